"""Re-wrap a markdown file at 118 columns (lists, paragraphs; tables, headings and code blocks verbatim):  python tools/wrap_md.py DESIGN.md > out.md"""
import re, textwrap, sys
W = 118
def wrap_md(text):
    out = []
    blocks = re.split(r"\n\s*\n", text.strip("\n"))
    in_code = False
    for b in blocks:
        lines = b.split("\n")
        if any(l.startswith("```") for l in lines) or in_code:
            # code block(s): keep verbatim (track fences)
            for l in lines:
                if l.startswith("```"):
                    in_code = not in_code
            out.append(b); continue
        if lines[0].startswith("#") and len(lines) == 1:
            out.append(b); continue
        if lines[0].startswith("|"):
            out.append(b); continue
        # split a block into items: a new item starts at a line beginning with list markers
        items, cur = [], []
        for l in lines:
            if re.match(r"^\s*([*-]|\d+[.)]|\(\d+\)|[a-z]\))\s", l) or l.startswith("#"):
                if cur: items.append(cur)
                cur = [l]
            else:
                cur.append(l)
        if cur: items.append(cur)
        res = []
        for it in items:
            first = it[0]
            if first.startswith("#"):
                res.append(first)
                rest = " ".join(x.strip() for x in it[1:]).strip()
                if rest: res.append(textwrap.fill(rest, W))
                continue
            m = re.match(r"^(\s*)(([*-]|\d+[.)]|\(\d+\)|[a-z]\))\s+)?", first)
            lead = m.group(1) or ""; mark = m.group(2) or ""
            body = " ".join([first[len(lead) + len(mark):].strip()] + [x.strip() for x in it[1:]])
            res.append(textwrap.fill(body, W, initial_indent=lead + mark, subsequent_indent=lead + " " * len(mark), break_long_words=False, break_on_hyphens=False))
        out.append("\n".join(res))
    return "\n\n".join(out) + "\n"
if __name__ == "__main__":
    sys.stdout.write(wrap_md(open(sys.argv[1]).read()))
