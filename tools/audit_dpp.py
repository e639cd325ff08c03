"""Build-time audit of hand-written DPP instructions (gs-sr_amd/csrc/gsr_blend_sp.hip).

The fused `v_*_dpp ... row_newbcast` forms are inline assembly, so hipcc does not insert the two wait states gfx9 needs between a
VALU write of a VGPR and a DPP read of it.  The kernel only feeds them long-lived constants; this script proves it on the generated
ISA: for every *_dpp instruction of the splat-parallel kernels it walks back over the preceding instructions until two wait states
have passed (one per instruction, N+1 per `s_nop N`) and fails if one of them is a VALU instruction whose destination overlaps the
DPP source register.  Usage: python tools/audit_dpp.py [file.s]  (without an argument it compiles gsr_blend_sp.hip itself)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(path):
    bad, ndpp = [], 0
    kernel, hist = None, []
    for ln in open(path):
        t = ln.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kernel, hist = m.group(1), []
            continue
        if not t or t[0] in ".;" or t.endswith(":") or kernel is None or "k_blend_bwd_sp" not in kernel:
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        ops = t.split(None, 1)
        mn = ops[0]
        args = [a.strip() for a in ops[1].split(",")] if len(ops) > 1 else []
        if "_dpp" in mn and len(args) >= 2:
            ndpp += 1
            src = regs(args[1].split()[0])
            ws = 0
            for pm, pa in reversed(hist):
                if ws >= 2:
                    break
                if pm.startswith("v_") and not pm.startswith("v_cmp") and pa and regs(pa[0].split()[0]) & src:
                    bad.append((kernel, t, pm + " " + ", ".join(pa)))
                ws += (int(pa[0], 0) + 1) if (pm == "s_nop" and pa) else 1
        hist.append((mn, args))
        if len(hist) > 8:
            hist.pop(0)
    return ndpp, bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        d = tempfile.mkdtemp()
        path = os.path.join(d, "sp.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=fast",
                               "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", path,
                               os.path.join(ROOT, "gs-sr_amd", "csrc", "gsr_blend_sp.hip")], stderr=subprocess.DEVNULL)
    n, bad = audit(path)
    print(f"audit_dpp: {n} DPP instructions checked, {len(bad)} hazards")
    for k, a, b in bad[:20]:
        print("  ", k[-30:], "|", b, "->", a)
    sys.exit(1 if bad else 0)
