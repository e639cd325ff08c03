"""Per-kernel resource table from the code objects (VERDICT r1 #8a): VGPRs, SGPRs, spills, static LDS, and the waves per SIMD those
allow on gfx950 (512 VGPRs per SIMD lane in granules of 8: MI355X_MICROARCH.md "Register files"; 160 KiB LDS per CU).
Compiles every csrc/*.hip with the Makefile's flags to assembly (no GPU needed) and parses the .amdhsa metadata.
    python tools/kernel_resources.py > profiles/r02_kernel_resources.json"""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gs-sr_amd", "csrc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "--cuda-device-only", "-S"]
PRE = ["-ffp-contract=off"]
BLEND = ["-ffp-contract=fast", "-fno-slp-vectorize"]
FLAGS = {"gsr_preprocess.hip": PRE, "gsr_extra.hip": PRE, "gsr_mvloss.hip": PRE, "gsr_tsdf_sparse.hip": PRE, "gsr_blend.hip": BLEND, "gsr_blend_sp.hip": BLEND}


def demangle(names):
    if not names:
        return names
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=60).stdout.split("\n")
        return [o.split("(")[0].replace("void ", "") for o in out[:len(names)]]
    except Exception:
        return names


def main():
    table = {}
    d = tempfile.mkdtemp()
    for f in sorted(os.listdir(SRC)):
        if not f.endswith(".hip"):
            continue
        out = os.path.join(d, f + ".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + COMMON + FLAGS.get(f, []) + ["-o", out, os.path.join(SRC, f)], stderr=subprocess.DEVNULL)
        txt = open(out).read()
        meta = txt[txt.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in txt else ""
        recs = re.split(r"\n  - ", meta)[1:]
        rows = []
        for r in recs:
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, r) or [None, None])[1]
            name = g("name")
            if not name:
                continue
            vg, sg, sp, lds = int(g("vgpr_count") or 0), int(g("sgpr_count") or 0), int(g("vgpr_spill_count") or 0), int(g("group_segment_fixed_size") or 0)
            ag = int(g("agpr_count") or 0)
            alloc = -(-(vg + ag) // 8) * 8
            rows.append((name, dict(vgpr=vg, agpr=ag, sgpr=sg, vgpr_spills=sp, lds_bytes=lds, max_threads=int(g("max_flat_workgroup_size") or 0),
                                    waves_per_simd_by_vgpr=min(8, 512 // max(alloc, 8)),
                                    workgroups_per_cu_by_lds=(163840 // lds) if lds else None)))
        for (n, row), dn in zip(rows, demangle([n for n, _ in rows])):
            table[f + ":" + dn] = row
    json.dump(table, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
