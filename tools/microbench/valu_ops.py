"""Per-opcode wave64 VALU issue rates (G instructions/s, whole device): python tools/microbench/valu_ops.py -> one JSON line."""
import ctypes, json, os, time
import torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvalu_ops.so"))
L.valu_ops_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(256, device="cuda")
names = ["v_fma_f32", "v_add_f32", "v_mov_b32", "v_exp_f32", "v_rcp_f32", "v_add_f32_dpp(quad_perm)", "v_cmp_lt_f32", "v_cndmask_b32(vcc)",
         "v_fma_f32 + v_exp_f32 interleaved (both counted)"]
res = {}
for mode, name in enumerate(names):
    best = 0.0
    for blocks in (256 * 8, 256 * 32):
        iters = 4000
        L.valu_ops_run(mode, blocks, 50, out.data_ptr(), None); torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.valu_ops_run(mode, blocks, iters, out.data_ptr(), None); torch.cuda.synchronize()
        best = max(best, blocks * 4 * iters * 16 / (time.perf_counter() - t0))
    res[name] = round(best / 1e9, 1)
print(json.dumps({"wave64_Ginst_per_s": res}))
