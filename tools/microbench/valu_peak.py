"""Measures wave64 VALU instructions per second (non-packed fp32) on the device: python tools/microbench/valu_peak.py -> one JSON line."""
import ctypes, json, os, time
import torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvalu_peak.so"))
L.valu_peak_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(256, device="cuda")
res = {}
for mode, name in ((0, "v_fma_f32"), (1, "v_mul_f32"), (3, "v_fma_f32, 16 of 64 lanes active"), (4, "v_fma_f32, 32 of 64 lanes active")):
    best = 0.0
    for blocks in (256 * 8, 256 * 16, 256 * 32):
        iters = 20000
        L.valu_peak_run(mode, blocks, 100, out.data_ptr(), None); torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.valu_peak_run(mode, blocks, iters, out.data_ptr(), None); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rate = blocks * 4 * iters * 16 / dt
        best = max(best, rate)
    res[name] = round(best / 1e9, 1)
print(json.dumps({"wave64_valu_Ginst_per_s": res, "device": torch.cuda.get_device_name(0)}))
