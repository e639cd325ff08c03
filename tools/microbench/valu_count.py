"""One launch with a KNOWN wave-level VALU instruction count, to calibrate rocprofv3's SQ_INSTS_VALU on this device."""
import ctypes, os, torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvalu_peak.so"))
L.valu_peak_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(256, device="cuda")
blocks, iters = 4096, 2000
L.valu_peak_run(0, blocks, iters, out.data_ptr(), None); torch.cuda.synchronize()
print("expected_wave_valu_insts", blocks * 4 * iters * 16)
