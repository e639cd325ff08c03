import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gs-sr_amd")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, scenes, hiprun, gsrast
sc = scenes.make_scene("surfel", 300000, 1920, 1080, seed=0, color_mode="precomp")
og = scenes.random_out_grads("surfel", 1920, 1080, seed=0, scale=1.0)
L = gsrast.lib()
out = (C.c_ulonglong * 8)()
L.gsr_debug_counters(out)
res = hiprun.run("surfel", sc, og, device="cuda:0")
torch.cuda.synchronize()
L.gsr_debug_counters(out)
wp, ok_l, none_ok, act = out[0], out[1], out[2], out[3]
print("wave-pairs after cull", wp, "with no contributing lane", none_ok, f"({100*none_ok/wp:.1f}%)", "avg ok lanes per processed pair", ok_l/max(1,wp-none_ok), "avg active lanes", act/wp)
