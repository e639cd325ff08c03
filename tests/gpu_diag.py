"""First-contact diagnostic for the GPU box: stage-by-stage comparison of the HIP path against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gs-sr_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import oracle, scenes, hiprun


def cmp(name, a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b); ref = np.abs(b).max()
    print("    %-18s maxabs %.3e  refmax %.3e  rel %.2e  frac>1e-4: %.2e" % (name, err.max(), ref, err.max() / (ref + 1e-30), (err > 1e-4).mean()))


def main():
    print(torch.cuda.get_device_name(0))
    for variant, cm, P, W, H in [("ewa", "precomp", 2000, 128, 96), ("ewa", "sh", 3000, 200, 120), ("plane", "precomp", 2000, 128, 96),
                                 ("surfel", "precomp", 2000, 128, 96), ("surfel", "sh", 5000, 330, 190)]:
        sc = scenes.make_scene(variant, P, W, H, seed=1, color_mode=cm, bg=(0.2, 0.4, 0.6), pose=1)
        og = scenes.random_out_grads(variant, W, H, seed=1, scale=1.0)
        print(f"== {variant} {cm} P={P} {W}x{H}")
        with oracle.Forward(sc, variant) as f:
            g = f.backward(**og)
            st = hiprun.run_raw(variant, sc)
            print("    R", st["R"], "oracle", f.R)
            print("    radii equal", np.array_equal(st["radii"], f.radii), " tiles_touched equal", np.array_equal(st["tiles_touched"], f.tiles_touched()))
            if st["R"] == f.R:
                print("    point_list equal", np.array_equal(st["point_list"], f.point_list()),
                      " tile_keys equal", np.array_equal(st["tile_keys"], (f.keys() >> np.uint64(32)).astype(np.uint32)))
                rr = f.ranges(); mine = st["ranges"]
                touched = rr[:, 1] > rr[:, 0]
                print("    ranges equal (touched tiles)", np.array_equal(mine[touched], rr[touched]), " untouched empty", bool(np.all(mine[~touched, 0] == mine[~touched, 1])))
            ft, nc = f.image_state()
            cmp("color", st["color"], f.color)
            cmp("final_T", st["final_T"], ft)
            print("    n_contrib equal frac", (st["n_contrib"] == nc).mean())
            if variant == "surfel":
                for ch in range(11): cmp(f"others[{ch}]", st["others"][ch], f.others[ch])
            if variant == "plane":
                cmp("out_all_map", st["all_map"], f.out_all_map); cmp("plane_depth", st["plane_depth"], f.plane_depth)
                print("    observe equal", np.array_equal(st["observe"], f.observe), np.abs(st["observe"] - f.observe).max())
            res = hiprun.run(variant, sc, og)
            cmp("color(api)", res["color"], f.color)
            gg = res["grads"]
            pairs = [("dL_dmeans3D", "dL_dmeans3D"), ("dL_dscales", "dL_dscales"), ("dL_drotations", "dL_drotations"), ("dL_dopacities", "dL_dopacity"), ("dL_dmeans2D", "dL_dmeans2D")]
            pairs.append(("dL_dshs", "dL_dsh") if cm == "sh" else ("dL_dcolors_precomp", "dL_dcolors"))
            if variant == "plane": pairs += [("dL_dall_map", "dL_dall_map"), ("dL_dmeans2D_abs", "dL_dmeans2D_abs")]
            for a, b in pairs:
                cmp(a, gg[a].reshape(g[b].shape), g[b])
    # timing at full size
    for variant in ["ewa", "surfel", "plane"]:
        sc = scenes.make_scene(variant, 300000, 1920, 1080, seed=0)
        og = scenes.random_out_grads(variant, 1920, 1080, seed=0)
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            res = hiprun.run(variant, sc, og)
            torch.cuda.synchronize(); t1 = time.time()
        print(f"full-size {variant}: fwd+bwd incl. host copies {1e3 * (t1 - t0):.1f} ms; color mean {res['color'].mean():.4f}")


if __name__ == "__main__":
    main()
