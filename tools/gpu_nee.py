"""Next-event estimation (PT_RENDER_NEE) against the reference estimator on the GPU: frame time at equal samples, rmse
against a converged image at equal samples, and the time to equal rmse.  Runs on the GPU box.
Usage: python tools/gpu_nee.py [cbox scene1_phong ...]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import PT_BVH_SORT_REFERENCE, PT_RENDER_NEE, HostScene, standins  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

SC = os.path.join(REPO, "tests", "golden", "scenes")
CONFIGS = {"cbox": (640, 480, 64), "scene1_phong": (640, 480, 64), "bunny": (640, 480, 64), "dragon_standin": (480, 270, 64)}


def main():
    for name in sys.argv[1:] or ["cbox", "scene1_phong", "dragon_standin"]:
        hs = standins.BUILDERS[name](SC) if name in standins.BUILDERS else HostScene.load(os.path.join(SC, name + ".pts"))
        d = hs.finalize(PT_BVH_SORT_REFERENCE)
        w, h, spp = CONFIGS[name]
        ds = dev.DeviceScene(d)
        ref_p = hs.render_params(w, h, 8192, seed=99)
        ref_q = ref_p.copy(); ref_q.flags = PT_RENDER_NEE
        ref = 0.5 * (ds.render(ref_p).astype(np.float64) + ds.render(ref_q))        # both estimators converge to the same image
        rel = float(np.abs(ds.render(ref_p).astype(np.float64).mean() - ds.render(ref_q).astype(np.float64).mean()) / ref.mean())
        out = {}
        for label, flags in (("reference estimator", 0), ("NEE", PT_RENDER_NEE)):
            p = hs.render_params(w, h, spp)
            p.flags = flags
            img = ds.render(p)
            ts = []
            for _ in range(5):
                ds.render(p)
                ts.append(ds.counters().kernel_ms)
            c = ds.counters()
            out[label] = (float(np.median(ts)), float(np.sqrt(((img - ref) ** 2).mean())), c.segments)
        (t0, e0, s0), (t1, e1, s1) = out["reference estimator"], out["NEE"]
        # rmse ~ 1/sqrt(spp): samples (and time) the reference estimator needs to reach NEE's rmse
        gain = (e0 / e1) ** 2 * t0 / t1
        print(f"{name:15s} {w}x{h}x{spp}: reference estimator {t0:8.3f} ms rmse {e0:.4f} | NEE {t1:8.3f} ms rmse {e1:.4f} | "
              f"time x{t1 / t0:.2f}, rmse x{e1 / e0:.2f}, equal-rmse speed-up x{gain:.2f}; converged means differ by {100 * rel:.3f} %", flush=True)
        ds.close()


if __name__ == "__main__":
    main()
