"""Kernel time against samples per pixel: the intercept is the fixed floor of a launch (staging + ramp + the tail of the
longest path), the slope the steady-state rate.  Usage: python tools/gpu_floor.py [scene ...]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

SC = os.path.join(REPO, "tests", "golden", "scenes")
for name in sys.argv[1:] or ["cbox"]:
    hs = HostScene.load(os.path.join(SC, name + ".pts"))
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    xs, ys = [], []
    for md in (50, 8, 1):
        for spp in (1, 2, 4, 8, 16, 32, 64, 128):
            p = hs.render_params(640, 480, spp)
            p.max_depth = md
            ts, seg = [], 0
            for _ in range(7):
                ds.render(p)
                c = ds.counters()
                ts.append(c.kernel_ms); seg = c.segments
            t = float(np.median(ts[2:]))
            print(f"{name} max_depth {md:2d} spp {spp:3d}: kernel {t * 1e3:8.1f} us  {seg / t / 1e3:9.1f} Msamples/s  grid {ds.info('grid')}", flush=True)
            if md == 50:
                xs.append(spp); ys.append(t)
        if md == 50:
            a, b = np.polyfit(xs[3:], ys[3:], 1)
            print(f"{name}: fit over spp>=8: {a * 1e3:.1f} us per spp + {b * 1e3:.1f} us floor", flush=True)
    ds.close()
