"""Small (interactive) launches: kernel time against blocks per CU.  Usage: python tools/gpu_small_launch.py"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

SC = os.path.join(REPO, "tests", "golden", "scenes")
for name in sys.argv[1:] or ["cbox", "scene1", "bunny"]:
    hs = HostScene.load(os.path.join(SC, name + ".pts"))
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    for spp in (1, 2, 4, 8):
        p = hs.render_params(640, 480, spp)
        row = []
        for bpc in (1, 2, 3, 4, 5, 6, 0):
            ds.set_option("blocks_per_cu", bpc)
            ts = []
            for _ in range(9):
                ds.render(p)
                ts.append(ds.counters().kernel_ms)
            row.append(f"bpc{bpc}(grid {ds.info('grid')}) {np.median(ts[2:]) * 1e3:7.1f}")
        print(f"{name} spp {spp}: " + "  ".join(row) + "  us", flush=True)
    ds.close()
