import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE
from pathtracer_cuda_interactive_amd import device as dev
for name in ("bunny", "teapot", "cbox"):
    hs = HostScene.load(f"tests/golden/scenes/{name}.pts")
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    p = hs.render_params(640, 480, 64)
    for bpc in (0, 5, 6, 7, 8):
        ds.set_option("blocks_per_cu", bpc)
        ts = []
        for r in range(6):
            ds.render(p); ts.append(ds.counters().kernel_ms)
        print(name, "blocks_per_cu", bpc, "occ", ds.info("occupancy"), "vgpr", ds.info("vgprs"), "median %.3f ms" % np.median(ts[1:]), flush=True)
    ds.close()
