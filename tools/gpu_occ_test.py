"""Kernel time against resident blocks per CU.  Usage: python tools/gpu_occ_test.py [scene:spp ...]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE
from pathtracer_cuda_interactive_amd import device as dev
for spec in sys.argv[1:] or ["scene1:16", "cbox:64", "teapot:16", "bunny:64"]:
    name, _, spp = spec.partition(":")
    hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", name + ".pts"))
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    p = hs.render_params(640, 480, int(spp or 64))
    res = {}
    for rnd in range(3):
        for bpc in (0, 2, 3, 4, 5, 6, 7, 8):
            ds.set_option("blocks_per_cu", bpc)
            ts = []
            for r in range(5):
                ds.render(p); ts.append(ds.counters().kernel_ms)
            res.setdefault(bpc, []).append(np.median(ts[1:]))
            occ, grid = ds.info("occupancy"), ds.info("grid")
            res.setdefault(("g", bpc), grid)
    print(f"{spec} occ {occ} vgpr {ds.info('vgprs')}: " + "  ".join(f"bpc{b} (grid {res[('g', b)]}) {np.median(res[b]):.3f}" for b in (0, 2, 3, 4, 5, 6, 7, 8)) + " ms", flush=True)
    ds.close()
