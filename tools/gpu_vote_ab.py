"""Vote burst against fixed bursts on LDS-resident scenes with spheres (interleaved rounds in one process).
Usage: python tools/gpu_vote_ab.py"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE
from pathtracer_cuda_interactive_amd import device as dev
for path in ("tests/data/mixed.xml", "tests/golden/scenes/scene1_phong.pts", "tests/golden/scenes/scene1.pts"):
    hs = HostScene.load(os.path.join(REPO, path))
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    p = hs.render_params(640, 480, 16)
    res = {}
    for rnd in range(4):
        for label, (t, i) in {"vote-6": (40, -6), "6+2": (40, 162), "3+1": (40, 3), "4+1": (40, 4)}.items():
            ds.set_option("v2_thresh", t); ds.set_option("v2_inner", i)
            ts = []
            for _ in range(5):
                ds.render(p); ts.append(ds.counters().kernel_ms)
            res.setdefault(label, []).append(np.median(ts[1:]))
    print(path, "residency", ds.info("residency"), " ".join(f"{k}: {np.median(v):.3f} ms" for k, v in res.items()), flush=True)
    ds.close()
