"""Frame time of the two big stand-in configs against the per-sample scratch budget (number of sample passes).
Usage: python tools/gpu_scratch_sweep.py"""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import PT_BVH_SORT_REFERENCE, standins
from pathtracer_cuda_interactive_amd import device as dev
sc = os.path.join(REPO, "tests", "golden", "scenes")
for name, (w, h, spp) in {"buddha_standin": (1280, 960, 256), "dragon_standin": (1920, 1080, 1024)}.items():
    hs = standins.BUILDERS[name](sc)
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    p = hs.render_params(w, h, spp)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    for gib in (1, 2, 4, 8, 16):
        ds.set_option("scratch_bytes", gib << 30)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ds.render_into(p, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        c = ds.counters()
        print(f"{name} scratch {gib:2d} GiB: passes {ds.info('passes')}, frame {np.median(ts[1:]) * 1e3:8.2f} ms (kernels {c.kernel_ms:.2f} + resolve {c.resolve_ms:.2f})", flush=True)
    ds.close()
