"""Per-rank frame time of the N-GPU weak-scaling shard shapes (rows r, r+N, ... at 64 N spp) on ONE GPU: what a rank of bench.py --gpus N\nrenders per step, without the gather.  Usage: python tools/gpu_shard_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE
from pathtracer_cuda_interactive_amd import device as dev
from pathtracer_cuda_interactive_amd import distributed as D
for name, (w, h, spp) in {"cbox": (640, 480, 64), "bunny": (640, 480, 64)}.items():
    hs = HostScene.load(os.path.join(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes"), name + ".pts"))
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    for world in (1, 2, 4, 8):
        p = hs.render_params(w, h, spp * world)
        ts = []
        for rank in sorted({0, world - 1}):
            q = D.shard_params(p, rank, world)
            out = torch.empty(q.num_rows(), w, 3, dtype=torch.float32, device="cuda")
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20):
                    ds.render_into(q, out.data_ptr())
                torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20 * 1e3
            ts.append(t)
        print(f"{name}: world {world}: per-rank frame {' / '.join(f'{t:.3f}' for t in ts)} ms (ranks 0 and {world - 1})", flush=True)
    ds.close()
