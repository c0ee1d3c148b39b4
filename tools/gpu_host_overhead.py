"""Host-side cost of a frame around the kernels: launch + sync, and the pt_get_counters call.  Usage: python tools/gpu_host_overhead.py"""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE
from pathtracer_cuda_interactive_amd import device as dev
hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", "cbox.pts"))
ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
for spp in (64, 2):
    p = hs.render_params(640, 480, spp)
    out = torch.empty((480, 640, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    N = 200
    for _ in range(10):
        ds.render_into(p, out.data_ptr(), stream); torch.cuda.synchronize(); ds.counters()
    t_launch = t_sync = t_cnt = 0.0
    kern = []
    for _ in range(N):
        t0 = time.perf_counter()
        ds.render_into(p, out.data_ptr(), stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        c = ds.counters()
        t3 = time.perf_counter()
        t_launch += t1 - t0; t_sync += t2 - t1; t_cnt += t3 - t2
        kern.append(c.kernel_ms + c.resolve_ms)
    print(f"cbox spp {spp}: enqueue {t_launch / N * 1e6:.1f} us, wait {t_sync / N * 1e6:.1f} us, counters() {t_cnt / N * 1e6:.1f} us; "
          f"total {(t_launch + t_sync + t_cnt) / N * 1e6:.1f} us per frame; kernels (HIP events) {np.mean(kern) * 1e3:.1f} us", flush=True)
ds.close()
