import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE
from pathtracer_cuda_interactive_amd import distributed as D
hs = HostScene.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", "cbox.pts"))
desc = hs.finalize(PT_BVH_SORT_REFERENCE)
params = hs.render_params(640, 480, 64)
R = D.ShardedRenderer(desc)
for _ in range(5):
    R.render(params, 0, 1); R.scene.counters()
N = 100
ta = tb = 0.0
km = []
torch.cuda.synchronize()
T0 = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter()
    f = R.render(params, 0, 1)
    t1 = time.perf_counter()
    c = R.scene.counters()
    t2 = time.perf_counter()
    ta += t1 - t0; tb += t2 - t1; km.append(c.kernel_ms + c.resolve_ms)
torch.cuda.synchronize()
T1 = time.perf_counter()
print(f"render() {ta / N * 1e6:.1f} us, counters() {tb / N * 1e6:.1f} us, loop {(T1 - T0) / N * 1e6:.1f} us per step, kernels {np.mean(km) * 1e3:.1f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    R.render(params, 0, 1)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
