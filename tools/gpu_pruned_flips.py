"""How often does closest-t pruning change a pixel?  Renders full-size frames with PT_TRAVERSAL_EXACT and
PT_TRAVERSAL_PRUNED for several seeds and counts differing pixels."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE, PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

for name, (w, h, spp) in {"cbox": (640, 480, 64), "bunny": (640, 480, 64), "teapot": (640, 480, 64), "scene1": (640, 480, 64)}.items():
    hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", name + ".pts"))
    d = hs.finalize(PT_BVH_SORT_REFERENCE)
    ds = dev.DeviceScene(d)
    ds.set_option("stats", 1)
    tot_px = tot_seg = 0
    for seed in range(1984, 1984 + 8):
        p = hs.render_params(w, h, spp, seed=seed)
        a = ds.render(p, traversal=PT_TRAVERSAL_EXACT)
        ca = ds.counters()
        b = ds.render(p, traversal=PT_TRAVERSAL_PRUNED)
        cb = ds.counters()
        npx = int((np.abs(a - b).max(axis=2) > 0).sum())
        tot_px += npx
        tot_seg += ca.segments
        print(f"{name} seed {seed}: {npx} differing pixels of {w*h}; max |diff| {np.abs(a-b).max():.4f}; node visits {ca.node_visits/ca.segments:.2f} -> {cb.node_visits/cb.segments:.2f}, "
              f"leaf tests {ca.leaf_tests/ca.segments:.2f} -> {cb.leaf_tests/cb.segments:.2f}; kernel {ca.kernel_ms:.2f} -> {cb.kernel_ms:.2f} ms", flush=True)
    print(f"== {name}: {tot_px} differing pixels in {tot_seg/1e6:.0f} M segments -> one per {tot_seg/max(tot_px,1)/1e6:.1f} M segments")
    ds.close()
