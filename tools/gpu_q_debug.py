"""One small render on trace_kernel_q (option kernel = 3) against the oracle, with the kernel's own diagnostics: which error bits
(if any), how many floats differ, the schedule counters.  For debugging the regrouping kernel on the GPU box."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob  # noqa: E402
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cbox"
w, h, spp = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (64, 48, 4)))
opts = dict(kv.split("=") for kv in sys.argv[5:])
hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", name + ".pts"))
d = hs.finalize(PT_BVH_SORT_REFERENCE)
p = hs.render_params(w, h, spp, seed=5)
want, cnt = ob.render(d, p)
ds = dev.DeviceScene(d)
ds.set_option("kernel", 3)
ds.set_option("stats", 1)
for k, v in opts.items():
    ds.set_option(k, int(v))
try:
    img = ds.render(p)
    print("render ok, kernel", ds.info("kernel"), "block", ds.info("block_threads"), "grid", ds.info("grid"), "lds", ds.info("lds_bytes"))
    bad = img.view(np.uint32) != want.view(np.uint32)
    print("differing floats:", int(bad.sum()), "of", bad.size, " max abs diff", float(np.abs(img - want).max()))
    c = ds.counters()
    print("paths", c.paths, cnt.paths, "segments", c.segments, cnt.segments, "kernel_ms", c.kernel_ms)
    print("qdiag", [ds.info("qdiag%d" % k) for k in range(11)])
except Exception as e:          # noqa: BLE001
    print("FAILED:", e)
finally:
    ds.close()
