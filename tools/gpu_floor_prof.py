"""Launch floor under rocprofv3 --kernel-trace: a fixed sequence of small launches whose GPU-side durations (kernel trace
begin/end) are compared with the HIP-event times the library reports.  Run as
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/gpu_floor_prof.py
then  python3 tools/gpu_floor_prof.py --parse OUT  prints one line per configuration."""
import csv
import glob
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = [("cbox", 64, 48, 1, 1), ("cbox", 640, 480, 1, 1), ("cbox", 640, 480, 1, 8), ("cbox", 640, 480, 1, 50), ("cbox", 640, 480, 2, 50),
           ("scene1", 640, 480, 1, 1), ("scene1", 640, 480, 2, 50), ("bunny", 640, 480, 1, 1), ("bunny", 640, 480, 2, 50)]
REPS = 6

if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tr = [r for r in rows if "trace_kernel" in r["Kernel_Name"]]
    rs = [r for r in rows if "resolve_kernel" in r["Kernel_Name"]]
    assert len(tr) == len(CONFIGS) * REPS, (len(tr), len(CONFIGS) * REPS)
    for k, cfg in enumerate(CONFIGS):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr[k * REPS + 1:(k + 1) * REPS]]
        g = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(tr[k * REPS + 1:(k + 1) * REPS], rs[k * REPS + 1:(k + 1) * REPS])]
        print(f"{cfg}: trace kernel {np.median(d):8.1f} us (min {min(d):8.1f}); gap to resolve start {np.median(g):6.1f} us")
    sys.exit(0)

sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

SC = os.path.join(REPO, "tests", "golden", "scenes")
scenes = {}
for name, w, h, spp, md in CONFIGS:
    if name not in scenes:
        hs = HostScene.load(os.path.join(SC, name + ".pts"))
        scenes[name] = (hs, dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE)))
    hs, ds = scenes[name]
    p = hs.render_params(w, h, spp)
    p.max_depth = md
    ts = []
    for _ in range(REPS):
        ds.render(p)
        ts.append(ds.counters().kernel_ms)
    print(f"{(name, w, h, spp, md)}: HIP-event kernel time {np.median(ts[1:]) * 1e3:8.1f} us", flush=True)
