"""Kernel-variant sweep on the GPU box: parity check (bit-exact vs oracle on a small frame) + interleaved
timing rounds of the headline configs (median/min of HIP-event kernel times; §5.4 rule 24: one process).
Usage: python tests/tools/gpu_tune.py [scene ...]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob  # noqa: E402
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE, PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

SC = os.path.join(REPO, "tests", "golden", "scenes")
CONFIGS = {"cbox": (640, 480, 64), "bunny": (640, 480, 64), "scene1": (640, 480, 16), "teapot": (640, 480, 16)}


def variants():
    yield ("v1", {"kernel": 1, "xcd_regions": 0, "octants": 1, "specialize": 1})
    yield ("v2 auto generic", {"kernel": 2, "v2_thresh": 0, "v2_inner": 0, "v2_minw": 0, "xcd_regions": 0, "octants": 1, "specialize": 0})
    yield ("v2 auto no-oct", {"kernel": 2, "v2_thresh": 0, "v2_inner": 0, "v2_minw": 0, "xcd_regions": 0, "octants": 0, "specialize": 1})
    yield ("v2 auto 1queue", {"kernel": 2, "v2_thresh": 0, "v2_inner": 0, "v2_minw": 0, "xcd_regions": 1, "octants": 1, "specialize": 1})
    for (t, i, w) in ((0, 0, 0), (40, -6, 6), (32, 4, 6), (40, 4, 6), (40, 3, 6), (40, 162, 6)):
        yield (f"v2 T{t} I{i} W{w}", {"kernel": 2, "v2_thresh": t, "v2_inner": i, "v2_minw": w, "xcd_regions": 0, "octants": 1, "specialize": 1})


def main():
    scenes = [a for a in sys.argv[1:] if not a.startswith("-")] or ["cbox", "bunny"]
    rounds = 5
    for name in scenes:
        hs = HostScene.load(os.path.join(SC, name + ".pts"))
        d = hs.finalize(PT_BVH_SORT_REFERENCE)
        ds = dev.DeviceScene(d)
        small = hs.render_params(64, 48, 4, seed=5)
        want, _ = ob.render(d, small)
        full = hs.render_params(*CONFIGS[name])
        vs = list(variants())
        ok = {}
        for label, opts in vs:
            for k, v in opts.items():
                ds.set_option(k, v)
            img = ds.render(small, traversal=PT_TRAVERSAL_EXACT)
            ok[label] = bool((img.view(np.uint32) == want.view(np.uint32)).all())
        for trav, tn in ((PT_TRAVERSAL_EXACT, "exact"), (PT_TRAVERSAL_PRUNED, "pruned")):
            times = {label: [] for label, _ in vs}
            regs = {}
            segs = 0
            for r in range(rounds):
                for label, opts in vs:
                    for k, v in opts.items():
                        ds.set_option(k, v)
                    ds.render(full, traversal=trav)
                    c = ds.counters()
                    times[label].append(c.kernel_ms)
                    segs = c.segments
                    regs[label] = (ds.info("vgprs_pruned" if trav == PT_TRAVERSAL_PRUNED else "vgprs"), ds.info("occupancy"))
            for label, _ in vs:
                t = np.array(times[label][1:])
                print(f"{name:7s} {tn:6s} {label:15s} parity={'OK ' if ok[label] else 'BAD'} vgpr={regs[label][0]:3d} occ={regs[label][1]} "
                      f"median {np.median(t):8.3f} ms  min {t.min():8.3f} ms  -> {segs / np.median(t) / 1e3:9.1f} Msamples/s", flush=True)
        ds.close()


if __name__ == "__main__":
    main()
