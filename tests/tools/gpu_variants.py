"""Option-set A/B on the GPU box, one process: every variant is a set of pt_scene_set_option values on the same library.
Per scene: bit-exact parity of every variant against the oracle on a small frame, then interleaved timing rounds of the
full configuration (HIP-event kernel time), then one STATS render per variant for the schedule diagnostics.
Usage: python tests/tools/gpu_variants.py --scenes bunny,buddha_standin@16 [--rounds 5] label:k=v,k=v [label:...] ..."""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob  # noqa: E402
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE, standins  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

SC = os.path.join(REPO, "tests", "golden", "scenes")
CONFIGS = {"cbox": (640, 480, 64), "bunny": (640, 480, 64), "scene1": (640, 480, 16), "teapot": (640, 480, 16), "scene4": (640, 480, 32),
           "scene1_phong": (640, 480, 16), "buddha_standin": (1280, 960, 256), "dragon_standin": (1920, 1080, 1024)}
RESET = {"kernel": 2, "q_target": 0, "q_swap": 0, "q_low": 0, "v2_thresh": 0, "v2_inner": 0, "v2_minw": 0, "blocks_per_cu": 0, "stats": 0, "lds_budget_kb": 0, "chunk": 0, "fast_tree": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="bunny")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    variants = []
    for v in a.variants:
        label, _, rest = v.partition(":")
        variants.append((label, {k: int(x) for k, x in (kv.split("=") for kv in rest.split(",") if kv)}))
    for spec in a.scenes.split(","):
        name, _, spp = spec.partition("@")
        hs = standins.BUILDERS[name](SC) if name in standins.BUILDERS else HostScene.load(os.path.join(SC, name + ".pts"))
        d = hs.finalize(PT_BVH_SORT_REFERENCE)
        ds = dev.DeviceScene(d)
        small = hs.render_params(64, 48, 4, seed=5)
        want, _ = ob.render(d, small)
        w, h, spp0 = CONFIGS[name]
        full = hs.render_params(w, h, int(spp) if spp else spp0)

        def apply(opts):
            for k, v in {**RESET, **opts}.items():
                ds.set_option(k, v)
        ok, times, info = {}, {l: [] for l, _ in variants}, {}
        for label, opts in variants:
            apply(opts)
            try:
                ok[label] = bool((ds.render(small).view(np.uint32) == want.view(np.uint32)).all())
            except Exception as e:                                 # variant not compiled in
                ok[label] = None
                print(f"{spec:18s} {label:22s} unavailable: {e}", flush=True)
        live = [(l, o) for l, o in variants if ok[l] is not None]
        segs = 0
        for r in range(a.rounds + 1):
            for label, opts in live:
                apply(opts)
                ds.render(full)
                c = ds.counters()
                if r:
                    times[label].append(c.kernel_ms)
                segs = c.segments
                info[label] = (ds.info("vgprs"), ds.info("blocks_per_cu"), ds.info("lds_bytes"), ds.info("top_nodes"))
        base = None
        for label, opts in live:
            t = np.array(times[label])
            med = float(np.median(t))
            base = base or med
            line = (f"{spec:18s} {label:22s} parity={'OK ' if ok[label] else 'BAD'} vgpr={info[label][0]:3d} bpc={info[label][1]} lds={info[label][2]:6d} "
                    f"top={info[label][3]:4d} median {med:9.3f} ms  min {t.min():9.3f} ms  {segs / med / 1e3:9.1f} Msamples/s  {100 * (med / base - 1):+6.1f} %")
            if not a.no_stats:
                apply({**opts, "stats": 1})
                ds.render(full)
                c = ds.counters()
                dg = [ds.info(f"diag{k}") for k in range(8)]
                # [0] iterations [1] scheduler phases [2] lanes served [3] inner steps [4] lanes in them [5] leaf steps [6] lanes in them [7] idle lane-slots
                line += (f" | inner steps {dg[3] / 1e6:7.1f} M x {dg[4] / max(dg[3], 1):4.1f} lanes, leaf steps {dg[5] / 1e6:6.1f} M x {dg[6] / max(dg[5], 1):4.1f} lanes,"
                         f" sched {dg[1] / 1e6:6.1f} M x {dg[2] / max(dg[1], 1):4.1f} lanes, iters {dg[0] / 1e6:6.1f} M,"
                         f" visits {c.node_visits / max(c.segments, 1):5.2f} + {c.leaf_tests / max(c.segments, 1):4.2f} per segment, rerun {ds.info('redo_segments')}")
            print(line, flush=True)
        ds.close()


if __name__ == "__main__":
    main()
