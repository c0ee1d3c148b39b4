"""Schedule diagnostics and launch timeline of trace_kernel_v2 (STATS build): where do the lane slots and the time go?
Usage: python tools/gpu_diag.py [spp ...]   (default 64 2)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE
from pathtracer_cuda_interactive_amd import device as dev
M64 = (1 << 64) - 1
spps = [int(a) for a in sys.argv[1:] if not a.startswith("md=")] or [64, 2]
max_depth = ([int(a[3:]) for a in sys.argv[1:] if a.startswith("md=")] or [50])[0]
for name in ("cbox", "bunny"):
    hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", name + ".pts"))
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    ds.set_option("stats", 1)
    for kv in filter(None, os.environ.get("PT_AB_OPTIONS", "").split(",")):
        ds.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for spp in spps:
        p = hs.render_params(640, 480, spp)
        p.max_depth = max_depth
        for _ in range(2):
            ds.render(p)
        c = ds.counters()
        it, sch, schl, ni, nil, nl, nll, wait = (ds.info(f"diag{k}") for k in range(8))
        steps = ni + nl
        print(f"{name} spp {spp}: kernel {c.kernel_ms:.3f} ms; wave iterations {it/1e6:.2f} M; scheduler phases {sch/1e6:.2f} M serving {schl/max(sch,1):.1f} lanes each "
              f"({c.segments/max(sch,1):.1f} segments per phase)")
        print(f"   inner steps {ni/1e6:.2f} M with {nil/max(ni,1):.1f} active lanes; leaf steps {nl/1e6:.2f} M with {nll/max(nl,1):.1f} active lanes; "
              f"waiting lanes per traversal step {wait/max(steps,1):.1f}; traversal-step lane utilisation {(nil+nll)/max(steps,1)/64:.3f}")
        t = [ds.info(f"diag{k}") & M64 for k in range(8, 16)]
        entry0, staged1, dry0, dry1, exit1, drain_sum, busy_sum, waves = (~t[0]) & M64, t[1], (~t[2]) & M64, t[3], t[4], t[5], t[6], t[7]
        us = lambda ticks: ticks / 100.0
        print(f"   timeline (us after the first wave entered): last wave staged {us(staged1 - entry0):.1f}; feed dry for the first wave {us(dry0 - entry0):.1f}, "
              f"for the last {us(dry1 - entry0):.1f}; last wave exits {us(exit1 - entry0):.1f}; {waves} waves: mean fed phase {us(busy_sum) / max(waves, 1):.1f}, "
              f"mean drain {us(drain_sum) / max(waves, 1):.1f}")
        dry = [ds.info(f"diag{16 + k}") for k in range(128)]
        ext = [ds.info(f"diag{144 + k}") for k in range(128)]
        last = max(k for k in range(128) if ext[k])
        first = min(k for k in range(128) if dry[k])
        print("   waves by 100-us bin after their entry (feed found dry / exit): " +
              "  ".join(f"{k / 10:.1f}ms {dry[k]}/{ext[k]}" for k in range(first, last + 1)) + f"   [bin width here: 100 us; {c.segments} segments]")
    ds.close()
