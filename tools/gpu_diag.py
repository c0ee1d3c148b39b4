"""Schedule diagnostics of trace_kernel_v2 (STATS build): where do the lane slots go?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE
from pathtracer_cuda_interactive_amd import device as dev
for name, (w, h, spp) in {"cbox": (640, 480, 64), "bunny": (640, 480, 64)}.items():
    hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", name + ".pts"))
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    ds.set_option("stats", 1)
    ds.render(hs.render_params(w, h, spp))
    c = ds.counters()
    it, sch, schl, ni, nil, nl, nll, wait = (ds.info(f"diag{k}") for k in range(8))
    steps = ni + nl
    print(f"{name}: kernel {c.kernel_ms:.2f} ms; wave iterations {it/1e6:.2f} M; scheduler phases {sch/1e6:.2f} M serving {schl/max(sch,1):.1f} lanes each "
          f"({c.segments/max(sch,1):.1f} segments per phase)")
    print(f"   inner steps {ni/1e6:.2f} M with {nil/max(ni,1):.1f} active lanes; leaf steps {nl/1e6:.2f} M with {nll/max(nl,1):.1f} active lanes; "
          f"waiting lanes per traversal step {wait/max(steps,1):.1f}; traversal-step lane utilisation {(nil+nll)/max(steps,1)/64:.3f}")
    ds.close()
