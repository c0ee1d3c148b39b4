"""Exploratory GPU run: device-vs-oracle parity on small renders + timings of the headline configs.
Usage (on the GPU box): python tools/gpu_explore.py [--quick]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob  # noqa: E402
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE, PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

SC = os.path.join(REPO, "tests", "golden", "scenes")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def main():
    quick = "--quick" in sys.argv
    rng = np.random.default_rng(1)
    x = (rng.random(200000, dtype=np.float32) * np.float32(6.2831855)).astype(np.float32)
    s_d, c_d = dev.debug_math(0, x)
    s_o, c_o = ob.sincos(x)
    print("sincos mismatches", int((bits(s_d) != bits(s_o)).sum()), int((bits(c_d) != bits(c_o)).sum()))
    xb = rng.random(200000, dtype=np.float32)
    yb = (rng.random(200000, dtype=np.float32) * 200).astype(np.float32)
    p_d, _ = dev.debug_math(1, xb, yb)
    p_o = ob.powf(xb, yb)
    print("pow mismatches", int((bits(p_d) != bits(p_o)).sum()))

    for name, (w, h, spp) in {"scene1": (64, 48, 8), "cbox": (64, 48, 8), "scene1_phong": (64, 48, 8), "teapot": (64, 48, 4), "bunny": (64, 48, 2)}.items():
        hs = HostScene.load(os.path.join(SC, name + ".pts"))
        d = hs.finalize(PT_BVH_SORT_REFERENCE)
        p = hs.render_params(w, h, spp)
        ref, cnt = ob.render(d, p)
        ds = dev.DeviceScene(d)
        for trav in (PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED):
            for fg in (0, 1):
                ds.set_option("force_global", fg)
                img = ds.render(p, traversal=trav)
                c = ds.counters()
                nm = int((bits(img) != bits(ref)).sum())
                print(f"{name} trav={trav} force_global={fg} lds_scene={ds.info('lds_scene')} mismatched floats={nm}/{img.size} "
                      f"maxabs={np.abs(img - ref).max():.3e} segs dev={c.segments} oracle={cnt.segments} paths={c.paths}")
        ds.close()

    for name, (w, h, spp) in {"cbox": (640, 480, 64), "bunny": (640, 480, 64), "scene1": (640, 480, 16)}.items():
        hs = HostScene.load(os.path.join(SC, name + ".pts"))
        d = hs.finalize(PT_BVH_SORT_REFERENCE)
        p = hs.render_params(w, h, spp)
        ds = dev.DeviceScene(d)
        print(name, "vgprs", ds.info("vgprs"), "scene_bytes", ds.info("scene_bytes"), "depth", ds.info("bvh_depth"))
        for trav in (PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED):
            for stats in (0, 1):
                ds.set_option("stats", stats)
                for rep in range(3):
                    img = ds.render(p, traversal=trav)
                    c = ds.counters()
                print(f"{name} {w}x{h} spp{spp} trav={trav} stats={stats}: kernel {c.kernel_ms:.3f} ms resolve {c.resolve_ms:.3f} ms "
                      f"segs {c.segments} -> {c.segments / c.kernel_ms / 1e3:.1f} Msamples/s; nodes/seg {c.node_visits / max(c.segments,1):.2f} "
                      f"leaves/seg {c.leaf_tests / max(c.segments,1):.2f} grid {ds.info('grid')} occ {ds.info('occupancy')} lds {ds.info('lds_bytes')} mean {img.mean():.6f}")
        if not quick and name == "cbox":
            for bpc in (1, 2, 4, 6, 8):
                ds.set_option("blocks_per_cu", bpc)
                ds.set_option("stats", 0)
                for rep in range(2):
                    ds.render(p, traversal=PT_TRAVERSAL_EXACT)
                    c = ds.counters()
                print(f"  blocks_per_cu={bpc}: {c.kernel_ms:.3f} ms -> {c.segments / c.kernel_ms / 1e3:.1f} Msamples/s")
        ds.close()


if __name__ == "__main__":
    main()
