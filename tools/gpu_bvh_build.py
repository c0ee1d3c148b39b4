"""Device BVH build (pt_bvh_build_device) against the host's reference builder: build time, tree depth, inner visits per
segment under exact traversal (the kernel's own counters) and frame time.  Runs on the GPU box.
Usage: python tools/gpu_bvh_build.py [bunny teapot buddha_standin ...]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import PT_BVH_SORT_REFERENCE, HostScene, standins  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402

SC = os.path.join(REPO, "tests", "golden", "scenes")
CONFIGS = {"cbox": (640, 480, 64), "bunny": (640, 480, 64), "teapot": (640, 480, 16), "buddha_standin": (1280, 960, 32),
           "dragon_standin": (960, 540, 32)}


def main():
    for name in sys.argv[1:] or ["teapot", "bunny", "buddha_standin"]:
        hs = standins.BUILDERS[name](SC) if name in standins.BUILDERS else HostScene.load(os.path.join(SC, name + ".pts"))
        t0 = time.perf_counter()
        d = hs.finalize(PT_BVH_SORT_REFERENCE)
        host_s = time.perf_counter() - t0                      # flatten + primitive boxes + median-split build (scene_build.cpp)
        w, h, spp = CONFIGS[name]
        p = hs.render_params(w, h, spp)
        trees = [("host median split (reference)", d, {"depth": hs.bvh_depth, "build_ms": host_s * 1e3, "wall_ms": host_s * 1e3})]
        for label, m in (("device LBVH", dev.PT_BVH_DEVICE_LBVH), ("device SAH", dev.PT_BVH_DEVICE_SAH)):
            dev.build_bvh_device(d, m)                           # warm-up: code objects, allocator
            t0 = time.perf_counter()
            d2, info = dev.build_bvh_device(d, m)
            info["wall_ms"] = (time.perf_counter() - t0) * 1e3   # uploads + build + copy of the nodes back to the host
            trees.append((label, d2, info))
        t0 = time.perf_counter()
        d3, info3 = dev.build_bvh_sweep(d)                       # the internal tree's builder (host threads)
        info3["wall_ms"] = (time.perf_counter() - t0) * 1e3
        trees.append(("host full-sweep SAH (pt_bvh_build_sweep)", d3, info3))
        base_img = None
        for label, dd, info in trees:
            ds = dev.DeviceScene(dd)
            ds.set_option("fast_tree", 0)                        # the tree as handed in ...
            ds.set_option("stats", 1)
            img = ds.render(p)
            c = ds.counters()
            ds.set_option("stats", 0)
            ts = []
            for _ in range(5):
                ds.render(p)
                ts.append(ds.counters().kernel_ms)
            ds.set_option("fast_tree", 1)                        # ... and what pt_scene_create makes of it
            ti = []
            for _ in range(5):
                ds.render(p)
                ti.append(ds.counters().kernel_ms)
            internal = f"with the internal tree over it {np.median(ti):8.3f} ms" if ds.info("fast_tree_on") else "no internal tree kept"
            if base_img is None:
                base_img = img
            diff_px = int((np.abs(img - base_img).max(axis=2) > 0).sum())
            print(f"{name:15s} {label:42s} prims {dd.num_shapes:8d} depth {info['depth']:3d} build {info['build_ms']:9.2f} ms (wall {info['wall_ms']:9.1f} ms) "
                  f"inner visits/segment {c.node_visits / c.segments:6.2f} leaf tests/segment {c.leaf_tests / c.segments:5.2f} "
                  f"frame {np.median(ts):8.3f} ms ({internal})  lds {ds.info('lds_bytes'):6d} top {ds.info('top_nodes'):4d} residency {ds.info('residency')}  pixels differing from the reference tree: {diff_px}",
                  flush=True)
            ds.close()


if __name__ == "__main__":
    main()
