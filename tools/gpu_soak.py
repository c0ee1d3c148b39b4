"""Soak test: many back-to-back frames over scenes / traversal modes / kernel variants / shard shapes; every frame must
be bit-identical to the first one of its configuration.  Catches rare scheduling-dependent bugs and hangs."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE, PT_RENDER_NEE, PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED
from pathtracer_cuda_interactive_amd import device as dev

rng = np.random.default_rng(0)
scenes = {}
for name in ("cbox", "scene1", "scene1_phong", "teapot", "bunny"):
    hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", name + ".pts"))
    scenes[name] = (hs, dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE)))
ref = {}
t0 = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for it in range(n):
    name = list(scenes)[it % len(scenes)]
    hs, ds = scenes[name]
    w, h, spp = [(160, 120, 4), (97, 61, 7), (320, 240, 2), (64, 64, 16)][(it // 5) % 4]
    kernel = [2, 1, 3, 2, 3][(it // 20) % 5]            # 3 = paths regrouped across waves (round 3): same frames
    trav = PT_TRAVERSAL_EXACT
    stride = [1, 3, 8][(it // 7) % 3]
    ds.set_option("kernel", kernel)
    ds.set_option("xcd_regions", (it // 11) % 2)
    ds.set_option("octants", (it // 13) % 2)
    ds.set_option("specialize", (it // 17) % 2)
    ds.set_option("top_cache", (it // 19) % 2)
    ds.set_option("item_order", (it // 23) % 2)
    ds.set_option("chunk", [0, 64, 128, 256][(it // 29) % 4])
    ds.set_option("lds_budget_kb", [0, 24, 39][(it // 31) % 3])
    ds.set_option("q_swap", [0, 4, 32][(it // 43) % 3])
    ds.set_option("q_target", [0, 200, 100000][(it // 47) % 3])
    ds.set_option("fast_tree", 0 if (it // 41) % 3 == 2 else 1)      # internal or caller's tree: the frame must not care
    nee = kernel == 2 and (it // 37) % 2 == 1
    p = hs.render_params(w, h, spp)
    p.flags = PT_RENDER_NEE if nee else 0
    p.row_begin, p.row_end, p.row_stride = it % stride, h, stride
    img = ds.render(p, traversal=trav)
    key = (name, w, h, spp, p.row_begin, stride, nee)
    if key not in ref:
        ref[key] = img
    elif not (ref[key].view(np.uint32) == img.view(np.uint32)).all():
        print("MISMATCH at iteration", it, key, "kernel", kernel)
        sys.exit(1)
    if it % 100 == 99:
        print(f"{it + 1} frames ok, {time.time() - t0:.1f} s", flush=True)
print("soak ok:", n, "frames,", len(ref), "distinct configurations")
