"""Soak test: many back-to-back frames over scenes / traversal modes / kernel variants / shard shapes; every frame must
be bit-identical to the first one of its configuration.  Catches rare scheduling-dependent bugs and hangs.
Every third iteration submits a BURST of 2-5 frames without a host sync in between (pt_render_async on one or two caller
streams, frames_in_flight 1-4): overlapping launches, slot reuse, partial grids — each frame of the burst must be that same image."""
import os, sys, time
import numpy as np
import torch
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
caller_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
frames = 0
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
    ds.set_option("frames_in_flight", [2, 1, 3, 4][(it // 3) % 4])
    key = (name, w, h, spp, p.row_begin, stride, nee)
    if it % 3 == 2:
        burst = [2, 3, 5, 4][(it // 9) % 4]
        rows = len(range(p.row_begin, h, stride))
        streams = caller_streams[: 1 + (it // 27) % 2]
        outs = [torch.empty(rows, w, 3, dtype=torch.float32, device="cuda") for _ in range(burst)]
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            ds.render_into(p, o.data_ptr(), stream=streams[k % len(streams)].cuda_stream, traversal=trav)
        torch.cuda.synchronize()
        frames += burst
        imgs = [o.cpu().numpy() for o in outs]
        if key not in ref:
            ref[key] = imgs[0]
        if not all((ref[key].view(np.uint32) == im.reshape(ref[key].shape).view(np.uint32)).all() for im in imgs):
            print("MISMATCH in a burst at iteration", it, key, "kernel", kernel, "burst", burst)
            sys.exit(1)
        img = ref[key]
    else:
        img = ds.render(p, traversal=trav)
        frames += 1
    if key not in ref:
        ref[key] = img
    elif not (ref[key].view(np.uint32) == img.view(np.uint32)).all():
        print("MISMATCH at iteration", it, key, "kernel", kernel)
        sys.exit(1)
    if it % 100 == 99:
        print(f"{it + 1} iterations ({frames} frames) ok, {time.time() - t0:.1f} s", flush=True)
print("soak ok:", n, "iterations,", frames, "frames,", len(ref), "distinct configurations")
