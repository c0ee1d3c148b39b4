"""Fuzz of the round-2 extensions: random scenes (all materials, spheres + triangles, emissive triangles and spheres, point
lights, sizes across every residency) rendered on the GPU and with the oracle
  (a) with next-event estimation (PT_RENDER_NEE) on the reference tree,
  (b) on the device-built LBVH and SAH trees (pt_bvh_build_device), with and without NEE;
every frame must be bit-identical to the oracle's on the same tree, every device-built tree a valid cover.
Usage: python tests/tools/gpu_fuzz2.py [n_scenes]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob
from conftest import random_scene
from test_device_bvh import check_tree, host_leaf_boxes
from pathtracer_cuda_interactive_amd import PT_RENDER_NEE
from pathtracer_cuda_interactive_amd import device as dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(77)
t0 = time.time()
by_res, frames, max_depth = {}, 0, 0
for k in range(n):
    n_tris = int(rng.choice([0, 2, 3, 7, 30, 45, 60, 150, 190, 230, 400, 900, 2500, 6000]))
    n_sph = int(rng.integers(0, 6)) if n_tris else int(rng.integers(1, 6))
    hs = random_scene(5000 + k, n_tris=n_tris, n_spheres=n_sph, emissive=bool(rng.integers(0, 4)))
    for _ in range(int(rng.integers(0, 3))):
        hs.add_point_light(rng.random(3) * 4 - 2, rng.random(3) * 8)
    d = hs.finalize(int(rng.integers(0, 2)))
    w, h, spp = [(48, 36, 3), (33, 17, 5), (64, 8, 2), (20, 50, 4)][k % 4]
    p = hs.render_params(w, h, spp, seed=int(rng.integers(0, 1 << 30)))
    p.max_depth = int(rng.choice([50, 50, 3, 1]))
    q = p.copy(); q.flags = PT_RENDER_NEE
    trees = [("reference", d)]
    for label, m in (("lbvh", dev.PT_BVH_DEVICE_LBVH), ("sah", dev.PT_BVH_DEVICE_SAH)):
        d2, info = dev.build_bvh_device(d, m)
        max_depth = max(max_depth, check_tree(info["nodes"], info["root"], d.num_shapes, host_leaf_boxes(hs)))
        trees.append((label, d2))
    for label, dd in trees:
        ds = dev.DeviceScene(dd)
        res = ds.info("residency")
        by_res[res] = by_res.get(res, 0) + 1
        for pp, what in ((p, "plain"), (q, "nee")):
            if label == "reference" and what == "plain":
                continue                                   # covered by gpu_fuzz.py
            want, cnt = ob.render(dd, pp)
            img = ds.render(pp)
            c = ds.counters()
            frames += 1
            if not (img.view(np.uint32) == want.view(np.uint32)).all() or (c.paths, c.segments) != (cnt.paths, cnt.segments):
                print(f"MISMATCH scene {k}: tris {n_tris} spheres {n_sph} tree {label} {what} residency {res} {w}x{h}x{spp}")
                sys.exit(1)
        ds.close()
    if k % 50 == 49:
        print(f"{k + 1} scenes ok, {frames} frames, {time.time() - t0:.1f} s", flush=True)
print(f"fuzz2 ok: {n} scenes, {frames} frames bit-identical to the oracle, residencies {dict(sorted(by_res.items()))}, deepest device-built tree {max_depth}")
