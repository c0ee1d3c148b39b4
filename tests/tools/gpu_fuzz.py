"""Fuzz: random scenes (all materials, spheres + triangles, sizes across every residency) rendered on the GPU and with the
oracle; frames and counters must be bit-identical in exact traversal, pruned traversal must stay within 1e-4 except for the
rare pixels it is allowed to flip.  Usage: python tests/tools/gpu_fuzz.py [n_scenes]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob
from conftest import random_scene
from test_fast_tree import scene_with_ties
from pathtracer_cuda_interactive_amd import PT_TRAVERSAL_PRUNED
from pathtracer_cuda_interactive_amd import device as dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2026)
t0 = time.time()
by_res = {}
flips = 0
reruns = 0
internal = 0
for k in range(n):
    n_tris = int(rng.choice([0, 2, 3, 7, 30, 45, 60, 150, 190, 230, 400, 900, 2500, 6000]))
    n_sph = int(rng.integers(0, 6)) if n_tris else int(rng.integers(1, 6))
    if k % 3 == 2:       # clouds of small triangles with doubled geometry: the kind of scene that keeps the internal tree, full of ties
        hs = scene_with_ties(1000 + k, n_tris=max(n_tris, 20))
    else:
        hs = random_scene(1000 + k, n_tris=n_tris, n_spheres=n_sph, emissive=bool(rng.integers(0, 2)))
    d = hs.finalize(int(rng.integers(0, 2)))
    w, h, spp = [(48, 36, 3), (33, 17, 5), (64, 8, 2), (20, 50, 4)][k % 4]
    p = hs.render_params(w, h, spp, seed=int(rng.integers(0, 1 << 30)))
    p.max_depth = int(rng.choice([50, 50, 3, 1]))
    want, cnt = ob.render(d, p)
    ds = dev.DeviceScene(d)
    ds.set_option("stats", 1)
    res = ds.info("residency")
    by_res[res] = by_res.get(res, 0) + 1
    img = ds.render(p)                       # default: scenes in global memory run on the internal tree with reference-order reruns
    reruns += ds.info("redo_segments")
    internal += ds.info("fast_tree_on")
    ok = bool((img.view(np.uint32) == want.view(np.uint32)).all())
    ds.set_option("fast_tree", 0)            # the caller's tree: counters must equal the oracle's
    img0 = ds.render(p)
    c = ds.counters()
    ok = ok and bool((img0.view(np.uint32) == want.view(np.uint32)).all())
    okc = (c.paths, c.segments, c.node_visits, c.leaf_tests) == (cnt.paths, cnt.segments, cnt.inner_pops, cnt.leaf_tri + cnt.leaf_sphere)
    pr = ds.render(p, traversal=PT_TRAVERSAL_PRUNED)
    bad = int((np.abs(pr - want).max(axis=2) > 1e-4).sum())
    flips += bad
    ds.close()
    if not (ok and okc) or bad > 2:
        print(f"MISMATCH scene {k}: tris {n_tris} spheres {n_sph} residency {res} {w}x{h}x{spp} exact {ok} counters {okc} pruned-diff-pixels {bad}")
        sys.exit(1)
    if k % 50 == 49:
        print(f"{k + 1} scenes ok, {time.time() - t0:.1f} s", flush=True)
print(f"fuzz ok: {n} scenes, residencies {dict(sorted(by_res.items()))}, pruned traversal flipped {flips} pixels in total, "
      f"{internal} scenes on the internal tree, {reruns} of their segments traced on the caller's tree (zero direction component)")
