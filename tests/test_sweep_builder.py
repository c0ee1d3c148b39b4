"""pt_bvh_build_sweep — the builder behind the library's internal tree (pt_tree_sweep.h) — is host code: the whole of it can be
checked here, without a GPU, by letting the ORACLE traverse its trees."""
import ctypes as C

import numpy as np
import pytest
from conftest import load_scene, random_scene
from test_device_bvh import check_tree, host_leaf_boxes

from pathtracer_cuda_interactive_amd import PtError
from pathtracer_cuda_interactive_amd import device as dev
from pathtracer_cuda_interactive_amd.ctypes_defs import PtBvhNode, PtSceneDesc


@pytest.mark.parametrize("name,w,h,spp,gain", [("cbox", 64, 48, 8, 0.75), ("scene4", 64, 48, 8, 0.9), ("teapot", 64, 48, 2, 0.35),
                                               ("tetrahedron", 33, 17, 4, 1.01), ("scene1", 40, 30, 4, 1.01),
                                               ("bunny", 48, 36, 2, 0.45)])       # 288 k primitives: the threaded build
def test_sweep_tree_is_a_valid_cover_and_cheaper_to_traverse(oracle, name, w, h, spp, gain):
    hs, d = load_scene(name)
    d2, info = dev.build_bvh_sweep(d)
    depth = check_tree(info["nodes"], info["root"], d.num_shapes, host_leaf_boxes(hs))
    assert depth == info["depth"]
    again, info2 = dev.build_bvh_sweep(d)
    assert info2["root"] == info["root"] and info2["nodes"].tobytes() == info["nodes"].tobytes()      # deterministic
    p = hs.render_params(w, h, spp, seed=3)
    ref_img, ref_cnt = oracle.render(d, p)
    img, cnt = oracle.render(d2, p)
    # same closest hits except where two primitives tie on t (the first one visited wins, scene.h:270)
    assert int((img != ref_img).any(axis=2).sum()) <= max(2, w * h // 200)
    if not (img != ref_img).any():
        # identical images -> identical paths, and every ray tests the SAME leaves on either tree (a leaf is tested iff the ray
        # hits its own box) — except the few rays with a zero direction component, whose 0 * inf slab products fall outside
        # that argument (the device reruns those on the caller's tree): about 2 in 1e5 segments of cbox
        assert cnt.paths == ref_cnt.paths and cnt.segments == ref_cnt.segments
        assert abs((cnt.leaf_tri + cnt.leaf_sphere) - (ref_cnt.leaf_tri + ref_cnt.leaf_sphere)) <= 1e-4 * (ref_cnt.leaf_tri + ref_cnt.leaf_sphere)
    assert cnt.inner_pops <= gain * ref_cnt.inner_pops, (cnt.inner_pops / cnt.segments, ref_cnt.inner_pops / ref_cnt.segments)


def _desc_with_leaf_boxes(hs, d, lo, hi):
    """A copy of `d` whose leaf boxes are replaced (the builder reads nothing else)."""
    nodes = hs.nodes_array().copy()
    leaf = np.flatnonzero(nodes["prim"] >= 0)
    nodes["bmin"][leaf] = lo[nodes["prim"][leaf]]
    nodes["bmax"][leaf] = hi[nodes["prim"][leaf]]
    d2 = PtSceneDesc()
    C.memmove(C.byref(d2), C.byref(d), C.sizeof(PtSceneDesc))
    d2.nodes = nodes.ctypes.data_as(C.POINTER(PtBvhNode))
    d2._keep = (nodes, d)
    return d2


def test_degenerate_inputs():
    hs = random_scene(3, n_tris=300, n_spheres=0)
    d = hs.finalize(0)
    n = d.num_shapes
    # every box the same: all cuts cost the same, the tie rule takes the middle one -> a balanced tree, not a chain
    lo = np.zeros((n, 3), np.float32)
    hi = np.ones((n, 3), np.float32)
    _, info = dev.build_bvh_sweep(_desc_with_leaf_boxes(hs, d, lo, hi))
    assert check_tree(info["nodes"], info["root"], n, (lo, hi)) == info["depth"] <= int(np.ceil(np.log2(n))) + 1
    # nested shells, each twice the size of the one before: the cheapest cut always peels off the outermost one; the depth
    # guard takes over with median cuts (2 log2 n + 16 levels of peeling at most)
    k = np.arange(n, dtype=np.float32)[:, None]
    hi = np.minimum(2.0 ** (k / 4), 1e30).astype(np.float32) * np.ones((1, 3), np.float32)
    lo = -hi
    _, info = dev.build_bvh_sweep(_desc_with_leaf_boxes(hs, d, lo, hi))
    depth = check_tree(info["nodes"], info["root"], n, (lo, hi))
    assert depth == info["depth"] <= 2 * int(np.ceil(np.log2(n))) + 16 + int(np.ceil(np.log2(n))) + 2
    # flat and point boxes
    lo = np.random.default_rng(1).random((n, 3)).astype(np.float32)
    hi = lo.copy()
    hi[::2, 0] += 0.5
    _, info = dev.build_bvh_sweep(_desc_with_leaf_boxes(hs, d, lo, hi))
    check_tree(info["nodes"], info["root"], n, (lo, hi))
    # a box that is not finite is refused
    hi[7, 1] = np.inf
    with pytest.raises(PtError):
        dev.build_bvh_sweep(_desc_with_leaf_boxes(hs, d, lo, hi))


def test_one_and_two_primitives():
    from pathtracer_cuda_interactive_amd import PT_MAT_DIFFUSE, HostScene
    for n in (1, 2):
        hs = HostScene()
        hs.set_camera((0, 0, 4.0), (0, 0, 0), (0, 1, 0), 45.0, 16, 16, 1)
        m = hs.add_material(PT_MAT_DIFFUSE, (0.5, 0.5, 0.5))
        for k in range(n):
            hs.add_sphere((k * 1.5, 0, 0), 0.5, m)
        d = hs.finalize(0)
        _, info = dev.build_bvh_sweep(d)
        assert check_tree(info["nodes"], info["root"], n, host_leaf_boxes(hs)) == info["depth"] == n


def _rays_through(hs, rng, n):
    """Rays that start anywhere around the scene and leave in any direction — no zero direction component."""
    na = hs.nodes_array()
    lo, hi = na["bmin"].min(axis=0), na["bmax"].max(axis=0)
    ext = np.maximum(hi - lo, 1e-3)
    o = (lo - 0.5 * ext + rng.random((n, 3)) * 2.0 * ext).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[d == 0] = np.float32(1e-3)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7] = o, d, 1e-4, np.inf
    return rays


@pytest.mark.parametrize("name,n_rays", [("cbox", 200000), ("scene4", 200000), ("teapot", 100000), ("bunny", 40000)])
def test_any_tree_over_the_same_leaf_boxes_reaches_the_same_leaves(oracle, name, n_rays):
    """The statement the internal tree stands on, checked ray by ray on the CPU: the reference's traversal (it never prunes)
    reaches a leaf iff the ray hits the leaf's own box, so the caller's tree, the sweep tree and a deliberately bad tree — a
    chain that peels one primitive off per level — all reach the SAME SET of leaves for every ray with finite 1/d.  Rays
    with a zero direction component are outside the statement (when such a ray starts exactly on a box plane of that axis —
    a bounce off an axis-aligned wall does — the slab test multiplies 0 by inf); the device traces those on the caller's tree."""
    hs, d = load_scene(name)
    rng = np.random.default_rng(11)
    rays = _rays_through(hs, rng, n_rays)
    inner0, leaf0, set0 = oracle.intersect_work(d, rays)
    d_sweep, _ = dev.build_bvh_sweep(d)
    inner1, leaf1, set1 = oracle.intersect_work(d_sweep, rays)
    assert np.array_equal(leaf0, leaf1) and np.array_equal(set0, set1)
    assert inner1.sum() < inner0.sum()
    assert leaf0.max() > 0 and (leaf0 > 0).mean() > 0.05                     # the rays do reach leaves
    if d.num_shapes <= 64:
        # the worst tree there is: primitive k against everything after it (depth N), boxes = exact unions
        lo, hi = host_leaf_boxes(hs)
        n = d.num_shapes
        nodes = np.zeros(2 * n - 1, dtype=dev.NODE_DTYPE)
        suffix_lo = np.minimum.accumulate(lo[::-1], axis=0)[::-1]
        suffix_hi = np.maximum.accumulate(hi[::-1], axis=0)[::-1]
        for k in range(n):                      # leaves 0..n-1
            nodes[k] = (lo[k], hi[k], -1, -1, k)
        for k in range(n - 1):                  # inner node n+k covers primitives k..n-1
            right = n - 1 if k == n - 2 else n + k + 1
            nodes[n + k] = (suffix_lo[k], suffix_hi[k], k, right, -1)
        d_chain = _desc_with_nodes(d, nodes, n)
        inner2, leaf2, set2 = oracle.intersect_work(d_chain, rays)
        assert np.array_equal(leaf0, leaf2) and np.array_equal(set0, set2)


def _desc_with_nodes(desc, nodes, root):
    d2 = PtSceneDesc()
    C.memmove(C.byref(d2), C.byref(desc), C.sizeof(PtSceneDesc))
    d2.nodes = nodes.ctypes.data_as(C.POINTER(PtBvhNode))
    d2.num_nodes = len(nodes)
    d2.root = root
    d2._keep = (nodes, desc)
    return d2


@pytest.mark.parametrize("name", ["scene4", "cbox", "teapot", "bunny"])
def test_sweep_trees_are_pinned(name):
    """Node pools of the internal tree for the fixture scenes, byte for byte (tests/golden/pins.json, written by
    make_golden_vectors.py): the builder is deterministic across runs, thread counts and builds — a change of it must show."""
    import hashlib
    import json
    import os

    from conftest import GOLDEN
    pin = json.load(open(os.path.join(GOLDEN, "pins.json")))["sweep_tree"][name]
    hs, d = load_scene(name)
    _, info = dev.build_bvh_sweep(d)
    assert (int(d.num_shapes), int(info["depth"])) == (pin["num_shapes"], pin["depth"])
    assert hashlib.md5(info["nodes"].tobytes()).hexdigest() == pin["md5"]
