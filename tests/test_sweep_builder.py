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
