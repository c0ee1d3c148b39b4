"""Exact traversal on the library's internal tree (pt_api.hip: validate_and_build / which_tree).

The reference's intersect() (scene.h:246-301) never prunes: a leaf is tested iff the ray passes the box test of every node
on the way down.  Where every box contains its children's boxes the slab test is monotone in the box, so that is: iff the
ray hits the leaf's OWN box — whatever tree sits above it.  The library therefore traverses scenes that live in global
memory on a surface-area tree of its own over the same leaf boxes.  What still depends on the visit order is handled the
reference's way: two candidates at equal t (the first VISITED wins, scene.h:270) are ordered by one box test at the node of
the caller's tree where their paths part, and rays with a zero direction component (1/d infinite, outside the argument)
are rerun on the caller's tree.  The result must equal the oracle ON THE CALLER'S TREE bit for bit; these tests aim at
exactly the cases the argument has to cover."""
import ctypes as C

import numpy as np
import pytest
from conftest import assert_bit_equal, assert_work_counters, load_scene, random_scene

from pathtracer_cuda_interactive_amd import PT_MAT_DIFFUSE, PT_TRAVERSAL_EXACT
from pathtracer_cuda_interactive_amd import device as dev
from pathtracer_cuda_interactive_amd.ctypes_defs import PtBvhNode, PtSceneDesc


def scene_with_ties(seed, n_tris=3000):
    """A cloud of small triangles (big enough for an internal tree, and of the kind where cuts of the Morton order beat the
    caller's median splits) with geometry built to tie: a few hundred of the triangles are there TWICE with different
    materials (equal t on every ray that hits them), and so is a fine grid of axis-aligned quads whose triangles share
    edges and vertices."""
    from pathtracer_cuda_interactive_amd import PT_MAT_MIRROR, PT_MAT_PHONG, HostScene
    rng = np.random.default_rng(seed)
    hs = HostScene()
    hs.set_camera((0, 0.5, 4.0), (0, 0, 0), (0, 1, 0), 50.0, 64, 48, 4)
    hs.set_background((0.4, 0.5, 0.6))
    grey = hs.add_material(PT_MAT_DIFFUSE, (0.6, 0.6, 0.5))
    red = hs.add_material(PT_MAT_DIFFUSE, (0.9, 0.1, 0.1))
    blue = hs.add_material(PT_MAT_PHONG, (0.1, 0.1, 0.9), exponent=20.0)
    mirror = hs.add_material(PT_MAT_MIRROR, (0.9, 0.9, 0.9))

    def cloud(n, size):
        c = (rng.random((n, 1, 3)) * 3 - 1.5).astype(np.float32)
        return (c + (rng.random((n, 3, 3)) - 0.5).astype(np.float32) * size).reshape(-1, 3).astype(np.float32)
    hs.add_mesh(cloud(n_tris, 0.2), np.arange(n_tris * 3, dtype=np.int32).reshape(-1, 3), grey)
    hs.add_mesh(cloud(200, 0.2), np.arange(600, dtype=np.int32).reshape(-1, 3), grey, radiance=(6.0, 5.0, 4.0))
    P = cloud(300, 0.4)
    I = np.arange(900, dtype=np.int32).reshape(-1, 3)
    hs.add_mesh(P, I, red)
    hs.add_mesh(P.copy(), I.copy(), blue)                                   # the same 300 triangles again
    hs.add_mesh(P[:450].copy(), I[:150].copy(), grey)                       # ... and half of them a third time: three-way ties
    m = GRID + 1
    g = np.linspace(-1.0, 1.0, m, dtype=np.float32)
    X, Y = np.meshgrid(g, g, indexing="xy")
    V = np.stack([X.ravel(), Y.ravel(), np.full(m * m, -0.5, np.float32)], axis=1).astype(np.float32)
    q = [(r * m + k, r * m + k + 1, (r + 1) * m + k + 1, (r + 1) * m + k) for r in range(GRID) for k in range(GRID)]
    T = np.array([[a, b, c_] for a, b, c_, d in q] + [[a, c_, d] for a, b, c_, d in q], dtype=np.int32)
    hs.add_mesh(V, T, red)
    hs.add_mesh(V.copy(), T.copy(), mirror)                                 # and the whole grid twice
    for k in range(3):
        hs.add_sphere(rng.random(3) * 2 - 1.0, 0.1 + 0.1 * float(rng.random()), [grey, mirror, blue][k])
    return hs


GRID = 32        # quads per side of the doubled grid (spacing 1/16: exactly representable, rays can be aimed at its vertices)


def probe_rays(hs, rng, n):
    """Random rays, rays along the axes and in the coordinate planes (zero components, both signs of zero), rays aimed at the
    grid's vertices and edges from an axis-aligned direction."""
    o = (rng.random((n, 3)) * 6 - 3).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    kind = rng.integers(0, 6, n)
    for k in range(n):
        if kind[k] == 1:                       # along one axis
            a = int(rng.integers(0, 3)); s = d[k, a]; d[k] = 0.0; d[k, a] = 1.0 if s > 0 else -1.0
        elif kind[k] == 2:                     # in a coordinate plane, the zero sometimes negative
            d[k, int(rng.integers(0, 3))] = -0.0 if rng.random() < 0.5 else 0.0
        elif kind[k] == 3:                     # straight down -z at a vertex / an edge of the doubled grid
            gx, gy = rng.integers(0, GRID + 1, 2)
            o[k] = (-1.0 + 2.0 / GRID * gx, -1.0 + 2.0 / GRID * gy if rng.random() < 0.5 else rng.random() * 2 - 1, 2.0)
            d[k] = (0.0, 0.0, -1.0)
        elif kind[k] == 4:                     # tilted at a grid vertex: finite 1/d, still ties between the triangles around it
            gx, gy = rng.integers(0, GRID + 1, 2)
            tgt = np.array([-1.0 + 2.0 / GRID * gx, -1.0 + 2.0 / GRID * gy, -0.5], np.float32)
            d[k] = tgt - o[k]
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7] = o, d, 1e-4, np.inf
    return rays


def desc_with_nodes(desc, nodes, root):
    d2 = PtSceneDesc()
    C.memmove(C.byref(d2), C.byref(desc), C.sizeof(PtSceneDesc))
    d2.nodes = nodes.ctypes.data_as(C.POINTER(PtBvhNode))
    d2.num_nodes = len(nodes)
    d2.root = root
    d2._keep = (nodes, desc)
    return d2


@pytest.mark.gpu
@pytest.mark.parametrize("seed,bvh", [(5, 0), (6, 1)])
def test_closest_hits_on_the_internal_tree_equal_the_reference_order_ray_by_ray(oracle, seed, bvh):
    hs = scene_with_ties(seed)
    d = hs.finalize(bvh)
    rays = probe_rays(hs, np.random.default_rng(seed), 60000)
    want_tuv, want_prim = oracle.intersect(d, rays)
    ds = dev.DeviceScene(d)
    try:
        assert ds.info("fast_tree") == 1
        tuv, prim = ds.intersect(rays, PT_TRAVERSAL_EXACT)
        reruns = ds.info("debug_reruns")
        ds.set_option("fast_tree", 0)
        tuv0, prim0 = ds.intersect(rays, PT_TRAVERSAL_EXACT)
        assert ds.info("debug_reruns") == 0
    finally:
        ds.close()
    assert np.array_equal(prim0, want_prim) and np.array_equal(tuv0.view(np.uint32), want_tuv.view(np.uint32))
    assert np.array_equal(prim, want_prim), f"{int((prim != want_prim).sum())} rays found another primitive"
    assert np.array_equal(tuv.view(np.uint32), want_tuv.view(np.uint32))
    # the doubled geometry and the axis-aligned rays did take the rerun path
    assert reruns > 1000, reruns
    assert (want_prim >= 0).sum() > 10000


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_render_with_ties_equals_the_oracle_on_the_callers_tree(oracle, seed):
    hs = scene_with_ties(seed, n_tris=3000)
    d = hs.finalize(seed & 1)
    p = hs.render_params(96, 72, 6, seed=seed)
    want, cnt = oracle.render(d, p)
    ds = dev.DeviceScene(d)
    try:
        ds.set_option("stats", 1)
        assert ds.info("fast_tree") == 1 and ds.info("residency") in (0, 3)
        img = ds.render(p, traversal=PT_TRAVERSAL_EXACT)
        c = ds.counters()
        reruns = ds.info("redo_segments")
        ds.set_option("top_cache", 0)
        img_plain = ds.render(p, traversal=PT_TRAVERSAL_EXACT)
    finally:
        ds.close()
    assert_bit_equal(img, want, "internal tree, ties")
    assert_bit_equal(img_plain, want, "internal tree, ties, no top-of-tree cache")
    assert (c.paths, c.segments) == (cnt.paths, cnt.segments)
    assert c.node_visits < cnt.inner_pops          # far fewer boxes than the caller's tree needs
    # and the ties are real: the oracle on the internal tree's own twin (same builder, its own visit order) sees other
    # primitives win in many pixels — the device, traversing that tree, still reproduces the caller's order
    d_sweep, _ = dev.build_bvh_sweep(d)
    other, _ = oracle.render(d_sweep, p)
    assert int((other != want).any(axis=2).sum()) > 50
    assert reruns < cnt.segments // 1000           # settled in place (ref_visits_first), not by rerunning the ray


@pytest.mark.gpu
def test_a_tree_whose_boxes_do_not_nest_is_traversed_as_given(oracle):
    """The equivalence needs nested boxes.  A caller's tree without them is legal for the reference (it tests whatever boxes
    it is given) — then there is no internal tree and the caller's is traversed, boxes as they are."""
    hs = random_scene(21, n_tris=4000, n_spheres=2)
    d = hs.finalize(0)
    nodes = hs.nodes_array().copy()
    inner = np.flatnonzero(nodes["prim"] < 0)
    root = int(d.root)
    rng = np.random.default_rng(3)
    picked = [int(k) for k in rng.choice(inner, 40, replace=False) if int(k) != root]
    for k in picked:                               # shrink some inner boxes: their children now stick out
        ctr = 0.5 * (nodes["bmin"][k] + nodes["bmax"][k])
        nodes["bmin"][k] = ctr + 0.6 * (nodes["bmin"][k] - ctr)
        nodes["bmax"][k] = ctr + 0.6 * (nodes["bmax"][k] - ctr)
    d2 = desc_with_nodes(d, nodes, root)
    p = hs.render_params(80, 60, 4, seed=2)
    want, cnt = oracle.render(d2, p)
    base, _ = oracle.render(d, p)
    assert (want != base).any()                    # the shrunken boxes do change what the reference would see
    ds = dev.DeviceScene(d2)
    try:
        ds.set_option("stats", 1)
        assert ds.info("fast_tree") == 0
        img = ds.render(p, traversal=PT_TRAVERSAL_EXACT)
        c = ds.counters()
        assert ds.info("redo_segments") == 0
    finally:
        ds.close()
    assert_bit_equal(img, want, "non-nested caller's tree")
    assert (c.segments, c.node_visits, c.leaf_tests) == (cnt.segments, cnt.inner_pops, cnt.leaf_tri + cnt.leaf_sphere)


@pytest.mark.gpu
def test_root_box_is_never_tested_so_it_may_be_anything(oracle):
    """scene.h:256 pushes the root without a box test; a caller's root box smaller than its children's must not matter —
    neither for the nestedness check nor for the result."""
    hs = scene_with_ties(22)
    d = hs.finalize(1)
    nodes = hs.nodes_array().copy()
    root = int(d.root)
    nodes["bmin"][root] = 0.0
    nodes["bmax"][root] = 0.0
    d2 = desc_with_nodes(d, nodes, root)
    p = hs.render_params(64, 48, 4, seed=4)
    want, _ = oracle.render(d2, p)
    ds = dev.DeviceScene(d2)
    try:
        assert ds.info("fast_tree") == 1
        img = ds.render(p, traversal=PT_TRAVERSAL_EXACT)
    finally:
        ds.close()
    assert_bit_equal(img, want, "degenerate root box")


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,spp,internal", [("bunny", 160, 120, 4, True), ("teapot", 128, 96, 4, True), ("cbox", 128, 96, 8, True),
                                                   ("scene4", 96, 72, 8, True), ("scene1", 96, 72, 8, False)])
def test_internal_tree_is_the_default_where_it_pays(oracle, name, w, h, spp, internal):
    """Meshes in global memory and LDS-resident scenes alike; a scene of a handful of primitives keeps the caller's tree."""
    hs, d = load_scene(name)
    p = hs.render_params(w, h, spp, seed=5)
    want, cnt = oracle.render(d, p)
    ds = dev.DeviceScene(d)
    try:
        ds.set_option("stats", 1)
        img = ds.render(p)
        c = ds.counters()
        assert ds.info("fast_tree") == ds.info("fast_tree_on") == (1 if internal else 0)
        assert_bit_equal(img, want, name)
        assert_work_counters(ds, c, cnt, oracle, d, p, name)
        if internal:
            assert 0 < ds.info("fast_tree_cost_permille") < 900
            assert ds.info("stack_entries") <= ds.info("fast_tree_depth")       # Strahler number, not depth
    finally:
        ds.close()


@pytest.mark.gpu
def test_sweep_tree_is_kept_only_where_it_touches_fewer_boxes(oracle):
    """Probe rays through both trees decide at scene creation (inner visits counted, info fast_tree_cost_permille).  A caller
    who hands in a tree the sweep tree cannot beat by 10 % — here: the sweep tree itself — keeps that topology; a scene in
    global memory then still gets the internal tree's WAY of traversing it (left child first on a short stack, leaves set
    aside, ties settled in place), which must not change a bit either."""
    hs, d0 = load_scene("teapot")
    d, _ = dev.build_bvh_sweep(d0)                            # the caller's tree = what the library would have built itself
    p = hs.render_params(96, 72, 4, seed=6)
    want, cnt = oracle.render(d, p)                           # the oracle on THAT tree (its own visit order, its own ties)
    ds = dev.DeviceScene(d)
    try:
        ds.set_option("stats", 1)
        assert ds.info("fast_tree_cost_permille") == 1000     # same tree: same visits
        assert ds.info("fast_tree") == 1 and ds.info("fast_tree_is_callers") == 1 and ds.info("residency") in (0, 3)
        assert ds.info("stack_entries") < ds.info("bvh_depth")                    # Strahler number + sentinel, not depth
        img = ds.render(p)
        c = ds.counters()
        assert_bit_equal(img, want, "caller's topology traversed the internal way")
        assert_work_counters(ds, c, cnt, oracle, d, p, "caller's topology")
        ds.set_option("fast_tree", 0)
        assert_bit_equal(ds.render(p), want, "caller's tree as given")
        assert_work_counters(ds, ds.counters(), cnt, oracle, d, p, "caller's tree as given")
    finally:
        ds.close()


@pytest.mark.gpu
def test_leaves_set_aside_is_an_internal_tree_schedule_only(oracle):
    """Scenes in global memory on the internal tree run the burst that sets leaves aside (v2_inner 1004): leaf tests in an
    order no tree prescribes (the default there is 1231: two rounds of 3 + 1).  Right on the internal tree (ties are settled by the caller's order whatever the test order),
    refused on the caller's tree, where the visit order IS the tie rule."""
    from pathtracer_cuda_interactive_amd import PT_ERR_INVALID_ARG, PtError
    hs, d = load_scene("teapot")
    p = hs.render_params(96, 72, 4, seed=8)
    want, cnt = oracle.render(d, p)
    ds = dev.DeviceScene(d)
    try:
        ds.set_option("stats", 1)
        for inner in (0, 1004, 1231, 4):         # 0 = automatic = 1231 here
            ds.set_option("v2_inner", inner)
            img = ds.render(p)
            assert_bit_equal(img, want, f"teapot v2_inner={inner}")
            assert_work_counters(ds, ds.counters(), cnt, oracle, d, p, f"teapot v2_inner={inner}")
        ds.set_option("fast_tree", 0)
        ds.set_option("v2_inner", 1004)
        with pytest.raises(PtError) as e:
            ds.render(p)
        assert e.value.status == PT_ERR_INVALID_ARG
        ds.set_option("v2_inner", 0)
        assert_bit_equal(ds.render(p), want, "teapot on the caller's tree")
    finally:
        ds.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,spp", [("cbox", 640, 480, 64), ("bunny", 640, 480, 64), ("buddha_standin", 1280, 960, 8),
                                          ("dragon_standin", 1920, 1080, 4), ("scene4", 640, 480, 32), ("teapot", 640, 480, 16)])
def test_full_size_frames_are_the_same_on_both_trees(name, w, h, spp):
    """BASELINE-size frames (too big for the oracle): internal tree == caller's tree, bit for bit, every pixel.  Events as rare
    as 1 ray in 1e7 — a zero direction component, a three-way tie — only show up at this size."""
    import os

    from conftest import SCENES
    from pathtracer_cuda_interactive_amd import HostScene, standins
    if name in standins.BUILDERS:
        hs = standins.BUILDERS[name](SCENES)
        d = hs.finalize(0)
    else:
        hs, d = load_scene(name)
    p = hs.render_params(w, h, spp)
    ds = dev.DeviceScene(d)
    try:
        assert ds.info("fast_tree_on") == 1
        ds.set_option("stats", 1)
        a = ds.render(p)
        reruns = ds.info("redo_segments")
        ds.set_option("stats", 0)
        ds.set_option("fast_tree", 0)
        b = ds.render(p)
    finally:
        ds.close()
    assert_bit_equal(a, b, f"{name} {w}x{h}x{spp}: internal tree vs caller's tree")
    if name in ("cbox", "buddha_standin", "dragon_standin"):
        assert reruns > 0            # rays with a zero direction component did occur in this frame


@pytest.mark.gpu
def test_more_blocks_per_cu_than_any_default_keeps_the_rerun_stacks_inside_their_buffer(oracle):
    """`blocks_per_cu` is a caller's option; the reference-order reruns (rays with a zero direction component: cbox has ~20 per
    million segments) index a global-memory stack by blockIdx.  The buffer is sized from the grid that is launched, so a grid
    of 12 blocks per CU renders the same bits as the default one — and an option beyond the documented range is refused."""
    from pathtracer_cuda_interactive_amd import PtError
    hs, d = load_scene("cbox")
    p = hs.render_params(640, 480, 8)
    want, cnt = oracle.render(d, p)
    ds = dev.DeviceScene(d)
    try:
        assert ds.info("fast_tree_on") == 1
        ds.set_option("stats", 1)
        for bpc in (0, 12, 2):
            ds.set_option("blocks_per_cu", bpc)
            img = ds.render(p)
            assert_bit_equal(img, want, f"cbox blocks_per_cu={bpc}")
            if bpc:
                assert ds.info("blocks_per_cu") == bpc and ds.info("grid") == bpc * ds.info("num_cus")
            assert ds.info("redo_segments") > 0          # the reruns did happen
        with pytest.raises(PtError):
            ds.set_option("blocks_per_cu", 33)
        with pytest.raises(PtError):
            ds.set_option("blocks_per_cu", -1)
    finally:
        ds.close()
