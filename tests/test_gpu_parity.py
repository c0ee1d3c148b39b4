"""GPU parity tests: the HIP path (through the C ABI, libpt_hip.so) against the CPU oracle and the
committed golden vectors.  Bar: BIT-EXACT fp32 (stronger than BASELINE.json's 1e-4 L-inf, which is
asserted as well) — any 1-ulp difference would be amplified into different paths."""
import glob
import os

import numpy as np
import pytest
from conftest import assert_work_counters, GOLDEN, assert_bit_equal, load_scene, random_scene
from test_golden_vectors import parse_image_name

from pathtracer_cuda_interactive_amd import (PT_ERR_BAD_SCENE, PT_ERR_INVALID_ARG, PT_ERR_UNSUPPORTED, PT_MAT_DIFFUSE,
                                             PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED, HostScene, PtError)
from pathtracer_cuda_interactive_amd import device as dev

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json: per-channel L-inf vs the oracle on identical PCG seeds
IMAGES = sorted(glob.glob(os.path.join(GOLDEN, "images", "*.npy")))
TRAVERSALS = [PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED]


@pytest.fixture(scope="module")
def dscenes():
    cache = {}

    def get(name):
        if name not in cache:
            _, d = load_scene(name)
            cache[name] = dev.DeviceScene(d)
        return cache[name]
    yield get
    for s in cache.values():
        s.close()


@pytest.mark.parametrize("path", IMAGES, ids=os.path.basename)
@pytest.mark.parametrize("trav", TRAVERSALS)
@pytest.mark.parametrize("force_global", [0, 1])
@pytest.mark.parametrize("kernel", [2, 1])
def test_device_reproduces_golden_image(dscenes, path, trav, force_global, kernel):
    """kernel 2 = decoupled traversal/shading scheduler (default), kernel 1 = segment-synchronous wavefront loop."""
    name, w, h, spp = parse_image_name(path)
    hs, _ = load_scene(name)
    ds = dscenes(name)
    ds.set_option("force_global", force_global)
    ds.set_option("kernel", kernel)
    try:
        img = ds.render(hs.render_params(w, h, spp), traversal=trav)
    finally:
        ds.set_option("force_global", 0)
        ds.set_option("kernel", 2)
    want = np.load(path)
    assert np.abs(img - want).max() <= TOL
    assert_bit_equal(img, want, f"{name} trav={trav} global={force_global} kernel={kernel}")


@pytest.mark.parametrize("thresh,inner,minw", [(40, -6, 6), (32, 4, 6), (40, 4, 6), (40, 3, 6), (40, 162, 6)])
def test_scheduler_variants_are_bit_identical(oracle, dscenes, thresh, inner, minw):
    """The scheduling knobs of trace_kernel_v2 change WHEN a lane runs, never what it computes."""
    hs, d = load_scene("cbox")
    ds = dscenes("cbox")
    p = hs.render_params(72, 54, 7, seed=3)
    want, cnt = oracle.render(d, p)
    for k, v in (("v2_thresh", thresh), ("v2_inner", inner), ("v2_minw", minw), ("stats", 1)):
        ds.set_option(k, v)
    try:
        img = ds.render(p)
        c = ds.counters()
    finally:
        for k, v in (("v2_thresh", 0), ("v2_inner", 0), ("v2_minw", 0), ("stats", 0)):     # 0 = automatic choice
            ds.set_option(k, v)
    assert_bit_equal(img, want, f"T{thresh} I{inner} W{minw}")
    assert_work_counters(ds, c, cnt, oracle, d, p)
    ds.set_option("v2_thresh", 17)
    with pytest.raises(PtError) as e:            # not compiled in
        ds.render(p)
    ds.set_option("v2_thresh", 0)
    assert e.value.status == PT_ERR_INVALID_ARG


@pytest.mark.parametrize("name,w,h,spp", [("cbox", 96, 72, 9), ("scene1", 80, 60, 12), ("scene1_phong", 64, 64, 10),
                                          ("teapot", 50, 40, 3), ("bunny", 40, 30, 2), ("tetrahedron", 33, 17, 5)])
def test_device_matches_live_oracle(oracle, dscenes, name, w, h, spp):
    hs, d = load_scene(name)
    p = hs.render_params(w, h, spp, seed=77)
    want, cnt = oracle.render(d, p)
    ds = dscenes(name)
    ds.set_option("stats", 1)
    img = ds.render(p, traversal=PT_TRAVERSAL_EXACT)
    c = ds.counters()
    assert_bit_equal(img, want, name)
    assert (c.paths, c.segments) == (cnt.paths, cnt.segments)
    assert_work_counters(ds, c, cnt, oracle, d, p, name)
    if ds.info("fast_tree"):
        ds.set_option("fast_tree", 0)               # the caller's tree: the oracle's own visit counts
        img = ds.render(p, traversal=PT_TRAVERSAL_EXACT)
        c = ds.counters()
        assert_bit_equal(img, want, name + " on the caller's tree")
        assert_work_counters(ds, c, cnt, oracle, d, p, name + " on the caller's tree")
        ds.set_option("fast_tree", 1)
    ds.set_option("stats", 0)
    # pruned traversal is NOT guaranteed bit-exact (a triangle's t can round below its box's entry distance: DESIGN.md §6):
    # a legitimate rounding flip may move a pixel by one path's radiance / spp, so this is a tolerance, not a bit test
    img2 = ds.render(p, traversal=PT_TRAVERSAL_PRUNED)
    diff_px = int((np.abs(img2 - want).max(axis=2) > 0).sum())
    assert diff_px <= 2, f"{diff_px} pixels differ between pruned and exact traversal"
    assert abs(float(img2.mean()) - float(want.mean())) < 1e-3 * max(float(want.mean()), 1e-6)


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_all_materials(oracle, seed):
    hs = random_scene(seed, n_tris=24 + 7 * seed, n_spheres=3 + seed % 3)
    d = hs.finalize()
    p = hs.render_params(56, 40, 6, seed=1000 + seed)
    want, cnt = oracle.render(d, p)
    assert cnt.emit > 0 and cnt.term_rr + cnt.term_miss > 0 and cnt.leaf_tri > 0 and cnt.leaf_sphere > 0
    ds = dev.DeviceScene(d)
    try:
        for trav in TRAVERSALS:
            for fg in (0, 1):
                ds.set_option("force_global", fg)
                assert_bit_equal(ds.render(p, traversal=trav), want, f"seed {seed} trav {trav} global {fg}")
    finally:
        ds.close()


@pytest.mark.parametrize("name", ["cbox", "scene1", "scene1_phong", "tetrahedron"])
def test_octant_node_tables_do_not_change_the_image(oracle, dscenes, name):
    """Small scenes keep 8 ray-octant copies of the node table in LDS (near/far planes pre-swapped, no per-visit
    selects).  Same arithmetic on the same operands: the frame must not change by a bit."""
    hs, d = load_scene(name)
    ds = dscenes(name)
    p = hs.render_params(70, 50, 6, seed=21)
    want, _ = oracle.render(d, p)
    assert ds.info("residency") == 2
    with_oct = ds.render(p)
    ds.set_option("octants", 0)
    try:
        assert ds.info("residency") == 1
        without = ds.render(p)
    finally:
        ds.set_option("octants", 1)
    assert_bit_equal(with_oct, want, name + " octants")
    assert_bit_equal(without, want, name + " single table")


@pytest.mark.parametrize("name", ["teapot", "bunny"])
def test_top_of_tree_cache_does_not_change_the_image(oracle, dscenes, name):
    """Scenes read from global memory keep the top levels of the BVH (breadth-first prefix of the node array) in LDS.
    The cached copy holds the same bytes, so frames and counters must not change by a bit, with the cache or without."""
    hs, d = load_scene(name)
    ds = dscenes(name)
    p = hs.render_params(64, 48, 5, seed=8)
    want, cnt = oracle.render(d, p)
    assert ds.info("residency") == 3 and ds.info("top_nodes") > 0
    assert_bit_equal(ds.render(p), want, name + " internal tree, top of it in LDS")       # the default path
    ds.set_option("top_cache", 0)
    assert_bit_equal(ds.render(p), want, name + " internal tree, all nodes from global memory")
    ds.set_option("top_cache", 1)
    ds.set_option("fast_tree", 0)                 # the caller's tree: the work counters must equal the oracle's
    ds.set_option("stats", 1)
    try:
        cached = ds.render(p)
        c1 = ds.counters()
        ds.set_option("top_cache", 0)
        assert ds.info("residency") == 0 and ds.info("top_nodes") == 0
        plain = ds.render(p)
        c0 = ds.counters()
    finally:
        ds.set_option("top_cache", 1)
        ds.set_option("stats", 0)
        ds.set_option("fast_tree", 1)
    assert_bit_equal(cached, want, name + " top of the tree in LDS")
    assert_bit_equal(plain, want, name + " all nodes from global memory")
    for c in (c0, c1):
        assert (c.paths, c.segments, c.node_visits, c.leaf_tests) == (cnt.paths, cnt.segments, cnt.inner_pops, cnt.leaf_tri + cnt.leaf_sphere)
    # the LDS budget of the cache (more cached nodes, fewer resident blocks) and the work-feed chunk only move speed
    tops = set()
    for kb in (24, 39, 52):
        ds.set_option("lds_budget_kb", kb)
        try:
            assert_bit_equal(ds.render(p), want, f"{name} lds budget {kb} KB")
            tops.add(ds.info("top_nodes"))
        finally:
            ds.set_option("lds_budget_kb", 0)
    assert len(tops) > 1
    for chunk in (64, 128, 256):
        ds.set_option("chunk", chunk)
        try:
            assert_bit_equal(ds.render(p), want, f"{name} chunk {chunk}")
        finally:
            ds.set_option("chunk", 0)


@pytest.mark.parametrize("n_tris,n_spheres", [(30, 3), (44, 2), (46, 1), (47, 1), (60, 4), (170, 4), (186, 2), (196, 4), (400, 4)])
def test_residency_thresholds(oracle, n_tris, n_spheres):
    """Scenes around the two size thresholds: 8 octant node tables in LDS (<= 48 inner nodes), one table in LDS
    (<= 36 KB of scene), global memory with the top of the tree in LDS (3) or without (0).  Every residency must render
    the oracle's image."""
    hs = random_scene(100 + n_tris, n_tris=n_tris, n_spheres=n_spheres)
    d = hs.finalize()
    p = hs.render_params(48, 36, 4, seed=n_tris)
    want, _ = oracle.render(d, p)
    ds = dev.DeviceScene(d)
    try:
        res = ds.info("residency")
        inner = ds.info("num_inner_nodes")
        expect = 2 if inner * 8 * 64 <= 24 * 1024 else (1 if ds.info("scene_bytes") <= 36 * 1024 else 3)
        assert res == expect, (res, inner, ds.info("scene_bytes"))
        assert_bit_equal(ds.render(p), want, f"{n_tris} tris residency {res}")
        if res == 3:
            assert 0 < ds.info("top_nodes") <= inner
            ds.set_option("top_cache", 0)
            assert ds.info("residency") == 0
            assert_bit_equal(ds.render(p), want, f"{n_tris} tris residency 0 (renumbered nodes, no cache)")
            ds.set_option("top_cache", 1)
        if res == 2:
            ds.set_option("octants", 0)
            assert_bit_equal(ds.render(p), want, f"{n_tris} tris residency 1")
        ds.set_option("force_global", 1)
        assert ds.info("residency") == 0
        assert_bit_equal(ds.render(p), want, f"{n_tris} tris residency 0")
    finally:
        ds.close()


def test_row_ranges_and_strides_tile_the_image(oracle, dscenes):
    hs, d = load_scene("cbox")
    ds = dscenes("cbox")
    p = hs.render_params(48, 37, 5)
    full = ds.render(p)
    assert_bit_equal(full, oracle.render(d, p)[0], "full")
    # contiguous bands
    parts = []
    for rb, re in [(0, 10), (10, 11), (11, 37)]:
        q = p.copy()
        q.row_begin, q.row_end = rb, re
        parts.append(ds.render(q))
    assert_bit_equal(np.concatenate(parts, axis=0), full, "bands")
    # interleaved rows, as the multi-GPU sharding uses them (rank r of n renders rows r, r+n, ...)
    n = 4
    out = np.zeros_like(full)
    for r in range(n):
        q = p.copy()
        q.row_begin, q.row_end, q.row_stride = r, p.height, n
        part = ds.render(q)
        assert part.shape[0] == len(range(r, p.height, n))
        assert_bit_equal(part, oracle.render(d, q)[0], f"stride rank {r}")
        out[r::n] = part
    assert_bit_equal(out, full, "interleaved")
    # empty selection
    q = p.copy()
    q.row_begin, q.row_end = 5, 5
    assert ds.render(q).shape == (0, 48, 3)


def test_multi_pass_equals_single_pass(dscenes):
    hs, _ = load_scene("cbox")
    ds = dscenes("cbox")
    p = hs.render_params(40, 30, 11)
    one = ds.render(p)
    assert ds.info("passes") == 1
    ds.set_option("scratch_bytes", 40 * 30 * 16 * 3)      # room for 3 samples per pass -> 4 passes
    many = ds.render(p)
    assert ds.info("passes") == 4
    ds.set_option("scratch_bytes", 0)
    assert_bit_equal(many, one, "multi-pass")


def test_sample_offset_and_stream_stride_address_absolute_samples(oracle, dscenes):
    """Samples [4,8) of an 8-spp stream layout rendered alone == the oracle doing the same."""
    hs, d = load_scene("scene1")
    ds = dscenes("scene1")
    p = hs.render_params(32, 24, 4)
    p.sample_offset, p.stream_stride = 4, 8
    assert_bit_equal(ds.render(p), oracle.render(d, p)[0], "offset")
    p0 = hs.render_params(32, 24, 4)
    assert np.abs(ds.render(p) - ds.render(p0)).max() > 0


def test_progressive_accumulation_matches_render_progressive_semantics(oracle, dscenes):
    """pt_render_accumulate == render_progressive (main.cu:64-89): accum (=|+=) sum of the call's new samples."""
    import torch
    hs, d = load_scene("cbox")
    ds = dscenes("cbox")
    W, H, total = 40, 30, 6
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    acc.fill_(123.0)                                  # must be overwritten by the first call
    want = None
    for k, (off, n) in enumerate([(0, 2), (2, 3), (5, 1)]):
        p = hs.render_params(W, H, n)
        p.sample_offset, p.stream_stride = off, total
        ds.accumulate_into(p, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
        part, _ = oracle.render(d, p, accumulate=True)
        want = part if k == 0 else (want + part).astype(np.float32)
    torch.cuda.synchronize()
    assert_bit_equal(acc.cpu().numpy(), want, "progressive")
    # ... and it converges to the offline image (different summation order: tolerance, not bits)
    off = ds.render(hs.render_params(W, H, total))
    np.testing.assert_allclose(acc.cpu().numpy() / total, off, rtol=1e-5, atol=1e-6)


def test_async_render_into_torch_tensor_on_side_stream(oracle, dscenes):
    import torch
    hs, d = load_scene("scene1")
    ds = dscenes("scene1")
    p = hs.render_params(64, 32, 3)
    st = torch.cuda.Stream()
    fb = torch.empty((32, 64, 3), dtype=torch.float32, device="cuda")
    with torch.cuda.stream(st):
        ds.render_into(p, fb.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert_bit_equal(fb.cpu().numpy(), oracle.render(d, p)[0], "async")
    c = ds.counters()
    assert c.paths == 64 * 32 * 3 and c.kernel_ms > 0


@pytest.mark.parametrize("max_depth,rr_depth", [(1, 5), (2, 0), (7, 2), (50, 49)])
def test_depth_limits(oracle, dscenes, max_depth, rr_depth):
    hs, d = load_scene("cbox")
    p = hs.render_params(32, 24, 4)
    p.max_depth, p.rr_depth = max_depth, rr_depth
    want, cnt = oracle.render(d, p)
    if max_depth <= 2:
        assert cnt.term_maxdepth > 0
    assert_bit_equal(dscenes("cbox").render(p), want, f"depth {max_depth}/{rr_depth}")


def test_edge_sizes(oracle, dscenes):
    hs, d = load_scene("cbox")
    ds = dscenes("cbox")
    for w, h, spp in [(1, 1, 1), (1, 7, 2), (65, 1, 1), (3, 3, 70), (7, 9, 3), (641, 17, 1), (33, 8, 2), (5, 16, 1)]:
        p = hs.render_params(w, h, spp)
        assert_bit_equal(ds.render(p), oracle.render(d, p)[0], f"{w}x{h}x{spp}")


def test_single_primitive_scenes(oracle):
    """Root is a leaf: no inner nodes at all (bvh.cu:18-25)."""
    for kind in ("sphere", "triangle"):
        hs = HostScene()
        hs.set_camera((0, 0, 3), (0, 0, 0), (0, 1, 0), 45, 24, 24, 4)
        m = hs.add_material(PT_MAT_DIFFUSE, (0.7, 0.6, 0.5))
        if kind == "sphere":
            hs.add_sphere((0, 0, 0), 1.0, m, radiance=(1, 2, 3))
        else:
            hs.add_mesh(np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.int32), m)
        d = hs.finalize()
        assert d.num_nodes == 1
        p = hs.render_params()
        want, cnt = oracle.render(d, p)
        assert cnt.closer_hits > 0
        ds = dev.DeviceScene(d)
        try:
            for trav in TRAVERSALS:
                assert_bit_equal(ds.render(p, traversal=trav), want, kind)
        finally:
            ds.close()


def test_axis_aligned_rays_and_degenerate_geometry(oracle):
    """Rays with exactly-zero direction components (1/0 = inf, 0*inf = NaN in the slab test, bbox.cuh:36-38),
    origins lying exactly on box planes, a zero-area triangle and a zero-radius sphere."""
    hs = HostScene()
    m = hs.add_material(PT_MAT_DIFFUSE, (0.8, 0.8, 0.8))
    quad = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32)
    hs.add_mesh(quad, np.array([[0, 1, 2], [0, 2, 3]], np.int32), m)
    hs.add_mesh(quad + np.float32([0, 0, -1]), np.array([[0, 1, 2], [0, 2, 3]], np.int32), m)
    hs.add_mesh(np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1]], np.float32), np.array([[0, 1, 2]], np.int32), m,
                normals=np.array([[0, 0, 1]] * 3, np.float32))                      # zero-area triangle
    hs.add_sphere((0.5, 0.5, 0.5), 0.0, m)                                          # zero-radius sphere
    hs.add_sphere((-0.5, 0.25, 0.5), 0.25, m)
    d = hs.finalize()
    rays = []
    for ox in (-1.0, -0.5, 0.0, 0.25, 1.0):            # origins on the planes x=-1, 0, 1 of the boxes
        for oy in (-1.0, 0.0, 0.5, 1.0):
            for dz in (-1.0, 1.0):
                rays.append([ox, oy, 2.0 * -dz, 0.0, 0.0, dz, 0.0, np.inf])
                rays.append([ox, oy, 0.0, 0.0, 0.0, dz, 1e-4, 3.4e38])
            rays.append([ox, -3.0, 0.5, 0.0, 1.0, 0.0, 0.0, np.inf])               # grazing along y inside plane x=ox
            rays.append([-3.0, oy, 0.5, 1.0, 0.0, 0.0, 0.0, np.inf])
            rays.append([ox, oy, 0.5, -0.0, 0.0, -1.0, 0.0, np.inf])               # negative zero component
    rays = np.array(rays, dtype=np.float32)
    tuv, prim = oracle.intersect(d, rays)
    assert (prim >= 0).sum() > 20
    ds = dev.DeviceScene(d)
    try:
        for trav in TRAVERSALS:
            t2, p2 = ds.intersect(rays, traversal=trav)
            assert (p2 == prim).all(), trav
            assert_bit_equal(t2, tuv, f"degenerate trav {trav}")
        hs.set_camera((0, 0, 4), (0, 0, 0), (0, 1, 0), 40, 33, 33, 3)       # odd size: centre pixel looks straight down -z
        p = hs.render_params()
        assert_bit_equal(ds.render(p), oracle.render(d, p)[0], "degenerate render")
    finally:
        ds.close()


def test_deep_unbalanced_bvh_uses_the_lds_stack(oracle):
    """A hand-made 'caterpillar' BVH 40 levels deep (the reference's stack cap is 64, scene.h:251)."""
    import ctypes as C

    from pathtracer_cuda_interactive_amd.ctypes_defs import PtBvhNode, PtSceneDesc
    n = 40
    hs = HostScene()
    hs.set_camera((0, 0, 6), (0, 0, 0), (0, 1, 0), 60, 40, 24, 3)
    m = hs.add_material(PT_MAT_DIFFUSE, (0.6, 0.7, 0.8))
    for k in range(n):
        hs.add_sphere((-4 + 8 * k / (n - 1), np.sin(k) * 1.5, np.cos(k * 1.7)), 0.3, m)
    d = hs.finalize()
    leaves = {int(nd["prim"]): nd for nd in hs.nodes_array() if nd["prim"] != -1}
    nodes = (PtBvhNode * (2 * n - 1))()
    for k in range(n):                                     # leaves 0..n-1
        nodes[k] = PtBvhNode(tuple(leaves[k]["bmin"]), tuple(leaves[k]["bmax"]), -1, -1, k)
    prev = 0
    for k in range(1, n):                                  # inner node n-1+k = (chain so far, leaf k)
        a, b = nodes[prev], nodes[k]
        lo = tuple(min(a.bmin[i], b.bmin[i]) for i in range(3))
        hi = tuple(max(a.bmax[i], b.bmax[i]) for i in range(3))
        nodes[n - 1 + k] = PtBvhNode(lo, hi, prev, k, -1)
        prev = n - 1 + k
    d2 = PtSceneDesc()
    C.memmove(C.byref(d2), C.byref(d), C.sizeof(PtSceneDesc))
    d2.nodes, d2.num_nodes, d2.root = nodes, 2 * n - 1, prev
    p = hs.render_params()
    want, cnt = oracle.render(d2, p)
    ds = dev.DeviceScene(d2)
    try:
        assert ds.info("bvh_depth") == n
        for trav in TRAVERSALS:
            assert_bit_equal(ds.render(p, traversal=trav), want, "caterpillar")
    finally:
        ds.close()


def test_invalid_descriptors_and_params_return_status_codes(dscenes):
    import ctypes as C

    from pathtracer_cuda_interactive_amd.ctypes_defs import PtSceneDesc
    hs, d = load_scene("cbox")

    def broken(**kw):
        b = PtSceneDesc()
        C.memmove(C.byref(b), C.byref(d), C.sizeof(PtSceneDesc))
        for k, v in kw.items():
            setattr(b, k, v)
        return b
    for bad in (broken(root=d.num_nodes), broken(num_nodes=d.num_nodes - 1), broken(num_shapes=0), broken(num_materials=0)):
        with pytest.raises(PtError) as e:
            dev.DeviceScene(bad)
        assert e.value.status == PT_ERR_BAD_SCENE and len(str(e.value)) > 20
    ds = dscenes("cbox")
    for field, val in (("spp", 0), ("width", 0), ("row_end", 10 ** 6), ("traversal", 9), ("sample_offset", -1)):
        p = hs.render_params(16, 16, 2)
        setattr(p, field, val)
        with pytest.raises(PtError) as e:
            ds.render(p)
        assert e.value.status == PT_ERR_INVALID_ARG, field
    with pytest.raises(PtError) as e:
        ds.set_option("no_such_option", 1)
    assert e.value.status == PT_ERR_INVALID_ARG
    # PCG stream = pixel*stride + sample_offset + s: samples beyond the stride would replay the next pixel's streams
    p = hs.render_params(16, 16, 4)
    p.sample_offset, p.stream_stride = 6, 8                       # 6 + 4 > 8
    with pytest.raises(PtError) as e:
        ds.render(p)
    assert e.value.status == PT_ERR_INVALID_ARG and "stream_stride" in str(e.value)
    import torch
    acc = torch.zeros(16 * 16 * 3, device="cuda")
    p = hs.render_params(16, 16, 2)                               # what pt_host_default_params produces: stride 0 -> spp
    ds.accumulate_into(p, acc.data_ptr())                         # first frame (offset 0): fine
    p.sample_offset = 2                                           # second frame with the default stride: streams would overlap
    with pytest.raises(PtError) as e:
        ds.accumulate_into(p, acc.data_ptr())
    assert e.value.status == PT_ERR_INVALID_ARG and "explicit stream_stride" in str(e.value)
    p.stream_stride = 1 << 16
    ds.accumulate_into(p, acc.data_ptr())                         # explicit upper bound: accepted
    torch.cuda.synchronize()
    import torch
    buf = torch.zeros(16 * 16 * 3, device="cuda")
    ds.set_option("scratch_bytes", 16 * 16 * 16)
    p = hs.render_params(16, 16, 5)
    with pytest.raises(PtError) as e:
        ds.accumulate_into(p, buf.data_ptr())
    ds.set_option("scratch_bytes", 0)
    assert e.value.status == PT_ERR_UNSUPPORTED


def test_scene_handles_release_their_device_memory():
    """pt_scene_destroy frees everything the handle owns (scene arrays, per-sample scratch, counters, events): creating,
    rendering and destroying many handles must not eat device memory (the reference never frees, scene.h:144-171)."""
    import torch
    hs, d = load_scene("teapot")
    p = hs.render_params(96, 64, 4)

    def cycle(n):
        for _ in range(n):
            ds = dev.DeviceScene(d)
            ds.render(p)
            ds.close()

    cycle(3)                                        # warm-up: allocator pools, code objects
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(60)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 60 create/destroy cycles"


@pytest.mark.gpu
def test_frame_times_of_frames_enqueued_without_a_host_sync(oracle):
    """pt_get_frame_times: the library keeps the HIP events of the last `timing_frames` render calls, so a caller may enqueue
    frame after frame on a stream (bench.py's timed loop) and read every frame's kernel time afterwards."""
    import torch
    hs, d = load_scene("cbox")
    p = hs.render_params(160, 120, 4)
    want, _ = oracle.render(d, p)
    ds = dev.DeviceScene(d)
    try:
        ds.set_option("timing_frames", 5)
        out = torch.empty((120, 160, 3), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(7):
            ds.render_into(p, out.data_ptr(), stream)
        k, r = ds.frame_times(16)
        assert len(k) == 5 and len(r) == 5 and (k > 0).all() and (r > 0).all()
        k2, _ = ds.frame_times(2)
        assert len(k2) == 2 and np.array_equal(k2, k[-2:])
        c = ds.counters()
        assert c.kernel_ms == k[-1]
        assert_bit_equal(out.cpu().numpy(), want, "last of seven frames enqueued back to back")
        with pytest.raises(PtError):
            ds.set_option("timing_frames", -1)
        # 0 = no timing events: same frame, same work counters, times read zero
        ds.set_option("timing_frames", 0)
        out.zero_()
        for _ in range(3):
            ds.render_into(p, out.data_ptr(), stream)
        c0 = ds.counters()
        assert (c0.paths, c0.segments) == (c.paths, c.segments) and c0.kernel_ms == 0 and c0.resolve_ms == 0
        assert_bit_equal(out.cpu().numpy(), want, "frames rendered without timing events")
        ds.set_option("timing_frames", 2)
        ds.render_into(p, out.data_ptr(), stream)
        assert ds.counters().kernel_ms > 0
    finally:
        ds.close()
