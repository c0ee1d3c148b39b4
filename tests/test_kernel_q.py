"""trace_kernel_q (option "kernel" = 3, csrc/pt_kernel_q.h): paths regrouped across the waves of a workgroup — traversal
waves hand finished paths to shading waves through LDS rings and take whatever ray is ready next.  Which lane or wave runs a
step must never change what is computed for a path: every frame must equal the oracle's (and therefore kernel 2's) bit for
bit, the work counters must agree, and no wait on a ring may run into its bound (the library turns that into PT_ERR_DEVICE).
Hot loop regrouped: /root/reference radiance.cuh:24-75 + scene.h:258-297."""
import numpy as np
import pytest
from conftest import assert_bit_equal, assert_work_counters, load_scene, random_scene

from pathtracer_cuda_interactive_amd import PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED
from pathtracer_cuda_interactive_amd import device as dev

pytestmark = pytest.mark.gpu


def q_scene(d, **opts):
    ds = dev.DeviceScene(d)
    ds.set_option("kernel", 3)
    for k, v in opts.items():
        ds.set_option(k, v)
    return ds


@pytest.mark.parametrize("name,w,h,spp", [("cbox", 96, 72, 9), ("scene1", 80, 60, 12), ("scene1_phong", 64, 64, 10),
                                          ("scene4", 64, 48, 6), ("tetrahedron", 33, 17, 5)])
def test_regrouped_kernel_matches_the_oracle(oracle, name, w, h, spp):
    hs, d = load_scene(name)
    p = hs.render_params(w, h, spp, seed=77)
    want, cnt = oracle.render(d, p)
    ds = q_scene(d, stats=1)
    try:
        img = ds.render(p, traversal=PT_TRAVERSAL_EXACT)
        c = ds.counters()
        assert ds.info("kernel") == 3 and ds.info("lds_scene") == 1
        assert_bit_equal(img, want, name)
        assert_work_counters(ds, c, cnt, oracle, d, p, name)
        for octants in (0, 1):                      # with and without the ray-octant node tables
            ds.set_option("octants", octants)
            assert_bit_equal(ds.render(p), want, f"{name} octants={octants}")
        ds.set_option("fast_tree", 0)               # on the caller's tree
        assert_bit_equal(ds.render(p), want, name + " caller's tree")
        assert ds.info("kernel") == 3
    finally:
        ds.close()


@pytest.mark.parametrize("target,swap,low", [(0, 0, 0), (64, 1, 1), (640, 64, 256), (100000, 8, 64), (200, 40, 1)])
def test_schedule_knobs_never_change_a_bit(oracle, target, swap, low):
    """q_target (paths in flight per workgroup; clamped to what the rings can hold), q_swap (lanes that must want an exchange)
    and q_low (when partial batches are shaded) move work in time and between waves only."""
    hs, d = load_scene("cbox")
    p = hs.render_params(120, 90, 7, seed=3)
    want, cnt = oracle.render(d, p)
    ds = q_scene(d, q_target=target, q_swap=swap, q_low=low)
    try:
        img = ds.render(p)
        c = ds.counters()
        assert_bit_equal(img, want, f"target {target} swap {swap} low {low}")
        assert (c.paths, c.segments) == (cnt.paths, cnt.segments)
    finally:
        ds.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_all_materials_regrouped(oracle, seed):
    """Spheres and triangles, all four materials, emissive shapes: the generic (SPEC = 0) and triangle-only instantiations."""
    hs = random_scene(100 + seed, n_tris=30 + 10 * seed, n_spheres=(0 if seed % 3 == 2 else 4))
    d = hs.finalize()
    p = hs.render_params(72, 54, 5, seed=seed)
    want, cnt = oracle.render(d, p)
    ds = q_scene(d)
    try:
        img = ds.render(p)
        c = ds.counters()
        assert ds.info("kernel") == 3
        assert_bit_equal(img, want, f"random scene {seed}")
        assert (c.paths, c.segments) == (cnt.paths, cnt.segments)
    finally:
        ds.close()


def test_what_the_regrouped_kernel_does_not_serve_runs_on_kernel_2(oracle):
    """Pruned traversal and next-event estimation keep to kernel 2 under option kernel = 3."""
    from pathtracer_cuda_interactive_amd import PT_RENDER_NEE
    hs, d = load_scene("cbox")
    p = hs.render_params(64, 48, 4)
    want, _ = oracle.render(d, p)
    ds = q_scene(d)
    try:
        img = ds.render(p, traversal=PT_TRAVERSAL_PRUNED)
        assert ds.info("kernel") == 2
        assert np.abs(img - want).max() <= 1e-4
        q = p.copy()
        q.flags = PT_RENDER_NEE
        ds.render(q)
        assert ds.info("kernel") == 2
        assert_bit_equal(ds.render(p), want, "cbox back on kernel 3")
        assert ds.info("kernel") == 3
    finally:
        ds.close()


@pytest.mark.parametrize("name,w,h,spp", [("teapot", 50, 40, 3), ("bunny", 40, 30, 2), ("cbox", 64, 48, 4)])
def test_regrouped_kernel_on_scenes_in_global_memory(oracle, name, w, h, spp):
    """Scenes read from global memory (cbox: forced there): 32-bit stacks, the top of the tree in LDS, leaves set aside on the internal
    tree, rings of half the size — on the internal tree, on the caller's tree, and without the top-of-tree cache."""
    hs, d = load_scene(name)
    p = hs.render_params(w, h, spp, seed=77)
    want, cnt = oracle.render(d, p)
    ds = q_scene(d, stats=1)
    try:
        if name == "cbox":
            ds.set_option("force_global", 1)
        img = ds.render(p)
        c = ds.counters()
        assert ds.info("kernel") == 3 and ds.info("lds_scene") == 0
        assert_bit_equal(img, want, name)
        assert_work_counters(ds, c, cnt, oracle, d, p, name)
        ds.set_option("top_cache", 0)
        assert_bit_equal(ds.render(p), want, name + " without the top-of-tree cache")
        ds.set_option("top_cache", 1)
        ds.set_option("fast_tree", 0)
        img = ds.render(p)
        c = ds.counters()
        assert_bit_equal(img, want, name + " caller's tree")
        assert (c.node_visits, c.leaf_tests) == (cnt.inner_pops, cnt.leaf_tri + cnt.leaf_sphere)
        assert ds.info("kernel") == 3
    finally:
        ds.close()


def test_small_launches_shards_passes_and_depth_limits(oracle):
    """Fewer paths than one workgroup holds, row shards, sample passes through a tiny scratch budget, depth limits,
    and two frames in a row on one handle."""
    hs, d = load_scene("cbox")
    ds = q_scene(d)
    try:
        for (w, h, spp) in ((1, 1, 1), (3, 2, 1), (17, 9, 2), (64, 1, 33)):
            p = hs.render_params(w, h, spp, seed=11)
            want, _ = oracle.render(d, p)
            assert_bit_equal(ds.render(p), want, f"{w}x{h}x{spp}")
            assert_bit_equal(ds.render(p), want, f"{w}x{h}x{spp} again")
        p = hs.render_params(80, 60, 6, seed=2)
        want, _ = oracle.render(d, p)
        out = np.zeros_like(want)
        for r in range(3):
            q = p.copy()
            q.row_begin, q.row_end, q.row_stride = r, 60, 3
            out[r::3] = ds.render(q)
        assert_bit_equal(out, want, "3 interleaved shards")
        ds.set_option("scratch_bytes", 80 * 60 * 16 * 2)          # two samples per pass
        assert_bit_equal(ds.render(p), want, "three sample passes")
        assert ds.info("passes") == 3
        ds.set_option("scratch_bytes", 0)
        for max_depth, rr_depth in ((1, 5), (3, 0), (50, 2)):
            q = hs.render_params(64, 48, 5, seed=9)
            q.max_depth, q.rr_depth = max_depth, rr_depth
            want, _ = oracle.render(d, q)
            assert_bit_equal(ds.render(q), want, f"max_depth {max_depth} rr_depth {rr_depth}")
    finally:
        ds.close()


def test_fullsize_cbox_frame_regrouped(oracle):
    """BASELINE.json configs[1] — cbox 640x480 spp=64 — whole frame against the oracle, every float; counters; reruns happen."""
    hs, d = load_scene("cbox")
    p = hs.render_params(640, 480, 64)
    want, cnt = oracle.render(d, p)
    ds = q_scene(d, stats=1)
    try:
        img = ds.render(p)
        c = ds.counters()
        assert_bit_equal(img, want, "cbox 640x480x64 on kernel 3")
        assert_work_counters(ds, c, cnt, oracle, d, p, "cbox full size")
        assert ds.info("redo_segments") > 0
        assert ds.info("block_threads") >= 256
    finally:
        ds.close()
