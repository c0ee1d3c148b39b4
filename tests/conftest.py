import os
import sys

import numpy as np
import pytest

TESTS = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(TESTS)
for p in (REPO, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(TESTS, "golden")
SCENES = os.path.join(GOLDEN, "scenes")
DATA = os.path.join(TESTS, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The native libraries are built in-tree and git-ignored: a fresh checkout has none.  Build what is missing
    # (hipcc cross-compiles gfx950 without a GPU); never rebuild the HIP library that travelled to the GPU box.
    from pathtracer_cuda_interactive_amd import _build
    _build.build_host()
    if not os.path.exists(_build.HIP_LIB):
        _build.build_hip()


def _gpu_available():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests never silently pass on a box without a GPU: they are skipped with a visible reason.
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device here (GPU tests run with -m gpu on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    bad = bits(a) != bits(b)
    if bad.any():
        idx = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.size} floats differ; first at {tuple(idx)}: "
                             f"{a[tuple(idx)]!r} vs {b[tuple(idx)]!r}; max abs diff {np.nanmax(np.abs(a - b)):.3e}")


def assert_work_counters(ds, c, cnt, oracle, d, p, what=""):
    """Device work counters of an EXACT render against the oracle's.  Paths and segments always agree.  Node visits and
    primitive tests agree with the oracle ON THE TREE THE DEVICE TRAVERSED: the caller's, or — where scene creation kept the
    library's internal tree (info fast_tree) and the option is on — that tree, which pt_bvh_build_sweep reproduces for the
    oracle (visit COUNTS do not depend on the visit order: exact traversal never prunes).  Rays with a zero direction
    component leave the internal tree at once and are rerun on the caller's (info redo_segments): they are the slack."""
    assert (c.paths, c.segments) == (cnt.paths, cnt.segments), what
    from pathtracer_cuda_interactive_amd import device as dev
    if ds.info("fast_tree") and ds.info("fast_tree_on") and ds.info("fast_tree_is_callers"):
        # the internal tree is the caller's own topology (the sweep tree did not win the probe): same nodes, visited in another
        # order — the counts are the caller's tree's
        slack = ds.info("redo_segments")
        assert abs(c.node_visits - cnt.inner_pops) <= 256 * slack, (what, c.node_visits, cnt.inner_pops, slack)
        assert abs(c.leaf_tests - (cnt.leaf_tri + cnt.leaf_sphere)) <= 256 * slack, what
    elif ds.info("fast_tree") and ds.info("fast_tree_on"):
        d_int, _ = dev.build_bvh_sweep(d)
        _, cnt_int = oracle.render(d_int, p)
        slack = ds.info("redo_segments")
        assert c.node_visits < cnt.inner_pops, what
        assert abs(c.node_visits - cnt_int.inner_pops) <= 256 * slack, (what, c.node_visits, cnt_int.inner_pops, slack)
        assert abs(c.leaf_tests - (cnt.leaf_tri + cnt.leaf_sphere)) <= 256 * slack, what
        # the same leaves are tested on any tree — by every ray the argument covers (all of them when nothing was rerun)
        assert abs(cnt_int.leaf_tri + cnt_int.leaf_sphere - (cnt.leaf_tri + cnt.leaf_sphere)) <= 256 * slack, (what, slack)
    else:
        assert c.node_visits == cnt.inner_pops and c.leaf_tests == cnt.leaf_tri + cnt.leaf_sphere, what


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    oracle_binding.build()
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def host_lib():
    from pathtracer_cuda_interactive_amd import _build, host
    _build.build_host()
    return host.lib()


_scene_cache = {}


def load_scene(name, sort_mode=None):
    """(HostScene, desc) for a .pts fixture; cached per (name, sort_mode)."""
    from pathtracer_cuda_interactive_amd import PT_BVH_SORT_REFERENCE, HostScene
    mode = PT_BVH_SORT_REFERENCE if sort_mode is None else sort_mode
    key = (name, mode)
    if key not in _scene_cache:
        hs = HostScene.load(os.path.join(SCENES, name + ".pts"))
        d = hs.finalize(mode)
        _scene_cache[key] = (hs, d)
    return _scene_cache[key]


def random_scene(seed, n_tris=40, n_spheres=4, emissive=True):
    """Procedural scene exercising every material and both primitive types."""
    from pathtracer_cuda_interactive_amd import (PT_MAT_DIFFUSE, PT_MAT_MIRROR, PT_MAT_PHONG, PT_MAT_PLASTIC, HostScene)
    rng = np.random.default_rng(seed)
    hs = HostScene()
    hs.set_camera((0, 0.5, 4.0), (0, 0, 0), (0, 1, 0), 50.0, 64, 48, 4)
    hs.set_background((0.4, 0.5, 0.6))
    mats = [hs.add_material(PT_MAT_DIFFUSE, rng.random(3) * 0.8 + 0.1),
            hs.add_material(PT_MAT_MIRROR, rng.random(3) * 0.5 + 0.5),
            hs.add_material(PT_MAT_PLASTIC, rng.random(3) * 0.8 + 0.1, eta=1.3 + float(rng.random()) * 0.5),
            hs.add_material(PT_MAT_PHONG, rng.random(3) * 0.8 + 0.1, exponent=float(rng.integers(2, 80))),
            hs.add_material(PT_MAT_DIFFUSE, (0.0, 0.0, 0.0))]
    if n_tris:
        c = (rng.random((n_tris, 1, 3)) * 4 - 2).astype(np.float32)
        P = (c + (rng.random((n_tris, 3, 3)) - 0.5).astype(np.float32) * 1.5).reshape(-1, 3).astype(np.float32)
        I = np.arange(n_tris * 3, dtype=np.int32).reshape(-1, 3)
        half = n_tris // 2
        hs.add_mesh(P[: half * 3], I[:half], mats[int(rng.integers(0, 4))])
        N = rng.standard_normal((n_tris * 3 - half * 3, 3)).astype(np.float32)
        N /= np.linalg.norm(N, axis=1, keepdims=True)
        hs.add_mesh(P[half * 3:], I[: n_tris - half], mats[int(rng.integers(0, 4))], normals=N,
                    radiance=(3.0, 2.5, 2.0) if emissive else None)
    for k in range(n_spheres):
        hs.add_sphere(rng.random(3) * 3 - 1.5, 0.2 + float(rng.random()) * 0.5, mats[k % 4],
                      radiance=(4.0, 4.0, 1.0) if (emissive and k == 0) else None)
    # big ground sphere so that most paths bounce
    hs.add_sphere((0, -101.5, 0), 100.0, mats[0])
    return hs
