"""pt_bvh_build_sweep_device (csrc/pt_sweep_build.hip): the library's internal tree built on the GPU must be the HOST builder's
tree byte for byte (pt_tree_sweep.h is the CPU-side checker; its own trees are pinned by md5 in tests/test_sweep_builder.py) —
every parity result obtained on the host-built tree then carries over.  Reference producer replaced: construct_bvh,
/root/reference/bvh.cu:16-54 (host code there too; README.md:123,132: 10-57 s of start-up)."""
import hashlib
import json
import os

import numpy as np
import pytest
from conftest import GOLDEN, load_scene, random_scene
from test_sweep_builder import _desc_with_leaf_boxes

from pathtracer_cuda_interactive_amd import PtError
from pathtracer_cuda_interactive_amd import device as dev

pytestmark = pytest.mark.gpu


def same_tree(d, what):
    _, host = dev.build_bvh_sweep(d)
    _, devi = dev.build_bvh_sweep(d, on_device=True)
    assert (devi["root"], devi["depth"]) == (host["root"], host["depth"]), what
    a, b = host["nodes"], devi["nodes"]
    if a.tobytes() != b.tobytes():
        bad = np.flatnonzero([x.tobytes() != y.tobytes() for x, y in zip(a, b)])
        raise AssertionError(f"{what}: {len(bad)} of {len(a)} nodes differ; first at {bad[0]}: host {a[bad[0]]} device {b[bad[0]]}")
    return host, devi


@pytest.mark.parametrize("name", ["scene4", "cbox", "teapot", "bunny", "scene1", "tetrahedron"])
def test_device_sweep_tree_is_the_host_builders_tree(name):
    hs, d = load_scene(name)
    host, devi = same_tree(d, name)
    pins = json.load(open(os.path.join(GOLDEN, "pins.json")))["sweep_tree"]
    if name in pins:
        assert hashlib.md5(devi["nodes"].tobytes()).hexdigest() == pins[name]["md5"]
    print(name, d.num_shapes, "prims: host", round(host["build_ms"], 2), "ms, device", round(devi["build_ms"], 3), "ms")


@pytest.mark.parametrize("seed", range(4))
def test_device_sweep_tree_on_random_scenes_and_tied_boxes(seed):
    """Spheres and triangles, duplicate geometry (equal centroids: the order falls to the ids), equal costs (the tie rules),
    zeros of both signs in the boxes (which operand a tie keeps decides the bits of an inner box)."""
    hs = random_scene(seed, n_tris=500 + 300 * seed, n_spheres=5)
    d = hs.finalize(0)
    same_tree(d, f"random scene {seed}")
    n = d.num_shapes
    rng = np.random.default_rng(seed)
    # every box the same -> every cut costs the same: the middle one must win at every node
    lo = np.zeros((n, 3), np.float32)
    hi = np.ones((n, 3), np.float32)
    same_tree(_desc_with_leaf_boxes(hs, d, lo, hi), "all boxes equal")
    # boxes on an integer grid with many exact ties and zeros of both signs
    lo = rng.integers(-3, 3, (n, 3)).astype(np.float32)
    lo[rng.random((n, 3)) < 0.2] = np.float32(-0.0)
    hi = lo + rng.integers(0, 3, (n, 3)).astype(np.float32)
    same_tree(_desc_with_leaf_boxes(hs, d, lo, hi), "integer grid with ties and signed zeros")
    # flat and point boxes
    lo = rng.random((n, 3)).astype(np.float32)
    hi = lo.copy()
    hi[::2, 0] += 0.5
    same_tree(_desc_with_leaf_boxes(hs, d, lo, hi), "flat and point boxes")


def test_one_two_and_three_primitives_on_the_device():
    from pathtracer_cuda_interactive_amd import PT_MAT_DIFFUSE, HostScene
    for n in (1, 2, 3):
        hs = HostScene()
        hs.set_camera((0, 0, 4.0), (0, 0, 0), (0, 1, 0), 45.0, 16, 16, 1)
        m = hs.add_material(PT_MAT_DIFFUSE, (0.5, 0.5, 0.5))
        for k in range(n):
            hs.add_sphere((k * 1.5, 0, 0), 0.5, m)
        same_tree(hs.finalize(0), f"{n} spheres")


def test_input_beyond_the_depth_guard_goes_back_to_the_host_builder():
    """Nested shells, each twice the size of the one before: the cheapest cut peels off the few outermost ones, level after
    level.  The host builder switches such a branch to median cuts past 2 log2 n + 16 levels; the device builder does not hold
    that fallback and says so (pt_scene_create then runs the host builder) — it never returns a different tree."""
    from pathtracer_cuda_interactive_amd import PT_MAT_DIFFUSE, HostScene
    hs = HostScene()
    hs.set_camera((0, 0, 4.0), (0, 0, 0), (0, 1, 0), 45.0, 16, 16, 1)
    m = hs.add_material(PT_MAT_DIFFUSE, (0.5, 0.5, 0.5))
    n = 120
    for k in range(n):
        hs.add_sphere((0, 0, 0), 0.5, m)
    d = hs.finalize(0)
    k = np.arange(n, dtype=np.float64)[:, None]
    raised = 0
    for ratio in (2.0 ** 0.25, 2.0):
        hi = (ratio ** k).astype(np.float32) * np.ones((1, 3), np.float32)
        d2 = _desc_with_leaf_boxes(hs, d, -hi, hi)
        _, host = dev.build_bvh_sweep(d2)
        try:
            _, devi = dev.build_bvh_sweep(d2, on_device=True)
            assert devi["nodes"].tobytes() == host["nodes"].tobytes() and devi["depth"] == host["depth"]
        except PtError as e:
            assert "UNSUPPORTED" in str(e)
            assert host["depth"] > 2 * 7 + 16                 # the host tree did run into the guard (log2 120 -> 7)
            raised += 1
    assert raised >= 1
    hi[7, 1] = np.inf
    with pytest.raises(PtError):
        dev.build_bvh_sweep(_desc_with_leaf_boxes(hs, d, -hi, hi), on_device=True)


def test_scene_creation_builds_its_internal_tree_on_the_device(oracle):
    """From 4,096 shapes up pt_scene_create runs the device builder (info "sweep_on_device"); the frame is the oracle's on the
    CALLER's tree bit for bit, as with the host-built internal tree; the stage timers of scene creation are reported."""
    from conftest import assert_bit_equal, assert_work_counters
    for name in ("teapot", "bunny"):
        hs, d = load_scene(name)
        ds = dev.DeviceScene(d)
        try:
            assert ds.info("sweep_on_device") == 1 and ds.info("fast_tree") == 1
            p = hs.render_params(48, 36, 2, seed=9)
            want, cnt = oracle.render(d, p)
            ds.set_option("stats", 1)
            img = ds.render(p)
            assert_bit_equal(img, want, name)
            assert_work_counters(ds, ds.counters(), cnt, oracle, d, p, name)
            us = [ds.info(f"create_us{k}") for k in range(7)]
            assert us[0] > 0 and sum(us[1:]) <= us[0] * 1.05
            print(name, "pt_scene_create wall us: total, prims, caller's tree, sweep, re-lay, uploads+probe, tie tables =", us)
        finally:
            ds.close()
    hs, d = load_scene("cbox")
    ds = dev.DeviceScene(d)
    try:
        assert ds.info("sweep_on_device") == 0 and ds.info("fast_tree") == 1          # 38 shapes: the host builder
    finally:
        ds.close()


def test_invalid_big_trees_are_refused_by_the_device_side_validation():
    """From 4,096 shapes up the caller's tree is checked on the device (pt_scene_prep.hip: relay_tree_device); a corrupted pool
    must come back as PT_ERR_BAD_SCENE — never as a fault or a hang: child index out of range, a node referenced twice, a
    primitive in two leaves / in none, a cycle, a leaf primitive id out of range."""
    import ctypes as C
    from pathtracer_cuda_interactive_amd import PT_ERR_BAD_SCENE
    from pathtracer_cuda_interactive_amd.ctypes_defs import PtBvhNode, PtSceneDesc
    hs, d = load_scene("teapot")
    base = hs.nodes_array().copy()
    inner = np.flatnonzero(base["prim"] == -1)
    leaf = np.flatnonzero(base["prim"] >= 0)
    root = int(d.root)

    def attempt(nodes):
        d2 = PtSceneDesc()
        C.memmove(C.byref(d2), C.byref(d), C.sizeof(PtSceneDesc))
        d2.nodes = nodes.ctypes.data_as(C.POINTER(PtBvhNode))
        d2._keep = (nodes, d)
        with pytest.raises(PtError) as e:
            dev.DeviceScene(d2).close()
        assert e.value.status == PT_ERR_BAD_SCENE, str(e.value)
        return str(e.value)

    n = base.copy(); n["left"][inner[5]] = len(base) + 7
    assert "out of range" in attempt(n)
    n = base.copy(); n["right"][inner[9]] = -3
    assert "out of range" in attempt(n)
    n = base.copy(); n["prim"][leaf[11]] = d.num_shapes + 5
    assert "primitive id out of range" in attempt(n)
    n = base.copy(); n["left"][inner[20]] = n["right"][inner[20]]                      # one child twice, the other orphaned
    attempt(n)
    n = base.copy(); n["prim"][leaf[3]] = n["prim"][leaf[4]]                           # a primitive in two leaves, another in none
    attempt(n)
    k = next(int(v) for v in inner if v != root and base["left"][v] in inner)          # a cycle: a node becomes its own grandchild
    n = base.copy(); n["left"][n["left"][k]] = k
    attempt(n)
    ds = dev.DeviceScene(d)                                                            # the untouched pool is fine
    assert ds.info("sweep_on_device") == 1
    ds.close()
