"""Per-ray closest-hit KATs: device traversal vs golden vectors and the live oracle."""
import glob
import os

import numpy as np
import pytest
from conftest import GOLDEN, assert_bit_equal, load_scene

from pathtracer_cuda_interactive_amd import PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED
from pathtracer_cuda_interactive_amd import device as dev

pytestmark = pytest.mark.gpu
RAYS = sorted(glob.glob(os.path.join(GOLDEN, "rays", "*.npz")))


@pytest.mark.parametrize("path", RAYS, ids=os.path.basename)
def test_device_reproduces_ray_kats(path):
    name = os.path.basename(path)[:-4]
    _, d = load_scene(name)
    z = np.load(path)
    ds = dev.DeviceScene(d)
    try:
        for trav in (PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED):
            tuv, prim = ds.intersect(z["rays"], traversal=trav)
            assert (prim == z["prim"]).all(), f"{name} trav {trav}: {(prim != z['prim']).sum()} prims differ"
            assert_bit_equal(tuv, z["tuv"], f"{name} trav {trav}")
    finally:
        ds.close()


@pytest.mark.parametrize("name", ["bunny", "teapot"])
def test_many_random_rays_against_live_oracle(oracle, name):
    hs, d = load_scene(name)
    rng = np.random.default_rng(5)
    n = 20000
    o = (rng.standard_normal((n, 3)) * 2).astype(np.float32)
    t = (rng.standard_normal((n, 3)) * 0.7).astype(np.float32)
    dirs = t - o
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays = np.concatenate([o, dirs.astype(np.float32), np.full((n, 1), 1e-4, np.float32),
                           np.full((n, 1), 3.0e38, np.float32)], axis=1).astype(np.float32)
    tuv, prim = oracle.intersect(d, rays)
    assert (prim >= 0).mean() > 0.2
    ds = dev.DeviceScene(d)
    try:
        for trav in (PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED):
            t2, p2 = ds.intersect(rays, traversal=trav)
            assert (p2 == prim).all()
            assert_bit_equal(t2, tuv, name)
    finally:
        ds.close()
