"""The committed golden vectors (tests/golden/, made by make_golden_vectors.py) against a fresh oracle
run: guards the oracle + host pipeline against drift, so the GPU tests can trust the files."""
import glob
import os
import re

import numpy as np
import pytest
from conftest import GOLDEN, assert_bit_equal, load_scene

IMAGES = sorted(glob.glob(os.path.join(GOLDEN, "images", "*.npy")))
RAYS = sorted(glob.glob(os.path.join(GOLDEN, "rays", "*.npz")))
BVH = sorted(glob.glob(os.path.join(GOLDEN, "bvh", "*_nodes.npy")))


def parse_image_name(path):
    m = re.match(r"(.+)_(\d+)x(\d+)_spp(\d+)\.npy", os.path.basename(path))
    return m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))


def test_goldens_present():
    assert len(IMAGES) >= 6 and len(RAYS) >= 6 and len(BVH) >= 2


@pytest.mark.parametrize("path", IMAGES, ids=os.path.basename)
def test_oracle_reproduces_golden_image(oracle, path):
    name, w, h, spp = parse_image_name(path)
    hs, d = load_scene(name)
    img, _ = oracle.render(d, hs.render_params(w, h, spp))
    assert_bit_equal(img, np.load(path), name)


@pytest.mark.parametrize("path", RAYS, ids=os.path.basename)
def test_oracle_reproduces_ray_kats(oracle, path):
    name = os.path.basename(path)[:-4]
    _, d = load_scene(name)
    z = np.load(path)
    tuv, prim = oracle.intersect(d, z["rays"])
    assert (prim == z["prim"]).all()
    assert_bit_equal(tuv, z["tuv"], name)
    assert (prim >= 0).sum() > (40 if name == "aabb_test" else 100)          # the KAT actually exercises hits (aabb_test: 60 small triangles)
    # libm-mode traversal involves no transcendental: identical hits
    tuv2, prim2 = oracle.intersect(d, z["rays"], math_mode=oracle.MATH_LIBM)
    assert (prim2 == prim).all()
    assert_bit_equal(tuv2, tuv, name + " libm")


@pytest.mark.parametrize("path", BVH, ids=os.path.basename)
def test_bvh_builder_reproduces_golden_nodes(path):
    name = os.path.basename(path)[: -len("_nodes.npy")]
    hs, _ = load_scene(name)
    got, want = hs.nodes_array(), np.load(path)
    assert got.shape == want.shape
    for f in ("left", "right", "prim"):
        assert (got[f] == want[f]).all(), f
    for f in ("bmin", "bmax"):
        assert_bit_equal(got[f], want[f], f)
