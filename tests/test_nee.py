"""Next-event estimation (pt_render_params.flags = PT_RENDER_NEE) — an EXTENSION: the reference samples no
light (point lights parsed and unused, light.h:5-8; occlusion query dead, scene.h:306-330), so nothing reference-held
pins it.  What is tested: (1) the estimator is the SAME image as the reference's BSDF-sampling estimator wherever all
light comes from area lights (expectation equal, variance lower); (2) point lights light a scene as the closed form
says; (3) the HIP path equals the oracle's NEE restatement bit for bit on every material, primitive and residency."""
import numpy as np
import pytest
from conftest import assert_bit_equal, load_scene, random_scene

from pathtracer_cuda_interactive_amd import (PT_ERR_INVALID_ARG, PT_ERR_UNSUPPORTED, PT_MAT_DIFFUSE, PT_RENDER_NEE,
                                             PT_TRAVERSAL_PRUNED, HostScene, PtError)


def with_nee(p):
    q = p.copy()
    q.flags = PT_RENDER_NEE
    return q


@pytest.mark.parametrize("name,w,h,spp", [("cbox", 64, 64, 1024), ("scene1_phong", 64, 48, 1024)])
def test_nee_is_the_same_image_with_less_noise(oracle, name, w, h, spp):
    hs, d = load_scene(name)
    p = hs.render_params(w, h, spp)
    a, ca = oracle.render(d, p)
    b, cb = oracle.render(d, with_nee(p))
    assert cb.shadow_rays > 0 and cb.nee_hits > 0 and ca.shadow_rays == 0
    assert cb.emit < ca.emit                                    # emission by BSDF sampling: camera rays / after specular bounces only
    blk = lambda x: x.astype(np.float64).reshape(8, h // 8, 8, w // 8, 3).mean(axis=(1, 3))
    A, B = blk(a), blk(b)
    assert np.abs(A - B).max() < 0.09 * A.mean()                # block means agree within Monte-Carlo noise (measured 0.02-0.045)
    assert abs(a.mean() - b.mean()) < 0.01 * a.mean()           # measured 0.0005-0.002
    # variance: at 16 spp the NEE frame is much closer to the converged image than the plain one (measured ~2x in rmse)
    ref = 0.5 * (a.astype(np.float64) + b)
    lo = hs.render_params(w, h, 16, seed=7)
    x, _ = oracle.render(d, lo)
    y, _ = oracle.render(d, with_nee(lo))
    assert np.sqrt(((y - ref) ** 2).mean()) < 0.75 * np.sqrt(((x - ref) ** 2).mean())


def point_light_scene():
    hs = HostScene()
    hs.set_camera((0, 3, 0.001), (0, 0, 0), (0, 1, 0), 40.0, 32, 32, 1)          # looking straight down on the plane y = 0
    hs.set_background((0, 0, 0))
    m = hs.add_material(PT_MAT_DIFFUSE, (0.6, 0.5, 0.4))
    P = np.float32([[-50, 0, -50], [50, 0, -50], [50, 0, 50], [-50, 0, 50]])
    hs.add_mesh(P, np.int32([[0, 2, 1], [0, 3, 2]]), m, normals=np.float32([[0, 1, 0]] * 4))
    hs.add_point_light((0.5, 2.0, -0.25), (30.0, 20.0, 10.0))
    return hs


def test_point_light_matches_the_closed_form(oracle):
    """One diffuse plane, black background, one point light: without NEE the frame is black (the reference never samples
    point lights); with NEE the radiance of a plane point is rho/pi * I * cos / d^2, exactly (no noise but the
    pixel jitter: a convex scene has no second bounce that finds light)."""
    hs = point_light_scene()
    d = hs.finalize()
    p = hs.render_params(32, 32, 64)
    dark, _ = oracle.render(d, p)
    assert dark.max() == 0.0
    img, cnt = oracle.render(d, with_nee(p))
    assert cnt.nee_hits > 0
    # closed form at the pixel centres
    from pathtracer_cuda_interactive_amd import camera_ray_data
    cam = camera_ray_data(hs.camera, 32, 32).astype(np.float64)
    og, tl, hz, vt = cam
    jj, ii = np.mgrid[0:32, 0:32]
    dirs = tl + hz * ((ii[..., None] + 0.5) / 32) - vt * ((jj[..., None] + 0.5) / 32) - og
    t = -og[1] / dirs[..., 1]
    x = og + dirs * t[..., None]
    L = np.array([0.5, 2.0, -0.25]) - x
    d2 = (L ** 2).sum(-1)
    cos = L[..., 1] / np.sqrt(d2)
    want = (np.array([0.6, 0.5, 0.4]) / np.pi) * np.array([30.0, 20.0, 10.0]) * (cos / d2)[..., None]
    np.testing.assert_allclose(img, want, rtol=0.02)


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,spp", [("cbox", 96, 72, 9), ("scene1_phong", 80, 60, 10), ("scene1", 64, 48, 6),
                                          ("teapot", 48, 36, 3), ("bunny", 40, 30, 2)])
def test_device_nee_matches_the_oracle(oracle, name, w, h, spp):
    from pathtracer_cuda_interactive_amd import device as dev
    hs, d = load_scene(name)
    p = with_nee(hs.render_params(w, h, spp, seed=41))
    want, cnt = oracle.render(d, p)
    ds = dev.DeviceScene(d)
    try:
        img = ds.render(p)
        c = ds.counters()
        assert_bit_equal(img, want, name + " NEE")
        assert (c.paths, c.segments) == (cnt.paths, cnt.segments)      # shadow rays are not "segments"
        ds.set_option("force_global", 1)
        assert_bit_equal(ds.render(p), want, name + " NEE, scene in global memory")
        ds.set_option("force_global", 0)
        ds.set_option("octants", 0)
        assert_bit_equal(ds.render(p), want, name + " NEE, one node table")
        ds.set_option("octants", 1)
        ds.set_option("stats", 1)
        assert_bit_equal(ds.render(p), want, name + " NEE, stats kernel")
        ds.set_option("stats", 0)
        plain = ds.render(hs.render_params(w, h, spp, seed=41))         # the default estimator is untouched
        assert_bit_equal(plain, oracle.render(d, hs.render_params(w, h, spp, seed=41))[0], name + " without NEE")
        q = p.copy()
        q.traversal = PT_TRAVERSAL_PRUNED
        with pytest.raises(PtError) as e:
            ds.render(q)
        assert e.value.status == PT_ERR_UNSUPPORTED
        q = p.copy()
        q.flags = 6
        with pytest.raises(PtError) as e:
            ds.render(q)
        assert e.value.status == PT_ERR_INVALID_ARG
    finally:
        ds.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_device_nee_on_random_scenes_with_every_material_and_both_light_kinds(oracle, seed):
    from pathtracer_cuda_interactive_amd import device as dev
    hs = random_scene(seed, n_tris=30 + 9 * seed, n_spheres=4)       # emissive triangles AND an emissive sphere, all four materials
    hs.add_point_light((1.0, 2.5, 0.5 * seed), (6.0, 5.0, 4.0))
    d = hs.finalize()
    p = with_nee(hs.render_params(56, 40, 6, seed=900 + seed))
    want, cnt = oracle.render(d, p)
    assert cnt.nee_hits > 0 and cnt.shadow_rays > cnt.nee_hits
    ds = dev.DeviceScene(d)
    try:
        for fg in (0, 1):
            ds.set_option("force_global", fg)
            assert_bit_equal(ds.render(p), want, f"seed {seed} global {fg}")
    finally:
        ds.close()
    hs2 = point_light_scene()
    d2 = hs2.finalize()
    p2 = with_nee(hs2.render_params(32, 32, 8))
    ds = dev.DeviceScene(d2)
    try:
        assert_bit_equal(ds.render(p2), oracle.render(d2, p2)[0], "point light plane")
    finally:
        ds.close()
