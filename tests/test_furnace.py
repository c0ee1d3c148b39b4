"""White-furnace check of the four BSDFs as the oracle restates them (reference scene.h:364-464).  PLASTIC is used by no scene
the reference ships, so no reference-held image pins it; what CAN be pinned is a property of the reference's own
formulas: with reflectance 1 the sampled weight value / pdf is 1 for DIFFUSE, F0 + (1 - F0)(1 - cos)^5 = 1 for MIRROR, and for
PLASTIC the specular lobe (weight 1, probability F) plus the diffuse lobe ((1 - F) rho c / ((1 - F.x) c) = 1) — so a convex
object lit by a constant background must show exactly the background.  A wrong Fresnel normalisation, a missing 1/pi or a
wrong pdf in the restatement breaks this.  PHONG loses the part of its lobe that points below the surface: darker, never brighter."""
import numpy as np
import pytest

from pathtracer_cuda_interactive_amd import PT_MAT_DIFFUSE, PT_MAT_MIRROR, PT_MAT_PHONG, PT_MAT_PLASTIC, HostScene


def furnace(mat, **kw):
    hs = HostScene()
    hs.set_camera((0, 0, 4.0), (0, 0, 0), (0, 1, 0), 40.0, 48, 48, 8)
    hs.set_background((0.5, 0.5, 0.5))
    hs.add_sphere((0, 0, 0), 1.0, hs.add_material(mat, (1.0, 1.0, 1.0), **kw))
    return hs, hs.finalize(0)


@pytest.mark.parametrize("mat,kw", [(PT_MAT_DIFFUSE, {}), (PT_MAT_MIRROR, {}), (PT_MAT_PLASTIC, {"eta": 1.5}), (PT_MAT_PLASTIC, {"eta": 2.4})])
@pytest.mark.parametrize("flavour", ["det", "libm"])
def test_a_white_object_in_a_furnace_shows_the_background(oracle, mat, kw, flavour):
    hs, d = furnace(mat, **kw)
    p = hs.render_params(48, 48, 8)
    img, cnt = oracle.render(d, p, math_mode=oracle.MATH_DET if flavour == "det" else oracle.MATH_LIBM)
    assert cnt.segments > cnt.paths                       # the sphere is hit
    assert np.abs(img - 0.5).max() < 2e-6, float(np.abs(img - 0.5).max())
    if mat == PT_MAT_PLASTIC:
        assert cnt.rng_draws > 3 * cnt.paths              # both lobes were sampled (1 draw for the lobe, 2 for the diffuse direction)


def test_phong_in_a_furnace_only_loses_energy(oracle):
    hs, d = furnace(PT_MAT_PHONG, exponent=10.0)
    img, _ = oracle.render(d, hs.render_params(48, 48, 64))
    assert img.max() <= 0.5 + 2e-6 and img[24, 24].mean() > 0.45 and img.min() >= 0.0
    assert img.mean() < 0.5                               # grazing pixels lose the part of the lobe below the surface


@pytest.mark.gpu
def test_furnace_on_the_device_is_the_oracles(oracle):
    from conftest import assert_bit_equal
    from pathtracer_cuda_interactive_amd import device as dev
    for mat, kw in ((PT_MAT_PLASTIC, {"eta": 1.5}), (PT_MAT_MIRROR, {}), (PT_MAT_DIFFUSE, {}), (PT_MAT_PHONG, {"exponent": 10.0})):
        hs, d = furnace(mat, **kw)
        p = hs.render_params(48, 48, 8)
        want, _ = oracle.render(d, p)
        ds = dev.DeviceScene(d)
        try:
            for kernel in (2, 3):
                ds.set_option("kernel", kernel)
                assert_bit_equal(ds.render(p), want, f"furnace material {mat} kernel {kernel}")
        finally:
            ds.close()
