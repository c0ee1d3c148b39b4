"""The only REFERENCE-HELD evidence of this renderer's output: four screenshots the reference ships in its own display
encoding (/root/reference/sample_images/cbox.png, scene1_phong.png, bunny.png and output/img.png), reduced to block means by
tests/golden/make_reference_image_pins.py (committed as tests/golden/reference_image_pins.json — data, not the PNGs).

What this pins, and what it does not: the reference images were rendered by the CUDA build with cuRAND XORWOW and an
unknown (large) number of accumulated samples, so agreement can only be STATISTICAL.  The estimator restated by the
oracle (LIBM flavour, the reference's per-pixel RNG use; and the DET flavour the GPU is checked against) and computed
by the HIP path must converge to the same image: diffuse + Phong + mirror materials (scene4: 30 spheres, 8 of them
mirrors with their Schlick weights and inter-reflections), spheres + triangles, triangle and sphere
area lights, a 288,094-primitive scene (two instances of a 144,046-triangle PLY mesh: parser, vertex normals, BVH build
and the global-memory traversal path), background, camera.  Block means agree to a fraction of one 8-bit display step on average and to a few steps in the
worst block (Monte-Carlo noise of our finite render; the reference's 8-bit quantisation).  Plastic is NOT covered by any
reference-held image (no shipped scene uses it).

Encoding (opengl_display.cpp:99-117): d = int(255.99 * clamp(sqrt(mean radiance), 0, 1)) per pixel.  Both sides are
compared as block means of the clamped LINEAR pixel values, expressed in display units 255.99 * sqrt(.), so that the
concave sqrt does not bias a render at another resolution or sample count.  Blocks that contain saturated pixels
(the light, the Phong highlights' cores) compare loosely when the resolution differs: the clamp acts per pixel."""
import json
import os

import numpy as np
import pytest
from conftest import GOLDEN, load_scene

PINS = json.load(open(os.path.join(GOLDEN, "reference_image_pins.json")))["images"]

# per-block tolerances in 8-bit display steps (of 255)
TOL_BLOCK = 6.0          # worst unsaturated block (measured: 3.2 - 5.4 at the CPU test's 256 spp)
TOL_BLOCK_SAT = 16.0     # blocks with saturated pixels, rendered at another resolution (measured: 12.2)
TOL_RMS = 1.5            # rms over the unsaturated blocks (measured: 0.63 - 0.85)
TOL_GLOBAL_REL = 0.01    # mean of the whole image per channel, relative (measured: <= 0.35 %)


def block_means(img, gy, gx):
    h, w = img.shape[:2]
    assert h % gy == 0 and w % gx == 0, (h, w, gy, gx)
    return np.clip(img.astype(np.float64), 0.0, 1.0).reshape(gy, h // gy, gx, w // gx, 3).mean(axis=(1, 3))


def compare(name, img, same_resolution):
    pin = PINS[name]
    gy, gx = pin["grid"]
    ref = np.array(pin["lin"], dtype=np.float64)
    sat = np.array(pin["sat"], dtype=np.float64)
    ours = block_means(img, gy, gx)
    dd = 255.99 * (np.sqrt(ours) - np.sqrt(ref))
    worst = np.abs(dd).max(axis=2)
    unsat = sat == 0
    stats = {"worst_unsat": float(worst[unsat].max()), "worst_all": float(worst.max()),
             "rms_unsat": float(np.sqrt((dd[unsat] ** 2).mean())),
             "global_rel": [float(v) for v in (ours.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))) / ref.mean(axis=(0, 1))]}
    assert stats["worst_unsat"] <= TOL_BLOCK, (name, stats)
    assert stats["worst_all"] <= (TOL_BLOCK if same_resolution else TOL_BLOCK_SAT), (name, stats)
    assert stats["rms_unsat"] <= TOL_RMS, (name, stats)
    assert max(abs(v) for v in stats["global_rel"]) <= TOL_GLOBAL_REL, (name, stats)
    return stats


def test_pins_carry_the_display_encoding_of_the_background():
    # corner pixels see only the default 0.5 background: int(255.99 * sqrt(0.5)) = 181 (opengl_display.cpp:105-111)
    # (bunny.xml sets its own background of 0.25: int(255.99 * sqrt(0.25)) = 127)
    for name, pin in PINS.items():
        assert pin["corner_pixel"] == ([127, 127, 127] if name == "bunny" else [181, 181, 181])
    assert int(255.99 * np.sqrt(np.float32(0.5))) == 181 and int(255.99 * np.sqrt(np.float32(0.25))) == 127


# Low resolution x many samples: one of our pixels covers 4 x 4 reference pixels, so the frame costs 1/16 of the
# paths and what is left of the Monte-Carlo noise is far below one display step per block.
CPU_CASES = {"cbox": (256, 256, 256), "scene1_phong": (320, 240, 256), "bunny": (320, 240, 48), "scene4": (320, 240, 128)}


@pytest.mark.parametrize("name", sorted(CPU_CASES))
@pytest.mark.parametrize("flavour", ["libm_per_pixel_rng", "det_per_sample_rng"])
def test_oracle_converges_to_the_reference_screenshot(oracle, name, flavour):
    """LIBM + per-pixel RNG = the reference's host semantics (main.cu:36-47, glibc sin/cos/pow); DET + per-(pixel,
    sample) RNG = the arithmetic contract the GPU is held to.  Both must be the same estimator as the CUDA build's."""
    w, h, spp = CPU_CASES[name]
    hs, d = load_scene(PINS[name]["scene"])
    p = hs.render_params(w, h, spp)
    if flavour == "libm_per_pixel_rng":
        img, _ = oracle.render(d, p, math_mode=oracle.MATH_LIBM, rng_mode=oracle.RNG_PER_PIXEL)
    else:
        img, _ = oracle.render(d, p, math_mode=oracle.MATH_DET, rng_mode=oracle.RNG_PER_SAMPLE)
    assert (img[0, 0] == np.float32(0.25 if name == "bunny" else 0.5)).all()
    compare(name, img, same_resolution=False)


# GPU leg: the screenshot's own resolution at 8,192 spp — our own Monte-Carlo noise is then far below one display step per block
# (cbox: 8.6 G paths, 1.5 s on one MI355X), and what is left in the residual is the reference's noise and its 8-bit quantisation.
# Tolerances = 1.25 x the residuals measured at that sample count (profiles/r03_reference_image_residuals.log), per screenshot:
#                 worst block (display steps)   rms over blocks   |mean| relative
GPU_SPP = 8192
GPU_TOL = {                                  # measured: worst / rms / |mean|        x 1.25
    "bunny":        (0.62, 0.315, 0.0028),    # 0.494 / 0.251 / 0.22 %
    "cbox":         (3.50, 0.275, 0.0020),    # 2.793 (a block on the light's edge; unsaturated blocks 1.07) / 0.220 / 0.15 %
    "scene1_phong": (0.72, 0.390, 0.0043),    # 0.569 / 0.311 / 0.34 %
    "scene4":       (0.61, 0.433, 0.0050),    # 0.488 / 0.346 / 0.39 %
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CPU_CASES))
def test_device_converges_to_the_reference_screenshot(name):
    """The HIP path at the screenshot's own resolution (1024x1024 / 1280x960 / 640x480): every block, saturated or not (the clamp
    acts on the same pixel footprints as in the reference), within 1.25 x the residual measured at this sample count."""
    from pathtracer_cuda_interactive_amd import device as dev
    pin = PINS[name]
    hs, d = load_scene(pin["scene"])
    p = hs.render_params(pin["width"], pin["height"], GPU_SPP)
    ds = dev.DeviceScene(d)
    try:
        img = ds.render(p)
    finally:
        ds.close()
    assert (img[0, 0] == np.float32(0.25 if name == "bunny" else 0.5)).all()
    stats = compare(name, img, same_resolution=True)
    print("RESIDUALS", name, GPU_SPP, "spp:", stats)
    tol_block, tol_rms, tol_mean = GPU_TOL[name]
    if tol_block is not None:
        assert stats["worst_all"] <= tol_block and stats["rms_unsat"] <= tol_rms, (name, stats)
        assert max(abs(v) for v in stats["global_rel"]) <= tol_mean, (name, stats)


# ---- can the pin fail?  Mutation tests (oracle-side, test-only switches: oracle/pt_oracle.h pt_oracle_set_mutation) -------------
#
# A statistical pin is only worth its detection power.  Each mutation below is a plausible restatement ERROR; the pin must
# reject it on at least one screenshot — or the table says that it cannot.  (Measured at the CPU cases' sample counts,
# statistics as printed by compare(); tolerances of the passing estimator: worst block 6, rms 1.5, mean 1 %.)
#
#   mutation (reference lines)                         cbox                     scene1_phong           scene4                 verdict
#   diffuse value without 1/pi (scene.h:370-375)       worst 233, mean +100..390 %   worst 199              worst 170              caught everywhere
#   roulette survivors not re-weighted                 worst 14.8, rms 1.69,    worst 5.6 (passes)     unchanged (passes)     caught by cbox
#     (radiance.cuh:68-74)                             mean(R) -2.5 %
#   Schlick exponent 4 instead of 5 (scene.h:333-336)  unchanged (no mirror)    unchanged              worst 9.6              caught by scene4
#   background 0.45 instead of the default 0.5         corner pixel 171 != 181  same                   same                   caught everywhere
#     (parse_scene.cpp:809)
#   unit_angle "fixed" to pi - 2 asin                  NOT VISIBLE: no normal of any screenshot scene changes (cbox's meshes have
#     (compute_normals.cpp:6, quirk kept: SURVEY H5b)  no obtuse corner between faces of different orientation; bunny.ply and the
#                                                      sphere scenes bring or need no computed normals) — this quirk is pinned
#                                                      by no reference-held image
#   CONTROL: roulette floor 0.25 instead of 0.5        passes                   passes                 passes                 an equally
#                                                      (another estimator of the same expectation must NOT trip the pin)      unbiased variant
MUT_DIFFUSE_NO_INV_PI, MUT_RR_NO_WEIGHT, MUT_SCHLICK_POW4, MUT_RR_FLOOR_QUARTER = 1, 2, 4, 8


def _render_mutated(oracle, name, bits, background=None):
    import ctypes as C
    from pathtracer_cuda_interactive_amd.ctypes_defs import PtSceneDesc
    L = oracle.lib()
    L.pt_oracle_set_mutation.argtypes = [C.c_int]
    L.pt_oracle_set_mutation.restype = C.c_int
    w, h, spp = CPU_CASES[name]
    hs, d = load_scene(PINS[name]["scene"])
    if background is not None:                      # a copy of the description: the cached one serves other tests
        d2 = PtSceneDesc()
        C.memmove(C.byref(d2), C.byref(d), C.sizeof(PtSceneDesc))
        d2.background[:] = background
        d2._keep = d
        d = d2
    p = hs.render_params(w, h, spp)
    L.pt_oracle_set_mutation(bits)
    try:
        img, _ = oracle.render(d, p)
    finally:
        L.pt_oracle_set_mutation(0)
    return img


def _fails(name, img):
    try:
        compare(name, img, same_resolution=False)
    except AssertionError as e:
        return str(e)
    return None


@pytest.mark.parametrize("bits,what,must_fail_on,must_pass_on", [
    (MUT_DIFFUSE_NO_INV_PI, "diffuse lobe without 1/pi", ["cbox", "scene1_phong", "scene4"], []),
    (MUT_RR_NO_WEIGHT, "roulette survivors not re-weighted", ["cbox"], ["scene4"]),
    (MUT_SCHLICK_POW4, "Schlick exponent 4", ["scene4"], ["cbox"]),
    (MUT_RR_FLOOR_QUARTER, "CONTROL: roulette floor 0.25 (same expectation)", [], ["cbox", "scene1_phong", "scene4"]),
])
def test_the_pin_rejects_wrong_estimators(oracle, bits, what, must_fail_on, must_pass_on):
    for name in must_fail_on:
        why = _fails(name, _render_mutated(oracle, name, bits))
        assert why is not None, f"the pin on {name} did not notice: {what}"
        print(f"{what}: rejected by {name}: {why[:200]}")
    for name in must_pass_on:
        why = _fails(name, _render_mutated(oracle, name, bits))
        assert why is None, f"{what} must not trip the pin on {name}: {why}"


def test_the_pin_rejects_a_wrong_default_background(oracle):
    """0.45 instead of parse_scene.cpp:809's 0.5: the corner pixel (pure background) is 171 display steps instead of the
    screenshots' 181, and every statistic is off."""
    for name in ("cbox", "scene4"):
        img = _render_mutated(oracle, name, 0, background=(0.45, 0.45, 0.45))
        assert int(255.99 * np.sqrt(img[0, 0, 0])) == 171 != PINS[name]["corner_pixel"][0]
        assert _fails(name, img) is not None


def test_the_unit_angle_quirk_is_visible_in_no_screenshot_scene():
    """compute_normals.cpp:6 has `(pi - 2) * asin(..)` where Nelson Max's weights need `pi - 2 * asin(..)`; the build keeps it
    (scene_build.cpp).  Would "fixing" it trip the pin?  It cannot: the corrected formula yields the SAME vertex normals for
    every mesh of cbox (each face brings its own vertices, or meets its neighbours at right angles), bunny.ply brings its own
    normals, and scene1_phong / scene4 hold spheres only.  So no reference-held image pins this quirk — recorded here."""
    import ctypes as C

    def unit_angle(u, v, fixed):
        f = np.float32
        s = f(0.5) * np.linalg.norm(v + u, axis=1).astype(f)
        neg = (f(np.pi) - f(2) * np.arcsin(s)) if fixed else (f(np.pi) - f(2)) * np.arcsin(s)
        pos = f(2) * np.arcsin(f(0.5) * np.linalg.norm(v - u, axis=1).astype(f))
        return np.where((u * v).sum(axis=1) < 0, neg, pos).astype(f)

    def unit(x):
        n = np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
        return np.where(n != 0, x / np.where(n == 0, 1, n), 0).astype(np.float32)

    def normals(P, I, fixed):
        v = [P[I[:, k]] for k in range(3)]
        n = np.cross(v[1] - v[0], v[2] - v[0]).astype(np.float32)
        ok = np.linalg.norm(n, axis=1) != 0
        n = unit(n)
        N = np.zeros_like(P)
        for k in range(3):
            a, b, c = v[k], v[(k + 1) % 3], v[(k + 2) % 3]
            np.add.at(N, I[ok, k], (n * unit_angle(unit(b - a), unit(c - a), fixed)[:, None])[ok])
        return unit(N)

    _, d = load_scene("cbox")
    assert d.num_meshes == 8
    for m in range(d.num_meshes):
        me = d.meshes[m]
        P = np.ctypeslib.as_array(me.positions, shape=(me.num_vertices, 3)).copy()
        I = np.ctypeslib.as_array(me.indices, shape=(me.num_faces, 3)).copy()
        N = np.ctypeslib.as_array(me.normals, shape=(me.num_vertices, 3)).copy()
        assert np.abs(normals(P, I, False) - N).max() < 1e-6       # the host library's normals ARE the quirk's
        assert np.abs(normals(P, I, True) - N).max() < 1e-6        # ... and the corrected formula's: nothing to see
    for name in ("scene1_phong", "scene4"):
        assert load_scene(PINS[name]["scene"])[1].num_meshes == 0
