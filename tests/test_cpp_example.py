"""examples/render_main.cpp: the offline path of the reference's main() on the bare C ABI (no Python in the loop)."""
import subprocess

import numpy as np
import pytest
from conftest import SCENES, _gpu_available, assert_bit_equal, load_scene

from pathtracer_cuda_interactive_amd import _build, read_pfm


@pytest.fixture(scope="module")
def binary():
    _build.build_host()
    return _build.build_example()


@pytest.mark.skipif(_gpu_available(), reason="only meaningful on a box without a GPU")
def test_example_fails_loudly_without_a_gpu(binary, tmp_path):
    r = subprocess.run([binary, f"{SCENES}/cbox.pts", str(tmp_path / "x.pfm"), "16", "12", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr
    assert subprocess.run([binary], capture_output=True).returncode == 2            # usage
    r = subprocess.run([binary, "/nonexistent.xml", "x.pfm"], capture_output=True, text=True)
    assert r.returncode == 1 and "scene load failed" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cbox", "scene1"])
def test_example_renders_the_oracle_image(oracle, binary, tmp_path, name):
    out = tmp_path / "frame.pfm"
    r = subprocess.run([binary, f"{SCENES}/{name}.pts", str(out), "48", "36", "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "GPU rendering took" in r.stdout and "Maximum BVH depth" in r.stdout
    hs, d = load_scene(name)
    want, _ = oracle.render(d, hs.render_params(48, 36, 5))
    assert_bit_equal(read_pfm(out), want, name)


@pytest.mark.gpu
def test_example_with_device_tree_and_nee(oracle, binary, tmp_path):
    """The two extensions through the bare C ABI: the tree built on the GPU, next-event estimation.  The oracle renders on
    the same device-built tree (the builder is deterministic, so the Python binding gets the very same nodes)."""
    from pathtracer_cuda_interactive_amd import PT_RENDER_NEE
    from pathtracer_cuda_interactive_amd import device as dev
    out = tmp_path / "frame.pfm"
    r = subprocess.run([binary, f"{SCENES}/teapot.pts", str(out), "48", "36", "4", "sah", "nee"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "device sah build" in r.stdout
    hs, d = load_scene("teapot")
    d2, _ = dev.build_bvh_device(d, dev.PT_BVH_DEVICE_SAH)
    p = hs.render_params(48, 36, 4)
    p.flags = PT_RENDER_NEE
    want, _ = oracle.render(d2, p)
    assert_bit_equal(read_pfm(out), want, "teapot, device SAH tree, NEE")


@pytest.mark.gpu
@pytest.mark.parametrize("lag", [0, 1, 2])
def test_interactive_example_accumulates_the_oracle_image(oracle, tmp_path, lag):
    """examples/interactive_main.cpp: the reference's interactive loop (render_progressive + display, main.cu:272-344) on the C ABI
    and the HIP runtime.  With the display one or two frames behind, consecutive frames overlap on the GPU; the accumulated image
    is the sum the oracle makes frame by frame, bit for bit, whatever the lag."""
    binary = _build.build_interactive_example()
    out = tmp_path / "accum.pfm"
    W, H, spf, frames = 48, 36, 2, 9
    r = subprocess.run([binary, f"{SCENES}/cbox.pts", str(out), str(W), str(H), str(spf), str(frames), str(lag)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "frames/s" in r.stdout
    hs, d = load_scene("cbox")
    want = None
    for f in range(frames):
        p = hs.render_params(W, H, spf)
        p.sample_offset, p.stream_stride = f * spf, spf * frames
        part, _ = oracle.render(d, p, accumulate=True)
        want = part if f == 0 else (want + part).astype(np.float32)
    want = (want * np.float32(1.0 / np.float32(spf * frames))).astype(np.float32)
    assert_bit_equal(read_pfm(out), want, f"interactive loop, display {lag} frame(s) behind")


@pytest.mark.skipif(_gpu_available(), reason="only meaningful on a box without a GPU")
def test_interactive_example_fails_loudly_without_a_gpu(tmp_path):
    binary = _build.build_interactive_example()
    r = subprocess.run([binary, f"{SCENES}/cbox.pts", str(tmp_path / "x.pfm"), "16", "12", "1", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr
    assert subprocess.run([binary], capture_output=True).returncode == 2            # usage
