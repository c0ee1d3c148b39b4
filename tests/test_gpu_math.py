"""Device arithmetic (csrc/pt_math.h compiled for gfx950) against the CPU oracle, bit for bit."""
import numpy as np
import pytest
from conftest import assert_bit_equal

from pathtracer_cuda_interactive_amd import device as dev

pytestmark = pytest.mark.gpu


def test_device_sincos_pow_pcg(oracle):
    rng = np.random.default_rng(3)
    x = (rng.random(500000, dtype=np.float32) * np.float32(6.2831855)).astype(np.float32)
    s, c = dev.debug_math(0, x)
    so, co = oracle.sincos(x)
    assert_bit_equal(s, so, "sin")
    assert_bit_equal(c, co, "cos")
    xb = np.concatenate([rng.random(300000, dtype=np.float32), np.array([0.0, 1.0, 1e-30, 1e-40], np.float32)])
    yb = np.concatenate([(rng.random(300000, dtype=np.float32) * 500).astype(np.float32), np.array([3, 3, 2, 1], np.float32)])
    p, _ = dev.debug_math(1, xb, yb)
    assert_bit_equal(p, oracle.powf(xb, yb), "pow")
    assert_bit_equal(p, dev.debug_math(1, xb, yb, host=True)[0], "pow host==device")
    streams = rng.integers(0, 2 ** 31, 1000, dtype=np.uint32)
    seeds = rng.integers(0, 2 ** 31, 1000, dtype=np.uint32)
    a, b = dev.debug_math(2, streams.view(np.float32), seeds.view(np.float32))
    ah, bh = dev.debug_math(2, streams.view(np.float32), seeds.view(np.float32), host=True)
    assert_bit_equal(a, ah, "pcg first")
    assert_bit_equal(b, bh, "pcg second")
    _, f, _ = oracle.pcg(int(streams[0]), int(seeds[0]), 2)
    assert a[0] == f[0] and b[0] == f[1]


def test_device_keeps_fp32_denormals_and_ieee_division():
    """pow(x,y) results in the denormal range must not be flushed (x86 oracle keeps them)."""
    x = np.array([1e-20, 1e-10, 0.001], np.float32)
    y = np.array([2.1, 4.2, 14.5], np.float32)
    p, _ = dev.debug_math(1, x, y)
    ph, _ = dev.debug_math(1, x, y, host=True)
    assert_bit_equal(p, ph, "denormal pow")
    assert (p > 0).all() and (p < 1.2e-38).all()
