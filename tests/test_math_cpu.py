"""The product's arithmetic header (csrc/pt_math.h, host half of libpt_hip.so) against the oracle's
independent restatement (oracle/pt_oracle_math.h): bit-for-bit, on the CPU.  The GPU half is checked
by tests/test_gpu_math.py."""
import os

import numpy as np
from conftest import assert_bit_equal

from pathtracer_cuda_interactive_amd import device as dev

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sincos_bit_exact_and_accurate(oracle):
    rng = np.random.default_rng(7)
    x = np.concatenate([(rng.random(200000, dtype=np.float32) * np.float32(6.2831855)).astype(np.float32),
                        np.array([0.0, 1e-30, 1e-8, np.pi / 4, np.pi / 2, np.pi, 3 * np.pi / 2, 6.2831855, 6.2831850],
                                 dtype=np.float32)])
    s, c = dev.debug_math(0, x, host=True)
    so, co = oracle.sincos(x)
    assert_bit_equal(s, so, "sin")
    assert_bit_equal(c, co, "cos")
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 2e-7


def test_pow_bit_exact_and_accurate(oracle):
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.random(100000, dtype=np.float32), np.array([0.0, 1.0, 1e-20, 0.5, 0.999999], dtype=np.float32)])
    y = np.concatenate([(rng.random(100000, dtype=np.float32) * 300).astype(np.float32),
                        np.array([2.0, 50.0, 3.0, 0.0, 1000.0], dtype=np.float32)])
    p, _ = dev.debug_math(1, x, y, host=True)
    assert_bit_equal(p, oracle.powf(x, y), "pow")
    ref = np.power(x.astype(np.float64), y.astype(np.float64))
    ok = ref > 1e-30
    assert (np.abs(p[ok] - ref[ok]) / ref[ok]).max() < 2e-7
    # exponents the Phong sampler uses: 1/(n+1)
    y2 = (1.0 / (np.float32(1.0) + (rng.random(x.size, dtype=np.float32) * 200))).astype(np.float32)
    p2, _ = dev.debug_math(1, x, y2, host=True)
    assert_bit_equal(p2, oracle.powf(x, y2), "pow small exponent")


def test_pcg_matches_oracle(oracle):
    streams = np.array([0, 1, 2, 307199, 12345678, 0x7fffffff], dtype=np.uint32)
    seeds = np.array([1984, 1984, 0, 1984, 42, 0xffffffff], dtype=np.uint32)
    a, b = dev.debug_math(2, streams.view(np.float32), seeds.view(np.float32), host=True)
    for k in range(len(streams)):
        _, f, _ = oracle.pcg(int(streams[k]), int(seeds[k]), 2)
        assert a[k] == f[0] and b[k] == f[1]


def test_fastdiv_matches_integer_division(tmp_path):
    """start_path maps a work item to (sample, row, column) with multiply-shift divisions (csrc/pt_layout.h FastDiv);
    they must agree with / for every operand below 2^30 — the launch caps work items and pixels at 2^30."""
    import subprocess
    exe = tmp_path / "fastdiv_check"
    src = os.path.join(REPO, "tests", "native", "fastdiv_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(REPO, "pathtracer_cuda_interactive_amd", "csrc"), src, "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
