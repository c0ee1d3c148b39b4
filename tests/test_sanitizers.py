"""AddressSanitizer + UBSan over the host scene pipeline and the oracle (CPU build; GPU ASan is not available)."""
import os
import shutil
import subprocess

import pytest
from conftest import REPO


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_pipeline_and_oracle_are_clean_under_asan_ubsan(tmp_path):
    host = os.path.join(REPO, "pathtracer_cuda_interactive_amd", "csrc", "host")
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith(".cpp")]
    exe = str(tmp_path / "driver")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off",
           f'-DREPO="{REPO}"', f'-DTMP="{tmp_path}"', "-I", os.path.join(REPO, "include"), "-I", os.path.join(REPO, "oracle"),
           os.path.join(REPO, "tests", "sanitize", "driver.cpp")] + srcs + ["-x", "c", os.path.join(REPO, "oracle", "pt_oracle.c"),
           "-lm", "-lpthread", "-pthread", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitizer driver done" in r.stdout
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_sweep_tree_builder_is_clean_under_sanitizers(tmp_path, san):
    """The internal tree's builder (csrc/pt_tree_sweep.h) is host code with a threaded path for big inputs: ASan + UBSan,
    and ThreadSanitizer for the shared arrays its tasks partition among themselves."""
    exe = str(tmp_path / "sweep_driver")
    cmd = ["g++", "-std=c++17", "-g", "-O1", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-ffp-contract=off", "-pthread",
           "-I", os.path.join(REPO, "include"), "-I", os.path.join(REPO, "pathtracer_cuda_interactive_amd", "csrc"),
           os.path.join(REPO, "tests", "sanitize", "sweep_driver.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sweep sanitizer driver done" in r.stdout
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
