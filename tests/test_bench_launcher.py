"""bench.py's multi-rank control flow on CPU: `python bench.py --gpus 2` must start its own ranks (torch.distributed.run
as a child), shard the rows, gather them to rank 0, all-reduce the timings and print ONE JSON line — exercised here with
the --stub-renderer test hook (gloo, CPU tensors; it renders nothing, so only structure is asserted, never numbers)."""
import json
import os
import subprocess
import sys

import pytest
from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def run_bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("n", [2, 3])
def test_bench_starts_its_own_ranks_and_prints_one_json_line(n):
    r = run_bench(["--gpus", str(n), "--steps", "2", "--warmup", "1", "--stub-renderer"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # exactly one line on stdout: rank 0's JSON
    out = json.loads(lines[0])
    assert out["stub"] is True and out["n_gpus"] == n and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["unit"] == "Msamples/s" and out["higher_is_better"] is True
    cfg = out["config"]
    assert cfg["spp"] == 64 * n                            # weak scaling: per-rank paths constant
    assert cfg["paths_per_step"] == 640 * 480 * 64 * n     # summed over ranks: every row rendered exactly once
    assert cfg["frame_rows_ok"] is True                    # gather + de-interleave put row j at position j
    assert "roofline" in out and "cpu_baseline" not in out


def test_world_size_mismatch_is_an_error():
    r = run_bench(["--gpus", "2", "--stub-renderer"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)


def test_without_a_gpu_the_real_renderer_refuses_to_run():
    import ctypes
    try:
        n = ctypes.c_int(0)
        if ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
            pytest.skip("a GPU is present")
    except OSError:
        pass
    r = run_bench(["--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
