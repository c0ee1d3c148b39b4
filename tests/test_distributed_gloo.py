"""The N>1 path on CPU: world_size 2 and 3 (ragged rows) over gloo.  Each rank renders its interleaved
rows (with the oracle standing in for the GPU, which is allowed in tests), the bands are gathered to
rank 0 and the assembled frame must equal the single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from conftest import REPO, TESTS, load_scene


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, scene, w, h, spp, out_path):
    for p in (REPO, TESTS):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import oracle_binding as ob
    from conftest import load_scene as ls

    from pathtracer_cuda_interactive_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hs, d = ls(scene)
        params = hs.render_params(w, h, spp)

        def render_rows(q):
            assert (q.row_begin, q.row_stride) == (rank, world)
            img, _ = ob.render(d, q, threads=1)
            return torch.from_numpy(img)
        frame = D.render_sharded(render_rows, params)
        if rank == 0:
            np.save(out_path, frame.numpy())
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 24), (3, 25)])
def test_sharded_render_equals_single_process(tmp_path, oracle, world, h):
    hs, d = load_scene("cbox")
    w, spp = 32, 3
    want, _ = oracle.render(d, hs.render_params(w, h, spp))
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), "cbox", w, h, spp, out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape == want.shape
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_shard_helpers():
    from pathtracer_cuda_interactive_amd import distributed as D
    hs, _ = load_scene("scene1")
    p = hs.render_params(8, 10, 1)
    assert [D.rows_of(r, 4, 10) for r in range(4)] == [3, 3, 2, 2]
    q = D.shard_params(p, 2, 4)
    assert (q.row_begin, q.row_end, q.row_stride, q.num_rows()) == (2, 10, 4, 2)
    assert (p.row_begin, p.row_end, p.row_stride) == (0, 0, 0)            # input untouched
    parts = torch.zeros((4, 3, 8, 3))
    for r in range(4):
        parts[r] = r
    full = D.assemble(parts, 10, 8, 4)
    assert full[:, 0, 0].tolist() == [0, 1, 2, 3, 0, 1, 2, 3, 0, 1]
    # world == 1 needs no process group
    out = D.render_sharded(lambda q: torch.full((q.num_rows(), q.width, 3), 7.0), p, rank=0, world=1)
    assert out.shape == (10, 8, 3)


def _gpu_worker(rank, world, port, scene, w, h, spp, out_path):
    """Two ranks share the one GPU of the test box: each renders its interleaved rows with the real ShardedRenderer
    (HIP path, device tensors) and the bands travel through render_sharded's gather.  The collective itself runs over
    gloo here (RCCL refuses two ranks on one device); the nccl backend is what bench.py uses on a multi-GPU node."""
    for p in (REPO, TESTS):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from conftest import load_scene as ls

    from pathtracer_cuda_interactive_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        hs, d = ls(scene)
        params = hs.render_params(w, h, spp)
        R = D.ShardedRenderer(d)
        for _ in range(2):                                  # twice: the band buffer is reused from frame to frame
            frame = D.render_sharded(lambda q: R.render_rows(q).cpu(), params)
        R.close()
        if rank == 0:
            np.save(out_path, frame.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,h", [(2, 48), (3, 50)])
def test_sharded_renderer_ranks_on_the_gpu(tmp_path, oracle, world, h):
    hs, d = load_scene("cbox")
    w, spp = 64, 4
    want, _ = oracle.render(d, hs.render_params(w, h, spp))
    out = str(tmp_path / "frame.npy")
    mp.spawn(_gpu_worker, args=(world, _free_port(), "cbox", w, h, spp, out), nprocs=world, join=True)
    got = np.load(out)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def _nccl_worker(rank, world, port, out_path):
    for p in (REPO, TESTS):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from pathtracer_cuda_interactive_amd import distributed as D
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        # the collectives bench.py issues on the nccl (= RCCL) backend, on a one-rank group: gather of a device tensor,
        # MAX / SUM all-reduce of the timing vector, barrier
        band = torch.arange(6 * 8 * 3, dtype=torch.float32, device="cuda").reshape(6, 8, 3)
        parts = torch.empty((1, 6, 8, 3), dtype=torch.float32, device="cuda")
        dist.gather(band, list(parts.unbind(0)), dst=0)
        t = torch.tensor([1.5, 2.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.barrier()
        torch.cuda.synchronize()
        ok = bool((parts[0] == band).all()) and t.tolist() == [1.5, 2.0]
        frame = D.assemble(parts, 6, 8, 1)
        np.save(out_path, np.array([ok and bool((frame == band).all())]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_backend_runs_the_collectives_bench_uses(tmp_path):
    """A one-GPU box cannot host two RCCL ranks, but it can prove that the nccl backend initialises with device_id and
    executes gather / all_reduce / barrier on device tensors — the calls of the N>1 path (bench.py, distributed.py)."""
    out = str(tmp_path / "ok.npy")
    mp.spawn(_nccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert bool(np.load(out)[0])
