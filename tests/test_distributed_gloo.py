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
