"""Multi-GPU decomposition of one frame: one process per GPU, rows interleaved over ranks,
one gather of the row bands to rank 0 (RCCL over xGMI when the backend is "nccl").

The reference has no multi-GPU code; the path shards naturally because pixels are
independent and the PCG stream is keyed by the GLOBAL pixel index, so the assembled frame is
bit-identical for any world size.  Interleaving (rank r renders rows r, r+N, r+2N, ...) balances
sky-only and geometry-heavy rows.  The only exchange step is the final gather:
rows/N x W x 3 floats per rank (e.g. 115 KB per rank for 640x480 on 8 GPUs).
"""
import torch
import torch.distributed as dist


def shard_params(params, rank, world):
    """pt_render_params of rank `rank`: rows rank, rank+world, ... of the full frame."""
    q = params.copy()
    q.row_begin, q.row_end, q.row_stride = int(rank), int(params.height), int(world)
    return q


def rows_of(rank, world, height):
    return len(range(rank, height, world))


def assemble(parts, height, width, world):
    """De-interleave gathered bands [world, max_rows, W, 3] into the frame [H, W, 3]."""
    if height % world == 0:
        # row j = k*world + r sits at parts[r, k]: one transposing copy (one kernel on the GPU instead of `world` of them)
        return parts.transpose(0, 1).reshape(height, width, 3)
    out = torch.empty((height, width, 3), dtype=parts.dtype, device=parts.device)
    for r in range(world):
        n = rows_of(r, world, height)
        out[r::world] = parts[r, :n]
    return out


def render_sharded(render_rows, params, rank=None, world=None, group=None, dst=0, force_collective=False):
    """Render one frame across the process group.

    render_rows(q) -> tensor [rows_of(rank), W, 3] (float32) for the shard params q; on the GPU path it
    wraps DeviceScene.render_into on the current stream, in the CPU tests it wraps the oracle.
    Returns the assembled frame on rank `dst`, None elsewhere.  world == 1 needs no process group — unless
    force_collective asks for the gather + de-interleave all the same (a one-rank group: what that step costs, measured).
    """
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    H, W = int(params.height), int(params.width)
    q = shard_params(params, rank, world)
    mine = render_rows(q)
    if world == 1 and not force_collective:
        return mine
    max_rows = rows_of(0, world, H)
    if mine.shape[0] != max_rows:                      # ragged tail: pad so every rank sends the same size
        pad = torch.zeros((max_rows - mine.shape[0], W, 3), dtype=mine.dtype, device=mine.device)
        mine = torch.cat([mine, pad], dim=0)
    mine = mine.contiguous()
    # One collective, the same on every rank: gather to `dst`.  Both backends this runs on implement it ("nccl" = RCCL
    # as grouped send/recv over xGMI, "gloo" for the CPU tests).  Communication errors propagate: a rank that failed must
    # exit non-zero rather than issue a different collective from its peers.
    if rank == dst:
        parts = torch.empty((world, max_rows, W, 3), dtype=mine.dtype, device=mine.device)
        dist.gather(mine, list(parts.unbind(0)), dst=dst, group=group)
        return assemble(parts, H, W, world)
    dist.gather(mine, None, dst=dst, group=group)
    return None


class ShardedRenderer:
    """GPU front-end: owns a DeviceScene on this rank's GPU and a reusable output band."""

    def __init__(self, desc, device_index=None):
        from .device import DeviceScene
        if device_index is not None:
            torch.cuda.set_device(device_index)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.scene = DeviceScene(desc)
        self._band = None

    def render_rows(self, q):
        rows = q.num_rows()
        if self._band is None or self._band.shape != (rows, q.width, 3):
            self._band = torch.empty((rows, q.width, 3), dtype=torch.float32, device=self.device)
        if rows:
            self.scene.render_into(q, self._band.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return self._band

    def render(self, params, rank=None, world=None, group=None, force_collective=False):
        return render_sharded(self.render_rows, params, rank, world, group, force_collective=force_collective)

    def close(self):
        self.scene.close()
