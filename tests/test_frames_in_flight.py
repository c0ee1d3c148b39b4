"""Frames in flight (option "frames_in_flight", include/pt_api.h at pt_render_async): the trace kernel of render call k+1 runs on
a stream of the handle's own while call k drains; what the caller sees stays in the order of the caller's stream.  Every frame
must come out bit-identical to the one-slot rendering, whatever the depth, the kernel, or the number of sample passes."""
import os

import numpy as np
import pytest

from pathtracer_cuda_interactive_amd import PT_BVH_SORT_REFERENCE, HostScene

pytestmark = pytest.mark.gpu
SCENES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def bits(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.fixture(scope="module")
def cbox():
    from pathtracer_cuda_interactive_amd import device as dev
    hs = HostScene.load(os.path.join(SCENES, "cbox.pts"))
    d = hs.finalize(PT_BVH_SORT_REFERENCE)
    ds = dev.DeviceScene(d)
    yield hs, d, ds
    ds.close()


def test_default_and_bounds(cbox):
    from pathtracer_cuda_interactive_amd import device as dev
    _, _, ds = cbox
    assert ds.info("frames_in_flight") == 2
    for bad in (0, 5, -1):
        with pytest.raises(dev.PtError):
            ds.set_option("frames_in_flight", bad)
    assert ds.info("frames_in_flight") == 2


@pytest.mark.parametrize("kernel", [2, 3, 1])
def test_frames_identical_at_every_depth(cbox, kernel):
    import torch
    hs, _, ds = cbox
    ds.set_option("kernel", kernel)
    frames = [hs.render_params(96, 64, 6, seed=11 + k) for k in range(9)]
    ds.set_option("frames_in_flight", 1)
    want = [ds.render(p).view(np.uint32).copy() for p in frames]
    stream = torch.cuda.Stream()
    try:
        for depth in (2, 3, 4, 1):
            ds.set_option("frames_in_flight", depth)
            outs = [torch.zeros(64, 96, 3, dtype=torch.float32, device="cuda") for _ in frames]
            torch.cuda.synchronize()
            for p, o in zip(frames, outs):
                ds.render_into(p, o.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
            for k, (o, w) in enumerate(zip(outs, want)):
                assert (bits(o) == w.reshape(64, 96, 3)).all(), f"kernel {kernel} depth {depth} frame {k}"
            c = ds.counters()                                    # the counters are those of the LAST call
            ds.set_option("frames_in_flight", 1)
            ds.render(frames[-1])
            c1 = ds.counters()
            assert (c.paths, c.segments) == (c1.paths, c1.segments)
    finally:
        ds.set_option("kernel", 2)
        ds.set_option("frames_in_flight", 2)


def test_same_output_buffer_reused_every_frame(cbox):
    """The resolve runs on the caller's stream: frame k+1 may not overwrite the buffer before a copy of frame k enqueued on that
    stream has read it."""
    import torch
    hs, _, ds = cbox
    frames = [hs.render_params(80, 48, 4, seed=3 + k) for k in range(8)]
    ds.set_option("frames_in_flight", 1)
    want = [ds.render(p).view(np.uint32).copy().reshape(48, 80, 3) for p in frames]
    ds.set_option("frames_in_flight", 2)
    stream = torch.cuda.Stream()
    out = torch.zeros(48, 80, 3, dtype=torch.float32, device="cuda")
    keep = torch.zeros(len(frames), 48, 80, 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for k, p in enumerate(frames):
            ds.render_into(p, out.data_ptr(), stream=stream.cuda_stream)
            keep[k].copy_(out)
    stream.synchronize()
    for k, w in enumerate(want):
        assert (bits(keep[k]) == w).all(), f"frame {k}"


def test_progressive_accumulation_in_flight(cbox):
    import torch
    hs, _, ds = cbox
    total, per = 24, 2
    res = {}
    for depth in (1, 2, 4):
        ds.set_option("frames_in_flight", depth)
        acc = torch.zeros(48, 64, 3, dtype=torch.float32, device="cuda")
        stream = torch.cuda.Stream()
        torch.cuda.synchronize()
        for s0 in range(0, total, per):
            p = hs.render_params(64, 48, per, seed=77)
            p.sample_offset = s0
            p.stream_stride = total
            ds.accumulate_into(p, acc.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        res[depth] = bits(acc).copy()
    ds.set_option("frames_in_flight", 2)
    assert (res[1] == res[2]).all() and (res[1] == res[4]).all()


def test_sample_passes_and_two_caller_streams(cbox):
    """Frames that need several sample passes stay on the caller's stream; slots are still rotated and guarded by their events,
    also when the caller alternates between two streams of its own."""
    import torch
    hs, _, ds = cbox
    p_big = hs.render_params(64, 48, 16, seed=5)
    p_small = hs.render_params(64, 48, 3, seed=6)
    ds.set_option("frames_in_flight", 1)
    ds.set_option("scratch_bytes", 0)
    want_big = ds.render(p_big).view(np.uint32).copy().reshape(48, 64, 3)
    want_small = ds.render(p_small).view(np.uint32).copy().reshape(48, 64, 3)
    try:
        ds.set_option("frames_in_flight", 3)
        ds.set_option("scratch_bytes", 64 * 48 * 16 * 5)           # 5 samples per pass: p_big runs in 4 passes, p_small in 1
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.zeros(48, 64, 3, dtype=torch.float32, device="cuda") for _ in range(10)]
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            ds.render_into(p_big if k % 3 == 0 else p_small, o.data_ptr(), stream=streams[k % 2].cuda_stream)
            if k % 3 == 0:
                assert ds.info("passes") == 4
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            assert (bits(o) == (want_big if k % 3 == 0 else want_small)).all(), f"frame {k}"
    finally:
        ds.set_option("scratch_bytes", 0)
        ds.set_option("frames_in_flight", 2)


def test_frame_times_of_overlapped_frames(cbox):
    import torch
    hs, _, ds = cbox
    p = hs.render_params(160, 120, 8, seed=2)
    ds.set_option("timing_frames", 6)
    try:
        out = torch.zeros(120, 160, 3, dtype=torch.float32, device="cuda")
        for _ in range(6):
            ds.render_into(p, out.data_ptr())
        k_ms, r_ms = ds.frame_times(6)
        assert len(k_ms) == 6 and all(t > 0 for t in k_ms) and all(t > 0 for t in r_ms)
    finally:
        ds.set_option("timing_frames", 1)


def test_grid_of_a_frame_that_overlaps_with_its_predecessor(cbox):
    """A frame launched while the previous one is still on the GPU leaves room for its successor: all but one block per CU with
    two slots, half of them with three; a frame that finds the GPU idle, or a caller's own blocks_per_cu, takes what it took before."""
    import torch
    hs, _, ds = cbox
    p = hs.render_params(640, 480, 16, seed=9)
    out = torch.zeros(480, 640, 3, dtype=torch.float32, device="cuda")
    ds.set_option("frames_in_flight", 1)
    ds.render_into(p, out.data_ptr())
    torch.cuda.synchronize()
    full = ds.info("blocks_per_cu")
    assert full == ds.info("occupancy") and full >= 2
    want = bits(out).copy()
    try:
        for depth, expect in ((2, full - 1), (3, max(1, full // 2))):
            ds.set_option("frames_in_flight", depth)
            torch.cuda.synchronize()
            ds.render_into(p, out.data_ptr())
            assert ds.info("blocks_per_cu") == full                 # nothing on the GPU: the whole chip
            for _ in range(3):
                ds.render_into(p, out.data_ptr())
            assert ds.info("blocks_per_cu") == expect
            torch.cuda.synchronize()
            assert (bits(out) == want).all()
            ds.set_option("blocks_per_cu", full)
            for _ in range(3):
                ds.render_into(p, out.data_ptr())
            assert ds.info("blocks_per_cu") == full
            ds.set_option("blocks_per_cu", 0)
            torch.cuda.synchronize()
            assert (bits(out) == want).all()
    finally:
        ds.set_option("blocks_per_cu", 0)
        ds.set_option("frames_in_flight", 2)
