#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (ray segments = intersect() calls per second) and ms/frame of the
path-tracing hot path on BASELINE.json's metric config — cbox 640x480 spp=64 on one MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene cbox|bunny|scene1] [--traversal exact|pruned]

One "step" = one full frame: trace kernel(s) + ordered resolve (+ gather to rank 0 when N>1).
N>1 (one rank per GPU): weak scaling — the frame stays 640x480 and the sample count grows to 64*N, rows are
interleaved over ranks (rank r: rows r, r+N, ...), so every GPU traces the same 19.66 M paths as the 1-GPU run; the
row bands are gathered to rank 0 over RCCL inside the timed region.  `python bench.py --gpus N` starts its own N
ranks (torch.distributed.run as a child process, before this process touches the GPU); under an external
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it runs as one of the ranks.

Prints ONE JSON line (rank 0).  `value` is the whole-job rate with the scene already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)

# BASELINE.json configs; bytes_per_segment = SURVEY §8d algorithmic bytes of the reference layout
# (60 B/inner pop + 120 B/triangle leaf + 32 B/sphere leaf + 60 B/closer triangle hit + 40 B shade), from the
# reference-traversal work counters (re-derived by the oracle in tests/test_oracle_pins.py for scene1/cbox).
WORKLOADS = {
    "cbox":   {"w": 640, "h": 480, "spp": 64, "bytes_per_segment": 1455.0, "label": "scenes/cbox/cbox.xml 640x480 spp=64 (BASELINE.json configs[1])"},
    "bunny":  {"w": 640, "h": 480, "spp": 64, "bytes_per_segment": 3625.0, "label": "scenes/bunny/bunny.xml 640x480 spp=64 (BASELINE.json configs[2])"},
    "scene1": {"w": 640, "h": 480, "spp": 16, "bytes_per_segment": 240.0, "label": "scenes/spheres/scene1.xml 640x480 spp=16 (BASELINE.json configs[0])"},
    # configs[3] / configs[4]: buddha.ply / dragon.ply are missing from the reference snapshot -> STAND-IN geometry
    # built from bunny instances (pathtracer_cuda_interactive_amd/standins.py); bytes/segment measured by the oracle on
    # every 16th row at 4 spp of the full-size frame (reference traversal counters, SURVEY §8d formula).
    "buddha_standin": {"w": 1280, "h": 960, "spp": 256, "bytes_per_segment": 2872.0, "builder": True,
                       "label": "STAND-IN for scenes/buddha/buddha.xml 1280x960 spp=256: 8 bunny instances = 1,152,368 tris (BASELINE.json configs[3])"},
    "dragon_standin": {"w": 1920, "h": 1080, "spp": 1024, "bytes_per_segment": 3515.0, "builder": True,
                       "label": "STAND-IN for scenes/dragon 1920x1080 spp=1024: Cornell shell + Phong and plastic bunnies (BASELINE.json configs[4])"},
}


def cpu_baseline(desc, params, name):
    """The oracle (CPU port of the same algorithm) timed on this box's host cores — reported, never shipped.
    Returns (cpu_baseline dict, oracle image of the sampled rows, row stride k of the sample)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_binding as ob
    ob.build()
    t0 = time.perf_counter()
    total_paths = params.width * params.height * params.spp
    budget = 60e6 if name != "cbox" and name != "scene1" else total_paths     # ~0.15-0.4 Mpaths/s/core on mesh scenes: 10-30 s on 16 cores
    if total_paths > budget:       # bound the sample: every k-th row of the frame at full spp
        k = int(np.ceil(total_paths / budget))
        q = params.copy()
        q.row_begin, q.row_end, q.row_stride = 0, params.height, k
        sample = f"every {k}th row of the {params.width}x{params.height} frame at spp={params.spp} ({q.num_rows()} rows)"
    else:
        k = 1
        q = params
        sample = f"full frame {params.width}x{params.height} spp={params.spp}"
    img, cnt = ob.render(desc, q, threads=ob.usable_cores())
    wall = time.perf_counter() - t0
    return ({"value": round(cnt.segments / cnt.seconds / 1e6, 3), "unit": "Msamples/s", "cores": int(cnt.threads_used),
             "kind": "port", "sample": sample, "seconds": round(cnt.seconds, 3), "wall_seconds": round(wall, 3),
             "segments": int(cnt.segments), "bytes_per_segment_measured": round(cnt.bytes_per_segment(), 1)}, img, k)


def parity_rows(frame, oracle_rows, k):
    """The LAST TIMED GPU frame against the oracle's rows of the same frame (rows 0, k, 2k, ... at the config's full spp): every
    float compared bit for bit.  The oracle is the checker here, never the thing measured."""
    gpu = frame[::k].detach().cpu().numpy()
    same = gpu.shape == oracle_rows.shape and bool((gpu.view(np.uint32) == oracle_rows.view(np.uint32)).all())
    out = {"rows": int(oracle_rows.shape[0]), "row_stride": int(k), "bit_identical": same}
    if not same and gpu.shape == oracle_rows.shape:
        out["differing_pixels"] = int((gpu.view(np.uint32) != oracle_rows.view(np.uint32)).any(axis=2).sum())
        out["max_abs_diff"] = float(np.abs(gpu - oracle_rows).max())
    return out


# README.md FPS of the reference on an RTX 3080 at its UI default of 2 samples per frame, 640x480, INCLUDING per-frame CPU
# tonemap + GL upload (README.md:10-11,112-113,121-124): the only numbers the reference publishes.
REFERENCE_FPS = {"scene1": (65, 80), "cbox": (55, 65), "bunny": (45, 50)}


def progressive_mode(args, hs, desc, wl):
    """Frames/s of the interactive accumulation loop (main.cu:272-344 without the window): each frame adds SPF samples
    per pixel into a device accumulation buffer with pt_render_accumulate; one host sync per frame, as a display would need."""
    import torch

    from pathtracer_cuda_interactive_amd.device import DeviceScene
    spf = args.progressive
    W, H = 640, 480
    ds = DeviceScene(desc)
    if not args.frame_timing:
        ds.set_option("timing_frames", 0)        # an interactive loop has no use for per-frame kernel times: no HIP events around the kernels
    for kv in args.set:
        k, v = kv.split("=")
        ds.set_option(k, int(v))
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    p = hs.render_params(W, H, spf)
    p.stream_stride = 1 << 20

    lag = max(0, args.display_lag)
    shown = [torch.empty_like(acc) for _ in range(lag + 1)]       # what a display would read: a copy of the accumulation buffer per frame
    done = [torch.cuda.Event() for _ in range(lag + 1)]

    def frame(k):
        p.sample_offset = k * spf
        ds.accumulate_into(p, acc.data_ptr(), stream)
        if lag == 0:
            torch.cuda.synchronize()                               # one host sync per frame
            return
        # a display that runs `lag` frames behind: frame k's image is copied out on the stream, the host waits for frame k - lag's
        shown[k % (lag + 1)].copy_(acc)
        done[k % (lag + 1)].record()
        if k >= lag:
            done[(k - lag) % (lag + 1)].synchronize()

    for k in range(args.warmup):
        frame(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    segs = 0
    for k in range(args.steps):
        frame(args.warmup + k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    c = ds.counters()
    fps = args.steps / elapsed
    ref = REFERENCE_FPS.get(args.scene)
    out = {"metric": f"progressive frames/s at {spf} samples per frame, {args.scene} 640x480 (render_progressive, main.cu:64-89)",
           "value": round(fps, 1), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": round(fps / (sum(ref) / 2), 1) if ref and spf == 2 else None, "dtype": "f32",
           "data": "synthetic: scene fixture, PCG seed 1984",
           "config": {"workload": f"{wl['label'].split(' (')[0].rsplit(' ', 2)[0]} 640x480, {spf} spp per frame, accumulate + "
                                  + ("host sync per frame" if lag == 0 else f"copy-out per frame, the host waits for the frame {lag} behind"),
                      "display_lag": lag, "frames_in_flight": int(ds.info("frames_in_flight")), "frame_timing_events": bool(args.frame_timing),
                      "kernel_ms_last_frame": round(c.kernel_ms, 4), "segments_last_frame": int(c.segments),
                      "reference_fps_rtx3080_with_ui": list(ref) if ref else None,
                      "note": "reference FPS includes its CPU tonemap + OpenGL upload per frame; this number has no display"}}
    print(json.dumps(out), flush=True)
    ds.close()


def launch_ranks(args):
    """`python bench.py --gpus N` as typed: start N ranks with torch.distributed.run as a CHILD process and relay rank 0's
    JSON line.  Nothing in this (parent) process has imported torch.cuda or called HIP, and it never exec()s."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this image
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    if r.returncode != 0 or line is None:
        raise SystemExit(r.returncode or 1)


class StubRenderer:
    """TEST HOOK (--stub-renderer): exercises bench.py's multi-rank control flow — launcher, sharding, gather, timing
    all-reduce, JSON — on a box without GPUs (gloo, CPU tensors).  It renders nothing: row j of the frame is filled with
    the value j, so the assembled frame proves the de-interleave; every number in the output line is meaningless and the
    line says `"stub": true`.  tests/test_bench_launcher.py is its only user."""

    class _Scene:
        class _C:
            kernel_ms, resolve_ms, segments, paths = 1.0, 0.1, 0, 0

        def __init__(self):
            self.c = self._C()

        def counters(self):
            return self.c

        def info(self, key):
            return 1 if key == "passes" else 0

        def set_option(self, key, value):
            pass

        def frame_times(self, n):
            return np.full(n, self.c.kernel_ms), np.full(n, self.c.resolve_ms)

        def close(self):
            pass

    def __init__(self, desc, hs):
        self.scene = self._Scene()

    def render(self, params, rank, world, force_collective=False):
        import torch

        from pathtracer_cuda_interactive_amd import distributed as D

        def render_rows(q):
            rows = torch.arange(q.row_begin, q.row_end, max(q.row_stride, 1), dtype=torch.float32)
            self.scene.c.paths = int(rows.numel()) * q.width * q.spp
            self.scene.c.segments = 2 * self.scene.c.paths
            return rows[:, None, None].expand(-1, q.width, 3).contiguous()
        return D.render_sharded(render_rows, params, rank, world, force_collective=force_collective)

    def close(self):
        pass


L1_TAG_LOOKUPS_MICROBENCH = 1.4    # L1 tag lookups per cycle per CU that tools/microbench/gather_nodes.hip reaches with this access
                                   # shape (a dependent chain of divergent 16-B loads; profiles/r02_microbench_gather.log)


def physical_roofline(scene, traversal, lds_scene, k_ms, launches):
    """The bound that physically limits the trace kernel, as a fraction <= 1:
      LDS-resident scenes  -> VALU lane throughput:  frac = (SQ_INSTS_VALU / t) / (1024 SIMDs x 2.4 GHz / 2 cycles) x lane utilisation
      scenes in global mem -> the vector L1's tag-lookup rate for divergent loads (one lookup per active lane and 16-B load):
                              frac = TCP_TOTAL_CACHE_ACCESSES / kernel cycles / 256 CUs / 1.4 (the rate the access shape reaches alone);
                              beside it the L2-miss (fabric) bandwidth:  (EA read bytes + write bytes) / t / 8 TB/s
    Instruction and byte counts per launch come from the committed rocprofv3 PMC run of the same command
    (profiles/rNN_<scene>_pmc.json — a launch of this config executes the same work every time); t = this run's
    HIP-event kernel time of one frame, which is `launches` trace_kernel launches when the frame runs in sample passes.
    Returns flat scalars only (they are merged into the top level of `roofline`)."""
    import glob
    cands = sorted(glob.glob(os.path.join(REPO, "profiles", f"r[0-9][0-9]_{scene}_pmc.json")))
    if traversal != "exact" or not cands:
        return {}
    prof = json.load(open(cands[-1]))
    c, dv = prof.get("counters_mean_per_launch", {}), prof.get("derived", {})
    t = k_ms * 1e-3
    out = {"pmc_source": os.path.relpath(cands[-1], REPO)}
    for k_src, k_dst in (("valu_lane_utilization", "lane_utilization"), ("valu_issue_busy", "valu_issue_busy_profiled"),
                         ("sq_wait_any_share_of_wave_cycles", "waves_waiting_share"), ("lds_bank_conflict_share", "lds_bank_conflict_share"),
                         ("l2_hit_rate", "l2_hit_rate")):
        if k_src in dv:
            out[k_dst] = round(float(dv[k_src]), 4)
    if "SQ_INSTS_VALU" in c:
        peak = 1024 * 2.4e9 / 2.0                       # wave64 VALU instructions per second: 2 cycles each on a SIMD-32
        out["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * launches / t / peak, 4)
    if lds_scene:
        if "valu_issue_frac" in out and "lane_utilization" in out:
            out["physical_bound"] = "valu_lane_throughput"
            out["physical_frac"] = round(out["valu_issue_frac"] * out["lane_utilization"], 4)
            out["physical_formula"] = "SQ_INSTS_VALU / kernel_s / (1024 x 2.4e9 / 2) x SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)"
        return out
    if "l2_miss_bytes_per_launch" in dv:
        gbs = dv["l2_miss_bytes_per_launch"] * launches / t / 1e9
        out["l2_miss_GBps"] = round(gbs, 1)
        out["l2_miss_frac_of_hbm_peak"] = round(gbs / HBM_PEAK_GBS, 4)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        rate = c["TCP_TOTAL_CACHE_ACCESSES_sum"] * launches / (t * 2.4e9) / 256.0
        out["l1_tag_lookups_per_cycle_per_cu"] = round(rate, 4)
        out["physical_bound"] = "l1_tag_lookup_rate"
        out["physical_frac"] = round(rate / L1_TAG_LOOKUPS_MICROBENCH, 4)
        out["physical_formula"] = "TCP_TOTAL_CACHE_ACCESSES / (kernel_s x 2.4e9) / 256 CUs / 1.4 lookups per cycle per CU (gather microbenchmark)"
    elif "l2_miss_frac_of_hbm_peak" in out:
        out["physical_bound"] = "l2_miss_bw"
        out["physical_frac"] = out["l2_miss_frac_of_hbm_peak"]
        out["physical_formula"] = "(32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B [TCC_EA0_RDREQ*] + 1024 x WRITE_SIZE) / kernel_s / 8e12"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="cbox", choices=sorted(WORKLOADS))
    ap.add_argument("--traversal", default="exact", choices=["exact", "pruned"])
    ap.add_argument("--tree", default="internal", choices=["internal", "caller"],
                    help="exact traversal: the library's internal tree over the caller's leaf boxes (default; bit-identical "
                         "results) or the caller's (reference median-split) tree itself")
    ap.add_argument("--kernel", type=int, default=0, choices=[0, 1, 2, 3],
                    help="trace kernel: 0 = the library's default, 2 = decoupled traversal / shading per wave, 3 = paths regrouped across "
                         "the waves of a workgroup (LDS-resident scenes), 1 = segment-synchronous")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="pt_scene_set_option before rendering (tuning runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-work-frames", action="store_true",
                    help="skip the two untimed counting frames behind config.work (profiling runs: only the timed kernel on the GPU)")
    ap.add_argument("--force-gather", action="store_true",
                    help="N=1 only: run every frame through the N>1 path — a one-rank nccl (RCCL) group, gather + de-interleave — "
                         "to measure what that step costs per frame")
    ap.add_argument("--stub-renderer", action="store_true",
                    help="TEST HOOK: run the multi-rank control flow on CPU (gloo) with a renderer that renders nothing")
    ap.add_argument("--frame-timing", action="store_true",
                    help="--progressive: keep the library's per-frame HIP timing events (four event records per frame; default off there)")
    ap.add_argument("--display-lag", type=int, default=0, metavar="FRAMES",
                    help="--progressive: 0 = one host sync per frame; n = every frame's image is copied out on the stream and the host waits "
                         "for the copy of the frame n behind (a display loop that runs n frames behind the renderer)")
    ap.add_argument("--progressive", type=int, default=0, metavar="SPF",
                    help="instead of the offline frame: time render_progressive-style frames of SPF samples each "
                         "(pt_render_accumulate; the reference UI's default is 2, main.cu:131) and report frames/s")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)          # before anything in this process touches the GPU

    import torch
    import torch.distributed as dist

    from pathtracer_cuda_interactive_amd import (PT_BVH_SORT_REFERENCE, PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED,
                                                 HostScene, _build)
    from pathtracer_cuda_interactive_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    stub = args.stub_renderer
    if not stub:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the path tracer has no CPU fallback")
        torch.cuda.set_device(local_rank)
    force_gather = bool(args.force_gather) and world == 1
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_DEBUG", "WARN")       # no RCCL version banner on stdout: rank 0 prints ONE JSON line there
        if force_gather and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)

    if not os.path.exists(_build.HIP_LIB) or not os.path.exists(_build.HOST_LIB):
        _build.build_all()

    wl = WORKLOADS[args.scene]
    scene_dir = os.path.join(REPO, "tests", "golden", "scenes")
    if wl.get("builder"):
        from pathtracer_cuda_interactive_amd import standins
        hs = standins.BUILDERS[args.scene](scene_dir)
    else:
        hs = HostScene.load(os.path.join(scene_dir, args.scene + ".pts"))
    desc = hs.finalize(PT_BVH_SORT_REFERENCE)           # the reference's own tree (same libstdc++ tie order)
    spp_total = wl["spp"] * world                        # weak scaling: per-GPU paths constant
    params = hs.render_params(wl["w"], wl["h"], spp_total)
    params.traversal = PT_TRAVERSAL_PRUNED if args.traversal == "pruned" else PT_TRAVERSAL_EXACT

    if args.progressive > 0:
        return progressive_mode(args, hs, desc, wl)

    R = StubRenderer(desc, hs) if stub else D.ShardedRenderer(desc)
    if not stub and args.tree == "caller":
        R.scene.set_option("fast_tree", 0)
    if not stub and args.kernel:
        R.scene.set_option("kernel", args.kernel)
    for kv in ([] if stub else args.set):
        k, v = kv.split("=")
        R.scene.set_option(k, int(v))
    # The timed loop enqueues frame after frame on the stream (N>1: each followed by the gather) with NO host sync inside: the
    # library keeps the HIP events of the last `steps` render calls (option "timing_frames"), read after the closing fence.
    # Every frame renders the same streams, so its work counters are the last frame's.
    R.scene.set_option("timing_frames", max(args.steps, 1))

    def step():
        return R.render(params, rank, world, force_collective=force_gather)

    def fence():
        if not stub:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    c_warm = R.scene.counters() if args.warmup else None
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):
        frame = step()
    fence()
    elapsed = time.perf_counter() - t0
    c_last = R.scene.counters()
    if c_warm is not None and (c_warm.segments, c_warm.paths) != (c_last.segments, c_last.paths):
        raise SystemExit("bench.py: two frames of the same configuration traced different work")
    kernel_ms, resolve_ms = R.scene.frame_times(args.steps)
    if len(kernel_ms) != args.steps:
        raise SystemExit("bench.py: the library did not keep the events of every timed frame")
    if os.environ.get("PT_BENCH_DUMP_FRAMES") and rank == 0:
        sys.stderr.write("kernel_ms per timed frame: " + " ".join(f"{t:.3f}" for t in kernel_ms) + "\n")
    segs = [c_last.segments] * args.steps
    paths = [c_last.paths] * args.steps

    t = torch.tensor([elapsed, float(sum(segs)), float(sum(paths)), float(np.mean(kernel_ms))], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total_segs, total_paths, kern = tmax[0].item(), tsum[1].item(), tsum[2].item(), tmax[3].item()
    else:
        total_segs, total_paths, kern = t[1].item(), t[2].item(), t[3].item()

    # Work the kernel really did per segment (one extra, untimed frame with the counting build of the kernel, rank 0's rows):
    # the algorithmic figures below price the REFERENCE's traversal of the REFERENCE's tree; this is what was traversed.
    work = None
    if not stub and not args.no_work_frames:
        R.scene.set_option("stats", 1)
        R.render(params, rank, world)
        cw = R.scene.counters()
        R.scene.set_option("stats", 0)
        if cw.segments:
            work = {"inner_visits_per_segment": round(cw.node_visits / cw.segments, 3),
                    "leaf_tests_per_segment": round(cw.leaf_tests / cw.segments, 3),
                    "segments_traced_on_the_callers_tree": int(R.scene.info("redo_segments"))}
            if R.scene.info("fast_tree_on"):
                # the same frame on the caller's tree: the visit count the algorithmic bytes are priced on
                R.scene.set_option("fast_tree", 0); R.scene.set_option("stats", 1)
                R.render(params, rank, world)
                cr = R.scene.counters()
                R.scene.set_option("stats", 0); R.scene.set_option("fast_tree", 1)
                work["callers_tree_inner_visits_per_segment"] = round(cr.node_visits / max(cr.segments, 1), 3)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_segs / elapsed / 1e6
        # roofline of the dominant kernel (trace_kernel) on rank 0: algorithmic bytes per launch / HIP-event duration
        seg_launch = float(np.mean(segs))
        k_ms = float(np.mean(kernel_ms))
        achieved = wl["bytes_per_segment"] * seg_launch / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        passes = int(R.scene.info("passes"))          # frames whose per-sample scratch exceeds the budget run in sample passes
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(f"{args.scene}:{args.traversal}")   # PMC bytes per trace_kernel launch
            if traffic is not None:
                traffic = int(traffic * passes)         # "launch" here = one frame = `passes` trace_kernel launches
        lds_scene = bool(R.scene.info("lds_scene"))
        # flat scalars from the committed rocprofv3 PMC run of this config (profiles/, tools/profile_gpu.sh) and this run's kernel time
        # Frames in flight (the library's default: 2): the trace kernel of frame k+1 starts while the last paths of frame k drain,
        # so a launch's HIP-event duration includes the time its blocks waited for the previous launch's to leave the CUs and the
        # durations of consecutive launches overlap.  `achieved` / `frac` keep that duration (per-launch duration);
        # the physical fraction is counted against the time the chip really spent per frame: the wall time of a step.
        in_flight = 1 if stub else int(R.scene.info("frames_in_flight"))
        overlapped = in_flight > 1 and passes == 1
        physical = {} if stub else physical_roofline(args.scene, args.traversal, lds_scene, ms_per_step if overlapped else k_ms, passes)
        if physical and overlapped:
            physical["physical_time_base"] = "ms_per_step (launches of consecutive frames overlap: frames_in_flight)"
        if overlapped:
            # launches per frame interval: ~2 when two frames share every CU.  `achieved` x this = algorithmic bytes per second of
            # the kernel as a whole (all resident launches together)
            physical["launch_overlap"] = round(k_ms / ms_per_step, 3)
            physical["achieved_all_resident_launches"] = round(achieved * k_ms / ms_per_step, 1)
        work_done_frac = None
        if work and "callers_tree_inner_visits_per_segment" in work:
            # "work avoided is not bandwidth achieved" (SURVEY §8d): the algorithmic figure with the inner visits this run really
            # made (60 B each in the reference layout) instead of the caller's tree's
            b_done = wl["bytes_per_segment"] - 60.0 * (work["callers_tree_inner_visits_per_segment"] - work["inner_visits_per_segment"])
            work_done_frac = round(b_done * seg_launch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    # SURVEY §8d's fields above, exactly as defined there: ALGORITHMIC bytes of the reference layout at the
                    # reference's visit counts per second against the HBM peak — not a physical fraction (those bytes are served
                    # by LDS / L1 / L2, and the internal tree avoids part of them).  The physical bound follows, as scalars.
                    "frac_kind": "algorithmic (SURVEY 8d), NOT physical: see physical_frac",
                    "work_done_frac": work_done_frac,
                    "inner_visits_per_segment": work.get("inner_visits_per_segment") if work else None,
                    "callers_tree_inner_visits_per_segment": work.get("callers_tree_inner_visits_per_segment") if work else None,
                    "leaf_tests_per_segment": work.get("leaf_tests_per_segment") if work else None,
                    "kernel": {1: "trace_kernel", 2: "trace_kernel_v2", 3: "trace_kernel_q"}.get(0 if stub else int(R.scene.info("kernel")), "stub"), "kernel_ms": round(k_ms, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                    "kernel_launches_per_step": passes,
                    "algorithmic_bytes_per_segment": wl["bytes_per_segment"], "segments_per_launch": int(seg_launch),
                    "note": ("scene staged in LDS (%d B): physical bound = VALU lane throughput, DESIGN.md 7" % R.scene.info("scene_bytes")) if lds_scene else
                            ("scene (%.1f MB) served by L1/L2/Infinity Cache: physical bound = L1 tag-lookup rate, DESIGN.md 7" % (R.scene.info("scene_bytes") / 1e6))}
        roofline.update(physical)
        out = {
            "metric": "Msamples/sec (rays x spp x bounces = intersect() calls per second), " + wl["label"].split(" (")[0],
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic: stand-in scene built from tests/golden/scenes/{bunny,cbox}.pts (the original PLY is missing from the reference), PCG seed 1984"
                     if wl.get("builder") else
                     "synthetic: scene fixture tests/golden/scenes/%s.pts (parsed from the reference's scene files), PCG seed 1984" % args.scene),
            "config": {"workload": wl["label"] + ("" if world == 1 else f"; weak scaling: spp = {wl['spp']} x {world} GPUs = {spp_total}, rows interleaved over ranks, RCCL gather to rank 0"),
                       "scene": args.scene, "width": wl["w"], "height": wl["h"], "spp": spp_total, "traversal": args.traversal,
                       "paths_per_step": int(total_paths / args.steps), "segments_per_step": int(total_segs / args.steps),
                       "mpaths_per_s": round(total_paths / elapsed / 1e6, 2), "kernel_ms": round(k_ms, 4),
                       "resolve_ms": round(float(np.mean(resolve_ms)), 4), "lds_scene": int(R.scene.info("lds_scene")),
                       "lds_bytes": int(R.scene.info("lds_bytes")),
                       "grid": int(R.scene.info("grid")), "blocks_per_cu": int(R.scene.info("blocks_per_cu")),
                       "blocks_per_cu_occupancy_limit": int(R.scene.info("occupancy")),
                       "vgprs": int(R.scene.info("vgprs_pruned" if args.traversal == "pruned" else "vgprs")),
                       # exact traversal runs on the library's internal tree where scene creation kept one: same leaves tested,
                       # same closest hits (ties in the caller's visit order), fewer inner visits; "caller" = --tree caller
                       "tree": ("stub" if stub else "internal" if R.scene.info("fast_tree_on") else "caller"),
                       "stack_entries": 0 if stub else int(R.scene.info("stack_entries")),
                       "host_sync_per_step": False, "forced_gather": force_gather, "frames_in_flight": in_flight,
                       "kernel_ms_note": ("launches of consecutive frames overlap on the chip (frames_in_flight): kernel_ms is the HIP-event "
                                          "duration of one launch, about roofline.launch_overlap frame intervals; ms_per_step is the frame interval")
                                         if overlapped else None,
                       "work": work,
                       "frame_mean": round(float(frame.mean().item()), 6)},
            "roofline": roofline,
        }
        if stub:
            out["stub"] = True
            out["config"]["frame_rows_ok"] = bool((frame[:, 0, 0] == torch.arange(wl["h"], dtype=torch.float32)).all())
        if world == 1 and not args.no_cpu_baseline and not stub:
            # the oracle's rows of the very frame that was timed: the CPU baseline AND the checker of the last timed GPU frame
            out["cpu_baseline"], oracle_rows, k = cpu_baseline(desc, hs.render_params(wl["w"], wl["h"], wl["spp"]), args.scene)
            out["parity_rows"] = parity_rows(frame, oracle_rows, k) if args.traversal == "exact" else None
        print(json.dumps(out), flush=True)
        if out.get("parity_rows") and not out["parity_rows"]["bit_identical"]:
            sys.stderr.write("bench.py: the timed GPU frame differs from the oracle: %r\n" % (out["parity_rows"],))
            R.close()
            raise SystemExit(3)
    R.close()
    if world > 1 or force_gather:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
