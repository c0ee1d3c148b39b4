"""Does the drain tail of one frame overlap with the start of the next?  Renders F frames of a scene (a) on one handle and one
stream, back to back, (b) on K handles of the same scene, each on its own HIP stream, frame i on handle i % K — and prints the
wall time per frame of both (host clock around a device synchronise; no host sync inside).  Every frame of (b) is compared
with the frame of (a) bit for bit.
Usage: python tools/gpu_overlap_probe.py [scene[@spp]] [frames] [handles]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
CONFIGS = {"cbox": (640, 480, 64), "bunny": (640, 480, 64), "scene1": (640, 480, 16), "teapot": (640, 480, 16)}


def main():
    import torch

    from pathtracer_cuda_interactive_amd import PT_BVH_SORT_REFERENCE, HostScene
    from pathtracer_cuda_interactive_amd import device as dev
    spec = sys.argv[1] if len(sys.argv) > 1 else "cbox"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    handles = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    name, _, spp = spec.partition("@")
    w, h, spp0 = CONFIGS[name]
    hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", name + ".pts"))
    d = hs.finalize(PT_BVH_SORT_REFERENCE)
    params = hs.render_params(w, h, int(spp) if spp else spp0)
    if handles <= 0:                 # only part two, with no stream in the process that it does not need (-1: on the null stream)
        one = dev.DeviceScene(d)
        extra = [torch.cuda.Stream() for _ in range(int(os.environ.get("PT_PROBE_EXTRA_STREAMS", "0")))]    # streams of the application's
        for e in extra:
            with torch.cuda.stream(e):
                torch.zeros(16, device="cuda").add_(1)
        caller = torch.cuda.Stream().cuda_stream if handles == 0 else None
        outs = [torch.empty(h, w, 3, dtype=torch.float32, device="cuda") for _ in range(2)]
        want = one.render(params).view(np.uint32).reshape(h, w, 3)
        burst = int(os.environ.get("PT_PROBE_BURST", "0"))
        if burst:
            print(f"bursts of {burst} frames, a device synchronise after each", flush=True)
        for depth in (1, 2, 3, 4, 2, 1):
            one.set_option("frames_in_flight", depth)
            ts = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(frames):
                    one.render_into(params, outs[i % 2].data_ptr(), stream=caller)
                    if burst and (i + 1) % burst == 0:
                        torch.cuda.synchronize()                 # a caller that submits `burst` frames and waits for them
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / frames * 1e3)
            same = all(bool((o.cpu().numpy().view(np.uint32) == want).all()) for o in outs)
            print(f"{spec}: one handle, caller on {'a stream of its own' if caller else 'the null stream'}, frames_in_flight {depth}: {min(ts[1:]):.4f} ms/frame "
                  f"(runs {' '.join(f'{t:.4f}' for t in ts)})   frames identical: {same}", flush=True)
        one.close()
        return
    scenes = [dev.DeviceScene(d) for _ in range(handles)]
    for s in scenes:
        s.set_option("frames_in_flight", 1)                      # part one: the overlap comes from the caller's streams alone
    streams = [torch.cuda.Stream() for _ in range(handles)]
    outs = [torch.empty(h, w, 3, dtype=torch.float32, device="cuda") for _ in range(handles)]

    def run(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            j = i % k
            scenes[j].render_into(params, outs[j].data_ptr(), stream=streams[j].cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3

    for k in (1, handles):
        run(k)                                                   # warm-up
    want = outs[0].cpu().numpy().view(np.uint32)
    for rep in range(3):
        a = run(1)
        b = run(handles)
        same = all(bool((o.cpu().numpy().view(np.uint32) == want).all()) for o in outs)
        print(f"{spec}: one stream {a:.4f} ms/frame   {handles} streams {b:.4f} ms/frame   ({(b / a - 1) * 100:+.1f} %)   frames identical: {same}", flush=True)
    # the same inside ONE handle: option frames_in_flight (the handle's trace kernels on its own streams, resolves on the caller's)
    one = scenes[0]
    for depth in (1, 2, 3, 4, 2, 1):
        one.set_option("frames_in_flight", depth)
        ts = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(frames):
                one.render_into(params, outs[i % 2].data_ptr(), stream=streams[0].cuda_stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / frames * 1e3)
        same = all(bool((o.cpu().numpy().view(np.uint32) == want).all()) for o in outs[:2])
        print(f"{spec}: one handle, one caller stream, frames_in_flight {depth}: {min(ts):.4f} ms/frame (runs {' '.join(f'{t:.4f}' for t in ts)})   frames identical: {same}", flush=True)
    for s in scenes:
        s.close()


if __name__ == "__main__":
    main()
