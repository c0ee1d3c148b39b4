"""Host time of one pt_render_accumulate call (enqueue only) against the GPU time of the frame it enqueues: is an interactive loop
of small frames bound by the host's launch path?  Usage: python tools/gpu_host_call_cost.py [scene] [spp]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from pathtracer_cuda_interactive_amd import PT_BVH_SORT_REFERENCE, HostScene
    from pathtracer_cuda_interactive_amd import device as dev
    name = sys.argv[1] if len(sys.argv) > 1 else "scene1"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", name + ".pts"))
    ds = dev.DeviceScene(hs.finalize(PT_BVH_SORT_REFERENCE))
    acc = torch.zeros(480, 640, 3, dtype=torch.float32, device="cuda")
    p = hs.render_params(640, 480, spp)
    p.stream_stride = 1 << 20
    stream = torch.cuda.Stream()
    n = 600
    for depth in (1, 2, 3):
        ds.set_option("frames_in_flight", depth)
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                p.sample_offset = k * spp
                ds.accumulate_into(p, acc.data_ptr(), stream.cuda_stream)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"{name} {spp} spp, frames_in_flight {depth}: host {1e6 * (t1 - t0) / n:.1f} us per call, all frames done after {1e6 * (t2 - t0) / n:.1f} us per frame", flush=True)
    ds.close()


if __name__ == "__main__":
    main()
