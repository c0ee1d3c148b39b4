#!/usr/bin/env python3
"""Offline render front-end (what the reference's main.cu does up to line 266, minus the window):
   python tools/render.py <scene.xml|scene.pts|buddha_standin|dragon_standin> [-o out.pfm|out.ppm] [--width W --height H --spp S]
                          [--traversal exact|pruned] [--seed 1984] [--bvh reference|lbvh|sah] [--nee]
Needs a GPU (no CPU fallback).  Multi-GPU: launch with torchrun; rows are interleaved over ranks, rank 0 writes."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("-o", "--output", default="render.pfm")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--spp", type=int)
    ap.add_argument("--seed", type=int, default=1984)
    ap.add_argument("--traversal", default="exact", choices=["exact", "pruned"])
    ap.add_argument("--bvh", default="reference", choices=["reference", "lbvh", "sah"],
                    help="reference: the host's reproduction of the reference tree (default); lbvh / sah: built on the GPU")
    ap.add_argument("--nee", action="store_true", help="next-event estimation (an extension: the reference samples no light)")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    from pathtracer_cuda_interactive_amd import (PT_BVH_SORT_REFERENCE, PT_RENDER_NEE, PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED,
                                                 HostScene, standins, write_image)
    from pathtracer_cuda_interactive_amd import device as dev
    from pathtracer_cuda_interactive_amd import distributed as D
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    t0 = time.perf_counter()
    hs = standins.BUILDERS[a.scene](os.path.join(REPO, "tests", "golden", "scenes")) if a.scene in standins.BUILDERS else HostScene.load(a.scene)
    desc = hs.finalize(PT_BVH_SORT_REFERENCE)
    depth = hs.bvh_depth
    if a.bvh != "reference":
        desc, info = dev.build_bvh_device(desc, dev.PT_BVH_DEVICE_SAH if a.bvh == "sah" else dev.PT_BVH_DEVICE_LBVH)
        depth = info["depth"]
    t1 = time.perf_counter()
    p = hs.render_params(a.width, a.height, a.spp, seed=a.seed)
    p.traversal = PT_TRAVERSAL_PRUNED if a.traversal == "pruned" else PT_TRAVERSAL_EXACT
    p.flags = PT_RENDER_NEE if a.nee else 0
    R = D.ShardedRenderer(desc)
    frame = R.render(p, rank, world)
    c = R.scene.counters()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if rank == 0:
        write_image(a.output, frame.cpu().numpy())
        print(f"{a.scene}: {desc.num_shapes} primitives, {a.bvh} BVH of depth {depth}; parse+build {t1 - t0:.2f} s; "
              f"{p.width}x{p.height} spp={p.spp}: kernel {c.kernel_ms:.2f} ms on rank 0 ({c.segments / c.kernel_ms / 1e3:.0f} Msamples/s), "
              f"upload+render+gather {t2 - t1:.3f} s -> {a.output}")
    R.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
