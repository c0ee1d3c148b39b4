"""Generates the golden vectors under tests/golden/ from the CPU oracle (deterministic math,
per-(pixel,sample) PCG streams) on the .pts scene fixtures:

  images/<scene>_<W>x<H>_spp<S>.npy    float32 [H,W,3]           — oracle image
  rays/<scene>.npz                     rays [n,8], tuv [n,3], prim [n] — per-ray closest-hit KATs
  bvh/<scene>_nodes.npy                structured node pool (reference std::sort tie order, bvh.cu:34-37)
  pins.json                            PCG KATs + scene topology + image statistics recorded by SURVEY §8c

The oracle itself is pinned against SURVEY §8c by tests/test_oracle_pins.py; these files
pin the oracle against drift and give the GPU tests vectors that need no CPU work."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_binding as ob  # noqa: E402
from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE  # noqa: E402

IMAGES = {"scene1": (64, 48, 4), "scene1_phong": (64, 48, 4), "cbox": (64, 48, 4), "teapot": (48, 48, 2),
          "bunny": (64, 48, 1), "tetrahedron": (48, 48, 4),
          # the rest of the scenes the reference ships (dragon / buddha need meshes that are missing from the snapshot)
          "scene0": (48, 48, 4), "scene0_spherical_light": (64, 48, 4), "scene1_spherical_light": (64, 48, 4),
          "scene2": (64, 48, 4), "scene3": (64, 48, 4), "scene4": (64, 48, 4), "aabb_test": (64, 48, 2),
          "single_triangle": (64, 48, 2)}
N_RAYS = 1024


def make_rays(hs, d, n, seed):
    """Half camera rays, half secondary rays leaving surface points in random directions."""
    rng = np.random.default_rng(seed)
    cam = hs.camera
    p = hs.render_params(cam.width, cam.height, 1)
    o = np.array(p.cam_origin, dtype=np.float32)
    tl = np.array(p.cam_top_left, dtype=np.float32)
    hz = np.array(p.cam_horizontal, dtype=np.float32)
    vt = np.array(p.cam_vertical, dtype=np.float32)
    uv = rng.random((n // 2, 2), dtype=np.float32)
    dirs = tl[None] + uv[:, :1] * hz[None] - uv[:, 1:] * vt[None] - o[None]
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    prim_rays = np.concatenate([np.broadcast_to(o, dirs.shape), dirs, np.zeros((n // 2, 1), np.float32),
                                np.full((n // 2, 1), np.inf, np.float32)], axis=1).astype(np.float32)
    tuv, prim = ob.intersect(d, prim_rays)
    hit = prim >= 0
    pts = (prim_rays[:, :3] + prim_rays[:, 3:6] * tuv[:, :1]).astype(np.float32)
    pts = np.where(hit[:, None], pts, o[None] + rng.standard_normal((n // 2, 3)).astype(np.float32))
    d2 = rng.standard_normal((n - n // 2, 3)).astype(np.float32)
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    sec = np.concatenate([pts[: n - n // 2], d2, np.full((n - n // 2, 1), 1e-4, np.float32),
                          np.full((n - n // 2, 1), np.finfo(np.float32).max, np.float32)], axis=1).astype(np.float32)
    return np.concatenate([prim_rays, sec], axis=0)


def sweep_tree_pins():
    import hashlib

    from pathtracer_cuda_interactive_amd import device as dev
    out = {}
    for name in ("scene4", "cbox", "teapot", "bunny"):
        hs = HostScene.load(os.path.join(HERE, "scenes", name + ".pts"))
        d = hs.finalize(PT_BVH_SORT_REFERENCE)
        _, info = dev.build_bvh_sweep(d)
        out[name] = {"num_shapes": int(d.num_shapes), "depth": int(info["depth"]), "md5": hashlib.md5(info["nodes"].tobytes()).hexdigest()}
    return out


if __name__ == "__main__":
    for sub in ("images", "rays", "bvh"):
        os.makedirs(os.path.join(HERE, sub), exist_ok=True)
    topo = {}
    for k, name in enumerate(IMAGES):
        hs = HostScene.load(os.path.join(HERE, "scenes", name + ".pts"))
        d = hs.finalize(PT_BVH_SORT_REFERENCE)
        w, h, spp = IMAGES[name]
        img, cnt = ob.render(d, hs.render_params(w, h, spp))
        np.save(os.path.join(HERE, "images", f"{name}_{w}x{h}_spp{spp}.npy"), img)
        rays = make_rays(hs, d, N_RAYS, 100 + k)
        tuv, prim = ob.intersect(d, rays)
        np.savez_compressed(os.path.join(HERE, "rays", name + ".npz"), rays=rays, tuv=tuv, prim=prim)
        if d.num_nodes <= 100:
            np.save(os.path.join(HERE, "bvh", name + "_nodes.npy"), hs.nodes_array())
        topo[name] = {"shapes": d.num_shapes, "meshes": d.num_meshes, "materials": d.num_materials,
                      "lights": d.num_lights, "nodes": d.num_nodes, "root": d.root, "depth": hs.bvh_depth}
        print(name, topo[name], "segments", cnt.segments, "hits", int((prim >= 0).sum()), "/", len(prim))
    pins = {
        "_source": "SURVEY.md §8c (values recorded from a host build of the reference's own headers) + the published pcg32 demo vector + topology from this build",
        "pcg": [
            {"stream": 1, "seed": 0x853c49e6748fea9b, "state": 0xf6e7b88658a69fc9, "inc": 0x3,
             "u32": [0x73c29fdb, 0xfbaa1ff7, 0xdb022af6, 0x12d7398c]},
            {"stream": 0, "seed": 1984, "state": 0xd376533ed32f1bee, "inc": 0x1,
             "u32": [0xb33f1a1b, 0xba7efb20, 0x35e75b63, 0xe8b62e0e],
             "f32": [0.700181603, 0.728500009, 0.210561395, 0.909029841]},
            {"stream": 307199, "seed": 1984, "state": 0x73fb18ddc2535d92, "inc": 0x95fff,
             "u32": [0x1181fd82, 0xfc486444, 0x250ed378, 0x594aa51b]},
            {"stream": 54, "seed": 42, "u32": [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e],
             "published": "pcg32-demo of the PCG reference implementation (pcg-c-basic): pcg32_srandom_r(&rng, 42u, 54u) -> first "
                          "six 32-bit outputs; independent of this repo and of the reference"},
        ],
        "survey_images_libm_per_pixel_rng": {
            "scene1": {"w": 640, "h": 480, "spp": 16, "mean": 0.325954931, "max": 1.67586458,
                       "px_320_240_hex": ["3ea335c2", "3de00123", "3d8c0048"], "sha256_16": "e3e2dab3ce1dbe76"},
            "cbox": {"w": 640, "h": 480, "spp": 64, "mean": 0.424828197, "max": 15.7954884,
                     "px_320_240_hex": ["3f01430c", "3e6917af", "3e1d8b0e"], "sha256_16": "e367250614e723d2"},
        },
        "survey_counters": {
            "scene1": {"segs_per_path": 2.097, "inner_per_seg": 2.60, "leaf_per_seg": 1.38, "max_stack": 3, "rng_per_path": 3.59, "bytes_per_seg": 240},
            "cbox": {"segs_per_path": 3.553, "inner_per_seg": 14.76, "leaf_per_seg": 3.93, "closer_per_seg": 0.963, "max_stack": 7, "rng_per_path": 7.79, "bytes_per_seg": 1455},
        },
        "survey_topology": {
            "scene1": {"shapes": 4, "nodes": 7, "root": 6, "depth": 3},
            "cbox": {"shapes": 38, "meshes": 8, "materials": 5, "lights": 2, "nodes": 75, "root": 74, "depth": 7},
            "bunny": {"shapes": 288094, "meshes": 3, "nodes": 576187, "depth": 20},
            "teapot": {"shapes": 15706, "nodes": 31411, "depth": 15},
            "scene4": {"shapes": 30, "nodes": 59, "depth": 6},
        },
        "topology": topo,
        # the library's internal tree (pt_bvh_build_sweep, host code): node pool bytes, so that a change of the builder shows
        "sweep_tree": sweep_tree_pins(),
    }
    with open(os.path.join(HERE, "pins.json"), "w") as f:
        json.dump(pins, f, indent=1)
