"""How many more boxes does exact traversal visit when the internal tree's inner boxes are conservative 16-bit (or 8-bit)
quantisations instead of the exact unions?  (DESIGN.md §9, next candidate for scenes in global memory: 32-B nodes, two L1 tag
lookups per visit instead of four.)  CPU only: the host sweep builder's tree over a scene's leaf boxes, camera rays + rays leaving
random surface points, traversal without pruning (a node is visited iff the ray hits its box, as the reference does), boxes
quantised top-down in each node's own DEQUANTISED frame, rounded outwards and checked with the dequantisation itself.
Usage: python tests/tools/sim_quantised_nodes.py [scene] [rays]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from pathtracer_cuda_interactive_amd import PT_BVH_SORT_REFERENCE, HostScene  # noqa: E402
from pathtracer_cuda_interactive_amd import device as dev  # noqa: E402


def slab_hit(lo, hi, org, inv):
    """bbox.cuh:35-61 in float32: didHit = tfar >= max(0, tnear)."""
    with np.errstate(invalid="ignore"):
        t0 = (lo - org) * inv
        t1 = (hi - org) * inv
    tn = np.minimum(t0, t1).max(axis=1)
    tf = np.maximum(t0, t1).min(axis=1)
    return tf >= np.maximum(tn, np.float32(0))


def count_visits(nodes, root, lo, hi, org, inv):
    """Inner visits and leaf-box hits per ray: pairs (ray, node) whose box was hit, level by level."""
    left, right, prim = nodes["left"], nodes["right"], nodes["prim"]
    rays = np.arange(len(org))
    node = np.full(len(org), root, dtype=np.int64)
    inner = leaf = 0
    while len(rays):
        inner += len(rays)                                      # every pair here is an inner node whose box the ray hit (or the root)
        kids = np.concatenate([left[node], right[node]])
        rr = np.concatenate([rays, rays])
        hit = slab_hit(lo[kids], hi[kids], org[rr], inv[rr])
        kids, rr = kids[hit], rr[hit]
        is_leaf = prim[kids] != -1
        leaf += int(is_leaf.sum())
        node, rays = kids[~is_leaf], rr[~is_leaf]
    return inner, leaf


def quantise(nodes, root, bits):
    """Conservative boxes: child boxes as `bits`-bit offsets in the parent's own dequantised box, top-down."""
    n = len(nodes)
    qlo = nodes["bmin"].astype(np.float32).copy()
    qhi = nodes["bmax"].astype(np.float32).copy()
    levels = (1 << bits) - 1
    frontier = np.array([root])
    left, right, prim = nodes["left"], nodes["right"], nodes["prim"]
    while len(frontier):
        par = frontier[prim[frontier] == -1]
        if not len(par):
            break
        nxt = []
        for kid in (left[par], right[par]):
            base, ext = qlo[par], (qhi[par] - qlo[par])
            step = (ext / np.float32(levels)).astype(np.float32)
            safe = np.where(step > 0, step, np.float32(1))
            a = np.floor((nodes["bmin"][kid] - base) / safe).clip(0, levels)
            b = np.ceil((nodes["bmax"][kid] - base) / safe).clip(0, levels)
            for _ in range(3):                                   # the kernel's own dequantisation must not cut into the exact box
                lo_d = (base + a.astype(np.float32) * step).astype(np.float32)
                hi_d = (base + b.astype(np.float32) * step).astype(np.float32)
                a = np.where(lo_d > nodes["bmin"][kid], np.maximum(a - 1, 0), a)
                b = np.where(hi_d < nodes["bmax"][kid], np.minimum(b + 1, levels), b)
            lo_d = np.minimum((base + a.astype(np.float32) * step).astype(np.float32), nodes["bmin"][kid])
            hi_d = np.maximum((base + b.astype(np.float32) * step).astype(np.float32), nodes["bmax"][kid])
            inner_kid = prim[kid] == -1
            # leaves keep their exact boxes (tested on their own, next to the primitive record); inner boxes are the conservative ones
            qlo[kid[inner_kid]], qhi[kid[inner_kid]] = lo_d[inner_kid], hi_d[inner_kid]
            nxt.append(kid[inner_kid])
        frontier = np.concatenate(nxt)
    assert (qlo <= nodes["bmin"]).all() and (qhi >= nodes["bmax"]).all()
    return qlo, qhi


def quantise_global(nodes, root, bits):
    """Every box (leaves' too: the parent holds both children's boxes) on ONE grid over the scene's box, rounded outwards."""
    levels = (1 << bits) - 1
    base = nodes["bmin"][root].astype(np.float32)
    step = ((nodes["bmax"][root] - base) / np.float32(levels)).astype(np.float32)
    safe = np.where(step > 0, step, np.float32(1))
    a = np.floor((nodes["bmin"] - base) / safe).clip(0, levels)
    b = np.ceil((nodes["bmax"] - base) / safe).clip(0, levels)
    for _ in range(3):
        lo_d = (base + a.astype(np.float32) * step).astype(np.float32)
        hi_d = (base + b.astype(np.float32) * step).astype(np.float32)
        a = np.where(lo_d > nodes["bmin"], np.maximum(a - 1, 0), a)
        b = np.where(hi_d < nodes["bmax"], np.minimum(b + 1, levels), b)
    lo_d = np.minimum((base + a.astype(np.float32) * step).astype(np.float32), nodes["bmin"])
    hi_d = np.maximum((base + b.astype(np.float32) * step).astype(np.float32), nodes["bmax"])
    return lo_d, hi_d


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    hs = HostScene.load(os.path.join(REPO, "tests", "golden", "scenes", name + ".pts"))
    d = hs.finalize(PT_BVH_SORT_REFERENCE)
    _, info = dev.build_bvh_sweep(d)
    nodes, root = info["nodes"], info["root"]
    rng = np.random.default_rng(7)
    p = hs.render_params(640, 480, 1)
    o = np.array(p.cam_origin, dtype=np.float32)
    tl, hz, vt = (np.array(v, dtype=np.float32) for v in (p.cam_top_left, p.cam_horizontal, p.cam_vertical))
    u, v = rng.random(n_rays // 2, dtype=np.float32), rng.random(n_rays // 2, dtype=np.float32)
    dirs = tl[None] + u[:, None] * hz[None] - v[:, None] * vt[None] - o[None]
    org = np.repeat(o[None], n_rays // 2, axis=0)
    # rays leaving random leaf boxes' centres in uniform directions (a stand-in for bounce rays)
    leaves = np.flatnonzero(nodes["prim"] != -1)
    pick = rng.choice(leaves, n_rays // 2)
    org2 = ((nodes["bmin"][pick] + nodes["bmax"][pick]) * np.float32(0.5)).astype(np.float32)
    d2 = rng.normal(size=(n_rays // 2, 3)).astype(np.float32)
    org = np.concatenate([org, org2]).astype(np.float32)
    dirs = np.concatenate([dirs, d2]).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = (np.float32(1) / dirs).astype(np.float32)
    lo, hi = nodes["bmin"].astype(np.float32), nodes["bmax"].astype(np.float32)
    i0, l0 = count_visits(nodes, root, lo, hi, org, inv)
    print(f"{name}: {len(nodes) // 2 + 1} primitives, {len(org)} rays; exact boxes: {i0 / len(org):.2f} inner visits, {l0 / len(org):.2f} leaf boxes hit per ray")
    for bits in (16, 12, 8):
        qlo, qhi = quantise(nodes, root, bits)
        i1, l1 = count_visits(nodes, root, qlo, qhi, org, inv)
        # lookups of 16 B: exact = 4 per inner visit + 3 per leaf (48-B primitive record); quantised = 2 per inner visit (32-B node),
        # and per leaf reached 2 for its exact box (24 B) + 3 for the primitive when that box is hit
        exact = 4 * i0 + 3 * l0
        quant = 2 * i1 + 2 * l1 + 3 * l0
        print(f"  {bits:2d}-bit conservative inner boxes, per-node frames: {i1 / len(org):.2f} inner visits (x{i1 / i0:.3f}), {l1 / len(org):.2f} leaves reached (x{l1 / max(l0, 1):.3f}); "
              f"16-B lookups per ray {exact / len(org):.1f} -> {quant / len(org):.1f} ({(quant / exact - 1) * 100:+.1f} %)")
    for bits in (16, 12):
        glo, ghi = quantise_global(nodes, root, bits)
        i2, l2 = count_visits(nodes, root, glo, ghi, org, inv)      # l2: leaves whose QUANTISED box is hit: each costs its exact box (2 lookups)
        quant = 2 * i2 + 2 * l2 + 3 * l0
        print(f"  {bits:2d}-bit boxes on one grid over the scene (32-B nodes, no frames): {i2 / len(org):.2f} inner visits (x{i2 / i0:.3f}), {l2 / len(org):.2f} leaves reached "
              f"(x{l2 / max(l0, 1):.3f}); 16-B lookups per ray {exact / len(org):.1f} -> {quant / len(org):.1f} ({(quant / exact - 1) * 100:+.1f} %)")


if __name__ == "__main__":
    main()
