"""CPU simulation of wave schedules for the trace kernel, fed with real per-path event traces from the oracle
(I inner visit, L leaf test, S shaded hit, M miss).  Estimates VALU wave-instructions per segment for
  v2  : one ray per lane, vote bursts, scheduler phase when >= THRESH lanes wait        (the shipped kernel)
  v3  : two rays per lane (the second parked in LDS); a lane whose ray waits for shading traverses its other ray
Costs are wave-instruction estimates: inner step 70, leaf step 60, scheduler phase 440 (shade 250 + refill 150 + begin 40),
ray swap 45, parked-job load/store 30.  Occupancy / latency effects are NOT modelled."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob  # noqa: E402
from conftest import load_scene  # noqa: E402

C_IN, C_LF, C_SCHED, C_SWAP, C_JOB = 70, 60, 440, 45, 30


def load_paths(name, n_pixels, spp, seed=0):
    hs, d = load_scene(name)
    p = hs.render_params(640, 480, spp)
    rng = np.random.default_rng(seed)
    # 64-pixel row strips, like the real work distribution
    xs = rng.integers(0, 640 - 64, n_pixels // 64)
    ys = rng.integers(0, 480, n_pixels // 64)
    xy = np.concatenate([np.stack([x + np.arange(64), np.full(64, y)], 1) for x, y in zip(xs, ys)])
    paths = ob.trace_pixels(d, p, xy).split("E")[:-1]
    out = []
    for t in paths:                       # -> list of segments, each a string of I/L ending with S or M
        segs, cur = [], ""
        for ch in t:
            cur += ch
            if ch in "SM":
                segs.append(cur)
                cur = ""
        out.append(segs)
    return out


class Ray:
    __slots__ = ("segs", "si", "pos")

    def __init__(self, segs):
        self.segs, self.si, self.pos = segs, 0, 0

    def kind(self):                       # 'I', 'L', or None when the traversal of the current segment is finished
        ch = self.segs[self.si][self.pos]
        return ch if ch in "IL" else None

    def step(self):
        self.pos += 1

    def shade(self):                      # returns True if the path continues
        self.si += 1
        self.pos = 0
        return self.si < len(self.segs)


def simulate(paths, mode, thresh=40, K=6):
    it = iter(paths)
    n = 64
    A = [None] * n                        # active rays
    B = [None] * n                        # parked rays (v3): (ray, state) state 'ready'|'wait'
    cost = steps_in = steps_lf = act_in = act_lf = sched = sched_lanes = 0
    work_left = True
    total_segs = sum(len(p) for p in paths)

    def new_ray():
        nonlocal work_left
        try:
            return Ray(next(it))
        except StopIteration:
            work_left = False
            return None

    while True:
        trav = [a is not None and a.kind() is not None for a in A]
        if mode == "v3":
            # rotate: finished/empty A with a ready B
            rotated = False
            for l in range(n):
                if not trav[l] and B[l] is not None and B[l][1] == "ready":
                    old = A[l]
                    A[l] = B[l][0]
                    B[l] = (old, "wait") if old is not None else None
                    trav[l] = True
                    rotated = True
            if rotated:
                cost += C_SWAP
        if mode == "v2":
            pend = sum(1 for l in range(n) if not trav[l] and (A[l] is not None or work_left))
            if pend >= thresh or not any(trav):
                if pend == 0:
                    break
                sched += 1
                sched_lanes += pend
                cost += C_SCHED
                for l in range(n):
                    if trav[l]:
                        continue
                    if A[l] is not None and not A[l].shade():
                        A[l] = None
                    if A[l] is None:
                        A[l] = new_ray()
        else:
            jobs = sum(1 for l in range(n) if (B[l] is not None and B[l][1] == "wait") or (not trav[l] and A[l] is not None)
                       or ((B[l] is None or (A[l] is None)) and work_left))
            if jobs >= thresh or not any(trav):
                if jobs == 0:
                    break
                sched += 1
                sched_lanes += jobs
                cost += C_SCHED + C_JOB
                for l in range(n):
                    if not trav[l] and A[l] is not None and B[l] is None:      # park the finished A
                        B[l] = (A[l], "wait")
                        A[l] = None
                    if B[l] is not None and B[l][1] == "wait":
                        r = B[l][0]
                        B[l] = (r, "ready") if r.shade() else None
                    if B[l] is None:
                        r = new_ray()
                        if r is not None:
                            B[l] = (r, "ready")
                    if not trav[l] and A[l] is None and B[l] is not None and B[l][1] == "ready":
                        A[l] = B[l][0]
                        B[l] = None
                        r = new_ray()
                        if r is not None:
                            B[l] = (r, "ready")
        for _ in range(K):
            kinds = [a.kind() if a is not None else None for a in A]
            n_in, n_lf = kinds.count("I"), kinds.count("L")
            if n_in == 0 and n_lf == 0:
                break
            k = "I" if n_in >= n_lf else "L"
            if k == "I":
                cost += C_IN; steps_in += 1; act_in += n_in
            else:
                cost += C_LF; steps_lf += 1; act_lf += n_lf
            for l in range(n):
                if kinds[l] == k:
                    A[l].step()
    return dict(cost_per_seg=cost / total_segs, inner_steps=steps_in, inner_act=act_in / max(steps_in, 1), leaf_steps=steps_lf,
                leaf_act=act_lf / max(steps_lf, 1), sched=sched, sched_lanes=sched_lanes / max(sched, 1), segs=total_segs)


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cbox"
    paths = load_paths(name, 64 * 24, 16)
    print(name, len(paths), "paths")
    for mode, th, K in (("v2", 40, 6), ("v2", 32, 6), ("v2", 24, 6), ("v3", 40, 6), ("v3", 32, 6), ("v3", 48, 6), ("v3", 56, 6), ("v3", 64, 6)):
        r = simulate(paths, mode, th, K)
        print(f"{mode} T{th} K{K}: {r['cost_per_seg']:.0f} wave-instr/seg; inner {r['inner_steps']} steps x {r['inner_act']:.1f} lanes; "
              f"leaf {r['leaf_steps']} x {r['leaf_act']:.1f}; sched {r['sched']} x {r['sched_lanes']:.1f} lanes")


def simulate_policy(paths, thresh, K, pick):
    """v2 structure with a pluggable step-kind policy pick(n_in, n_lf) -> 'I' | 'L'."""
    it = iter(paths)
    n = 64
    A = [None] * n
    cost = steps_in = steps_lf = act_in = act_lf = sched = sched_lanes = 0
    work_left = True
    total_segs = sum(len(p) for p in paths)
    while True:
        trav = [a is not None and a.kind() is not None for a in A]
        pend = sum(1 for l in range(n) if not trav[l] and (A[l] is not None or work_left))
        if pend >= thresh or not any(trav):
            if pend == 0:
                break
            sched += 1; sched_lanes += pend; cost += C_SCHED
            for l in range(n):
                if trav[l]:
                    continue
                if A[l] is not None and not A[l].shade():
                    A[l] = None
                if A[l] is None:
                    try:
                        A[l] = Ray(next(it))
                    except StopIteration:
                        work_left = False
        for _ in range(K):
            kinds = [a.kind() if a is not None else None for a in A]
            n_in, n_lf = kinds.count("I"), kinds.count("L")
            if n_in == 0 and n_lf == 0:
                break
            k = pick(n_in, n_lf)
            if k == "I":
                cost += C_IN; steps_in += 1; act_in += n_in
            else:
                cost += C_LF; steps_lf += 1; act_lf += n_lf
            for l in range(n):
                if kinds[l] == k:
                    A[l].step()
    return cost / total_segs, steps_in, act_in / max(steps_in, 1), steps_lf, act_lf / max(steps_lf, 1), sched, sched_lanes / max(sched, 1)


if __name__ == "__main__" and "--policies" in sys.argv:
    paths = load_paths(sys.argv[1], 64 * 24, 16)
    pol = {"vote": lambda a, b: "I" if a >= b else "L"}
    for th in (8, 12, 16, 20, 24, 28):
        pol[f"leaf>= {th}"] = (lambda th: lambda a, b: "L" if (b >= th or a == 0) else "I")(th)
    for w in (1.25, 1.5, 2.0, 3.0):
        pol[f"vote w{w}"] = (lambda w: lambda a, b: "I" if (a >= w * b and a > 0) or b == 0 else "L")(w)
        pol[f"vote 1/w{w}"] = (lambda w: lambda a, b: "L" if (b >= w * a and b > 0) or a == 0 else "I")(w)
    for name, f in pol.items():
        for th, K in ((40, 6), (48, 8)):
            c, si, ai, sl, al, sc, scl = simulate_policy(paths, th, K, f)
            print(f"{name:12s} T{th} K{K}: {c:.1f} wave-instr/seg; inner {si} x {ai:.1f}; leaf {sl} x {al:.1f}; sched {sc} x {scl:.1f}")
