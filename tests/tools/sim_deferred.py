"""CPU simulation of a schedule variant for trace_kernel_v2: deferred leaf tests.
In exact traversal the box tests never look at the current best hit, so a lane can put the leaves it meets into a small
per-lane FIFO and keep visiting inner nodes; leaf steps then serve the oldest queued leaf of every lane that has one
(the per-lane order of leaf tests — and with it the tie-breaking — is unchanged).
Event traces come from the oracle (tests/tools/sim_schedule.py).  Costs are wave-instruction estimates from the ISA."""
import sys

from sim_schedule import Ray, load_paths

C_IN, C_LF, C_SCHED = 70, 100, 440
C_ENQ = 8          # extra instructions of an inner step that can enqueue + pop
C_DEQ = 6


def simulate(paths, mode, thresh=40, burst=(4, 1), qcap=4):
    it = iter(paths)
    n = 64
    A = [None] * n
    Q = [0] * n                                    # queued leaves per lane
    cost = steps_in = steps_lf = act_in = act_lf = sched = sched_lanes = 0
    work_left = True
    total_segs = sum(len(p) for p in paths)
    deferred = mode != "base"

    def traversing(l):
        return A[l] is not None and (A[l].kind() is not None or Q[l] > 0)

    while True:
        trav = [traversing(l) for l in range(n)]
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

        def inner_step():
            nonlocal cost, steps_in, act_in
            act = 0
            for l in range(n):
                a = A[l]
                if a is None:
                    continue
                did = False
                if deferred and a.kind() == "L" and Q[l] < qcap:
                    Q[l] += 1; a.step(); did = True
                if a.kind() == "I":
                    a.step(); did = True
                    # the real step leaves a leaf child in `cur`; it is enqueued by the next step
                act += did
            if act:
                cost += C_IN + (C_ENQ if deferred else 0); steps_in += 1; act_in += act
            return act

        def leaf_step():
            nonlocal cost, steps_lf, act_lf
            act = 0
            for l in range(n):
                a = A[l]
                if a is None:
                    continue
                if deferred:
                    if Q[l] > 0:
                        Q[l] -= 1; act += 1
                elif a.kind() == "L":
                    a.step(); act += 1
            if act:
                cost += C_LF + (C_DEQ if deferred else 0); steps_lf += 1; act_lf += act
            return act

        if mode in ("base", "fixed"):
            for _ in range(burst[0]):
                inner_step()
            for _ in range(burst[1]):
                leaf_step()
        else:   # "vote": K steps, each the kind more lanes can use; leaf demand = lanes with queued leaves, weight w
            K, w = burst
            for _ in range(K):
                n_in = sum(1 for l in range(n) if A[l] is not None and (A[l].kind() == "I" or (A[l].kind() == "L" and Q[l] < qcap)))
                n_lf = sum(1 for l in range(n) if A[l] is not None and Q[l] > 0)
                n_blocked = sum(1 for l in range(n) if A[l] is not None and Q[l] > 0 and (A[l].kind() is None or (A[l].kind() == "L" and Q[l] >= qcap)))
                if n_in == 0 and n_lf == 0:
                    break
                if n_in >= w * n_lf and n_in > n_blocked:
                    inner_step()
                else:
                    leaf_step()
    return dict(cost_per_seg=cost / total_segs, inner_steps=steps_in, inner_act=act_in / max(steps_in, 1), leaf_steps=steps_lf,
                leaf_act=act_lf / max(steps_lf, 1), sched=sched, sched_lanes=sched_lanes / max(sched, 1), segs=total_segs)


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cbox"
    paths = load_paths(name, 64 * 24, 16)
    print(name, len(paths), "paths;", "costs: inner", C_IN, "leaf", C_LF, "sched", C_SCHED, "enqueue +", C_ENQ, "dequeue +", C_DEQ)
    runs = [("base", 40, (4, 1), 0), ("base", 32, (4, 1), 0), ("base", 40, (3, 1), 0)]
    for q in (2, 4, 8):
        for b in ((4, 1), (5, 1), (6, 1), (8, 2), (6, 2)):
            runs.append(("fixed", 40, b, q))
    for q in (4, 8):
        for b in ((6, 0.5), (6, 0.75), (6, 1.0), (8, 0.75)):
            runs.append(("vote", 40, b, q))
    for mode, th, b, q in runs:
        r = simulate(paths, mode, th, b, q)
        print(f"{mode:5s} T{th} burst{b} qcap{q}: {r['cost_per_seg']:.1f} wave-instr/seg; inner {r['inner_steps']} x {r['inner_act']:.1f} lanes; "
              f"leaf {r['leaf_steps']} x {r['leaf_act']:.1f}; sched {r['sched']} x {r['sched_lanes']:.1f}", flush=True)
