"""CPU simulation (real event traces from the oracle, tests/tools/sim_schedule.py) of a schedule with R rays per lane that ALL traverse:
in a step of one kind a lane advances whichever of its rays needs that kind; the scheduler phase serves one ray per lane.
Costs: inner step 45, leaf step 65, scheduler phase 440 wave-instructions, + c_sel for picking a ray's registers per step.
-> profiles/r03_schedule_simulation_two_rays.log"""
import sys
sys.path.insert(0,'/root/repo/tests/tools'); sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import sim_schedule as S
Ray=S.Ray
def sim(paths, rays_per_lane, thresh, n_in=6, n_lf=2, c_in=45, c_lf=65, c_sched=440, c_sel=0):
    it=iter(paths); n=64; R=rays_per_lane
    A=[[None]*R for _ in range(n)]
    cost=steps_in=steps_lf=act_in=act_lf=sched=sched_l=0
    work_left=True; total=sum(len(p) for p in paths)
    def new_ray():
        nonlocal work_left
        try: return Ray(next(it))
        except StopIteration:
            work_left=False; return None
    while True:
        # lanes that can make progress in a scheduler phase: have a finished ray, or an empty slot and work left
        def idle_slot(l):
            for j in range(R):
                a=A[l][j]
                if a is None:
                    if work_left: return j
                elif a.kind() is None: return j
            return -1
        pend=[idle_slot(l) for l in range(n)]
        npend=sum(1 for p in pend if p>=0)
        anytrav=any(a is not None and a.kind() is not None for l in range(n) for a in A[l])
        if npend>=thresh or not anytrav:
            if npend==0: break
            sched+=1; sched_l+=npend; cost+=c_sched+c_sel
            for l in range(n):
                j=pend[l]
                if j<0: continue
                a=A[l][j]
                if a is not None and not a.shade(): A[l][j]=None
                if A[l][j] is None: A[l][j]=new_ray()
        for kind,cnt,c in (("I",n_in,c_in),("L",n_lf,c_lf)):
            for _ in range(cnt):
                act=0
                for l in range(n):
                    for a in A[l]:
                        if a is not None and a.kind()==kind:
                            a.step(); act+=1; break
                if act:
                    cost+=c+c_sel
                    if kind=="I": steps_in+=1; act_in+=act
                    else: steps_lf+=1; act_lf+=act
    return cost/total, steps_in, act_in/max(steps_in,1), steps_lf, act_lf/max(steps_lf,1), sched, sched_l/max(sched,1)
name=sys.argv[1] if len(sys.argv)>1 else "cbox"
paths=S.load_paths(name, 64*24, 16)
print(name,len(paths),"paths (caller's tree traces)")
for R,th,sel in ((1,40,0),(2,40,14),(2,48,14),(2,32,14),(2,56,14),(3,48,20)):
    for (ni,nl) in ((6,2),(4,2),(8,3)):
        r=sim(paths,R,th,ni,nl,c_sel=sel)
        print(f"R{R} T{th} {ni}+{nl}: {r[0]:.0f} wave-instr/seg; inner {r[1]} x {r[2]:.1f}; leaf {r[3]} x {r[4]:.1f}; sched {r[5]} x {r[6]:.1f}")
