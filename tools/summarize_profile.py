"""Condenses a gpurun_out/prof_<tag>/ directory (tools/profile_gpu.sh) into tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_pmc.json           per-launch means of every PMC counter for trace_kernel + derived figures
  profiles/traffic.json             key "<scene>:<traversal>" -> HBM bytes per trace_kernel launch (read by bench.py)
Usage: python tools/summarize_profile.py <tag> <scene> <traversal>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, scene, traversal):
    src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
    dst = os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)
    # gpurun merges new files next to those of earlier runs: only the newest file of every pass counts
    newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)
    ks = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    shutil.copy(ks, os.path.join(dst, tag + "_kernel_stats.csv"))
    kern = {}
    for r in csv.DictReader(open(ks)):
        if "trace_kernel" in r["Name"]:
            kern = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
    agg = collections.defaultdict(list)
    meta = {}
    for pass_dir in glob.glob(os.path.join(src, "pmc_*")):
        f = newest(os.path.join(pass_dir, "*", "*_counter_collection.csv"))
        for r in csv.DictReader(open(f)):
            if "trace_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "lds_block_size": int(r["LDS_Block_Size"]),
                        "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "scratch": int(r["Scratch_Size"])}
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    d = {}
    # What rocprofv3 reports per dispatch is NOT the kernel's footprint: LDS_Block_Size counts static LDS only (this kernel's
    # LDS is dynamic) and VGPR_Count comes out at half the allocation.  The authoritative figures are the ones the library
    # reports for the very launch (pt_scene_get_info), which bench.py prints: taken from the profiled run's own JSON line.
    dispatch = {"rocprofv3_reported": meta}
    bl = os.path.join(src, "bench_line.json")
    if os.path.exists(bl) and os.path.getsize(bl):
        cfg = json.load(open(bl)).get("config", {})
        dispatch.update({k: cfg[k] for k in ("grid", "lds_bytes", "vgprs", "blocks_per_cu", "blocks_per_cu_occupancy_limit", "lds_scene") if k in cfg})
    meta = dispatch
    if all(k in c for k in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
        # exact bytes the L2 requested from the fabric (Infinity Cache / HBM), by request size
        d["l2_miss_read_bytes"] = 32.0 * c["TCC_EA0_RDREQ_32B_sum"] + 64.0 * c["TCC_EA0_RDREQ_64B_sum"] + 128.0 * c["TCC_EA0_RDREQ_128B_sum"]
        d["l2_miss_read_requests_by_size"] = {"32B": c["TCC_EA0_RDREQ_32B_sum"], "64B": c["TCC_EA0_RDREQ_64B_sum"], "128B": c["TCC_EA0_RDREQ_128B_sum"]}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE = requests x 64 B, so 128-B requests are
        # tallied at half their bytes -> doubled here.  (Checked on bunny: every fabric read of this kernel IS a 128-B request,
        # TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ, so the doubling is exact; profiles/r02_bunny_memsys_pmc_baseline.json.)
        # WRITE_SIZE is exact for 16-B-per-lane stores (the per-sample float4 stores here).
        d["hbm_read_bytes"] = 2.0 * c["FETCH_SIZE"] * 1024
        d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
        d["l2_miss_bytes_per_launch"] = d.get("l2_miss_read_bytes", d["hbm_read_bytes"]) + d["hbm_write_bytes"]
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["valu_lane_utilization"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VALU" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                 # summed over the 8 XCDs
        d["kernel_cycles"] = cycles
        d["valu_issue_busy"] = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cycles)    # wave64 VALU = 2 cycles on a SIMD-32, 1024 SIMDs
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "kernel_cycles" in d:
        d["l1_tag_lookups_per_cycle_per_cu"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / d["kernel_cycles"] / 256.0
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in c:
                d[k.lower() + "_share_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
        d["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    out = {"tag": tag, "scene": scene, "traversal": traversal, "command": f"python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --scene {scene}",
           "kernel_trace": kern, "dispatch": meta, "counters_mean_per_launch": c, "derived": d}
    json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
    tpath = os.path.join(dst, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    if "hbm_bytes_per_launch" in d:
        traffic[f"{scene}:{traversal}"] = round(d["hbm_bytes_per_launch"])
        json.dump(traffic, open(tpath, "w"), indent=1)
    print(json.dumps({"kernel": kern, "derived": d}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
