#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output (gpurun_out/<run>/vm_results.db) into the small
summaries kept under profiles/: per-kernel stats (CSV) and PMC counters of the render kernel (JSON).

  python3 tools/summarize_profiles.py stats gpurun_out/prof4/vm_results.db profiles/r01_bench_kernel_stats_v3.csv
  python3 tools/summarize_profiles.py pmc gpurun_out/pmc4 profiles/r01_pmc_v3_many_lights_50spp.json [algorithmic_bytes] [command text]
"""
import csv
import glob
import json
import os
import sqlite3
import sys


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        w.writerows(rows)
    for r in rows[:4]:
        print(r[0][:60], r[1], round(r[3], 1), "us avg")


def pmc(root, out, algorithmic=None, command=None):
    res = {"command": (command or "rocprofv3 --pmc <pass> -- python3 tools/prof_run.py --spp 50 --batches 2") +
                      " (one run per pass; values of the LAST render_kernel<false, ...> dispatch, i.e. the re-ordered launch)", "passes": {}}
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        db = os.path.join(d, "vm_results.db")
        if not os.path.exists(db):
            continue
        c = sqlite3.connect(db)
        rows = c.execute("select dispatch_id, counter_name, value, duration from counters_collection "
                         "where kernel_name like '%render_kernel<false%'").fetchall()
        if not rows:
            continue
        last = max(r[0] for r in rows)
        agg = {}
        for disp, name, value, dur in rows:
            if disp == last:
                agg[name] = agg.get(name, 0.0) + float(value)
                agg["_duration_ns"] = dur
        kn = c.execute("select distinct kernel_name from counters_collection where kernel_name like '%render_kernel<false%'").fetchall()
        res["kernel"] = kn[-1][0] if kn else None
        res["passes"][os.path.basename(d)] = agg
    P = res["passes"]
    der = {}
    if "f" in P and "w" in P:
        F = P["f"]["FETCH_SIZE"] * 1024
        W = P["w"]["WRITE_SIZE"] * 1024
        der.update({"fetch_bytes_reported": F, "fetch_bytes_x2_gfx950_correction": 2 * F, "write_bytes": W,
                    "hbm_traffic_per_launch": 2 * F + W,
                    "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH doubled per MI355X_MICROARCH.md (16 B/lane loads); the access "
                            "pattern here is gathered 16 B/lane, so the absolute is uncalibrated. Infinity-Cache hits are "
                            "counted, so this is fabric traffic, an upper bound on HBM traffic."})
        if algorithmic:
            der["algorithmic_bytes_per_launch"] = int(algorithmic)
    if "p2" in P and "SQ_THREAD_CYCLES_VALU" in P["p2"]:
        der["valu_lane_utilisation"] = P["p2"]["SQ_THREAD_CYCLES_VALU"] / (64 * P["p2"]["SQ_ACTIVE_INST_VALU"])
    if "p2" in P and "p4" in P and "GRBM_GUI_ACTIVE" in P["p4"]:
        # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles; 256 CUs x 4 SIMDs; a wave64 VALU instruction holds its SIMD 4 cycles
        cycles = P["p4"]["GRBM_GUI_ACTIVE"] / 8.0
        der["launch_cycles_per_xcd"] = cycles
        der["valu_busy"] = P["p2"]["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles)
        if "valu_lane_utilisation" in der:
            der["valu_useful_issue"] = der["valu_busy"] * der["valu_lane_utilisation"]
        der["valu_wave_instructions"] = P["p2"]["SQ_ACTIVE_INST_VALU"]
    if "p3" in P and "TCC_REQ_sum" in P["p3"]:
        der["l2_hit_rate"] = P["p3"]["TCC_HIT_sum"] / P["p3"]["TCC_REQ_sum"]
    if "p1" in P and "p2" in P and "SQ_WAVE_CYCLES" in P["p1"]:
        wc = P["p1"]["SQ_WAVE_CYCLES"]
        der["wave_cycle_shares"] = {"issuing": P["p2"]["SQ_ACTIVE_INST_ANY"] / wc, "waiting_on_memory_or_barrier": P["p2"]["SQ_WAIT_ANY"] / wc,
                                    "issue_stalled": P["p1"]["SQ_WAIT_INST_ANY"] / wc}
    if "p1" in P and "launch_cycles_per_xcd" in der and "SQ_WAVE_CYCLES" in P["p1"]:
        der["resident_waves_per_simd"] = P["p1"]["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * der["launch_cycles_per_xcd"])  # counter ticks every 4 cycles
    if "p4" in P and "TCP_TOTAL_CACHE_ACCESSES_sum" in P["p4"]:
        der["l1_accesses"] = P["p4"]["TCP_TOTAL_CACHE_ACCESSES_sum"]
        der["l1_to_l2_read_requests"] = P["p4"]["TCP_TCC_READ_REQ_sum"]
    if "hbm_traffic_per_launch" in der and "f" in P:
        der["launch_ms"] = P["f"]["_duration_ns"] / 1e6
        der["fabric_GBps"] = der["hbm_traffic_per_launch"] / P["f"]["_duration_ns"]
        if algorithmic:
            der["traffic_over_algorithmic"] = der["hbm_traffic_per_launch"] / float(algorithmic)
    if "p5" in P and "SQ_INSTS_VALU_FLOPS_FP32" in P["p5"]:
        # gfx950's FLOPS counters add, per wave-level instruction, the fp32 operations ONE lane performs (packed = 2, fma = 2):
        # x 64 = what the SIMD issued for, x the lane utilisation = what the active lanes computed
        f = P["p5"]["SQ_INSTS_VALU_FLOPS_FP32"] + P["p5"].get("SQ_INSTS_VALU_FLOPS_FP32_TRANS", 0.0)
        der["fp32_flops_issued"] = 64.0 * f
        if "valu_lane_utilisation" in der:
            der["fp32_flops_active_lanes"] = 64.0 * f * der["valu_lane_utilisation"]
        der["fp32_tflops_issued"] = 64.0 * f / P["p5"]["_duration_ns"] / 1e3
        if "valu_lane_utilisation" in der:
            der["fp32_tflops_active_lanes"] = der["fp32_tflops_issued"] * der["valu_lane_utilisation"]
        der["valu_instruction_mix_wave_level"] = {k[len("SQ_INSTS_VALU_"):].lower(): P["p5"][k] for k in P["p5"] if k.startswith("SQ_INSTS_VALU_") and "FLOPS" not in k}
    res["derived"] = der
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(der, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None, sys.argv[5] if len(sys.argv) > 5 else None)
