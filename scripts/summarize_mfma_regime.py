"""One row per kernel variant from the passes of scripts/collect_mfma_regime.sh.

mfma_busy_of_gpu_active = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the fraction of GPU-active
cycles a SIMD's matrix pipe is busy (launch ramps and tails count as idle) -- the formula of DESIGN.md section 2.3.
mfma_busy_of_sq_busy divides by SQ_BUSY_CYCLES / 32 instead (cycles in which some wave is resident).
The two passes of a variant are different runs: durations and clocks come from the GRBM pass.
"""
import collections
import csv
import json
import os
import sys

root = sys.argv[1]


def load(sub, want):
    path = os.path.join(root, sub, "p_counter_collection.csv")
    per = collections.defaultdict(float)
    dur = 0.0
    seen = set()
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if not want(r["Kernel_Name"]):
            continue
        per[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per["launches"] = len(seen)
    per["duration_ns"] = dur
    return per


def row(prefix, want, flops_per_launch_set=None):
    sq, gr = load(prefix + "_sq", want), load(prefix + "_grbm", want)
    if not sq or not gr or not gr["duration_ns"]:
        return None
    o = {"launches": int(gr["launches"]), "duration_ms_total": gr["duration_ns"] / 1e6}
    o["effective_clock_GHz"] = gr["GRBM_GUI_ACTIVE"] / 8 / gr["duration_ns"]
    # counters of the SQ pass scaled to the GRBM pass by launch count (same workload)
    scale = gr["launches"] / max(sq["launches"], 1)
    o["mfma_busy_of_gpu_active"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] * scale / (1024 * gr["GRBM_GUI_ACTIVE"] / 8)
    o["mfma_busy_of_sq_busy"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * sq["SQ_BUSY_CYCLES"] / 32)
    o["wave_wait_fraction"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
    o["wave_issue_stall_fraction"] = sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
    o["wave_active_fraction"] = sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
    o["lds_bank_conflict_fraction"] = sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"] if sq["SQ_LDS_IDX_ACTIVE"] else 0.0
    # 32x32x16 MFMA = 32 busy cycles and 32768 flop
    o["TFLOPs_from_counters"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] * scale / 32 * 32768 / (gr["duration_ns"] * 1e-9) / 1e12
    return o


out = {
    "mips_filter_f16<QW=2, 8 waves> (shipped)": row("qw2", lambda n: "mips_filter_f16<2" in n),
    "mips_filter_f16<QW=4, 4 waves> (experiment)": row("qw4", lambda n: "mips_filter_f16<4" in n),
    "bare mfma loop, random operands": row("bare0", lambda n: "mfma_loop" in n),
    "bare mfma loop, zero operands": row("bare1", lambda n: "mfma_loop" in n),
    "note": "2032 queries x 18M rows (scripts/dev_search_timing.py: 13 searches per pass); bare loop: 4 launches of ~8 ms",
}
print(json.dumps(out, indent=1))
