"""Condense the rocprofv3 CSVs of scripts/collect_profiles.sh into small files for profiles/.

Writes (next to the inputs): kernel_stats.csv (our kernels + the GEMMs), pmc_summary.json with
per-kernel counter sums and the derived numbers quoted in DESIGN.md, and pmc_traffic.json
(HBM bytes per search of the mips_filter kernel; FETCH_SIZE doubled per the gfx950 note in
MI355X_MICROARCH.md section HBM).
"""
import collections
import csv
import json
import os
import sys

root = sys.argv[1]


def short(name):
    for key in ("mips_filter_f16", "topk_merge", "merge_lists", "prep_queries", "finalize_topk", "attention_fwd",
                "bias_gelu", "bias_residual_layernorm", "embed_layernorm", "pool_project", "Cijk_"):
        if key in name:
            return key if key != "Cijk_" else "hipblaslt_gemm(" + name.split("_MT")[1].split("_")[0] + ")" if "_MT" in name else "hipblaslt_gemm"
    return None


stats_in = os.path.join(root, "trace", "bench_kernel_stats.csv")
if os.path.exists(stats_in):
    rows = list(csv.DictReader(open(stats_in)))
    with open(os.path.join(root, "kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            n = short(r["Name"])
            if n:
                w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

summary = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_grbm"):
    path = os.path.join(root, sub, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if not n or n.startswith("hipblaslt"):
            continue
        per[n][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[n].add(r["Dispatch_Id"])
        if (sub, r["Dispatch_Id"]) not in seen:
            seen.add((sub, r["Dispatch_Id"]))
            dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n in per:
        d = summary.setdefault(n, {})
        d.setdefault("launches_" + sub, len(launches[n]))
        d.setdefault("duration_ns_" + sub, dur[n])
        d.update(per[n])

f = summary.get("mips_filter_f16", {})
derived = {}
searches = None
if "launches_pmc_fetch" in f:
    # bench.py --steps 2 --warmup 1 runs 3 searches of 8 rounds each
    searches = 3
    derived["searches_profiled"] = searches
    derived["FETCH_SIZE_KB_per_search_raw"] = f["FETCH_SIZE"] / searches
    derived["hbm_read_bytes_per_search"] = 2 * f["FETCH_SIZE"] * 1024 / searches   # gfx950: FETCH_SIZE counts 1/2
if "WRITE_SIZE" in f and searches:
    derived["hbm_write_bytes_per_search_uncalibrated"] = f["WRITE_SIZE"] * 1024 / searches
if "GRBM_GUI_ACTIVE" in f:
    clk = f["GRBM_GUI_ACTIVE"] / 8 / f["duration_ns_pmc_grbm"]   # 8 XCD instances summed
    derived["effective_clock_GHz_profiled"] = clk
if "SQ_VALU_MFMA_BUSY_CYCLES" in f and "SQ_BUSY_CYCLES" in f:
    # SQ_BUSY_CYCLES is summed over 32 shader engines; 1024 SIMDs
    cycles = f["SQ_BUSY_CYCLES"] / 32
    derived["mfma_pipe_busy_fraction"] = f["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles)
    derived["lds_active_fraction"] = f["SQ_LDS_IDX_ACTIVE"] / (256 * cycles)
    derived["wave_wait_fraction"] = f["SQ_WAIT_ANY"] / f["SQ_WAVE_CYCLES"]
    derived["wave_issue_stall_fraction"] = f["SQ_WAIT_INST_ANY"] / f["SQ_WAVE_CYCLES"]
summary["derived_mips_filter"] = derived
json.dump(summary, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
if "hbm_read_bytes_per_search" in derived:
    json.dump({"hbm_bytes_per_search": derived["hbm_read_bytes_per_search"],
               "note": "mips_filter_f16: 2 x FETCH_SIZE (gfx950 correction) summed over the rounds of one search; "
                       "algorithmic bytes are N*256 = 4.608e9"},
              open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
