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
    if "mips_filter_f16<1" in name:
        return "mips_filter_f16_qw1"      # the HBM-bound small-batch instantiation (bench.py scan_small_batch)
    if "mips_filter_i8_pairs<1" in name or "mips_filter_i8<1" in name:
        return "mips_filter_i8_qw1"
    for key in ("mips_filter_i8", "mips_filter_f16", "quantise_rows_i8", "column_stats", "prep_queries_i8", "topk_merge", "merge_sorted_lists", "merge_lists", "kmeans_assign", "segmented_mean", "bootstrap_scores", "bootstrap_select", "prep_queries",
                "finalize_topk", "gemm_tn_f16", "attention_fwd", "attention_cls_fwd", "bias_gelu", "bias_residual_layernorm",
                "embed_layernorm", "pool_project", "cls_dense_mfma", "small_dense_mfma", "gather_rows_kernel", "stream_copy", "stream_read", "mfma_loop", "Cijk_"):
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
def bench_line(log):
    """The JSON line bench.py printed under the profiler (last line of the pass's log)."""
    try:
        for line in reversed(open(os.path.join(root, log)).read().strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    except Exception:
        pass
    return {}


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
        if not n:
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

def derive(f, fp16_beside=False):
    """per-search figures of one filter kernel from its counter sums (fp16_beside: the fp16 scan the bench times beside the
    nomination scan -- its rounds per search are in the line's fp16_scan object, 6 where the int8 rounds are 8)"""
    derived = {}
    searches = None
    if "launches_pmc_fetch" in f:
        # full-size searches in the pass = launches of the QW=2 instantiation / rounds per search (bench line)
        line = bench_line("pmc_fetch.log")
        rounds = line.get("fp16_scan", {}).get("rounds", 6) if fp16_beside else line.get("config", {}).get("rounds", 8)
        searches = f["launches_pmc_fetch"] / rounds
        derived["rounds_per_search"] = rounds
        derived["searches_profiled"] = searches
        derived["FETCH_SIZE_KB_per_search_raw"] = f["FETCH_SIZE"] / searches
        derived["hbm_read_bytes_per_search"] = 2 * f["FETCH_SIZE"] * 1024 / searches   # gfx950: FETCH_SIZE counts 1/2
    if "WRITE_SIZE" in f and searches:
        derived["hbm_write_bytes_per_search_uncalibrated"] = f["WRITE_SIZE"] * 1024 / searches
    if "GRBM_GUI_ACTIVE" in f:
        clk = f["GRBM_GUI_ACTIVE"] / 8 / f["duration_ns_pmc_grbm"]   # 8 XCD instances summed
        derived["effective_clock_GHz_profiled"] = clk
        derived["avg_launch_us_profiled"] = f["duration_ns_pmc_grbm"] / max(f.get("launches_pmc_grbm", 1), 1) / 1e3
    if "SQ_VALU_MFMA_BUSY_CYCLES" in f and "SQ_BUSY_CYCLES" in f:
        # SQ_BUSY_CYCLES is summed over 32 shader engines; 1024 SIMDs
        cycles = f["SQ_BUSY_CYCLES"] / 32
        # two denominators: cycles in which a wave is resident somewhere (SQ busy), and all GPU-active cycles of the
        # launches (GRBM_GUI_ACTIVE / 8 XCDs, from the GRBM pass scaled by launch count) -- the latter counts launch ramps
        # and tails as idle and is the figure DESIGN.md section 2.3 quotes
        derived["mfma_pipe_busy_of_sq_busy"] = f["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles)
        if "GRBM_GUI_ACTIVE" in f and f.get("launches_pmc_grbm"):
            scale = f["launches_pmc_grbm"] / max(f.get("launches_pmc_sq", 1), 1)
            derived["mfma_pipe_busy_of_gpu_active"] = f["SQ_VALU_MFMA_BUSY_CYCLES"] * scale / (1024 * f["GRBM_GUI_ACTIVE"] / 8)
        derived["lds_active_fraction"] = f["SQ_LDS_IDX_ACTIVE"] / (256 * cycles)
        derived["wave_wait_fraction"] = f["SQ_WAIT_ANY"] / f["SQ_WAVE_CYCLES"]
        derived["wave_issue_stall_fraction"] = f["SQ_WAIT_INST_ANY"] / f["SQ_WAVE_CYCLES"]
    return derived


# the dominant kernel of the timed search: the int8 nomination scan when the bench ran it, with the fp16 scan beside it
nominated = "mips_filter_i8" in summary
derived = derive(summary.get("mips_filter_i8" if nominated else "mips_filter_f16", {}))
derived["kernel"] = "mips_filter_i8" if nominated else "mips_filter_f16"
if nominated and "mips_filter_f16" in summary:
    summary["derived_mips_filter_f16"] = derive(summary["mips_filter_f16"], fp16_beside=True)
g = summary.get("mips_filter_i8_qw1" if nominated else "mips_filter_f16_qw1", {})
if "FETCH_SIZE" in g:
    # bench.py's scan_small_batch leg: 6 searches of 32 queries (3 plain + 3 with HIP-event brackets) per scan
    derived["small_batch_hbm_read_bytes_per_search"] = 2 * g["FETCH_SIZE"] * 1024 / 6
summary["derived_mips_filter"] = derived

# encoder passes (collect_profiles.sh: enc_pmc_sq / enc_pmc_grbm run bench.py with a token-size search)
enc = {}
for sub in ("enc_pmc_sq", "enc_pmc_grbm"):
    path = os.path.join(root, sub, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    seen = set()
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if not n or n.startswith("mips_") or n in ("topk_merge", "prep_queries", "finalize_topk", "merge_lists"):
            continue
        d = enc.setdefault(n, collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        if (sub, r["Dispatch_Id"]) not in seen:
            seen.add((sub, r["Dispatch_Id"]))
            d["duration_ns_" + sub] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            d["launches_" + sub] += 1
derived_enc = {}
tot_busy = tot_cycles = 0.0
for n, d in enc.items():
    o = {}
    if d.get("SQ_BUSY_CYCLES"):
        cycles = d["SQ_BUSY_CYCLES"] / 32
        o["mfma_pipe_busy_of_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles)
        o["share_of_encoder_gpu_cycles"] = cycles
        tot_busy += d["SQ_VALU_MFMA_BUSY_CYCLES"]
        tot_cycles += cycles
    if d.get("GRBM_GUI_ACTIVE") and d.get("duration_ns_enc_pmc_grbm"):
        o["effective_clock_GHz_profiled"] = d["GRBM_GUI_ACTIVE"] / 8 / d["duration_ns_enc_pmc_grbm"]
        o["avg_duration_us"] = d["duration_ns_enc_pmc_grbm"] / d["launches_enc_pmc_grbm"] / 1e3
    derived_enc[n] = o
for o in derived_enc.values():
    if "share_of_encoder_gpu_cycles" in o and tot_cycles:
        o["share_of_encoder_gpu_cycles"] /= tot_cycles
if tot_cycles:
    derived_enc["whole_encoder"] = {"mfma_pipe_busy_of_sq_busy": tot_busy / (1024 * tot_cycles)}
if derived_enc:
    summary["derived_encoder"] = derived_enc
# encoder HBM traffic per full-size step (collect_profiles.sh: enc_pmc_fetch / enc_pmc_write, bench.py --skip-varlen):
# all encoder kernels of the pass together, over the number of steps (= embed_layernorm launches)
enc_traffic = {}
for sub, counter in (("enc_pmc_fetch", "FETCH_SIZE"), ("enc_pmc_write", "WRITE_SIZE")):
    path = os.path.join(root, sub, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    total, steps, per_kernel = 0.0, set(), collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if r["Counter_Name"] != counter or not n or n.startswith("mips_") or n.startswith("bootstrap") or \
                n in ("topk_merge", "prep_queries", "finalize_topk", "merge_lists"):
            continue
        total += float(r["Counter_Value"])
        per_kernel[n] += float(r["Counter_Value"])
        if n == "embed_layernorm":
            steps.add(r["Dispatch_Id"])
    if steps:
        scale = 2 * 1024 if counter == "FETCH_SIZE" else 1024     # KB; gfx950: FETCH_SIZE counts half
        key = "read" if counter == "FETCH_SIZE" else "write_uncalibrated"
        enc_traffic["steps_profiled_" + key] = len(steps)
        enc_traffic["hbm_%s_bytes_per_step" % key] = total * scale / len(steps)
        enc_traffic["hbm_%s_bytes_per_step_by_kernel" % key] = {k: v * scale / len(steps) for k, v in per_kernel.items()}
if enc_traffic:
    summary["derived_encoder_traffic"] = enc_traffic
json.dump(summary, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
if "hbm_read_bytes_per_search" in derived:
    traffic = {"hbm_bytes_per_search": derived["hbm_read_bytes_per_search"], "nomination": nominated,
               "note": derived["kernel"] + ": 2 x FETCH_SIZE (gfx950 correction) summed over the rounds of one search; algorithmic "
                       "bytes are N*256 = 4.608e9 (fp16 rows; the int8 nomination scan streams N*128 = 2.304e9 of its copy)"}
    if "derived_mips_filter_f16" in summary and "hbm_read_bytes_per_search" in summary["derived_mips_filter_f16"]:
        traffic["fp16_scan_hbm_bytes_per_search"] = summary["derived_mips_filter_f16"]["hbm_read_bytes_per_search"]
    if "hbm_read_bytes_per_step" in enc_traffic:
        traffic["encode_hbm_read_bytes_per_step"] = enc_traffic["hbm_read_bytes_per_step"]
        traffic["encode_step_shape"] = [512, 128]
        traffic["encode_note"] = "all kernels of one 512 x 128 bert-base encode step, 2 x FETCH_SIZE"
    json.dump(traffic, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
print(json.dumps(summary.get('derived_encoder', {}), indent=1))
