"""Copy one scripts/final_run.sh collection from gpurun_out/ into profiles/ under the round's names, stamp the PMC traffic
file with the commit whose kernels were measured, and print the figures profiles/README.md quotes.
usage: python scripts/install_profiles.py <run, e.g. r04c> <round, e.g. r04> [commit]"""
import csv, json, shutil, subprocess, sys

run, rnd = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
P, F = f"gpurun_out/{run}_prof", f"gpurun_out/{run}_final"
for src, dst in [(f"{P}/kernel_stats.csv", "kernel_stats.csv"), (f"{P}/kernel_stats_full.csv", "kernel_stats_full.csv"),
                 (f"{P}/bench_under_rocprof.json", "bench_under_rocprof.json"), (f"{P}/pmc_summary.json", "pmc_summary.json"),
                 (f"{F}/bench_default.json", "bench_unprofiled.json"), (f"{F}/bench_default_full.json", "bench_unprofiled_full.json"),
                 (f"{F}/gpu_tests.txt", "gpu_tests.txt"), (f"{F}/bench_rccl_1rank_cabi.json", "bench_rccl_1rank_cabi.json"),
                 (f"{F}/bench_rccl_1rank_torch.json", "bench_rccl_1rank_torch.json"),
                 (f"{F}/bench_2ranks_one_gpu_gloo.json", "bench_2ranks_one_gpu_gloo.json")]:
    shutil.copy(src, f"profiles/{rnd}_{dst}")
t = json.load(open(f"{P}/pmc_traffic.json"))
t["measured_on_commit"] = commit
for dst in ("profiles/pmc_traffic.json", f"profiles/{rnd}_pmc_traffic.json"):
    json.dump(t, open(dst, "w"), indent=1)
print("stamped", commit)
for r in list(csv.DictReader(open(f"profiles/{rnd}_kernel_stats.csv")))[:8]:
    print(f'  {r["Name"][:34]:34s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us')
for f in ("bench_under_rocprof", "bench_unprofiled", "bench_rccl_1rank_cabi", "bench_rccl_1rank_torch", "bench_2ranks_one_gpu_gloo"):
    b = json.load(open(f"profiles/{rnd}_{f}.json"))
    print(f'  {f}: {b["value"]:.0f} q/s, {b["ms_per_step"]:.3f} ms/step, filter {b["roofline"]["filter_ms_per_search"]:.3f} ms, frac {b["roofline"]["frac"]:.4f}')
b = json.load(open(f"profiles/{rnd}_bench_unprofiled.json"))
e = b["encode"]
print("  fp16_scan beside it:", b.get("fp16_scan"))
print("  shard_sweep", [round(p, 3) for p in b["shard_sweep"]["ms_per_search"]], "large_k",
      {k: round(v["ms_per_search"], 2) for k, v in b["large_k"].items()}, "small batch frac", round(b["scan_small_batch"]["roofline"]["frac"], 3))
print("  encode", round(e["value"]), round(e["roofline"]["frac"], 4), "varlen", round(e["varlen"]["value"]), "corpus_1m", round(e["corpus_1m"]["value"]),
      "cli loops", round(e["cli_text"]["encode_loop"]["passages_per_s"]), round(e.get("cli_text_non_ascii", e["cli_text"])["encode_loop"]["passages_per_s"]),
      "seq512", e.get("seq512"), "batch300", e.get("batch300"), "gemm", e.get("gemm_kernel"))
print("  cli_eval", round(b["search_cli_eval"]["value"], 2), "s; kmeans", round(b["kmeans"]["ms_per_iteration"], 1), "ms; online",
      round(b["online"]["k80"]["ms_per_question"], 2), round(b["online"]["k5000"]["ms_per_question"], 2), "ms; float32", round(b["float32_index"]["value"]),
      "; cpu", round(b["cpu_baseline"]["value"]), "q/s; peaks", {k: round(v) for k, v in b["peak_measured"].items() if isinstance(v, float)})
s = json.load(open(f"profiles/{rnd}_pmc_summary.json"))
d = s["derived_mips_filter"]
print("  filter PMC: hbm bytes/search", round(d["hbm_read_bytes_per_search"] / 1e9, 3), "GB, mfma busy of active", round(d["mfma_pipe_busy_of_gpu_active"], 3),
      "of sq-busy", round(d["mfma_pipe_busy_of_sq_busy"], 3), "clock", round(d["effective_clock_GHz_profiled"], 2), "stall", round(d["wave_issue_stall_fraction"], 2))
if "derived_mips_filter_f16" in s:
    print("  fp16 scan PMC:", {a: (round(v, 3) if isinstance(v, float) else v) for a, v in s["derived_mips_filter_f16"].items()})
for k in ("gemm_tn_f16", "hipblaslt_gemm(256x256x64)", "whole_encoder"):
    if k in s.get("derived_encoder", {}):
        print("  encoder PMC", k, {a: round(v, 3) for a, v in s["derived_encoder"][k].items()})
