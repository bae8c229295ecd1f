"""bench.py the way the driver runs it at N > 1: `python bench.py --gpus N` with no launcher around it starts its own
ranks (a child torch.distributed.run) and prints rank 0's one JSON line.  The test box has one GPU, so the two ranks
share it and exchange over gloo (RCCL refuses two ranks per device); the kernels, the sharding and the merge are the
real ones, and the merged result must be the single-rank result bit for bit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(gpus, extra_env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--rows", "400000", "--queries", "64",
           "--steps", "2", "--warmup", "1", "--skip-encode", "--skip-float32", "--skip-cpu", "--skip-extras"]
    env = dict(os.environ, **(extra_env or {}))
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]            # exactly one line on stdout
    assert len(lines[0]) <= 6000                        # the driver keeps ~8 KB of tail: every leg must be inside it
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(gpu_device):
    one = _run(1)
    two = _run(2, {"PROQA_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["world_size"] == 2 and two["config"]["backend"] == "gloo"
    ranks = two["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and ranks[0]["pid"] != ranks[1]["pid"]
    assert ranks[0]["rows"] == [0, 200000] and ranks[1]["rows"] == [200000, 400000]
    assert "sharded_step" in two and two["sharded_step"]["rows_per_rank"] == 200000
    assert two["scaling"] == "strong" and two["value"] > 0
    # sharded == unsharded, bit for bit (ids and scores of all 64 x 80 results)
    assert two["result"] == one["result"]


def test_two_ranks_also_time_query_shards_over_replicated_rows(gpu_device):
    """N > 1 without --skip-extras: beside the row-sharded `value` the line carries the same job with the rows replicated and
    the queries sharded (65 queries on 2 ranks: 33 + 32, the second slice padded), its result equal to the row-sharded one."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "400000", "--queries", "65", "--steps", "2",
           "--warmup", "1", "--skip-encode", "--skip-float32", "--skip-cpu"]
    env = dict(os.environ, PROQA_DIST_BACKEND="gloo")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 6000
    line = json.loads(lines[0])
    qs = line["query_shards"]
    assert "error" not in qs, qs
    assert qs["queries_per_rank"] == 33 and qs["rows_per_rank"] == 400000 and qs["value"] > 0
    assert qs["ids_equal"] and qs["scores_equal"]


def test_the_full_line_fits_the_drivers_tail(gpu_device):
    """The default N = 1 run with every leg (shrunk: 1M rows, 3 steps, 20 k passages) prints one line of <= 6000 bytes that
    still carries the headline, the roofline object, the fp16 scan beside the nomination scan, and the secondary legs."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "1000000", "--queries", "600", "--steps", "3", "--warmup", "1",
           "--encode-steps", "2", "--corpus-passages", "4096", "--cli-passages", "4096"]
    env = dict(os.environ)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 6000, len(lines[0])
    line = json.loads(lines[0])
    for key in ("roofline", "cpu_baseline", "scan_small_batch", "shard_sweep", "large_k", "search_cli_eval", "kmeans", "online",
                "float32_index", "encode", "recall_parity"):
        assert key in line, key
    assert line["roofline"]["nomination"] == "int8" and line["fp16_scan"]["ids_equal"] and line["fp16_scan"]["scores_equal"]
    # the headline fraction is priced against the peak of the instruction that ran (int8 MFMA), the fp16-equivalent figure
    # beside it; the re-scoring gather and the non-filter part of the step are reported
    rf = line["roofline"]
    assert rf["peak"] == 5000.0 and rf["unit"] == "TOP/s" and rf["instruction"] == "v_mfma_i32_32x32x32_i8"
    assert abs(rf["frac"] - rf["achieved"] / 5000.0) < 1e-3 * rf["frac"]
    assert abs(rf["fp16_equivalent_frac"] - 2.0 * rf["frac"]) < 1e-3 * rf["fp16_equivalent_frac"]
    assert rf["hbm_bytes_per_row_scanned"] == 128 and rf["rescore_gather_bytes"] > 0 and rf["chain_ms_per_search"] > 0
    assert line["fp16_scan"]["frac"] > 0
    for key in ("mfma_i8_TOPs_8ms_random_operands", "mfma_i8_TOPs_8ms_zero_operands", "mfma_i8_16x16x64_TOPs_8ms_random",
                "mfma_f16_TFLOPs_8ms_random_operands"):
        assert line["peak_measured"][key] > 100.0, key
    assert line["encode"]["gemm_kernel"]


def test_bench_refuses_a_launcher_of_another_size(gpu_device):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "1000"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
