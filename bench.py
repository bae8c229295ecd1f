"""Benchmark of the hot path on MI355X: exhaustive top-80 MIPS over an 18M x 128 fp16 index
(BASELINE.json configs[2]/[3]) as the headline line, plus the bert-base encode leg
(configs[1]) as a secondary object of the same JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A search "step" is one pass of all 2032 queries over the whole corpus (row-sharded over the N
ranks, one RCCL all-gather of the per-shard top-80 lists, GPU merge).  Inputs are synthetic,
generated on the device from fixed seeds and resident in HBM before the timed region.  An encode
"step" is one batch of 512 pre-tokenised 128-token passages through the bert-base tower.
Rank 0 prints ONE JSON line of at most ~6 KB (the driver keeps the tail of stdout): short machine keys, floats rounded to
five significant digits, no prose.  What the keys mean:

  value / ms_per_step      whole-job queries/s of the timed steps (default scan: the int8 nomination rounds + exact re-scoring)
  roofline                 dominant kernel of the timed search: the ALGORITHMIC 2 Q N d multiply-add flops over the HIP-event time
                           of its filter launches, against the dense peak of the instruction the kernel issues (int8 MFMA for
                           the nomination scan, fp16 MFMA for the fp16 scan); .fp16_equivalent_frac = the same flops against
                           the fp16 peak; .traffic = HBM bytes of those launches from the committed PMC pass
                           (profiles/pmc_traffic.json, stamped with its commit), null when that pass measured the other scan;
                           .rescore_gather_bytes = fp16 rows the merges gather for the exact re-scoring
  fp16_scan                the same search with the nomination switched off (mips_filter_f16), digests compared
  config.leap_rank /       leaping rounds: the rank (< k) of the running lists the rounds tested against, and the same search on
  config.ordinary_rounds   ordinary rounds (thresholds at the k-th best) timed beside it, digests compared
  query_shards             N > 1 only: the same job with the rows REPLICATED on every rank and the queries sharded (one all-gather
                           of the result rows, no rank merge); `value` stays the row-sharded figure of BASELINE configs[3]
  scan_small_batch         32 queries over the same rows: the HBM-bound regime (algorithmic bytes = rows x 256 B)
  shard_sweep              N=1 timing of the per-rank search of a G-rank job (first N/G rows, no collective); .pipelined_ms_per_search:
                           the same searches as a stream of batches on two handles / two streams (PipelinedSearcher)
  large_k                  the reference's large-k callers (trec_process.py:76, online_sampler.py:113) on the same rows
  search_cli_eval          the eval_retrieval.py command line end to end (index file -> printed Recall lines), stage split
  peak_measured            stream / MFMA micro-benchmarks of this box (float4 copy / read of 2 GiB; register-resident MFMA loops)
  kmeans, online           SURVEY section 8(f) rows on the same resident rows
  cpu_baseline             the NumPy restatement of eval_retrieval.py:98-104 on this host's cores, bounded sample
  recall_parity            GPU vs that restatement: top-k id overlap on the same sample
  float32_index, encode    the exact-float32 index leg and the bert-base encode leg (configs[1]) with their own rooflines
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F16_TFLOPS = 2500.0   # dense fp16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_MFMA_I8_TOPS = 5000.0      # dense int8 MFMA peak (2 x the fp16 rate: v_mfma_i32_32x32x32_i8), same guide
PEAK_HBM_GBS = 8000.0           # HBM3E spec peak, same table
D = 128
GEN_CHUNK = 250_000             # rows per seeded generation chunk (divides every 18M/G shard)
ENCODE_GFLOP_PER_PASSAGE = 22.35  # 128-token passage through bert-base


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=18_000_000)
    p.add_argument("--queries", type=int, default=2032)
    p.add_argument("--topk", type=int, default=80)
    p.add_argument("--encode-steps", type=int, default=8)
    p.add_argument("--encode-batch", type=int, default=512)
    p.add_argument("--seq-len", type=int, default=128)
    p.add_argument("--skip-encode", action="store_true")
    p.add_argument("--corpus-passages", type=int, default=1_000_000,
                   help="passages of the corpus-scale encode leg (BASELINE.json configs[1]); 0 skips it")
    p.add_argument("--cli-passages", type=int, default=200_000,
                   help="text passages of the command-line encode leg (encode.cli_text); 0 skips it")
    p.add_argument("--skip-float32", action="store_true", help="skip the exact-float32 index leg")
    p.add_argument("--skip-cli-eval", action="store_true", help="skip the eval_retrieval.py command-line leg (search_cli_eval)")
    p.add_argument("--skip-cpu", action="store_true")
    p.add_argument("--skip-varlen", action="store_true", help="encode leg: full-length steps only (PMC traffic passes)")
    p.add_argument("--skip-extras", action="store_true",
                   help="skip the shard sweep and the measured-peak micro-benchmarks (profiler passes: keeps the kernel "
                        "statistics to the searches of the headline workload)")
    p.add_argument("--transport", choices=["torch", "cabi"], default=os.environ.get("PROQA_SHARDED_TRANSPORT", "torch"),
                   help="who runs the all-gather of the sharded search: torch.distributed (RCCL backend) or the "
                        "library's own RCCL communicator (proqa_sharded_search_device)")
    p.add_argument("--force-collective", action="store_true",
                   help="run the all-gather + list merge even with one rank (exercises the RCCL path on one GPU)")
    return p.parse_args()


def gen_rows(lo, hi, device):
    """Rows [lo, hi) of the synthetic corpus: chunk c is randn(seed=1000+c) rounded to fp16, so
    the corpus is the same for every sharding."""
    out = torch.empty((hi - lo, D), dtype=torch.float16, device=device)
    g = torch.Generator(device=device)
    c0, c1 = lo // GEN_CHUNK, (hi + GEN_CHUNK - 1) // GEN_CHUNK
    for c in range(c0, c1):
        g.manual_seed(1000 + c)
        chunk = torch.randn((GEN_CHUNK, D), generator=g, device=device, dtype=torch.float32).to(torch.float16)
        a, b = max(lo, c * GEN_CHUNK), min(hi, (c + 1) * GEN_CHUNK)
        out[a - lo:b - lo] = chunk[a - c * GEN_CHUNK:b - c * GEN_CHUNK]
    return out


def gen_queries(nq, device):
    g = torch.Generator(device=device)
    g.manual_seed(1)
    return torch.randn((nq, D), generator=g, device=device, dtype=torch.float32).to(torch.float16)


def timed(fn, steps, warmup, world, device):
    """W untimed steps, then exactly K steps between barrier+synchronize; max over ranks (seconds)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def host_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU
    boxes expose 256 logical CPUs under a 16-CPU quota; oversubscribing the quota slows BLAS down)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_search_baseline(xb_sample, xq_sample, k, rows_total, cores):
    """The reference's CPU path (eval_retrieval.py:98-104 restated in NumPy: float32 upcast, BLAS
    sgemm, exact top-k with running thresholds, query blocks spread over `cores` threads like faiss'
    OpenMP search) timed on a bounded sample and scaled linearly in corpus rows."""
    from oracle import search_oracle
    xb = xb_sample.cpu().numpy()
    xq = xq_sample.cpu().numpy()
    search_oracle.topk_ip_threaded(xq[:64], xb[:8192], k, workers=cores)  # warm the pools
    t0 = time.perf_counter()
    Do, Io = search_oracle.topk_ip_threaded(xq, xb, k, workers=cores)
    dt = time.perf_counter() - t0
    qps_sample = xq.shape[0] / dt
    return qps_sample * (xb.shape[0] / rows_total), dt, Do, Io


def float32_leg(args, device, world, rank, lo, hi):
    """The same search over the UNROUNDED float32 corpus/queries (an '<f4' index, eval_retrieval.py:99-104):
    values fp16 cannot hold put the index in exact-float32 mode (fp16 scan + float32 re-scoring)."""
    from proqa_amd.index import ShardedIndexFlatIP
    n, nq, k = args.rows, args.queries, args.topk
    sharded = ShardedIndexFlatIP(n)
    g = torch.Generator(device=device)
    c0, c1 = lo // GEN_CHUNK, (hi + GEN_CHUNK - 1) // GEN_CHUNK
    for c in range(c0, c1):
        g.manual_seed(1000 + c)
        chunk = torch.randn((GEN_CHUNK, D), generator=g, device=device, dtype=torch.float32)
        a, b = max(lo, c * GEN_CHUNK), min(hi, (c + 1) * GEN_CHUNK)
        sharded.add_local(chunk[a - c * GEN_CHUNK:b - c * GEN_CHUNK].contiguous())
    g.manual_seed(1)
    xq = torch.randn((nq, D), generator=g, device=device, dtype=torch.float32)
    result = {}

    def step():
        result["DI"] = sharded.search(xq, k)

    steps = max(3, args.steps // 4)
    dt = timed(step, steps, 1, world, device)
    out = {"metric": "queries/sec top-%d over a float32 index (exact-float32 mode)" % k, "value": nq * steps / dt,
           "unit": "queries/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": "f16 scan + f32 rows, f64 re-scoring",
           "exact_f32": bool(sharded.local_index.exact_f32), "fallback_rounds": sharded.local_index.last_stats()["fallback_rounds"]}
    if rank == 0 and world == 1 and not args.skip_cpu:
        from oracle import search_oracle
        ns, qs = min(GEN_CHUNK, hi - lo), 64        # float64-accumulated oracle on the first chunk of the same rows
        from proqa_amd.index import IndexFlatIP
        ix = IndexFlatIP(128)
        g.manual_seed(1000)
        rows = torch.randn((GEN_CHUNK, D), generator=g, device=device, dtype=torch.float32)[:ns].contiguous()
        ix.add(rows)
        Dg, Ig = ix.search_device(xq[:qs], k)
        Do, Io = search_oracle.topk_ip_exact(xq[:qs].cpu().numpy(), rows.cpu().numpy(), k)
        out["parity"] = {"sample": f"{qs} q x {ns} float32 rows vs the float64-accumulated oracle",
                         "ids_identical": bool((Ig.cpu().numpy() == Io).all()),
                         "scores_identical": bool((Dg.cpu().numpy() == Do).all())}
    return out


def encode_leg(args, device, world, rank):
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
    B, S = args.encode_batch, args.seq_len
    sd = random_state_dict(BERT_BASE, seed=0)
    model = BertForRetriever(BERT_BASE, device=device)
    model.load_state_dict(sd)
    if os.environ.get("PROQA_TUNE_GEMM"):          # opt-in rocBLAS solution tuning (off in the default run)
        model.tune_gemms(True)
    g = torch.Generator(device=device)
    g.manual_seed(rank)
    ids = torch.randint(1000, 30522, (B, S), generator=g, device=device, dtype=torch.int64)
    ids[:, 0], ids[:, -1] = 101, 102
    mask = torch.ones((B, S), dtype=torch.bool, device=device)
    batch = {"input_ids": ids, "input_mask": mask}
    outs = []
    # same stream discipline as proqa_amd.get_embed.predict: ONE compute stream (concurrent
    # stream-K GEMMs on two streams deadlock on ragged batch sizes; see N_STREAMS there)
    n_streams = int(os.environ.get("PROQA_ENCODE_STREAMS", "1"))
    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
    counter = [0]

    def step():
        s = streams[counter[0] % n_streams]
        counter[0] += 1
        with torch.cuda.stream(s):
            outs.append(model.get_embed(batch, False, check_mask=False)["embed"])
        if len(outs) > 2 * n_streams:
            outs.pop(0)

    dt = timed(step, args.encode_steps, 2, world, device)
    pps = world * B * args.encode_steps / dt
    # flops actually executed: the last layer runs its attention output / dense blocks / LayerNorms on the
    # [CLS] row only, i.e. (14.55 - 3.54) MFLOP x (S-1) tokens fewer than the reference's 22.35 GFLOP at S=128
    per_tok_layer = 2 * 768 * 2304 + 2 * 768 * 768 + 4 * 768 * 3072 + 4 * S * 768
    executed = (12 * S * per_tok_layer - (S - 1) * (per_tok_layer - 2 * 768 * 2304)) / 1e9
    tf = pps / world * executed / 1e3   # per-GPU TFLOP/s
    res = {
        "metric": "passages/sec encoded", "value": pps, "unit": "passages/s", "ms_per_step": dt / args.encode_steps * 1e3,
        "steps": args.encode_steps, "scaling": "weak", "dtype": "f16",
        # bert-base-uncased shape (12 x 768 x 12 x 3072), pre-tokenised ids, fp16 weights / activations, fp32 accumulate,
        # random N(0, 0.02) weights (BASELINE configs[1])
        "config": {"workload": f"bert-base shape, {B} x {S} pre-tokenised ids, fp16, random weights (configs[1])", "batch": B,
                   "seq_len": S},
        "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf / PEAK_MFMA_F16_TFLOPS, "traffic": encode_traffic(B, S),
                     "gflop_per_passage_executed": executed, "gflop_per_passage_reference": ENCODE_GFLOP_PER_PASSAGE,
                     "note": "flops executed per step / step time (the last layer is evaluated on the [CLS] rows "
                             "only); the dense-layer GEMMs are rocBLAS calls made by libproqa_hip.so"},
    }
    # which dense path ran: the hipBLASLt kernel pinned by name for the large layers, or rocblas_gemm_ex (lt_gemm.cpp falls
    # back silently when the pinned kernel is absent from the loaded library)
    try:
        name = model.gemm_kernels().get(False, "")   # the passage tower
    except Exception as e:   # noqa: BLE001
        name = f"unknown ({type(e).__name__})"
    res["gemm_kernel_full"] = name or "rocblas_gemm_ex"
    res["gemm_kernel"] = (name if len(name) <= 72 else name[:52] + ".." + name[-18:]) or "rocblas_gemm_ex"
    if args.skip_varlen:     # (the encoder's PMC traffic passes want full-size steps only)
        return res
    # the reference's own operating points (retrieval/config.py:25 --max_seq_length 512; get_para_embed.sh:5 batch 300),
    # each with the roofline on executed flops (attention is 4 S 768 flop per token and layer: 16 x as much at S = 512) and
    # the parity of four rows against the NumPy oracle
    for key, B2, S2 in (("seq512", 64, 512), ("batch300", 300, 128)):
        ids2 = torch.randint(1000, 30522, (B2, S2), generator=g, device=device, dtype=torch.int64)
        ids2[:, 0], ids2[:, -1] = 101, 102
        batch2 = {"input_ids": ids2, "input_mask": torch.ones((B2, S2), dtype=torch.bool, device=device)}
        keep = []

        def step2():
            keep.append(model.get_embed(batch2, False, check_mask=False)["embed"])
            if len(keep) > 2:
                keep.pop(0)

        # (the better of two runs of K steps: one run of 8 steps of ~7.5 ms is at the mercy of a single host hiccup --
        # a collection of this round read 29 k passages/s here between four that read 37-40 k)
        dt2 = min(timed(step2, args.encode_steps, 2, world, device), timed(step2, args.encode_steps, 0, world, device))
        ptl = 2 * 768 * 2304 + 2 * 768 * 768 + 4 * 768 * 3072 + 4 * S2 * 768
        ex2 = (12 * S2 * ptl - (S2 - 1) * (ptl - 2 * 768 * 2304)) / 1e9
        tf2 = B2 * args.encode_steps / dt2 * ex2 / 1e3
        res[key] = {"value": world * B2 * args.encode_steps / dt2, "unit": "passages/s", "batch": B2, "seq_len": S2,
                    "ms_per_step": dt2 / args.encode_steps * 1e3, "tokens_per_s": world * B2 * S2 * args.encode_steps / dt2,
                    "roofline": {"bound": "mfma", "achieved": tf2, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf2 / PEAK_MFMA_F16_TFLOPS, "gflop_per_passage_executed": ex2}}
        if rank == 0 and world == 1 and not args.skip_cpu:
            from oracle import bert_oracle
            sd_np_ = {k_: v_.numpy() for k_, v_ in sd.items()}
            ref2 = bert_oracle.get_embed(sd_np_, ids2[:4].cpu().numpy(), np.ones((4, S2), bool), False, 12, 12)
            res[key]["parity_max_abs_err_vs_oracle"] = float(np.abs(keep[-1][:4].float().cpu().numpy() - ref2).max())
        del ids2, batch2, keep
    # variable-length variant: lengths ~ U[32, S], right-padded as em_collate does;
    # the lengths are known on the host (predict() takes them from the collated batch), padding is skipped
    lens_host = torch.randint(32, S + 1, (B,), generator=torch.Generator().manual_seed(rank)).tolist()
    lens_dev = torch.tensor(lens_host, device=device)
    vmask = torch.arange(S, device=device)[None, :] < lens_dev[:, None]
    vids = torch.where(vmask, ids, torch.zeros_like(ids))
    vids[torch.arange(B, device=device), lens_dev - 1] = 102
    vbatch = {"input_ids": vids, "input_mask": vmask}

    def vstep():
        s = streams[counter[0] % n_streams]
        counter[0] += 1
        with torch.cuda.stream(s):
            outs.append(model.get_embed(vbatch, False, check_mask=False, seq_lens_host=lens_host)["embed"])
        if len(outs) > 2 * n_streams:
            outs.pop(0)

    # (the better of two runs, as above: one collection of round 6 read 48 k passages/s here between runs that read 64-66 k)
    vdt = min(timed(vstep, args.encode_steps, 2, world, device), timed(vstep, args.encode_steps, 0, world, device))
    res["varlen"] = {"value": world * B * args.encode_steps / vdt, "unit": "passages/s",
                     "ms_per_step": vdt / args.encode_steps * 1e3, "mean_len": float(np.mean(lens_host)),
                     "tokens_per_s": world * float(np.sum(lens_host)) * args.encode_steps / vdt,
                     "workload": f"same model, batch {B}, lengths ~U[32,{S}] right-padded to {S}; valid tokens packed"}
    step()   # leave a full-length batch in outs[-1] for the parity check below

    if rank == 0 and world == 1 and not args.skip_cpu:   # CPU baselines: single-GPU runs only (contract)
        from oracle import bert_oracle, bert_torch_cpu
        # parity: the pinned NumPy oracle on 4 passages of the timed batch
        nb = 4
        sd_np = {k: v.numpy() for k, v in sd.items()}
        ids_np, mask_np = ids[:nb].cpu().numpy(), mask[:nb].cpu().numpy()
        ref = bert_oracle.get_embed(sd_np, ids_np, mask_np, False, 12, 12)
        got = outs[-1][:nb].float().cpu().numpy()
        res["parity_max_abs_err_vs_oracle"] = float(np.abs(got - ref).max())
        # CPU baseline (SURVEY 8d iii): the torch-CPU fp32 restatement (threaded like the reference's own CPU
        # execution) on a bounded sample: one warm-up pass, then 3 passes of 128 passages, best taken
        cb = 128
        ids_cpu, mask_cpu = ids[:cb].cpu(), mask[:cb].cpu()
        torch.set_num_threads(host_cores())
        bert_torch_cpu.get_embed(sd, ids_cpu[:32], mask_cpu[:32], False, 12, 12)
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            bert_torch_cpu.get_embed(sd, ids_cpu, mask_cpu, False, 12, 12)
            best = min(best, time.perf_counter() - t0)
        res["cpu_baseline"] = {"value": cb / best, "unit": "passages/s", "cores": torch.get_num_threads(),
                               "kind": "port",
                               "sample": f"{cb} passages x {S} tokens, oracle/bert_torch_cpu.py fp32, best of 3"}
    if args.corpus_passages > 0:
        res["corpus_1m"] = corpus_leg(args, device, world, rank, model, sd)
    if args.cli_passages > 0 and world == 1:
        del model
        torch.cuda.empty_cache()
        res["cli_text"] = cli_text_leg(args, device, sd)
        # the same command line on a corpus in which 30 % of the passages hold accented words, Greek / Cyrillic words,
        # Unicode punctuation and CJK runs (Wikipedia-like): the native tokenizer has to keep the GPU fed there too
        res["cli_text_non_ascii"] = cli_text_leg(args, device, sd, non_ascii=0.3, n_pass=max(20000, args.cli_passages // 2))
    return res


def cli_text_leg(args, device, sd, non_ascii=0.0, n_pass=None):
    """Encode throughput through the product's own command line on TEXT ("report separately with real
    text"): 200 000 synthetic ~100-word passages in a JSON-lines file -> proqa_amd.get_embed.main (JSONL read, WordPiece
    tokenisation on --eval-workers threads, collate, upload, encode, D2H, np.save), the reference's
    retrieval/get_embed.py:29-139 end to end.  Reported: passages/s of the whole call and of its encode loop, the rate the
    loader alone reaches (tokenise-only), and the fraction of the loop the GPU sat idle (HIP events around get_embed)."""
    import shutil
    import tempfile
    from proqa_amd import get_embed as ge
    from proqa_amd.datasets import JsonlTexts, TextBatchLoader, TokenizeCollate
    from proqa_amd.retriever import BERT_BASE
    n_pass = n_pass or args.cli_passages
    d = tempfile.mkdtemp(prefix="proqa_cli_")
    try:
        rng = np.random.default_rng(7)
        letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))

        def word():
            return "".join(rng.choice(letters, size=int(rng.integers(3, 10))))
        # a bert-base-shaped vocabulary: the special tokens where bert-base-uncased has them, single characters and
        # their continuations, then whole words and word continuations up to 30 522 entries
        vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"[unused{i}]" for i in range(99, 994)]
        chars = [chr(c) for c in range(33, 127) if not chr(c).isupper()]
        vocab += chars + ["##" + c for c in chars if c.isalnum()]
        words = set()
        while len(words) < 16000:
            words.add(word())
        words = sorted(words)
        pieces = set()
        while len(vocab) + len(words) + len(pieces) < BERT_BASE["vocab_size"]:
            pieces.add("##" + word()[:int(rng.integers(2, 5))])
        # non-ASCII workload: 600 Greek / Cyrillic words and 400 CJK ideographs take the place of as many word pieces
        foreign = []
        if non_ascii > 0:
            greek = [chr(c) for c in range(0x3B1, 0x3CA) if c != 0x3C2]
            cyr = [chr(c) for c in range(0x430, 0x450)]
            fw = set()
            while len(fw) < 600:
                alpha = greek if len(fw) % 2 else cyr
                fw.add("".join(rng.choice(alpha, size=int(rng.integers(3, 9)))))
            foreign = sorted(fw) + [chr(c) for c in range(0x4E00, 0x4E00 + 400)]
        vocab += words + sorted(pieces)[:BERT_BASE["vocab_size"] - len(vocab) - len(words) - len(foreign)] + foreign
        assert len(vocab) == BERT_BASE["vocab_size"] and len(set(vocab)) == len(vocab)
        model_dir = os.path.join(d, "bert-base-synthetic")
        os.makedirs(model_dir)
        with open(os.path.join(model_dir, "vocab.txt"), "w") as f:
            f.write("\n".join(vocab) + "\n")
        with open(os.path.join(model_dir, "config.json"), "w") as f:
            json.dump(dict(BERT_BASE, model_type="bert"), f)
        torch.save(sd, os.path.join(d, "ckpt.pt"))
        warr = np.array(words)
        oov = np.array([word() + word() for _ in range(20000)])          # out-of-vocabulary: split into several pieces
        accents = str.maketrans({"a": "\u00e1", "e": "\u00e9", "o": "\u00f6", "u": "\u00fc", "n": "\u00f1", "c": "\u00e7"})
        farr = np.array(foreign[:600]) if foreign else None
        with open(os.path.join(d, "paras.txt"), "w", encoding="utf-8") as f:
            for i0 in range(0, n_pass, 10000):
                m = min(10000, n_pass - i0)
                known = warr[rng.integers(0, len(warr), (m, 88))]
                unknown = oov[rng.integers(0, len(oov), (m, 12))]
                mixed = rng.random(m) < non_ascii
                for r in range(m):
                    ws = list(known[r]) + list(unknown[r])
                    if mixed[r]:
                        # a passage of the non-ASCII share: a third of its words accented (they fold to the plain words), a
                        # tenth Greek / Cyrillic, an en dash, curly quotes and a run of CJK ideographs
                        for j in range(0, len(ws), 3):
                            ws[j] = ws[j].translate(accents).capitalize() if j % 2 else ws[j].translate(accents)
                        for j in range(5, len(ws), 10):
                            ws[j] = str(farr[int(rng.integers(0, len(farr)))])
                        ws[7] = "\u2013"
                        ws[20] = "\u201c" + ws[20] + "\u201d"
                        ws[40] = "".join(foreign[600 + int(v)] for v in rng.integers(0, 400, 6))
                    text = " ".join(ws)
                    f.write(json.dumps({"id": f"p{i0 + r}", "text": text[:1].upper() + text[1:] + "."}, ensure_ascii=False) + "\n")
        cores = host_cores()
        argv = ["--do_predict", "--predict_batch_size", str(args.encode_batch), "--bert_model_name", model_dir, "--fp16",
                "--predict_file", os.path.join(d, "paras.txt"), "--init_checkpoint", os.path.join(d, "ckpt.pt"),
                "--embed_save_path", os.path.join(d, "para_embed.npy"), "--eval-workers", str(cores),
                "--max_seq_length", str(args.seq_len)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out_path = ge.main(argv)
        dt = time.perf_counter() - t0
        st = dict(ge.LAST_RUN_STATS)
        emb = np.load(out_path, mmap_mode="r")
        assert emb.shape == (n_pass, 128) and emb.dtype == np.float16
        # the loader alone: the same dataset / collate / worker count, nothing consumed on the GPU
        from transformers import BertTokenizer
        tok = BertTokenizer.from_pretrained(model_dir)
        ds = JsonlTexts(os.path.join(d, "paras.txt"), 30, args.seq_len, False)
        n_tok = min(n_pass, 60000)
        native_share = None
        if non_ascii > 0:      # how many of the mixed passages the library's own WordPiece tokenises itself
            probe = TokenizeCollate(tok, ds.max_length, parallel=True, native_threads=max(1, cores - 2))
            import ctypes
            from proqa_amd import _lib
            lib_, h_ = probe._native_handle()
            sample = [ds[i] for i in range(min(n_pass, 20000))]
            raw = [t.encode("utf-8") for t in sample]
            ids_ = np.empty((len(raw), ds.max_length), dtype=np.int64)
            lens_ = np.empty(len(raw), dtype=np.int32)
            _lib.check(lib_.proqa_wordpiece_encode_batch(h_, (ctypes.c_char_p * len(raw))(*raw),
                                                         np.fromiter(map(len, raw), dtype=np.int64, count=len(raw)).ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                         len(raw), ds.max_length, ids_.ctypes.data, lens_.ctypes.data, max(1, cores - 2)))
            native_share = float((lens_ >= 0).mean())
        loader = TextBatchLoader(ds, args.encode_batch,
                                 TokenizeCollate(tok, ds.max_length, parallel=True, native_threads=max(1, cores - 2)), prefetch=8,
                                 lo=0, hi=n_tok)
        t1 = time.perf_counter()
        tokens = 0
        for b in loader:
            tokens += sum(b["seq_lens"])
        t_tok = time.perf_counter() - t1
        return {"metric": "passages/sec encoded from text through the get_embed.py command line", "value": n_pass / dt,
                "unit": "passages/s", "passages": n_pass, "seconds": dt, "loader_workers": st.get("loader_workers"),
                "host_cores": cores, "mean_tokens_per_passage": tokens / n_tok,
                "encode_loop": {"passages_per_s": st["passages"] / st["loop_seconds"], "seconds": st["loop_seconds"],
                                "gpu_busy_seconds": st["gpu_busy_seconds"], "loader_wait_seconds": st["loader_wait_seconds"],
                                "feed_seconds": st["feed_seconds"], "upload_seconds": st["upload_seconds"],
                                "gpu_idle_fraction": max(0.0, 1.0 - st["gpu_busy_seconds"] / st["loop_seconds"])},
                "non_ascii_passage_fraction": non_ascii, "native_wordpiece": native_share,
                "tokenise_only": {"passages_per_s": n_tok / t_tok, "passages": n_tok,
                                  "note": "the same loader (TextBatchLoader: one producer thread; libproqa_hip.so's table-driven "
                                          "WordPiece on cores - 2 threads) with nothing consumed on the GPU"},
                "workload": f"{n_pass} synthetic passages of 100 words (88 in-vocabulary, 12 split into word pieces) in a "
                            f"JSON-lines file, bert-base-shaped 30 522-entry vocabulary, max_seq_length {args.seq_len}, batch "
                            f"{args.encode_batch}; whole call = JSONL read + tokeniser + model/checkpoint load + encode + np.save"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def corpus_leg(args, device, world, rank, model, sd):
    """BASELINE.json configs[1] as written: N synthetic 128-token passages, batch 512, fp16 -> .npy index, through
    the product's own loop -- proqa_amd.get_embed.predict (get_embed.py:142-172: batches to the GPU, get_embed,
    embeddings kept on the device, torch.cat) and npy.save (get_embed.py:138-139: D2H + file write).  The timed
    region covers all of it; tokenisation is excluded (pre-tokenised ids, as in the encode leg).  Each rank
    encodes N/world passages and writes its own file (the reference's sharded get_embed runs do the same)."""
    import tempfile
    from proqa_amd import npy
    from proqa_amd.get_embed import predict
    B, S = args.encode_batch, args.seq_len
    n_local = args.corpus_passages // world
    g = torch.Generator().manual_seed(1234 + rank)
    pool = []                                       # 8 distinct full batches, cycled: 1M x 128 ids would be 1 GB of host memory
    for _ in range(8):
        ids = torch.randint(1000, 30522, (B, S), generator=g, dtype=torch.int64)
        ids[:, 0], ids[:, -1] = 101, 102
        pool.append({"input_ids": ids.pin_memory(), "input_mask": torch.ones((B, S), dtype=torch.bool).pin_memory()})

    def loader():
        done = 0
        i = 0
        while done < n_local:
            m = min(B, n_local - done)              # the ragged last batch (1M = 1953 x 512 + 64)
            b = pool[i % len(pool)]
            yield b if m == B else {k: v[:m] for k, v in b.items()}
            done += m
            i += 1

    model.half()
    out_dir = tempfile.mkdtemp(prefix="proqa_bench_")
    path = os.path.join(out_dir, f"para_embed_{rank}.npy")
    predict(args, model, ({k: v[:64] for k, v in pool[0].items()} for _ in range(2)), device, is_query_embed=False)  # warm-up
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    embeds = predict(args, model, loader(), device, is_query_embed=False)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    local = embeds.cpu().numpy()
    t_d2h = time.perf_counter() - t0
    npy.save(path, local)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = {"metric": "passages/sec encoded, corpus run", "value": n_local * world / dt, "unit": "passages/s",
           "passages": n_local * world, "batch": B, "seq_len": S, "seconds": dt,
           "seconds_encode_loop": t_gpu, "seconds_d2h": t_d2h - t_gpu, "seconds_npy_write": dt - t_d2h if world == 1 else None,
           "bytes_written_per_rank": os.path.getsize(path), "dtype": str(local.dtype),
           "workload": f"{n_local * world} pre-tokenised {S}-token passages, batch {B}, get_embed.predict -> torch.cat -> "
                       f"D2H -> npy.save, all inside the timed region (BASELINE.json configs[1])"}
    if rank == 0 and not args.skip_cpu:
        from oracle import bert_oracle
        rows = [0, 1, n_local // 2, n_local - 1]     # first batch, a middle batch, the ragged last batch
        sd_np = {k: v.numpy() for k, v in sd.items()}
        back = npy.load(path)
        err = 0.0
        for r in rows:
            b = pool[(r // B) % len(pool)]
            ids_np = b["input_ids"][r % B:r % B + 1].numpy()
            ref = bert_oracle.get_embed(sd_np, ids_np, np.ones_like(ids_np, dtype=bool), False, 12, 12)
            err = max(err, float(np.abs(back[r].astype(np.float32) - ref[0]).max()))
        out["parity_max_abs_err_vs_oracle"] = err
        out["parity_rows"] = rows
    os.remove(path)
    os.rmdir(out_dir)
    return out


def cli_eval_leg(args, device, xb, xq, I_top, k):
    """The drop-in command line of the search half, end to end, at the headline size (the reference's
    retrieval/eval_retrieval.py:88-123): the resident corpus is written ONCE to a '<f2' para_embed.npy (outside every timed
    region), with the 2032 query embeddings, a QA file, a row -> doc-id sidecar and a sqlite DB of passages in which the
    answer of every second question is planted in that question's best passage (so the printed recall must be exactly
    0.5: a check of the id map, not of the search).  Timed: `python eval_retrieval.py ...` as a child process (imports,
    scorer-pool fork, file -> HBM load, search, id map, scoring), its own per-stage clock, the loader alone with a cold and
    a warm page cache next to a dd-style single-thread read of the same file, and the host-pointer IndexFlatIP.search."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    from proqa_amd import npy
    from proqa_amd.index import IndexFlatIP
    n, nq = xb.shape[0], xq.shape[0]
    base = os.environ.get("PROQA_BENCH_TMP") or tempfile.gettempdir()
    need = n * 256 + (1 << 30)
    if shutil.disk_usage(base).free < need:
        return {"skipped": f"{base} has less than {need >> 20} MiB free for the index file"}
    d = tempfile.mkdtemp(prefix="proqa_cli_eval_", dir=base)
    try:
        t_prep = time.perf_counter()
        index_path = os.path.join(d, "para_embed.npy")
        with open(index_path, "wb") as f:                   # what np.save writes, streamed from the device in pieces
            np.lib.format.write_array_header_1_0(f, {"descr": "<f2", "fortran_order": False, "shape": (n, 128)})
            for r0 in range(0, n, 2_000_000):
                f.write(xb[r0:r0 + 2_000_000].cpu().numpy().data)
        assert npy.stat(index_path)["rows"] == n
        np.save(os.path.join(d, "q_embed.npy"), xq.cpu().numpy())
        # passages: 100 words of a 5000-word vocabulary; row r of the index is passage r % n_docs
        rng = np.random.default_rng(3)
        n_docs = 200_000
        letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
        vocab = np.array(["".join(rng.choice(letters, size=int(rng.integers(3, 10)))) for _ in range(5000)])
        words = vocab[rng.integers(0, len(vocab), (n_docs, 100))]
        docs = [" ".join(w) for w in words]
        top1 = I_top[:, 0].cpu().numpy()
        with open(os.path.join(d, "qa.txt"), "w") as f:
            for q in range(nq):
                ans = f"answq{q} tokq{q}"
                if q % 2 == 0:
                    docs[int(top1[q]) % n_docs] += f" {ans.capitalize()} ."
                f.write(json.dumps({"question": f"question number {q}", "answer": [ans, "never matches xyz"]}) + "\n")
        conn = sqlite3.connect(os.path.join(d, "paras.db"))
        conn.execute("CREATE TABLE documents (id PRIMARY KEY, text)")
        conn.executemany("INSERT INTO documents VALUES (?,?)", ((f"d{i:07d}", t) for i, t in enumerate(docs)))
        conn.commit()
        conn.close()
        # sidecar of gen_index_id_map.write_sidecar, generated in bulk: line r = '"d%07d"\n' % (r % n_docs)
        ids = (np.arange(n, dtype=np.int64) % n_docs)
        blob = np.empty((n, 11), dtype=np.uint8)
        blob[:, 0], blob[:, 1], blob[:, 9], blob[:, 10] = ord('"'), ord("d"), ord('"'), ord("\n")
        for j in range(7):
            blob[:, 8 - j] = ord("0") + (ids // 10 ** j) % 10
        blob.tofile(os.path.join(d, "idx_id.ids"))
        np.save(os.path.join(d, "idx_id.off.npy"), np.arange(n + 1, dtype=np.int64) * 11)
        # the text sidecar of gen_index_id_map.write_text_sidecar, in bulk: every distinct passage once, row r -> span of
        # passage r % n_docs (rows with the same text share it)
        enc = [t.encode("utf-8") for t in docs]
        ends = np.cumsum([len(b_) for b_ in enc], dtype=np.int64)
        with open(os.path.join(d, "idx_id.txt"), "wb") as f:
            f.write(b"".join(enc))
        doc_spans = np.stack([ends - np.array([len(b_) for b_ in enc], dtype=np.int64), ends], axis=1)
        np.save(os.path.join(d, "idx_id.txtoff.npy"), doc_spans[ids])
        del blob, ids, words, docs, enc, doc_spans
        t_prep = time.perf_counter() - t_prep

        def drop_cache():
            fd = os.open(index_path, os.O_RDONLY)
            try:
                os.fsync(fd)
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            finally:
                os.close(fd)

        def dd_read():
            buf = bytearray(32 << 20)
            t0 = time.perf_counter()
            with open(index_path, "rb", buffering=0) as f:
                while f.readinto(buf):
                    pass
            return n * 256 / (time.perf_counter() - t0) / 1e9

        def load_only():
            ix = IndexFlatIP(128, capacity=n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix.add_npy(index_path)
            dt = time.perf_counter() - t0
            return ix, n * 256 / dt / 1e9

        t0 = time.perf_counter()
        drop_cache()
        t_drop = time.perf_counter() - t0
        dd_cold = dd_read()
        dd_warm = dd_read()
        drop_cache()
        ix, load_cold = load_only()
        ix.close()
        ix, load_warm = load_only()
        # the host-pointer search of the faiss call shape (numpy in, numpy out) on the index just loaded
        xq_np = xq.cpu().numpy()
        ix.search(xq_np, k)
        t0 = time.perf_counter()
        for _ in range(5):
            Dh, Ih = ix.search(xq_np, k)
        host_ms = (time.perf_counter() - t0) / 5 * 1e3
        ids_equal = bool((Ih == I_top.cpu().numpy()).all())
        ix.close()
        del ix
        torch.cuda.empty_cache()

        stats_path = os.path.join(d, "stats.json")
        cmd = [sys.executable, os.path.join(ROOT, "eval_retrieval.py"), os.path.join(d, "qa.txt"), index_path,
               os.path.join(d, "q_embed.npy"), os.path.join(d, "paras.db"), "--topk", str(k), "--num-workers", "10",
               "--idx-id-map", os.path.join(d, "idx_id.ids")]
        env = {kk: v for kk, v in os.environ.items() if kk not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env["PROQA_STATS_JSON"] = stats_path
        # first the reference's route for the texts (row -> doc id -> one sqlite query per hit), then the same command line
        # with the index's text sidecar in use (the default when <stem>.txt sits next to the id map): same printed lines
        t0 = time.perf_counter()
        proc_db = subprocess.run(cmd + ["--no-text-sidecar"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall_db = time.perf_counter() - t0
        if proc_db.returncode != 0:
            return {"error": proc_db.stderr[-800:]}
        st_db = json.load(open(stats_path))
        t0 = time.perf_counter()
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if proc.returncode != 0:
            return {"error": proc.stderr[-800:]}
        st = json.load(open(stats_path))
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("Top ")]
        same_lines = lines == [ln for ln in proc_db.stdout.splitlines() if ln.startswith("Top ")]
        recall = {ln.split()[1]: float(ln.split(": ")[1].split()[0]) for ln in lines}
        # the reference's own id map (json.load of {"<row>": id}, eval_retrieval.py:73-74) on a 1M-entry sample
        sample = 1_000_000
        jpath = os.path.join(d, "idx_id_sample.json")
        with open(jpath, "w") as f:
            json.dump({str(i): f"d{i % n_docs:07d}" for i in range(sample)}, f)
        t0 = time.perf_counter()
        with open(jpath) as f:
            m = json.load(f)
        t_json = time.perf_counter() - t0
        del m
        return {
            "metric": "seconds for the eval_retrieval.py command line, end to end", "value": wall, "unit": "s",
            "higher_is_better": False, "rows": n, "queries": nq, "topk": k, "index_file_bytes": os.path.getsize(index_path),
            # stages: imports + scorer-pool fork / HIP start + index allocation / index file -> HBM / search incl. query upload
            # and result download / row -> doc id / scoring pool
            "stages_seconds": {"imports_fork": st.get("startup_seconds"), "hip_start": st.get("gpu_init_seconds"),
                               "load_index": st.get("load_seconds"),
                               "search": st.get("search_seconds"),
                               "idx2id": st.get("idx2id_seconds"), "scoring": st.get("scoring_seconds"),
                               "main_total": st.get("total_seconds"), "process_wall": wall},
            "load_GBs_in_cli": st.get("load_gbs"), "text_sidecar": st.get("text_sidecar"),
            "sqlite_route": {"value": wall_db, "idx2id": st_db.get("idx2id_seconds"), "scoring": st_db.get("scoring_seconds"),
                             "same_printed_lines": same_lines},
            "loader": {"cold_GBs": load_cold, "warm_GBs": load_warm, "dd_style_read_cold_GBs": dd_cold, "dd_style_read_warm_GBs": dd_warm,
                       "cache_drop_seconds": t_drop,
                       "note": "proqa_index_add_npy (4 reader threads -> pinned ring -> HBM) on the whole file (cold = also the first call of the process); dd-style = one thread "
                               "readinto() of 32 MiB blocks, nothing uploaded; cold = after fsync + posix_fadvise(DONTNEED), "
                               "best effort (a tmpfs cannot be dropped)"},
            "host_api_search": {"ms_per_search": host_ms, "queries_per_s": nq / host_ms * 1e3, "ids_equal_device_search": ids_equal,
                                "note": "IndexFlatIP.search(numpy xq) -> numpy D, I: the faiss call shape incl. query upload and result download"},
            "idx2id_json_route": {"entries_sample": sample, "json_load_seconds": t_json,
                                  "note": f"the reference's idx_id.json costs this per million rows ({n / 1e6:.0f}x at this index); "
                                          "the CLI run above used the .ids sidecar"},
            "recall_printed": recall, "recall_expected": 0.5, "scorer_processes": st.get("scorer_processes"),
            "prepare_seconds_untimed": t_prep,
            "workload": f"{n} x 128 fp16 para_embed.npy, {nq} questions, top-{k}, {n_docs} passages of 100 words in sqlite, "
                        f"row r -> passage r % {n_docs}; answers planted for even questions in their top-1 passage"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line and the child's exit code.  Nothing in this process has
    touched the GPU at this point (no HIP call before the child exists, and no exec of a process that initialised it)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank child exited with code {proc.returncode}"
                         f"{'' if lines else ' and printed no result line'}\n")
        raise SystemExit(proc.returncode or 1)
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    raise SystemExit(0)


def kmeans_leg(args, device, xb):
    """SURVEY section 8(f) row 1 on the driver's clock: retrieval/group_paras.py:20-53 at its default shape -- 10 000
    centroids over 10M x 128 passage embeddings (1000 points per centroid: faiss' sub-sampling bound, so every point
    trains) -- as Lloyd iterations of proqa_amd.group_paras.KMeans: 3 warm + 5 timed.  Roofline on kmeans_assign
    (2 n k 144 flop per iteration executed by the nominating pass -- the hi fp16 halves of the fp32 centroids plus the norm
    step; the few per cent of undecided points go through hi + lo + norm, 272, once more); the final assignment is checked
    against the NumPy restatement of faiss' search on a sample of the points."""
    from proqa_amd.group_paras import KMeans
    n = min(10_000_000, xb.shape[0])
    k = 10_000 if n >= 1_000_000 else max(8, n // 1000)
    warm, timed_it = 3, 5
    x = xb[:n]
    km = KMeans(128, k, niter=warm + timed_it, max_points_per_centroid=n // k + 1, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    km.train(x)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    it_s = float(np.mean(km.iter_seconds[warm:]))
    assign_s = float(np.mean(km.assign_ms[warm:])) / 1e3
    flops = 2.0 * n * k * 144
    tf = flops / assign_s / 1e12
    out = {"metric": "k-means Lloyd iterations/sec (group_paras.py shape)", "value": 1.0 / it_s, "unit": "iterations/s",
           "ms_per_iteration": it_s * 1e3, "points": n, "centroids": k, "iterations_warm": warm, "iterations_timed": timed_it,
           "seconds_whole_train_call": total, "dtype": "f16 points x f16 (hi) centroids, f32 accumulate; hi + lo for the points a hi-only lead cannot decide",
           "objective": km.obj[-1],
           "roofline": {"bound": "mfma", "kernel": "kmeans_assign", "achieved": tf, "peak": PEAK_MFMA_F16_TFLOPS,
                        "unit": "TFLOP/s", "frac": tf / PEAK_MFMA_F16_TFLOPS, "assign_ms": assign_s * 1e3,
                        "achieved_algorithmic": 2.0 * n * k * 128 / assign_s / 1e12,
                        "frac_algorithmic": 2.0 * n * k * 128 / assign_s / 1e12 / PEAK_MFMA_F16_TFLOPS,
                        "note": "frac counts the flops EXECUTED by the nominating pass, 2 n k 144 per iteration (the hi fp16 halves of "
                                "the fp32 centroids + the norm step; the undecided few per cent of the points, which run hi + lo + norm "
                                "once more, are not counted); frac_algorithmic the 2 n k 128 of the reference's fp32 search; both over "
                                "the HIP-event time of the assign launches (mean of the timed iterations); rounds 1-3 executed 2 n k 272"}}
    if not args.skip_cpu:
        from oracle import kmeans_oracle
        ns = min(20_000, n)
        sel = torch.linspace(0, n - 1, ns, device=device).long()
        xs = x[sel].contiguous()
        D, I = km.assign(xs)
        cen = km.centroids.cpu().numpy()
        xs_np = xs.cpu().numpy()
        t1 = time.perf_counter()
        Do, Io = [], []
        for r0 in range(0, ns, 2000):                       # blocks: the float64 distance matrix is k columns wide
            d_, i_ = kmeans_oracle.assign(xs_np[r0:r0 + 2000], cen, True)
            Do.append(d_)
            Io.append(i_)
        cpu_s = time.perf_counter() - t1
        Do, Io = np.concatenate(Do), np.concatenate(Io)
        Dg, Ig = D.cpu().numpy(), I.cpu().numpy()
        out["parity"] = {"sample": f"{ns} points x {k} centroids vs oracle/kmeans_oracle.py",
                         "assignment_agreement": float((Ig == Io).mean()),
                         "objective_rel_diff": float(abs(Dg.astype(np.float64).sum() - Do.astype(np.float64).sum()) /
                                                     max(Do.astype(np.float64).sum(), 1e-30))}
        out["cpu_baseline"] = {"value": ns / cpu_s / n, "unit": "iterations/s (assign step only)", "cores": host_cores(),
                               "kind": "port", "sample": f"assign {ns} points to {k} centroids (NumPy, {cpu_s:.1f} s), scaled to {n}"}
    return out


def online_leg(args, device, index, n_rows):
    """SURVEY section 8(f) row 4 on the driver's clock: the retrieval step of qa/online_sampler.py:104-121 for ONE question
    -- encode with the query tower (bert-base shape, random weights), exact search over the resident index, row gather
    (para_embed[I]) -- through proqa_amd.online_retriever.OnlineRetriever; k = 80 and k = 5000 (the sampler's training k).
    Median of 50 questions each."""
    from proqa_amd.online_retriever import OnlineRetriever
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
    model = BertForRetriever(BERT_BASE, device=device)
    model.load_state_dict(random_state_dict(BERT_BASE, seed=0))
    from proqa_amd.online_retriever import GraphedQuestionEncoder
    r = OnlineRetriever(np.float16, None, device=device, index=index)
    graphed = GraphedQuestionEncoder(model)
    host_rng = np.random.default_rng(5)
    g = torch.Generator(device=device).manual_seed(5)
    mask = torch.ones((1, 16), dtype=torch.bool, device=device)
    out = {"metric": "ms per question: encode + exact top-k + row gather (online sampler step)", "unit": "ms",
           "rows": n_rows, "question_tokens": 16, "questions": 50}
    # the captured HIP graph of the forward (GraphedQuestionEncoder: one launch call instead of ~90) beside the plain call
    # the leg times: the same bits, and -- measured -- the same time: the forward is bound by the GPU's dispatch of ~90
    # dependent microsecond kernels, not by the host's launch calls (ABLATIONS R5.9)
    replay = []
    for i in range(55):
        tok = host_rng.integers(1000, 30522, 16).tolist()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        q = graphed(tok)
        torch.cuda.synchronize()
        replay.append(time.perf_counter() - t0)
    out["encode_ms_graph_replay"] = float(np.median(replay[5:]) * 1e3)
    t = torch.tensor([tok], dtype=torch.int64, device=device)
    out["graph_equals_plain"] = bool(torch.equal(q, model.get_embed({"input_ids": t, "input_mask": mask}, True)["embed"]))
    for k in (80, 5000):
        enc, ret = [], []
        for i in range(55):
            tok = torch.randint(1000, 30522, (1, 16), generator=g, device=device)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            q = model.get_embed({"input_ids": tok, "input_mask": mask}, True, check_mask=False, seq_lens_host=[16])["embed"]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            idx, _, rows = r.retrieve(q, k)
            t2 = time.perf_counter()
            if i >= 5:
                enc.append(t1 - t0)
                ret.append(t2 - t1)
        assert idx.shape == (min(k, n_rows),) and rows.shape == (min(k, n_rows), 128)
        out[f"k{k}"] = {"ms_per_question": float(np.median(np.array(enc) + np.array(ret)) * 1e3),
                        "encode_ms": float(np.median(enc) * 1e3), "retrieve_ms": float(np.median(ret) * 1e3)}
    out["value"] = out["k5000"]["ms_per_question"]
    out["higher_is_better"] = False
    return out


def query_shards_leg(args, n, nq, k, world, rank, device, xq, ids_sha, scores_sha, digest):
    """rows replicated, queries sharded (proqa_amd.index.QueryShardedIndexFlatIP): every rank searches its nq / world queries
    over all n rows, one all-gather of the result rows (timed with it).  Collective-safe: the ranks agree that every one of
    them built its replica before the first collective of the timed loop."""
    from proqa_amd.index import QueryShardedIndexFlatIP
    ok, err, ix, rows = 1, None, None, None
    try:
        rows = gen_rows(0, n, device)
        ix = QueryShardedIndexFlatIP()
        ix.adopt(rows)
        ix.prepare()
    except Exception as e:   # (e.g. no room for the replica beside the row shard on a shared GPU)
        ok, err = 0, repr(e)[:200]
    flag = torch.tensor([ok], dtype=torch.int32, device=device if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        return {"error": err or "another rank could not build its replica"}
    got = {}

    def step():
        got["DI"] = ix.search(xq, k)

    steps = max(5, args.steps // 2)
    dt = timed(step, steps, 2, world, device)
    out = {"value": nq * steps / dt, "unit": "queries/s", "ms_per_step": dt / steps * 1e3, "queries_per_rank": (nq + world - 1) // world,
           "rows_per_rank": n, "exchange": "all-gather of the result rows, no rank merge",
           "ids_equal": digest(got["DI"][1]) == ids_sha, "scores_equal": digest(got["DI"][0]) == scores_sha}
    ix.close()
    del rows
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    # stdout carries exactly ONE line, the JSON result: everything else that writes to file descriptor 1 (RCCL prints a
    # five-line version banner there when a communicator is created) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} under a launcher with WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or (args.force_collective and "MASTER_ADDR" in os.environ):
        backend = os.environ.get("PROQA_DIST_BACKEND", "nccl")                          # "nccl" is RCCL on ROCm
        dist.init_process_group(backend=backend, **({"device_id": device} if backend == "nccl" else {}))

    from proqa_amd.index import IndexFlatIP, ShardedIndexFlatIP, shard_bounds

    n, nq, k = args.rows, args.queries, args.topk
    lo, hi = shard_bounds(n, world, rank)
    xb = gen_rows(lo, hi, device)
    xq = gen_queries(nq, device)
    sharded = ShardedIndexFlatIP(n, preallocate=False, transport=args.transport)
    sharded.adopt_local(xb)
    sharded.prepare()      # the int8 copy of this rank's rows now (otherwise the first enqueued step's finish builds it)
    result = {}

    def step():
        result["DI"] = sharded.search(xq, k, force_collective=args.force_collective)

    dt = timed(step, args.steps, args.warmup, world, device)
    qps = nq * args.steps / dt
    # kernel-level timing for the roofline: three more searches with HIP events bracketing every
    # mips_filter launch on the search stream (kept out of the timed region: ~60 us per search)
    local = sharded.local_index

    def filter_ms_of_three():
        local.set_profiling(True)
        filt = []
        for _ in range(3):
            step()
            filt.append(local.last_stats()["filter_ms"])
        local.set_profiling(False)
        st_ = local.last_stats()
        st_["filter_ms"] = float(np.mean(filt))
        return st_

    st = filter_ms_of_three()
    import hashlib
    digest = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()   # noqa: E731
    ids_sha, scores_sha = digest(result["DI"][1]), digest(result["DI"][0])
    # roofline of the dominant kernel: ALGORITHMIC flops 2*Q*N_local*d of one search (the fp16 inner products the result is
    # made of) over the HIP-event time of its filter launches (recorded on the search stream).  When the
    # rounds ran on the int8 copy of the rows (nomination + exact re-scoring: the result is the fp16 scan's bit for bit), the
    # same flops are also priced as executed int8 operations against the int8 peak, and the fp16 scan of the same search is
    # timed beside it (fp16_scan) with its digests compared.
    flops = 2.0 * nq * (hi - lo) * D
    filter_s = st["filter_ms"] / 1e3
    tflops = flops / filter_s / 1e12
    nominated = bool(st.get("nomination"))
    fp16_scan = None
    if nominated:
        local.configure_nomination("off")
        n16 = max(3, args.steps // 2)
        dt16 = timed(step, n16, 1, world, device)
        st16 = filter_ms_of_three()
        local.configure_nomination("auto")
        t16 = flops / (st16["filter_ms"] / 1e3) / 1e12
        fp16_scan = {"value": nq * n16 / dt16, "unit": "queries/s", "ms_per_step": dt16 / n16 * 1e3, "filter_ms_per_search": st16["filter_ms"],
                     "achieved": t16, "frac": t16 / PEAK_MFMA_F16_TFLOPS, "kernel": "mips_filter_f16",
                     "candidates_per_query": st16["candidates"] / max(nq, 1), "rounds": st16["rounds"],
                     "ids_equal": digest(result["DI"][1]) == ids_sha, "scores_equal": digest(result["DI"][0]) == scores_sha}
        step()   # (leave the nominated result in place for the legs below)
    # the same search on ordinary rounds (thresholds at the k-th best) beside the leaping ones
    # (not under --skip-extras: the profile passes of scripts/collect_profiles.sh count launches per search)
    ordinary = None
    if st.get("leap_rank") and not args.skip_extras:
        local.configure_leap("off")
        n_ord = max(3, args.steps // 2)
        dt_ord = timed(step, n_ord, 1, world, device)
        st_ord = local.last_stats()
        local.configure_leap("auto")
        ordinary = {"ms_per_step": dt_ord / n_ord * 1e3, "rounds": st_ord["rounds"],
                    "ids_equal": digest(result["DI"][1]) == ids_sha, "scores_equal": digest(result["DI"][0]) == scores_sha}
        step()

    # what ran where: every rank reports its process, GPU and shard (the record proves N ranks on N devices); the
    # digests of the merged result let two runs of the same workload (e.g. --gpus 1 and --gpus 2) be compared bit for bit
    props = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "pid": os.getpid(), "device": local_rank, "pci_bus_id": getattr(props, "pci_bus_id", None), "rows": [lo, hi]}
    ranks = [me]
    if dist.is_initialized():
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    comm_info = None
    if getattr(sharded, "_comm", None) is not None:
        import ctypes
        from proqa_amd import _lib
        ws_, rk_ = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().proqa_comm_info(sharded._comm, ctypes.byref(ws_), ctypes.byref(rk_)))
        comm_info = {"world_size": ws_.value, "rank": rk_.value}

    # The roofline is that of the instruction the dominant kernel ISSUES: v_mfma_i32_32x32x32_i8 against the dense int8 peak
    # when the rounds scanned the int8 copy (the 2 Q N d multiply-adds are executed as int8 operations), the fp16 MFMA peak
    # for the fp16 scan.  `fp16_equivalent_frac` prices the same algorithmic flops against the fp16 peak (the inner products
    # the result is made of): a secondary figure, not a fraction of the kernel's roofline.  filter_ms_per_search is the scan
    # alone; the exact re-scoring of the nominated rows runs inside the topk_merge launches and is part of
    # `chain_ms_per_search` (step - filter: bootstrap, merges incl. re-scoring, finalize, launch gaps, the host wait).
    row_bytes_scanned = D * (1 if nominated else 2)
    hbm_gbs = (hi - lo) * row_bytes_scanned / filter_s / 1e9
    peak = PEAK_MFMA_I8_TOPS if nominated else PEAK_MFMA_F16_TFLOPS
    roofline = {"bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TOP/s" if nominated else "TFLOP/s",
                "frac": tflops / peak, "traffic": pmc_traffic((hi - lo) / n, nominated),
                "traffic_commit": pmc_traffic_commit(), "traffic_age_commits": pmc_traffic_age(),
                "kernel": "mips_filter_i8" if nominated else "mips_filter_f16", "nomination": "int8" if nominated else None,
                "instruction": "v_mfma_i32_32x32x32_i8" if nominated else "v_mfma_f32_32x32x16_f16",
                "filter_ms_per_search": st["filter_ms"], "chain_ms_per_search": dt / args.steps * 1e3 - st["filter_ms"],
                "hbm_bytes_per_row_scanned": row_bytes_scanned, "hbm_achieved_GBs": hbm_gbs}
    if nominated:
        roofline["fp16_equivalent_frac"] = tflops / PEAK_MFMA_F16_TFLOPS
        roofline["nominated_per_query"] = st["nominated"] / max(nq, 1)
        # what the exact re-scoring gathers beside the scan: one 256-byte fp16 row per nominated row (in the merges)
        roofline["rescore_gather_bytes"] = float(st["nominated"]) * D * 2
    line = {
        "metric": "queries/sec top-80 MIPS over 18M x 128 index", "value": qps, "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        # what the timed path computes in: the f16 scan (fp32 accumulate), or int8 MFMA rounds that nominate + the same f16
        # arithmetic on the nominated rows (the result is the f16 scan's, bit for bit: fp16_scan.ids_equal / scores_equal)
        "dtype": "int8 nomination + f16 re-scoring" if nominated else "f16",
        "data": "synthetic",
        "config": {"workload": f"top-{k} MIPS, {nq} q x {n} x 128 fp16 rows in HBM (BASELINE configs[2]; x{world} row shards: configs[3])",
                   "rows": n, "queries": nq, "topk": k, "parallelism": f"corpus-row-shard x{world}",
                   "exchange": ("none" if world == 1 and not args.force_collective else
                                f"all-gather top-{k}, {args.transport}, "
                                f"{dist.get_backend() if dist.is_initialized() else 'rccl (library communicator)'}"),
                   "rounds": st["rounds"], "fallback_rounds": st["fallback_rounds"],
                   # leaping rounds (thresholds at rank leap_rank < k of the running lists, verified by the merges) and the
                   # same search on ordinary rounds timed beside them
                   "leap_rank": st.get("leap_rank"), "ordinary_rounds": ordinary,
                   "candidates_per_query": st["candidates"] / max(nq, 1),
                   "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                   "backend": dist.get_backend() if dist.is_initialized() else None,
                   "library_communicator": comm_info, "ranks": ranks},
        "result": {"ids_sha256": ids_sha[:16], "scores_sha256": scores_sha[:16]},   # first 16 hex digits of the digests
        "roofline": roofline,
    }
    if fp16_scan:
        line["fp16_scan"] = fp16_scan

    # N > 1: the SAME job with the rows replicated and the QUERIES sharded -- every rank holds all N rows (+ their int8 copy:
    # 6.9 GB of 288 at 18M rows) and searches its nq / world queries; one all-gather of the result rows, no rank merge.
    # `value` above stays the row-sharded figure BASELINE configs[3] prescribes; this is what the machine's memory allows
    # (DESIGN.md section 7).  Any failure here is reported in the object and never costs the line.
    if world > 1 and dist.is_initialized() and not args.skip_extras:
        line["query_shards"] = query_shards_leg(args, n, nq, k, world, rank, device, xq, ids_sha, scores_sha, digest)

    if dist.is_initialized():
        # where a sharded step goes: the rank-local search alone (max over ranks) vs the whole step with the exchange
        n_loc = max(5, args.steps // 2)
        dt_loc = timed(lambda: sharded.local_index.search_device(xq, k, idx_offset=lo), n_loc, 1, world, device)
        line["sharded_step"] = {"local_search_ms": dt_loc / n_loc * 1e3, "exchange_and_merge_ms": dt / args.steps * 1e3 - dt_loc / n_loc * 1e3,
                                "rows_per_rank": hi - lo}

    # the HBM-bound regime of the same kernel (north_star: "achieved HBM GB/s for the MIPS scan"):
    # 32 queries over the same shard -- intensity 32 flop/B, the corpus stream is the bound
    nq_small = 32

    def small_filter_ms():
        for profiled in (False, True):
            local.set_profiling(profiled)
            small = [None] * 3
            for i in range(3):
                local.search_device(xq[:nq_small], k)
                small[i] = local.last_stats()["filter_ms"]
        local.set_profiling(False)
        return float(np.mean(small)) / 1e3, bool(local.last_stats().get("nomination"))

    small_s, small_nom = small_filter_ms()
    row_bytes = D * (1 if small_nom else 2)          # what the scan streams per row: the int8 copy, or the fp16 rows
    small_gbs = (hi - lo) * row_bytes / small_s / 1e9
    # achieved = bytes the scan STREAMS per search (rows x 128 B of the int8 copy under the nomination scan, rows x 256 B
    # of fp16 otherwise) over the HIP-event time of all filter launches of the search; fp16_rows_GBs prices the same time
    # against the algorithmic bytes of SURVEY 8(d) (rows x 256 B): it may exceed the HBM peak, the bytes are not read
    line["scan_small_batch"] = {
        "queries": nq_small, "filter_ms_per_search": small_s * 1e3,
        "roofline": {"bound": "hbm", "achieved": small_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": small_gbs / PEAK_HBM_GBS, "kernel": "mips_filter_i8<QW=1>" if small_nom else "mips_filter_f16<QW=1>",
                     "bytes_per_row": row_bytes, "fp16_rows_GBs": (hi - lo) * D * 2 / small_s / 1e9}}
    if small_nom:
        local.configure_nomination("off")
        s16, _ = small_filter_ms()
        local.configure_nomination("auto")
        g16 = (hi - lo) * D * 2 / s16 / 1e9
        line["scan_small_batch"]["fp16_scan"] = {"filter_ms_per_search": s16 * 1e3, "achieved": g16, "frac": g16 / PEAK_HBM_GBS}

    if world == 1 and not args.skip_extras:
        # per-rank work of the strong-scaling runs, timed on this one GPU: the same queries over the first
        # N/G rows (G = 1, 2, 4, 8) -- what a rank of a G-GPU job does before the all-gather
        sweep = []
        for g_ in (1, 2, 4, 8):
            rows_g = n // g_
            ix = IndexFlatIP(128)
            ix.adopt_device(xb[:rows_g])
            dt_g = timed(lambda: ix.search_device(xq, k), max(5, args.steps // 2), 2, 1, device)
            ms_g = dt_g / max(5, args.steps // 2) * 1e3
            sweep.append((g_, rows_g, ms_g))
            ix.close()
        # the same per-rank searches as a STREAM of batches: two index handles on two streams, the next search enqueued
        # before the host waits for the current one (proqa_amd.index.PipelinedSearcher) -- a throughput figure for callers
        # with many batches; the headline and ms_per_search above stay one search at a time
        from proqa_amd.index import PipelinedSearcher
        piped = []
        for g_ in (1, 2, 4, 8):
            ps = PipelinedSearcher(xb[:n // g_])
            n_b = max(10, args.steps)
            for _ in ps.search_batches([xq] * 4, k):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in ps.search_batches([xq] * n_b, k):
                pass
            torch.cuda.synchronize()
            piped.append((time.perf_counter() - t0) / n_b * 1e3)
            ps.close()
            del ps
        torch.cuda.empty_cache()
        # N=1 timing of the per-rank search of a G-rank job (no collective): parallel arrays
        line["shard_sweep"] = {"ranks": [p_[0] for p_ in sweep], "rows_per_rank": [p_[1] for p_ in sweep],
                               "ms_per_search": [p_[2] for p_ in sweep], "pipelined_ms_per_search": piped}
        # the reference's large-k callers on the same index: retrieval/trec_process.py:76 (6980 MS MARCO dev queries,
        # top-10000 of 8.8M passages) and qa/online_sampler.py:113 (one question, k = 5000)
        large = {}
        for name, rows_l, nq_l, k_l in (("trec_top10000", min(n, 8_800_000), 6980, 10000), ("one_question_top5000", n, 1, 5000)):
            ix = IndexFlatIP(128)
            ix.adopt_device(xb[:rows_l])
            xq_l = gen_queries(nq_l, device)
            dt_l = timed(lambda: ix.search_device(xq_l, k_l), 3, 1, 1, device)
            st_l = ix.last_stats()
            large[name] = {"rows": rows_l, "queries": nq_l, "topk": k_l, "ms_per_search": dt_l / 3 * 1e3,
                           "rounds": st_l["rounds"], "fallback_rounds": st_l["fallback_rounds"]}
            ix.close()
            del ix, xq_l
        torch.cuda.empty_cache()
        line["large_k"] = large
        if not args.skip_cli_eval:
            line["search_cli_eval"] = cli_eval_leg(args, device, xb, xq, result["DI"][1], k)
        line["peak_measured"] = measured_peaks(device)
        # the other "next" rows of SURVEY section 8(f), on the same resident rows
        line["kmeans"] = kmeans_leg(args, device, xb)
        line["online"] = online_leg(args, device, sharded.local_index, hi - lo)
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.skip_cpu:   # CPU baselines: single-GPU runs only (contract)
        # CPU baseline + id parity on a bounded sample of the same workload
        cores = host_cores()
        ns, qs = hi - lo, nq   # rank 0's whole shard: ~10 s of NumPy/BLAS work on 16 usable cores at 18M rows
        cpu_qps, cdt, Do, Io = cpu_search_baseline(xb[:ns], xq[:qs], k, n, cores)
        ix = IndexFlatIP(128)
        ix.adopt_device(xb[:ns])
        Dg, Ig = ix.search_device(xq[:qs], k)
        Ig = Ig.cpu().numpy()
        rec = {f"overlap@{c}": float(np.mean([len(set(a[:c]) & set(b[:c])) / c for a, b in zip(Ig, Io)]))
               for c in (5, 20, 80)}
        line["cpu_baseline"] = {"value": cpu_qps, "unit": "queries/s", "cores": cores, "kind": "port",
                                "sample": f"{qs} q x {ns} rows, NumPy eval_retrieval.py:98-104 ({cdt:.1f} s), scaled to {n} rows"}
        line["recall_parity"] = dict(rec, sample=f"GPU vs NumPy oracle id overlap, {qs} q x {ns} rows",
                                     max_abs_score_diff=float(np.abs(Dg.cpu().numpy() - Do).max()))
    if world > 1:
        dist.barrier()

    del sharded, xb
    torch.cuda.empty_cache()
    if not args.skip_float32:
        line["float32_index"] = float32_leg(args, device, world, rank, lo, hi)
    if not args.skip_encode:
        torch.cuda.empty_cache()
        enc = encode_leg(args, device, world, rank)
        line["encode"] = enc

    if rank == 0:
        sys.stdout.flush()
        full = json.dumps(line)
        try:   # the unabridged record, for the builder (gpurun_out/ is scratch)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
        small = compact(line)
        text = json.dumps(small, separators=(",", ":"))
        # the driver keeps the tail of stdout: should the line still outgrow LINE_BYTES, secondary detail goes first
        for path in LINE_TRIM_ORDER:
            if len(text) <= LINE_BYTES:
                break
            node = small
            for key in path[:-1]:
                node = node.get(key, {}) if isinstance(node, dict) else {}
            if isinstance(node, dict) and node.pop(path[-1], None) is not None:
                text = json.dumps(small, separators=(",", ":"))
        print(f"bench.py: JSON line {len(text)} bytes (unabridged {len(full)})", file=sys.stderr)
        os.write(real_stdout, (text + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


LINE_BYTES = 6000   # the printed line stays below this (tests/test_bench_gpu.py)
LINE_TRIM_ORDER = [("encode", "cli_text_non_ascii"), ("search_cli_eval", "loader"), ("encode", "cli_text", "tokenise_only"),
                   ("kmeans", "cpu_baseline"), ("search_cli_eval", "host_api_search"), ("online",),
                   ("encode", "corpus_1m"), ("encode", "cli_text"), ("kmeans",), ("float32_index",), ("large_k",), ("peak_measured",)]
MAX_STRING = 96   # longer strings are prose: they live in the docstring above / DESIGN.md, not in the line
# keys of the unabridged record (gpurun_out/bench_full.json) that the printed line leaves out below its top level: prose,
# restatements of the command line, intermediate timings a stage split already covers
NESTED_DROP = {"note", "metric", "gemm_kernel_full", "host_cores", "loader_workers", "mean_tokens_per_passage", "feed_seconds", "upload_seconds",
               "loader_wait_seconds", "gpu_busy_seconds", "bytes_written_per_rank", "parity_rows", "prepare_seconds_untimed",
               "index_file_bytes", "cache_drop_seconds", "iterations_warm", "iterations_timed", "scorer_processes", "seconds_d2h",
               "seconds_npy_write", "dd_style_read_cold_GBs", "dd_style_read_warm_GBs", "idx2id_json_route", "recall_expected",
               "objective", "tokens_per_s", "gflop_per_passage_reference", "entries_sample", "higher_is_better", "seconds_encode_loop",
               "traffic_age_commits", "points", "centroids", "non_ascii_passage_fraction", "recall_printed", "hbm_bytes_algorithmic",
               "question_tokens", "questions", "graph_equals_plain"}


def compact(obj, depth=0, parent=None):
    """The printed form of the result: no prose ("note", nested "metric" / "workload" strings), floats at five significant
    digits; `workload` survives in `config`, `sample` in `cpu_baseline` / `parity` (the contract asks for them)."""
    if isinstance(obj, dict):
        out = {}
        for k, v in obj.items():
            if depth > 0 and k in NESTED_DROP:
                continue
            if depth > 1 and k in ("unit", "passages", "dtype", "steps", "scaling") and parent not in ("roofline", "cpu_baseline"):
                continue   # sub-legs inherit them from their leg (a roofline / cpu_baseline object keeps its unit)
            if k == "workload" and parent != "config":
                continue
            if isinstance(v, str) and len(v) > MAX_STRING and k not in ("workload", "sample"):
                continue
            if v is None and depth > 0 and k != "traffic":   # "not applicable" below the top level: the key's absence says it
                continue
            out[k] = compact(v, depth + 1, k)
        return out
    if isinstance(obj, (list, tuple)):
        return [compact(v, depth + 1, parent) for v in obj]
    if isinstance(obj, float):   # five significant digits; four below the second level
        return float(f"{obj:.5g}" if depth <= 2 else f"{obj:.4g}") if obj == obj and abs(obj) != float("inf") else None
    if isinstance(obj, str) and len(obj) > MAX_STRING:
        return obj[:MAX_STRING - 3] + "..."
    return obj


def measured_peaks(device):
    """Stream and MFMA micro-benchmarks of libproqa_hip.so on this box: the ceilings a HIP kernel
    reaches here, next to the spec peaks the roofline fractions are priced against."""
    import ctypes
    from proqa_amd import _lib
    lib = _lib.load()
    nbytes = 2 << 30
    buf = torch.empty(2 * nbytes, dtype=torch.uint8, device=device)
    buf.zero_()
    torch.cuda.synchronize()
    out = {}
    v = ctypes.c_double()
    st = _lib.current_stream_ptr()
    for name, kind in (("hbm_copy_GBs", 0), ("hbm_read_GBs", 1)):
        _lib.check(lib.proqa_microbench_stream(buf.data_ptr(), nbytes, kind, 5, st, ctypes.byref(v)))
        out[name] = v.value
    del buf
    for name, ms, zero in (("mfma_f16_TFLOPs_8ms_random_operands", 8.0, 0), ("mfma_f16_TFLOPs_8ms_zero_operands", 8.0, 1)):
        _lib.check(lib.proqa_microbench_mfma(ms, zero, st, ctypes.byref(v)))
        out[name] = v.value
    # the int8 instruction the nomination scan issues, and the 16x16x64 shape the guide quotes its int8 ceiling for
    for name, zero, shape in (("mfma_i8_TOPs_8ms_random_operands", 0, 0), ("mfma_i8_TOPs_8ms_zero_operands", 1, 0),
                              ("mfma_i8_16x16x64_TOPs_8ms_random", 0, 1)):
        _lib.check(lib.proqa_microbench_mfma_i8_shape(8.0, zero, shape, st, ctypes.byref(v)))
        out[name] = v.value
    out["note"] = ("float4 grid-stride copy / read of 2 GiB; 4 independent v_mfma_f32_32x32x16_f16 (v_mfma_i32_32x32x32_i8, "
                   "_16x16x64_i8) chains per wave from registers, 8 waves per CU; best of repeated launches")
    return out


def encode_traffic(batch, seq_len):
    """HBM bytes one 512 x 128 encode step reads, all its kernels together (committed rocprofv3 PMC pass,
    FETCH_SIZE x 2 per the gfx950 note); None for other shapes."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        return t.get("encode_hbm_read_bytes_per_step") if (batch, seq_len) == tuple(t.get("encode_step_shape", ())) else None
    except Exception:
        return None


def pmc_traffic(shard_fraction, nominated=False):
    """HBM bytes the filter launches of one search read (committed rocprofv3 PMC pass of the 1-GPU run, FETCH_SIZE x 2 per
    the gfx950 note), scaled to this rank's share of the corpus; None when the committed pass measured the other scan."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if bool(t.get("nomination")) != bool(nominated):
            return None
        return t.get("hbm_bytes_per_search") * shard_fraction
    except Exception:
        return None


def pmc_traffic_age():
    """Commits that touched csrc/mips_kernels.hip since the commit profiles/pmc_traffic.json was measured on (None if git
    cannot tell): a non-zero value says the `traffic` reading predates the kernel that ran."""
    import subprocess
    c = pmc_traffic_commit()
    if not c:
        return None
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-list", "--count", f"{c}..HEAD", "--", "proqa_amd/csrc/mips_kernels.hip"],
                             capture_output=True, text=True, timeout=10)
        return int(out.stdout.strip()) if out.returncode == 0 and out.stdout.strip() else None
    except Exception:
        return None


def pmc_traffic_commit():
    """The commit profiles/pmc_traffic.json was measured on (stamped when the profile set is refreshed): `traffic` is a
    committed counter reading, not a measurement of this run, and goes stale when a kernel changes."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get("measured_on_commit")
    except Exception:
        return None


if __name__ == "__main__":
    main()
