"""Encode a JSON-lines file of questions or passages into a [N,128] .npy index.

Drop-in for /root/reference/retrieval/get_embed.py (load_saved :22-27, main :29-139,
predict :142-172): same flags (proqa_amd.config), same input files, same output format.

    python get_embed.py --do_predict --predict_batch_size 512 --bert_model_name bert-base-uncased \
        --fp16 --predict_file F --init_checkpoint CKPT [--is_query_embed] --embed_save_path OUT

Differences that are intended: the model runs on the MI355X kernels of libproqa_hip.so (no
apex); under torchrun (WORLD_SIZE > 1) or --local_rank the input rows are split into contiguous
ranges, one per GPU, and every rank writes its slice of a pre-sized .npy — the reference's
DataParallel/DDP wrappers have no get_embed and cannot do this.
"""
import json
import os
import sys
import random

import numpy as np
import torch

from . import npy
from .config import get_args
from .datasets import JsonlTexts, TextBatchLoader, TokenizeCollate
from .retriever import BertForRetriever, config_from_dict
from .utils import move_to_cuda


def load_saved(model, path):
    """torch state_dict checkpoint -> model; tolerates DataParallel's 'module.' prefix."""
    state_dict = torch.load(path, map_location="cpu")
    state_dict = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
    model.load_state_dict(state_dict)
    return model


def load_bert_config(name_or_dir):
    """config.json of a local model directory, else transformers' BertConfig.from_pretrained."""
    cfg_path = os.path.join(name_or_dir, "config.json")
    if os.path.isfile(cfg_path):
        with open(cfg_path) as f:
            return config_from_dict(json.load(f))
    from transformers import BertConfig
    return config_from_dict(BertConfig.from_pretrained(name_or_dir).to_dict())


def _dist_env(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.local_rank != -1:
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", max(args.local_rank, 0)))
        return world, rank, local_rank
    return 1, 0, 0


LAST_RUN_STATS = {}     # filled by main(): see predict(stats=...) (bench.py's encode.cli_text reads it)


def usable_cpus():
    """CPUs this process may use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


# Batches in flight.  Two streams overlapped one batch's HBM-bound kernels with the other's GEMMs for
# ~2.5 %, but two hipBLASLt stream-K GEMMs running concurrently deadlock (their workgroups spin on
# partial tiles of partners that cannot be scheduled) as soon as M is not a multiple of the tile --
# any ragged batch -- so the encoder stays on ONE side stream (kernel launches are asynchronous: the
# host collates and uploads batch i+1 while batch i computes).
N_STREAMS = 1


def _right_padded(mask):
    """em_collate pads on the right: every mask row must be a prefix of True (checked on the host)."""
    return mask.shape[1] < 2 or not bool((mask[:, 1:] & ~mask[:, :-1]).any())


class _PinnedStager:
    """Uploads host batches through a small ring of pinned buffers that are allocated once.

    A loader with pin_memory=True pins every batch in a thread of this process with torch's (thread-pool) copy into a
    freshly pinned allocation.  Here a batch is memcpy'd into a slot of the ring (0.6 MB, ~50 us) and uploaded from there;
    a slot is reused once the upload that read it has completed."""

    def __init__(self, device, slots=4):
        self.device = device
        self.slots = [dict(bufs={}, event=None) for _ in range(slots)]
        self.next = 0

    def upload(self, batch):
        if all(not torch.is_tensor(v) or v.is_pinned() or v.is_cuda for v in batch.values()):
            return move_to_cuda(batch)                      # the caller pinned it (or it is on the GPU already)
        slot = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        if slot["event"] is not None:
            slot["event"].synchronize()
        out = {}
        for k, v in batch.items():
            if not torch.is_tensor(v) or v.is_cuda:
                out[k] = v
                continue
            buf = slot["bufs"].get(k)
            if buf is None or buf.dtype != v.dtype or buf.numel() < v.numel():
                buf = torch.empty(max(v.numel(), 1), dtype=v.dtype, pin_memory=True)
                slot["bufs"][k] = buf
            staged = buf[:v.numel()].view(v.shape)
            # (a plain memcpy: torch's copy_ goes through its intra-op thread pool above 32k elements, and that pool is as
            # wide as the machine -- 256 spinning threads on the GPU boxes exhaust the process's CPU quota within a batch)
            np.copyto(staged.numpy(), v.contiguous().numpy())
            out[k] = staged.to(self.device, non_blocking=True)
        slot["event"] = torch.cuda.Event()
        slot["event"].record()
        return out


def predict(args, model, eval_dataloader, device, fp16=False, is_query_embed=True, stats=None):
    """The reference's hot loop: move batch to the GPU, get_embed, keep embeddings on device.
    Batches alternate between N_STREAMS HIP streams (order of the results is kept).
    A batch may carry its valid lengths as the host list 'seq_lens' (TokenizeCollate does); a batch of the reference's
    shape (ids + mask only) has them taken from the mask.
    stats (optional dict): filled with 'batches', 'passages', 'gpu_busy_seconds' (HIP-event time of the get_embed calls),
    'loop_seconds' (wall time of the loop incl. the wait for the last batch), 'loader_wait_seconds' (blocked in the
    loader) and 'feed_seconds' (upload + launches) -- what a run needs to tell whether the host kept the GPU fed."""
    import time
    model.eval()
    if fp16:
        model.half()
    chunks = []
    streams = [torch.cuda.Stream(device=device) for _ in range(N_STREAMS)]
    main = torch.cuda.current_stream(device)
    for s in streams:
        s.wait_stream(main)     # pool streams are non-blocking: order them after the weight preparation on `main`
    events = []
    stager = _PinnedStager(device)
    t_loop = time.perf_counter()
    t_wait = t_feed = t_upload = 0.0
    it = iter(eval_dataloader)
    i = -1
    while True:
        t0 = time.perf_counter()
        try:
            batch = next(it)
        except StopIteration:
            break
        i += 1
        t1 = time.perf_counter()
        t_wait += t1 - t0
        lens = batch.pop("seq_lens", None) if isinstance(batch, dict) else None
        if lens is None:
            if not _right_padded(batch["input_mask"]):
                raise ValueError("input_mask must be right-padded (a prefix of True per row), as em_collate produces")
            lens = batch["input_mask"].sum(dim=1).tolist()      # host-side: lets the encoder skip the padding
        s = streams[i % N_STREAMS]
        with torch.cuda.stream(s), torch.no_grad():
            batch_to_feed = stager.upload(batch)
            t_upload += time.perf_counter() - t1
            if stats is not None:
                events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                events[-1][0].record()
            chunks.append(model.get_embed(batch_to_feed, is_query_embed, check_mask=False,
                                          seq_lens_host=lens)["embed"])
            if stats is not None:
                events[-1][1].record()
        t_feed += time.perf_counter() - t1
    for s in streams:
        main.wait_stream(s)
    if stats is not None:
        torch.cuda.synchronize(device)
        stats.update(batches=len(chunks), passages=int(sum(c.shape[0] for c in chunks)),
                     gpu_busy_seconds=sum(a.elapsed_time(b) for a, b in events) / 1e3,
                     loop_seconds=time.perf_counter() - t_loop, loader_wait_seconds=t_wait, feed_seconds=t_feed, upload_seconds=t_upload)
    if chunks:
        embeds = torch.cat(chunks)
    else:
        embeds = torch.empty((0, 128), dtype=model.out_dtype, device=device)
    model.train()
    return embeds


def main(argv=None):
    args = get_args(argv)
    is_query_embed = args.is_query_embed

    if args.accumulate_gradients < 1:
        raise ValueError("Invalid accumulate_gradients parameter: {}, should be >= 1".format(
            args.accumulate_gradients))
    if not args.do_train and not args.do_predict:
        raise ValueError("At least one of `do_train` or `do_predict` must be True.")
    if args.do_train:
        raise ValueError("proqa_amd.get_embed implements the encode path only; training is out of scope")
    if not args.predict_file:
        raise ValueError("If `do_predict` is True, then `predict_file` must be specified.")
    if args.no_cuda:
        raise RuntimeError("--no_cuda: proqa_amd has no CPU path (the reference's move_to_cuda is "
                           "unconditional as well, retrieval/utils.py:11)")

    world, rank, local_rank = _dist_env(args)
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible: the encode path has no CPU fallback")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and not torch.distributed.is_initialized():
        # RCCL ("nccl") on ROCm; only barriers are issued.  PROQA_DIST_BACKEND=gloo lets two ranks
        # share one GPU (RCCL refuses that), which is how the GPU test exercises this path.
        torch.distributed.init_process_group(backend=os.environ.get("PROQA_DIST_BACKEND", "nccl"))

    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)

    # (read by the tokenizer library's thread pool when it is first used)
    os.environ.setdefault("RAYON_NUM_THREADS", str(max(1, min(args.eval_workers, usable_cpus() - 2))))
    from transformers import BertTokenizer
    bert_config = load_bert_config(args.bert_model_name)
    model = BertForRetriever(bert_config, args, device=device)
    tokenizer = BertTokenizer.from_pretrained(args.bert_model_name)

    # (the sentences EmDataset would tokenise, parsed line by line in the loader's producer thread: datasets.JsonlTexts)
    dataset = JsonlTexts(args.predict_file, args.max_query_length, args.max_seq_length, is_query_embed)
    n_total = len(dataset)
    lo, hi = (n_total * rank) // world, (n_total * (rank + 1)) // world
    # The loader hands over whole tokenised batches (datasets.TokenizeCollate: same ids / masks as EmDataset + em_collate)
    # from a background thread of this process (datasets.TextBatchLoader); --eval-workers is the size of the tokenizer's
    # thread pool, capped two below the CPUs this process may use: one for the thread that launches the kernels, one for
    # the producer.  --eval-workers 0 tokenises in the consumer thread, batch by batch.
    texts = dataset
    workers = max(0, min(args.eval_workers, usable_cpus() - 2))
    # (plain-ASCII sentences are tokenised by the library's own WordPiece on `workers` threads, the rest by the tokenizer)
    collate = TokenizeCollate(tokenizer, dataset.max_length, parallel=workers > 1, native_threads=workers)
    if workers > 0:
        loader = TextBatchLoader(texts, args.predict_batch_size, collate, prefetch=8, lo=lo, hi=hi)
    else:
        loader = (collate([texts[i] for i in range(b0, min(b0 + args.predict_batch_size, hi))])
                  for b0 in range(lo, hi, args.predict_batch_size))

    assert args.init_checkpoint != ""
    model = load_saved(model, args.init_checkpoint)
    model.to(device)

    # output dtype: fp16 under --fp16 (apex O1 emits half) or --efficient_eval (.half()), else fp32
    want_half = args.fp16 or args.efficient_eval
    if args.embed_dtype != "auto":
        want_half = args.embed_dtype in ("float16", "fp16", "f2")
    model.half() if want_half else model.float()

    LAST_RUN_STATS.clear()
    embeds = predict(args, model, loader, device, fp16=args.efficient_eval, is_query_embed=is_query_embed, stats=LAST_RUN_STATS)
    LAST_RUN_STATS["loader_workers"] = workers
    # which dense path the large layers ran on: the hipBLASLt kernel pinned by name, or rocblas_gemm_ex (the library falls
    # back silently when the loaded hipBLASLt does not hold the pinned kernel: a ROCm point release can change that)
    try:
        kernel = model.gemm_kernels().get(bool(is_query_embed), "")
        LAST_RUN_STATS["gemm_kernel"] = kernel or "rocblas_gemm_ex"
        if rank == 0:
            print("Dense layers: " + (f"hipBLASLt kernel {kernel}" if kernel else
                                      "rocblas_gemm_ex (no pinned hipBLASLt kernel in the loaded library, or batches below 4096 token rows)"),
                  file=sys.stderr)
    except Exception:   # noqa: BLE001  (a log line must not fail the run)
        pass
    del loader
    local = embeds.cpu().numpy()
    out_path = npy.save_path(args.embed_save_path)
    if world == 1:
        npy.save(out_path, local)
    else:
        if rank == 0:
            npy.create(out_path, n_total, 128, local.dtype)
        torch.distributed.barrier()
        npy.write_rows(out_path, lo, local)
        torch.distributed.barrier()
    return out_path


if __name__ == "__main__":
    main()
