"""Top-k retrieval + answer-recall evaluation.

Drop-in for /root/reference/retrieval/eval_retrieval.py:

    python eval_retrieval.py RAW_DATA INDEXPATH QUERY_EMBED DB [--topk 80] [--num-workers 10]
    torchrun --nproc-per-node 8 eval_retrieval.py ...     # the corpus row-sharded over 8 MI355X (see search())

Same inputs (QA JSON-lines, para_embed.npy, query .npy, sqlite DB, idx_id.json), same stdout
lines `Top {k} Recall for {n} QA pairs: {mean} ...`.  The faiss.IndexFlatIP search (:102-104)
is replaced by the MI355X exhaustive top-k of libproqa_hip.so; the string-match scoring
(:27-65) stays on the host in a process pool, forked BEFORE the GPU is touched as in the
reference (:92-96).
"""
import argparse
import json
import os
from collections import defaultdict
from functools import partial
from multiprocessing import Pool as ProcessPool
from multiprocessing.util import Finalize

import numpy as np

from .basic_tokenizer import SimpleTokenizer
from .utils import DocDB, normalize

PROCESS_TOK = None
PROCESS_DB = None
PROCESS_TEXTS = None    # TextSidecar of the index (row -> passage text), when one sits next to the id map
DEFAULT_IDX_ID = "../pretrained_models/idx_id.json"   # hard-coded in the reference (:69)
FIXED_CUTOFFS = (5, 10, 20, 50)


def init(db_path, text_sidecar=None):
    """Per-worker state: a tokenizer and a sqlite connection (and, when the index comes with one, its text sidecar)."""
    global PROCESS_TOK, PROCESS_DB, PROCESS_TEXTS
    PROCESS_TOK = SimpleTokenizer()
    Finalize(PROCESS_TOK, PROCESS_TOK.shutdown, exitpriority=100)
    PROCESS_DB = DocDB(db_path)
    Finalize(PROCESS_DB, PROCESS_DB.close, exitpriority=100)
    if text_sidecar:
        from .gen_index_id_map import TextSidecar
        PROCESS_TEXTS = TextSidecar(text_sidecar)


def _words(text):
    return PROCESS_TOK.tokenize(normalize(text)).words(uncased=True)


def _fold(text):
    """Lower-cased with both lower-case sigmas made one: a token's .lower() and the same characters inside the lower-cased
    paragraph can differ in nothing but the final-sigma rule (every other lower-case mapping is per character)."""
    return text.lower().replace("\u03c2", "\u03c3")


def _may_match(needles, folded_para):
    """False only if NO alias can occur in the paragraph: an alias matches as a contiguous run of tokens, every token is a
    substring of the (normalised) paragraph, so every token of a matching alias occurs in the folded paragraph as a folded
    substring.  An alias without tokens matches every paragraph (the reference's loop finds the empty run at i = 0)."""
    for needle in needles:
        if all(tok in folded_para for tok in needle):
            return True
    return False


def para_has_answer(answer, para, return_matched=False, needles=None):
    """True if any answer alias occurs in the paragraph as a contiguous token sequence (eval_retrieval.py:27-45 of the
    reference).  The paragraph is tokenised only if some alias has all its tokens somewhere in it -- a plain substring test on
    the folded text that cannot reject a match (`_may_match`); most of a question's top-k paragraphs hold none of them."""
    norm = normalize(para)
    if needles is None:
        needles = [_words(alias) for alias in answer]
    if not _may_match([[_fold(t) for t in needle] for needle in needles], _fold(norm)):
        return (False, "") if return_matched else False
    tokens = PROCESS_TOK.tokenize(norm)
    text = tokens.words(uncased=True)
    for needle in needles:
        span = len(needle)
        for i in range(len(text) - span + 1):
            if text[i:i + span] == needle:
                if return_matched:
                    return True, tokens.slice(i, i + span).untokenize()
                return True
    return (False, "") if return_matched else False


def _score_paras(answer, paras, topk):
    needles = [_words(alias) for alias in answer]      # once per question, not once per paragraph
    hits = [int(para_has_answer(answer, p, needles=needles)) for p in paras]
    scores = {str(topk): int(sum(hits) > 0)}
    for c in FIXED_CUTOFFS:
        scores[str(c)] = int(sum(hits[:c]) > 0)
    return scores


def get_score(answer_doc, topk=80):
    """Hit flags of one question at cut-offs topk, 5, 10, 20, 50 (computed on the top-k list)."""
    _question, answer, doc_ids = answer_doc
    paras = [PROCESS_DB.get_doc_text(doc_id) for doc_id in doc_ids][:topk]
    return _score_paras(answer, paras, topk)


def get_score_rows(answer_rows, topk=80):
    """get_score with the paragraphs taken from the index's text sidecar by ROW (gen_index_id_map.write_text_sidecar) instead
    of row -> doc id -> one sqlite query per hit."""
    _question, answer, rows = answer_rows
    paras = [PROCESS_TEXTS[int(r)] for r in rows][:topk]
    return _score_paras(answer, paras, topk)


def convert_idx2id(idxs, mapping_path=DEFAULT_IDX_ID):
    """Row indices of the index -> document ids via idx_id.json ({"<row>": doc_id}), or via the
    memory-mapped sidecar written by gen_index_id_map (path ending in .ids): same result."""
    if mapping_path.endswith(".ids"):
        from .gen_index_id_map import SidecarMap
        return SidecarMap(mapping_path).take(np.asarray(idxs))    # one gather + one parse for all ids
    with open(mapping_path) as f:
        idx_id = json.load(f)
    return [[idx_id[str(int(i))] for i in row] for row in idxs]


def dist_env():
    """(world_size, rank, local_rank) of a torchrun launch; (1, 0, 0) for a plain `python eval_retrieval.py`."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 1, 0, 0
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


LAST_RUN_STATS = {}     # filled by search() / main(): seconds per stage of the last run (bench.py's search.cli_eval reads it)


def search(indexpath, query_embed, topk, allow_rounding=False, readers=0, shard="rows"):
    """np.load + IndexFlatIP.add + search of the reference (:98-104), on the GPU.

    The index file goes from disk to HBM inside the library (proqa_index_add_npy: reader threads -> pinned ring -> PCIe,
    no host copy of the corpus).  fp16 .npy files (get_embed.py --fp16) are scanned as they are.  float32 files whose
    values fp16 cannot hold are searched in exact-float32 mode (float32 copies of the rows in HBM, fp16 scan + exact
    re-scoring) unless allow_rounding asks for the rounded, faster variant.

    Under torchrun (WORLD_SIZE > 1) the corpus is row-sharded: every rank loads ONLY rows [r*N/G, (r+1)*N/G) of the file
    into its GPU, all ranks search all queries, one all-gather of the per-shard lists (RCCL over xGMI) and a merge give
    every rank the result of the single-GPU search, bit for bit (ShardedIndexFlatIP).
    shard="queries": every rank loads ALL rows and searches its slice of the queries -- one all-gather of result rows, no
    merge (QueryShardedIndexFlatIP: what 288 GB per GPU allow)."""
    import time
    from . import npy
    xq = npy.load(query_embed)
    info = npy.stat(indexpath)
    if info["cols"] != 128 or xq.shape[1] != 128:
        raise ValueError("embeddings must be 128-d")
    world, rank, local_rank = dist_env()
    t0 = time.perf_counter()
    if world == 1:
        from .index import IndexFlatIP
        index = IndexFlatIP(128, capacity=info["rows"])         # (the first HIP call of the process: runtime start-up)
        if allow_rounding:
            index.allow_rounding(True)
        t0b = time.perf_counter()
        index.add_npy(indexpath, 0, info["rows"], readers)
        t1 = time.perf_counter()
        D, I = index.search(xq, topk)
        rows_here = info["rows"]
    else:
        import torch
        import torch.distributed as dist
        from .index import QueryShardedIndexFlatIP, ShardedIndexFlatIP
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        if not dist.is_initialized():
            # RCCL ("nccl") on ROCm.  PROQA_DIST_BACKEND=gloo lets two ranks share one GPU (RCCL refuses that), which is
            # how the GPU test exercises this path.
            dist.init_process_group(backend=os.environ.get("PROQA_DIST_BACKEND", "nccl"))
        if shard == "queries":
            index = QueryShardedIndexFlatIP()
        else:
            index = ShardedIndexFlatIP(info["rows"])
        if allow_rounding:
            index.local_index.allow_rounding(True)
        t0b = time.perf_counter()
        if shard == "queries":
            index.add_npy(indexpath, readers)
        else:
            index.add_local_npy(indexpath, readers)
        t1 = time.perf_counter()
        D, I = index.search(torch.from_numpy(np.ascontiguousarray(xq)).cuda(), topk)
        D, I = D.cpu().numpy(), I.cpu().numpy()
        rows_here = info["rows"] if shard == "queries" else index.hi - index.lo
    t2 = time.perf_counter()
    row_bytes = 128 * (2 if info["dtype"] == np.float16 else 4)
    LAST_RUN_STATS.update(world=world, rows=int(info["rows"]), rows_this_rank=int(rows_here), queries=int(xq.shape[0]),
                          gpu_init_seconds=t0b - t0, load_seconds=t1 - t0b, load_gbs=rows_here * row_bytes / max(t1 - t0b, 1e-9) / 1e9,
                          search_seconds=t2 - t1)
    index.close()
    return D, I


def finish_distributed():
    """After the sharded search: ranks leave the process group together (rank 0 goes on to score alone)."""
    import sys
    if dist_env()[0] == 1 and "torch.distributed" not in sys.modules:
        return               # a plain single-process run never imported PyTorch: nothing to leave
    try:
        import torch.distributed as dist
    except ImportError:
        return
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("raw_data", type=str, default=None)
    parser.add_argument("indexpath", type=str, default=None)
    parser.add_argument("query_embed", type=str, default=None)
    parser.add_argument("db", type=str, default=None)
    parser.add_argument("--topk", type=int, default=80)
    parser.add_argument("--num-workers", type=int, default=10)
    parser.add_argument("--idx-id-map", type=str, default=DEFAULT_IDX_ID,
                        help="idx_id.json (the reference reads ../pretrained_models/idx_id.json)")
    parser.add_argument("--allow-fp16-rounding", action="store_true",
                        help="round float32 embeddings to fp16 instead of searching them in exact-float32 mode")
    parser.add_argument("--no-text-sidecar", action="store_true",
                        help="fetch the passage texts from the sqlite DB by document id even if a text sidecar (<stem>.txt, "
                             "gen_index_id_map --texts) sits next to the id map")
    parser.add_argument("--shard", choices=("rows", "queries"), default="rows",
                        help="under torchrun: row shards + all-gather + merge (default), or every rank loads all rows and "
                             "searches its slice of the queries (no merge; 6.9 GB per GPU at 18M rows)")
    parser.add_argument("--dump-results", type=str, default=None,
                        help="also write the search result (D float32 [Q,k], I int64 [Q,k]) to this .npz (not in the reference)")
    return parser


def main(argv=None):
    import time
    args = build_parser().parse_args(argv)
    world, rank, _local_rank = dist_env()
    LAST_RUN_STATS.clear()
    t_start = time.perf_counter()
    if rank != 0:
        # a shard of the corpus and the collective; rank 0 maps ids, scores and prints
        search(args.indexpath, args.query_embed, args.topk, allow_rounding=args.allow_fp16_rounding, shard=args.shard)
        finish_distributed()
        return []
    with open(args.raw_data) as f:
        qas = [json.loads(line) for line in f.readlines()]
    questions = [item["question"] for item in qas]
    answers = [item["answer"] for item in qas]

    # the passage texts by row, when the index was built with them (gen_index_id_map --texts): no id mapping, no sqlite
    from .gen_index_id_map import text_sidecar_of
    text_sidecar = None if args.no_text_sidecar else text_sidecar_of(args.idx_id_map)
    # fork the scorer pool before any HIP call (and before the process group's threads exist)
    processes = ProcessPool(processes=args.num_workers, initializer=init, initargs=[args.db, text_sidecar])
    try:
        t0 = time.perf_counter()
        D, I = search(args.indexpath, args.query_embed, args.topk, allow_rounding=args.allow_fp16_rounding, shard=args.shard)
        finish_distributed()
        if args.dump_results:
            np.savez(args.dump_results, D=D, I=I)
        t1 = time.perf_counter()
        if text_sidecar:
            t2 = t1
            assert len(I) == len(questions) == len(answers)
            results = processes.map(partial(get_score_rows, topk=args.topk), zip(questions, answers, I.tolist()))
        else:
            retrieval_results = convert_idx2id(I, args.idx_id_map)
            t2 = time.perf_counter()
            assert len(retrieval_results) == len(questions) == len(answers)
            results = processes.map(partial(get_score, topk=args.topk), zip(questions, answers, retrieval_results))
        t3 = time.perf_counter()
    finally:
        processes.close()
        processes.join()
    LAST_RUN_STATS.update(startup_seconds=t0 - t_start, search_total_seconds=t1 - t0, idx2id_seconds=t2 - t1,
                          scoring_seconds=t3 - t2, scorer_processes=args.num_workers, text_sidecar=bool(text_sidecar))

    aggregate = defaultdict(list)
    for r in results:
        for key, value in r.items():
            aggregate[key].append(value)
    lines = []
    for key, values in aggregate.items():
        line = "Top {} Recall for {} QA pairs: {} ...".format(key, len(values), np.mean(values))
        print(line)
        lines.append(line)
    LAST_RUN_STATS["total_seconds"] = time.perf_counter() - t_start
    if os.environ.get("PROQA_STATS_JSON"):      # bench.py's search_cli_eval runs this command line as a child process
        with open(os.environ["PROQA_STATS_JSON"], "w") as f:
            json.dump(LAST_RUN_STATS, f)
    return lines


if __name__ == "__main__":
    main()
