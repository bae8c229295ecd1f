"""Top-k retrieval + answer-recall evaluation.

Drop-in for /root/reference/retrieval/eval_retrieval.py:

    python eval_retrieval.py RAW_DATA INDEXPATH QUERY_EMBED DB [--topk 80] [--num-workers 10]

Same inputs (QA JSON-lines, para_embed.npy, query .npy, sqlite DB, idx_id.json), same stdout
lines `Top {k} Recall for {n} QA pairs: {mean} ...`.  The faiss.IndexFlatIP search (:102-104)
is replaced by the MI355X exhaustive top-k of libproqa_hip.so; the string-match scoring
(:27-65) stays on the host in a process pool, forked BEFORE the GPU is touched as in the
reference (:92-96).
"""
import argparse
import json
from collections import defaultdict
from functools import partial
from multiprocessing import Pool as ProcessPool
from multiprocessing.util import Finalize

import numpy as np

from .basic_tokenizer import SimpleTokenizer
from .utils import DocDB, normalize

PROCESS_TOK = None
PROCESS_DB = None
DEFAULT_IDX_ID = "../pretrained_models/idx_id.json"   # hard-coded in the reference (:69)
FIXED_CUTOFFS = (5, 10, 20, 50)


def init(db_path):
    """Per-worker state: a tokenizer and a sqlite connection."""
    global PROCESS_TOK, PROCESS_DB
    PROCESS_TOK = SimpleTokenizer()
    Finalize(PROCESS_TOK, PROCESS_TOK.shutdown, exitpriority=100)
    PROCESS_DB = DocDB(db_path)
    Finalize(PROCESS_DB, PROCESS_DB.close, exitpriority=100)


def _words(text):
    return PROCESS_TOK.tokenize(normalize(text)).words(uncased=True)


def para_has_answer(answer, para, return_matched=False):
    """True if any answer alias occurs in the paragraph as a contiguous token sequence."""
    tokens = PROCESS_TOK.tokenize(normalize(para))
    text = tokens.words(uncased=True)
    for alias in answer:
        needle = _words(alias)
        span = len(needle)
        for i in range(len(text) - span + 1):
            if text[i:i + span] == needle:
                if return_matched:
                    return True, tokens.slice(i, i + span).untokenize()
                return True
    return (False, "") if return_matched else False


def get_score(answer_doc, topk=80):
    """Hit flags of one question at cut-offs topk, 5, 10, 20, 50 (computed on the top-k list)."""
    _question, answer, doc_ids = answer_doc
    paras = [PROCESS_DB.get_doc_text(doc_id) for doc_id in doc_ids][:topk]
    hits = [int(para_has_answer(answer, p)) for p in paras]
    scores = {str(topk): int(sum(hits) > 0)}
    for c in FIXED_CUTOFFS:
        scores[str(c)] = int(sum(hits[:c]) > 0)
    return scores


def convert_idx2id(idxs, mapping_path=DEFAULT_IDX_ID):
    """Row indices of the index -> document ids via idx_id.json ({"<row>": doc_id}), or via the
    memory-mapped sidecar written by gen_index_id_map (path ending in .ids): same result."""
    if mapping_path.endswith(".ids"):
        from .gen_index_id_map import SidecarMap
        idx_id = SidecarMap(mapping_path)
        return [[idx_id[int(i)] for i in row] for row in idxs]
    with open(mapping_path) as f:
        idx_id = json.load(f)
    return [[idx_id[str(int(i))] for i in row] for row in idxs]


def search(indexpath, query_embed, topk, chunk_rows=1 << 21, allow_rounding=False):
    """np.load + IndexFlatIP.add + search of the reference, on the GPU (index streamed from an mmap).

    fp16 .npy files (get_embed.py --fp16) are scanned as they are.  float32 files whose values fp16
    cannot hold are searched in exact-float32 mode (float32 copies of the rows in HBM, fp16 scan +
    exact re-scoring) unless allow_rounding asks for the rounded, faster variant."""
    from . import npy
    from .index import IndexFlatIP
    xq = npy.load(query_embed)
    info = npy.stat(indexpath)
    if info["cols"] != 128 or xq.shape[1] != 128:
        raise ValueError("embeddings must be 128-d")
    xb = npy.memmap(indexpath)
    index = IndexFlatIP(128, capacity=info["rows"])
    if allow_rounding:
        index.allow_rounding(True)
    for r0 in range(0, info["rows"], chunk_rows):
        index.add(np.ascontiguousarray(xb[r0:r0 + chunk_rows]))
    return index.search(xq, topk)


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
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    with open(args.raw_data) as f:
        qas = [json.loads(line) for line in f.readlines()]
    questions = [item["question"] for item in qas]
    answers = [item["answer"] for item in qas]

    # fork the scorer pool before any HIP call
    processes = ProcessPool(processes=args.num_workers, initializer=init, initargs=[args.db])
    try:
        D, I = search(args.indexpath, args.query_embed, args.topk, allow_rounding=args.allow_fp16_rounding)
        retrieval_results = convert_idx2id(I, args.idx_id_map)
        assert len(retrieval_results) == len(questions) == len(answers)
        results = processes.map(partial(get_score, topk=args.topk), zip(questions, answers, retrieval_results))
    finally:
        processes.close()
        processes.join()

    aggregate = defaultdict(list)
    for r in results:
        for key, value in r.items():
            aggregate[key].append(value)
    lines = []
    for key, values in aggregate.items():
        line = "Top {} Recall for {} QA pairs: {} ...".format(key, len(values), np.mean(values))
        print(line)
        lines.append(line)
    return lines


if __name__ == "__main__":
    main()
