"""MS MARCO (TREC 2019) preprocessing and the top-10000 retrieval that labels the training queries.

Drop-in for /root/reference/retrieval/trec_process.py: the same three functions with the same arguments,
file formats and printed line.  `retrieve_topk` (:69-94) is the caller of the LARGE-k search: np.load of
both .npy files, faiss.IndexFlatIP.search(xq, 10000), then per query the list of passage rows and the
0/1 label of each (row in the query's `labels`).  The search runs on the MI355X (the one-pass large-k path
of libproqa_hip.so); there is no CPU path.  The labelling, a Python `in` over a list per
retrieved row in the reference (70 M list scans for the 6980 dev queries), is one vectorised membership test
per query here -- same output.
"""
import json
from collections import defaultdict

import numpy as np

TOPK = 10000   # hard-coded in the reference (:76)


def prepare_corpus(path="../data/trec-2019/collection.tsv", save_path="../data/trec-2019/msmarco_paras.txt"):
    """collection.tsv (`pid<TAB>text`) -> JSON-lines {"text", "id"} in file order (reference :8-17).
    Like the reference, a line that does not split into exactly two fields is an error."""
    with open(path) as f, open(save_path, "w") as g:
        for line in f:
            pid, text = line.strip().split("\t")
            g.write(json.dumps({"text": text, "id": int(pid)}) + "\n")


def extract_labels(input="../data/trec-2019/qrels.train.tsv", output="../data/trec-2019/msmacro-train.txt",
                   queries="../data/trec-2019/queries.train.tsv"):
    """qrels (`qid 0 pid 1`) + queries (`qid<TAB>text`) -> JSON-lines {"question", "labels", "qid"}, one per query
    that has a relevant passage, in order of first appearance in the qrels; a trailing '?' is dropped from the
    question (reference :19-46).  Prints the two counts the reference prints."""
    qid2query = {}
    with open(queries) as f:
        for line in f:
            fields = line.strip().split("\t")
            q = fields[1]
            if q.endswith("?"):
                q = q[:-1]
            qid2query[int(fields[0])] = q
    print(len(qid2query))

    qid2ground = defaultdict(list)
    with open(input) as f:
        for line in f:
            fields = line.strip().split("\t")
            qid2ground[int(fields[0])].append(int(fields[2]))
    print(len(qid2ground))

    with open(output, "w") as g:
        for qid, labels in qid2ground.items():
            g.write(json.dumps({"question": qid2query[qid], "labels": labels, "qid": qid}) + "\n")


def label_rows(I, labels):
    """para_labels of the reference (:85): 1 where the retrieved row is one of the query's labels.
    Slots the search left empty (I = -1: fewer rows than k) are no label's row."""
    return np.isin(I, np.asarray(labels, dtype=np.int64)).astype(np.int64)


def retrieve_topk(index_path="../data/trec-2019/embeds/msmarco_paras_embed.npy",
                  query_embeds="../data/trec-2019/embeds/msmarco-train-query.npy",
                  query_input="../data/trec-2019/msmacro-train.txt", output="../data/trec-2019/processed/train.txt",
                  topk=TOPK, allow_rounding=False):
    """Top-`topk` passages of every query, written as the query's JSON line + "para_embed_idx" (the rows, best first)
    and "para_labels" (0/1 per row); prints `Avg recall: {fraction of queries with a label among their rows}`.
    Returns that fraction (the reference returns None)."""
    from .eval_retrieval import dist_env, finish_distributed, search
    _D, I = search(index_path, query_embeds, topk, allow_rounding=allow_rounding)
    finish_distributed()
    if dist_env()[1] != 0:      # under torchrun the corpus is row-sharded over the ranks; rank 0 labels and writes
        return None
    with open(query_input) as f:
        raw_data = [json.loads(line) for line in f]
    if len(raw_data) < I.shape[0]:
        raise IndexError(f"{I.shape[0]} query embeddings but {len(raw_data)} lines in {query_input}")   # reference: raw_data[idx]

    covered = []
    with open(output, "w") as g:
        for idx in range(I.shape[0]):
            sample = raw_data[idx]
            rows = I[idx]
            labels = label_rows(rows, sample["labels"])
            sample["para_embed_idx"] = rows.tolist()
            sample["para_labels"] = labels.tolist()
            covered.append(int(labels.sum() > 0))
            g.write(json.dumps(sample) + "\n")
    recall = float(np.mean(covered))
    print(f"Avg recall: {recall}")
    return recall


if __name__ == "__main__":
    retrieve_topk()
