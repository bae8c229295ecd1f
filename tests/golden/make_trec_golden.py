"""Golden vectors for proqa_amd/trec_process.py, made by running the REFERENCE's trec_process.py.

Run in the build container only (needs /root/reference):   python tests/golden/make_trec_golden.py

The reference imports faiss (absent here); `retrieve_topk` is run with a stand-in module whose IndexFlatIP is the
NumPy oracle (exact scores, score descending / row ascending) -- what the golden pins is everything AROUND the
search: file formats, the order and content of the per-query lists, the labels, the printed line.  Inputs are
regenerated from seeds by the test (tests/test_trec_process*.py: `trec_inputs`); the outputs are committed as text
(the two preprocessing files) or as SHA-256 + a few leading values (the 10000-wide lists).
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/retrieval"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from trec_inputs import trec_inputs  # noqa: E402  (seeded inputs shared with the tests)

COLLECTION = "0\tThe Eiffel Tower is in Paris.\n1\tA passage with \"quotes\" and a back\\slash.\n7\tÜmlauts and 日本語 text.\n" \
             "3\tThe last line has trailing spaces.   \n"
QUERIES = "11\twhere is the eiffel tower?\n5\twhat is a passage\n42\tunused query?\n8\twho wrote it??\n"
QRELS = "11\t0\t0\t1\n5\t0\t1\t1\n11\t0\t3\t1\n8\t0\t7\t1\n"


def main():
    from oracle import search_oracle

    class IndexFlatIP:   # the stand-in for the absent library
        def __init__(self, d):
            self.xb = np.zeros((0, d), np.float32)

        def add(self, xb):
            self.xb = np.concatenate([self.xb, xb])

        def search(self, xq, k):
            return search_oracle.topk_ip(xq, self.xb, k)

    faiss = types.ModuleType("faiss")
    faiss.IndexFlatIP = IndexFlatIP
    sys.modules["faiss"] = faiss
    sys.path.insert(0, REF)
    import trec_process as ref
    sys.path.remove(REF)

    out = {"collection_tsv": COLLECTION, "queries_tsv": QUERIES, "qrels_tsv": QRELS}
    with tempfile.TemporaryDirectory() as tmp:
        for name, text in (("collection.tsv", COLLECTION), ("queries.tsv", QUERIES), ("qrels.tsv", QRELS)):
            with open(os.path.join(tmp, name), "w") as f:
                f.write(text)
        ref.prepare_corpus(os.path.join(tmp, "collection.tsv"), os.path.join(tmp, "paras.txt"))
        out["prepare_corpus"] = open(os.path.join(tmp, "paras.txt")).read()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref.extract_labels(input=os.path.join(tmp, "qrels.tsv"), output=os.path.join(tmp, "train.txt"),
                               queries=os.path.join(tmp, "queries.tsv"))
        out["extract_labels"] = open(os.path.join(tmp, "train.txt")).read()
        out["extract_labels_stdout"] = buf.getvalue()

        paras, queries, qfile = trec_inputs(tmp)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref.retrieve_topk(index_path=paras, query_embeds=queries, query_input=qfile, output=os.path.join(tmp, "processed.txt"))
        blob = open(os.path.join(tmp, "processed.txt"), "rb").read()
        first = json.loads(blob.split(b"\n")[0])
        out["retrieve_topk"] = {"n": 12000, "nq": 12, "seed": 77, "stdout": buf.getvalue(),
                                "output_sha256": hashlib.sha256(blob).hexdigest(), "output_bytes": len(blob),
                                "first_keys": list(first.keys()), "first_rows": first["para_embed_idx"][:16],
                                "first_label_sum": int(sum(first["para_labels"]))}
    with open(os.path.join(HERE, "trec_golden.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("trec golden ok:", out["retrieve_topk"]["stdout"].strip(), out["retrieve_topk"]["output_bytes"], "bytes")


if __name__ == "__main__":
    main()
