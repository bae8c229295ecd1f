"""Seeded inputs of the trec_process.retrieve_topk golden (shared by tests/golden/make_trec_golden.py and the tests)."""
import json
import os

import numpy as np


def trec_inputs(tmp, n=12000, nq=12, seed=77):
    """Seeded embeddings + query file for retrieve_topk: integer-valued fp16 (exact scores, many ties)."""
    rng = np.random.default_rng(seed)
    xb = rng.integers(-3, 4, (n, 128)).astype(np.float16)
    xq = rng.integers(-3, 4, (nq, 128)).astype(np.float16)
    np.save(os.path.join(tmp, "paras.npy"), xb)
    np.save(os.path.join(tmp, "queries.npy"), xq)
    with open(os.path.join(tmp, "queries.txt"), "w") as f:
        for q in range(nq):
            labels = [] if q % 5 == 4 else [int(v) for v in rng.integers(0, n, 1 + q % 3)]
            f.write(json.dumps({"question": f"question {q}", "labels": labels, "qid": 1000 + q}) + "\n")
    return os.path.join(tmp, "paras.npy"), os.path.join(tmp, "queries.npy"), os.path.join(tmp, "queries.txt")
