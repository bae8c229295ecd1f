"""End-to-end latency of OnlineRetriever per question (encode + search + id map + row gather), 18M-row index."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from proqa_amd.online_retriever import OnlineRetriever
from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
xb = rng.standard_normal((n, 128), dtype=np.float32).astype(np.float16)
ids = [f"doc-{i}" for i in range(n)]
r = OnlineRetriever(xb, ids, device=dev)
model = BertForRetriever(BERT_BASE, device=dev); model.load_state_dict(random_state_dict(BERT_BASE, seed=0))
tok = torch.randint(1000, 30522, (1, 16), device=dev); mask = torch.ones((1, 16), dtype=torch.bool, device=dev)
for k in (5, 80, 5000):
    for _ in range(3):
        q = model.get_embed({"input_ids": tok, "input_mask": mask}, True, check_mask=False, seq_lens_host=[16])["embed"]
        r.retrieve(q, k)
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 20
    te = 0.0
    for _ in range(reps):
        t1 = time.perf_counter()
        q = model.get_embed({"input_ids": tok, "input_mask": mask}, True, check_mask=False, seq_lens_host=[16])["embed"]
        torch.cuda.synchronize(); te += time.perf_counter() - t1
        out = r.retrieve(q, k)
    dt = (time.perf_counter() - t0) / reps
    print(f"k={k}: {dt*1e3:.3f} ms per question (encode {te/reps*1e3:.3f} ms, retrieve {(dt - te/reps)*1e3:.3f} ms)")
