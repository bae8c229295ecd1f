"""dev: latency of encoding one question (bert-base shape) with and without the captured forward (PROQA_ENCODER_GRAPH)."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
dev = torch.device("cuda:0")
model = BertForRetriever(BERT_BASE, device=dev); model.load_state_dict(random_state_dict(BERT_BASE, seed=0))
for B, L in ((1, 16), (1, 30), (4, 32), (1, 128), (8, 32)):
    tok = torch.randint(1000, 30522, (B, L), device=dev); mask = torch.ones((B, L), dtype=torch.bool, device=dev)
    for _ in range(5): model.get_embed({"input_ids": tok, "input_mask": mask}, True, check_mask=False, seq_lens_host=[L] * B)
    torch.cuda.synchronize(); ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        q = model.get_embed({"input_ids": tok, "input_mask": mask}, True, check_mask=False, seq_lens_host=[L] * B)["embed"]
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"graph={os.environ.get('PROQA_ENCODER_GRAPH','1')} B={B} L={L}: median {sorted(ts)[25]*1e3:.3f} ms  sum {q.float().sum().item():.4f}")
