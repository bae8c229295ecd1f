"""Encoder latency/throughput vs batch size (bert-base shape, random weights)."""
import sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
dev = torch.device("cuda", 0)
model = BertForRetriever(BERT_BASE, device=dev); model.load_state_dict(random_state_dict(BERT_BASE, seed=0))
for B, S in [(1, 16), (1, 128), (4, 32), (8, 32), (16, 32), (32, 32), (32, 128), (128, 128), (512, 128)]:
    ids = torch.randint(1000, 30522, (B, S), device=dev); mask = torch.ones((B, S), dtype=torch.bool, device=dev)
    batch = {"input_ids": ids, "input_mask": mask}
    lens = [S] * B
    for _ in range(3): model.get_embed(batch, True, check_mask=False, seq_lens_host=lens)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 50
    for _ in range(n): model.get_embed(batch, True, check_mask=False, seq_lens_host=lens)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"B={B:4d} S={S:4d}: {dt*1e3:8.3f} ms per batch  {B/dt:9.0f} seq/s")
