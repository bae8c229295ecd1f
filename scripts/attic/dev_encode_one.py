"""One question at a time through the encoder (bert-base shape, random weights): wall time per call, for rocprofv3."""
import sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
dev = torch.device("cuda", 0)
B, S, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
model = BertForRetriever(BERT_BASE, device=dev); model.load_state_dict(random_state_dict(BERT_BASE, seed=0))
ids = torch.randint(1000, 30522, (B, S), device=dev); mask = torch.ones((B, S), dtype=torch.bool, device=dev)
batch = {"input_ids": ids, "input_mask": mask}
for _ in range(5): model.get_embed(batch, True, check_mask=False, seq_lens_host=[S] * B)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): model.get_embed(batch, True, check_mask=False, seq_lens_host=[S] * B)
t_issue = (time.perf_counter() - t0) / n
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"B={B} S={S}: {dt*1e3:.3f} ms per call (host issue {t_issue*1e3:.3f} ms)")
