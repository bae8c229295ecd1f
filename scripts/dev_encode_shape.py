"""One encode shape on its own (dev; for rocprofv3 --kernel-trace --stats): usage dev_encode_shape.py [batch] [seq_len] [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
model = BertForRetriever(BERT_BASE, device=dev)
model.load_state_dict(random_state_dict(BERT_BASE, seed=0))
g = torch.Generator(device=dev).manual_seed(1)
ids = torch.randint(1000, 30522, (B, S), generator=g, device=dev)
batch = {"input_ids": ids, "input_mask": torch.ones_like(ids, dtype=torch.bool)}
for _ in range(2):
    model.get_embed(batch, False, check_mask=False, seq_lens_host=[S] * B)
torch.cuda.synchronize()
t = time.time()
for _ in range(steps):
    model.get_embed(batch, False, check_mask=False, seq_lens_host=[S] * B)
torch.cuda.synchronize()
dt = (time.time() - t) / steps
print(f"{B} x {S}: {dt * 1e3:.3f} ms per step, {B / dt:.0f} passages/s")
