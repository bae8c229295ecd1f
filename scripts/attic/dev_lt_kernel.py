import sys, torch
sys.path.insert(0, ".")
from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
dev = torch.device("cuda:0")
cfg = dict(BERT_BASE, num_hidden_layers=2)
m = BertForRetriever(cfg, device=dev); m.load_state_dict(random_state_dict(cfg, seed=0))
ids = torch.randint(1000, 30000, (512, 128), device=dev); mask = torch.ones((512, 128), dtype=torch.bool, device=dev)
print("before:", m.gemm_kernels())
m.get_embed({"input_ids": ids, "input_mask": mask}, False)
torch.cuda.synchronize()
print("after:", {k: v[:90] for k, v in m.gemm_kernels().items()})
