"""The handle's library dense layer (pinned hipBLASLt kernel where it applies) against an fp32 product over a grid of shapes."""
import sys, torch
sys.path.insert(0, ".")
from proqa_amd import _lib
from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
dev = torch.device("cuda:0")
cfg = dict(BERT_BASE, num_hidden_layers=1)
m = BertForRetriever(cfg, device=dev); m.load_state_dict(random_state_dict(cfg, seed=0))
lib = _lib.load()
h = m.towers[False]._handle
g = torch.Generator(device=dev).manual_seed(0)
for M in (4096, 4352, 6400, 7936, 65536):
    for N in (256, 768, 2304, 3072):
        for K in (64, 128, 192, 256, 320, 768, 3072):
            x = torch.randn((M, K), generator=g, device=dev).half(); w = (torch.randn((N, K), generator=g, device=dev) * 0.05).half()
            out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
            _lib.check(lib.proqa_encoder_dense(h, x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, _lib.current_stream_ptr()))
            torch.cuda.synchronize()
            rows = torch.randint(0, M, (256,), device=dev)
            ref = x[rows].float() @ w.float().t()
            err = (out[rows].float() - ref).abs().max().item()
            bad = err > 2e-3 * max(1.0, ref.abs().max().item()) or not torch.isfinite(out).all()
            if bad: print(f"M={M} N={N} K={K}: max abs err {err:.4g}  BAD")
print("pinned:", m.gemm_kernels()[False][:80], "-- grid done")
