"""A/B timing of the attention kernel with and without the folded projection bias (same process)."""
import sys
import torch
sys.path.insert(0, ".")
from proqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
B, S, NH = 512, 128, 12
H = NH * 64
qkv = torch.randn((B * S, 3 * H), device=dev).half()
bias = torch.randn((3 * H,), device=dev).half()
lens = torch.full((B,), S, dtype=torch.int32, device=dev)
out = torch.empty((B * S, H), dtype=torch.float16, device=dev)
st = torch.cuda.current_stream().cuda_stream
for name, bp in (("no bias", None), ("bias", bias.data_ptr()), ("no bias", None), ("bias", bias.data_ptr())):
    for _ in range(5):
        lib.proqa_attention_ex_f16(qkv.data_ptr(), bp, lens.data_ptr(), None, B, S, NH, 0, out.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.proqa_attention_ex_f16(qkv.data_ptr(), bp, lens.data_ptr(), None, B, S, NH, 0, out.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/50*1e3:.1f} us")
