import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.arange(10, dtype=torch.uint8, device="cuda")
out = torch.empty((1, 10), dtype=torch.uint8, device="cuda")
dist.all_gather_into_tensor(out, x)
torch.cuda.synchronize()
print("rccl all_gather_into_tensor ok", out.tolist(), dist.get_backend())
dist.barrier(); dist.destroy_process_group()
