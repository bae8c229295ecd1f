"""The assignment step alone (proqa_kmeans_assign_device) at group_paras.py's default shape: ms per call, the number of points
the nominating pass left undecided, and a digest of the labels (must not depend on PROQA_KMEANS_TWO_PASS)."""
import ctypes, hashlib, sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd import _lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    x[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
lib = _lib.load()
h = ctypes.c_void_p(); _lib.check(lib.proqa_kmeans_create(128, n, k, ctypes.byref(h)))
lab = torch.empty(n, dtype=torch.int32, device=dev); dist = torch.empty(n, dtype=torch.float32, device=dev)
for name, cen in (("centroids = random points", x[torch.randperm(n, generator=g, device=dev)[:k]].float().contiguous()),
                  (f"centroids = means of {n // k} random points", x[:k * (n // k)].float().view(k, n // k, 128).mean(1).contiguous())):
    f = lambda: _lib.check(lib.proqa_kmeans_assign_device(h, x.data_ptr(), n, cen.data_ptr(), 1, lab.data_ptr(), dist.data_ptr(), _lib.current_stream_ptr()))
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 5 * 1e3
    digest = hashlib.sha256(lab.cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"{name}: {ms:.1f} ms per assignment; labels {digest}; objective {dist.double().sum().item():.6e}")
    # hinted: the exact labels as the hint (a converged Lloyd loop), and labels of which 30 % are wrong
    for hname, hint in (("own labels", lab.clone()), ("30 % random", torch.where(torch.rand(n, device=dev) < 0.3, torch.randint(0, k, (n,), device=dev, dtype=torch.int32), lab))):
        lab2 = torch.empty_like(lab)
        fh = lambda: _lib.check(lib.proqa_kmeans_assign_hinted_device(h, x.data_ptr(), n, cen.data_ptr(), 1, hint.data_ptr(), lab2.data_ptr(), dist.data_ptr(), _lib.current_stream_ptr()))
        for _ in range(2): fh()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): fh()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 5 * 1e3
        print(f"   hint = {hname}: {ms:.1f} ms; labels {hashlib.sha256(lab2.cpu().numpy().tobytes()).hexdigest()[:10]}")
