"""What a grid-wide barrier costs on this box (proqa_microbench_grid_sync): one cooperative launch of G workgroups with N
barriers, against N + 1 dependent ordinary launches (dev; MI355X)."""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from proqa_amd import _lib  # noqa: E402

lib = _lib.load()
torch.zeros(1, device="cuda")
v = ctypes.c_double()
st = _lib.current_stream_ptr()
for grid in (24, 96, 256):
    row = []
    for n in (0, 1, 10, 84, 168):
        _lib.check(lib.proqa_microbench_grid_sync(grid, n, st, ctypes.byref(v)))
        row.append((n, v.value))
    per = (row[-1][1] - row[1][1]) / (row[-1][0] - row[1][0])
    print(f"grid {grid:3d}: " + "  ".join(f"{n} syncs {us:7.1f} us" for n, us in row) + f"   -> {per:.2f} us per barrier, launch {row[0][1]:.1f} us")
# dependent ordinary launches for comparison
x = torch.zeros(1024, device="cuda")
for _ in range(10):
    x.add_(1)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(900):
    x.add_(1)
torch.cuda.synchronize()
print(f"900 dependent tiny torch kernels: {(time.perf_counter() - t) / 900 * 1e6:.2f} us each")
