"""Where the first proqa_index_add_npy of a process spends its time: run per piece size in fresh processes.
usage: python scripts/dev_loader_timing.py [rows]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, os
sys.path.insert(0, %r)
t0 = time.perf_counter()
import numpy as np
from proqa_amd.index import IndexFlatIP
t1 = time.perf_counter()
path, n, readers = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ix = IndexFlatIP(128, capacity=n)
t2 = time.perf_counter()
ix.add_npy(path, readers=readers)
t3 = time.perf_counter()
ix2 = IndexFlatIP(128, capacity=n)
t4 = time.perf_counter()
ix2.add_npy(path, readers=readers)
t5 = time.perf_counter()
print("piece_mb", os.environ.get("PROQA_LOADER_PIECE_MB"), "readers", readers, "import %%.3f create %%.3f first_add %%.3f (%%.1f GB/s) create2 %%.3f second_add %%.3f (%%.1f GB/s)" %% (
    t1 - t0, t2 - t1, t3 - t2, n * 256 / (t3 - t2) / 1e9, t4 - t3, t5 - t4, n * 256 / (t5 - t4) / 1e9))
''' % ROOT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18_000_000
import numpy as np
path = "/tmp/dev_loader.npy"
with open(path, "wb") as f:
    np.lib.format.write_array_header_1_0(f, {"descr": "<f2", "fortran_order": False, "shape": (n, 128)})
    blk = np.random.default_rng(0).standard_normal((1_000_000, 128)).astype(np.float16)
    for r0 in range(0, n, 1_000_000):
        f.write(blk[:min(1_000_000, n - r0)].data)
for mb in (8,):   # (the piece size was a build-time experiment: PROQA_LOADER_PIECE_MB is gone, see profiles/r04_loader_timing.txt)
    for readers in (2, 4, 8):
        env = dict(os.environ, PROQA_LOADER_PIECE_MB=str(mb))
        subprocess.run([sys.executable, "-c", CHILD, path, str(n), str(readers)], env=env)
os.remove(path)
