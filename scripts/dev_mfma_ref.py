"""Bare-MFMA reference launches for the PMC passes (random operands, ~8 ms each): what the matrix pipe sustains on
this box with nothing else running -- the row next to mips_filter_f16 in profiles/r02_mfma_regime.json."""
import ctypes
import sys

sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from proqa_amd import _lib  # noqa: E402

lib = _lib.load()
v = ctypes.c_double()
for zero in ((0, 1) if len(sys.argv) < 2 else (int(sys.argv[1]),)):
    _lib.check(lib.proqa_microbench_mfma(8.0, zero, _lib.current_stream_ptr(), ctypes.byref(v)))
    print("zero_operands" if zero else "random_operands", v.value, "TFLOP/s")
