"""int8 MFMA rate of a loop shaped like mips_filter_i8's unit -- 8 MFMAs + n plain VALU instructions per wave, four waves per
SIMD -- over n (dev; MI355X).  proqa_microbench_mfma_i8_valu."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from proqa_amd import _lib  # noqa: E402

lib = _lib.load()
torch.zeros(1, device="cuda")
v = ctypes.c_double()
st = _lib.current_stream_ptr()
_lib.check(lib.proqa_microbench_mfma_i8(8.0, 0, st, ctypes.byref(v)))
print(f"bare loop (2 waves per SIMD, 4 chains): {v.value:.0f} TOP/s")
base = None
for rep in range(2):
    for n in (0, 8, 16, 24, 32, 48, 64):
        _lib.check(lib.proqa_microbench_mfma_i8_valu(8.0, n, st, ctypes.byref(v)))
        base = base or v.value
        # cycles of the SIMD's matrix issue one VALU instruction costs: per trip the four waves issue 4 x 8 MFMAs = 1024 matrix
        # cycles; time per trip scales with base / rate
        extra = 1024.0 * (base / v.value - 1.0)
        print(f"rep {rep}: 8 MFMAs + {n:2d} VALU per wave and trip: {v.value:7.0f} TOP/s ({v.value / base:.3f} of the VALU-free loop; "
              f"+{extra:6.0f} cycles per trip = {extra / (4 * n) if n else 0:.2f} cycles per VALU instruction)")
