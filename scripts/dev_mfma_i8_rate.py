"""Gate 0(b) of the int8 nomination scan: sustained rate of v_mfma_i32_32x32x32_i8 against v_mfma_f32_32x32x16_f16 in the
same register-resident loop on the same box (random and all-zero operands, 2 / 8 / 20 ms launches)."""
import ctypes

import torch

from proqa_amd import _lib

lib = _lib.load()
torch.cuda.init()
st = _lib.current_stream_ptr()
for ms in (2.0, 8.0, 20.0):
    for zero in (0, 1):
        f = ctypes.c_double()
        i = ctypes.c_double()
        _lib.check(lib.proqa_microbench_mfma(ms, zero, st, ctypes.byref(f)))
        _lib.check(lib.proqa_microbench_mfma_i8(ms, zero, st, ctypes.byref(i)))
        print(f"launch ~{ms:4.1f} ms, {'zero  ' if zero else 'random'} operands: fp16 {f.value:7.1f} TFLOP/s   int8 {i.value:7.1f} TOP/s   ratio {i.value / f.value:.3f}")
