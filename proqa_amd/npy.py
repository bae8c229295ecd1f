"""NumPy-facing wrappers of the C-ABI .npy reader/writer (proqa_npy_* in include/proqa_hip.h).

File format contract: /root/reference/retrieval/get_embed.py:139 (np.save) and
retrieval/eval_retrieval.py:99-100 (np.load) — 2-D C-order '<f2' / '<f4', format v1.0.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import PROQA_F16, PROQA_F32

_NP = {PROQA_F16: np.float16, PROQA_F32: np.float32}


def _code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float16:
        return PROQA_F16
    if dtype == np.float32:
        return PROQA_F32
    raise TypeError(f"only float16/float32 embedding files are supported, got {dtype}")


def save_path(path):
    """np.save appends '.npy' when the name lacks it; keep that behaviour."""
    return path if path.endswith(".npy") else path + ".npy"


def stat(path):
    info = _lib.NpyInfo()
    _lib.check(_lib.load().proqa_npy_stat(path.encode(), ctypes.byref(info)))
    return {"rows": info.rows, "cols": info.cols, "dtype": _NP[info.dtype], "data_offset": info.data_offset}


def load(path, row0=0, nrows=None):
    """Read rows [row0, row0+nrows) of a .npy embedding matrix through the C reader."""
    info = stat(path)
    if nrows is None:
        nrows = info["rows"] - row0
    out = np.empty((nrows, info["cols"]), dtype=info["dtype"])
    _lib.check(_lib.load().proqa_npy_read_rows(path.encode(), row0, nrows, out.ctypes.data, out.nbytes))
    return out


def memmap(path):
    """Zero-copy view of the data region (for chunked upload of a multi-GB index)."""
    info = stat(path)
    return np.memmap(path, dtype=info["dtype"], mode="r", offset=info["data_offset"],
                     shape=(info["rows"], info["cols"]))


def save(path, array):
    array = np.ascontiguousarray(array)
    if array.ndim != 2:
        raise ValueError("embedding matrices are 2-D")
    path = save_path(path)
    _lib.check(_lib.load().proqa_npy_write(path.encode(), array.ctypes.data, array.shape[0], array.shape[1],
                                           _code(array.dtype)))
    return path


def create(path, rows, cols, dtype):
    path = save_path(path)
    _lib.check(_lib.load().proqa_npy_create(path.encode(), rows, cols, _code(dtype)))
    return path


def write_rows(path, row0, array):
    array = np.ascontiguousarray(array)
    if array.ndim != 2:
        raise ValueError("embedding matrices are 2-D")
    _lib.check(_lib.load().proqa_npy_write_rows(path.encode(), row0, array.shape[0], array.ctypes.data,
                                                array.shape[1], _code(array.dtype)))
