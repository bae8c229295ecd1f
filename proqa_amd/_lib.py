"""ctypes binding of libproqa_hip.so (C ABI declared in include/proqa_hip.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is visible the
calls raise.  PyTorch is used only as the owner of device memory and streams; raw device
pointers (tensor.data_ptr()) and the current stream handle cross the boundary.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libproqa_hip.so")

PROQA_F16 = 0
PROQA_F32 = 1
EMBED_DIM = 128

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t
c_char_p = ctypes.c_char_p


class ProqaError(RuntimeError):
    """A libproqa_hip call returned a negative status."""

    def __init__(self, code, message):
        super().__init__(f"libproqa_hip error {code}: {message}")
        self.code = code


class SearchStats(ctypes.Structure):
    _fields_ = [("rounds", ctypes.c_int32), ("fallback_rounds", ctypes.c_int32),
                ("candidates", ctypes.c_int64), ("filter_ms", c_float), ("total_ms", c_float),
                ("nominated", ctypes.c_int64), ("nomination", ctypes.c_int32), ("nomination_state", ctypes.c_int32),
                ("leap_rank", ctypes.c_int32), ("leap_state", ctypes.c_int32)]


class BertLayer(ctypes.Structure):
    """proqa_bert_layer: device fp16 pointers of one encoder layer."""
    _fields_ = [(n, c_void_p) for n in ("qkv_w", "qkv_b", "ao_w", "ao_b", "ln1_g", "ln1_b", "ff1_w", "ff1_b",
                                        "ff2_w", "ff2_b", "ln2_g", "ln2_b")]


class BertWeights(ctypes.Structure):
    """proqa_bert_weights"""
    _fields_ = [("hidden", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("n_heads", ctypes.c_int32),
                ("intermediate", ctypes.c_int32), ("max_position", ctypes.c_int32), ("vocab", c_int64),
                ("layer_norm_eps", c_float),
                ("word_emb", c_void_p), ("pos_emb", c_void_p), ("type_emb", c_void_p),
                ("emb_ln_g", c_void_p), ("emb_ln_b", c_void_p),
                ("layers", ctypes.POINTER(BertLayer)),
                ("pool_w", c_void_p), ("pool_b", c_void_p), ("proj_w", c_void_p), ("proj_b", c_void_p)]


ENC_CLS_ONLY_LAST = 1
ENC_PACKED = 2


class NpyInfo(ctypes.Structure):
    _fields_ = [("rows", c_int64), ("cols", c_int64), ("dtype", ctypes.c_int32),
                ("data_offset", c_int64)]


# name -> (restype, argtypes); mirrors include/proqa_hip.h one to one
SIGNATURES = {
    "proqa_last_error": (c_char_p, []),
    "proqa_abi_version": (c_int, []),
    "proqa_device_info": (c_int, [ctypes.POINTER(c_int), c_char_p, c_size_t]),
    "proqa_index_create": (c_int, [c_int, c_int64, ctypes.POINTER(c_void_p)]),
    "proqa_index_add": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "proqa_index_add_npy": (c_int, [c_void_p, c_char_p, c_int64, c_int64, c_int]),
    "proqa_index_add_device": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "proqa_index_adopt_device": (c_int, [c_void_p, c_void_p, c_int64]),
    "proqa_index_rows_changed": (c_int, [c_void_p]),
    "proqa_index_prepare": (c_int, [c_void_p, c_void_p]),
    "proqa_index_allow_rounding": (c_int, [c_void_p, c_int]),
    "proqa_index_is_exact_f32": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "proqa_index_ntotal": (c_int, [c_void_p, ctypes.POINTER(c_int64)]),
    "proqa_index_reset": (c_int, [c_void_p]),
    "proqa_index_free": (c_int, [c_void_p]),
    "proqa_index_search": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "proqa_index_search_device": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int64,
                                          c_void_p, c_void_p, c_void_p]),
    "proqa_index_search_begin_device": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int64,
                                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "proqa_index_search_finish": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "proqa_index_reconstruct_batch_device": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    "proqa_index_last_stats": (c_int, [c_void_p, ctypes.POINTER(SearchStats)]),
    "proqa_index_set_profiling": (c_int, [c_void_p, c_int]),
    "proqa_index_configure": (c_int, [c_void_p, c_int, c_int]),
    "proqa_index_configure_nomination": (c_int, [c_void_p, c_int]),
    "proqa_index_configure_leap": (c_int, [c_void_p, c_int]),
    "proqa_leap_plan": (c_int, [c_int64, c_int64, c_int, c_int64, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "proqa_index_configure_bootstrap": (c_int, [c_void_p, c_int]),
    "proqa_topk_merge_device": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p,
                                        c_void_p, c_void_p]),
    "proqa_topk_merge_strided_device": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_int64, c_void_p,
                                                c_void_p, c_void_p]),
    "proqa_encoder_create": (c_int, [ctypes.POINTER(BertWeights), ctypes.POINTER(c_void_p)]),
    "proqa_encoder_free": (c_int, [c_void_p]),
    "proqa_encoder_set_gemm_tuning": (c_int, [c_void_p, c_int]),
    "proqa_encoder_gemm_kernel": (c_int, [c_void_p, c_char_p, c_size_t]),
    "proqa_encoder_workspace": (c_int, [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t)]),
    "proqa_encoder_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "proqa_encoder_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_int,
                                      c_void_p]),
    "proqa_embed_layernorm_f16": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int64,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                          c_void_p, c_void_p]),
    "proqa_embed_layernorm_varlen_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64,
                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                                 c_void_p, c_void_p]),
    "proqa_attention_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "proqa_attention_cls_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "proqa_attention_varlen_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "proqa_attention_cls_varlen_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "proqa_attention_ex_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                       c_void_p]),
    "proqa_gemm_tn_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "proqa_bias_gelu_f16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "proqa_bias_residual_layernorm_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  c_float, c_int64, c_int, c_void_p, c_void_p]),
    "proqa_pool_project_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "proqa_kmeans_create": (c_int, [c_int, c_int64, c_int, ctypes.POINTER(c_void_p)]),
    "proqa_kmeans_free": (c_int, [c_void_p]),
    "proqa_kmeans_assign_device": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                           c_void_p]),
    "proqa_kmeans_assign_hinted_device": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                  c_void_p]),
    "proqa_kmeans_update_device": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "proqa_rand_perm": (c_int, [c_int64, c_int64, c_void_p]),
    "proqa_npy_stat": (c_int, [c_char_p, ctypes.POINTER(NpyInfo)]),
    "proqa_npy_read_rows": (c_int, [c_char_p, c_int64, c_int64, c_void_p, c_size_t]),
    "proqa_npy_write": (c_int, [c_char_p, c_void_p, c_int64, c_int64, c_int]),
    "proqa_npy_create": (c_int, [c_char_p, c_int64, c_int64, c_int]),
    "proqa_npy_write_rows": (c_int, [c_char_p, c_int64, c_int64, c_void_p, c_int64, c_int]),
    "proqa_wordpiece_create": (c_int, [c_char_p, c_size_t, c_int, ctypes.POINTER(c_void_p)]),
    "proqa_wordpiece_free": (c_int, [c_void_p]),
    "proqa_wordpiece_encode_batch": (c_int, [c_void_p, ctypes.POINTER(c_char_p), ctypes.POINTER(c_int64), c_int64, c_int,
                                             c_void_p, c_void_p, c_int]),
    "proqa_wordpiece_encode_jsonl_batch": (c_int, [c_void_p, ctypes.POINTER(c_char_p), ctypes.POINTER(c_int64), c_int64,
                                                   c_char_p, c_int, c_void_p, c_void_p, c_int]),
    "proqa_comm_get_unique_id": (c_int, [c_void_p]),
    "proqa_comm_create": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "proqa_comm_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "proqa_comm_free": (c_int, [c_void_p]),
    "proqa_sharded_search_device": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p,
                                            c_void_p, c_void_p]),
    "proqa_sharded_block_layout": (c_int, [c_int64, c_int, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t),
                                           ctypes.POINTER(c_size_t)]),
    "proqa_topk_merge_gathered_device": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "proqa_microbench_stream": (c_int, [c_void_p, c_size_t, c_int, c_int, c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "proqa_microbench_mfma": (c_int, [ctypes.c_double, c_int, c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "proqa_microbench_mfma_i8": (c_int, [ctypes.c_double, c_int, c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "proqa_microbench_mfma_i8_shape": (c_int, [ctypes.c_double, c_int, c_int, c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "proqa_microbench_mfma_i8_valu": (c_int, [ctypes.c_double, c_int, c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "proqa_microbench_grid_sync": (c_int, [c_int, c_int, c_void_p, ctypes.POINTER(ctypes.c_double)]),
}

_lock = threading.Lock()
_lib = None


def _promote_hip_runtime():
    """Make ONE HIP runtime's (and its rocBLAS') symbols global before libproqa_hip.so is opened.

    libproqa_hip.so carries no DT_NEEDED on libamdhip64 / librocblas (see proqa_amd/build.py).  When torch is
    importable its bundled runtime is the one that owns every tensor we are handed, so that copy
    is promoted; otherwise the system ROCm runtime is used.
    """
    dirs = []
    try:
        # torch's bundled runtime, located WITHOUT importing torch (1-2 s the eval_retrieval.py command line does not need);
        # if torch is imported later in the process it finds the same libraries already loaded
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            dirs.append(os.path.join(list(spec.submodule_search_locations)[0], "lib"))
    except Exception:
        pass
    dirs += ["", "/opt/rocm/lib"]
    errors = []
    for d in dirs:
        hip = os.path.join(d, "libamdhip64.so") if d else "libamdhip64.so"
        if d and not os.path.exists(hip):
            continue
        try:
            handle = ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
            # the encoder's GEMMs: the rocBLAS that belongs to THIS runtime (same directory)
            ctypes.CDLL(os.path.join(d, "librocblas.so") if d else "librocblas.so", mode=ctypes.RTLD_GLOBAL)
            return handle
        except OSError as e:  # pragma: no cover - depends on the machine
            errors.append(f"{hip}: {e}")
    raise RuntimeError("no HIP runtime (libamdhip64.so + librocblas.so) could be loaded: " + "; ".join(errors))


def load():
    """Return the bound library; raises if it has not been built (no silent fallback)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            # not a fallback: the same hipcc build __graft_entry__.build() runs, for a checkout that was never built
            try:
                from . import build as _build
                _build.build()
            except Exception as e:
                raise RuntimeError(
                    f"{LIB_PATH} is missing and could not be built ({e}): build it with `python -m proqa_amd.build` "
                    "(or __graft_entry__.build()); proqa_amd has no CPU fallback") from e
        _promote_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI drifted
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.proqa_abi_version() != 7:
            raise RuntimeError("libproqa_hip.so ABI version mismatch; rebuild it")
        _lib = lib
        return lib


def check(status):
    if status != 0:
        msg = load().proqa_last_error()
        raise ProqaError(status, msg.decode("utf-8", "replace") if msg else "")
    return status


def current_stream_ptr():
    """hipStream_t of torch's current stream on the current device, as an int."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def device_info():
    lib = load()
    n = c_int(0)
    buf = ctypes.create_string_buffer(256)
    check(lib.proqa_device_info(ctypes.byref(n), buf, 256))
    return n.value, buf.value.decode()


def require_gpu():
    """Raise unless a gfx950 device is visible; called by every product entry point."""
    n, arch = device_info()
    if not arch.startswith("gfx950"):
        raise RuntimeError(f"proqa_amd targets MI355X (gfx950); found '{arch}'")
    return n, arch
