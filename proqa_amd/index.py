"""Exact inner-product index on MI355X, with the call shape of faiss.IndexFlatIP.

The reference searches with

    index = faiss.IndexFlatIP(d); index.add(xb); D, I = index.search(xq, k)

(/root/reference/retrieval/eval_retrieval.py:102-104, also retrieval/trec_process.py:74-76).
`IndexFlatIP` here keeps that shape (numpy in, numpy out) over libproqa_hip.so;
`ShardedIndexFlatIP` row-shards the corpus over the ranks of a torch.distributed group (one
process per GPU, RCCL over xGMI), all-gathers the per-shard (score, id) lists once and merges
them on the GPU.  Neither class has a CPU path.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import PROQA_F16, PROQA_F32, EMBED_DIM


def _np_dtype_code(arr):
    if arr.dtype == np.float16:
        return PROQA_F16
    if arr.dtype == np.float32:
        return PROQA_F32
    raise TypeError(f"embeddings must be float16 or float32, got {arr.dtype}")


def _torch_dtype_code(t):
    import torch
    if t.dtype == torch.float16:
        return PROQA_F16
    if t.dtype == torch.float32:
        return PROQA_F32
    raise TypeError(f"embeddings must be float16 or float32, got {t.dtype}")


def _as_matrix(x, d, what):
    x = np.ascontiguousarray(x)
    if x.ndim != 2 or x.shape[1] != d:
        raise ValueError(f"{what} must have shape [n, {d}], got {x.shape}")
    return x


QUERY_BATCH = 16384   # queries per library call: bounds the per-search candidate store in HBM


class IndexFlatIP:
    """Brute-force maximum-inner-product index; rows live in HBM as fp16."""

    def __init__(self, d=EMBED_DIM, capacity=0):
        self._lib = _lib.load()
        _lib.require_gpu()
        self.d = int(d)
        handle = ctypes.c_void_p()
        _lib.check(self._lib.proqa_index_create(self.d, int(capacity), ctypes.byref(handle)))
        self._h = handle
        self._adopted = None  # keeps an adopted tensor alive

    # -- faiss-shaped API ---------------------------------------------------------------
    @property
    def ntotal(self):
        n = ctypes.c_int64()
        _lib.check(self._lib.proqa_index_ntotal(self._h, ctypes.byref(n)))
        return n.value

    def add(self, xb):
        """Append rows.  numpy [n,d] float16/float32 (host) or a CUDA torch tensor."""
        if _is_torch(xb):
            return self.add_device(xb)
        xb = _as_matrix(xb, self.d, "xb")
        _lib.check(self._lib.proqa_index_add(self._h, xb.ctypes.data, xb.shape[0], _np_dtype_code(xb)))

    def add_npy(self, path, row0=0, n=-1, readers=0):
        """np.load(path)[row0:row0+n] + add, streamed by the library (reader threads -> pinned ring -> HBM; n < 0: to the
        last row).  The file's dtype ('<f2' / '<f4') follows the precision rules of add()."""
        _lib.check(self._lib.proqa_index_add_npy(self._h, str(path).encode(), int(row0), int(n), int(readers)))

    def search(self, xq, k):
        """(D float32 [nq,k], I int64 [nq,k]); scores descending, ties by ascending row."""
        if _is_torch(xq):
            D, I = self.search_device(xq, k)
            return D.cpu().numpy(), I.cpu().numpy()
        xq = _as_matrix(xq, self.d, "xq")
        nq = xq.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        for q0 in range(0, max(nq, 1), QUERY_BATCH):
            part = xq[q0:q0 + QUERY_BATCH]
            _lib.check(self._lib.proqa_index_search(self._h, part.ctypes.data, part.shape[0], _np_dtype_code(xq),
                                                    int(k), D[q0:].ctypes.data, I[q0:].ctypes.data))
        return D, I

    def reset(self):
        _lib.check(self._lib.proqa_index_reset(self._h))
        self._adopted = None

    # -- device-resident variants -------------------------------------------------------
    def add_device(self, xb):
        import torch
        if not xb.is_cuda:
            raise ValueError("add_device expects a CUDA tensor")
        xb = xb.contiguous()
        if xb.dim() != 2 or xb.shape[1] != self.d:
            raise ValueError(f"xb must have shape [n, {self.d}], got {tuple(xb.shape)}")
        with torch.cuda.device(xb.device):
            _lib.check(self._lib.proqa_index_add_device(self._h, xb.data_ptr(), xb.shape[0], _torch_dtype_code(xb),
                                                        _lib.current_stream_ptr()))

    def adopt_device(self, xb):
        """Search caller-owned fp16 rows in place (no copy); the tensor is kept alive here.

        The rows must not change while adopted: searches of k <= 128 over >= 65536 rows scan an int8 copy of them
        (+128 B per row) that is built once.  After writing into the tensor call `rows_changed()` -- or switch the copy
        off with `configure_nomination("off")`; otherwise results can silently miss rows."""
        import torch
        if not (xb.is_cuda and xb.dtype == torch.float16 and xb.is_contiguous() and xb.dim() == 2
                and xb.shape[1] == self.d):
            raise ValueError("adopt_device expects a contiguous CUDA float16 [n, d] tensor")
        torch.cuda.current_stream().synchronize()
        _lib.check(self._lib.proqa_index_adopt_device(self._h, xb.data_ptr(), xb.shape[0]))
        self._adopted = xb

    def rows_changed(self):
        """The adopted tensor was modified in place: the int8 copy of the rows is rebuilt before its next use."""
        _lib.check(self._lib.proqa_index_rows_changed(self._h))

    def prepare(self):
        """Build now what the first search would otherwise build (the int8 copy of the rows: 128 B per row, ~3 ms per
        18M rows, one host wait).  Optional; `search` / `search_device` build it on demand, the enqueued search of
        ShardedIndexFlatIP leaves it to the end of its first step."""
        import torch
        _lib.check(self._lib.proqa_index_prepare(self._h, _lib.current_stream_ptr() if torch.cuda.is_available() else None))

    def search_device(self, xq, k, idx_offset=0, out=None):
        """CUDA tensor in, CUDA tensors out: (D float32 [nq,k], I int64 [nq,k]); `out` = (D, I) to write into."""
        import torch
        if not xq.is_cuda:
            raise ValueError("search_device expects a CUDA tensor")
        xq = xq.contiguous()
        if xq.dim() != 2 or xq.shape[1] != self.d:
            raise ValueError(f"xq must have shape [nq, {self.d}], got {tuple(xq.shape)}")
        nq = xq.shape[0]
        if out is not None:
            D, I = out
            if (tuple(D.shape), D.dtype, tuple(I.shape), I.dtype) != ((nq, k), torch.float32, (nq, k), torch.int64) or \
                    not (D.is_contiguous() and I.is_contiguous() and D.device == xq.device and I.device == xq.device):
                raise ValueError("out must be contiguous (float32 [nq,k], int64 [nq,k]) tensors on xq's device")
        else:
            D = torch.empty((nq, k), dtype=torch.float32, device=xq.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=xq.device)
        with torch.cuda.device(xq.device):
            for q0 in range(0, max(nq, 1), QUERY_BATCH):
                part = xq[q0:q0 + QUERY_BATCH]
                _lib.check(self._lib.proqa_index_search_device(self._h, part.data_ptr(), part.shape[0],
                                                               _torch_dtype_code(xq), int(k), int(idx_offset),
                                                               D[q0:].data_ptr(), I[q0:].data_ptr(),
                                                               _lib.current_stream_ptr()))
        return D, I

    def reconstruct_batch_device(self, ids, dtype=None, idx_offset=0):
        """Rows of the index by id (faiss reconstruct_batch) without leaving the GPU: ids int64 CUDA tensor of any shape
        (e.g. the I of search_device) -> [*ids.shape, d] tensor of `dtype` (torch.float16, the stored rows, or
        torch.float32); ids outside the index (-1 of a short result) give zero rows."""
        import torch
        if not ids.is_cuda or ids.dtype != torch.int64:
            raise ValueError("reconstruct_batch_device expects an int64 CUDA tensor")
        dtype = dtype or torch.float16
        ids = ids.contiguous()
        out = torch.empty(tuple(ids.shape) + (self.d,), dtype=dtype, device=ids.device)
        with torch.cuda.device(ids.device):
            _lib.check(self._lib.proqa_index_reconstruct_batch_device(self._h, ids.data_ptr(), ids.numel(), int(idx_offset),
                                                                      out.data_ptr(), _torch_dtype_code(out),
                                                                      _lib.current_stream_ptr()))
        return out

    # -- introspection / tuning ---------------------------------------------------------
    def last_stats(self):
        st = _lib.SearchStats()
        _lib.check(self._lib.proqa_index_last_stats(self._h, ctypes.byref(st)))
        return {"rounds": st.rounds, "fallback_rounds": st.fallback_rounds, "candidates": st.candidates,
                "filter_ms": st.filter_ms, "total_ms": st.total_ms, "nominated": st.nominated,
                "nomination": bool(st.nomination),
                "nomination_state": ("off", "on", "suspended")[st.nomination_state] if 0 <= st.nomination_state <= 2 else None,
                "leap_rank": st.leap_rank,
                "leap_state": ("off", "on", "paused")[st.leap_state] if 0 <= st.leap_state <= 2 else None}

    def set_profiling(self, enable=True):
        _lib.check(self._lib.proqa_index_set_profiling(self._h, 1 if enable else 0))

    @property
    def exact_f32(self):
        """True once a float32 row or query that fp16 cannot hold has switched the index to
        exact-float32 mode (float32 copies of the rows, fp16 scan + exact re-scoring)."""
        v = ctypes.c_int(0)
        _lib.check(self._lib.proqa_index_is_exact_f32(self._h, ctypes.byref(v)))
        return bool(v.value)

    def allow_rounding(self, allow=True):
        """Round float32 inputs to fp16 instead of switching to exact-float32 mode (faster merge, half
        the memory; results are then those of the rounded vectors).  Off by default: the reference
        searches float32."""
        _lib.check(self._lib.proqa_index_allow_rounding(self._h, 1 if allow else 0))

    def configure(self, first_slab_rows=0, growth=0):
        """Round schedule: rows of the first (dense) slab and the growth factor of later slabs."""
        _lib.check(self._lib.proqa_index_configure(self._h, first_slab_rows, growth))

    def configure_bootstrap(self, rows):
        """Rows covered by the dense bootstrap (exact top-k of the first rows from a score matrix); 0 disables it."""
        _lib.check(self._lib.proqa_index_configure_bootstrap(self._h, int(rows)))

    def configure_nomination(self, mode):
        """The int8 nomination scan of the k <= 128 rounds (see proqa_hip.h): 0 / "off" = fp16 scan only, 1 / "auto"
        (default: a search that over-nominates suspends the int8 rounds, later searches re-probe them; `last_stats()
        ["nomination_state"]`, one stderr line per switch under PROQA_LOG=1), 2 / "always".  The result is the fp16
        scan's either way."""
        mode = {"off": 0, "auto": 1, "always": 2}.get(mode, mode)
        _lib.check(self._lib.proqa_index_configure_nomination(self._h, int(mode)))

    def configure_leap(self, mode):
        """Leaping rounds of the k <= 128 searches (see proqa_hip.h): thresholds at a rank j < k of the running lists, a
        third to a half of the rounds; a round that falls short is re-scanned (the result never depends on it) and pauses
        the leaps of this index (`last_stats()["leap_rank"]`, `["leap_state"]`).  0 / "off", 1 / "auto" (default)."""
        mode = {"off": 0, "auto": 1}.get(mode, mode)
        _lib.check(self._lib.proqa_index_configure_leap(self._h, int(mode)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.proqa_index_free(self._h)
            self._h = None
            self._adopted = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class PipelinedSearcher:
    """A STREAM of query batches over one set of rows, two searches in flight.

    One search is a chain of ~20 dependent launches (bootstrap, rounds of scan + merge, finalize) and a host wait; at the
    shard sizes of a multi-GPU job the chain is a third of the step.  Two IndexFlatIP handles over the same adopted rows,
    each on its own stream, take the batches in turn: batch i+1 is enqueued (proqa_index_search_begin_device) before the
    host waits for batch i (proqa_index_search_finish), so one search's small launches run beside the other's large
    scans.  Results are those of the one-call search, in order.  Cost: every handle builds its own int8 copy of the rows
    (+128 B per row each) and its own workspace.  Measured: 1.08 x the call-by-call rate at 18M rows, 1.19 x at 2.25M
    (2032 queries, k = 80; profiles/ABLATIONS.md R6.3).  The reference searches once (eval_retrieval.py:102-104); this is
    for callers that have many batches (an evaluation loop over several query files, the online sampler)."""

    def __init__(self, xb, d=EMBED_DIM):
        import torch
        self.handles = [IndexFlatIP(d), IndexFlatIP(d)]
        self.streams = [torch.cuda.Stream(device=xb.device), torch.cuda.Stream(device=xb.device)]
        for h in self.handles:
            h.adopt_device(xb)
            h.prepare()
        self.d = d

    def search_batches(self, batches, k, idx_offset=0):
        """Generator: for every CUDA tensor [nq, d] of `batches` one (D float32 [nq,k], I int64 [nq,k]), in order; a batch's
        result is complete (host-synchronised) when it is yielded."""
        import torch
        lib = _lib.load()
        k = int(k)
        pending = []
        try:
            for n, xq in enumerate(batches):
                slot = n & 1
                if len(pending) == 2:
                    yield self._finish(pending.pop(0))
                if not xq.is_cuda or xq.dim() != 2 or xq.shape[1] != self.d:
                    raise ValueError(f"batches must be CUDA tensors of shape [nq, {self.d}]")
                if xq.shape[0] > QUERY_BATCH:
                    raise ValueError(f"at most {QUERY_BATCH} queries per batch")
                xq = xq.contiguous()
                nq = xq.shape[0]
                D = torch.empty((nq, k), dtype=torch.float32, device=xq.device)
                I = torch.empty((nq, k), dtype=torch.int64, device=xq.device)
                st = self.streams[slot]
                st.wait_stream(torch.cuda.current_stream(xq.device))  # the batch may still be on its way on the caller's stream
                with torch.cuda.device(xq.device):
                    _lib.check(lib.proqa_index_search_begin_device(self.handles[slot]._h, xq.data_ptr(), nq, _torch_dtype_code(xq),
                                                                   k, int(idx_offset), D.data_ptr(), I.data_ptr(), None,
                                                                   st.cuda_stream))
                pending.append((slot, xq, D, I))
            while pending:
                yield self._finish(pending.pop(0))
        finally:
            # a consumer that stops early (or an error above): the searches still in flight write into tensors this frame
            # is about to drop -- complete them first
            for item in pending:
                try:
                    self._finish(item)
                except Exception:
                    pass

    def _finish(self, item):
        slot, _xq, D, I = item
        _lib.check(_lib.load().proqa_index_search_finish(self.handles[slot]._h, None))
        return D, I

    def last_stats(self):
        return [h.last_stats() for h in self.handles]

    def close(self):
        for h in self.handles:
            h.close()


def merge_topk_device(D_parts, I_parts):
    """Merge [n_parts, nq, k] per-shard lists (ascending shard order) into [nq, k] on the GPU.  The parts need not be
    sorted inside (this entry point sorts); slots with I = -1 are missing rows wherever they sit."""
    import torch
    lib = _lib.load()
    D_parts = D_parts.contiguous()
    I_parts = I_parts.contiguous()
    n_parts, nq, k = D_parts.shape
    D = torch.empty((nq, k), dtype=torch.float32, device=D_parts.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=D_parts.device)
    with torch.cuda.device(D_parts.device):
        _lib.check(lib.proqa_topk_merge_device(D_parts.data_ptr(), I_parts.data_ptr(), n_parts, nq, k,
                                               D.data_ptr(), I.data_ptr(), _lib.current_stream_ptr()))
    return D, I


def shard_bounds(n_rows, world_size, rank):
    """Contiguous row range of `rank`: rows [r*N/G, (r+1)*N/G)."""
    lo = (n_rows * rank) // world_size
    hi = (n_rows * (rank + 1)) // world_size
    return lo, hi


class QueryShardedIndexFlatIP:
    """The other decomposition of the multi-GPU search: every rank holds ALL rows, the QUERIES are sharded.

    An 18M x 128 fp16 index and its int8 copy are 6.9 GB of a GPU's 288 GB.  Rank r searches queries [r nq / G, (r + 1) nq / G)
    over the whole corpus (IndexFlatIP, i.e. the HBM-bound small-batch regime for the usual 2032 queries on 8 ranks) and the
    ranks all-gather their result rows: no rank merge, nothing to exchange but the answer, and every query's result is the
    single-GPU search's by construction.  Measured on one GPU: 254 queries over 18M rows 0.72 ms, against 0.81-0.83 ms for
    2032 queries over a 2.25M-row shard before its all-gather and merge (`bench.py --gpus N` reports
    this decomposition as `query_shards` beside the row-sharded `value` that BASELINE configs[3] prescribes).
    `local_search(xq, k) -> (D, I)` is injectable so that the plumbing runs under gloo on CPU in the tests."""

    def __init__(self, d=EMBED_DIM, group=None, local_search=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.d = d
        self._local_search = local_search
        self._index = IndexFlatIP(d) if local_search is None else None

    @property
    def local_index(self):
        return self._index

    def add(self, xb):
        """ALL rows of the corpus, on every rank (host array or CUDA tensor)."""
        self._index.add(xb)

    def add_npy(self, path, readers=0):
        """every rank loads the whole .npy index file"""
        self._index.add_npy(path, 0, -1, readers)

    def adopt(self, xb):
        """search all rows in place (a contiguous CUDA fp16 tensor, see IndexFlatIP.adopt_device)"""
        self._index.adopt_device(xb)

    def prepare(self):
        if self._index is not None:
            self._index.prepare()

    def search(self, xq, k):
        """All ranks call with the same queries [nq, d]; returns (D float32 [nq,k], I int64 [nq,k]) on xq's device, on every
        rank.  ONE collective: the ranks' result rows (ids and scores as one byte buffer per rank, padded to equal slices)."""
        import torch
        nq = xq.shape[0]
        per = (nq + self.world_size - 1) // self.world_size
        q0, q1 = min(self.rank * per, nq), min((self.rank + 1) * per, nq)
        mine = xq[q0:q1].contiguous()
        if q1 > q0:
            D, I = self._local_search(mine, k) if self._local_search is not None else self._index.search_device(mine, k)
        else:
            D = torch.empty((0, k), dtype=torch.float32, device=xq.device)
            I = torch.empty((0, k), dtype=torch.int64, device=xq.device)
        if self.world_size == 1:
            return D, I
        Dp = torch.zeros((per, k), dtype=torch.float32, device=D.device)
        Ip = torch.full((per, k), -1, dtype=torch.int64, device=D.device)
        Dp[:q1 - q0] = D
        Ip[:q1 - q0] = I
        n_i, n_d = Ip.numel() * 8, Dp.numel() * 4
        block = torch.cat([Ip.view(torch.uint8).reshape(-1), Dp.view(torch.uint8).reshape(-1)])
        gathered = torch.empty((self.world_size, n_i + n_d), dtype=torch.uint8, device=D.device)
        if self.dist.get_backend(self.group) == "nccl":     # RCCL
            self.dist.all_gather_into_tensor(gathered, block, group=self.group)
        elif gathered.is_cuda:                               # gloo with two ranks on one GPU (tests): staged through the host
            parts = [torch.empty_like(block, device="cpu") for _ in range(self.world_size)]
            self.dist.all_gather(parts, block.cpu(), group=self.group)
            gathered = torch.stack(parts).to(D.device)
        else:
            self.dist.all_gather(list(gathered.unbind(0)), block, group=self.group)
        I_all = gathered[:, :n_i].contiguous().view(torch.int64).reshape(self.world_size * per, k)
        D_all = gathered[:, n_i:].contiguous().view(torch.float32).reshape(self.world_size * per, k)
        return D_all[:nq].contiguous(), I_all[:nq].contiguous()

    def close(self):
        if self._index is not None:
            self._index.close()
            self._index = None


class ShardedIndexFlatIP:
    """Row-sharded exact index over a torch.distributed process group (one rank per GPU).

    Every rank holds rows [lo, hi) of the corpus and all queries; `search` runs the local exact
    top-k with global row ids, all-gathers the [nq, k] (score, id) lists of every rank (the only
    collective on the path) and merges them with the same ordering rule, so the result is
    bit-identical to the single-GPU search.  `local_search` / `merge` are injectable so the
    distributed plumbing can be exercised with the gloo backend on CPU in the tests.
    """

    def __init__(self, n_total, d=EMBED_DIM, group=None, local_search=None, merge=None, preallocate=True,
                 transport="torch"):
        """transport: who runs the all-gather -- "torch" (torch.distributed, RCCL under the nccl backend) or "cabi"
        (libproqa_hip.so's own RCCL communicator, proqa_sharded_search_device: the path a caller without PyTorch
        uses; torch.distributed only carries the 128-byte communicator id to the ranks here)."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = int(n_total)
        self.d = d
        self.lo, self.hi = shard_bounds(self.n_total, self.world_size, self.rank)
        self._local_search = local_search
        self._merge = merge or merge_topk_device
        self._index = None
        self._comm = None
        if transport not in ("torch", "cabi"):
            raise ValueError(f"unknown transport {transport!r}")
        self.transport = transport
        if local_search is None:
            self._index = IndexFlatIP(d, capacity=(self.hi - self.lo) if preallocate else 0)
        if transport == "cabi":
            if self._index is None:
                raise ValueError('transport="cabi" needs the built-in HIP searcher')
            self._comm = self._create_comm()

    def _create_comm(self):
        """proqa_comm over the ranks of the group: rank 0 draws the id, every rank joins (a collective)."""
        lib = _lib.load()
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.proqa_comm_get_unique_id(ident))
        if self.world_size > 1:
            box = [ident.raw]
            self.dist.broadcast_object_list(box, src=self.dist.get_global_rank(self.group, 0) if self.group else 0,
                                            group=self.group)
            ident = ctypes.create_string_buffer(box[0], 128)
        handle = ctypes.c_void_p()
        _lib.check(lib.proqa_comm_create(ident, self.world_size, self.rank, ctypes.byref(handle)))
        return handle

    def add_local(self, xb_local):
        """Add this rank's rows (exactly rows [lo, hi) of the corpus, in order)."""
        if self._index is None:
            raise RuntimeError("add_local is only available with the built-in HIP searcher")
        self._index.add(xb_local)
        if self._index.ntotal > self.hi - self.lo:
            raise ValueError("more rows added than this rank's shard holds")

    def add_local_npy(self, path, readers=0):
        """Load this rank's rows [lo, hi) of the .npy index file straight into HBM (every rank reads only its range)."""
        if self._index is None:
            raise RuntimeError("add_local_npy is only available with the built-in HIP searcher")
        if self._index.ntotal:
            raise ValueError("this rank's shard already holds rows")
        self._index.add_npy(path, self.lo, self.hi - self.lo, readers)

    def adopt_local(self, xb_local):
        """Search this rank's rows [lo, hi) in place: a contiguous CUDA fp16 tensor, not copied (and not to be modified
        while adopted: see IndexFlatIP.adopt_device / rows_changed)."""
        if self._index is None:
            raise RuntimeError("adopt_local is only available with the built-in HIP searcher")
        if xb_local.shape[0] != self.hi - self.lo:
            raise ValueError(f"rank {self.rank} holds rows [{self.lo}, {self.hi}), got {xb_local.shape[0]} rows")
        self._index.adopt_device(xb_local)

    def prepare(self):
        """Every rank builds the int8 copy of its shard now (IndexFlatIP.prepare) instead of at the end of its first
        search step."""
        if self._index is not None:
            self._index.prepare()

    @property
    def local_index(self):
        return self._index

    def search(self, xq, k, force_collective=False):
        """All ranks call with the same queries; returns torch tensors (D, I) on xq's device.
        force_collective: go through the all-gather and the merge even with a single rank."""
        import torch
        if self._comm is not None:
            return self._search_cabi(xq, k)
        exchange = self.world_size > 1 or force_collective
        if exchange and not self.dist.is_initialized():
            raise RuntimeError('force_collective with transport="torch" needs an initialised torch.distributed process '
                               'group (transport="cabi" brings its own RCCL communicator)')
        if exchange and self._local_search is None and self._merge is merge_topk_device and \
                self.dist.get_backend(self.group) == "nccl":
            return self._search_in_place(xq, k)
        if self._local_search is not None:
            D, I = self._local_search(xq, k, self.lo)
        else:
            D, I = self._index.search_device(xq, k, idx_offset=self.lo)
        if not exchange:
            return D, I
        # ONE collective: ids (int64) and scores (float32) travel as one byte buffer per rank;
        # rank-ordered slices of the gathered buffer are exactly the [n_parts, nq, k] layout the merge consumes
        n_i, n_d = I.numel() * 8, D.numel() * 4
        mine = torch.cat([I.contiguous().view(torch.uint8).reshape(-1), D.contiguous().view(torch.uint8).reshape(-1)])
        gathered = torch.empty((self.world_size, n_i + n_d), dtype=torch.uint8, device=D.device)
        if self.dist.get_backend(self.group) == "nccl":     # RCCL: one flat collective, no staging copies
            self.dist.all_gather_into_tensor(gathered, mine, group=self.group)
        else:                                                # gloo (CPU tests, two ranks on one GPU)
            self.dist.all_gather(list(gathered.unbind(0)), mine, group=self.group)
        I_all = gathered[:, :n_i].contiguous().view(torch.int64).reshape((self.world_size,) + tuple(I.shape))
        D_all = gathered[:, n_i:].contiguous().view(torch.float32).reshape((self.world_size,) + tuple(D.shape))
        return self._merge(D_all, I_all)

    def _search_in_place(self, xq, k):
        """The RCCL exchange without staging copies and without a host round trip in the middle: the local search is
        only ENQUEUED (proqa_index_search_begin_device) and writes its ids and scores straight into this rank's block
        [ids | scores | status word, 16 B] of the send buffer; ONE all_gather_into_tensor and the strided merge
        (proqa_topk_merge_strided_device), which reads the receive buffer where it lies, go onto the stream right behind
        it; the host waits once, at the end (proqa_index_search_finish).  The status words of all ranks come back with
        the blocks: a rank whose candidate lists overflowed (status 1) rewrites its list in _finish and every rank runs
        the exchange once more; a rank whose local search failed (status 0xFFFFFFFF) still enters the collective, and
        every rank raises instead of waiting for it."""
        import torch
        if not xq.is_cuda:
            raise ValueError("sharded search expects a CUDA tensor")
        xq = xq.contiguous()
        if xq.dim() != 2 or xq.shape[1] != self.d:
            raise ValueError(f"xq must have shape [nq, {self.d}], got {tuple(xq.shape)}")
        k = int(k)
        if k <= 0:
            raise ValueError(f"k={k}")
        nq = xq.shape[0]
        D_out = torch.empty((nq, k), dtype=torch.float32, device=xq.device)
        I_out = torch.empty((nq, k), dtype=torch.int64, device=xq.device)
        # (every rank sees the same nq: with no query nobody enters the collective; above QUERY_BATCH queries the search
        # runs slice by slice like IndexFlatIP.search_device -- the candidate store of the library grows with nq)
        for q0 in range(0, nq, QUERY_BATCH):
            self._exchange_slice(xq[q0:q0 + QUERY_BATCH], k, D_out[q0:q0 + QUERY_BATCH], I_out[q0:q0 + QUERY_BATCH])
        return D_out, I_out

    def _exchange_slice(self, xq, k, D_out, I_out):
        """begin -> all-gather -> strided merge -> finish for one slice of at most QUERY_BATCH queries; D_out / I_out are
        contiguous [nq, k] views that receive the merged result."""
        import torch
        lib = _lib.load()
        nq = xq.shape[0]
        sizes = [ctypes.c_size_t() for _ in range(3)]
        _lib.check(lib.proqa_sharded_block_layout(nq, k, *[ctypes.byref(v) for v in sizes]))
        n_i, n_d, block = (v.value for v in sizes)
        # the send / receive buffers are kept from call to call (one set per slice size: the full slice and the tail)
        key = (nq, k, xq.device)
        cache = self.__dict__.setdefault("_xchg", {})
        if key not in cache:
            if len(cache) >= 2:
                cache.clear()
            cache[key] = (torch.empty(block, dtype=torch.uint8, device=xq.device),
                          torch.empty((self.world_size, block), dtype=torch.uint8, device=xq.device),
                          torch.zeros(self.world_size, dtype=torch.int32).pin_memory())
        mine, gathered, status_host = cache[key]
        status_dev = mine.data_ptr() + n_i + n_d
        failure = None
        with torch.cuda.device(xq.device):
            try:
                _lib.check(lib.proqa_index_search_begin_device(self._index._h, xq.data_ptr(), nq, _torch_dtype_code(xq), k,
                                                               int(self.lo), mine.data_ptr() + n_i, mine.data_ptr(), status_dev,
                                                               _lib.current_stream_ptr()))
            except Exception as e:       # still enter the collective: the other ranks are already on their way into it
                failure = e
            for _attempt in range(2):
                if failure is not None:
                    mine[n_i + n_d:n_i + n_d + 4] = 0xFF
                self.dist.all_gather_into_tensor(gathered, mine, group=self.group)
                # (the merge kernel drops every rank's status word into the pinned host buffer: no copy on the stream)
                _lib.check(lib.proqa_topk_merge_gathered_device(gathered.data_ptr(), self.world_size, nq, k,
                                                                status_host.data_ptr(), D_out.data_ptr(), I_out.data_ptr(),
                                                                _lib.current_stream_ptr()))
                if failure is None:
                    try:
                        _lib.check(lib.proqa_index_search_finish(self._index._h, None))
                    except Exception as e:
                        failure = e
                torch.cuda.current_stream().synchronize()
                status = status_host.tolist()
                if -1 in status:
                    raise failure if failure is not None else RuntimeError(
                        f"sharded search: rank {status.index(-1)} failed in its local search")
                if not any(status):
                    if failure is not None:     # the lists were exchanged (nobody waits), but this rank's result may be incomplete
                        raise failure
                    return
                if failure is None:
                    mine[n_i + n_d:n_i + n_d + 4] = 0
        raise RuntimeError("sharded search: the ranks did not agree on a final result")

    def _search_cabi(self, xq, k):
        """proqa_sharded_search_device: local search, RCCL all-gather and merge inside the library."""
        import torch
        if not xq.is_cuda:
            raise ValueError("sharded search expects a CUDA tensor")
        xq = xq.contiguous()
        nq = xq.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=xq.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=xq.device)
        lib = _lib.load()
        with torch.cuda.device(xq.device):
            for q0 in range(0, max(nq, 1), QUERY_BATCH):
                part = xq[q0:q0 + QUERY_BATCH]
                _lib.check(lib.proqa_sharded_search_device(self._index._h, self._comm, part.data_ptr(), part.shape[0],
                                                           _torch_dtype_code(xq), int(k), int(self.lo),
                                                           D[q0:].data_ptr(), I[q0:].data_ptr(),
                                                           _lib.current_stream_ptr()))
        return D, I

    def close(self):
        if getattr(self, "_comm", None):
            _lib.load().proqa_comm_free(self._comm)
            self._comm = None
        if getattr(self, "_index", None) is not None:
            self._index.close()
            self._index = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
