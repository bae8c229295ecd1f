"""The int8 nomination scan (mips_filter_i8 + MFMA re-scoring in topk_merge) must be invisible: every result is the fp16
scan's, bit for bit -- ids AND scores -- and therefore the oracle's on integer-valued corpora.

Replaces the same call site as the fp16 scan: /root/reference/retrieval/eval_retrieval.py:98-104
(`faiss.IndexFlatIP.add` / `.search`).  Mode "always" makes shards below the automatic 65536-row minimum and batches of
<= 256 queries take the int8 rounds, so that the oracle still finishes in seconds."""
import numpy as np
import pytest

from oracle import search_oracle

pytestmark = pytest.mark.gpu


def _int_corpus(rng, n, lo=-4, hi=4):
    return rng.integers(lo, hi + 1, (n, 128)).astype(np.float16)


def _both(xb, xq, k, add=None, idx_offset=0):
    """(D, I, stats) of the fp16 scan and of the int8 nomination scan on the same rows"""
    import torch
    from proqa_amd.index import IndexFlatIP
    out = []
    tq = torch.from_numpy(xq).cuda()
    for mode in ("off", "always"):
        ix = IndexFlatIP(128)
        ix.configure_nomination(mode)
        (add or (lambda index, rows: index.add(rows)))(ix, xb)
        D, I = ix.search_device(tq, k, idx_offset=idx_offset)
        out.append((D.cpu().numpy(), I.cpu().numpy(), ix.last_stats(), ix))
    return out


@pytest.mark.parametrize("n,nq,k,lo,hi", [(70000, 600, 80, -4, 4), (20000, 300, 80, -4, 4), (100000, 257, 5, -8, 8),
                                          (40000, 512, 128, 0, 1), (9000, 40, 80, -4, 4), (70000, 1100, 1, -2, 2),
                                          # either side of the rule that picks the nominating merge (k x growth <= 200: the
                                          # 1024-key, eight-per-CU form), and k = 128 on the 1024-thread merge of a small batch
                                          (70000, 600, 100, -4, 4), (70000, 600, 101, -4, 4), (90000, 200, 128, -4, 4)])
def test_integer_corpora_ids_and_scores_identical_to_the_oracle(gpu_device, n, nq, k, lo, hi):
    rng = np.random.default_rng(n + nq + k)
    xb, xq = _int_corpus(rng, n, lo, hi), _int_corpus(rng, nq, lo, hi)
    (D0, I0, st0, _), (D1, I1, st1, _) = _both(xb, xq, k)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    assert st1["nomination"] and not st0["nomination"] and st1["nominated"] >= nq * min(k, 1)
    np.testing.assert_array_equal(I1, Io)
    np.testing.assert_array_equal(D1, Do)
    np.testing.assert_array_equal(I0, Io)


@pytest.mark.parametrize("dist", ["normal", "shifted", "anisotropic", "constant_dims", "tiny", "huge"])
def test_float_corpora_bit_identical_to_the_fp16_scan(gpu_device, dist):
    """Random fp16 data in the shapes that stress the quantiser: a large common component (what the centring is for), dimensions
    of very different scale (per-dimension equalisation), constant dimensions (they quantise to nothing; their contribution
    sits in the query's offset), values near the bottom / top of the fp16 range."""
    rng = np.random.default_rng(5)
    n, nq, k = 150000, 520, 80
    xb = rng.standard_normal((n, 128)).astype(np.float32)
    xq = rng.standard_normal((nq, 128)).astype(np.float32)
    if dist == "shifted":
        xb += 20.0
    elif dist == "anisotropic":
        xb *= np.exp(rng.uniform(-4, 4, 128)).astype(np.float32)
    elif dist == "constant_dims":
        xb[:, ::3] = 7.5
        xb[:, 1] = 2000.0
    elif dist == "tiny":
        xb *= 1e-3
        xq *= 1e-2
    elif dist == "huge":
        xb *= 40.0
        xq *= 8.0
    xb, xq = xb.astype(np.float16), xq.astype(np.float16)
    (D0, I0, st0, _), (D1, I1, st1, _) = _both(xb, xq, k)
    assert st1["nomination"] and st1["fallback_rounds"] == 0
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1.view(np.int32), D0.view(np.int32))
    # and the fp16 scan is the oracle's search (same tolerance as test_random_fp16_matches_oracle)
    Do, Io = search_oracle.topk_ip(xq[:64], xb, k)
    np.testing.assert_allclose(D1[:64], Do, rtol=1e-5, atol=1e-4 * max(1.0, float(np.abs(Do).max())))
    assert np.mean([len(set(a) & set(b)) / k for a, b in zip(I1[:64], Io)]) > 1 - 1e-4
    # the margin is rigorous, not generous: a few times the fp16 scan's candidates (scripts/dev_int8_margin.py)
    assert st1["nominated"] < 8 * max(st0["candidates"], nq * k), (st1, st0)


def test_a_row_the_int8_score_undersells_by_most_of_the_margin_is_still_returned(gpu_device):
    """A planted row whose rounding residuals all point along the query: its int8 score falls short of its exact score by 45 % of
    the rigorous margin (0.44 against 0.97 in score units), and its exact score sits 0.04 above the k-th best.  A scan that
    lowered its threshold by less than 40 % of the bound would lose it."""
    rng = np.random.default_rng(17)
    n, nq, k = 80000, 300, 10
    xb = (rng.integers(-128, 129, (n, 128)) / 64.0).astype(np.float16)          # fp16-exact grid, |x| <= 2
    xq = (rng.integers(-64, 65, (nq, 128)) / 64.0).astype(np.float16)
    x32, q32 = xb.astype(np.float32), xq.astype(np.float32)
    # the quantiser of csrc/mips_kernels.hip, restated: centre, scale every dimension to +-127, round
    mu = x32.mean(axis=0)
    c = np.abs(x32 - mu).max(axis=0)
    q = 0                                                                         # the query the row is planted for
    tau = np.sort(x32 @ q32[q])[-k]
    # the planted row: a multiple of the query on the int8 grid plus a residual of 0.49 grid steps in every dimension, signed
    # like the query's weight in that dimension (so that every rounding error lowers the int8 score); the multiple is raised
    # until the exact score is just above tau
    w = q32[q] * c / 127.0
    s_q = np.abs(w).max() / 127.0
    qi = np.clip(np.rint(w / s_q), -127, 127)
    best = None
    for alpha in np.arange(0.0, 120.0, 0.05):
        base_i = np.clip(np.rint(alpha * q32[q] / np.abs(q32[q]).max()), -100, 100)
        row = (mu + (c / 127.0) * (base_i + 0.49 * np.sign(w))).astype(np.float16).astype(np.float32)
        score = float(row @ q32[q])
        if score > tau:
            xi = np.clip(np.rint((row - mu) * 127.0 / c), -127, 127)
            best = (score - tau, row, score - (float(q32[q] @ mu) + s_q * float(qi @ xi)))
            break
    assert best is not None and best[0] < 0.2
    xb[n // 2] = best[1].astype(np.float16)
    x32 = xb.astype(np.float32)
    u = w / (np.abs(w).max() / 127.0)
    v = (x32 - x32.mean(axis=0)) * 127.0 / np.abs(x32 - x32.mean(axis=0)).max(axis=0)
    margin = (np.abs(w).max() / 127.0) * (np.linalg.norm(u) * np.linalg.norm(v - np.rint(v), axis=1).max()
                                          + np.linalg.norm(u - np.rint(u)) * np.linalg.norm(np.rint(v), axis=1).max())
    assert best[2] > 0.4 * margin, (best[2], margin)      # a real stress of the bound, not a row it covers ten times over
    (D0, I0, _, _), (D1, I1, st1, _) = _both(xb, xq, k)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    assert st1["nomination"]
    assert n // 2 in Io[q]                                 # the planted row belongs to the answer ...
    np.testing.assert_array_equal(I1, I0)                  # ... and the nomination scan returns it
    np.testing.assert_array_equal(D1.view(np.int32), D0.view(np.int32))
    assert np.mean([len(set(a) & set(b)) / k for a, b in zip(I1, Io)]) > 1 - 1e-3


def test_data_that_does_not_quantise_goes_back_to_the_fp16_scan(gpu_device):
    """An outlier row in EVERY 32-row block (-2000 in all dimensions where the bulk is within +-4; the queries are non-negative,
    so the outliers themselves never rank) stretches every block's scale until the bulk rows all quantise to zero -- a few
    outlier rows would only cost their own blocks (test_float_corpora... / test_full_size_properties).  The first search
    nominates everything (lists overflow, the fp16 overflow-safe path finishes it -- the result is still exact); the
    automatic mode then stays with the fp16 scan until the rows change."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(23)
    n, nq, k = 70000, 300, 20
    xb, xq = _int_corpus(rng, n), _int_corpus(rng, nq, 0, 4)
    xb[5::32] = -2000.0
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    ix = IndexFlatIP(128)
    ix.add(xb)
    tq = torch.from_numpy(xq).cuda()
    stats = []
    for _ in range(2):
        D, I = ix.search_device(tq, k)
        np.testing.assert_array_equal(I.cpu().numpy(), Io)
        np.testing.assert_array_equal(D.cpu().numpy(), Do)
        stats.append(ix.last_stats())
    assert stats[0]["nomination"] and (stats[0]["fallback_rounds"] > 0 or stats[0]["nominated"] > nq * 4096)
    assert not stats[1]["nomination"] and stats[1]["fallback_rounds"] == 0
    assert stats[0]["nomination_state"] == stats[1]["nomination_state"] == "suspended"
    ix.add(xb[:1000])                                        # the rows changed: the copy is rebuilt, the verdict forgotten
    D, I = ix.search_device(tq, k)
    assert ix.last_stats()["nomination"]
    Do2, Io2 = search_oracle.topk_ip(xq, np.concatenate([xb, xb[:1000]]), k)
    np.testing.assert_array_equal(I.cpu().numpy(), Io2)
    np.testing.assert_array_equal(D.cpu().numpy(), Do2)


def test_adversarial_order_overflows_into_the_fp16_path(gpu_device):
    """Scores that rise with the row number: every row is nominated, the lists of the int8 rounds overflow and the fp16
    overflow-safe path re-scans them -- exact, with the fallback counted."""
    rng = np.random.default_rng(3)
    n, nq, k = 90000, 300, 80
    xb = rng.integers(-1, 2, (n, 128)).astype(np.float16)
    xb[:, 0] = np.minimum(np.arange(n) // 7, 2000)
    xq = rng.integers(0, 2, (nq, 128)).astype(np.float16)
    xq[:, 0] = 1
    (D0, I0, _, _), (D1, I1, st1, _) = _both(xb, xq, k)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(I1, Io)
    np.testing.assert_array_equal(D1, Do)
    assert st1["nomination"] and st1["fallback_rounds"] > 0


def test_non_finite_rows_switch_the_scan_off(gpu_device):
    """inf / NaN rows cannot be centred or scaled: the int8 copy is marked unusable and the fp16 scan (heap rule for
    non-finite scores) answers, also in mode "always"."""
    rng = np.random.default_rng(31)
    n, nq, k = 70000, 300, 80
    xb, xq = _int_corpus(rng, n), _int_corpus(rng, nq)
    xq[:, 0] = 1.0
    xb[rng.choice(n, 30, replace=False), 0] = np.inf
    xb[rng.choice(n, 30, replace=False), 1] = np.nan
    (D0, I0, _, _), (D1, I1, st1, _) = _both(xb, xq, k)
    Do, Io = search_oracle.topk_ip_heap(xq, xb, k)
    assert not st1["nomination"]
    np.testing.assert_array_equal(I1, Io)
    np.testing.assert_array_equal(D1, Do)


def test_non_finite_queries_still_get_the_heap_rule(gpu_device):
    rng = np.random.default_rng(37)
    n, nq, k = 70000, 300, 10
    xb, xq = _int_corpus(rng, n), _int_corpus(rng, nq)
    xq[5, 3] = np.inf
    xq[9, 4] = np.nan
    xq[11] = 0
    (D0, I0, _, _), (D1, I1, st1, _) = _both(xb, xq, k)
    Do, Io = search_oracle.topk_ip_heap(xq, xb, k)
    np.testing.assert_array_equal(I1, Io)
    np.testing.assert_array_equal(D1, Do)


def test_shards_offsets_and_the_deferred_search(gpu_device):
    """Shard-local int8 copies (each shard centres and scales on its own rows), global ids, the rank merge, and the enqueued
    form of the search (`search_begin` / `finish`, what the sharded step uses)."""
    import torch
    from proqa_amd.index import IndexFlatIP, merge_topk_device
    rng = np.random.default_rng(41)
    n, nq, k = 210000, 400, 80
    xb, xq = _int_corpus(rng, n), _int_corpus(rng, nq)
    tq = torch.from_numpy(xq).cuda()
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    parts = []
    for lo, hi in ((0, 70000), (70000, 140001), (140001, n)):
        ix = IndexFlatIP(128)
        ix.configure_nomination("always")
        ix.add(torch.from_numpy(xb[lo:hi]).cuda())
        parts.append(ix.search_device(tq, k, idx_offset=lo))
        assert ix.last_stats()["nomination"]
    D, I = merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy(), Do)


def test_rows_added_after_a_search_rebuild_the_copy(gpu_device):
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(43)
    xb, xq = _int_corpus(rng, 120000), _int_corpus(rng, 300)
    tq = torch.from_numpy(xq).cuda()
    ix = IndexFlatIP(128)
    ix.configure_nomination("always")
    ix.add(xb[:70000])
    D, I = ix.search_device(tq, 80)
    np.testing.assert_array_equal(I.cpu().numpy(), search_oracle.topk_ip(xq, xb[:70000], 80)[1])
    ix.add(xb[70000:] * 3)                    # a different scale: the old per-dimension maxima no longer hold
    both = np.concatenate([xb[:70000], xb[70000:] * 3])
    D, I = ix.search_device(tq, 80)
    Do, Io = search_oracle.topk_ip(xq, both, 80)
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy(), Do)
    ix.reset()
    ix.add(xb[:66000])
    D, I = ix.search_device(tq, 80)
    np.testing.assert_array_equal(I.cpu().numpy(), search_oracle.topk_ip(xq, xb[:66000], 80)[1])
    assert ix.last_stats()["nomination"]


def test_automatic_mode_picks_the_scan_by_shape(gpu_device):
    """default mode: the int8 rounds for k <= 128 on shards of >= 65536 rows (any batch size); the fp16 scan for everything
    else (large k, small shards, exact-float32 indexes)."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(47)
    xb = _int_corpus(rng, 70000)
    ix = IndexFlatIP(128)
    ix.add(xb)
    for nq, k, want in ((300, 80, True), (256, 80, True), (1, 5, True), (300, 129, False), (300, 2000, False)):
        xq = _int_corpus(rng, nq)
        D, I = ix.search_device(torch.from_numpy(xq).cuda(), k)
        assert ix.last_stats()["nomination"] == want, (nq, k)
        Do, Io = search_oracle.topk_ip(xq, xb, k)
        np.testing.assert_array_equal(I.cpu().numpy(), Io)
    small = IndexFlatIP(128)
    small.add(xb[:60000])
    small.search_device(torch.from_numpy(_int_corpus(rng, 300)).cuda(), 80)
    assert not small.last_stats()["nomination"]
    f32 = IndexFlatIP(128)
    f32.add(xb.astype(np.float32) + np.float32(1e-4))        # values fp16 cannot hold: exact-float32 mode
    assert f32.exact_f32
    f32.search_device(torch.from_numpy(_int_corpus(rng, 300)).cuda(), 80)
    assert not f32.last_stats()["nomination"]


def _zero_queries(nq):
    return np.zeros((nq, 128), np.float16)


def test_a_suspended_scan_is_probed_again_and_resumes(gpu_device):
    """Automatic mode is not sticky: ONE batch that over-nominates (all-zero queries: every row's integer score ties with the
    threshold, every row is nominated, the lists overflow) suspends the int8 rounds -- the next seven eligible searches run
    on the fp16 rows, the eighth probes the int8 rounds again and, the rows being perfectly quantisable, resumes them.  A
    probe that fails doubles the distance to the next one (8, 16, ...).  Every result is the oracle's either way."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(53)
    n, nq, k = 70000, 300, 80
    xb = _int_corpus(rng, n)
    ix = IndexFlatIP(128)
    ix.add(xb)
    bad = torch.from_numpy(_zero_queries(nq)).cuda()
    Dz, Iz = search_oracle.topk_ip(_zero_queries(nq), xb, k)

    def search_bad():
        D, I = ix.search_device(bad, k)
        np.testing.assert_array_equal(I.cpu().numpy(), Iz)
        np.testing.assert_array_equal(D.cpu().numpy(), Dz)
        return ix.last_stats()

    def search_good(seed):
        xq = _int_corpus(np.random.default_rng(seed), nq)
        D, I = ix.search_device(torch.from_numpy(xq).cuda(), k)
        Do, Io = search_oracle.topk_ip(xq, xb, k)
        np.testing.assert_array_equal(I.cpu().numpy(), Io)
        np.testing.assert_array_equal(D.cpu().numpy(), Do)
        return ix.last_stats()

    st = search_good(0)
    assert st["nomination"] and st["nomination_state"] == "on" and st["fallback_rounds"] == 0
    st = search_bad()
    assert st["nomination"] and st["nomination_state"] == "suspended"
    assert st["fallback_rounds"] > 0 or st["nominated"] > nq * 4096
    seen = [search_good(100 + i) for i in range(20)]
    # searches 1-7 after the suspension: fp16 rows; the 8th probes and resumes; from then on the int8 rounds again
    assert [s["nomination"] for s in seen] == [False] * 7 + [True] * 13
    assert [s["nomination_state"] for s in seen] == ["suspended"] * 7 + ["on"] * 13
    assert all(s["fallback_rounds"] == 0 for s in seen)
    # a probe that fails: suspended again, the next probe twice as far away
    assert search_bad()["nomination_state"] == "suspended"
    pattern = [search_bad()["nomination"] for _ in range(8 + 16)]
    assert pattern == [False] * 7 + [True] + [False] * 15 + [True]
    # ... and the rows changing starts afresh
    ix.add(xb[:1000])
    xq = _int_corpus(np.random.default_rng(7), nq)
    D, I = ix.search_device(torch.from_numpy(xq).cuda(), k)
    assert ix.last_stats()["nomination"] and ix.last_stats()["nomination_state"] == "on"
    np.testing.assert_array_equal(I.cpu().numpy(), search_oracle.topk_ip(xq, np.concatenate([xb, xb[:1000]]), k)[1])


def test_the_switches_of_the_automatic_mode_are_logged(gpu_device):
    """PROQA_LOG=1: one stderr line when the int8 rounds are suspended, one when a probe starts, one when they resume."""
    import subprocess
    import sys
    import os
    code = r"""
import numpy as np, torch
from proqa_amd.index import IndexFlatIP
rng = np.random.default_rng(1)
xb = rng.integers(-4, 5, (70000, 128)).astype(np.float16)
ix = IndexFlatIP(128); ix.add(xb)
good = torch.from_numpy(rng.integers(-4, 5, (300, 128)).astype(np.float16)).cuda()
ix.search_device(good, 80)
ix.search_device(torch.zeros((300, 128), dtype=torch.float16, device="cuda"), 80)
for _ in range(8):
    ix.search_device(good, 80)
print(ix.last_stats()["nomination_state"])
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PROQA_LOG="1", PYTHONPATH=root), capture_output=True,
                       text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.strip().splitlines()[-1] == "on"
    lines = [ln for ln in p.stderr.splitlines() if ln.startswith("[proqa] ")]
    assert sum("int8 nomination scan suspended" in ln for ln in lines) == 1, lines
    assert sum("re-probing the int8 nomination scan" in ln for ln in lines) == 1, lines
    assert sum("int8 nomination scan resumed" in ln for ln in lines) == 1, lines
    q = subprocess.run([sys.executable, "-c", code], env=dict({k_: v for k_, v in os.environ.items() if k_ != "PROQA_LOG"},
                                                             PYTHONPATH=root), capture_output=True, text=True, timeout=600, cwd=root)
    assert q.returncode == 0 and "[proqa]" not in q.stderr


def test_an_enqueued_search_never_builds_the_copy_its_finish_does(gpu_device):
    """proqa_index_search_begin_device keeps its contract on a fresh index: it returns while the stream is still busy with
    the work enqueued before it (no synchronisation, no allocation of the int8 copy), the search runs on the fp16 rows with
    status word 0, and _finish -- the host wait the caller pays anyway -- builds the copy: the second enqueued search scans
    it.  proqa_index_prepare builds it ahead of time instead."""
    import ctypes
    import torch
    from proqa_amd import _lib
    from proqa_amd.index import IndexFlatIP
    lib = _lib.load()
    rng = np.random.default_rng(59)
    n, nq, k = 70000, 300, 80
    xb, xq = _int_corpus(rng, n), _int_corpus(rng, nq)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    tq = torch.from_numpy(xq).cuda()
    busy_a = torch.randn((8192, 8192), device="cuda")

    def begin_finish(ix, expect_busy):
        D = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
        I = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
        status = torch.full((4,), 77, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(12):                      # ~100 ms of work ahead of the search on its stream
            busy_a @ busy_a
        _lib.check(lib.proqa_index_search_begin_device(ix._h, tq.data_ptr(), nq, 0, k, 0, D.data_ptr(), I.data_ptr(),
                                                       status.data_ptr(), _lib.current_stream_ptr()))
        still_busy = not torch.cuda.current_stream().query()
        _lib.check(lib.proqa_index_search_finish(ix._h, None))
        np.testing.assert_array_equal(I.cpu().numpy(), Io)
        np.testing.assert_array_equal(D.cpu().numpy(), Do)
        assert status[0].item() == 0
        if expect_busy:
            assert still_busy, "search_begin_device waited for the stream"
        return ix.last_stats()

    ix = IndexFlatIP(128)
    ix.add(torch.from_numpy(xb).cuda())
    ix.configure_nomination("off")
    ix.search_device(tq, k)                      # the workspace of this batch size exists from here on
    ix.configure_nomination("auto")
    st = begin_finish(ix, True)
    assert not st["nomination"] and st["nomination_state"] == "on" and st["fallback_rounds"] == 0
    st = begin_finish(ix, False)                 # the copy was built by the first _finish (this search's candidate store
    assert st["nomination"] and st["nominated"] > 0   # may still grow: the first int8 search of its size on the handle)
    st = begin_finish(ix, True)
    assert st["nomination"]
    # the explicit form: prepare() right after the rows are in place
    ix2 = IndexFlatIP(128)
    ix2.add(torch.from_numpy(xb).cuda())
    ix2.configure_nomination("off")
    ix2.search_device(tq, k)
    ix2.configure_nomination("auto")
    ix2.prepare()
    ix2.prepare()                                # idempotent
    assert begin_finish(ix2, False)["nomination"]
    assert begin_finish(ix2, True)["nomination"]
    # a small shard / nomination off: prepare is a no-op
    small = IndexFlatIP(128)
    small.add(xb[:5000])
    small.prepare()
    small.search_device(tq, k)
    assert not small.last_stats()["nomination"]


def test_rows_changed_after_writing_into_adopted_rows(gpu_device):
    """Adopted rows are searched in place and the int8 copy is keyed on them: a caller that overwrites rows says so
    (rows_changed) and the next search rebuilds the copy -- a planted row that only exists after the write is found."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(61)
    n, nq, k = 80000, 300, 80
    xb, xq = _int_corpus(rng, n), _int_corpus(rng, nq)
    t = torch.from_numpy(xb).cuda()
    tq = torch.from_numpy(xq).cuda()
    ix = IndexFlatIP(128)
    ix.adopt_device(t)
    D, I = ix.search_device(tq, k)
    assert ix.last_stats()["nomination"]
    np.testing.assert_array_equal(I.cpu().numpy(), search_oracle.topk_ip(xq, xb, k)[1])
    xb2 = xb.copy()
    xb2[40001] = xq[0] * 4                      # the best match of query 0 by far, and a row far outside the old scales
    xb2[123] = 0
    t.copy_(torch.from_numpy(xb2))
    ix.rows_changed()
    D, I = ix.search_device(tq, k)
    Do, Io = search_oracle.topk_ip(xq, xb2, k)
    assert ix.last_stats()["nomination"] and Io[0, 0] == 40001
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy(), Do)


def test_near_ties_at_large_magnitude_are_the_fp16_scans_bits(gpu_device):
    """Scores of ~1e6 whose gaps are of the order of the fp32 accumulation's own rounding (2^-24 x 1e6 = 0.06): a common
    component of ~90 per dimension plus perturbations of a few fp16 ulps.  The nomination threshold's slack must cover the
    exact score's rounding (advisor finding, round 5: 2^-16 ||q|| max||x|| left no headroom; now 2^-14): ids and score bits of
    the int8 path equal the fp16 scan's."""
    rng = np.random.default_rng(67)
    n, nq, k = 150000, 300, 80
    base = rng.uniform(60, 120, 128).astype(np.float32)
    xb = (base + rng.integers(-3, 4, (n, 128)) * 0.0625).astype(np.float16)      # fp16 spacing at 64..128 is 0.0625
    xq = (base + rng.integers(-8, 9, (nq, 128)) * 0.0625).astype(np.float16)
    (D0, I0, st0, _), (D1, I1, st1, _) = _both(xb, xq, k)
    assert st1["nomination"] and not st0["nomination"]
    assert float(np.abs(D0).min()) > 3e5
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1.view(np.int32), D0.view(np.int32))
    # the exact scores really are near-tied at that magnitude: neighbouring results differ by less than 1e-5 relative
    assert float(np.median((D0[:, :-1] - D0[:, 1:]) / D0[:, :-1])) < 1e-5


@pytest.mark.parametrize("nq", [1, 7, 32, 33, 64, 65, 100, 128, 129, 200, 256])
def test_small_batches_row_split_launches_match_the_oracle(gpu_device, nq):
    """<= 128 queries run the row-split form of the int8 scan (1 / 2 / 4 query blocks replicated over the eight waves of a
    workgroup, which share the units of the stream and the lists of their queries through LDS counters); 129 .. 256 the
    one-block-per-wave form.  Ids and scores are the oracle's on an integer corpus either way -- also for k = 1 and with a
    ragged last chunk."""
    rng = np.random.default_rng(1000 + nq)
    n = 131072 + 77 * nq + 5
    xb, xq = _int_corpus(rng, n), _int_corpus(rng, nq)
    for k in (80, 1):
        (D0, I0, st0, _), (D1, I1, st1, _) = _both(xb, xq, k)
        Do, Io = search_oracle.topk_ip(xq, xb, k)
        assert st1["nomination"] and st1["fallback_rounds"] == 0
        np.testing.assert_array_equal(I1, Io)
        np.testing.assert_array_equal(D1, Do)
        np.testing.assert_array_equal(I0, Io)


@pytest.mark.parametrize("nq", [5, 40, 120])
def test_row_split_lists_that_overflow_go_to_the_fp16_path(gpu_device, nq):
    """Scores that rise with the row number make every unit a hit: the shared lists of a row-split launch fill up, the
    overflow word is raised and the fp16 overflow-safe path finishes the round -- exact, with the fallback counted."""
    rng = np.random.default_rng(7 + nq)
    n, k = 90000, 80
    xb = rng.integers(-1, 2, (n, 128)).astype(np.float16)
    xb[:, 0] = np.minimum(np.arange(n) // 7, 2000)
    xq = rng.integers(0, 2, (nq, 128)).astype(np.float16)
    xq[:, 0] = 1
    (D0, I0, _, _), (D1, I1, st1, _) = _both(xb, xq, k)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(I1, Io)
    np.testing.assert_array_equal(D1, Do)
    assert st1["nomination"] and st1["fallback_rounds"] > 0


def test_rows_that_change_between_all_searches_stop_rebuilding_the_copy(gpu_device):
    """add, search, add, search, ...: the int8 copy (two passes over ALL rows + an allocation) would be rebuilt for every
    single search.  After two copies in a row that served one search each, automatic mode leaves the first search after a
    change on the fp16 rows and rebuilds only when a second search finds the rows unchanged.  Exact either way."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(71)
    nq, k = 300, 40
    xq = _int_corpus(rng, nq)
    tq = torch.from_numpy(xq).cuda()
    parts = [_int_corpus(rng, 70000)] + [_int_corpus(rng, 3000) for _ in range(6)]
    ix = IndexFlatIP(128)
    seen = []
    for i, part in enumerate(parts):
        ix.add(part)
        D, I = ix.search_device(tq, k)
        Do, Io = search_oracle.topk_ip(xq, np.concatenate(parts[:i + 1]), k)
        np.testing.assert_array_equal(I.cpu().numpy(), Io)
        np.testing.assert_array_equal(D.cpu().numpy(), Do)
        seen.append(ix.last_stats()["nomination"])
    # builds 1-3 happen (the third finds two short-lived predecessors), from then on a changed index is searched on fp16 rows
    assert seen == [True, True, True, False, False, False, False], seen
    D, I = ix.search_device(tq, k)                      # the rows stayed: now the copy is rebuilt
    assert ix.last_stats()["nomination"]
    np.testing.assert_array_equal(I.cpu().numpy(), search_oracle.topk_ip(xq, np.concatenate(parts), k)[1])
    for _ in range(3):                                  # ... and a copy that serves several searches resets the verdict
        ix.search_device(tq, k)
    ix.add(parts[1])
    ix.search_device(tq, k)
    assert ix.last_stats()["nomination"]


def test_rows_that_overflow_the_small_merge_move_to_the_large_one_not_to_the_fp16_scan(gpu_device):
    """The 1024-key nominating merge (eight workgroups per CU) is chosen where a round is expected to nominate a quarter of
    what it holds.  Rows that nominate more overflow it -- the round is re-scanned exactly -- and the index then moves to the
    2048-key merge instead of suspending the int8 rounds.  Forced here: the rule's limit raised so that rounds growing by 8
    (~1100 nominations per query) take the small merge."""
    import subprocess
    import sys
    import os
    code = r"""
import numpy as np, torch
from oracle import search_oracle
from proqa_amd.index import IndexFlatIP
rng = np.random.default_rng(3)
xb = rng.standard_normal((400000, 128)).astype(np.float16)
xq = rng.standard_normal((600, 128)).astype(np.float16)
ix = IndexFlatIP(128); ix.configure(256, 8); ix.add(xb)
tq = torch.from_numpy(xq).cuda()
Do, Io = search_oracle.topk_ip(xq, xb, 80)
for i in range(3):
    D, I = ix.search_device(tq, 80)
    st = ix.last_stats()
    ov = np.mean([len(set(a) & set(b)) / 80 for a, b in zip(I.cpu().numpy(), Io)])
    print("RESULT", i, st["nomination"], st["nomination_state"], st["fallback_rounds"], float(ov) > 1 - 1e-4)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PROQA_LOG="1", PROQA_NOM_SMALL_MERGE_LIMIT="100000", PYTHONPATH=root)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [ln.split()[1:] for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
    assert len(rows) == 3
    # search 0: int8 rounds with the small merge, overflow -> exact by the overflow-safe path, NOT suspended
    assert rows[0][1] == "True" and rows[0][2] == "on" and int(rows[0][3]) > 0 and rows[0][4] == "True", rows
    # searches 1, 2: int8 rounds with the large merge, nothing overflows
    for r in rows[1:]:
        assert r[1] == "True" and r[2] == "on" and int(r[3]) == 0 and r[4] == "True", rows
    assert sum("overflowed the 1024-key merge" in ln for ln in p.stderr.splitlines()) == 1, p.stderr[-1500:]
    assert "suspended" not in p.stderr


@pytest.mark.parametrize("n,nq,k", [(400000, 1, 5000), (2000000, 32, 2000), (1000000, 7, 5000), (200000, 5, 700)])
def test_one_pass_large_k_for_a_few_queries_scans_the_int8_copy(gpu_device, n, nq, k):
    """k in the thousands for <= 32 queries (online_sampler.py:113: one question, k = 5000): the one launch over the shard is an
    HBM stream -- it runs on the int8 copy (half the bytes), the nominated rows are re-scored by rescore_nominated_lists and
    the compact merge sorts what beats the sampled threshold (taken while rows >= 16 x queries x the rows expected above the
    threshold: the gather of the nominated rows must stay small beside the bytes saved).  Ids and score bits are the fp16
    launch's."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(n + nq + k)
    xb = rng.standard_normal((n, 128)).astype(np.float16)
    xq = rng.standard_normal((nq, 128)).astype(np.float16)
    tq = torch.from_numpy(xq).cuda()
    out = []
    for mode in ("off", "auto"):
        ix = IndexFlatIP(128)
        ix.configure_nomination(mode)
        ix.add(xb)
        D, I = ix.search_device(tq, k)
        out.append((D.cpu().numpy(), I.cpu().numpy(), ix.last_stats()))
        ix.close()
    (D0, I0, st0), (D1, I1, st1) = out
    assert not st0["nomination"] and st1["nomination"] and st1["nominated"] >= nq * k, (st0, st1)
    assert st0["fallback_rounds"] == 0 and st1["fallback_rounds"] == 0
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1.view(np.uint32), D0.view(np.uint32))
