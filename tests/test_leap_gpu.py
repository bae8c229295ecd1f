"""Leaping rounds (thresholds at a rank j < k of the running lists: proqa_index_configure_leap, mips_index.cpp plan_leap) must be
invisible: ids and scores are those of the ordinary rounds bit for bit, whether a leap holds or falls short and is re-scanned.

Replaces the same call site as every search: /root/reference/retrieval/eval_retrieval.py:98-104 (`faiss.IndexFlatIP.add` /
`.search`)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import search_oracle

pytestmark = pytest.mark.gpu


def _search(xb, xq, k, leap, nomination="auto", growth=0):
    import torch
    from proqa_amd.index import IndexFlatIP
    ix = IndexFlatIP(128)
    ix.configure_leap(leap)
    ix.configure_nomination(nomination)
    if growth:
        ix.configure(growth=growth)
    ix.add(xb)
    D, I = ix.search_device(torch.from_numpy(xq).cuda(), k)
    return D.cpu().numpy(), I.cpu().numpy(), ix.last_stats(), ix


@pytest.mark.parametrize("n,nq,k,nomination", [(300000, 600, 80, "always"), (300000, 600, 80, "off"), (1000000, 33, 10, "always"),
                                               (400000, 2000, 128, "always"), (200000, 1, 80, "always"), (120000, 300, 8, "off"),
                                               (66000, 700, 80, "auto"), (250000, 257, 100, "always"), (250000, 130, 101, "off")])
def test_leaping_rounds_give_the_ordinary_rounds_result(gpu_device, n, nq, k, nomination):
    rng = np.random.default_rng(n + nq + k)
    xb = rng.standard_normal((n, 128)).astype(np.float16)
    xq = rng.standard_normal((nq, 128)).astype(np.float16)
    D0, I0, st0, _ = _search(xb, xq, k, "off", nomination)
    D1, I1, st1, _ = _search(xb, xq, k, "auto", nomination)
    assert st0["leap_rank"] == 0 and st0["leap_state"] == "off"
    # rows in random order: the leap holds (a shortfall has probability 1e-8 per query and round)
    assert 0 < st1["leap_rank"] < k and st1["leap_state"] == "on" and st1["fallback_rounds"] == 0, st1
    assert st1["rounds"] <= st0["rounds"] and st1["nomination"] == st0["nomination"]
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1.view(np.uint32), D0.view(np.uint32))


def _sorted_rows(rng, n, nq):
    """rows sorted by their component along u, queries along u: every later row scores lower than the first ones"""
    u = rng.standard_normal(128)
    u /= np.linalg.norm(u)
    xb = rng.standard_normal((n, 128)).astype(np.float32)
    xb = xb[np.argsort(-(xb @ u), kind="stable")].astype(np.float16)
    xq = (6.0 * u[None, :] + 0.3 * rng.standard_normal((nq, 128))).astype(np.float16)
    return xb, xq


@pytest.mark.parametrize("nq,nomination", [(300, "always"), (2000, "auto"), (600, "off")])
def test_a_leap_that_falls_short_is_rescanned_and_pauses_the_leaps(gpu_device, nq, nomination):
    import torch
    rng = np.random.default_rng(11)
    n, k = 300000, 80
    xb, xq = _sorted_rows(rng, n, nq)
    D0, I0, st0, _ = _search(xb, xq, k, "off", nomination)
    D1, I1, st1, ix = _search(xb, xq, k, "auto", nomination)
    assert st0["fallback_rounds"] == 0
    assert st1["leap_rank"] > 0 and st1["fallback_rounds"] > 0 and st1["leap_state"] == "paused", st1
    if nomination == "always":   # a leap's shortfall says nothing about the int8 copy
        assert st1["nomination_state"] == "on"
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1.view(np.uint32), D0.view(np.uint32))
    # the next 16 searches take ordinary rounds, the 17th leaps again, falls short again and pauses for 32
    tq = torch.from_numpy(xq).cuda()
    for i in range(16):
        D, I = ix.search_device(tq, k)
        st = ix.last_stats()
        assert st["leap_rank"] == 0 and st["fallback_rounds"] == 0 and st["rounds"] == st0["rounds"], (i, st)
        assert st["leap_state"] == ("paused" if i < 15 else "on")
    np.testing.assert_array_equal(I.cpu().numpy(), I0)
    D, I = ix.search_device(tq, k)
    st = ix.last_stats()
    assert st["leap_rank"] > 0 and st["fallback_rounds"] > 0 and st["leap_state"] == "paused"
    np.testing.assert_array_equal(I.cpu().numpy(), I0)
    np.testing.assert_array_equal(D.cpu().numpy().view(np.uint32), D0.view(np.uint32))
    for i in range(3):
        ix.search_device(tq, k)
        assert ix.last_stats()["leap_rank"] == 0
    # new rows: the pause belongs to the old ones.  (Rows in random order behind the sorted ones: these queries' best rows all
    # sit in the first part, the leap holds or not -- either way the result is the ordinary rounds')
    more = rng.standard_normal((100000, 128)).astype(np.float16)
    ix.add(more)
    D, I = ix.search_device(tq, k)
    assert ix.last_stats()["leap_rank"] > 0
    D2, I2, _, _ = _search(np.concatenate([xb, more]), xq, k, "off", nomination)
    np.testing.assert_array_equal(I.cpu().numpy(), I2)
    np.testing.assert_array_equal(D.cpu().numpy().view(np.uint32), D2.view(np.uint32))


def _planted(rng, n, nq, k, which):
    """random rows; for the queries `which`, k rows among the first few hundred score far above everything else: their
    running lists are final after the bootstrap, no later row beats the score at rank j -- those queries end every leaping round
    short, all others are verified"""
    xb = rng.standard_normal((n, 128)).astype(np.float32)
    xq = rng.standard_normal((nq, 128)).astype(np.float32)
    for i, q in enumerate(which):
        r0 = 50 + 100 * i
        # (1.2 x the query: ~150 against a best ordinary score of ~40, and few OTHER queries score high on these rows)
        xb[r0:r0 + k] = 1.2 * xq[q] + 0.02 * rng.standard_normal((k, 128))
    return xb.astype(np.float16), xq.astype(np.float16)


@pytest.mark.parametrize("nq,nomination,which", [(600, "always", (3, 17, 599)), (40, "always", (0,)), (600, "off", (5, 300)),
                                                 (2000, "auto", tuple(range(0, 2000, 250)))])
def test_a_few_short_queries_are_searched_again_by_themselves(gpu_device, nq, nomination, which):
    rng = np.random.default_rng(nq)
    n, k = 300000, 80
    xb, xq = _planted(rng, n, nq, k, which)
    D0, I0, st0, _ = _search(xb, xq, k, "off", nomination)
    D1, I1, st1, ix = _search(xb, xq, k, "auto", nomination)
    # (a shortfall this cheap does not pause the leaps at once: three strikes each, a pause at eight)
    assert st1["leap_rank"] > 0 and st1["fallback_rounds"] > 0 and st1["leap_state"] == "on", st1
    # (the short queries alone ran again -- a small batch on ordinary rounds -- not the flagged slabs for all queries, which is
    # four quarters per flagged round)
    assert st1["fallback_rounds"] < 4 * st1["rounds"], st1
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1.view(np.uint32), D0.view(np.uint32))
    for q in which:
        assert set(I1[q]) == set(range(50 + 100 * which.index(q), 50 + 100 * which.index(q) + k))
    import torch
    tq = torch.from_numpy(xq).cuda()
    states = []
    for _ in range(3):
        D, I = ix.search_device(tq, k)
        st = ix.last_stats()
        states.append((st["leap_rank"] > 0, st["leap_state"]))
        np.testing.assert_array_equal(I.cpu().numpy(), I0)
    # strikes 3, 6, 9 -> the third shortfall pauses; the search after it takes ordinary rounds
    assert states == [(True, "on"), (True, "paused"), (False, "paused")], states


def test_the_deferred_search_searches_its_short_queries_again(gpu_device):
    import torch
    from proqa_amd.index import PipelinedSearcher
    rng = np.random.default_rng(8)
    xb, xq = _planted(rng, 300000, 512, 80, (1, 100, 257, 511))
    D0, I0, _, _ = _search(xb, xq, 80, "off")
    ps = PipelinedSearcher(torch.from_numpy(xb).cuda())
    outs = list(ps.search_batches([torch.from_numpy(xq[:256]).cuda(), torch.from_numpy(xq[256:]).cuda()], 80))
    np.testing.assert_array_equal(np.concatenate([o[1].cpu().numpy() for o in outs]), I0)
    np.testing.assert_array_equal(np.concatenate([o[0].cpu().numpy() for o in outs]).view(np.uint32), D0.view(np.uint32))
    assert all(st["fallback_rounds"] > 0 for st in ps.last_stats())
    ps.close()


@pytest.mark.parametrize("nq,k,rank,nomination", [(600, 80, 2, "always"), (40, 80, 3, "always"), (300, 128, 1, "off"), (600, 16, 2, "auto"),
                                                  (1, 80, 2, "always")])
def test_thresholds_at_a_rank_far_too_high_are_caught_by_the_merge(gpu_device, monkeypatch, nq, k, rank, nomination):
    """PROQA_LEAP_RANK (developer switch, read at every search): thresholds at rank 1-3 -- almost no row beats them, most
    queries end a round with fewer than k rows above their threshold, many with none at all (the merge's early exit)."""
    rng = np.random.default_rng(nq + k)
    n = 300000
    xb = rng.standard_normal((n, 128)).astype(np.float16)
    xq = rng.standard_normal((nq, 128)).astype(np.float16)
    D0, I0, st0, _ = _search(xb, xq, k, "off", nomination)
    monkeypatch.setenv("PROQA_LEAP_RANK", str(rank))
    monkeypatch.setenv("PROQA_LEAP_ROUNDS", "3")
    D1, I1, st1, _ = _search(xb, xq, k, "auto", nomination)
    assert st1["leap_rank"] == rank and st1["fallback_rounds"] > 0, st1
    assert st1["leap_state"] == ("paused" if nq > 256 else "on")   # (<= 256 short queries: searched again by themselves, a cheap shortfall)
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1.view(np.uint32), D0.view(np.uint32))


@pytest.mark.parametrize("n,nq,k", [(200000, 600, 80), (150000, 50, 20)])
def test_integer_corpora_with_leaps_are_the_oracles(gpu_device, n, nq, k):
    """integer scores tie in thousands: k rows rarely BEAT the score at rank j -- leaps fall short, the re-scan is exact"""
    rng = np.random.default_rng(n + nq)
    xb = rng.integers(-4, 5, (n, 128)).astype(np.float16)
    xq = rng.integers(-4, 5, (nq, 128)).astype(np.float16)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    for nomination in ("always", "off"):
        D1, I1, st1, _ = _search(xb, xq, k, "auto", nomination)
        assert st1["leap_rank"] > 0
        np.testing.assert_array_equal(I1, Io)
        np.testing.assert_array_equal(D1, Do)


def test_a_configured_growth_keeps_its_ordinary_rounds(gpu_device):
    rng = np.random.default_rng(2)
    xb = rng.standard_normal((200000, 128)).astype(np.float16)
    xq = rng.standard_normal((300, 128)).astype(np.float16)
    _, _, st, _ = _search(xb, xq, 80, "auto", growth=3)
    assert st["leap_rank"] == 0


def test_the_deferred_search_rescans_a_leap_that_fell_short(gpu_device):
    """PipelinedSearcher: begin / finish on two handles -- the shortfall is found by _finish, which re-scans"""
    import torch
    from proqa_amd.index import PipelinedSearcher
    rng = np.random.default_rng(4)
    xb, xq = _sorted_rows(rng, 300000, 512)
    D0, I0, _, _ = _search(xb, xq, 80, "off")
    ps = PipelinedSearcher(torch.from_numpy(xb).cuda())
    outs = list(ps.search_batches([torch.from_numpy(xq[:256]).cuda(), torch.from_numpy(xq[256:]).cuda()], 80))
    I = np.concatenate([o[1].cpu().numpy() for o in outs])
    D = np.concatenate([o[0].cpu().numpy() for o in outs])
    np.testing.assert_array_equal(I, I0)
    np.testing.assert_array_equal(D.view(np.uint32), D0.view(np.uint32))
    ps.close()


def test_the_leaps_are_logged(gpu_device):
    """PROQA_LOG=1: one stderr line for the plan, one when a leap falls short"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import numpy as np, torch
from proqa_amd.index import IndexFlatIP
rng = np.random.default_rng(11)
u = rng.standard_normal(128); u /= np.linalg.norm(u)
xb = rng.standard_normal((200000, 128)).astype(np.float32)
xb = xb[np.argsort(-(xb @ u), kind="stable")].astype(np.float16)
xq = (6.0 * u[None, :] + 0.3 * rng.standard_normal((64, 128))).astype(np.float16)
ix = IndexFlatIP(128); ix.add(xb)
ix.search_device(torch.from_numpy(xq).cuda(), 80)
print(ix.last_stats())
"""
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PROQA_LOG="1", PYTHONPATH=root), capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert "leaping rounds behind a bootstrap" in p.stderr and "a leaping round found fewer than k rows" in p.stderr, p.stderr
    # 64 short queries of 64: searched again by themselves; with PROQA_LEAP_RESCUE_MAX=0 the flagged slabs are re-scanned
    assert "searched again on ordinary rounds" in p.stderr and "leaps go on (3 strikes of 8)" in p.stderr, p.stderr
    q = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PROQA_LOG="1", PROQA_LEAP_RESCUE_MAX="0", PYTHONPATH=root),
                       capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr
    assert "ordinary rounds for the next 16 searches" in q.stderr and "searched again" not in q.stderr, q.stderr
