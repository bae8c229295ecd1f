"""Leaping rounds, the arithmetic (CPU): proqa_leap_plan is the planner proqa_index_configure_leap's automatic mode runs.  Its
rank must be the smallest whose shortfall probability -- P(fewer than k - j of the new rows beat the score at rank j of the
rows seen) -- stays below 1e-8 under the negative-binomial law the design states, and that law must be what exchangeable
rows actually do (a simulation on ranks alone: no scores, no GPU).

The search these plans are for replaces /root/reference/retrieval/eval_retrieval.py:102-104 (`IndexFlatIP.search`)."""
import ctypes

import numpy as np
import pytest
from scipy import stats

from proqa_amd import _lib


def _plan(rows, boot, k, queries, nominating=True):
    lib = _lib.load()
    rounds, rank = ctypes.c_int(), ctypes.c_int()
    per_round, p = ctypes.c_double(), ctypes.c_double()
    _lib.check(lib.proqa_leap_plan(rows, boot, k, queries, 1 if nominating else 0, ctypes.byref(rounds), ctypes.byref(rank),
                                   ctypes.byref(per_round), ctypes.byref(p)))
    return rounds.value, rank.value, per_round.value, p.value


@pytest.mark.parametrize("rows,boot,k,queries,nominating", [(18_000_000, 8192, 80, 2032, True), (2_250_000, 8192, 80, 2032, True),
                                                            (18_000_000, 8192, 80, 1, True), (2_250_000, 8192, 80, 32, True),
                                                            (1_000_000, 4096, 10, 300, False), (400_000, 8192, 128, 2000, True),
                                                            (66_000, 8192, 80, 700, False)])
def test_the_rank_is_the_smallest_one_under_the_negative_binomial_law(rows, boot, k, queries, nominating):
    rounds, rank, per_round, p = _plan(rows, boot, k, queries, nominating)
    assert rounds >= 1 and 0 < rank < k
    rho = (rows / boot) ** (1.0 / rounds)
    assert per_round == pytest.approx(rank * (rho - 1.0), rel=1e-9)
    # scipy: failures before the j-th success, success probability 1 / rho; a shortfall is fewer than k - j of them
    law = lambda j: stats.nbinom.cdf(k - j - 1, j, 1.0 / rho)   # noqa: E731
    assert p == pytest.approx(law(rank), rel=1e-6) and p <= 1e-8
    assert law(rank - 1) > 1e-8                                  # one rank lower would not do
    # what the round's merge must hold at five sigma (x 2.5 for the int8 scan's over-nomination)
    spread = 1.0 + 5.0 / np.sqrt(rank)
    assert per_round * spread * (2.5 if nominating else 1.0) <= 2048


def test_the_measured_schedules():
    """the round counts the interleaved sweeps found fastest (profiles/ABLATIONS.md R6.13)"""
    assert _plan(18_000_000, 8192, 80, 2032)[:2] == (4, 33)
    assert _plan(2_250_000, 8192, 80, 2032)[:2] == (3, 34)
    assert _plan(18_000_000, 8192, 80, 1)[0] == 3 and _plan(2_250_000, 8192, 80, 1)[0] == 2
    # no leap where the design says so
    assert _plan(18_000_000, 8192, 200, 2032)[0] == 0 and _plan(18_000_000, 8192, 4, 2032)[0] == 0
    assert _plan(8192, 8192, 80, 100)[0] == 0
    lib = _lib.load()
    assert lib.proqa_leap_plan(0, 8192, 80, 1, 1, None, None, None, None) != 0


@pytest.mark.parametrize("k,j,rho", [(80, 20, 5.0), (80, 16, 8.0), (16, 8, 4.0), (128, 31, 6.0)])
def test_exchangeable_rows_follow_that_law(k, j, rho):
    """Ranks only: n0 rows seen, (rho - 1) n0 to come, all exchangeable.  The number of new rows that beat the j-th best of the
    rows seen is the number of new rows among the overall best before the j-th old one appears -- draw the origin (old / new)
    of the best rows one by one (without replacement: hypergeometric, which the negative binomial approximates for n0 >> k).
    The shortfall frequency must match the closed sum the planner uses; ranks with probabilities around 1e-2 so that 200 000
    trials can see them."""
    rng = np.random.default_rng(k * 1000 + j)
    n0, trials, depth = 8192, 200_000, 8 * k
    n1 = int((rho - 1.0) * n0)
    # origin of the `depth` best rows of n0 + n1 exchangeable rows: old with probability (old left) / (all left)
    old_left = np.full(trials, n0, dtype=np.int64)
    all_left = np.full(trials, n0 + n1, dtype=np.int64)
    old_seen = np.zeros(trials, dtype=np.int64)
    new_before_jth_old = np.zeros(trials, dtype=np.int64)
    for _ in range(depth):
        is_old = rng.random(trials) * all_left < old_left
        counting = old_seen < j
        new_before_jth_old += counting & ~is_old
        old_seen += is_old
        old_left -= is_old
        all_left -= 1
    assert (old_seen >= j).all()                       # deep enough: every trial reached its j-th old row
    short = float(np.mean(new_before_jth_old < k - j))
    law = float(stats.nbinom.cdf(k - j - 1, j, n0 / (n0 + n1)))
    assert 1e-3 < law < 0.2
    assert short == pytest.approx(law, rel=0.12, abs=4 * np.sqrt(law / trials))
