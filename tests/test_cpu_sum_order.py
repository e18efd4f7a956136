"""How much of the parity claim rests on the un-pinned float summation order ("what's weak" 1).

The reference computes every squared distance with Eigen's `squaredNorm()` (treequantizer.hpp:197,655,
vectorquantizer.hpp:109); Eigen is not vendored, so its reduction order is not fixed by anything the reference holds.
The oracle (and the HIP kernels) use the literal sequential order (sum mode 0).  Here the oracle's sum-order probes
(oracle/pqt_oracle.cpp sqdist(): the SSE2-packet orders Eigen would use under the reference's own build flags, an AVX order
and an FMA-contracted loop) are run against mode 0 on the BASELINE-shaped fixtures and the agreement is MEASURED and asserted:

  * query side (same index, different query arithmetic): candidate SETS, visiting sequences and top-100 id lists;
  * build side (insert = id() + prepareReranking under the other order): bin ids and line codes of the database vectors.

A last-ulp difference can only change a candidate set by swapping two bins across the boundVectors cut or two cells across
the W-best cut; the numbers below say how often that happens on these data (printed with -s).
"""
import numpy as np
import pytest

from common import CONFIGS, fixture
from oracle import Oracle

BV_BB = {"cfg2_small": [(300, 500), (20000, 500)], "cfg3_small": [(400, 500), (4096, 512)], "tools_default": [(2000, 500)]}
MODES = {1: "sse2 linear2", 2: "sse2 tree", 3: "sse2 linear1", 4: "avx linear2", 5: "sequential+fma"}
# asserted bounds (measured: 1.0 everywhere on these fixtures; the bounds leave room for a boundary swap on other seeds)
MIN_SET_AGREEMENT = 0.95
MIN_BIN_AGREEMENT = 0.995
MIN_CODE_AGREEMENT = 0.98


def agreement(o, queries, bv, bb, mode):
    o.set_sort_mode(1)
    try:
        o.set_sum_mode(0)
        ref_u = [o.query_unsorted(q, bv, bb)[0] for q in queries]
        ref_s = [o.query(q, bv, bb)[0][:100] for q in queries]
        o.set_sum_mode(mode)
        got_u = [o.query_unsorted(q, bv, bb)[0] for q in queries]
        got_s = [o.query(q, bv, bb)[0][:100] for q in queries]
    finally:
        o.set_sum_mode(0)
        o.set_sort_mode(0)
    n = len(queries)
    same_set = sum(set(a.tolist()) == set(b.tolist()) for a, b in zip(ref_u, got_u)) / n
    same_seq = sum(np.array_equal(a, b) for a, b in zip(ref_u, got_u)) / n
    same_top = sum(np.array_equal(a, b) for a, b in zip(ref_s, got_s)) / n
    return same_set, same_seq, same_top


@pytest.mark.parametrize("name", list(BV_BB))
def test_candidate_sets_under_other_sum_orders(name):
    f = fixture(name)
    for bv, bb in BV_BB[name]:
        for mode, label in MODES.items():
            same_set, same_seq, same_top = agreement(f.oracle, f.queries, bv, bb, mode)
            print("%s query(%d,%d) sum order %d (%s): candidate sets identical %.3f, visiting sequence %.3f, top-100 ids %.3f"
                  % (name, bv, bb, mode, label, same_set, same_seq, same_top))
            assert same_set >= MIN_SET_AGREEMENT, (name, bv, bb, mode)


def test_tables_really_differ_between_sum_orders():
    """The probe is not vacuous: a large share of the L1virt entries change in their last bits."""
    f = fixture("cfg2_small")
    o = f.oracle
    base = o.stage_l1(f.queries[0])[0].copy()
    try:
        for mode in MODES:
            o.set_sum_mode(mode)
            v = o.stage_l1(f.queries[0])[0]
            frac = float((v.view(np.uint32) != base.view(np.uint32)).mean())
            assert 0.05 < frac < 0.9, (mode, frac)
            assert np.allclose(v, base, rtol=1e-6)
    finally:
        o.set_sum_mode(0)


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_small"])
def test_database_build_under_other_sum_orders(name):
    """insert() under the SSE2 order: which vectors land in another bin / get another line code."""
    f = fixture(name)
    c = CONFIGS[name]
    n = 4000
    for mode in (1, 5):
        o2 = Oracle(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], heur_keep=1)
        o2.set_codebooks(f.cb1, f.cb2)
        o2.set_sum_mode(mode)
        o2.insert(f.base[:n])
        codes2 = o2.export_codes()
        bins2 = np.array([o2.bin_id(v) for v in f.base[:n]], np.uint32)
        bins0 = np.array([f.oracle.bin_id(v) for v in f.base[:n]], np.uint32)
        bin_same = float((bins2 == bins0).mean())
        code_same = float((codes2 == f.codes[:n]).all(1).mean())
        word_same = float((codes2 == f.codes[:n]).mean())
        ab_same = float(((codes2 & 0xffff) == (f.codes[:n] & 0xffff)).mean())
        print("%s insert under sum order %d: bin id identical %.4f, code rows identical %.4f, code words %.4f, (A,B) pairs %.4f"
              % (name, mode, bin_same, code_same, word_same, ab_same))
        assert bin_same >= MIN_BIN_AGREEMENT
        assert ab_same >= MIN_CODE_AGREEMENT
