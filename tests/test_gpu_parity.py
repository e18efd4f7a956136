"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact.  Tables, bin ids, candidate sequences, ADC distances and the sorted result lists must be
identical (f32 compared as raw bits).  Order comparisons use the oracle in its reference-faithful mode
(std::sort); when the engine reports exact float ties at a stage the oracle's stable mode is the canonical
order (see DESIGN.md "ties").
"""
import numpy as np
import pytest

from common import CONFIGS, fixture, pqt_pkg, sift_like

pytestmark = pytest.mark.gpu

BV_BB = {"tools_default": (2000, 500), "cfg2_small": (300, 500), "cfg2_dense": (1500, 800), "wrap": (100, 1000), "odd": (400, 144), "ties": (500, 400), "big_coarse": (600, 64), "cfg3_small": (400, 500)}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module", params=[(c, m) for c in CONFIGS for m in ("fused", "staged")], ids=lambda p: "%s-%s" % p)
def pair(request):
    """Every test runs on both launch structures: the wave-per-query fused kernels and the staged kernels."""
    name, mode = request.param
    f = fixture(name)
    idx = f.hip_index()
    idx.set_option("fused", 1 if mode == "fused" else 0)
    yield name, f, idx
    idx.close()


def test_library_loaded_and_device():
    pkg = pqt_pkg()
    assert pkg.lib().pqt_device_count() >= 1


def test_triangle_known_answers_on_device():
    """run.cu:33-104 -- the reference's only golden vectors, evaluated by a kernel."""
    pkg = pqt_pkg()
    a2 = np.array([1, 2, 2, 2, 2, 5], np.float32)
    b2 = np.array([2, 2, 2, 5, 5, 2], np.float32)
    c2 = np.array([1, 4, 2, 9, 1, 1], np.float32)
    lam_exp = np.array([1, .5, .5, 2 / 3, 2, -1], np.float32)
    d_exp = np.array([1, 1, 1.5, 1, 1, 1], np.float32)
    _, ratio, _, _ = pkg.dev_triangle(a2, b2, c2, np.zeros(6, np.float32))
    assert np.all(np.abs(ratio - lam_exp) < 1e-5)  # equal(): triangle.cuh:112 tolerance
    dist, _, _, _ = pkg.dev_triangle(a2, b2, c2, ratio)
    assert np.all(np.abs(dist - d_exp) < 1e-5)
    # lambda codec sweep of run.cu:106-113 against the oracle's codec (bit exact)
    from oracle.oracle import _lib
    L = _lib()
    f = (np.arange(-100, 100) / 10.0).astype(np.float32)
    _, _, u16, rt = pkg.dev_triangle(f, f, f, f)
    for i, x in enumerate(f):
        assert int(u16[i]) == L.pqo_lambda_encode(float(x))
        assert bits(rt[i:i + 1])[0] == bits(np.array([L.pqo_lambda_decode(int(u16[i]))], np.float32))[0]


def test_coarse_table_bit_exact(pair):
    _, f, idx = pair
    assert np.array_equal(bits(idx.coarse()), bits(f.oracle.coarse()))


def test_stage_tables_bit_exact(pair):
    name, f, idx = pair
    Bv, Bb = BV_BB[name]
    idx.set_option("fused", 0)  # the fused traversal keeps the segment lists on chip; the staged kernels expose them
    idx.query(f.queries, Bv, Bb, 10)
    st = idx.stats()
    dbg = idx.debug_read(len(f.queries), cands=False)
    o = f.oracle
    for qi, q in enumerate(f.queries):
        virt, _, _ = o.stage_l1(q)
        assert np.array_equal(bits(dbg["l1virt"][qi]), bits(virt)), "L1virt differs q=%d" % qi
        l1, l2, _, d2, order = o.stage_segments(q)
        if st["ties_l1"] == 0 and st["ties_l2"] == 0:
            for p in range(o.P):
                srt = order[p]
                assert np.array_equal(bits(dbg["seg_d2"][qi, p]), bits(d2[p][srt]))
                assert np.array_equal(dbg["seg_bin"][qi, p], (l1[p][srt] * o.C2 + l2[p][srt]).astype(np.uint32))


def _oracle_lists(f, q, Bv, Bb, stable):
    o = f.oracle
    o.set_sort_mode(1 if stable else 0)
    try:
        u_ids, u_d = o.query_unsorted(q, Bv, Bb)
        s_ids, s_d = o.query(q, Bv, Bb)
    finally:
        o.set_sort_mode(0)
    return u_ids, u_d, s_ids, s_d


def test_candidates_and_full_sorted_list(pair):
    """Candidate index sequence (visiting order), ADC distances, and the fully sorted list == oracle."""
    name, f, idx = pair
    Bv, Bb = BV_BB[name]
    qn = len(f.queries)
    kfull = 8192  # > 4096 -> full-sort path; larger than any candidate list of these fixtures
    ids, dist, cnt = idx.query(f.queries, Bv, Bb, kfull)
    st = idx.stats()
    dbg = idx.debug_read(qn, segs=False)  # (the segment lists stay on chip in the fused traversal)
    any_ties = st["ties_l1"] or st["ties_l2"] or st["ties_bins"]
    total = 0
    for qi, q in enumerate(f.queries):
        u_ids, u_d, s_ids, s_d = _oracle_lists(f, q, Bv, Bb, stable=bool(any_ties))
        n = int(cnt[qi])
        total += n
        assert n == len(u_ids), "candidate count differs q=%d: %d vs %d" % (qi, n, len(u_ids))
        assert n <= kfull
        assert np.array_equal(dbg["cand_idx"][qi, :n], u_ids), "candidate sequence differs q=%d" % qi
        assert np.array_equal(bits(dbg["cand_dist"][qi, :n]), bits(u_d)), "ADC distances differ q=%d" % qi
        # sorted list: distances must agree exactly; ids must agree wherever the distance is unique
        assert np.array_equal(bits(dist[qi, :n]), bits(s_d))
        if st["ties_final"] == 0 and not any_ties:
            assert np.array_equal(ids[qi, :n], s_ids)
        else:
            so = f.oracle
            so.set_sort_mode(1)
            try:
                st_ids, _ = so.query(q, Bv, Bb)
            finally:
                so.set_sort_mode(0)
            assert np.array_equal(ids[qi, :n], st_ids)
        assert np.all(ids[qi, n:] == 0xffffffff) and np.all(np.isinf(dist[qi, n:]))
    assert st["candidates"] == total


@pytest.mark.parametrize("k", [1, 10, 100, 129, 1000, 4096])
def test_topk_select_path(pair, k):
    name, f, idx = pair
    Bv, Bb = BV_BB[name]
    ids, dist, cnt = idx.query(f.queries, Bv, Bb, k)
    st = idx.stats()
    stable = bool(st["ties_l1"] or st["ties_l2"] or st["ties_bins"] or st["ties_final"])
    for qi, q in enumerate(f.queries):
        f.oracle.set_sort_mode(1 if stable else 0)
        try:
            s_ids, s_d = f.oracle.query(q, Bv, Bb)
        finally:
            f.oracle.set_sort_mode(0)
        assert int(cnt[qi]) == len(s_ids)
        kk = min(k, len(s_ids))
        assert np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk]))
        # with ties beyond position k the select path must still return the stable-first ones
        f.oracle.set_sort_mode(1)
        try:
            st_ids, _ = f.oracle.query(q, Bv, Bb)
        finally:
            f.oracle.set_sort_mode(0)
        assert np.array_equal(ids[qi, :kk], st_ids[:kk])
        assert np.all(ids[qi, kk:] == 0xffffffff)


@pytest.mark.parametrize("bv,bb", [(0, 1), (0, 7), (1, 64), (10 ** 9, 64), (50, 0)])
def test_edge_bounds(pair, bv, bb):
    """Bv = 0 (first non-empty bin still taken: strict '>'), Bb = 1, Bb = 0, Bv beyond everything."""
    name, f, idx = pair
    bb = min(bb, len(f.heur))
    bvq = min(bv, 2 ** 31)
    qs = f.queries[:8]
    ids, dist, cnt = idx.query(qs, bvq, bb, 6000)
    for qi, q in enumerate(qs):
        f.oracle.set_sort_mode(1)
        try:
            s_ids, s_d = f.oracle.query(q, bvq, bb)
        finally:
            f.oracle.set_sort_mode(0)
        n = int(cnt[qi])
        assert n == len(s_ids)
        m = min(n, 6000)
        assert np.array_equal(ids[qi, :m], s_ids[:m])
        assert np.array_equal(bits(dist[qi, :m]), bits(s_d[:m]))


def test_single_query_and_ragged_batches(pair):
    name, f, idx = pair
    Bv, Bb = BV_BB[name]
    full_ids, full_d, full_c = idx.query(f.queries, Bv, Bb, 50)
    for qn in (1, 3, 17):
        ids, d, c = idx.query(f.queries[:qn], Bv, Bb, 50)
        assert np.array_equal(ids, full_ids[:qn]) and np.array_equal(bits(d), bits(full_d[:qn])) and np.array_equal(c, full_c[:qn])


def test_heuristic_built_by_library_matches_oracle(pair):
    """a3 at every fixture shape, the BASELINE ones included ((W*C2)^P = 64^4 tuples for cfg2 and cfg3): the prefix the
    library builds (pqt_index_build_heuristic) == the prefix the oracle's own prepareHeuristic restatement built."""
    name, f, idx = pair
    if f.oracle.max_multi_index > (1 << 26):
        pytest.skip("full tuple space too large for a unit test")
    rows = len(f.heur)
    idx.build_heuristic(rows)
    assert np.array_equal(idx.heuristic(rows), f.heur)
    idx.set_heuristic(f.heur)


def test_sharded_two_way_equals_unsharded(pair):
    """Range-shard the database over two handles (same GPU), query each shard, merge: identical to unsharded."""
    import torch
    name, f, idx = pair
    Bv, Bb = BV_BB[name]
    k = 64
    n = f.oracle.num_vectors
    cut = n // 3
    shards = [f.hip_index(shard=(0, cut)), f.hip_index(shard=(cut, n))]
    try:
        ref_ids, ref_d, ref_c = idx.query(f.queries, Bv, Bb, k)
        q = torch.from_numpy(f.queries).cuda()
        qn = q.shape[0]
        I = torch.empty((2, qn, k), dtype=torch.int32, device="cuda")
        Dd = torch.empty((2, qn, k), dtype=torch.float32, device="cuda")
        Pp = torch.empty((2, qn, k), dtype=torch.int32, device="cuda")
        Cc = torch.empty((2, qn), dtype=torch.int32, device="cuda")
        for runs in (0, 1):  # 1: bin runs (with the global positions of their first members) instead of candidate lists
            for s, sh in enumerate(shards):
                sh.set_option("bin_runs", runs)
                sh.query_shard_dev(q, Bv, Bb, k, I[s], Dd[s], Pp[s], Cc[s], sync=True)
            oI = torch.empty((qn, k), dtype=torch.int32, device="cuda")
            oD = torch.empty((qn, k), dtype=torch.float32, device="cuda")
            shards[0].merge_topk_dev(2, qn, k, I, Dd, Pp, oI, oD, sync=True)  # [shard][QN][k] layout (shard_stride 0 = QN*k)
            got_ids = oI.cpu().numpy().view(np.uint32)
            got_d = oD.cpu().numpy()
            assert np.array_equal(Cc[0].cpu().numpy().view(np.uint32), ref_c)  # every shard sees the global count
            assert np.array_equal(Cc[1].cpu().numpy().view(np.uint32), ref_c)
            assert np.array_equal(got_ids, ref_ids)
            assert np.array_equal(bits(got_d), bits(ref_d))
    finally:
        for sh in shards:
            sh.close()


def test_assign_encode_matches_oracle_insert(pair):
    """Offline row: bin ids and 4-byte line codes of the database vectors, bit exact (insert/prepareReranking)."""
    import torch
    name, f, idx = pair
    n = min(2000, f.base.shape[0])
    x = torch.from_numpy(f.base[:n]).cuda()
    ob = torch.empty(n, dtype=torch.int32, device="cuda")
    oc = torch.empty((n, f.cfg["LP"]), dtype=torch.int32, device="cuda")
    idx.assign_encode_dev(x, ob, oc)
    torch.cuda.synchronize()
    got_bins = ob.cpu().numpy().view(np.uint32)
    got_codes = oc.cpu().numpy().view(np.uint32)
    # oracle: bin of vector i from the exported bins
    bin_of = np.zeros(f.oracle.num_vectors, np.uint32)
    off = 0
    for b, s in zip(f.bin_ids, f.bin_sizes):
        bin_of[f.members[off:off + s]] = b
        off += s
    assert np.array_equal(got_bins, bin_of[:n])
    assert np.array_equal(got_codes, f.codes[:n])


def test_tie_fixture_really_has_ties_and_matches_stable_order():
    """The engine's tie order is the stable order: on a fixture with duplicated vectors and centroids
    the counters are non-zero and every list equals the oracle in stable mode."""
    f = fixture("ties")
    for mode in (1, 0):
        idx = f.hip_index()
        idx.set_option("fused", mode)
        try:
            ids, dist, cnt = idx.query(f.queries, 500, 400, 100)
            st = idx.stats()
            assert st["ties_final"] > 0 and st["ties_l2"] > 0
            f.oracle.set_sort_mode(1)
            try:
                for qi, q in enumerate(f.queries):
                    s_ids, s_d = f.oracle.query(q, 500, 400)
                    kk = min(100, len(s_ids))
                    assert int(cnt[qi]) == len(s_ids)
                    assert np.array_equal(ids[qi, :kk], s_ids[:kk])
                    assert np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk]))
            finally:
                f.oracle.set_sort_mode(0)
        finally:
            idx.close()


def test_empty_and_degenerate_inputs():
    """Empty database, zero queries, k larger than anything, a bin list with one vector."""
    pkg = pqt_pkg()
    f = fixture("odd")
    c = f.cfg
    idx = pkg.PqtIndex(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"])
    try:
        idx.set_codebooks(f.cb1, f.cb2)
        idx.set_heuristic(f.heur)
        with pytest.raises(pkg.PqtError):  # no bins / codes yet: loud failure, no silent empty result
            idx.query(f.queries[:2], 10, 10, 5)
        idx.set_bins(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32))
        idx.set_lines(np.zeros((0, c["LP"]), np.uint32))
        ids, dist, cnt = idx.query(f.queries[:3], 10, 50, 7)
        assert np.all(cnt == 0) and np.all(ids == 0xffffffff) and np.all(np.isinf(dist))
        ids, dist, cnt = idx.query(np.zeros((0, c["D"]), np.float32), 10, 50, 7)
        assert ids.shape == (0, 7)
        # one vector
        b = f.oracle.bin_id(f.base[0])
        idx.set_bins(np.array([b], np.uint32), np.array([1], np.uint32), np.array([0], np.uint32))
        idx.set_lines(f.codes[:1])
        ids, dist, cnt = idx.query(f.base[:1], 0, len(f.heur), 3)
        assert cnt[0] == 1 and ids[0, 0] == 0 and np.all(ids[0, 1:] == 0xffffffff)
    finally:
        idx.close()


def test_invalid_arguments_are_rejected():
    pkg = pqt_pkg()
    with pytest.raises(pkg.PqtError):
        pkg.PqtIndex(128, 3, 16, 8, 4, 32)      # dim % p != 0
    with pytest.raises(pkg.PqtError):
        pkg.PqtIndex(128, 2, 4, 8, 5, 32)       # w > c1
    f = fixture("cfg2_small")
    idx = f.hip_index()
    try:
        with pytest.raises(pkg.PqtError):
            idx.query(f.queries[:2], 10, len(f.heur) + 1, 5)  # bound_bins beyond the heuristic prefix held
        with pytest.raises(pkg.PqtError):
            idx.set_bins(np.array([5, 5], np.uint32), np.array([1, 1], np.uint32), np.array([0, 1], np.uint32))  # duplicate bin id
    finally:
        idx.close()


@pytest.mark.parametrize("raw_u8", [False, True])
@pytest.mark.parametrize("k", [10, 100, 300])
def test_exact_rerank_against_raw_vectors(k, raw_u8):
    """Row 8f-4: the first k approximate results re-ordered by exact squared L2 (f32, summed left to right), ties keep
    the previous order -- against a numpy restatement with the same summation order."""
    import torch
    f = fixture("tools_default")
    idx = f.hip_index()
    try:
        q = torch.from_numpy(f.queries).cuda()
        qn = q.shape[0]
        oi = torch.empty((qn, k), dtype=torch.int32, device="cuda")
        od = torch.empty((qn, k), dtype=torch.float32, device="cuda")
        idx.query_dev(q, 2000, 500, k, oi, od, sync=True) if k <= 4096 else None
        raw = torch.from_numpy(f.base.astype(np.uint8) if raw_u8 else f.base).cuda()
        ri = torch.empty_like(oi)
        rd = torch.empty_like(od)
        idx.rerank_exact_dev(q, k, oi, raw, ri, rd, sync=True)
        ids = oi.cpu().numpy().view(np.uint32)
        got_i = ri.cpu().numpy().view(np.uint32)
        got_d = rd.cpu().numpy()
        for qi in range(qn):
            valid = ids[qi] != 0xffffffff
            cand = ids[qi][valid]
            x = f.base[cand]
            s = np.zeros(len(cand), np.float32)
            for d in range(f.cfg["D"]):  # same left-to-right f32 accumulation as the kernel
                df = (f.queries[qi, d] - x[:, d]).astype(np.float32)
                s = (s + df * df).astype(np.float32)
            order = np.lexsort((np.arange(len(cand)), s))
            n = len(cand)
            assert np.array_equal(got_i[qi, :n], cand[order])
            assert np.array_equal(bits(got_d[qi, :n]), bits(s[order]))
            assert np.all(got_i[qi, n:] == 0xffffffff)
    finally:
        idx.close()


def test_bins_overflow_pass_dense_database():
    """Staged bins kernel: more populated bins than the first pass's LDS arena (1024) -> the query is redone by the
    full-size second pass; results still equal the oracle."""
    from common import Fixture
    # few, fat cells: almost every enumerated bin is populated; W*C2 = 48 -> 2304 tuples, all enumerated
    def uniform(n, D, seed):
        return np.random.default_rng(seed).integers(0, 256, (n, D)).astype(np.float32)

    f = Fixture(D=16, P=2, C1=8, C2=6, W=8, LP=4, n_base=30000, n_query=8, seed=91, heur_rows=2304, train=3000, data=uniform)
    idx = f.hip_index()
    try:
        idx.set_option("fused", 0)
        ids, dist, cnt = idx.query(f.queries, 10 ** 6, 2304, 200)  # Bv beyond the database: every populated bin is included
        st = idx.stats()
        assert st["bins_nonempty"] / len(f.queries) > 1024, "fixture no longer overflows the first pass"
        f.oracle.set_sort_mode(1)
        try:
            for qi, q in enumerate(f.queries):
                s_ids, s_d = f.oracle.query(q, 10 ** 6, 2304)
                kk = min(200, len(s_ids))
                assert int(cnt[qi]) == len(s_ids)
                assert np.array_equal(ids[qi, :kk], s_ids[:kk])
                assert np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk]))
        finally:
            f.oracle.set_sort_mode(0)
    finally:
        idx.close()


def test_small_scratch_budget_chunks_the_batch():
    """A 1 MiB candidate arena forces the batch through several chunks of queries: identical results."""
    f = fixture("tools_default")
    idx = f.hip_index()
    try:
        ref = idx.query(f.queries, 2000, 500, 100)
        for mode in (1, 0):
            idx.set_option("fused", mode)
            idx.set_option("scratch_mb", 1)   # stride ~3000 slots * 8 B -> ~40 queries per chunk
            got = idx.query(f.queries, 2000, 500, 100)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])) and np.array_equal(got[2], ref[2])
            big = idx.query(f.queries, 2000, 500, 5000)  # full-sort path, chunked too
            idx.set_option("scratch_mb", 4096)
            big_ref = idx.query(f.queries, 2000, 500, 5000)
            assert np.array_equal(big[0], big_ref[0]) and np.array_equal(bits(big[1]), bits(big_ref[1]))
    finally:
        idx.close()


@pytest.mark.parametrize("name", ["tools_default", "ties", "cfg2_small", "cfg3_small"])
def test_schedule_and_ordering_variants_change_nothing(name):
    """A batch large enough that every wavefront slot of the fused rerank gets several queries (the balancing order and
    the LDS tickets are active): identical results with the schedule off, with all rows ordered in the traversal, and
    equal to the small-batch results of the same queries."""
    f = fixture(name)
    bv, bb = BV_BB[name]
    idx = f.hip_index()
    try:
        small = idx.query(f.queries, bv, bb, 50)
        reps = (5000 + len(f.queries) - 1) // len(f.queries)
        big_q = np.tile(f.queries, (reps, 1))
        ref = idx.query(big_q, bv, bb, 50)
        n = len(f.queries)
        for r in range(reps):
            assert np.array_equal(ref[0][r * n:(r + 1) * n], small[0]) and np.array_equal(bits(ref[1][r * n:(r + 1) * n]), bits(small[1]))
            assert np.array_equal(ref[2][r * n:(r + 1) * n], small[2])
        st_ref = idx.stats()
        # "static_shapes" 0: the run-time-shape traversal instead of the compile-time instantiation of the two BASELINE shapes
        # "bin_runs" 1: the traversal hands bin runs to the rerank instead of materialising the candidate list
        for opt in ("balance", "order_all_rows", "static_shapes", "bin_runs"):
            idx.set_option(opt, 1 if opt in ("order_all_rows", "bin_runs") else 0)
            got = idx.query(big_q, bv, bb, 50)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])) and np.array_equal(got[2], ref[2])
            st = idx.stats()
            for key in ("candidates", "bins_visited", "bins_nonempty", "ties_l1", "ties_l2", "ties_final"):
                assert st[key] == st_ref[key], key
            idx.set_option(opt, 0 if opt in ("order_all_rows", "bin_runs") else 1)
    finally:
        idx.close()


@pytest.mark.parametrize("bb", [513, 700, 1300, 2304])
@pytest.mark.parametrize("bv", [300, 10 ** 6])
def test_wide_enumeration_in_the_fused_traversal(bb, bv):
    """boundBins in (512, 4096]: the fused traversal enumerates the rows in blocks of 512 and orders the populated ones
    (list of <= 512; queries with more are handed to the workgroup-per-query bins kernel).  A dense fixture puts the
    queries in all three regimes (<= 128 populated rows, <= 512, overflow); results, counts and statistics equal the
    staged kernels' and the oracle's."""
    from common import Fixture

    def uniform(n, D, seed):
        return np.random.default_rng(seed).integers(0, 256, (n, D)).astype(np.float32)

    # W*C2 = 48 -> 2304 tuples; 30000 vectors over <= 48^2 cells of which a query enumerates the bb nearest
    f = Fixture(D=16, P=2, C1=8, C2=6, W=8, LP=4, n_base=1500 if bb < 1000 else 30000, n_query=24, seed=92, heur_rows=2304, train=1500, data=uniform)
    idx = f.hip_index()
    try:
        idx.set_option("fused", 1)
        got = idx.query(f.queries, bv, bb, 100)
        st = idx.stats()
        idx.set_option("fused", 0)
        ref = idx.query(f.queries, bv, bb, 100)
        st_ref = idx.stats()
        assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])) and np.array_equal(got[2], ref[2])
        for key in ("candidates", "bins_visited", "bins_nonempty", "ties_l1", "ties_l2", "ties_final"):
            assert st[key] == st_ref[key], key
        f.oracle.set_sort_mode(1)
        try:
            for qi, q in enumerate(f.queries[:6]):
                s_ids, s_d = f.oracle.query(q, bv, bb)
                kk = min(100, len(s_ids))
                assert int(got[2][qi]) == len(s_ids)
                assert np.array_equal(got[0][qi, :kk], s_ids[:kk]) and np.array_equal(bits(got[1][qi, :kk]), bits(s_d[:kk]))
        finally:
            f.oracle.set_sort_mode(0)
    finally:
        idx.close()


@pytest.mark.parametrize("name", ["tools_default", "wrap"])
def test_wide_enumeration_sharded_equals_unsharded(name):
    """Two range shards with boundBins = 1000 (wide fused traversal, sharded variant): merged result identical to the
    unsharded engine."""
    import torch
    f = fixture(name)
    Bv, Bb, k = 2000, 1000, 64
    n = f.oracle.num_vectors
    cut = n // 3
    idx = f.hip_index()
    shards = [f.hip_index(shard=(0, cut)), f.hip_index(shard=(cut, n))]
    try:
        ref_ids, ref_d, ref_c = idx.query(f.queries, Bv, Bb, k)
        q = torch.from_numpy(f.queries).cuda()
        qn = q.shape[0]
        I = torch.empty((2, qn, k), dtype=torch.int32, device="cuda")
        Dd = torch.empty((2, qn, k), dtype=torch.float32, device="cuda")
        Pp = torch.empty((2, qn, k), dtype=torch.int32, device="cuda")
        Cc = torch.empty((2, qn), dtype=torch.int32, device="cuda")
        for s_, sh in enumerate(shards):
            sh.query_shard_dev(q, Bv, Bb, k, I[s_], Dd[s_], Pp[s_], Cc[s_], sync=True)
        oI = torch.empty((qn, k), dtype=torch.int32, device="cuda")
        oD = torch.empty((qn, k), dtype=torch.float32, device="cuda")
        shards[0].merge_topk_dev(2, qn, k, I, Dd, Pp, oI, oD, sync=True)
        assert np.array_equal(Cc[0].cpu().numpy().view(np.uint32), ref_c)
        assert np.array_equal(oI.cpu().numpy().view(np.uint32), ref_ids)
        assert np.array_equal(bits(oD.cpu().numpy()), bits(ref_d))
    finally:
        idx.close()
        for sh in shards:
            sh.close()


@pytest.mark.parametrize("name,bb", [("tools_default", 500), ("odd", 144), ("tools_default", 900)])
def test_hashed_db_triple_without_aliasing_equals_exact_bins(name, bb):
    """The CUDA tools' database triple (prefix, counts, dbIdx over HASH_SIZE slots, PerturbationProTree.hh:66) with a
    hash size above every bin id (slot == bin id, nothing aliases) must give the results of the exact-key bin table:
    covers the `% HASH_SIZE` path of the traversal, its presence bitmap and the wide enumeration."""
    f = fixture(name)
    bv = BV_BB[name][0]
    pkg = pqt_pkg()
    c = f.cfg
    max_id = int(np.max(f.bin_ids)) if len(f.bin_ids) else 0
    hash_size = max_id + 17
    assert hash_size < 2 ** 31
    prefix = np.zeros(hash_size, np.uint32)
    counts = np.zeros(hash_size, np.uint32)
    starts = np.concatenate([[0], np.cumsum(f.bin_sizes)[:-1]]).astype(np.uint32) if len(f.bin_sizes) else np.zeros(0, np.uint32)
    prefix[f.bin_ids] = starts
    counts[f.bin_ids] = f.bin_sizes
    ref_idx = f.hip_index()
    idx = pkg.PqtIndex(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], device=0)
    try:
        idx.set_codebooks(f.cb1, f.cb2)
        idx.set_heuristic(f.heur)
        idx.set_db_hashed(prefix, counts, f.members, hash_size)
        idx.set_lines(f.codes)
        ref = ref_idx.query(f.queries, bv, bb, 64)
        for mode in (1, 0):
            idx.set_option("fused", mode)
            got = idx.query(f.queries, bv, bb, 64)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])) and np.array_equal(got[2], ref[2])
    finally:
        idx.close()
        ref_idx.close()


@pytest.mark.parametrize("LP", [4, 8, 16])
def test_improving_candidates_fill_the_pending_buffer(LP):
    """with 4 or 8 line parts the fused rerank+select appended up to 512 keys per batch behind a best
    list of k keys in a 512-slot area.  Worst case for the pending buffer: more than 1024 candidates whose distances
    DEcrease in visiting order (every candidate beats tau).  Two big bins, members re-ordered by descending ADC distance
    to query 0 (any member order is a valid bin list); results must equal the oracle's for every k."""
    from common import Fixture
    f = Fixture(D=16, P=1, C1=4, C2=2, W=1, LP=LP, n_base=9000, n_query=6, seed=91, heur_rows=2, train=1500)
    o = f.oracle
    u_ids, u_d = o.query_unsorted(f.queries[0], 10 ** 6, 2)
    assert len(u_ids) > 1024
    rank = {int(v): -float(d) for v, d in zip(u_ids, u_d)}  # descending distance first
    members = f.members.copy()
    off = 0
    for sz in f.bin_sizes.tolist():
        seg = members[off:off + sz]
        if int(seg[0]) in rank:
            members[off:off + sz] = np.array(sorted(seg.tolist(), key=lambda v: (rank[v], v)), np.uint32)
        off += sz
    o.import_bins(f.bin_ids, f.bin_sizes, members)
    f.members = members
    idx = f.hip_index()
    try:
        o.set_sort_mode(1)
        u2, d2 = o.query_unsorted(f.queries[0], 10 ** 6, 2)
        first = int(f.bin_sizes[list(f.bin_ids).index(o.bin_id(f.base[int(u2[0])]))])
        assert np.all(np.diff(d2[:first]) <= 0), "fixture: first bin not in descending-distance order"
        for k in (1, 37, 100, 128):
            ids, dist, cnt = idx.query(f.queries, 10 ** 6, 2, k)
            for qi, q in enumerate(f.queries):
                s_ids, s_d = o.query(q, 10 ** 6, 2)
                n = min(k, len(s_ids))
                assert int(cnt[qi]) == len(s_ids)
                assert np.array_equal(bits(dist[qi, :n]), bits(s_d[:n])), (LP, k, qi)
                assert np.array_equal(ids[qi, :n], s_ids[:n]), (LP, k, qi)
    finally:
        o.set_sort_mode(0)
        idx.close()


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_small", "big_coarse"])
def test_adc_bias_mode_same_sets_distances_by_its_own_formula(name):
    """Opt-in pqt_index_set_option("adc_bias", 1) (SURVEY App. C "E-alt"): distance = sum_p (b + l*(a-b)) + bias[row] with
    bias[row] = sum_p (l*l*c - l*c) precomputed per database row.  Contract: candidate SETS (counts, membership) identical to the
    default mode; the distances are bit-identical to a numpy f32 restatement of that formula on the oracle's candidates
    and agree with the reference-association distances to f32 rounding; the top-k differs from the default only among
    near-equal distances."""
    f = fixture(name)
    bv, bb = BV_BB[name]
    k = 100
    o = f.oracle
    coarse = o.coarse()
    idx = f.hip_index()
    try:
        ex_ids, ex_d, ex_c = idx.query(f.queries, bv, bb, k)
        idx.set_option("adc_bias", 1)
        ids, dist, cnt = idx.query(f.queries, bv, bb, k)
        assert np.array_equal(cnt, ex_c)
        o.set_sort_mode(1)
        overlap = []
        for qi, q in enumerate(f.queries):
            u_ids, u_d = o.query_unsorted(q, bv, bb)
            n = len(u_ids)
            assert int(cnt[qi]) == n
            if n == 0:
                continue
            virt = o.stage_l1(q)[0].reshape(o.LP, o.C1)
            w = f.codes[u_ids]  # [n][LP]
            A, B = (w & 0xff).astype(np.int64), ((w >> 8) & 0xff).astype(np.int64)
            lam = (w >> 16).astype(np.float32) * np.float32(8.0 / 65536.0) - np.float32(4.0)
            acc = np.zeros(n, np.float32)
            bias = np.zeros(n, np.float32)
            for p in range(o.LP):
                sb, sa = virt[p, A[:, p]], virt[p, B[:, p]]
                l = lam[:, p]
                acc = acc + (sb + l * (sa - sb))
                c = coarse[p, A[:, p], B[:, p]]
                bias = bias + (l * l * c - l * c)
            d = acc + bias
            assert np.allclose(d, u_d, rtol=2e-5, atol=2e-2), "bias-mode distances drifted from the reference association"
            order = np.lexsort((np.arange(n), d))[:k]
            kk = len(order)
            assert np.array_equal(bits(dist[qi, :kk]), bits(d[order])), qi
            assert np.array_equal(ids[qi, :kk], u_ids[order]), qi
            assert np.all(ids[qi, kk:] == 0xffffffff)
            from collections import Counter  # multisets: an aliased (wrapped) bin is visited twice and repeats its ids
            overlap.append(sum((Counter(ids[qi, :kk].tolist()) & Counter(ex_ids[qi, :kk].tolist())).values()) / kk)
            # an id that left the top-k sits within rounding distance of the k-th
            kth = float(ex_d[qi, kk - 1])
            exact_of = dict(zip(u_ids.tolist(), u_d.tolist()))
            for v in ids[qi, :kk].tolist():
                assert exact_of[v] <= kth * (1 + 1e-4) + 1e-2
        assert np.mean(overlap) > 0.98
        idx.set_option("adc_bias", 0)
        ids2, dist2, _ = idx.query(f.queries, bv, bb, k)
        assert np.array_equal(ids2, ex_ids) and np.array_equal(bits(dist2), bits(ex_d))
    finally:
        o.set_sort_mode(0)
        idx.close()


@pytest.mark.parametrize("k", [130, 300, 2048, 4096])
def test_fused_big_k_select_shrinks_and_matches_oracle(k):
    """128 < k <= 4096 (queryKNN(.., 4096) of the reference front-end): workgroup-per-query fused rerank+select.  A dense
    database gives every query ~30 k candidates, i.e. several times the LDS key array (2*NP2(k) keys), so the exact
    block-wide radix select + compaction runs repeatedly; the first candidates are also the worst ones for it when
    distances decrease.  Results == oracle, and == the staged kernels."""
    from common import Fixture

    def uniform(n, D, seed):
        return np.random.default_rng(seed).integers(0, 256, (n, D)).astype(np.float32)

    f = Fixture(D=16, P=2, C1=8, C2=6, W=8, LP=4, n_base=30000, n_query=6, seed=92, heur_rows=2304, train=3000, data=uniform)
    idx = f.hip_index()
    try:
        ids, dist, cnt = idx.query(f.queries, 10 ** 6, 2304, k)
        assert int(cnt.min()) > 2 * 4096
        f.oracle.set_sort_mode(1)
        try:
            for qi, q in enumerate(f.queries):
                s_ids, s_d = f.oracle.query(q, 10 ** 6, 2304)
                assert int(cnt[qi]) == len(s_ids)
                assert np.array_equal(bits(dist[qi]), bits(s_d[:k])), (k, qi)
                assert np.array_equal(ids[qi], s_ids[:k]), (k, qi)
        finally:
            f.oracle.set_sort_mode(0)
        idx.set_option("fused", 0)
        ids_s, dist_s, cnt_s = idx.query(f.queries, 10 ** 6, 2304, k)
        assert np.array_equal(ids_s, ids) and np.array_equal(bits(dist_s), bits(dist)) and np.array_equal(cnt_s, cnt)
    finally:
        idx.close()


def test_big_coarse_exact_filter_matches_workgroup_kernel_and_falls_back_on_tie_clusters():
    """Shapes whose coarse table does not fit the LDS (BASELINE configs[2]/[3]) run the band-filtered exact rerank by default
    (pqt_rs_query MODE 2): identical to the workgroup-per-query exact kernel.  A database of 600 copies of ONE vector per
    bin puts hundreds of exactly tied candidates around the k-th distance, so the band overflows the wave's 256-entry list
    and the query must take the fallback (plain exact kernel) -- still the oracle's result."""
    from common import Fixture
    for name in ("cfg3_small", "big_coarse"):
        f = fixture(name)
        bv, bb = BV_BB[name]
        idx = f.hip_index()
        try:
            a = idx.query(f.queries, bv, bb, 100)
            assert idx.stats()["filter_fallbacks"] == 0
            idx.set_option("exact_filter", 0)
            b = idx.query(f.queries, bv, bb, 100)
            assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
            # bin runs (default at the configs[2]/[3] shape) against the materialised candidate list
            idx.set_option("exact_filter", 1)
            idx.set_option("bin_runs", 0)
            c = idx.query(f.queries, bv, bb, 100)
            assert np.array_equal(a[0], c[0]) and np.array_equal(bits(a[1]), bits(c[1])) and np.array_equal(a[2], c[2])
        finally:
            idx.close()

    def clustered(n, D, seed):
        # 20 fixed prototypes; half of the vectors are exact copies of one (-> identical line codes, exactly tied
        # distances, 300 per prototype in the database), the other half are noisy (so every tree cell can be trained)
        protos = np.random.default_rng(777).integers(0, 256, (20, D)).astype(np.float32)
        rng = np.random.default_rng(seed)
        x = protos[rng.integers(0, 20, n)]
        noisy = rng.random(n) < 0.5
        x[noisy] = np.clip(np.rint(x[noisy] + rng.normal(0, 25, (int(noisy.sum()), D))), 0, 255)
        return x.astype(np.float32)

    # C1 = 32: candidate lists; C1 = 64 with 32 line parts: the variant that receives bin runs (the fallback kernel then
    # expands the runs itself)
    for C1 in (32, 64):
        f = Fixture(D=64, P=2, C1=C1, C2=4, W=2, LP=32, n_base=12000, n_query=8, seed=68, heur_rows=64, train=3000, data=clustered)
        idx = f.hip_index()
        try:
            ids, dist, cnt = idx.query(f.queries, 10 ** 6, 64, 100)
            st = idx.stats()
            assert st["filter_fallbacks"] > 0, "fixture no longer overflows the band"
            f.oracle.set_sort_mode(1)
            for qi, q in enumerate(f.queries):
                s_ids, s_d = f.oracle.query(q, 10 ** 6, 64)
                kk = min(100, len(s_ids))
                assert int(cnt[qi]) == len(s_ids)
                assert np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk])), (C1, qi)
                assert np.array_equal(ids[qi, :kk], s_ids[:kk]), (C1, qi)
        finally:
            f.oracle.set_sort_mode(0)
            idx.close()


def test_config5_shape_follows_the_reference_no_bins_enumerated():
    """BASELINE configs[4] shape (d=256 p=8 c1=128 c2=64): (W*C2)^P = 64^8 wraps to 0 in the reference's uint arithmetic
    (treequantizer.hpp:40-41), so its orderBins enumerates min(boundBins, 0) = 0 rows and every query returns an empty
    list.  The engine does the same (no error, no invented semantics); the build side (insert) still works and matches."""
    import torch
    from oracle import Oracle
    D, P, C1, C2, W, LP = 256, 8, 128, 64, 1, 32
    rng = np.random.default_rng(5)
    cb1 = rng.uniform(0, 255, (C1, D)).astype(np.float32)
    cb2 = rng.uniform(0, 255, (P, C1, C2, D // P)).astype(np.float32)
    base = rng.integers(0, 256, (300, D)).astype(np.float32)
    o = Oracle(D, P, C1, C2, W, LP, heur_keep=16)
    assert o.max_multi_index == 0
    o.set_codebooks(cb1, cb2)
    o.insert(base)
    idx = pqt_pkg().PqtIndex(D, P, C1, C2, W, LP)
    try:
        idx.set_codebooks(cb1, cb2)
        idx.build_heuristic(500)
        ids_b, sizes_b, members = o.export_bins()
        idx.set_bins(ids_b, sizes_b, members)
        idx.set_lines(o.export_codes())
        ids, dist, cnt = idx.query(base[:8], 20000, 500, 10)
        assert np.all(cnt == 0) and np.all(ids == 0xffffffff) and np.all(np.isinf(dist))
        for q in base[:8]:
            assert len(o.query(q, 20000, 500)[0]) == 0
        x = torch.from_numpy(base).cuda()
        bins = torch.empty(300, dtype=torch.int32, device="cuda")
        codes = torch.empty((300, LP), dtype=torch.int32, device="cuda")
        idx.assign_encode_dev(x, bins, codes)
        torch.cuda.synchronize()
        assert np.array_equal(bins.cpu().numpy().view(np.uint32), np.array([o.bin_id(v) for v in base], np.uint32))
        assert np.array_equal(codes.cpu().numpy().view(np.uint32), o.export_codes())
    finally:
        idx.close()


@pytest.mark.parametrize("name", ["tools_default", "cfg2_small"])
def test_cuda_style_heuristic_mode(name):
    """Optional mode (SURVEY 8f-4 tail): the CUDA library's prepareDistSequence order (pqt/ProTree.cu:128-207) as the
    traversal heuristic.  The table equals a numpy restatement (f32 sum of sqrt(digit) in part order, ties by tuple index),
    and queries under it equal the oracle's when the oracle is handed the same table (everything else is cpu_version)."""
    f = fixture(name)
    c = f.cfg
    P, WC = c["P"], c["W"] * c["C2"]
    b = min(16, WC)
    n_vec = b ** P
    i = np.arange(n_vec, dtype=np.int64)
    digits = np.stack([(i // b ** p) % b for p in range(P)], 1)
    key = np.zeros(n_vec, np.float32)
    for p in range(P):
        key = key + np.sqrt(digits[:, p].astype(np.float32))
    order = np.lexsort((i, key))
    rows = min(n_vec, 65536, 400)
    want = digits[order[:rows]].astype(np.uint32)
    idx = f.hip_index()
    try:
        idx.build_heuristic_cuda(WC, rows)
        assert np.array_equal(idx.heuristic(rows), want)
        ids, dist, cnt = idx.query(f.queries, 500, rows, 50)
        f.oracle.set_heuristic(want)
        f.oracle.set_sort_mode(1)
        for qi, q in enumerate(f.queries):
            s_ids, s_d = f.oracle.query(q, 500, rows)
            kk = min(50, len(s_ids))
            assert int(cnt[qi]) == len(s_ids)
            assert np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk])) and np.array_equal(ids[qi, :kk], s_ids[:kk])
    finally:
        f.oracle.set_heuristic(f.heur)
        f.oracle.set_sort_mode(0)
        idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,dc,knobs", [
    ("cfg2_small", 512, (800, 500, 50)),    # the grid width test/test1B.cpp:941 passes; 64 entries per part list
    ("cfg2_small", 64, (300, 1500, 100)),   # complete 64 x 64 orders (4096 cells, zero filled behind), 1500 rows
    ("cfg3_small", 512, (2000, 400, 100)),  # W = 1, C2 = 64; MODE 2 rerank behind it
    ("wrap", 512, (500, 300, 64)),          # 32 entries per part list (sample positions inside), aliased bin ids
])
def test_2d_anisotropic_sequences_mode(name, dc, knobs):
    """Optional mode (SURVEY 8f-4 tail): the CUDA 1B path's 2-D anisotropic sequences (pqt/ProTree.cu:50-126) and their per-query use
    (pqt/PerturbationProTree.cu:2839-3100) as the choice of the enumerated rows -- pqt_index_build_heuristic_2d + pqt_k_rows_2d against the
    checker's restatement (oracle.build_heuristic_2d): candidate counts, distances and ids of every query; everything behind the rows is
    cpu_version on both sides.  Switching back to a shared table restores the default results."""
    f = fixture(name)
    Bv, Bb, k = knobs
    idx = f.hip_index()
    try:
        base_ids, base_dist, base_cnt = idx.query(f.queries, Bv, min(Bb, f.heur_rows), k)
        idx.build_heuristic_2d(dc)
        f.oracle.build_heuristic_2d(dc)
        f.oracle.set_sort_mode(1)
        ids, dist, cnt = idx.query(f.queries, Bv, Bb, k)
        assert "traverse=staged" in idx.last_path()
        differs = 0
        for qi, q in enumerate(f.queries):
            rows = f.oracle.rows_2d(q, Bb)
            real = rows[rows[:, 0] != 0xffffffff]
            assert len(real) > 0 and int(real.max()) < f.cfg["W"] * f.cfg["C2"]
            s_ids, s_d = f.oracle.query(q, Bv, Bb)
            kk = min(k, len(s_ids))
            assert int(cnt[qi]) == len(s_ids), "candidate count differs q=%d: %d vs %d" % (qi, int(cnt[qi]), len(s_ids))
            assert np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk])) and np.array_equal(ids[qi, :kk], s_ids[:kk]), "results differ q=%d" % qi
            differs += int(cnt[qi]) != int(base_cnt[qi])
        assert differs > 0  # (the mode does change what is enumerated)
        idx.set_heuristic(f.heur)
        ids2, dist2, cnt2 = idx.query(f.queries, Bv, min(Bb, f.heur_rows), k)
        assert np.array_equal(ids2, base_ids) and np.array_equal(bits(dist2), bits(base_dist)) and np.array_equal(cnt2, base_cnt)
    finally:
        f.oracle.set_heuristic(f.heur)
        f.oracle.set_sort_mode(0)
        idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,knobs,opts", [
    ("cfg2_small", (800, 500, 100), {}),                     # LDS coarse table (MODE 0), short fused traversal
    ("cfg2_small", (800, 500, 100), {"bin_runs": 1}),        # ... with bin runs instead of candidate lists
    ("cfg2_small", (300, 1500, 17), {}),                     # wide traversal (1500 rows)
    ("cfg3_small", (2000, 400, 100), {}),                    # MODE 2 filter + bin runs (6/12-wave workgroups)
    ("cfg3_small", (2000, 400, 100), {"exact_filter": 0}),   # workgroup-per-query kernel (no wave schedule at all)
    ("wrap", (500, 300, 64), {}),                            # LP = 8
    ("big_coarse", (400, 64, 33), {}),                       # coarse table through L2
])
def test_rerank_schedules_agree_on_large_batches(name, knobs, opts):
    """More queries than wavefront slots (> 256 CUs x 12): the three rerank schedules (static round-robin, workgroup-local lists,
    global pools fed by the traversal's registration lists) must give identical results; a sample is checked against the oracle,
    and the same through two range shards merged."""
    import torch
    f = fixture(name)
    rng = np.random.default_rng(4242)
    qn = 3300
    pick = rng.integers(0, f.base.shape[0], qn)
    queries = np.clip(np.rint(f.base[pick] + rng.normal(0, 6, (qn, f.base.shape[1]))), 0, 255).astype(np.float32)
    bv, bb, k = knobs
    bb = min(bb, f.heur.shape[0])
    idx = f.hip_index()
    n = f.base.shape[0]
    shards = [f.hip_index(shard=(0, n // 3)), f.hip_index(shard=(n // 3, n))]
    try:
        for h in [idx] + shards:
            for o, v in opts.items():
                h.set_option(o, v)
        ref = None
        for bal in (2, 1, 0):
            idx.set_option("balance", bal)
            got = idx.query(queries, bv, bb, k)
            if ref is None:
                ref = got
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(got[2], ref[2]), bal
        f.oracle.set_sort_mode(1)
        try:
            for qi in range(0, qn, 97):
                s_ids, s_d = f.oracle.query(queries[qi], bv, bb)
                kk = min(k, len(s_ids))
                assert int(ref[2][qi]) == len(s_ids)
                assert np.array_equal(ref[0][qi, :kk], s_ids[:kk]) and np.array_equal(ref[1][qi, :kk].view(np.uint32), s_d[:kk].view(np.uint32))
        finally:
            f.oracle.set_sort_mode(0)
        # two range shards, schedule 2 on each, merged
        q = torch.from_numpy(queries).cuda()
        torch.cuda.synchronize()
        pack = torch.empty((2, 3, qn, k), dtype=torch.int32, device="cuda")
        cnt = torch.empty((2, qn), dtype=torch.int32, device="cuda")
        for s, sh in enumerate(shards):
            sh.query_shard_dev(q, bv, bb, k, pack[s, 0], pack[s, 1].view(torch.float32), pack[s, 2], cnt[s], sync=True)
        oi = torch.empty((qn, k), dtype=torch.int32, device="cuda")
        od = torch.empty((qn, k), dtype=torch.float32, device="cuda")
        shards[0].merge_topk_dev(2, qn, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oi, od, sync=True, shard_stride=3 * qn * k)
        assert np.array_equal(oi.cpu().numpy().view(np.uint32), ref[0])
        assert np.array_equal(od.cpu().numpy().view(np.uint32), ref[1].view(np.uint32))
    finally:
        for h in [idx] + shards:
            h.close()


# ---- round 3: the path taken is asserted, not only the result -------------------------------------------
@pytest.mark.parametrize("shape,expect,expect_bias", [
    # C1 = 128, LP = 32: 16 KB of L1virt per wave -> the 6-wave MODE 2 / MODE 1 kernels (used to fall to the staged rerank
    # silently because the 12-wave LDS size was tested)
    ((64, 2, 128, 4, 2, 32), "rerank=mode2-nw6", "rerank=mode1-nw6"),
    # C1 = 256, LP = 16: same L1virt size per wave
    ((64, 2, 256, 4, 2, 16), "rerank=mode2-nw6", "rerank=mode1-nw6"),
    # BASELINE configs[2]/[3] shape: 12 waves + bin runs
    ((128, 4, 64, 64, 1, 32), "rerank=mode2-nw12-runs", "rerank=mode1-nw12-runs"),
    # BASELINE configs[1] shape: the coarse table lives in LDS
    ((128, 4, 32, 32, 2, 16), "rerank=lds-table-xcode", "rerank=mode1-nw12"),
])
def test_kernel_path_taken_is_the_documented_one(shape, expect, expect_bias):
    from common import Fixture
    D, P, C1, C2, W, LP = shape
    f = Fixture(D=D, P=P, C1=C1, C2=C2, W=W, LP=LP, n_base=5000, n_query=8, seed=4242 + C1 + LP, heur_rows=64, train=2500)
    idx = f.hip_index()
    try:
        ids, dist, cnt = idx.query(f.queries, 600, 64, 33)
        path = idx.last_path()
        assert "traverse=fused" in path and expect in path.split(), path
        st = idx.stats()
        if "mode2" in expect:
            assert st["filter_fallbacks"] == 0
        f.oracle.set_sort_mode(1)
        try:
            for qi, q in enumerate(f.queries):
                s_ids, s_d = f.oracle.query(q, 600, 64)
                kk = min(33, len(s_ids))
                assert int(cnt[qi]) == len(s_ids)
                assert np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk])) and np.array_equal(ids[qi, :kk], s_ids[:kk]), (path, qi)
        finally:
            f.oracle.set_sort_mode(0)
        idx.set_option("adc_bias", 1)
        ids_b, dist_b, cnt_b = idx.query(f.queries, 600, 64, 33)
        pb = idx.last_path()
        assert expect_bias in pb.split(), pb
        assert np.array_equal(cnt_b, cnt)
        # same candidate sets; distances follow MODE 1's own association: close to the reference's, not bit-equal in general
        assert np.allclose(np.where(np.isinf(dist_b), 0, dist_b), np.where(np.isinf(dist), 0, dist), rtol=1e-4, atol=1e-2)
        idx.set_option("adc_bias", 0)
        idx.set_option("fused", 0)
        idx.query(f.queries, 600, 64, 33)
        assert "traverse=staged" in idx.last_path() and "rerank=staged-select" in idx.last_path()
    finally:
        idx.close()


def test_stage_timing_period_getters_fall_back_to_the_last_timed_call():
    """stage_timing = N > 1: untimed calls record no events; pqt_get_rerank_launch_ms / pqt_get_stats report the most recent
    timed call instead of failing / returning zeros."""
    f = fixture("cfg2_small")
    idx = f.hip_index()
    try:
        idx.set_option("stage_timing", 4)
        for i in range(3):  # call 0 is timed, calls 1 and 2 are not
            idx.query(f.queries, 300, 500, 10)
        ms = idx.rerank_launch_ms()
        assert len(ms) == 1 and ms[0] > 0
        st = idx.stats()
        assert st["ms_total"] > 0 and st["ms_rerank"] > 0
        assert idx.stage_ms_history(8).shape[0] == 1
        idx.set_option("stage_timing", 0)
        for i in range(40):  # the whole ring untimed: nothing to report, but no error
            idx.query(f.queries[:2], 300, 500, 10)
        assert len(idx.rerank_launch_ms()) == 0 and idx.stats()["ms_total"] == 0
    finally:
        idx.close()


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_small", "wrap", "odd", "ties", "big_coarse"])
def test_overlapped_halves_equal_the_single_piece_call(name):
    """"overlap" = 1 (2 pieces) / n: pqt_query runs the batch as pieces on their own streams, the later ones on view handles that
    share the index arrays.  Results, candidate counts, statistics and the debug read-back (candidate sets of BOTH halves) equal the
    one-piece call's; a timed call (stage events) stays in one piece."""
    f = fixture(name)
    idx = f.hip_index()
    try:
        Bv, Bb = BV_BB[name]
        for k in (10, 100):
            idx.set_option("overlap", 0)
            idx.set_option("stage_timing", 0)
            i0, d0, c0 = idx.query(f.queries, Bv, Bb, k)
            assert "overlap" not in idx.last_path()
            st0 = idx.stats()
            with_cands = "-runs" not in idx.last_path()  # (bin runs: no candidate list is materialised)
            r0 = idx.debug_read(len(f.queries), cands=with_cands, segs=False, dists=False)
            idx.set_option("overlap", 1)
            for rep in range(2):  # (the first overlapped call creates the view handle)
                i1, d1, c1 = idx.query(f.queries, Bv, Bb, k)
                assert "overlap=2-pieces" in idx.last_path(), idx.last_path()
                assert np.array_equal(i0, i1) and np.array_equal(bits(d0), bits(d1)) and np.array_equal(c0, c1)
            st1 = idx.stats()
            for key in ("queries", "candidates", "bins_nonempty", "bins_visited", "ties_final", "filter_fallbacks"):
                assert st0[key] == st1[key], (key, st0[key], st1[key])
            r1 = idx.debug_read(len(f.queries), cands=with_cands, segs=False, dists=False)
            assert np.array_equal(r1["ncand"], r0["ncand"]) and np.array_equal(bits(r1["l1virt"]), bits(r0["l1virt"]))
            if with_cands:
                for qi in range(len(f.queries)):
                    assert np.array_equal(r1["cand_idx"][qi, :r1["ncand"][qi]], r0["cand_idx"][qi, :r0["ncand"][qi]]), qi
            idx.set_option("stage_timing", 1)  # every call carries events: one piece
            i2, d2, c2 = idx.query(f.queries, Bv, Bb, k)
            assert "overlap" not in idx.last_path()
            assert np.array_equal(i0, i2) and np.array_equal(bits(d0), bits(d2))
        # one query cannot be split; an odd batch splits unevenly
        idx.set_option("stage_timing", 0)
        i3, d3, c3 = idx.query(f.queries[:1], Bv, Bb, 10)
        assert "overlap" not in idx.last_path()
        i4, d4, c4 = idx.query(f.queries[:3], Bv, Bb, 10)
        assert "overlap=2-pieces" in idx.last_path()
        idx.set_option("overlap", 0)
        i5, d5, c5 = idx.query(f.queries[:3], Bv, Bb, 10)
        assert np.array_equal(i4, i5) and np.array_equal(bits(d4), bits(d5)) and np.array_equal(c4, c5)
        # three and four pieces (option value = number of pieces), statistics and read-back over all of them
        i6, d6, c6 = idx.query(f.queries, Bv, Bb, 10)
        st6 = idx.stats()
        for pieces in (3, 4):
            idx.set_option("overlap", pieces)
            i7, d7, c7 = idx.query(f.queries, Bv, Bb, 10)
            want = min(pieces, len(f.queries))
            assert ("overlap=%d-pieces" % want) in idx.last_path(), idx.last_path()
            assert np.array_equal(i6, i7) and np.array_equal(bits(d6), bits(d7)) and np.array_equal(c6, c7)
            st7 = idx.stats()
            assert st7["queries"] == st6["queries"] and st7["candidates"] == st6["candidates"]
            r7 = idx.debug_read(len(f.queries), cands=False, segs=False, dists=False)
            assert np.array_equal(r7["ncand"].astype(np.int64), c6.astype(np.int64))
    finally:
        idx.close()


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_small", "ties", "big_coarse", "wrap"])
def test_tie_statistics_count_adjacent_equal_distances_of_the_result_lists(name):
    """pqt_stats.ties_final = adjacent equal-distance pairs inside the returned lists, summed over the batch.  The kernels sum it per
    wavefront / per workgroup before ONE atomic (pqt_count_ties; one atomic per lane serialised the whole rerank launch): every kernel
    family must still report the exact count -- fused wave-per-query (k <= 128), short-list + block-wide (128 < k <= 4096), staged."""
    f = fixture(name)
    idx = f.hip_index()
    try:
        Bv, Bb = BV_BB[name]
        qs = np.concatenate([f.queries] * 8)  # several queries per wavefront slot and equal lists in different workgroups
        for k, opts in ((100, {}), (7, {}), (300, {}), (300, {"small_lists": 0}), (100, {"fused": 0})):
            for o, v in opts.items():
                idx.set_option(o, v)
            ids, dist, cnt = idx.query(qs, Bv, Bb, k)
            st = idx.stats()
            want = 0
            for qi in range(len(qs)):
                kk = min(k, int(cnt[qi]))
                d = bits(dist[qi, :kk])
                want += int((d[1:] == d[:-1]).sum())
            assert st["ties_final"] == want, (name, k, opts, idx.last_path(), st["ties_final"], want)
            for o in opts:
                idx.set_option(o, 1)
    finally:
        idx.close()


def test_one_launch_query_kernel_equals_the_two_launches():
    """"one_launch" = 1 (SIFT1M shape): traversal + rerank/select of a query by one wavefront in one launch (pqt_k_query_fused) --
    identical lists, counts and statistics, stage events around the single dispatch; shapes it does not cover keep the two launches."""
    f = fixture("cfg2_small")
    idx = f.hip_index()
    try:
        Bv, Bb = BV_BB["cfg2_small"]
        qs = np.concatenate([f.queries] * 40)  # several queries per wavefront
        for k in (1, 10, 100):
            idx.set_option("one_launch", 0)
            i0, d0, c0 = idx.query(qs, Bv, Bb, k)
            assert "one-launch" not in idx.last_path()
            st0 = idx.stats()
            idx.set_option("one_launch", 1)
            i1, d1, c1 = idx.query(qs, Bv, Bb, k)
            assert "one-launch" in idx.last_path(), idx.last_path()
            st1 = idx.stats()
            assert np.array_equal(i0, i1) and np.array_equal(bits(d0), bits(d1)) and np.array_equal(c0, c1)
            for key in ("queries", "candidates", "bins_nonempty", "ties_final"):
                assert st0[key] == st1[key], key
            assert st1["ms_total"] > 0 and st1["ms_rerank"] > 0
        i2, d2, c2 = idx.query(qs, Bv, 1024, 10)  # wide traversal: not covered
        assert "one-launch" not in idx.last_path()
        i3, d3, c3 = idx.query(qs, Bv, Bb, 300)   # k > 128: not covered
        assert "one-launch" not in idx.last_path()
    finally:
        idx.close()
    g = fixture("cfg3_small")
    jdx = g.hip_index()
    try:
        jdx.set_option("one_launch", 1)
        jdx.query(g.queries, *BV_BB["cfg3_small"], 10)
        assert "one-launch" not in jdx.last_path()
    finally:
        jdx.close()


def test_query_candidates_entry_point_returns_the_whole_sorted_list():
    """pqt_query_candidates by name (oracle-parity entry): the reference's whole sorted candidate list per query,
    true lengths in out_count, lists longer than cap cut after cap entries; a missing out_count is rejected."""
    import ctypes as C
    import torch
    pkg = pqt_pkg()
    f = fixture("tools_default")
    idx = f.hip_index()
    try:
        Bv, Bb = BV_BB["tools_default"]
        q = torch.from_numpy(f.queries).cuda()
        qn = q.shape[0]
        for cap in (8192, 64):
            oi = torch.empty((qn, cap), dtype=torch.int32, device="cuda")
            od = torch.empty((qn, cap), dtype=torch.float32, device="cuda")
            oc = torch.empty(qn, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            idx.query_candidates_dev(q, Bv, Bb, cap, oi, od, oc, sync=True)
            gi, gd, gc = oi.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy().view(np.uint32)
            f.oracle.set_sort_mode(1)
            try:
                for qi in range(qn):
                    s_ids, s_d = f.oracle.query(f.queries[qi], Bv, Bb)
                    assert int(gc[qi]) == len(s_ids)
                    n = min(cap, len(s_ids))
                    assert np.array_equal(gi[qi, :n], s_ids[:n]) and np.array_equal(bits(gd[qi, :n]), bits(s_d[:n]))
            finally:
                f.oracle.set_sort_mode(0)
        rc = pkg.lib().pqt_query_candidates(idx.h, q.data_ptr(), qn, Bv, Bb, 64, oi.data_ptr(), od.data_ptr(), None, None, 1)
        assert rc == -1 and b"out_count" in pkg.lib().pqt_last_error()
    finally:
        idx.close()


@pytest.mark.parametrize("name", ["tools_default", "cfg2_small", "cfg3_small", "wrap", "odd", "ties", "big_coarse"])
@pytest.mark.parametrize("cap", [128, 3, 256])
def test_query_sharded_traversal_equals_replicated_traversal(name, cap):
    """pqt_traverse_bins on ONE shard + pqt_query_shard_bins on EVERY shard == pqt_query_shard (each shard traversing itself):
    ids, distance bits, global visiting positions, counts.  The bin lists are shard independent (every shard writes the same
    bytes) and equal the lists derived from the oracle's traversal; a small capacity forces the traverse-it-yourself fallback;
    bound_bins > 512 runs the wide traversal on both sides."""
    import torch
    from test_cpu_sharding_gloo import OracleShardEngine
    f = fixture(name)
    n = f.base.shape[0]
    cuts = [0, n // 3, n]
    shards = [f.hip_index(shard=(cuts[r], cuts[r + 1])) for r in range(2)]
    try:
        q = torch.from_numpy(f.queries).cuda()
        qn, k = q.shape[0], 50
        knobs = [BV_BB[name]]
        if CONFIGS[name]["heur_rows"] >= 1024:
            knobs.append((10 ** 6, 1000))  # wide traversal
        for bv, bb in knobs:
            lists = []
            for sh in shards:
                b = torch.zeros((qn, cap + 1), dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                sh.traverse_bins_dev(q, bv, bb, cap, b, sync=True)
                lists.append(b.cpu().numpy().view(np.uint64))
            # only the used part of a row is defined
            for qi in range(qn):
                m = int(lists[0][qi, cap]) & 0xffffffff
                assert lists[0][qi, cap] == lists[1][qi, cap]
                if m != 0xffffffff:
                    assert np.array_equal(lists[0][qi, :m], lists[1][qi, :m]), (name, bv, bb, qi)
            st = shards[0].stats() if False else None
            ref = []
            for sh in shards:
                o = [torch.empty((qn, k), dtype=torch.int32, device="cuda"), torch.empty((qn, k), dtype=torch.float32, device="cuda"),
                     torch.empty((qn, k), dtype=torch.int32, device="cuda"), torch.empty(qn, dtype=torch.int32, device="cuda")]
                sh.query_shard_dev(q, bv, bb, k, o[0], o[1], o[2], o[3], sync=True)
                ref.append([t.cpu().numpy() for t in o])
                ties = sh.stats()["ties_bins"]
            if not ties and name != "ties":
                exp = torch.zeros((qn, cap + 1), dtype=torch.int64)
                OracleShardEngine(f, 0, n).traverse_bins(torch.from_numpy(f.queries), bv, bb, cap, exp)
                exp = exp.numpy().view(np.uint64)
                for qi in range(qn):
                    m = int(exp[qi, cap]) & 0xffffffff
                    assert exp[qi, cap] == lists[0][qi, cap] or (m == 0xffffffff and (int(lists[0][qi, cap]) & 0xffffffff) == 0xffffffff), (name, bv, bb, qi)
                    if m != 0xffffffff:
                        assert np.array_equal(exp[qi, :m], lists[0][qi, :m]), (name, bv, bb, qi)
            over = sum(1 for qi in range(qn) if (int(lists[0][qi, cap]) & 0xffffffff) == 0xffffffff)
            if cap == 3:
                assert over > 0, "the small capacity was meant to overflow some lists"
            bins_dev = torch.from_numpy(lists[0].view(np.int64)).cuda()
            for s_, sh in enumerate(shards):
                o = [torch.empty((qn, k), dtype=torch.int32, device="cuda"), torch.empty((qn, k), dtype=torch.float32, device="cuda"),
                     torch.empty((qn, k), dtype=torch.int32, device="cuda"), torch.empty(qn, dtype=torch.int32, device="cuda")]
                torch.cuda.synchronize()
                sh.query_shard_bins_dev(q, bv, bb, k, bins_dev, cap, o[0], o[1], o[2], o[3], sync=True)
                assert "traverse=bins-resolved" in sh.last_path(), sh.last_path()
                got = [t.cpu().numpy() for t in o]
                assert np.array_equal(got[3], ref[s_][3]), (name, bv, bb, s_, "counts")
                assert np.array_equal(got[0], ref[s_][0]) and np.array_equal(got[1].view(np.uint32), ref[s_][1].view(np.uint32)), (name, bv, bb, s_)
                assert np.array_equal(got[2], ref[s_][2]), (name, bv, bb, s_, "positions")
    finally:
        for sh in shards:
            sh.close()


@pytest.mark.parametrize("name,nsh", [("tools_default", 2), ("cfg2_small", 3), ("cfg3_small", 4), ("wrap", 5), ("odd", 2)])
def test_multi_handle_equals_single_index(name, nsh):
    """pqt_multi_*: ONE handle over nsh range shards in one process (all on device 0 here) -- query slices traversed by
    different shards, bin lists exchanged by copies, per-shard top-k merged -- returns the single-index result bit for bit, with
    the sharded and the replicated traversal, for k below and above 128, through the device- and the host-pointer entry."""
    import torch
    pkg = pqt_pkg()
    f = fixture(name)
    c = f.cfg
    ref = f.hip_index()
    m = pkg.PqtMulti(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], [0] * nsh)
    try:
        m.set_codebooks(f.cb1, f.cb2)
        m.set_heuristic(f.heur)
        m.set_bins(f.bin_ids, f.bin_sizes, f.members)
        m.set_lines(f.codes)
        n = f.base.shape[0]
        assert [m.shard_range(s) for s in range(nsh)] == [(n * s // nsh, n * (s + 1) // nsh) for s in range(nsh)]
        Bv, Bb = BV_BB[name]
        q = torch.from_numpy(f.queries).cuda()
        qn = q.shape[0]
        for k in (10, 100, 300, 5000):  # 5000: nsh * k keys do not fit the LDS merge -> the ranked merge
            r_ids, r_d, r_c = ref.query(f.queries, Bv, Bb, k)
            for rep in (0, 1):
                m.set_option("replicated_traversal", rep)
                oi = torch.empty((qn, k), dtype=torch.int32, device="cuda")
                od = torch.empty((qn, k), dtype=torch.float32, device="cuda")
                oc = torch.empty(qn, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                m.query_dev(q, Bv, Bb, k, oi, od, oc, sync=True)
                assert np.array_equal(oi.cpu().numpy().view(np.uint32), r_ids), (name, k, rep)
                assert np.array_equal(bits(od.cpu().numpy()), bits(r_d)), (name, k, rep)
                assert np.array_equal(oc.cpu().numpy().view(np.uint32), r_c), (name, k, rep)
                assert ("bins-resolved" in m.shard_last_path(nsh - 1)) == (rep == 0)
            h_ids, h_d, h_c = m.query(f.queries, Bv, Bb, k)
            assert np.array_equal(h_ids, r_ids) and np.array_equal(bits(h_d), bits(r_d)) and np.array_equal(h_c, r_c)
        # two batches back to back without a host synchronisation in between reuse the exchange buffers correctly
        oi2 = [torch.empty((qn, 50), dtype=torch.int32, device="cuda") for _ in range(2)]
        od2 = [torch.empty((qn, 50), dtype=torch.float32, device="cuda") for _ in range(2)]
        m.set_option("replicated_traversal", 0)
        qrev = q.flip(0).contiguous()
        torch.cuda.synchronize()
        m.query_dev(q, Bv, Bb, 50, oi2[0], od2[0])
        m.query_dev(qrev, Bv, Bb, 50, oi2[1], od2[1], sync=True)
        r_ids, r_d, _ = ref.query(f.queries, Bv, Bb, 50)
        assert np.array_equal(oi2[0].cpu().numpy().view(np.uint32), r_ids) and np.array_equal(oi2[1].cpu().numpy().view(np.uint32), r_ids[::-1])
        assert np.array_equal(bits(od2[1].cpu().numpy()), bits(r_d[::-1]))
    finally:
        m.close()
        ref.close()


@pytest.mark.parametrize("name,bv", [("tools_default", 300), ("tools_default", 10 ** 6), ("cfg2_small", 10 ** 6), ("cfg2_dense", 1500), ("cfg2_dense", 10 ** 6), ("cfg3_small", 10 ** 6), ("odd", 400)])
@pytest.mark.parametrize("k", [129, 600, 4096])
def test_short_lists_sorted_by_one_wavefront_long_ones_by_the_block_kernel(name, bv, k):
    """128 < k <= 4096: candidate lists of <= 1024 entries are evaluated and sorted by one wavefront (pqt_k_rerank_sort_small), longer
    ones are handed on -- 1025..2048 to a second wave-per-query pass at the SIFT1M shape, the rest to the block-wide select kernel --
    all against the oracle, and identical to the block-wide kernel alone."""
    f = fixture(name)
    idx = f.hip_index()
    try:
        bb = min(CONFIGS[name]["heur_rows"], 1024)
        if name == "cfg2_dense":
            bb = 4096  # enough bins for lists of 1025..2048 candidates (bv = 1500: cut after the bin that crosses 1500) and beyond
        ids, dist, cnt = idx.query(f.queries, bv, bb, k)
        path = idx.last_path()
        handed = idx.stats()["filter_fallbacks"]
        if CONFIGS[name]["LP"] in (16, 32):
            assert "+small-lists" in path, path
            assert handed == int((cnt > 1024).sum()), (handed, cnt)  # exactly the long lists were set aside by the first pass
            # SIFT1M shape: a second wave-per-query pass takes the lists of 1025..2048 candidates, the block-wide kernel the rest
            assert ("+mid-lists" in path) == (name in ("cfg2_small", "cfg2_dense")), path
            if name == "cfg2_dense":
                assert int(((cnt > 1024) & (cnt <= 2048)).sum()) > 0, cnt  # the second pass is exercised
                if bv > 1500:
                    assert int((cnt > 2048).sum()) > 0, cnt            # and so is the hand-over to the block-wide kernel
        f.oracle.set_sort_mode(1)
        try:
            for qi, q in enumerate(f.queries):
                s_ids, s_d = f.oracle.query(q, bv, bb)
                kk = min(k, len(s_ids))
                assert int(cnt[qi]) == len(s_ids)
                assert np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk])) and np.array_equal(ids[qi, :kk], s_ids[:kk]), (name, bv, k, qi)
                assert np.all(ids[qi, kk:] == 0xffffffff) and np.all(np.isinf(dist[qi, kk:]))
        finally:
            f.oracle.set_sort_mode(0)
        idx.set_option("small_lists", 0)
        ids2, dist2, cnt2 = idx.query(f.queries, bv, bb, k)
        assert "+small-lists" not in idx.last_path()
        assert np.array_equal(ids2, ids) and np.array_equal(bits(dist2), bits(dist)) and np.array_equal(cnt2, cnt)
    finally:
        idx.close()


def test_multi_handle_on_distinct_devices_equals_single_index():
    """the one-handle multi-GPU path with every shard on its OWN device -- hipMemcpyPeerAsync, hipDeviceEnablePeerAccess
    and hipStreamWaitEvent on another device's event really execute.  Needs >= 2 GPUs; the 1-GPU test box skips (the same-device
    variant above covers the protocol there)."""
    import torch
    pkg = pqt_pkg()
    ndev = pkg.lib().pqt_device_count()
    if ndev < 2:
        pytest.skip("needs >= 2 gfx950 devices (this box has %d)" % ndev)
    f = fixture("cfg3_small")
    c = f.cfg
    ref = f.hip_index()
    nsh = min(ndev, 4)
    m = pkg.PqtMulti(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], list(range(nsh)))
    try:
        m.set_codebooks(f.cb1, f.cb2)
        m.set_heuristic(f.heur)
        m.set_bins(f.bin_ids, f.bin_sizes, f.members)
        m.set_lines(f.codes)
        Bv, Bb = BV_BB["cfg3_small"]
        for qn in (f.queries.shape[0], 5):  # the second, smaller batch reuses the buffers; a third, larger one grows them
            r_ids, r_d, r_c = ref.query(f.queries[:qn], Bv, Bb, 100)
            h_ids, h_d, h_c = m.query(f.queries[:qn], Bv, Bb, 100)
            assert np.array_equal(h_ids, r_ids) and np.array_equal(bits(h_d), bits(r_d)) and np.array_equal(h_c, r_c)
        big = np.concatenate([f.queries, f.queries[::-1]])
        r_ids, r_d, _ = ref.query(big, Bv, Bb, 100)
        h_ids, h_d, _ = m.query(big, Bv, Bb, 100)
        assert np.array_equal(h_ids, r_ids) and np.array_equal(bits(h_d), bits(r_d))
    finally:
        m.close()
        ref.close()


def test_multi_handle_heuristics_reach_every_shard_and_hostile_row_counts_are_clamped():
    """(1) the CUDA-order table (prepareDistSequence(maxCluster, groupParts)) must be on EVERY shard -- the traversal is
    sharded by query slice, so a table on shard 0 only orders one slice's bins differently; (2) pqt_multi_build_heuristic with
    rows = 2^32 ("all rows") clamps to the table instead of throwing across the C-ABI; (3) pqt_multi_query_host with qn = 0 is OK."""
    pkg = pqt_pkg()
    f = fixture("tools_default")
    c = f.cfg
    ref = f.hip_index()
    m = pkg.PqtMulti(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], [0, 0, 0])
    try:
        m.set_codebooks(f.cb1, f.cb2)
        m.set_bins(f.bin_ids, f.bin_sizes, f.members)
        m.set_lines(f.codes)
        Bv, Bb = BV_BB["tools_default"]
        ref.build_heuristic_cuda(c["C2"] * c["W"], 4096)
        m.build_heuristic_cuda(c["C2"] * c["W"], 4096)
        rows = min(4096, min(16, c["C2"] * c["W"]) ** c["P"])
        want = ref.heuristic(rows)
        L = pkg.lib()
        for s in range(3):
            got = np.zeros((rows, c["P"]), np.uint32)
            assert L.pqt_index_get_heuristic(L.pqt_multi_shard(m.h, s), got.ctypes.data_as(pkg.u32p), rows) == 0
            assert np.array_equal(got, want), s
        bb = min(Bb, rows)
        r_ids, r_d, r_c = ref.query(f.queries, Bv, bb, 50)
        h_ids, h_d, h_c = m.query(f.queries, Bv, bb, 50)
        assert np.array_equal(h_ids, r_ids) and np.array_equal(bits(h_d), bits(r_d)) and np.array_equal(h_c, r_c)
        # "all rows": (W*C2)^P = 32^2 = 1024 tuples exist
        m.build_heuristic(2 ** 32)
        ref.build_heuristic(2 ** 32)
        r_ids, r_d, _ = ref.query(f.queries, Bv, 1024, 50)
        h_ids, h_d, _ = m.query(f.queries, Bv, 1024, 50)
        assert np.array_equal(h_ids, r_ids) and np.array_equal(bits(h_d), bits(r_d))
        z = np.zeros((0, c["D"]), np.float32)
        e_ids, e_d, e_c = m.query(z, Bv, Bb, 10)
        assert e_ids.shape == (0, 10)
    finally:
        m.close()
        ref.close()


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_small"])
def test_view_handle_serves_a_second_batch_in_flight(name):
    """pqt_index_create_view: a second handle on the same loaded index (own scratch / stream / statistics).  Two batches enqueued back
    to back on two streams -- one on the owner, one on the view -- return what the owner alone returns; the view follows the owner's
    later changes (a longer heuristic prefix); loading into a view and asking it to build shared data are refused."""
    import torch
    pkg = pqt_pkg()
    f = fixture(name)
    idx = f.hip_index()
    try:
        v = idx.view()
        Bv, Bb = BV_BB[name]
        q = torch.from_numpy(f.queries).cuda()
        qn, k = q.shape[0], 64
        h = qn // 2
        with pytest.raises(pkg.PqtError):
            v.set_codebooks(f.cb1, f.cb2)
        o_v = [torch.empty((qn, k), dtype=torch.int32, device="cuda"), torch.empty((qn, k), dtype=torch.float32, device="cuda"), torch.empty(qn, dtype=torch.int32, device="cuda")]
        with pytest.raises(pkg.PqtError):  # the owner has not served a call yet: the bin-ordered line store does not exist
            v.query_dev(q, Bv, Bb, k, *o_v, sync=True)
        r_ids, r_d, r_c = idx.query(f.queries, Bv, Bb, k)
        s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
        o_a = [torch.empty((qn, k), dtype=torch.int32, device="cuda"), torch.empty((qn, k), dtype=torch.float32, device="cuda"), torch.empty(qn, dtype=torch.int32, device="cuda")]
        torch.cuda.synchronize()
        for rep in range(3):
            idx.query_dev(q[:h], Bv, Bb, k, o_a[0][:h], o_a[1][:h], o_a[2][:h], stream=s0.cuda_stream)
            v.query_dev(q[h:], Bv, Bb, k, o_a[0][h:], o_a[1][h:], o_a[2][h:], stream=s1.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(o_a[0].cpu().numpy().view(np.uint32), r_ids) and np.array_equal(bits(o_a[1].cpu().numpy()), bits(r_d))
        assert np.array_equal(o_a[2].cpu().numpy().view(np.uint32), r_c)
        assert v.stats()["queries"] == qn - h and idx.stats()["queries"] == h
        # the view sees what the owner loads later
        rows = min(CONFIGS[name]["heur_rows"], 2 * Bb)
        idx.set_heuristic(f.heur[:rows])
        r2 = idx.query(f.queries, Bv, rows, k)
        v.query_dev(q, Bv, rows, k, *o_v, sync=True)
        assert np.array_equal(o_v[0].cpu().numpy().view(np.uint32), r2[0]) and np.array_equal(bits(o_v[1].cpu().numpy()), bits(r2[1]))
    finally:
        idx.close()


@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_reference_sort_and_scan_self_checks_through_the_library_primitives(n):
    """The reference's only sort / scan vectors (pqt/bitonicSort.cuh:213-252): sortTestLarge sorts the values N - tid that carry
    tid and expects payload[tid] == N - tid - 1, scanTestLarge expects the exclusive scan of ones == tid, for N = 1024, 2048, 4096 --
    fed to the primitives stage a8 is made of: the in-register wave network, the block-wide bitonic network, both radix selects, both scans."""
    pkg = pqt_pkg()
    want = (n - 1 - np.arange(n)).astype(np.uint32)
    if n <= 2048:
        assert np.array_equal(pkg.debug_sort_scan(0, n)[:n], want), "pqt_wave_sort_u64"
    assert np.array_equal(pkg.debug_sort_scan(1, n)[:n], want), "pqt_bitonic_sort_u64"
    if n == 1024:
        for m in (512, 1024):  # the j-th smallest value is j + 1 = N - payload
            assert np.array_equal(pkg.debug_sort_scan(2, m)[:m], (m - 1 - np.arange(m)).astype(np.uint32)), "pqt_wave_kth_u64"
        for m in (64, 128, 256, 512):
            assert np.array_equal(pkg.debug_sort_scan(0, m)[:m], (m - 1 - np.arange(m)).astype(np.uint32)), "pqt_wave_sort_u64 (short)"
    got = pkg.debug_sort_scan(3, n)[:n]
    asked = np.arange(0, n, n >> 6)
    assert np.array_equal(got[asked], want[asked]) and (np.delete(got, asked) == 0xffffffff).all(), "pqt_block_kth_u64"
    assert np.array_equal(pkg.debug_sort_scan(4, n)[:n], np.arange(n, dtype=np.uint32)), "pqt_wave_incl_scan"
    s5 = pkg.debug_sort_scan(5, n)
    assert np.array_equal(s5[:n], np.arange(n, dtype=np.uint32)) and s5[n] == n, "pqt_block_excl_scan"


def test_config5_shape_throughput_mode_matches_the_checker_with_the_same_prefix():
    """BASELINE configs[4] shape with the row limit lifted ("enumerate_beyond_wrap" -- NO reference counterpart: the reference's uint32
    count of heuristic rows wraps to 0 and it enumerates nothing): with the best-first prefix of the sum-of-squares order supplied to
    the engine and to the checker (whose limit is lifted the same way) the whole path -- a1/a2 at cb1 = 128 KB / cb2 = 8 MB, bin ids
    in uint32 wrap-around (only parts 0..2 reach them), the cut, MODE 2 over a 2 MB coarse table, top-k -- is bit-identical."""
    import importlib
    import torch
    from oracle import Oracle
    hp = importlib.import_module("product-quantization-tree_amd.heuristic")
    D, P, C1, C2, W, LP = 256, 8, 128, 64, 1, 32
    n, nq = 4000, 24
    data = np.concatenate([sift_like(n + 2000, 128, 901), sift_like(n + 2000, 128, 902)], 1)
    rng = np.random.default_rng(9)
    cb1 = data[n:n + C1].copy()
    S = D // P
    pick = rng.integers(0, n, (C1, C2))
    cb2 = np.stack([data[pick][:, :, p * S:(p + 1) * S] for p in range(P)]).astype(np.float32)
    base = data[:n]
    queries = np.clip(np.rint(base[rng.integers(0, n, nq)] + rng.normal(0, 3, (nq, D))), 0, 255).astype(np.float32)
    rows = 600
    prefix = hp.heuristic_prefix_best_first(W * C2, P, rows)
    assert prefix.shape == (rows, P) and (prefix[0] == 0).all() and int((prefix[1:9] ** 2).sum()) == 8  # the eight unit tuples follow the origin
    o = Oracle(D, P, C1, C2, W, LP, heur_keep=16)
    o.set_codebooks(cb1, cb2)
    o.insert(base)
    o.lift_tuple_wrap(rows)
    o.set_heuristic(prefix)
    idx = pqt_pkg().PqtIndex(D, P, C1, C2, W, LP)
    try:
        idx.set_codebooks(cb1, cb2)
        idx.set_option("enumerate_beyond_wrap", 1)
        idx.set_heuristic(prefix)
        idx.set_bins(*o.export_bins())
        idx.set_lines(o.export_codes())
        o.set_sort_mode(1)
        total = 0
        for bv, bb, k in ((200, 600, 32), (50, 100, 8), (10 ** 6, 600, 200)):
            ids, dist, cnt = idx.query(queries, bv, bb, k)
            for qi in range(nq):
                s_ids, s_d = o.query(queries[qi], bv, bb)
                kk = min(k, len(s_ids))
                assert int(cnt[qi]) == len(s_ids), (bv, bb, qi, int(cnt[qi]), len(s_ids))
                assert np.array_equal(ids[qi, :kk], s_ids[:kk]) and np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk])), (bv, bb, k, qi)
                total += len(s_ids)
        assert total > 10 * nq  # the lists are not empty: rows really are enumerated now
        # default behaviour restored by the option: the reference's empty lists
        idx.set_option("enumerate_beyond_wrap", 0)
        _, _, cnt0 = idx.query(queries, 200, 600, 8)
        assert np.all(cnt0 == 0)
    finally:
        idx.close()


def test_cross_lane_primitives_exchange_and_scan():
    """pqt_lane_xor_u32<LM> (the compare-exchange partner of every sorting network: DPP moves for LM <= 8, the gfx950 row / half swaps
    v_permlane16_swap / v_permlane32_swap for LM = 16, 32) and the DPP wave scan, against their definitions."""
    pkg = pqt_pkg()
    got = pkg.debug_sort_scan(6, 64)
    lane = np.arange(64, dtype=np.uint64)
    for i in range(6):
        want = ((np.uint64(0x9e3779b9) * ((lane ^ np.uint64(1 << i)) + np.uint64(1))) & np.uint64(0xffffffff)).astype(np.uint32)
        assert np.array_equal(got[i * 64:(i + 1) * 64], want), "lane ^ %d" % (1 << i)
    vals = ((np.uint64(0x9e3779b9) * (lane + np.uint64(1))) & np.uint64(0xffffffff)).astype(np.uint32)
    assert np.array_equal(got[6 * 64:7 * 64], np.concatenate([vals[1:], vals[63:]])), "lane + 1"
    assert np.array_equal(got[7 * 64:8 * 64], np.sort(vals | np.uint32(1))), "pqt_wave_sort_u32"
    v = ((lane * np.uint64(2654435761)) & np.uint64(0xffffffff)) >> np.uint64(24)
    assert np.array_equal(pkg.debug_sort_scan(7, 64)[:64], np.cumsum(v).astype(np.uint32))
    # the traversal's part sorts: four 64-key lists sorted at once, one per 16-lane row, 4 keys per lane
    e = np.arange(256, dtype=np.uint64)
    keys = (((np.uint64(0x9e3779b9) * (e + np.uint64(1))) & np.uint64(0xffffffff)).astype(np.uint32) | np.uint32(1)).reshape(4, 64)
    assert np.array_equal(pkg.debug_sort_scan(8, 256)[:256].reshape(4, 64), np.sort(keys, axis=1)), "pqt_row_sort64_u32"


@pytest.mark.parametrize("name", ["cfg2_small", "cfg2_dense", "cfg3_small"])
def test_row_parallel_part_sorts_equal_the_one_list_at_a_time_sorts(name):
    """Compile-time shapes: the four second-level part lists of a query are sorted together, one per 16-lane row (pqt_row_sort64_u32, keys
    = distance key with the position in its low 6 bits); a query in which two neighbours of a sorted list agree in the upper 26 bits takes
    the one-list-at-a-time code that settles such pairs exactly.  Option exact_part_sorts = 1 sends EVERY query through that code: the
    result lists must be the same bit for bit, and the checker's."""
    f = fixture(name)
    idx = f.hip_index()
    try:
        Bv, Bb = BV_BB[name]
        f.oracle.set_sort_mode(1)
        got = {}
        for ex in (0, 1):
            idx.set_option("exact_part_sorts", ex)
            got[ex] = idx.query(f.queries, Bv, Bb, 128)
            assert "-shape" in idx.last_path(), idx.last_path()
        assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got[0], got[1])), name
        ids, dist, cnt = got[1]
        for qi, q in enumerate(f.queries):
            s_ids, s_d = f.oracle.query(q, Bv, Bb)
            kk = min(128, len(s_ids))
            assert int(cnt[qi]) == len(s_ids) and np.array_equal(ids[qi, :kk], s_ids[:kk]) and np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk])), (name, qi)
    finally:
        idx.set_option("exact_part_sorts", 0)
        f.oracle.set_sort_mode(0)
        idx.close()


@pytest.mark.parametrize("name", ["cfg2_small", "cfg2_dense", "ties", "wrap"])
def test_xcode_rows_and_the_one_pass_selection_change_no_bit(name):
    """The exact rerank with the LDS coarse table at C1 = 32 reads the X-code copy of the line store (coarse offset precomputed in the
    code word, two candidates per packed instruction, 16 wavefronts per workgroup) and selects once over 32-bit distance keys held by
    visiting position; option xcode = 0 runs the plain store with the 12-wavefront kernel.  Both must return the checker's lists bit for
    bit -- including exact distance ties (fixture "ties": duplicated vectors), lists longer than the 32-bit key area (a second,
    threshold-filtered phase), and k at both ends."""
    f = fixture(name)
    idx = f.hip_index()
    try:
        Bv0, Bb = BV_BB[name]
        f.oracle.set_sort_mode(1)
        for Bv, k in ((Bv0, 100), (10 ** 6, 128), (10 ** 6, 1), (Bv0, 7)):
            got = {}
            for xc in (1, 0):
                idx.set_option("xcode", xc)
                got[xc] = idx.query(f.queries, Bv, Bb, k)
                path = idx.last_path()
                want_x = xc == 1 and CONFIGS[name]["C1"] == 32 and CONFIGS[name]["LP"] * 32 * 32 * 4 <= 65536
                assert ("-xcode" in path) == want_x, (name, xc, path)
            assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got[0], got[1])), (name, Bv, k)
            ids, dist, cnt = got[1]
            for qi, q in enumerate(f.queries):
                s_ids, s_d = f.oracle.query(q, Bv, Bb)
                kk = min(k, len(s_ids))
                assert int(cnt[qi]) == len(s_ids)
                assert np.array_equal(ids[qi, :kk], s_ids[:kk]) and np.array_equal(bits(dist[qi, :kk]), bits(s_d[:kk])), (name, Bv, k, qi)
        if name == "cfg2_dense":
            assert int(cnt.max()) > 768  # the second phase (lists beyond the 768 32-bit slots of a 16-wavefront workgroup) is exercised
    finally:
        f.oracle.set_sort_mode(0)
        idx.close()


@pytest.mark.parametrize("knobs", [(400, 500), (5000, 500), (10 ** 6, 512), (3000, 64), (1, 500)])
def test_shared_row_pass_changes_no_bit(knobs):
    """BASELINE configs[2]/[3] shape: the rows of a bin read once for all the queries -- and all the visits of one query: the bin id drops
    part 3 at this shape, (C1*C2)^3 wraps to 0 in uint32 (treequantizer.hpp:45-49,572), so a query visits the same bin once per aliased
    tuple -- that include it (pqt_shared_rows.h, option "shared_rows"; automatic only for line stores of 1 GiB and more).  Same ids, distance
    bits and counts as the wave-per-query filter kernel and as the checker; range shards as well; the queries whose run list does not fit
    the hand-over (64 runs) evaluate their rows in the selection kernel."""
    import torch
    bv, bb = knobs
    f = fixture("cfg3_small")
    idx = f.hip_index()
    n = f.oracle.num_vectors
    shards = [f.hip_index(shard=(0, n // 3)), f.hip_index(shard=(n // 3, n))]
    try:
        k = 100
        a = idx.query(f.queries, bv, bb, k)
        assert "-shared" not in idx.last_path()
        idx.set_option("shared_rows", 1)
        b = idx.query(f.queries, bv, bb, k)
        assert "rerank=mode2-nw12-runs-shared" in idx.last_path(), idx.last_path()
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
        # the checker (duplicates of an aliased bin appear once per visit, in visiting order)
        f.oracle.set_sort_mode(1)
        dup = 0
        for qi, q in enumerate(f.queries):
            s_ids, s_d = f.oracle.query(q, bv, bb)
            kk = min(k, len(s_ids))
            assert int(b[2][qi]) == len(s_ids)
            assert np.array_equal(b[0][qi, :kk], s_ids[:kk]) and np.array_equal(bits(b[1][qi, :kk]), bits(s_d[:kk])), qi
            dup += len(s_ids) - len(np.unique(s_ids))
        if bv >= 3000 and bb >= 500:
            assert dup > 0, "fixture no longer visits a bin twice"
        # range shards with the pass, merged
        q = torch.from_numpy(f.queries).cuda()
        qn = q.shape[0]
        I = torch.empty((2, qn, k), dtype=torch.int32, device="cuda")
        Dd = torch.empty((2, qn, k), dtype=torch.float32, device="cuda")
        Pp = torch.empty((2, qn, k), dtype=torch.int32, device="cuda")
        Cc = torch.empty((2, qn), dtype=torch.int32, device="cuda")
        for s, sh in enumerate(shards):
            sh.set_option("shared_rows", 1)
            sh.query_shard_dev(q, bv, bb, k, I[s], Dd[s], Pp[s], Cc[s], sync=True)
            assert "-shared" in sh.last_path(), sh.last_path()
        oI = torch.empty((qn, k), dtype=torch.int32, device="cuda")
        oD = torch.empty((qn, k), dtype=torch.float32, device="cuda")
        shards[0].merge_topk_dev(2, qn, k, I, Dd, Pp, oI, oD, sync=True)
        assert np.array_equal(oI.cpu().numpy().view(np.uint32), a[0]) and np.array_equal(bits(oD.cpu().numpy()), bits(a[1]))
        # a view of the index (second batch in flight) runs the pass on its own scratch
        v = idx.view()
        try:
            c = v.query(f.queries[::-1].copy(), bv, bb, k)
            assert "-shared" in v.last_path()
            assert np.array_equal(c[0], a[0][::-1]) and np.array_equal(bits(c[1]), bits(a[1][::-1]))
        finally:
            v.close()
    finally:
        f.oracle.set_sort_mode(0)
        idx.close()
        for sh in shards:
            sh.close()


def test_shared_row_pass_keeps_the_tie_cluster_fallback():
    """The band overflow of the filter (hundreds of exactly tied candidates around the k-th distance) still sends the query to the plain
    exact kernel when the distances come from the shared-row pass."""
    from common import Fixture

    def clustered(n, D, seed):
        protos = np.random.default_rng(777).integers(0, 256, (20, D)).astype(np.float32)
        rng = np.random.default_rng(seed)
        x = protos[rng.integers(0, 20, n)]
        noisy = rng.random(n) < 0.5
        x[noisy] = np.clip(np.rint(x[noisy] + rng.normal(0, 25, (int(noisy.sum()), D))), 0, 255)
        return x.astype(np.float32)

    f = Fixture(D=64, P=2, C1=64, C2=4, W=2, LP=32, n_base=12000, n_query=8, seed=68, heur_rows=64, train=3000, data=clustered)
    idx = f.hip_index()
    try:
        a = idx.query(f.queries, 10 ** 6, 64, 100)
        fa = idx.stats()["filter_fallbacks"]
        idx.set_option("shared_rows", 1)
        b = idx.query(f.queries, 10 ** 6, 64, 100)
        assert "-shared" in idx.last_path()
        assert idx.stats()["filter_fallbacks"] == fa and fa > 0
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
    finally:
        idx.close()


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_small"])
def test_wide_enumeration_with_the_lds_first_level_of_the_bitmap(name):
    """Option "filter_l1" (pqt_k_traverse_f1, opt-in: measured no faster): the wide enumeration asks the folded first level of the presence bitmap
    in LDS before the bitmap word itself; a clear first-level bit proves the bit clear, so nothing may change."""
    f = fixture(name)
    idx = f.hip_index()
    try:
        for bv, bb in ((400, 4096), (3000, 1024)):
            if bb > f.heur.shape[0]:
                idx.build_heuristic(bb)
            a = idx.query(f.queries, bv, bb, 64)
            idx.set_option("filter_l1", 1)
            pa = idx.last_path()
            b = idx.query(f.queries, bv, bb, 64)
            pb = idx.last_path()
            idx.set_option("filter_l1", 0)
            # the token says pqt_k_traverse_f1 really ran (the launcher falls back silently when the first level does not exist or fit)
            assert "fused-wide" in pa and "-f1" not in pa and "fused-wide" in pb and "-f1" in pb and "-f1c" not in pb, (pa, pb)
            assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2]), (bv, bb)
    finally:
        idx.close()
