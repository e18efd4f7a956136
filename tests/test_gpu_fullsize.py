"""Full-size checks at BASELINE.json configs[1] (SIFT1M shape: N = 1 M, d=128 p=4 c1=c2=32 lineparts=16, 10 k queries).

The oracle cannot finish this size in test time, so parity is checked through size-independent properties of the
path plus an oracle spot-check on a sample of the batch:
  * sortedness: every result list is ascending in distance, padding only at the tail, a repeated id (wrapped-bin
    aliasing, as in the reference) repeats its distance;
  * prefix property: the top-10 list is the prefix of the top-100 list (same bounds);
  * idempotence / determinism: the same batch twice, and the batch split in ragged pieces, give identical bytes;
  * structure independence: wave-per-query fused kernels == workgroup-per-query staged kernels, bit for bit;
  * count identity: the per-query candidate counts sum to the engine's own statistic; every count respects the
    reference's cut rule bound (count <= Bv + largest bin);
  * membership: every returned id belongs to a bin that the oracle's traversal of that query visits (sample);
  * oracle spot-check: 64 queries of the batch, full equality of ids and distance bits.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    import importlib
    import torch
    import bench
    pkg = importlib.import_module("product-quantization-tree_amd")
    w = bench.WORKLOADS["sift1m"]
    idx, base, meta = bench.build_index(pkg, w, 0)
    idx.build_heuristic(500)
    queries = bench.sift_like(w["qn"], w["D"], 0xC0DE03, torch.device("cuda", 0))
    torch.cuda.synchronize()  # the library enqueues on its own stream when handed torch's (NULL) default stream
    assert meta["max_bin"] < 5000, "degenerate database (data synthesis and build kernel out of order?)"
    yield pkg, w, idx, base, meta, queries
    idx.close()


def run(idx, q, bv, bb, k):
    import torch
    qn = q.shape[0]
    oi = torch.empty((qn, k), dtype=torch.int32, device=q.device)
    od = torch.empty((qn, k), dtype=torch.float32, device=q.device)
    oc = torch.empty(qn, dtype=torch.int32, device=q.device)
    torch.cuda.synchronize()  # q may be the product of a torch op still in flight on the default stream; the library uses its own
    idx.query_dev(q, bv, bb, k, oi, od, oc, sync=True)
    return oi.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy().view(np.uint32)


def test_fullsize_properties(big):
    pkg, w, idx, base, meta, queries = big
    bv, bb = 20000, 500
    ids, dist, cnt = run(idx, queries, bv, bb, 100)
    qn = ids.shape[0]
    n_valid = np.minimum(cnt, 100)
    # sortedness + padding + uniqueness
    for qi in range(0, qn, 7):
        n = int(n_valid[qi])
        d = dist[qi, :n]
        assert np.all(d[1:] >= d[:-1])
        assert np.all(ids[qi, n:] == 0xffffffff) and np.all(np.isinf(dist[qi, n:]))
        # (C1*C2)^P = 2^40 wraps in uint32: tuples that differ only in the lost high digits alias to one bin, which is
        # then visited twice -- the reference's list holds such ids twice too; a repeated id repeats its distance
        for v in np.unique(ids[qi, :n]):
            assert len(np.unique(dist[qi, :n][ids[qi, :n] == v].view(np.uint32))) == 1
        assert ids[qi, :n].max(initial=0) < w["n_base"]
    # count identity and the cut-rule bound
    st = idx.stats()
    assert int(cnt.astype(np.int64).sum()) == st["candidates"]
    assert int(cnt.max()) <= bv + meta["max_bin"]
    # prefix property
    ids10, dist10, cnt10 = run(idx, queries, bv, bb, 10)
    assert np.array_equal(ids10, ids[:, :10]) and np.array_equal(dist10.view(np.uint32), dist[:, :10].view(np.uint32))
    assert np.array_equal(cnt10, cnt)
    # idempotence + ragged split
    ids2, dist2, cnt2 = run(idx, queries, bv, bb, 100)
    assert np.array_equal(ids2, ids) and np.array_equal(dist2.view(np.uint32), dist.view(np.uint32))
    pieces = [run(idx, queries[a:b], bv, bb, 100) for a, b in ((0, 1), (1, 4097), (4097, qn))]
    assert np.array_equal(np.concatenate([p[0] for p in pieces]), ids)
    assert np.array_equal(np.concatenate([p[1] for p in pieces]).view(np.uint32), dist.view(np.uint32))
    # structure independence
    idx.set_option("fused", 0)
    try:
        ids_s, dist_s, cnt_s = run(idx, queries, bv, bb, 100)
    finally:
        idx.set_option("fused", 1)
    assert np.array_equal(ids_s, ids) and np.array_equal(dist_s.view(np.uint32), dist.view(np.uint32)) and np.array_equal(cnt_s, cnt)
    # schedule independence: static round-robin, workgroup-local lists, global pools (the default) -- also behind the wide traversal
    for knobs in ((bv, bb), (4096, 4096)):
        idx.build_heuristic(max(knobs[1], 500))
        ref = None
        for bal in (2, 1, 0):
            idx.set_option("balance", bal)
            try:
                got = run(idx, queries, knobs[0], knobs[1], 100)
            finally:
                idx.set_option("balance", -1)
            if ref is None:
                ref = got
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(got[2], ref[2]), (knobs, bal)
        if knobs == (bv, bb):
            assert np.array_equal(ref[0], ids) and np.array_equal(ref[2], cnt)
    # a tighter vector bound really cuts, and the cut list is a prefix-in-visiting-order subset of the uncut one
    ids_c, dist_c, cnt_c = run(idx, queries[:256], 50, bb, 100)
    assert np.all(cnt_c <= cnt[:256]) and np.any(cnt_c < cnt[:256])
    assert np.all(cnt_c <= 50 + meta["max_bin"])


def test_fullsize_oracle_spot_check(big):
    """64 queries of the 10 k batch against the oracle loaded with the same 1 M-vector index."""
    from oracle import Oracle
    pkg, w, idx, base, meta, queries = big
    o = Oracle(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], heur_keep=500)  # the checker builds its OWN 64^4-tuple table
    assert np.array_equal(o.heuristic(500), idx.heuristic(500))
    o.set_codebooks(meta["cb1"], meta["cb2"])
    o.import_bins(meta["bin_ids"], meta["sizes"], meta["members"])
    o.import_codes(idx._keep[0].cpu().numpy().view(np.uint32))
    sample = np.arange(0, queries.shape[0], queries.shape[0] // 64)[:64]
    ids, dist, cnt = run(idx, queries[sample], 20000, 500, 100)
    qh = queries[sample].cpu().numpy()
    o.set_sort_mode(1)
    size_of = dict(zip(meta["bin_ids"].tolist(), range(len(meta["bin_ids"]))))
    starts = np.concatenate([[0], np.cumsum(meta["sizes"].astype(np.int64))])
    for i in range(len(sample)):
        s_ids, s_d = o.query(qh[i], 20000, 500)
        n = min(100, len(s_ids))
        assert int(cnt[i]) == len(s_ids)
        assert np.array_equal(ids[i, :n], s_ids[:n])
        assert np.array_equal(dist[i, :n].view(np.uint32), s_d[:n].view(np.uint32))
        # membership: every returned id lives in a bin the traversal of this query enumerates
        bins, _, _ = o.stage_bins(qh[i], 500)
        members = set()
        for b in bins.tolist():
            j = size_of.get(b)
            if j is not None:
                members.update(meta["members"][starts[j]:starts[j + 1]].tolist())
        assert set(ids[i, :n].tolist()) <= members
    # sensitivity of the result to the un-pinned float summation order (tests/test_cpu_sum_order.py) at full size: the same
    # 64 queries with the oracle's SSE2-packet order (what Eigen selects under the reference's own build flags) and with an
    # FMA-contracted loop, against the sequential order the engine implements
    ref_sets = []
    for i in range(len(sample)):
        ref_sets.append((set(o.query_unsorted(qh[i], 20000, 500)[0].tolist()), o.query(qh[i], 20000, 500)[0][:100]))
    try:
        for mode in (1, 5):
            o.set_sum_mode(mode)
            same_set = same_top = 0
            for i in range(len(sample)):
                same_set += set(o.query_unsorted(qh[i], 20000, 500)[0].tolist()) == ref_sets[i][0]
                same_top += np.array_equal(o.query(qh[i], 20000, 500)[0][:100], ref_sets[i][1])
            print("1M index, sum order %d: candidate sets identical %d/64, top-100 id lists identical %d/64" % (mode, same_set, same_top))
            assert same_set >= 61  # >= 0.95 (tests/test_cpu_sum_order.py MIN_SET_AGREEMENT)
    finally:
        o.set_sum_mode(0)


# =====================================================================================================================
# BASELINE.json configs[2]/[3] shape (d=128 p=4 c1=c2=64 lineparts=32, 128-byte code rows, workgroup-per-query rerank with
# the group-major store), chunk-built at 10 M vectors: the sizes at which the cfg3/cfg4 code paths run for real.
# =====================================================================================================================
@pytest.fixture(scope="module")
def big3():
    import importlib
    import torch
    import bench
    pkg = importlib.import_module("product-quantization-tree_amd")
    w = bench.WORKLOADS["synth10m"]
    idx, base, meta = bench.build_index(pkg, w, 0)
    assert base is None  # chunk-built: the raw vectors are never resident as a whole
    idx.build_heuristic(4096)
    queries = bench.sift_like(512, w["D"], 0xC0DE03, torch.device("cuda", 0))
    torch.cuda.synchronize()
    assert meta["max_bin"] < 50000, "degenerate database (data synthesis and build kernel out of order?)"
    yield pkg, w, idx, meta, queries
    idx.close()


@pytest.mark.parametrize("bv,bb", [(20000, 500), (4096, 4096)])
def test_cfg3_fullsize_properties(big3, bv, bb):
    pkg, w, idx, meta, queries = big3
    ids, dist, cnt = run(idx, queries, bv, bb, 100)
    qn = ids.shape[0]
    n_valid = np.minimum(cnt, 100)
    for qi in range(qn):
        n = int(n_valid[qi])
        d = dist[qi, :n]
        assert np.all(d[1:] >= d[:-1])
        assert np.all(ids[qi, n:] == 0xffffffff) and np.all(np.isinf(dist[qi, n:]))
        assert ids[qi, :n].max(initial=0) < w["n_base"]
    st = idx.stats()
    assert int(cnt.astype(np.int64).sum()) == st["candidates"]
    assert int(cnt.max()) <= bv + meta["max_bin"]
    assert int(cnt.max()) > 1000  # the rerank really works on long lists here
    ids10, dist10, cnt10 = run(idx, queries, bv, bb, 10)
    assert np.array_equal(ids10, ids[:, :10]) and np.array_equal(dist10.view(np.uint32), dist[:, :10].view(np.uint32))
    ids2, dist2, cnt2 = run(idx, queries, bv, bb, 100)
    assert np.array_equal(ids2, ids) and np.array_equal(dist2.view(np.uint32), dist.view(np.uint32))
    pieces = [run(idx, queries[a:b], bv, bb, 100) for a, b in ((0, 1), (1, 130), (130, qn))]
    assert np.array_equal(np.concatenate([p[0] for p in pieces]), ids)
    assert np.array_equal(np.concatenate([p[1] for p in pieces]).view(np.uint32), dist.view(np.uint32))
    # structure independence: workgroup-per-query fused rerank (group-major store) == staged kernels == wave-per-query kernel
    for opt, val, back in (("fused", 0, 1), ("wg_rerank", 0, 1)):
        idx.set_option(opt, val)
        try:
            ids_s, dist_s, cnt_s = run(idx, queries[:128], bv, bb, 100)
        finally:
            idx.set_option(opt, back)
        assert np.array_equal(ids_s, ids[:128]) and np.array_equal(dist_s.view(np.uint32), dist[:128].view(np.uint32)) and np.array_equal(cnt_s, cnt[:128]), opt
    # k > 128 (staged select) agrees with the fused top-100 on its prefix
    ids_k, dist_k, _ = run(idx, queries[:64], bv, bb, 1000)
    assert np.array_equal(ids_k[:, :100], ids[:64]) and np.array_equal(dist_k[:, :100].view(np.uint32), dist[:64].view(np.uint32))


def test_cfg3_fullsize_oracle_spot_check(big3):
    """64 queries against the oracle loaded with the same 10 M-vector index; the oracle builds its own heuristic table."""
    from oracle import Oracle
    pkg, w, idx, meta, queries = big3
    o = Oracle(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], heur_keep=4096)
    assert np.array_equal(o.heuristic(4096), idx.heuristic(4096))
    o.set_codebooks(meta["cb1"], meta["cb2"])
    o.import_bins(meta["bin_ids"], meta["sizes"], meta["members"])
    o.import_codes(idx._keep[0].cpu().numpy().view(np.uint32))
    o.set_sort_mode(1)
    qh = queries[:64].cpu().numpy()
    for bv, bb in ((20000, 500), (4096, 4096)):
        ids, dist, cnt = run(idx, queries[:64], bv, bb, 100)
        for i in range(64):
            s_ids, s_d = o.query(qh[i], bv, bb)
            n = min(100, len(s_ids))
            assert int(cnt[i]) == len(s_ids), (bv, bb, i)
            assert np.array_equal(dist[i, :n].view(np.uint32), s_d[:n].view(np.uint32)), (bv, bb, i)
            assert np.array_equal(ids[i, :n], s_ids[:n]), (bv, bb, i)


def test_cfg3_fullsize_eight_way_shards_equal_unsharded(big3):
    """The north-star layout on one device: 8 range shards described from the shard's side (pqt_index_set_bins_local with
    the counts of sharding.merge_bin_counts), each queried with pqt_query_shard, merged with pqt_merge_topk at 8 x k."""
    import importlib
    import torch
    pkg, w, idx, meta, queries = big3
    sharding = importlib.import_module("product-quantization-tree_amd.sharding")
    n, world, k = w["n_base"], 8, 100
    dev = queries.device
    bin_of_vec = torch.empty(n, dtype=torch.int64, device=dev)
    bin_of_vec[torch.from_numpy(meta["members"].astype(np.int64)).to(dev)] = \
        torch.repeat_interleave(torch.from_numpy(meta["bin_ids"].astype(np.int64)), torch.from_numpy(meta["sizes"].astype(np.int64))).to(dev)
    codes = idx._keep[0]
    ranges = [sharding.shard_range(r, world, n) for r in range(world)]
    local = [sharding.local_bin_lists(bin_of_vec[lo:hi], lo) for lo, hi in ranges]
    shards = []
    try:
        for r, (lo, hi) in enumerate(ranges):
            uk, gs, low, ls = sharding.merge_bin_counts([l[0] for l in local], [l[1] for l in local], r)
            assert int(gs.sum()) == n and int(ls.sum()) == hi - lo
            sh = pkg.PqtIndex(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], device=0)
            sh.set_codebooks(meta["cb1"], meta["cb2"])
            sh.build_heuristic(4096)
            sh.set_bins_local(uk.cpu().numpy(), gs.cpu().numpy(), low.cpu().numpy(), ls.cpu().numpy(), local[r][2].cpu().numpy(), n)
            sh.set_lines_dev(codes[lo:hi], lo)
            shards.append(sh)
        q = queries[:256]
        qn = q.shape[0]
        for bv, bb in ((20000, 500), (4096, 4096)):
            ref_ids, ref_d, ref_c = run(idx, q, bv, bb, k)
            pack = torch.empty((world, 3, qn, k), dtype=torch.int32, device=dev)
            Cc = torch.empty((world, qn), dtype=torch.int32, device=dev)
            tot_local = 0
            for s, sh in enumerate(shards):
                sh.query_shard_dev(q, bv, bb, k, pack[s, 0], pack[s, 1].view(torch.float32), pack[s, 2], Cc[s], sync=True)
                tot_local += sh.stats()["candidates"]
                assert np.array_equal(Cc[s].cpu().numpy().view(np.uint32), ref_c)  # every shard sees the global count
            assert tot_local == int(ref_c.astype(np.int64).sum())  # the shards partition the candidates
            # query-sharded traversal: shard s traverses query slice s only, the lists are concatenated (the all-gather), and every
            # shard's pqt_query_shard_bins reproduces its pqt_query_shard output -- with the schedule-2 registration coming from
            # the resolve kernel instead of the traversal
            cap, qs = sharding.BIN_CAP, (qn + world - 1) // world
            bins_all = torch.zeros((world * qs, cap + 1), dtype=torch.int64, device=dev)
            for s, sh in enumerate(shards):
                lo_q, hi_q = min(s * qs, qn), min((s + 1) * qs, qn)
                if hi_q > lo_q:
                    sh.traverse_bins_dev(q[lo_q:hi_q], bv, bb, cap, bins_all[lo_q:hi_q], sync=True)
            assert int(((bins_all[:qn, cap] & 0xffffffff) == 0xffffffff).sum()) == 0  # no list overflows 128 bins at this shape
            for s, sh in enumerate(shards):
                o = [torch.empty((qn, k), dtype=torch.int32, device=dev), torch.empty((qn, k), dtype=torch.float32, device=dev),
                     torch.empty((qn, k), dtype=torch.int32, device=dev), torch.empty(qn, dtype=torch.int32, device=dev)]
                sh.query_shard_bins_dev(q, bv, bb, k, bins_all, cap, o[0], o[1], o[2], o[3], sync=True)
                assert "traverse=bins-resolved" in sh.last_path()
                assert torch.equal(o[0], pack[s, 0]) and torch.equal(o[1].view(torch.int32), pack[s, 1]) and torch.equal(o[2], pack[s, 2]) and torch.equal(o[3], Cc[s]), (bv, bb, s)
            oI = torch.empty((qn, k), dtype=torch.int32, device=dev)
            oD = torch.empty((qn, k), dtype=torch.float32, device=dev)
            shards[0].merge_topk_dev(world, qn, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oI, oD, sync=True, shard_stride=3 * qn * k)
            assert np.array_equal(oI.cpu().numpy().view(np.uint32), ref_ids), (bv, bb)
            assert np.array_equal(oD.cpu().numpy().view(np.uint32), ref_d.view(np.uint32)), (bv, bb)
    finally:
        for sh in shards:
            sh.close()
