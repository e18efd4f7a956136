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
    yield pkg, w, idx, base, meta, queries
    idx.close()


def run(idx, q, bv, bb, k):
    import torch
    qn = q.shape[0]
    oi = torch.empty((qn, k), dtype=torch.int32, device=q.device)
    od = torch.empty((qn, k), dtype=torch.float32, device=q.device)
    oc = torch.empty(qn, dtype=torch.int32, device=q.device)
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
    # a tighter vector bound really cuts, and the cut list is a prefix-in-visiting-order subset of the uncut one
    ids_c, dist_c, cnt_c = run(idx, queries[:256], 50, bb, 100)
    assert np.all(cnt_c <= cnt[:256]) and np.any(cnt_c < cnt[:256])
    assert np.all(cnt_c <= 50 + meta["max_bin"])


def test_fullsize_oracle_spot_check(big):
    """64 queries of the 10 k batch against the oracle loaded with the same 1 M-vector index."""
    from oracle import Oracle
    pkg, w, idx, base, meta, queries = big
    o = Oracle(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], heur_keep=1)
    o.set_heuristic(idx.heuristic(500))
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
