"""Seeded sweep over odd shapes: every template path of the fused and staged kernels against the oracle.

Covers what the named fixtures do not: P = 1 / 8, W*C2 in (64,128] and (128,256] (2 and 4 keys per lane in the a2 sort),
lineparts 4 / 8 (1 and 2 code vectors per row), non-power-of-two C1 in the fused rerank, segment lengths that are not a
multiple of 4 (no cb2 tiles), C2 = 1, W = C1, boundBins around the 512 limit of the fused traversal, k around 128.
"""
import numpy as np
import pytest

from common import Fixture

pytestmark = pytest.mark.gpu

#        D   P  C1  C2  W  LP   n     bv    bb    k
SHAPES = [
    (32,  1, 16, 8,  4, 4,  3000, 300,  32,   20),    # P = 1, LP = 4
    (64,  8, 4,  4,  2, 8,  3000, 200,  300,  128),   # P = 8
    (48,  2, 12, 64, 2, 12, 4000, 500,  400,  64),    # W*C2 = 128 (2 keys/lane), C1 not a power of two, LP = 12
    (64,  2, 8,  64, 4, 16, 4000, 800,  512,  100),   # W*C2 = 256 (4 keys/lane), boundBins = 512 exactly
    (40,  2, 10, 6,  3, 10, 3000, 300,  513,  129),   # just above both fused limits -> staged kernels, S = 20
    (36,  2, 6,  5,  6, 6,  2500, 400,  200,  7),     # W = C1, S = 18 (not a multiple of 4: no cb2 tiles), LP = 6 (scalar code reads)
    (32,  4, 16, 1,  3, 8,  3000, 150,  81,   50),    # C2 = 1
    (128, 4, 32, 16, 2, 32, 5000, 1000, 500,  100),   # 128-byte rows with coarse in LDS? (32*32*32*4 = 128 KB -> staged slices)
    (64,  2, 64, 4,  2, 32, 4000, 600,  64,   33),    # C1 = 64, LP = 32: workgroup-per-query rerank
    (24,  3, 5,  3,  2, 6,  1500, 100,  216,  10),    # everything odd
    (64,  2, 128, 4, 2, 32, 6000, 600,  64,   33),    # C1 = 128, LP = 32 (configs[4]'s first level): 16 KB of L1virt per wave, 6-wave workgroups
    (128, 4, 64, 64, 1, 32, 6000, 900,  300,  100),   # BASELINE configs[2]/[3] shape at a different seed
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "D%d_P%d_C%dx%d_W%d_LP%d" % s[:6])
@pytest.mark.parametrize("mode", ["fused", "staged"])
def test_shape_sweep(shape, mode):
    D, P, C1, C2, W, LP, n, bv, bb, k = shape
    rows = min(bb, (W * C2) ** P)
    f = Fixture(D=D, P=P, C1=C1, C2=C2, W=W, LP=LP, n_base=n, n_query=10, seed=1000 + D + 7 * C1 + LP, heur_rows=rows,
                train=min(n, 2500))
    idx = f.hip_index()
    try:
        idx.set_option("fused", 1 if mode == "fused" else 0)
        ids, dist, cnt = idx.query(f.queries, bv, bb, k)
        st = idx.stats()
        f.oracle.set_sort_mode(1)
        try:
            for qi, q in enumerate(f.queries):
                s_ids, s_d = f.oracle.query(q, bv, bb)
                kk = min(k, len(s_ids))
                assert int(cnt[qi]) == len(s_ids), (qi, int(cnt[qi]), len(s_ids))
                assert np.array_equal(dist[qi, :kk].view(np.uint32), s_d[:kk].view(np.uint32)), qi
                assert np.array_equal(ids[qi, :kk], s_ids[:kk]), qi
                assert np.all(ids[qi, kk:] == 0xffffffff)
        finally:
            f.oracle.set_sort_mode(0)
        assert st["candidates"] == int(cnt.astype(np.int64).sum())
    finally:
        idx.close()


def _random_case(seed):
    """A random valid (shape, knobs, options) combination; (W*C2)^P kept small enough for the oracle's full tuple table."""
    rng = np.random.default_rng(9000 + seed)
    while True:
        D = int(rng.choice([32, 64, 128]))
        P = int(rng.choice([1, 2, 4]))
        LP = int(rng.choice([4, 8, 16, 32]))
        if D % P or D % LP or LP % P:
            continue
        C1 = int(rng.choice([4, 8, 16, 32, 64]))
        C2 = int(rng.choice([2, 4, 8, 16, 32, 64]))
        W = int(rng.integers(1, min(C1, 4) + 1))
        if W * C2 > 256 or (W * C2) ** P > (1 << 20) or (W * C2) ** P < 8:
            continue
        break
    n = int(rng.integers(1500, 6000))
    bv = int(rng.choice([0, 50, 500, 10 ** 6]))
    bb = int(rng.integers(1, min(4096, (W * C2) ** P) + 1))
    k = int(rng.choice([1, 17, 100, 128, 129, 700, 4096]))
    opts = {"bin_runs": int(rng.choice([-1, 0, 1])), "exact_filter": int(rng.integers(0, 2)), "static_shapes": int(rng.integers(0, 2)),
            "order_all_rows": int(rng.integers(0, 2)), "balance": int(rng.integers(0, 3)), "wg_rerank": int(rng.integers(0, 2))}
    return (D, P, C1, C2, W, LP, n, bv, bb, k), opts


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PQT_FUZZ_N", "24"))))
def test_random_shapes_knobs_and_options(seed):
    """Randomised sweep (seeded): shape x bounds x k x every result-neutral option, against the oracle."""
    (D, P, C1, C2, W, LP, n, bv, bb, k), opts = _random_case(seed)
    f = Fixture(D=D, P=P, C1=C1, C2=C2, W=W, LP=LP, n_base=n, n_query=6, seed=2000 + seed, heur_rows=bb, train=min(n, 2000))
    idx = f.hip_index()
    try:
        for name, val in opts.items():
            idx.set_option(name, val)
        ids, dist, cnt = idx.query(f.queries, bv, bb, k)
        f.oracle.set_sort_mode(1)
        try:
            for qi, q in enumerate(f.queries):
                s_ids, s_d = f.oracle.query(q, bv, bb)
                kk = min(k, len(s_ids))
                ctx = (seed, (D, P, C1, C2, W, LP, n, bv, bb, k), opts, qi)
                assert int(cnt[qi]) == len(s_ids), ctx
                assert np.array_equal(dist[qi, :kk].view(np.uint32), s_d[:kk].view(np.uint32)), ctx
                assert np.array_equal(ids[qi, :kk], s_ids[:kk]), ctx
                assert np.all(ids[qi, kk:] == 0xffffffff), ctx
        finally:
            f.oracle.set_sort_mode(0)
    finally:
        idx.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PQT_FUZZ_SHARD_N", "10"))))
def test_random_range_shards_merge_to_the_unsharded_result(seed):
    """Randomised: 2..5 range shards at random cut points (some of them empty), every result-neutral option, merged with
    pqt_merge_topk == the unsharded engine on the same index."""
    import torch
    (D, P, C1, C2, W, LP, n, bv, bb, k), opts = _random_case(500 + seed)
    k = min(k, 700)
    rng = np.random.default_rng(77 + seed)
    world = int(rng.integers(2, 6))
    cuts = np.sort(rng.integers(0, n + 1, world - 1)).tolist()
    bounds = [0] + cuts + [n]
    f = Fixture(D=D, P=P, C1=C1, C2=C2, W=W, LP=LP, n_base=n, n_query=6, seed=3000 + seed, heur_rows=bb, train=min(n, 2000))
    ref = f.hip_index()
    shards = [f.hip_index(shard=(bounds[r], bounds[r + 1])) for r in range(world)]
    try:
        for name, val in opts.items():
            for h in shards + [ref]:
                h.set_option(name, val)
        ref_ids, ref_d, ref_c = ref.query(f.queries, bv, bb, k)
        q = torch.from_numpy(f.queries).cuda()
        qn = q.shape[0]
        pack = torch.empty((world, 3, qn, k), dtype=torch.int32, device="cuda")
        cnt = torch.empty((world, qn), dtype=torch.int32, device="cuda")
        for s, sh in enumerate(shards):
            sh.query_shard_dev(q, bv, bb, k, pack[s, 0], pack[s, 1].view(torch.float32), pack[s, 2], cnt[s], sync=True)
            assert np.array_equal(cnt[s].cpu().numpy().view(np.uint32), ref_c), (seed, s)
        oi = torch.empty((qn, k), dtype=torch.int32, device="cuda")
        od = torch.empty((qn, k), dtype=torch.float32, device="cuda")
        shards[0].merge_topk_dev(world, qn, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oi, od, sync=True, shard_stride=3 * qn * k)
        ctx = (seed, (D, P, C1, C2, W, LP, n, bv, bb, k), opts, bounds)
        assert np.array_equal(oi.cpu().numpy().view(np.uint32), ref_ids), ctx
        assert np.array_equal(od.cpu().numpy().view(np.uint32), ref_d.view(np.uint32)), ctx
    finally:
        for h in shards + [ref]:
            h.close()
