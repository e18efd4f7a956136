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
