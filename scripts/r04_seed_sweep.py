"""Seed sweep at the two compile-time traversal shapes (row-parallel part sorts): HIP lists against the checker, bit for bit, and against
the one-list-at-a-time part sorts (option exact_part_sorts).  Run from the repository root on a GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from common import Fixture
bad = 0
for shape in (dict(D=128, P=4, C1=32, C2=32, W=2, LP=16), dict(D=128, P=4, C1=64, C2=64, W=1, LP=32)):
    for seed in range(301, 313):
        f = Fixture(n_base=12000, n_query=48, seed=seed, heur_rows=512, train=5000, **shape)
        idx = f.hip_index()
        f.oracle.set_sort_mode(1)
        for bv, bb, k in ((400, 500, 100), (10 ** 6, 512, 128)):
            idx.set_option("exact_part_sorts", 0)
            a = idx.query(f.queries, bv, bb, k)
            assert "-shape" in idx.last_path(), idx.last_path()
            idx.set_option("exact_part_sorts", 1)
            b = idx.query(f.queries, bv, bb, k)
            same = all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))
            ok = True
            for qi, q in enumerate(f.queries):
                s_ids, s_d = f.oracle.query(q, bv, bb)
                kk = min(k, len(s_ids))
                ok &= int(a[2][qi]) == len(s_ids) and np.array_equal(a[0][qi, :kk], s_ids[:kk]) and np.array_equal(a[1][qi, :kk].view(np.uint32), s_d[:kk].view(np.uint32))
            if not (same and ok):
                bad += 1
            print("C1=%d seed %d (%d, %d, %d): both sorts identical %s, checker identical %s, ties l2 %d" % (shape["C1"], seed, bv, bb, k, same, ok, idx.stats()["ties_l2"]), flush=True)
        idx.close()
print("FAILURES:", bad)
