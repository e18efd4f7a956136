#!/usr/bin/env python3
"""Writes tests/golden/dump_small.{tree,bins} + dump_small_expected.npz.

An index dump pair in the reference's own on-disk formats (treequantizer::saveTree / saveBins, treequantizer.hpp:699-774)
at the reference tools' default template parameters (cpu_version/tools/query.cpp:10-15: D=128 C1=16 C2=8 P=2 W=4 LP=32),
written by the oracle restatement, plus the sorted candidate lists the oracle returns for 16 queries at
query(1500, 400).  tests/test_gpu_tools.py loads the dumps through the product's loadTree/loadBins and must reproduce the
lists; tests/test_cpu_oracle.py re-loads them into the oracle.  A dump pair written by a real build of the reference
(where Eigen exists) can be dropped in at the same paths: the expected lists then come from that build's query() output.
NOT reference-pinned as committed (the writer is the restatement)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def fixture():
    from common import Fixture
    return Fixture(D=128, P=2, C1=16, C2=8, W=4, LP=32, n_base=3000, n_query=16, seed=4242, heur_rows=1024, train=1500)


def inputs_only(tmp):
    """oracle/make_ref_fixtures.sh step 1: the inputs a genuine reference build needs (tree dump in the reference's format, raw vectors)."""
    f = fixture()
    f.oracle.save_tree(os.path.join(tmp, "dump_small.tree"))
    f.base.astype(np.float32).tofile(os.path.join(tmp, "base.raw"))
    f.queries.astype(np.float32).tofile(os.path.join(tmp, "queries.raw"))
    open(os.path.join(tmp, "n"), "w").write(str(f.base.shape[0]))
    open(os.path.join(tmp, "nq"), "w").write(str(f.queries.shape[0]))


def install_reference(tmp):
    """oracle/make_ref_fixtures.sh step 4: the reference's own saveBins output and query() lists become the golden fixture."""
    import shutil
    f = fixture()
    raw = np.fromfile(os.path.join(tmp, "ref.lists"), np.uint32)
    ids, dist, n_each, o = [], [], [], 0
    for _ in range(f.queries.shape[0]):
        m = int(raw[o]); o += 1
        pairs = raw[o:o + 2 * m].reshape(m, 2); o += 2 * m
        ids.append(pairs[:, 0].copy()); dist.append(pairs[:, 1].copy().view(np.float32)); n_each.append(m)
    shutil.copy(os.path.join(tmp, "dump_small.tree"), os.path.join(HERE, "dump_small.tree"))
    shutil.copy(os.path.join(tmp, "ref.bins"), os.path.join(HERE, "dump_small.bins"))
    np.savez_compressed(os.path.join(HERE, "dump_small_expected.npz"), queries=f.queries, n_each=np.array(n_each, np.uint32),
                        ids=np.concatenate(ids), dist=np.concatenate(dist), cfg=np.array([128, 2, 16, 8, 4, 32], np.uint32),
                        bv_bb=np.array([1500, 400], np.uint32), pinned_by=np.array(["reference"]))
    print("installed a genuine reference run as tests/golden/dump_small.*")


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--inputs-only":
        return inputs_only(sys.argv[2])
    if len(sys.argv) == 3 and sys.argv[1] == "--install-reference":
        return install_reference(sys.argv[2])
    f = fixture()
    f.oracle.save_tree(os.path.join(HERE, "dump_small.tree"))
    f.oracle.save_bins(os.path.join(HERE, "dump_small.bins"))
    bv, bb = 1500, 400
    f.oracle.set_sort_mode(1)
    outs = [f.oracle.query(q, bv, bb) for q in f.queries]
    n_each = np.array([len(o[0]) for o in outs], np.uint32)
    np.savez_compressed(os.path.join(HERE, "dump_small_expected.npz"), queries=f.queries, n_each=n_each,
                        ids=np.concatenate([o[0] for o in outs]), dist=np.concatenate([o[1] for o in outs]),
                        cfg=np.array([128, 2, 16, 8, 4, 32], np.uint32), bv_bb=np.array([bv, bb], np.uint32))
    print("dump fixture written:", {n: os.path.getsize(os.path.join(HERE, n)) for n in ("dump_small.tree", "dump_small.bins", "dump_small_expected.npz")})


if __name__ == "__main__":
    main()
