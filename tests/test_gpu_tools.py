"""End-to-end through the kept front-end: files in the reference's formats -> tool_createdb -> tool_query.

tool_createdb's .bins dump must be byte-identical to the oracle's saveBins (treequantizer.hpp:745-774) for the
same tree and dataset; tool_query's recall lines must equal the recall computed from the oracle's result lists.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from common import ROOT, fixture

pytestmark = pytest.mark.gpu
HOST = os.path.join(ROOT, "product-quantization-tree_amd", "host")


def write_umem(path, arr, dtype):
    arr = np.ascontiguousarray(arr, dtype)
    hdr = ("%d\n%d\n" % arr.shape).encode().ljust(20, b"\0")
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(arr.tobytes())


def test_createdb_then_query_roundtrip(tmp_path):
    if not os.path.exists(os.path.join(HOST, "tool_query")):
        subprocess.check_call(["make", "-C", HOST])
    f = fixture("tools_default")
    c = f.cfg
    os.chdir(tmp_path)
    pre = "t_%d_%d_%d_%d" % (c["D"], c["P"], c["C1"], c["C2"])
    # .ppqt: ASCII header dim,p,p2,C1,C2,nDBs + 1 separator byte + cb1 + cb2 (PerturbationProTree.cu:60-116)
    with open(pre + ".ppqt", "wb") as fh:
        fh.write(("%d\n%d\n%d\n%d\n%d\n%d\n" % (c["D"], c["P"], c["P"], c["C1"], c["C2"], 1)).encode())
        fh.write(f.cb1.tobytes())
        fh.write(f.cb2.tobytes())
    write_umem("base.umem", f.base, np.uint8)
    write_umem("query.umem", f.queries, np.uint8)
    args = ["--c1", str(c["C1"]), "--c2", str(c["C2"]), "--p", str(c["P"]), "--dim", str(c["D"]), "--lineparts", str(c["LP"]),
            "--basename", "t", "--w", str(c["W"])]
    out = subprocess.run([os.path.join(HOST, "tool_createdb")] + args + ["--dataset", "base.umem"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    f.oracle.save_bins("oracle.bins")
    assert open(pre + ".bins", "rb").read() == open("oracle.bins", "rb").read(), ".bins dump differs from the oracle's saveBins"
    # ground truth for recall = oracle's own top-1 (any fixed id list works: both sides are scored the same way)
    bv, bb, nvec = 2000, 500, 128
    lists = []
    f.oracle.set_sort_mode(1)
    for q in f.queries:
        ids, _ = f.oracle.query(q, bv, bb)
        lists.append(ids)
    f.oracle.set_sort_mode(0)
    rng = np.random.default_rng(0)
    gt = np.array([l[min(len(l) - 1, int(rng.integers(0, 40)))] for l in lists], np.int32).reshape(-1, 1)
    write_umem("gt.imem", gt, np.int32)
    out = subprocess.run([os.path.join(HOST, "tool_query")] + args + ["--queryset", "query.umem", "--groundtruth", "gt.imem",
                         "--boundvectors", str(bv), "--boundbins", str(bb), "--nvec", str(nvec)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    got = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"@R(\d+): ([0-9.eE+-]+)", out.stdout)}
    for R in (1, 10, 100):
        want = np.mean([gt[i, 0] in lists[i][:min(R, nvec)] for i in range(len(lists))])
        assert abs(got[R] - want) < 1e-5, (R, got[R], want)  # the tool prints 6 significant digits
    assert "avg. query time" in out.stdout


def test_tool_reports_missing_codebook(tmp_path):
    if not os.path.exists(os.path.join(HOST, "tool_query")):
        subprocess.check_call(["make", "-C", HOST])
    os.chdir(tmp_path)
    write_umem("q.umem", np.zeros((2, 128)), np.uint8)
    out = subprocess.run([os.path.join(HOST, "tool_query"), "--queryset", "q.umem", "--basename", "nope"], capture_output=True, text=True)
    assert out.returncode == 1 and "you need to generate a codebook first" in out.stdout


def test_createdb_trains_tree_identical_to_oracle(tmp_path):
    """tool_createdb --train: k-means by centroid splitting (GPU E step, host M step) == the oracle's restatement of
    productquantizer/vectorquantizer::generate, bit for bit (codebook dump compared as bytes)."""
    if not os.path.exists(os.path.join(HOST, "tool_createdb")):
        subprocess.check_call(["make", "-C", HOST])
    from common import sift_like
    from oracle import Oracle
    D, P, C1, C2, LP = 32, 2, 8, 4, 4
    data = sift_like(3000, D, 123)
    os.chdir(tmp_path)
    write_umem("base.umem", data, np.uint8)
    args = ["--c1", str(C1), "--c2", str(C2), "--p", str(P), "--dim", str(D), "--lineparts", str(LP), "--basename", "tr",
            "--dataset", "base.umem", "--train", "2000"]
    out = subprocess.run([os.path.join(HOST, "tool_createdb")] + args, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    raw = open("tr_%d_%d_%d_%d.ppqt" % (D, P, C1, C2), "rb").read()
    hdr = ("%d\n%d\n%d\n%d\n%d\n%d\n" % (D, P, P, C1, C2, 1)).encode()
    assert raw.startswith(hdr)
    body = np.frombuffer(raw[len(hdr):], np.float32)
    o = Oracle(D, P, C1, C2, 2, LP, heur_keep=16)
    o.train(data[:2000])
    cb1, cb2 = o.codebooks()
    assert np.array_equal(body[:C1 * D].view(np.uint32), cb1.ravel().view(np.uint32))
    assert np.array_equal(body[C1 * D:].view(np.uint32), cb2.ravel().view(np.uint32))
    # and the database built with that tree equals the oracle's
    o.insert(data)
    o.save_bins("oracle.bins")
    assert open("tr_%d_%d_%d_%d.bins" % (D, P, C1, C2), "rb").read() == open("oracle.bins", "rb").read()


def test_class_surface_loadtree_loadbins_query(tmp_path):
    """pqt::PerturbationProTree used like the reference's treequantizer (cpu_version/tools/query.cpp): loadTree,
    loadBins, query(boundVectors, boundBins, vec, out) per vector, queryKNN per batch, saveTree/saveBins round trip."""
    if not os.path.exists(os.path.join(HOST, "test_classes")):
        subprocess.check_call(["make", "-C", HOST])
    f = fixture("tools_default")
    c = f.cfg
    os.chdir(tmp_path)
    f.oracle.save_tree("o.tree")
    f.oracle.save_bins("o.bins")
    nq, bv, bb = 12, 1500, 400
    f.queries[:nq].astype(np.float32).tofile("q.raw")
    out = subprocess.run([os.path.join(HOST, "test_classes"), str(c["D"]), str(c["P"]), str(c["LP"]), str(c["W"]), "o.tree", "o.bins",
                          "q.raw", str(nq), str(bv), str(bb), "res.bin"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    assert out.stdout.split()[:3] == ["ok", str(c["C1"]), str(c["C2"])]
    assert open("res.bin.tree", "rb").read() == open("o.tree", "rb").read()
    assert open("res.bin.bins", "rb").read() == open("o.bins", "rb").read()
    raw = np.fromfile("res.bin", np.uint32)
    pos = 0
    f.oracle.set_sort_mode(1)
    try:
        tops = []
        for i in range(nq):
            n = int(raw[pos]); pos += 1
            pairs = raw[pos:pos + 2 * n].reshape(n, 2); pos += 2 * n
            ids, d = f.oracle.query(f.queries[i], bv, bb)
            assert n == len(ids)
            assert np.array_equal(pairs[:, 0], ids)
            assert np.array_equal(pairs[:, 1], d.view(np.uint32))
            tops.append((ids[:16], d[:16]))
    finally:
        f.oracle.set_sort_mode(0)
    ri = raw[pos:pos + nq * 16].reshape(nq, 16); pos += nq * 16
    rd = raw[pos:pos + nq * 16].reshape(nq, 16)
    for i in range(nq):
        kk = len(tops[i][0])
        assert np.array_equal(ri[i, :kk], tops[i][0]) and np.array_equal(rd[i, :kk], tops[i][1].view(np.uint32))


def test_committed_dump_pair_loads_and_reproduces_expected_lists(tmp_path):
    """tests/golden/dump_small.{tree,bins}: an index dump pair in the reference's on-disk formats (written by the oracle's
    saveTree/saveBins; a pair written by a real reference build can be dropped in at the same paths).  Loaded through the
    product's loadTree/loadBins (C++ class surface -> C-ABI -> HIP) it must reproduce the committed candidate lists -- no
    oracle call at test time."""
    if not os.path.exists(os.path.join(HOST, "test_classes")):
        subprocess.check_call(["make", "-C", HOST])
    g = os.path.join(ROOT, "tests", "golden")
    exp = np.load(os.path.join(g, "dump_small_expected.npz"))
    D, P, C1, C2, W, LP = (int(v) for v in exp["cfg"])
    bv, bb = (int(v) for v in exp["bv_bb"])
    nq = exp["queries"].shape[0]
    os.chdir(tmp_path)
    exp["queries"].astype(np.float32).tofile("q.raw")
    out = subprocess.run([os.path.join(HOST, "test_classes"), str(D), str(P), str(LP), str(W), os.path.join(g, "dump_small.tree"),
                          os.path.join(g, "dump_small.bins"), "q.raw", str(nq), str(bv), str(bb), "res.bin"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    assert out.stdout.split()[:3] == ["ok", str(C1), str(C2)]
    # the dumps survive a load/save round trip byte for byte
    assert open("res.bin.tree", "rb").read() == open(os.path.join(g, "dump_small.tree"), "rb").read()
    assert open("res.bin.bins", "rb").read() == open(os.path.join(g, "dump_small.bins"), "rb").read()
    raw = np.fromfile("res.bin", np.uint32)
    pos = off = 0
    for i in range(nq):
        n = int(raw[pos]); pos += 1
        pairs = raw[pos:pos + 2 * n].reshape(n, 2); pos += 2 * n
        assert n == int(exp["n_each"][i]), i
        assert np.array_equal(pairs[:, 1], exp["dist"][off:off + n].view(np.uint32)), i
        assert np.array_equal(pairs[:, 0], exp["ids"][off:off + n]), i
        off += n
