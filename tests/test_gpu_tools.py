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
    # the same dumps range-sharded over three handles behind the one front-end (all on device 0 here; `--gpus 8` on a node):
    # identical recall lines
    out3 = subprocess.run([os.path.join(HOST, "tool_query")] + args + ["--queryset", "query.umem", "--groundtruth", "gt.imem",
                          "--boundvectors", str(bv), "--boundbins", str(bb), "--nvec", str(nvec), "--devices", "0,0,0"], capture_output=True, text=True)
    assert out3.returncode == 0, out3.stderr + out3.stdout
    assert "range-sharded over 3 devices" in out3.stdout
    assert [l for l in out3.stdout.splitlines() if l.startswith("@R")] == [l for l in out.stdout.splitlines() if l.startswith("@R")]


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
    assert out.stdout.splitlines()[-1].split()[:3] == ["ok", str(c["C1"]), str(c["C2"])]
    assert "multi ok 2" in out.stdout  # the same dumps behind one object over two range shards: identical queryKNN / query() results
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


def test_class_prepare2DDistSequence_single_and_two_shards(tmp_path):
    """The class method with the reference's signature (ProTree.hh:69; test/test1B.cpp:941 calls prepare2DDistSequence(512)) on a p = 4
    index: queryKNN under the 2-D anisotropic sequences equals the checker's restatement, and one object over two range shards returns the
    same bytes as the single-device object (test_classes compares them itself)."""
    if not os.path.exists(os.path.join(HOST, "test_classes")):
        subprocess.check_call(["make", "-C", HOST])
    f = fixture("cfg2_small")
    c = f.cfg
    os.chdir(tmp_path)
    f.oracle.save_tree("o.tree")
    f.oracle.save_bins("o.bins")
    nq, bv, bb = 10, 800, 500
    f.queries[:nq].astype(np.float32).tofile("q.raw")
    out = subprocess.run([os.path.join(HOST, "test_classes"), str(c["D"]), str(c["P"]), str(c["LP"]), str(c["W"]), "o.tree", "o.bins",
                          "q.raw", str(nq), str(bv), str(bb), "res.bin"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "2d ok" in out.stdout and "multi ok 2" in out.stdout
    raw = np.fromfile("res.bin.2d", np.uint32)
    ri, rd = raw[:nq * 16].reshape(nq, 16), raw[nq * 16:].reshape(nq, 16)
    f.oracle.build_heuristic_2d(512)
    f.oracle.set_sort_mode(1)
    try:
        for i in range(nq):
            ids, d = f.oracle.query(f.queries[i], bv, bb)
            kk = min(16, len(ids))
            assert kk > 0 and np.array_equal(ri[i, :kk], ids[:kk]) and np.array_equal(rd[i, :kk], d[:kk].view(np.uint32))
    finally:
        f.oracle.set_heuristic(f.heur)
        f.oracle.set_sort_mode(0)


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
    assert out.stdout.splitlines()[-1].split()[:3] == ["ok", str(C1), str(C2)]
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


def _hashed_from_bins(bin_ids, bin_sizes, members, hs):
    """Python restatement of the .bins -> (.prefix, .count, .dbIdx) translation: slot = bin id % hashsize, bins sharing a slot
    concatenated in ascending id order (the dense hashed CSR of the CUDA library, PerturbationProTree.hh:11-12)."""
    starts = np.concatenate([[0], np.cumsum(bin_sizes.astype(np.int64))])
    order = sorted(range(len(bin_ids)), key=lambda b: (int(bin_ids[b]) % hs, int(bin_ids[b])))
    prefix, counts, dbidx = np.zeros(hs, np.uint32), np.zeros(hs, np.uint32), []
    for b in order:
        slot = int(bin_ids[b]) % hs
        if counts[slot] == 0:
            prefix[slot] = len(dbidx)
        counts[slot] += bin_sizes[b]
        dbidx.extend(members[starts[b]:starts[b + 1]].tolist())
    return prefix, counts, np.array(dbidx, np.uint32)


def test_createdb_multi_chunk_and_both_dump_families(tmp_path):
    """tool_createdb over the WHOLE dataset in 3 chunks == 1 chunk == the oracle's saveBins, byte for byte; the CUDA library's
    dump family (.prefix/.count/.dbIdx/_<LP>.lines) is the documented translation of the exact bins, and tool_query answers
    identically from either family when the hash does not alias."""
    if not os.path.exists(os.path.join(HOST, "tool_query")):
        subprocess.check_call(["make", "-C", HOST])
    f = fixture("tools_default")
    c = f.cfg
    n = f.base.shape[0]
    hs = 20011  # prime > (C1*C2)^P = 16384 possible bin ids: no two bins share a slot
    outs = {}
    for tag, chunk in (("one", n), ("three", (n + 2) // 3)):
        d = tmp_path / tag
        d.mkdir()
        os.chdir(d)
        pre = "t_%d_%d_%d_%d" % (c["D"], c["P"], c["C1"], c["C2"])
        with open(pre + ".ppqt", "wb") as fh:
            fh.write(("%d\n%d\n%d\n%d\n%d\n%d\n" % (c["D"], c["P"], c["P"], c["C1"], c["C2"], 1)).encode())
            fh.write(f.cb1.tobytes())
            fh.write(f.cb2.tobytes())
        write_umem("base.umem", f.base, np.uint8)
        args = ["--c1", str(c["C1"]), "--c2", str(c["C2"]), "--p", str(c["P"]), "--dim", str(c["D"]), "--lineparts", str(c["LP"]),
                "--basename", "t", "--w", str(c["W"]), "--hashsize", str(hs)]
        out = subprocess.run([os.path.join(HOST, "tool_createdb")] + args + ["--dataset", "base.umem", "--chunksize", str(chunk)],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr + out.stdout
        assert ("chunks %d" % (1 if tag == "one" else 3)) in out.stdout
        outs[tag] = {ext: open(pre + ext, "rb").read() for ext in (".bins", ".prefix", ".count", ".dbIdx", "_%d.lines" % c["LP"])}
    assert outs["one"] == outs["three"], "a 3-chunk build differs from the 1-chunk build"
    f.oracle.save_bins(str(tmp_path / "oracle.bins"))
    assert outs["one"][".bins"] == open(tmp_path / "oracle.bins", "rb").read()
    prefix, counts, dbidx = _hashed_from_bins(f.bin_ids, f.bin_sizes, f.members, hs)
    assert outs["one"][".prefix"] == prefix.tobytes() and outs["one"][".count"] == counts.tobytes() and outs["one"][".dbIdx"] == dbidx.tobytes()
    assert outs["one"]["_%d.lines" % c["LP"]] == f.codes.tobytes()
    # tool_query from the hashed family == from the .bins dump
    write_umem("query.umem", f.queries, np.uint8)
    gt = np.array([[int(f.oracle.query(q, 2000, 500)[0][0])] for q in f.queries], np.int32)
    write_umem("gt.imem", gt, np.int32)
    res = {}
    for hashed in ("0", "1"):
        out = subprocess.run([os.path.join(HOST, "tool_query")] + args + ["--queryset", "query.umem", "--groundtruth", "gt.imem", "--boundvectors", "2000",
                             "--boundbins", "500", "--nvec", "64", "--hashed", hashed], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr + out.stdout
        assert ("read t_%d_%d_%d_%d.prefix" % (c["D"], c["P"], c["C1"], c["C2"]) in out.stdout) == (hashed == "1")
        res[hashed] = [l for l in out.stdout.splitlines() if l.startswith("@R")]
    assert res["0"] == res["1"] and res["0"][0] == "@R1: 1"


def test_class_surface_device_pointers_and_getters(tmp_path):
    """buildKBestDB with a DEVICE pointer (the reference's signature) == the oracle's insert; getDBIdx / getLine /
    getBinPrefix / getBinCounts hand out device arrays that agree with the dumps."""
    if not os.path.exists(os.path.join(HOST, "test_classes")):
        subprocess.check_call(["make", "-C", HOST])
    f = fixture("tools_default")
    c = f.cfg
    n, hs = f.base.shape[0], 20011
    os.chdir(tmp_path)
    f.oracle.save_tree("o.tree")
    f.oracle.save_bins("o.bins")
    f.queries[:4].astype(np.float32).tofile("q.raw")
    f.base.astype(np.float32).tofile("b.raw")
    out = subprocess.run([os.path.join(HOST, "test_classes"), str(c["D"]), str(c["P"]), str(c["LP"]), str(c["W"]), "o.tree", "o.bins",
                          "q.raw", "4", "1500", "400", "res.bin", "b.raw", str(n), str(hs)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    assert open("res.bin.devbuild.bins", "rb").read() == open("o.bins", "rb").read()
    prefix, counts, dbidx = _hashed_from_bins(f.bin_ids, f.bin_sizes, f.members, hs)
    assert open("res.bin.h.prefix", "rb").read() == prefix.tobytes() and open("res.bin.h.count", "rb").read() == counts.tobytes()
    assert open("res.bin.h.dbIdx", "rb").read() == dbidx.tobytes()
    g = np.fromfile("res.bin.getters", np.uint32)
    g_dbidx, g_codes = g[:n], g[n:n + n * c["LP"]].reshape(n, c["LP"])
    o2 = n + n * c["LP"] + 2 * hs
    g_prefix, g_counts = g[n + n * c["LP"]:n + n * c["LP"] + hs], g[n + n * c["LP"] + hs:o2]
    g_codes_bin = g[o2:o2 + n * c["LP"]].reshape(n, c["LP"])
    assert np.array_equal(g_dbidx, f.members)  # ids grouped by bin, bins in ascending id order (std::map order)
    assert np.array_equal(g_codes, f.codes)  # getLine(): row i = code of VECTOR i, like the reference's d_lineLambda and the .lines dump
    assert np.array_equal(g_codes_bin, f.codes[f.members])  # getLineBinOrder(): row i = code of getDBIdx()[i]
    assert np.array_equal(g_prefix, prefix) and np.array_equal(g_counts, counts)


def test_bench_dataset_dir_real_data_leg(tmp_path):
    """bench.py --dataset-dir: the reference's own pipeline on TEXMEX-format files (scripts/prepare_data.sh:3 fetches them; here a
    small fabricated set in the same formats): tool_createdb trains the tree and builds the database, the engine's recall@1/@10/@100
    (cpu_version/tools/query.cpp:29-82) equals the checker's on the same index, id lists identical."""
    import json
    import sys
    from common import sift_like
    rng = np.random.default_rng(7)
    base, learn = sift_like(20000, 128, 901), sift_like(3000, 128, 902)
    pick = rng.integers(0, base.shape[0], 64)
    queries = np.clip(np.rint(base[pick] + rng.normal(0, 5, (64, 128))), 0, 255).astype(np.float32)
    d2 = ((queries[:, None, :] - base[None, :, :]) ** 2).sum(-1)
    gt = np.argsort(d2, axis=1, kind="stable")[:, :100].astype(np.int32)

    def write_vecs(path, a, item):
        a = np.ascontiguousarray(a, item)
        rec = np.empty((a.shape[0], 4 + a.shape[1] * a.itemsize), np.uint8)
        rec[:, :4] = np.frombuffer(np.int32(a.shape[1]).tobytes(), np.uint8)
        rec[:, 4:] = a.view(np.uint8).reshape(a.shape[0], -1)
        rec.tofile(path)
    write_vecs(tmp_path / "sift_base.fvecs", base, np.float32)
    write_vecs(tmp_path / "sift_learn.fvecs", learn, np.float32)
    write_vecs(tmp_path / "sift_query.bvecs", queries, np.uint8)  # both element types are read
    write_vecs(tmp_path / "sift_groundtruth.ivecs", gt, np.int32)
    if not os.path.exists(os.path.join(HOST, "tool_createdb")):
        subprocess.check_call(["make", "-C", HOST])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dataset-dir", str(tmp_path), "--dataset-train", "3000", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert d["data"].startswith("real:") and d["value"] > 0
    assert c["id_lists_identical_frac"] == 1.0
    assert c["engine"] == c["checker_on_same_index"]
    assert c["engine"]["recall@100"] >= c["engine"]["recall@1"] and c["engine"]["recall@100"] > 0.5


@pytest.mark.parametrize("nvec", [16, 100, 4096])
def test_frontend_queryKNN_packed_handover_equals_engine_and_whole_array_copy(nvec):
    """pqt::PerturbationProTree::queryKNN (the call of tool_query.cpp:155): a large sparse result is packed on the device, only the filled
    prefixes cross PCIe and host threads write the padding; small or dense results are copied whole.  Either way the resized vectors must
    hold exactly what the engine's padded device arrays hold (ids, distance bits, 0xffffffff / +inf padding --
    PerturbationProTree.cu:8278-8281 pads too).  Runs in a child process per variant (the pack threshold is read once per process)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import importlib, os, sys, json
        import numpy as np, torch
        sys.path.insert(0, os.path.join(%r, "tests"))
        from common import fixture
        fe_mod = importlib.import_module("product-quantization-tree_amd.frontend")
        nvec = %d
        f = fixture("cfg2_small")
        c = f.cfg
        devs = tuple(int(x) for x in os.environ.get("PQT_TEST_FE_DEVICES", "0").split(","))
        fe = fe_mod.FrontEnd(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], f.cb1, f.cb2, f.bin_ids, f.bin_sizes, f.members, f.codes, devices=devs)
        q = torch.from_numpy(f.queries).cuda()
        qn, bv, bb = q.shape[0], 3000, 512
        tm, oi, od = fe.queryKNN(q.data_ptr(), qn, nvec, bv, bb, reps=2)
        idx = f.hip_index()
        idx.build_heuristic(bb)
        gi = torch.empty((qn, nvec), dtype=torch.int32, device="cuda"); gd = torch.empty((qn, nvec), dtype=torch.float32, device="cuda"); gc = torch.empty(qn, dtype=torch.int32, device="cuda")
        idx.query_dev(q, bv, bb, nvec, gi, gd, gc, sync=True)
        cnt = gc.cpu().numpy()
        ok = bool(np.array_equal(oi, gi.cpu().numpy().view(np.uint32)) and np.array_equal(od.view(np.uint32), gd.cpu().numpy().view(np.uint32)))
        filled = int(np.minimum(cnt, nvec).sum())
        pad_ok = all((oi[r, cnt[r]:] == 0xffffffff).all() and np.isinf(od[r, cnt[r]:]).all() for r in np.nonzero(cnt < nvec)[0][:8])
        print("RESULT " + json.dumps({"ok": ok, "pad_ok": pad_ok, "packed": tm["packed"], "bytes": tm["d2h_bytes"], "filled": filled, "qn": qn, "short": int((cnt < nvec).sum())}))
    """ % (ROOT, nvec))
    res = {}
    variants = [("packed", {"PQT_FRONTEND_PACK_MIN_BYTES": "0"}), ("whole", {"PQT_FRONTEND_LEGACY_COPY": "1"}), ("default", {})]
    if nvec == 4096:  # the same through the one-object multi-shard handle (two range shards on this device): its own stream, then the compaction
        variants.append(("packed_two_shards", {"PQT_FRONTEND_PACK_MIN_BYTES": "0", "PQT_TEST_FE_DEVICES": "0,0"}))
    for variant, env_extra in variants:
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        r = __import__("json").loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        assert r["ok"] and r["pad_ok"], (variant, r)
        res[variant] = r
    r = res["packed"]
    if nvec == 4096:
        assert r["short"] > 0 and r["packed"] == 1.0 and r["bytes"] == (r["qn"] + 1) * 4 + 2 * 4 * r["filled"], r  # only the filled prefixes crossed PCIe
    else:
        assert r["packed"] == 0.0  # every row is full: more than half of the padded size is data -> whole-array copy
    assert res["whole"]["packed"] == 0.0 and res["whole"]["bytes"] == 2 * 4 * res["whole"]["qn"] * nvec
    assert res["default"]["packed"] == 0.0  # 32 queries: below the 8 MB threshold
    if nvec == 4096:
        assert res["packed_two_shards"]["packed"] == 1.0


def test_frontend_two_batches_in_flight_and_padding_memory():
    """queryKNNAsync / queryKNNCollect (two batches in flight on the index and a view of it, the loop of host/tool_query.cpp)
    and the padding memory of the hand-over (only the slots the previous batch filled beyond the new one's prefix are re-padded when the
    caller hands the same vectors back, tool_query.cpp:149-154).  Two different batches alternate, so rows shrink and grow between
    consecutive hand-overs into the same vectors: every collected batch must equal the engine's padded arrays bit for bit.  A caller that
    writes into the padding between calls sees its bytes survive with the memory on and restored with setKeepPadding(false); a row whose
    sentinel slots were overwritten is padded whole; a failing call leaves the object usable."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import importlib, os, sys, json
        import numpy as np, torch
        sys.path.insert(0, os.path.join(%r, "tests"))
        from common import fixture
        fe_mod = importlib.import_module("product-quantization-tree_amd.frontend")
        nvec = 4096
        f = fixture("cfg2_small")
        c = f.cfg
        fe = fe_mod.FrontEnd(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], f.cb1, f.cb2, f.bin_ids, f.bin_sizes, f.members, f.codes, devices=(0,))
        qa = torch.from_numpy(f.queries).cuda()
        qb = torch.from_numpy(np.ascontiguousarray(f.queries[::-1] * 0.5 + 20.0)).cuda()
        qn, bv, bb = qa.shape[0], 3000, 512
        idx = f.hip_index()
        idx.build_heuristic(bb)
        def engine(q):
            gi = torch.empty((qn, nvec), dtype=torch.int32, device="cuda"); gd = torch.empty((qn, nvec), dtype=torch.float32, device="cuda"); gc = torch.empty(qn, dtype=torch.int32, device="cuda")
            idx.query_dev(q, bv, bb, nvec, gi, gd, gc, sync=True)
            return gi.cpu().numpy().view(np.uint32), gd.cpu().numpy().view(np.uint32), gc.cpu().numpy()
        ea, eb = engine(qa), engine(qb)
        res = {"rows_differ": int((ea[2] != eb[2]).sum())}
        for reps in (1, 2, 4, 5):
            ms, oi, od = fe.queryKNN_inflight(qa.data_ptr(), qb.data_ptr(), qn, nvec, bv, bb, reps=reps, keep_padding=True)
            want = ea if reps %% 2 == 1 else eb
            res["inflight_%%d" %% reps] = bool(np.array_equal(oi, want[0]) and np.array_equal(od.view(np.uint32), want[1]))
        # the synchronous call on top of the same slots, alternating batches into the same vectors
        ok = True
        for q, want in ((qa, ea), (qb, eb), (qa, ea), (qa, ea)):
            tm, oi, od = fe.queryKNN(q.data_ptr(), qn, nvec, bv, bb, reps=1)
            ok &= bool(np.array_equal(oi, want[0]) and np.array_equal(od.view(np.uint32), want[1]) and tm["packed"] == 1.0)
        res["sync_alternating"] = ok
        # a caller that scribbles INSIDE the padding (not on a sentinel slot): survives with the memory on ...
        short = ea[2] < nvec - 2
        fe.scribble(nvec, nvec - 2, 7)
        tm, oi, od = fe.queryKNN(qa.data_ptr(), qn, nvec, bv, bb, reps=1)
        res["scribble_survives"] = bool((oi[short, nvec - 2] == 7).all() and np.array_equal(np.delete(oi, nvec - 2, 1), np.delete(ea[0], nvec - 2, 1)))
        # ... a caller whose storage no longer shows the sentinels (last slot overwritten: what assign / fill / a reallocation at the same
        # address look like) gets the whole row padded again, memory on or not (ADVICE r05)
        fe.scribble(nvec, nvec - 1, 9)
        tm, oi, od = fe.queryKNN(qa.data_ptr(), qn, nvec, bv, bb, reps=1)
        res["lost_sentinel_repads_row"] = bool(np.array_equal(oi, ea[0]) and np.array_equal(od.view(np.uint32), ea[1]))
        fe.scribble(nvec, nvec - 2, 7)
        fe.set_keep_padding(False)
        tm, oi, od = fe.queryKNN(qa.data_ptr(), qn, nvec, bv, bb, reps=1)
        res["repaired"] = bool(np.array_equal(oi, ea[0]) and np.array_equal(od.view(np.uint32), ea[1]))
        # a call that fails inside pqt_query (more than 8192 heuristic rows: PQT_ERR_LIMIT) leaves the object usable: the next call, synchronous or two in
        # flight, answers as if nothing had happened (ADVICE r05: the slot stayed busy and the ticket counter ahead)
        failed = 0
        for _ in range(3):
            try:
                fe.queryKNN(qa.data_ptr(), qn, nvec, bv, 9000, reps=1)
            except RuntimeError:
                failed += 1
        res["bad_call_throws"] = failed == 3
        tm, oi, od = fe.queryKNN(qb.data_ptr(), qn, nvec, bv, bb, reps=1)
        res["good_after_bad"] = bool(np.array_equal(oi, eb[0]) and np.array_equal(od.view(np.uint32), eb[1]))
        ms, oi, od = fe.queryKNN_inflight(qa.data_ptr(), qb.data_ptr(), qn, nvec, bv, bb, reps=3, keep_padding=True)
        res["inflight_after_bad"] = bool(np.array_equal(oi, ea[0]) and np.array_equal(od.view(np.uint32), ea[1]))
        res["short_rows"] = int(short.sum())
        print("RESULT " + json.dumps(res))
    """ % (ROOT,))
    env = dict(os.environ, PQT_FRONTEND_PACK_MIN_BYTES="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = __import__("json").loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r["rows_differ"] > 0 and r["short_rows"] > 0, r
    assert all(v for k_, v in r.items() if k_ not in ("rows_differ", "short_rows")), r


def test_tool_query_async_loop_equals_sync_loop(tmp_path):
    """host/tool_query: the default loop (next batch issued before the current one is collected) prints the recall the one-batch-at-a-time loop
    (--sync 1, the reference's form) prints; several batches (4096 queries each) so that both slots and the short last batch are used."""
    os.chdir(tmp_path)
    from common import sift_like
    base = sift_like(30000, 128, 91)
    queries = sift_like(9000, 128, 92)
    write_umem("base.umem", base, np.uint8)
    write_umem("query.umem", queries, np.uint8)
    args = ["--basename", "db", "--dim", "128", "--p", "4", "--c1", "16", "--c2", "16", "--w", "2", "--lineparts", "16"]
    out = subprocess.run([os.path.join(HOST, "tool_createdb")] + args + ["--dataset", "base.umem", "--train", "4000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    gt = np.zeros((queries.shape[0], 1), np.int32)
    d2 = ((queries[:200, None, :] - base[None, :3000, :]) ** 2).sum(-1)  # (a partial ground truth is enough: both loops must print the same numbers)
    gt[:200, 0] = d2.argmin(1)
    write_umem("gt.imem", gt, np.int32)
    outs = []
    for sync in ("0", "1"):
        o = subprocess.run([os.path.join(HOST, "tool_query")] + args + ["--queryset", "query.umem", "--groundtruth", "gt.imem", "--boundvectors", "2000", "--boundbins", "500",
                                                                      "--nvec", "256", "--sync", sync], capture_output=True, text=True)
        assert o.returncode == 0, o.stdout + o.stderr
        outs.append([l for l in o.stdout.splitlines() if l.startswith("@R")])
    assert outs[0] == outs[1] and len(outs[0]) == 6, outs
