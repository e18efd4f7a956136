"""CPU tests: the oracle against every golden vector that pins it, file formats, and restatement invariants."""
import os
import struct

import numpy as np
import pytest

from common import fixture, Fixture
from oracle import Oracle, ref_helper, ref_triangle
from oracle.oracle import _lib

G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_run_cu_known_answers():
    """run.cu:33-104: project/dist known answers with the reference's tolerance equal(.,1e-5) (triangle.cuh:112)."""
    k = np.load(os.path.join(G, "run_cu_known.npz"))
    L = _lib()
    eps = float(k["eps"][0])
    for a2, b2, c2, lam, d2 in zip(k["a2"], k["b2"], k["c2"], k["lam"], k["d2"]):
        l = L.pqo_calc_ratio(float(a2), float(b2), float(c2))
        assert abs(l - lam) < eps
        assert abs(L.pqo_extract_distance(float(a2), float(b2), float(c2), l) - d2) < eps
    # lambda sweep of run.cu:106-113: codec round trip error < one quantisation step inside [-4,4), clamps outside
    for f in k["sweep"]:
        u = L.pqo_lambda_encode(float(f))
        r = L.pqo_lambda_decode(u)
        if -4 <= f < 4:
            assert 0 <= f - r < 8.0 / 65536 + 1e-6
        elif f >= 4:
            assert u == 65535
        else:
            assert u == 0


def test_oracle_line_math_matches_genuine_reference_functions():
    """Golden outputs of cpu_version/helper.hpp and pqt/triangle.cuh (tests/golden/make_golden.py) -- bit exact."""
    g = np.load(os.path.join(G, "ref_line_math.npz"))
    L = _lib()
    a, b, c, lam = g["a"], g["b"], g["c"], g["lam"]
    ed = np.array([L.pqo_extract_distance(float(x), float(y), float(z), float(l)) for x, y, z, l in zip(a, b, c, lam)], np.float32)
    assert np.array_equal(bits(ed), bits(g["extract_distance"]))
    assert np.array_equal(bits(ed), bits(g["tri_dist"]))  # CPU and CUDA headers agree on the formula
    cr = np.array([L.pqo_calc_ratio(float(x), float(y), float(z)) for x, y, z in zip(a, b, c)], np.float32)
    assert np.array_equal(bits(cr), bits(g["calc_ratio"]))
    assert np.array_equal(bits(cr), bits(g["tri_project"]))
    us = np.array([L.pqo_lambda_encode(float(l)) for l in lam], np.uint16)
    assert np.array_equal(us, g["to_ushort"]) and np.array_equal(us, g["to_ushort_triangle"])
    packed = np.array([L.pqo_code_pack(int(x), int(y), float(l)) for x, y, l in zip(g["pa"], g["pb"], lam)], np.uint32)
    assert np.array_equal(packed, g["packed"])
    assert np.array_equal(packed & 0xff, g["unpack_a"]) and np.array_equal((packed >> 8) & 0xff, g["unpack_b"])
    dec = np.array([L.pqo_lambda_decode(int(u)) for u in range(65536)], np.float32)
    assert np.array_equal(bits(dec), bits(g["decode_all_u16"]))
    assert np.array_equal(bits(dec[packed >> 16]), bits(g["unpack_lambda"]))
    for bi, base in enumerate(g["pow_bases"]):
        for e in range(9):
            assert L.pqo_upow(int(base), e) == int(g["pow_table"][bi, e])
    assert int(g["sizeof_code"][0]) == 4


def test_live_reference_functions_when_built():
    """When oracle/_ref is present (built from /root/reference), compare live on fresh random inputs."""
    H, T = ref_helper(), ref_triangle()
    if H is None or T is None:
        pytest.skip("oracle/_ref not built here")
    L = _lib()
    rng = np.random.default_rng(5)
    for _ in range(2000):
        a, b, c = (float(np.float32(rng.uniform(0, 1e5))) for _ in range(3))
        l = float(np.float32(rng.uniform(-6, 6)))
        assert bits(np.float32(L.pqo_extract_distance(a, b, c, l))) == bits(np.float32(H.ref_extract_distance(a, b, c, l)))
        assert bits(np.float32(L.pqo_calc_ratio(a, b, c + 1e-3))) == bits(np.float32(H.ref_calc_ratio(a, b, c + 1e-3)))
        assert L.pqo_lambda_encode(l) == H.ref_to_ushort(l) == T.reftri_to_ushort(l)
        assert bits(np.float32(T.reftri_dist(a, b, c, l))) == bits(np.float32(L.pqo_extract_distance(a, b, c, l)))


def test_oracle_regression_fixture():
    g = np.load(os.path.join(G, "oracle_small.npz"))
    D, P, C1, C2, W, LP = (int(x) for x in g["cfg"])
    o = Oracle(D, P, C1, C2, W, LP, heur_keep=64)
    o.set_codebooks(g["cb1"], g["cb2"])
    assert np.array_equal(o.heuristic(), g["heur"])
    assert np.array_equal(bits(o.coarse()), bits(g["coarse"]))
    o.insert(g["base"])
    ids, sizes, members = o.export_bins()
    assert np.array_equal(ids, g["bin_ids"]) and np.array_equal(sizes, g["bin_sizes"]) and np.array_equal(members, g["members"])
    assert np.array_equal(o.export_codes(), g["codes"])
    bv, bb = (int(x) for x in g["bv_bb"])
    off = 0
    for qi, q in enumerate(g["queries"]):
        i, d = o.query(q, bv, bb)
        n = int(g["n_each"][qi])
        assert len(i) == n
        assert np.array_equal(i, g["ids"][off:off + n]) and np.array_equal(bits(d), bits(g["dist"][off:off + n]))
        off += n


def test_heuristic_order_and_prefix():
    """prepareHeuristic (treequantizer.hpp:75-127): tuples sorted by squared norm; the comment's example prefix."""
    o = Oracle(8, 2, 4, 5, 2, 2, heur_keep=100)  # base 10, P=2 -> 100 tuples, like the (9,2) example in the source comment
    h = o.heuristic()
    assert h.shape == (100, 2)
    n = (h.astype(np.int64) ** 2).sum(1)
    assert np.all(np.diff(n) >= 0)
    assert sorted(map(tuple, h.tolist())) == sorted((a, b) for a in range(10) for b in range(10))
    assert h[0].tolist() == [0, 0] and h[-1].tolist() == [9, 9]
    assert o.max_multi_index == 100


def test_uint32_wraparound_of_bin_ids():
    """(C1*C2)^p wraps mod 2^32 exactly like pow<uint> (treequantizer.hpp:45-49, matlab/readme.md:26)."""
    f = fixture("wrap")
    L = _lib()
    assert L.pqo_upow(1024, 3) == (1024 ** 3) % 2 ** 32 and L.pqo_upow(1024, 4) == 0
    cfg = f.cfg
    powers = [(cfg["C1"] * cfg["C2"]) ** p % 2 ** 32 for p in range(cfg["P"])]
    q = f.base[7]
    virt, l1, order = f.oracle.stage_l1(q)
    S = cfg["D"] // cfg["P"]
    want = 0
    for p in range(cfg["P"]):
        c = int(order[p, 0])
        d = ((q[p * S:(p + 1) * S][None] - f.cb2[p, c]) ** 2).sum(1)
        want = (want + (c * cfg["C2"] + int(d.argmin())) * powers[p]) % 2 ** 32
    assert f.oracle.bin_id(q) == want


def test_tree_and_bins_file_formats(tmp_path):
    """.tree: 5 x u32 (D,C1,C2,P,W) + cb1 + cb2 (treequantizer.hpp:699-737); .bins: nbins, {id,n,members}, nvec, LP, codes (:745-774)."""
    f = fixture("odd")
    o = f.oracle
    tp, bp = str(tmp_path / "t.tree"), str(tmp_path / "t.bins")
    o.save_tree(tp)
    o.save_bins(bp)
    raw = open(tp, "rb").read()
    c = f.cfg
    assert struct.unpack("<5I", raw[:20]) == (c["D"], c["C1"], c["C2"], c["P"], c["W"])
    assert len(raw) == 20 + 4 * (c["C1"] * c["D"] + c["C1"] * c["C2"] * c["D"])
    assert np.array_equal(np.frombuffer(raw[20:20 + 4 * c["C1"] * c["D"]], np.float32), f.cb1.ravel())
    assert np.array_equal(np.frombuffer(raw[20 + 4 * c["C1"] * c["D"]:], np.float32), f.cb2.ravel())
    rawb = open(bp, "rb").read()
    nb = struct.unpack("<I", rawb[:4])[0]
    assert nb == o.num_bins
    assert len(rawb) == 4 + 8 * nb + 4 * o.num_vectors + 8 + 4 * o.num_vectors * c["LP"]
    o2 = Oracle(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], heur_keep=f.heur_rows)
    o2.load_tree(tp)
    o2.load_bins(bp)
    assert np.array_equal(o2.export_codes(), f.codes)
    for q in f.queries[:8]:
        i1, d1 = o.query(q, 300, 100)
        i2, d2 = o2.query(q, 300, 100)
        assert np.array_equal(i1, i2) and np.array_equal(bits(d1), bits(d2))
    o3 = Oracle(c["D"], c["P"], c["C1"] , c["C2"], c["W"], c["LP"] , heur_keep=4)
    with pytest.raises(RuntimeError):
        o3.load_tree(str(tmp_path / "missing.tree"))


def test_cut_rule_finish_the_bin_then_stop():
    """rerankVectors (treequantizer.hpp:450-477): whole bins are taken; stop after the bin in which count > Bv."""
    f = fixture("tools_default")
    o = f.oracle
    size_of = dict(zip(f.bin_ids.tolist(), f.bin_sizes.tolist()))
    for q in f.queries[:12]:
        for bv in (0, 1, 100, 1500):
            bin_ids, dist, seq = o.stage_bins(q, 500)
            n = 0
            for s in seq:
                n += size_of.get(int(bin_ids[s]), 0)
                if n > bv:
                    break
            ids, _ = o.query_unsorted(q, bv, 500)
            assert len(ids) == n


@pytest.mark.parametrize("name", ["tools_default", "cfg2_small", "wrap", "odd"])
def test_std_sort_and_stable_sort_agree_without_ties(name):
    """The reference's std::sort order is unique unless keys tie; fixtures are tie-free -> both oracle modes agree."""
    f = fixture(name)
    for q in f.queries[:8]:
        f.oracle.set_sort_mode(0)
        a = f.oracle.query(q, 500, min(500, len(f.heur)))
        f.oracle.set_sort_mode(1)
        b = f.oracle.query(q, 500, min(500, len(f.heur)))
        f.oracle.set_sort_mode(0)
        uniq = len(np.unique(a[1])) == len(a[1])
        assert np.array_equal(bits(a[1]), bits(b[1]))
        if uniq:
            assert np.array_equal(a[0], b[0])


def test_query_batch_matches_single_queries():
    f = fixture("odd")
    ids, dist, cnt = f.oracle.query_batch(f.queries, 200, 100, 16, nthreads=2)
    for qi, q in enumerate(f.queries):
        i, d = f.oracle.query(q, 200, 100)
        k = min(16, len(i))
        assert cnt[qi] == len(i)
        assert np.array_equal(ids[qi, :k], i[:k]) and np.array_equal(bits(dist[qi, :k]), bits(d[:k]))


def test_train_restatement_smoke():
    """generate() (treequantizer.hpp:155-177): k-means by splitting produces a usable tree (self-retrieval)."""
    rng = np.random.default_rng(3)
    data = np.rint(rng.uniform(0, 255, (600, 16))).astype(np.float32)
    o = Oracle(16, 2, 4, 2, 2, 4, heur_keep=16)
    o.train(data)
    cb1, cb2 = o.codebooks()
    assert np.isfinite(cb1).all() and np.isfinite(cb2).all()
    o.insert(data)
    hit = 0
    for i in range(20):
        ids, _ = o.query(data[i], 600, 16)
        hit += int(i in ids[:50])
    assert hit >= 15


def test_invalid_parameters_rejected():
    with pytest.raises(ValueError):
        Oracle(128, 3, 16, 8, 4, 32)  # D % P != 0 (static_assert treequantizer.hpp:23)
    with pytest.raises(ValueError):
        Oracle(128, 2, 4, 8, 5, 32)   # W > C1 (:25)


def test_converters_roundtrip(tmp_path):
    """fvecs / bvecs / ivecs -> .umem / .imem (convert/*.cpp of the reference): header text + payload at byte 20."""
    import subprocess
    host = os.path.join(os.path.dirname(G), "..", "product-quantization-tree_amd", "host")
    for t in ("convert_fvecs", "convert_bvecs", "convert_ivecs"):
        subprocess.check_call(["make", "-C", host, t], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(1)
    n, d = 37, 24

    def write_vecs(path, arr, dt):
        with open(path, "wb") as f:
            for r in arr:
                f.write(np.int32(d).tobytes())
                f.write(np.asarray(r, dt).tobytes())

    vals = rng.integers(0, 256, (n, d))
    write_vecs(tmp_path / "a.fvecs", vals, np.float32)
    write_vecs(tmp_path / "a.bvecs", vals, np.uint8)
    ivals = rng.integers(-5, 10 ** 6, (n, d))
    write_vecs(tmp_path / "a.ivecs", ivals, np.int32)
    for tool, flag, outflag, ext, dt, want in (("convert_fvecs", "fvecs", "umem", "fu", np.uint8, vals),
                                               ("convert_bvecs", "bvecs", "umem", "bu", np.uint8, vals),
                                               ("convert_ivecs", "ivecs", "imem", "ii", np.int32, ivals)):
        out = str(tmp_path / ("o." + ext))
        subprocess.check_call([os.path.join(host, tool), "--" + flag, str(tmp_path / ("a." + flag)), "--" + outflag, out,
                               "--chunkSize", "10"], stdout=subprocess.DEVNULL)
        raw = open(out, "rb").read()
        assert raw[:20].rstrip(b"\0") == ("%d\n%d\n" % (n, d)).encode()
        assert np.array_equal(np.frombuffer(raw[20:], dt).reshape(n, d), want.astype(dt))
    # a float file that is not uint8-valued must be refused, not silently truncated
    write_vecs(tmp_path / "b.fvecs", vals + 0.5, np.float32)
    r = subprocess.run([os.path.join(host, "convert_fvecs"), "--fvecs", str(tmp_path / "b.fvecs"), "--umem", str(tmp_path / "b.umem")],
                       capture_output=True)
    assert r.returncode != 0


def test_committed_dump_pair_reloads_into_the_oracle():
    """tests/golden/dump_small.{tree,bins} (reference on-disk formats) re-loaded by the oracle's loadTree/loadBins reproduce
    the committed candidate lists: guards the restatement (and the dump readers) against drift between machines/compilers."""
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = np.load(os.path.join(g, "dump_small_expected.npz"))
    D, P, C1, C2, W, LP = (int(v) for v in exp["cfg"])
    bv, bb = (int(v) for v in exp["bv_bb"])
    from oracle import Oracle
    o = Oracle(D, P, C1, C2, W, LP, heur_keep=bb)
    o.load_tree(os.path.join(g, "dump_small.tree"))
    o.load_bins(os.path.join(g, "dump_small.bins"))
    o.set_sort_mode(1)
    off = 0
    for i, q in enumerate(exp["queries"]):
        ids, d = o.query(q, bv, bb)
        n = int(exp["n_each"][i])
        assert len(ids) == n
        assert np.array_equal(ids, exp["ids"][off:off + n]) and np.array_equal(d.view(np.uint32), exp["dist"][off:off + n].view(np.uint32))
        off += n


def test_2d_anisotropic_sequences_tables_and_rows():
    """The checker's restatement of the CUDA library's 2-D anisotropic sequences (pqt/ProTree.cu:50-126 and
    pqt/PerturbationProTree.cu:2839-3100), checked against what the CUDA text says on its own terms: every order is the ascending
    (x^0.8 + s*y^0.8, cell) sequence for its slope (numpy restatement with the same f32 steps), a complete grid is a permutation
    followed by zeros, a query's rows are tuples of ranks inside the part lists (or "no bin"), the first row is the all-best tuple,
    and the slope classes equal roundf(log(slope) / log(1.2)) + 5 away from the class boundaries."""
    f = fixture("wrap")
    o = f.oracle
    try:
        for dc in (64, 512):
            o.build_heuristic_2d(dc)
            seq = o.heuristic_2d()
            n_vec = dc * dc
            keep = min(n_vec, 65536)
            i = np.arange(n_vec, dtype=np.int64)
            x, y = (i % dc).astype(np.float32), (i // dc).astype(np.float32)
            for slope in range(10):
                s = np.float32((0.9 * float(np.float32(1.2))) ** (slope - 5))
                key = (np.power(x, np.float32(0.8)) + s * np.power(y, np.float32(0.8))).astype(np.float32)
                got = seq[slope, :keep].astype(np.int64)
                assert len(np.unique(got)) == keep and got.max() < n_vec
                kg = key[got]
                # ascending in the key, ties in cell order (libm's powf and numpy's may differ in the last bit: compare the order through
                # the keys numpy computes, allowing equal-or-adjacent keys to swap)
                assert np.all(np.diff(kg.astype(np.float64)) >= -np.abs(kg[1:]).astype(np.float64) * 2.0 ** -20)
                if keep < 65536:
                    assert not seq[slope, keep:].any()
            WC = f.cfg["W"] * f.cfg["C2"]
            for q in f.queries[:6]:
                rows = o.rows_2d(q, 600)
                real = rows[rows[:, 0] != 0xffffffff]
                assert len(real) > 100 and real.max() < min(64, WC)
                assert tuple(rows[0]) == (0, 0, 0, 0)  # cell 0 of every order is (0, 0) on both levels
                assert len(np.unique(real, axis=0)) == len(real)  # a tuple is enumerated once
        # slope classes
        thr = np.array([np.float32(1.2) ** np.float32(j - 4.5) for j in range(9)], np.float32)
        rng = np.random.default_rng(5)
        sl = np.exp(rng.uniform(np.log(0.2), np.log(5.0), 4000)).astype(np.float32)
        t = np.log(sl.astype(np.float64)) / np.log(1.2)
        far = np.abs(t - np.floor(t) - 0.5) > 1e-3
        want = np.clip(np.rint(t) + 5, 0, 9).astype(np.int64)
        got = (sl[:, None] >= thr[None, :]).sum(1)
        assert np.array_equal(got[far], want[far])
    finally:
        o.set_heuristic(f.heur)


def test_format_layer_against_the_reference_compiled_fixture(tmp_path):
    """f2 pinned by the reference's OWN code: tests/golden/ref_formats.npz holds what convert/filehelper.hpp,
    utils/filereader.hpp and cpu_version/filehelper.hpp -- compiled from /root/reference as oracle/_ref/libref_format{,_cpu}.so by
    tests/golden/make_golden.py -- wrote and read for seeded arrays.  (1) The host layer's converters produce the reference writer's bytes
    for bvecs -> .umem and ivecs -> .imem.  (2) The host layer's reader (utils/filereader.hpp, the one tool_query / tool_createdb use) returns
    what the reference's FileReader<float | uint8_t | int> returned, whole files and (num, offset) windows.  (3) fvecs: the reference's
    converter stores the FLOAT payload (convert_fvecs.cpp:14,61) although its readers take one BYTE per value (the fixture's
    filereader_f32_of_float_payload is not the data); the host layer's converter follows the READER side: uint8 payload, = the reference
    writer's bytes for the uint8-cast values, and reads back as the values."""
    import subprocess
    z = np.load(os.path.join(G, "ref_formats.npz"))
    host = os.path.join(os.path.dirname(G), "..", "product-quantization-tree_amd", "host")
    for t in ("convert_fvecs", "convert_bvecs", "convert_ivecs", "read_mem"):
        subprocess.check_call(["make", "-C", host, t], stdout=subprocess.DEVNULL)
    # both copies of the reference's format code agree with each other
    for nm in ("umem_bytes", "imem_bytes", "fmem_bytes", "jegou_f32", "jegou_u8", "jegou_i32", "jegou_batch_u8_start9_num6", "read_u8_len640_off384"):
        assert np.array_equal(z[nm], z["cpu_" + nm]), nm
    # the reference's TEXMEX readers return the arrays the files were made from
    assert np.array_equal(z["jegou_f32"], z["f32"]) and np.array_equal(z["jegou_u8"], z["u8"]) and np.array_equal(z["jegou_i32"], z["i32"])
    assert np.array_equal(z["jegou_batch_u8_start9_num6"], z["u8"][9:15])
    assert tuple(z["jegou_header_u8"]) == z["u8"].shape and tuple(z["jegou_header_f32"]) == z["f32"].shape and tuple(z["jegou_header_i32"]) == z["i32"].shape
    # (1) converters against the reference writer's bytes
    for nm in ("fvecs", "bvecs", "ivecs"):
        (tmp_path / ("a." + nm)).write_bytes(z[nm + "_bytes"].tobytes())
    for tool, flag, outflag, want in (("convert_bvecs", "bvecs", "umem", z["umem_bytes"]), ("convert_ivecs", "ivecs", "imem", z["imem_bytes"])):
        out = tmp_path / ("o_" + flag + "." + outflag)
        subprocess.check_call([os.path.join(host, tool), "--" + flag, str(tmp_path / ("a." + flag)), "--" + outflag, str(out), "--chunkSize", "7"], stdout=subprocess.DEVNULL)
        assert np.array_equal(np.frombuffer(out.read_bytes(), np.uint8), want), tool
    # (2) the host reader on the REFERENCE-written files
    (tmp_path / "ref.umem").write_bytes(z["umem_bytes"].tobytes())
    (tmp_path / "ref.imem").write_bytes(z["imem_bytes"].tobytes())

    def read_mem(path, as_, dt, num=None, off=0):
        out = tmp_path / "raw.bin"
        cmd = [os.path.join(host, "read_mem"), "--in", str(path), "--as", as_, "--out", str(out)]
        if num is not None:
            cmd += ["--num", str(num), "--offset", str(off)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return tuple(int(x) for x in r.stdout.split()), np.frombuffer(out.read_bytes(), dt)

    nd, v = read_mem(tmp_path / "ref.umem", "f32", np.float32)
    assert nd == tuple(z["filereader_f32_of_umem_nd"]) and np.array_equal(v.reshape(nd), z["filereader_f32_of_umem"])
    _, v = read_mem(tmp_path / "ref.umem", "f32", np.float32, 4, 11)
    assert np.array_equal(v.reshape(4, -1), z["filereader_f32_of_umem_num4_off11"])
    _, v = read_mem(tmp_path / "ref.umem", "u8", np.uint8)
    assert np.array_equal(v.reshape(nd), z["filereader_u8_of_umem"])
    # read<uint8_t>(len, element offset) of the reference = the same window of the payload
    assert np.array_equal(z["read_u8_len640_off384"], z["u8"].reshape(-1)[384:384 + 640])
    nd, v = read_mem(tmp_path / "ref.imem", "i32", np.int32)
    assert nd == tuple(z["filereader_i32_of_imem_nd"]) and np.array_equal(v.reshape(nd), z["filereader_i32_of_imem"])
    _, v = read_mem(tmp_path / "ref.imem", "i32", np.int32, 5, 7)
    assert np.array_equal(v.reshape(5, -1), z["filereader_i32_of_imem_num5_off7"])
    assert np.array_equal(z["read_i32_len200_off700"], z["i32"].reshape(-1)[700:900])
    r = subprocess.run([os.path.join(host, "read_mem"), "--in", str(tmp_path / "missing.umem"), "--as", "f32", "--out", str(tmp_path / "x")], capture_output=True)
    assert r.returncode == 3  # std::runtime_error like the reference's reader (the fixture generator asserts the reference throws)
    # (3) fvecs
    assert not np.array_equal(z["filereader_f32_of_float_payload"], z["f32"])  # the reference's converter/reader pair does not round-trip
    out = tmp_path / "o_fvecs.umem"
    subprocess.check_call([os.path.join(host, "convert_fvecs"), "--fvecs", str(tmp_path / "a.fvecs"), "--umem", str(out)], stdout=subprocess.DEVNULL)
    raw = np.frombuffer(out.read_bytes(), np.uint8)
    hdr = ("%d\n%d\n" % z["f32"].shape).encode().ljust(20, b"\0")
    assert raw[:20].tobytes() == hdr and raw[:20].tobytes() == z["fmem_bytes"][:20].tobytes()  # the reference writer's header for this shape
    assert np.array_equal(raw[20:], z["f32"].astype(np.uint8).reshape(-1))
    nd, v = read_mem(out, "f32", np.float32)
    assert nd == z["f32"].shape and np.array_equal(v.reshape(nd), z["f32"])


def test_reference_format_libraries_still_match_the_fixture():
    """Where /root/reference is present (the authoring container) the format fixture is re-derived from the compiled reference and compared
    with the committed one: the fixture cannot drift from the code it claims to come from.  Skipped on the GPU box."""
    from oracle import ref_format
    if not os.path.isdir("/root/reference/convert") or ref_format() is None:
        pytest.skip("reference sources not present here")
    import ctypes as C
    z = np.load(os.path.join(G, "ref_formats.npz"))
    F = ref_format()
    import tempfile
    p = os.path.join(tempfile.mkdtemp(), "a.umem").encode()
    u8 = np.ascontiguousarray(z["u8"])
    assert F.reffmt_write_u8(p, u8.shape[0], u8.shape[1], u8.ctypes.data, u8.size, 0) == 0
    assert np.array_equal(np.frombuffer(open(p, "rb").read(), np.uint8), z["umem_bytes"])
    n_, d_ = C.c_uint(), C.c_uint()
    r = np.zeros(u8.shape, np.float32)
    assert F.reffmt_filereader_f32(p, r.ctypes.data, C.byref(n_), C.byref(d_), u8.shape[0], 0) == 0
    assert np.array_equal(r, z["filereader_f32_of_umem"])
