#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the authoring container only (needs /root/reference for oracle/_ref).

  ref_line_math.npz   outputs of the GENUINE reference functions (cpu_version/helper.hpp compiled as
                      oracle/_ref/libref_helper.so, pqt/triangle.cuh as oracle/_ref/libref_triangle.so) on seeded
                      inputs: extractDistance, calcRatio, code_t pack/unpack, toUShort, pow<uint>, dist, project.
                      These pin the oracle's (and through it the kernels') line-quantisation arithmetic.
  run_cu_known.npz    the six known-answer triples of run.cu:33-104 (data transcribed from the reference's test:
                      inputs a2,b2,c2 and expected lambda,d2) and the lambda sweep inputs of run.cu:106-113.
  ref_formats.npz     the file-format layer run through the GENUINE reference code (convert/filehelper.hpp, utils/filereader.hpp and
                      cpu_version/filehelper.hpp compiled as oracle/_ref/libref_format{,_cpu}.so): the bytes its writers produce for
                      .umem / .imem / .fmem, what its readers (read<T>, header, FileReader<float|uint8_t|int>) return for them and for
                      TEXMEX fvecs / bvecs / ivecs files (readJegou*, readBatchJegou).  Data only: arrays in, bytes / arrays out.
  oracle_small.npz    a small seeded index + the oracle's own query outputs (NOT reference-pinned: guards the
                      restatement against drift between machines/compilers).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_format, ref_helper, ref_triangle  # noqa: E402


def write_vecs_file(path, arr, dtype):
    """TEXMEX layout (corpus-texmex.irisa.fr): per vector int32 dim + dim values."""
    arr = np.ascontiguousarray(arr, dtype)
    with open(path, "wb") as f:
        for row in arr:
            f.write(np.int32(arr.shape[1]).tobytes())
            f.write(row.tobytes())


def make_ref_formats():
    import ctypes as C
    import tempfile
    F, Fc = ref_format(False), ref_format(True)
    assert F is not None and Fc is not None, "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(0xF11E)
    u8 = rng.integers(0, 256, (37, 128)).astype(np.uint8)     # 128-dimensional: the reference's bvecs readers hard-wire 132-byte records
    i32 = rng.integers(-5, 10 ** 6, (23, 100)).astype(np.int32)
    f32 = rng.integers(0, 256, (19, 24)).astype(np.float32)   # integer-valued floats like SIFT fvecs
    out = dict(u8=u8, i32=i32, f32=f32)
    d = tempfile.mkdtemp()
    n_, d_ = C.c_uint(), C.c_uint()

    def fb(path):
        return np.frombuffer(open(path, "rb").read(), np.uint8).copy()

    for tag, lib in (("", F), ("cpu_", Fc)):
        # writers (convert/filehelper.hpp:252-282): header text, padding to byte 20, payload
        pu, pi, pf = (os.path.join(d, tag + x).encode() for x in ("a.umem", "a.imem", "a.fmem"))
        assert lib.reffmt_write_u8(pu, u8.shape[0], u8.shape[1], u8.ctypes.data, u8.size, 0) == 0
        assert lib.reffmt_write_i32(pi, i32.shape[0], i32.shape[1], i32.ctypes.data, i32.size, 0) == 0
        assert lib.reffmt_write_f32(pf, f32.shape[0], f32.shape[1], f32.ctypes.data, f32.size, 0) == 0
        out[tag + "umem_bytes"], out[tag + "imem_bytes"], out[tag + "fmem_bytes"] = fb(pu), fb(pi), fb(pf)
        # readers of the same header: read<T> (len elements from element offset), header
        for p_, nm in ((pu, "umem"), (pi, "imem"), (pf, "fmem")):
            assert lib.reffmt_header(p_, C.byref(n_), C.byref(d_)) == 0
            out[tag + nm + "_header"] = np.array([n_.value, d_.value], np.uint32)
        r = np.zeros(5 * 128, np.uint8)
        assert lib.reffmt_read_u8(pu, C.byref(n_), C.byref(d_), r.ctypes.data, r.size, 3 * 128) == 0
        out[tag + "read_u8_len640_off384"] = r
        r = np.zeros(200, np.int32)
        assert lib.reffmt_read_i32(pi, C.byref(n_), C.byref(d_), r.ctypes.data, r.size, 700) == 0
        out[tag + "read_i32_len200_off700"] = r
        r = np.zeros(48, np.float32)
        assert lib.reffmt_read_f32(pf, C.byref(n_), C.byref(d_), r.ctypes.data, r.size, 24) == 0
        out[tag + "read_f32_len48_off24"] = r
        assert lib.reffmt_header(os.path.join(d, "missing.umem").encode(), C.byref(n_), C.byref(d_)) == 1  # throws std::runtime_error
        # TEXMEX readers
        pfv, pbv, piv = (os.path.join(d, tag + x) for x in ("a.fvecs", "a.bvecs", "a.ivecs"))
        write_vecs_file(pfv, f32, np.float32)
        write_vecs_file(pbv, u8, np.uint8)
        write_vecs_file(piv, i32, np.int32)
        if not tag:
            out["fvecs_bytes"], out["bvecs_bytes"], out["ivecs_bytes"] = fb(pfv), fb(pbv), fb(piv)
        for p_, t, arr in ((pfv, "f32", f32), (pbv, "u8", u8), (piv, "i32", i32)):
            assert getattr(lib, "reffmt_jegou_header_" + t)(p_.encode(), C.byref(n_), C.byref(d_)) == 0
            out[tag + "jegou_header_" + t] = np.array([n_.value, d_.value], np.uint32)
            r = np.zeros(arr.shape, arr.dtype)
            assert getattr(lib, "reffmt_jegou_" + t)(p_.encode(), r.ctypes.data, C.byref(n_), C.byref(d_)) == 0
            out[tag + "jegou_" + t] = r
        r = np.zeros((6, 128), np.uint8)
        assert lib.reffmt_jegou_batch_u8(pbv.encode(), 9, 6, 128, r.ctypes.data) == 0
        out[tag + "jegou_batch_u8_start9_num6"] = r
    # utils/filereader.hpp: FileReader<T> reads a UINT8 payload and widens it; FileReader<int> reads int32
    pu, pi, pf = (os.path.join(d, x).encode() for x in ("a.umem", "a.imem", "a.fmem"))
    r = np.zeros(u8.shape, np.float32)
    assert F.reffmt_filereader_f32(pu, r.ctypes.data, C.byref(n_), C.byref(d_), u8.shape[0], 0) == 0
    out["filereader_f32_of_umem"], out["filereader_f32_of_umem_nd"] = r, np.array([n_.value, d_.value], np.uint32)
    r = np.zeros((4, 128), np.float32)
    assert F.reffmt_filereader_f32(pu, r.ctypes.data, C.byref(n_), C.byref(d_), 4, 11) == 0
    out["filereader_f32_of_umem_num4_off11"] = r
    r = np.zeros(u8.shape, np.uint8)
    assert F.reffmt_filereader_u8(pu, r.ctypes.data, C.byref(n_), C.byref(d_), u8.shape[0], 0) == 0
    out["filereader_u8_of_umem"] = r
    r = np.zeros(i32.shape, np.int32)
    assert F.reffmt_filereader_i32(pi, r.ctypes.data, C.byref(n_), C.byref(d_), i32.shape[0], 0) == 0
    out["filereader_i32_of_imem"], out["filereader_i32_of_imem_nd"] = r, np.array([n_.value, d_.value], np.uint32)
    r = np.zeros((5, 100), np.int32)
    assert F.reffmt_filereader_i32(pi, r.ctypes.data, C.byref(n_), C.byref(d_), 5, 7) == 0
    out["filereader_i32_of_imem_num5_off7"] = r
    # the reference's convert_fvecs writes the FLOAT payload (convert_fvecs.cpp:14,61,64: out_t = float, write(...)) = a.fmem above;
    # its query-side reader FileReader<float> takes the first n*d BYTES of that payload for the values:
    r = np.zeros(f32.shape, np.float32)
    assert F.reffmt_filereader_f32(pf, r.ctypes.data, C.byref(n_), C.byref(d_), f32.shape[0], 0) == 0
    out["filereader_f32_of_float_payload"] = r
    assert F.reffmt_filereader_f32(os.path.join(d, "missing.umem").encode(), None, C.byref(n_), C.byref(d_), 0, 0) == 1
    np.savez_compressed(os.path.join(HERE, "ref_formats.npz"), **out)


def main():
    H, T = ref_helper(), ref_triangle()
    assert H is not None and T is not None, "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(0xBEEF)
    n = 4096
    # squared triangle sides like the engine sees them (non-negative, integer-ish and fractional), plus edge cases
    a = np.concatenate([rng.uniform(0, 3e5, n // 2), np.rint(rng.uniform(0, 1e4, n // 2))]).astype(np.float32)
    b = np.concatenate([rng.uniform(0, 3e5, n // 2), np.rint(rng.uniform(0, 1e4, n // 2))]).astype(np.float32)
    c = np.concatenate([rng.uniform(1e-3, 3e5, n // 2), np.rint(rng.uniform(1, 1e4, n // 2))]).astype(np.float32)
    lam = np.concatenate([rng.uniform(-5, 5, n - 16), np.array([-4, 4, -4.0001, 3.9999998, 0, -0.0, 1, -1, 4.5, -4.5, 3.99993, 1e-7, -1e-7, 2, 0.5, 7.9], np.float32)]).astype(np.float32)
    ed = np.array([H.ref_extract_distance(float(x), float(y), float(z), float(l)) for x, y, z, l in zip(a, b, c, lam)], np.float32)
    cr = np.array([H.ref_calc_ratio(float(x), float(y), float(z)) for x, y, z in zip(a, b, c)], np.float32)
    us = np.array([H.ref_to_ushort(float(l)) for l in lam], np.uint16)
    us_tri = np.array([T.reftri_to_ushort(float(l)) for l in lam], np.uint16)
    pa = rng.integers(0, 256, n).astype(np.uint32)
    pb = rng.integers(0, 256, n).astype(np.uint32)
    packed = np.array([H.ref_code_pack(int(x), int(y), float(l)) for x, y, l in zip(pa, pb, lam)], np.uint32)
    ua = np.array([H.ref_code_a(int(p)) for p in packed], np.uint32)
    ub = np.array([H.ref_code_b(int(p)) for p in packed], np.uint32)
    ul = np.array([H.ref_code_lambda(int(p)) for p in packed], np.float32)
    all_u16 = np.arange(65536, dtype=np.uint32)
    dec = np.array([T.reftri_to_float(int(u)) for u in all_u16], np.float32)
    td = np.array([T.reftri_dist(float(x), float(y), float(z), float(l)) for x, y, z, l in zip(a, b, c, lam)], np.float32)
    tp = np.array([T.reftri_project(float(x), float(y), float(z)) for x, y, z in zip(a, b, c)], np.float32)
    pw = np.array([[H.ref_upow(x, e) for e in range(9)] for x in (2, 3, 64, 128, 256, 1024, 4096, 65535)], np.uint32)
    np.savez_compressed(os.path.join(HERE, "ref_line_math.npz"), a=a, b=b, c=c, lam=lam, extract_distance=ed, calc_ratio=cr,
                        to_ushort=us, to_ushort_triangle=us_tri, pa=pa, pb=pb, packed=packed, unpack_a=ua, unpack_b=ub,
                        unpack_lambda=ul, decode_all_u16=dec, tri_dist=td, tri_project=tp,
                        pow_bases=np.array([2, 3, 64, 128, 256, 1024, 4096, 65535], np.uint32), pow_table=pw,
                        sizeof_code=np.array([H.ref_sizeof_code()], np.uint32))
    # run.cu known answers (inputs and expected values as written in the reference's test)
    np.savez(os.path.join(HERE, "run_cu_known.npz"),
             a2=np.array([1, 2, 2, 2, 2, 5], np.float32), b2=np.array([2, 2, 2, 5, 5, 2], np.float32),
             c2=np.array([1, 4, 2, 9, 1, 1], np.float32), lam=np.array([1, .5, .5, 0.666666666, 2, -1], np.float32),
             d2=np.array([1, 1, 1.5, 1, 1, 1], np.float32), eps=np.array([1e-5], np.float32),
             sweep=(np.arange(-100, 100) / np.float32(10.0)).astype(np.float32))
    # oracle self-regression fixture
    from common import Fixture
    f = Fixture(D=32, P=2, C1=8, C2=4, W=2, LP=4, n_base=1500, n_query=8, seed=77, heur_rows=64, train=800)
    outs = [f.oracle.query(q, 120, 64) for q in f.queries]
    n_each = np.array([len(o[0]) for o in outs], np.uint32)
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), cb1=f.cb1, cb2=f.cb2, base=f.base, queries=f.queries,
                        heur=f.heur, bin_ids=f.bin_ids, bin_sizes=f.bin_sizes, members=f.members, codes=f.codes,
                        n_each=n_each, ids=np.concatenate([o[0] for o in outs]), dist=np.concatenate([o[1] for o in outs]),
                        coarse=f.oracle.coarse(), cfg=np.array([32, 2, 8, 4, 2, 4], np.uint32), bv_bb=np.array([120, 64], np.uint32))
    make_ref_formats()
    print("golden fixtures written")


if __name__ == "__main__":
    main()
