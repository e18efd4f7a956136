#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the authoring container only (needs /root/reference for oracle/_ref).

  ref_line_math.npz   outputs of the GENUINE reference functions (cpu_version/helper.hpp compiled as
                      oracle/_ref/libref_helper.so, pqt/triangle.cuh as oracle/_ref/libref_triangle.so) on seeded
                      inputs: extractDistance, calcRatio, code_t pack/unpack, toUShort, pow<uint>, dist, project.
                      These pin the oracle's (and through it the kernels') line-quantisation arithmetic.
  run_cu_known.npz    the six known-answer triples of run.cu:33-104 (data transcribed from the reference's test:
                      inputs a2,b2,c2 and expected lambda,d2) and the lambda sweep inputs of run.cu:106-113.
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

from oracle import ref_helper, ref_triangle  # noqa: E402


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
    print("golden fixtures written")


if __name__ == "__main__":
    main()
