"""Round 6: parity tests of the OPT-IN kernels that were built while the GPU pool was closed to this repository:
pqt_k_sr_adc2, the range scan + merge of the selection, the cooperative filter scan (pqt_k_pair_scan), the compacted first level of the wide
traversal, and the hand-back / capacity cases of the shared-row pass.  None of them has run on a device yet; the file sorts LAST so that
`pytest -x` has already run every test of the default path when it gets here.  Bar as everywhere: ids, distance bits and counts identical
to the default kernels' (which the rest of the suite compares with the CPU checker)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from common import fixture, pqt_pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


@pytest.mark.parametrize("knobs", [(400, 500), (5000, 500), (10 ** 6, 512), (1, 500)])
def test_shared_row_evaluating_kernels_agree_bit_for_bit(knobs):
    """Round 6: pqt_k_sr_adc2 (tables of two queries interleaved as float2, one ds_read_b64 per look-up and pair; a row's 16-byte piece decoded
    once for all the queries of a chunk) writes the same filter distances as pqt_k_sr_adc at EVERY visiting position -- the whole cand_dist
    array is compared, not only the results that come out of the selection -- and the statistics launch of the pass adds up."""
    bv, bb = knobs
    f = fixture("cfg3_small")
    idx = f.hip_index()
    try:
        qn, k = f.queries.shape[0], 100
        idx.set_option("shared_rows", 1)
        idx.set_option("sr_stats", 1)
        out, dist = {}, {}
        for kern in (1, 2, 1):
            idx.set_option("sr_kernel", kern)
            out[kern] = idx.query(f.queries, bv, bb, k)
            assert "-shared" in idx.last_path(), idx.last_path()
            dist[kern] = idx.debug_read_dist(qn)
            st = idx.shared_rows_stats()
        a, b = out[1], out[2]
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
        # the selection's scan over position ranges of a query (2 or 4 per query, 4 or 8 requests in flight) + the merge of the range lists:
        # the same best list, hence the same everything
        fb0 = int(idx.stats()["filter_fallbacks"])
        for kern, split, depth in ((1, 2, 4), (2, 4, 8), (1, 1, 8), (2, 4, 4)):
            idx.set_option("sr_kernel", kern)
            idx.set_option("sr_scan_split", split)
            idx.set_option("sr_scan_depth", depth)
            c = idx.query(f.queries, bv, bb, k)
            assert "-shared" in idx.last_path(), idx.last_path()
            assert np.array_equal(a[0], c[0]) and np.array_equal(bits(a[1]), bits(c[1])) and np.array_equal(a[2], c[2]), (kern, split, depth)
            assert int(idx.stats()["filter_fallbacks"]) == fb0
        idx.set_option("sr_scan_split", 1)
        idx.set_option("sr_scan_depth", 4)
        (n1, d1), (n2, d2) = dist[1], dist[2]
        assert np.array_equal(n1, n2) and int(n1.sum()) > 0
        for q in range(qn):
            assert np.array_equal(bits(d1[q, :n1[q]]), bits(d2[q, :n2[q]])), q
        # the statistics of the last batch: nothing dropped, sums consistent
        assert st["capacity_flag"] == 0 and st["pairs"] >= st["bins"] > 0 and st["rows_read"] >= st["distinct_rows"] > 0 and st["items"] > 0
        assert st["rows_read"] <= st["pairs"] * int(idx.stats()["max_bin"])
        assert st["uncovered_queries"] <= qn and st["distances_written"] <= int(n1.sum())
        if st["uncovered_queries"] == 0:
            assert st["distances_written"] == int(n1.sum())
    finally:
        idx.close()


def test_shared_row_pass_hands_back_what_it_cannot_hold():
    """(a) a per-batch bin table that fills up (10-bit table, one probe per pair: every collision gives up), (b) queries with
    more than 64 runs (short bins, large vector bound: the traversal writes the plain list, nRuns = 0xffffffff) -- the pass does not cover
    them, they are handed back to the exact list kernels, and every id, distance bit and count is what the wave-per-query kernel returns.
    The statistics say how many queries were not covered; the capacity flag stays down (its caps are worst-case bounds)."""
    f = fixture("cfg3_small")
    idx = f.hip_index()
    try:
        qn, k = f.queries.shape[0], 100
        for bv, bb, probes, want_uncovered in ((400, 500, 1, True), (5000, 500, 1, True), (10 ** 6, 512, 128, None), (5000, 500, 128, None)):
            idx.set_option("shared_rows", 0)
            a = idx.query(f.queries, bv, bb, k)
            idx.set_option("shared_rows", 1)
            idx.set_option("sr_stats", 1)
            idx.set_option("sr_slot_bits", 10)
            idx.set_option("sr_probes", probes)
            for kern in (1, 2):
                idx.set_option("sr_kernel", kern)
                b = idx.query(f.queries, bv, bb, k)
                assert "-shared" in idx.last_path(), idx.last_path()
                st = idx.shared_rows_stats()
                fb = int(idx.stats()["filter_fallbacks"])
                assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2]), (bv, bb, probes, kern)
                assert st["capacity_flag"] == 0
                assert fb >= st["uncovered_queries"], (fb, st)
                if want_uncovered:
                    assert st["uncovered_queries"] > 0, st
            idx.set_option("sr_slot_bits", 0)
            idx.set_option("sr_probes", 128)
            idx.set_option("sr_kernel", 1)
        # (b) more than 64 runs per query: at least one knob set of this fixture must produce such queries
        idx.set_option("shared_rows", 1)
        b = idx.query(f.queries, 10 ** 6, 512, k)
        st = idx.shared_rows_stats()
        nb = idx.stats()
        assert st["uncovered_queries"] > 0 or nb["bins_nonempty"] <= 64 * qn, (st, nb)
    finally:
        idx.close()


@pytest.mark.parametrize("knobs", [(400, 500), (5000, 500), (10 ** 6, 512), (3000, 64), (1, 500), (130, 500)])
def test_cooperative_filter_scan_changes_no_bit(knobs):
    """Round 6: option "coop_rerank" -- two wavefronts per query around ONE LDS copy of its table take alternate
    batches of the candidates (pqt_k_pair_scan, 16 wavefronts per CU instead of 12), their two best lists are merged (pqt_k_sr_merge) and the
    band launch finishes the query.  Same ids, distance bits and counts as the wave-per-query filter kernel, unsharded, on range shards
    (merged), through a view; lists shorter than one batch (the second wavefront of a pair has nothing to do), plain candidate lists (more
    than 64 runs) and the tie-cluster hand-back included; pqt_get_stats reports no wavefront that gave up waiting for its partner."""
    import torch
    bv, bb = knobs
    f = fixture("cfg3_small")
    idx = f.hip_index()
    n = f.oracle.num_vectors
    shards = [f.hip_index(shard=(0, n // 3)), f.hip_index(shard=(n // 3, n))]
    try:
        k = 100
        a = idx.query(f.queries, bv, bb, k)
        fa = int(idx.stats()["filter_fallbacks"])
        assert "-coop" not in idx.last_path()
        idx.set_option("coop_rerank", 1)
        b = idx.query(f.queries, bv, bb, k)
        assert "rerank=mode2-nw12-runs-coop" in idx.last_path(), idx.last_path()
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
        assert int(idx.stats()["filter_fallbacks"]) == fa
        q = torch.from_numpy(f.queries).cuda()
        qn = q.shape[0]
        I = torch.empty((2, qn, k), dtype=torch.int32, device="cuda")
        Dd = torch.empty((2, qn, k), dtype=torch.float32, device="cuda")
        Pp = torch.empty((2, qn, k), dtype=torch.int32, device="cuda")
        Cc = torch.empty((2, qn), dtype=torch.int32, device="cuda")
        for s_, sh in enumerate(shards):
            sh.set_option("coop_rerank", 1)
            sh.query_shard_dev(q, bv, bb, k, I[s_], Dd[s_], Pp[s_], Cc[s_], sync=True)
            assert "-coop" in sh.last_path(), sh.last_path()
            sh.stats()
        oI = torch.empty((qn, k), dtype=torch.int32, device="cuda")
        oD = torch.empty((qn, k), dtype=torch.float32, device="cuda")
        shards[0].merge_topk_dev(2, qn, k, I, Dd, Pp, oI, oD, sync=True)
        assert np.array_equal(oI.cpu().numpy().view(np.uint32), a[0]) and np.array_equal(bits(oD.cpu().numpy()), bits(a[1]))
        v = idx.view()
        try:
            c = v.query(f.queries[::-1].copy(), bv, bb, k)
            assert "-coop" in v.last_path()
            assert np.array_equal(c[0], a[0][::-1]) and np.array_equal(bits(c[1]), bits(a[1][::-1]))
            v.stats()
        finally:
            v.close()
        # the pass has priority where both are asked for
        idx.set_option("shared_rows", 1)
        d_ = idx.query(f.queries, bv, bb, k)
        assert "-shared" in idx.last_path() and "-coop" not in idx.last_path()
        assert np.array_equal(a[0], d_[0]) and np.array_equal(bits(a[1]), bits(d_[1]))
    finally:
        idx.close()
        for sh in shards:
            sh.close()


def test_cooperative_filter_scan_keeps_the_tie_cluster_fallback():
    """the band overflow (hundreds of exactly tied candidates around the k-th distance) reaches the exact list kernels from the band launch
    behind the cooperative scan as well"""
    from common import Fixture

    def clustered(n, D, seed):
        protos = np.random.default_rng(777).integers(0, 256, (20, D)).astype(np.float32)
        rng = np.random.default_rng(seed)
        x = protos[rng.integers(0, 20, n)]
        noisy = rng.random(n) < 0.5
        x[noisy] = np.clip(np.rint(x[noisy] + rng.normal(0, 25, (int(noisy.sum()), D))), 0, 255)
        return x.astype(np.float32)

    f = Fixture(D=64, P=2, C1=64, C2=4, W=2, LP=32, n_base=12000, n_query=8, seed=68, heur_rows=64, train=3000, data=clustered)
    idx = f.hip_index()
    try:
        a = idx.query(f.queries, 10 ** 6, 64, 100)
        fa = int(idx.stats()["filter_fallbacks"])
        idx.set_option("coop_rerank", 1)
        b = idx.query(f.queries, 10 ** 6, 64, 100)
        assert "-coop" in idx.last_path(), idx.last_path()
        assert int(idx.stats()["filter_fallbacks"]) == fa and fa > 0
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
    finally:
        idx.close()


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_small"])
def test_wide_enumeration_with_the_compacted_first_level(name):
    """Option "filter_l1" = 2 (round 6): the rows that pass the LDS first level of the presence bitmap are compacted and the bitmap is asked
    for full wavefronts of them only (8-16 gather instructions per 4096-row query instead of 64); nothing may change."""
    f = fixture(name)
    idx = f.hip_index()
    try:
        for bv, bb in ((400, 4096), (3000, 1024), (10 ** 6, 2048)):
            if bb > f.heur.shape[0]:
                idx.build_heuristic(bb)
            a = idx.query(f.queries, bv, bb, 64)
            idx.set_option("filter_l1", 2)
            c = idx.query(f.queries, bv, bb, 64)
            pc = idx.last_path()
            idx.set_option("filter_l1", 0)
            assert "fused-wide" in pc and "-f1c" in pc, pc
            assert np.array_equal(a[0], c[0]) and np.array_equal(bits(a[1]), bits(c[1])) and np.array_equal(a[2], c[2]), (bv, bb)
    finally:
        idx.close()


def test_hbm_leg_prices_the_shared_row_pass_on_deduplicated_bytes():
    """when the shared-row pass runs, roofline.frac of the leg prices what one launch of pqt_k_sr_adc must move at the least
    (distinct rows + the distances it writes, counted on the device in the run), stays below 1, and SURVEY 8(d)'s per-candidate bytes ride
    along as a speed-up (algorithmic_equivalent_*), not as a fraction.  Small stand-in workload with the pass forced on."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu", "--hbm-workload", "synth1m", "--option", "shared_rows=1",
                          "--no-live-traffic"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    leg = d["config"]["hbm_roofline_leg"]
    assert "error" not in leg, leg
    for knobs in ("knobs_20000_500", "knobs_4096_4096"):
        e = leg[knobs]
        r = e["roofline"]
        assert "-shared" in e["kernel_path"] and r["kernel"] == "pqt_k_sr_adc", (e["kernel_path"], r["kernel"])
        dd = r["deduplicated"]
        assert dd["capacity_flag"] == 0 and 0 < dd["distinct_rows"] <= dd["rows_read_by_the_kernel"] and dd["distances_written"] > 0
        assert dd["bytes_per_launch"] == dd["distinct_rows"] * (4 * 32 + 4) + 4 * dd["distances_written"] == r["algorithmic_bytes_per_launch"]
        assert 0 < r["frac"] < 1 and abs(r["achieved"] - dd["bytes_per_launch"] / r["avg_launch_ms"] / 1e6) < 1e-6 * r["achieved"] + 1e-9
        assert dd["algorithmic_equivalent_speedup"] >= 1.0 and dd["survey_8d_bytes_per_launch"] >= dd["bytes_per_launch"]
        assert 0 < e["path_frac_of_hbm_peak"] < 1
        assert 0 < r["selection_kernel"]["frac"] < 1 and r["selection_kernel"]["bytes_per_launch"] > 0


@pytest.mark.parametrize("name,nsh", [("cfg2_small", 3), ("cfg3_small", 2)])
def test_multi_handle_two_batches_in_flight_on_two_lanes(name, nsh):
    """pqt_multi_query_lane -- lane 1 is a view of every shard with streams, events and exchange buffers of its own, so two
    batches are in flight behind ONE multi handle; issued on two streams without a host synchronisation in between, alternating lanes, every
    batch returns the single-index result bit for bit."""
    import torch
    pkg = pqt_pkg()
    f = fixture(name)
    c = f.cfg
    ref = f.hip_index()
    m = pkg.PqtMulti(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], [0] * nsh)
    try:
        m.set_codebooks(f.cb1, f.cb2)
        m.set_heuristic(f.heur)
        m.set_bins(f.bin_ids, f.bin_sizes, f.members)
        m.set_lines(f.codes)
        bv, bb, k = 400, 500, 50
        q = torch.from_numpy(f.queries).cuda()
        qrev = q.flip(0).contiguous()
        qn = q.shape[0]
        r_ids, r_d, r_c = ref.query(f.queries, bv, bb, k)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [[torch.empty((qn, k), dtype=torch.int32, device="cuda"), torch.empty((qn, k), dtype=torch.float32, device="cuda"), torch.empty(qn, dtype=torch.int32, device="cuda")]
                for _ in range(2)]
        torch.cuda.synchronize()
        for step in range(6):
            lane = step & 1
            qq = qrev if lane else q
            m.query_lane_dev(lane, qq, bv, bb, k, outs[lane][0], outs[lane][1], outs[lane][2], stream=streams[lane].cuda_stream)
        torch.cuda.synchronize()
        for lane in (0, 1):
            want = (r_ids[::-1], r_d[::-1], r_c[::-1]) if lane else (r_ids, r_d, r_c)
            assert np.array_equal(outs[lane][0].cpu().numpy().view(np.uint32), want[0]), lane
            assert np.array_equal(bits(outs[lane][1].cpu().numpy()), bits(want[1])), lane
            assert np.array_equal(outs[lane][2].cpu().numpy().view(np.uint32), want[2]), lane
        # the plain entry is lane 0 and still answers after the lanes were used
        m.query_dev(q, bv, bb, k, outs[0][0], outs[0][1], outs[0][2], sync=True)
        assert np.array_equal(outs[0][0].cpu().numpy().view(np.uint32), r_ids)
    finally:
        m.close()
        ref.close()


def test_frontend_two_shards_two_batches_in_flight():
    """the kept C++ front-end with setDevices (two shards on device 0): queryKNNAsync / queryKNNCollect now keep two batches in flight on the
    two lanes of the multi handle; every collected batch equals the single-index engine's padded arrays"""
    import importlib
    import torch
    fe_mod = importlib.import_module("product-quantization-tree_amd.frontend")
    f = fixture("cfg2_small")
    c = f.cfg
    nvec, bv, bb = 256, 3000, 512
    fe = fe_mod.FrontEnd(c["D"], c["P"], c["C1"], c["C2"], c["W"], c["LP"], f.cb1, f.cb2, f.bin_ids, f.bin_sizes, f.members, f.codes, devices=(0, 0))
    idx = f.hip_index()
    try:
        qa = torch.from_numpy(f.queries).cuda()
        qb = torch.from_numpy(np.ascontiguousarray(f.queries[::-1] * 0.5 + 20.0)).cuda()
        qn = qa.shape[0]
        idx.build_heuristic(bb)

        def engine(q):
            gi = torch.empty((qn, nvec), dtype=torch.int32, device="cuda"); gd = torch.empty((qn, nvec), dtype=torch.float32, device="cuda"); gc = torch.empty(qn, dtype=torch.int32, device="cuda")
            idx.query_dev(q, bv, bb, nvec, gi, gd, gc, sync=True)
            return gi.cpu().numpy().view(np.uint32), gd.cpu().numpy().view(np.uint32)
        ea, eb = engine(qa), engine(qb)
        for reps in (1, 2, 5):
            ms, oi, od = fe.queryKNN_inflight(qa.data_ptr(), qb.data_ptr(), qn, nvec, bv, bb, reps=reps, keep_padding=True)
            want = ea if reps % 2 == 1 else eb
            assert np.array_equal(oi, want[0]) and np.array_equal(od.view(np.uint32), want[1]), reps
        tm, oi, od = fe.queryKNN(qa.data_ptr(), qn, nvec, bv, bb, reps=2)
        assert np.array_equal(oi, ea[0]) and np.array_equal(od.view(np.uint32), ea[1])
    finally:
        fe.close()
        idx.close()


def test_100m_shape_pass_statistics_and_opt_in_kernels_at_full_size():
    """BASELINE configs[2] at full size (100 M vectors, chunk-built like bench.py's hbm leg; 2000 queries): the device statistics of the pass
    split `filter_fallbacks` into queries the pass did not cover and near-tie band overflows (the bound of the default-suite
    test had to be loosened to 40 without saying which kind rose), the capacity flag stays down, and the round's opt-in kernels -- pqt_k_sr_adc2,
    the range scan with the deeper queue, both together -- return the default kernels' ids, distance bits and counts at both knob sets."""
    import importlib
    import torch
    sys.path.insert(0, ROOT)
    import bench
    if torch.cuda.get_device_properties(0).total_memory < 64 << 30:
        pytest.skip("needs a 100 M-vector index in HBM")
    pkg = importlib.import_module("product-quantization-tree_amd")
    w = bench.WORKLOADS["synth100m"]
    idx, base, meta = bench.build_index(pkg, w, 0)
    try:
        idx.build_heuristic(4096)
        q = bench.sift_like(2000, w["D"], 0xC0DE03, torch.device("cuda", 0))
        qn, k = q.shape[0], 100

        def run():
            oi = torch.empty((qn, k), dtype=torch.int32, device=q.device); od = torch.empty((qn, k), dtype=torch.float32, device=q.device); oc = torch.empty(qn, dtype=torch.int32, device=q.device)
            idx.query_dev(q, bv, bb, k, oi, od, oc, sync=True)
            return oi.cpu().numpy(), od.cpu().numpy().view(np.uint32), oc.cpu().numpy()
        for bv, bb in ((20000, 500), (4096, 4096)):
            idx.set_option("sr_stats", 1)
            a = run()
            assert "-shared" in idx.last_path(), idx.last_path()
            st, fb = idx.shared_rows_stats(), int(idx.stats()["filter_fallbacks"])
            assert st["capacity_flag"] == 0 and st["distinct_rows"] > 0 and st["rows_read"] >= st["distinct_rows"]
            assert st["distances_written"] <= int(a[2].astype(np.int64).sum())
            # uncovered = more than 64 runs or a full table (none expected at this shape: a query's candidates sit in a few long bins);
            # the rest of the fall-backs are near-tie bands beyond the 256 slots (1-5 per 10 k-query batch measured in round 5)
            assert st["uncovered_queries"] <= 4 and fb >= st["uncovered_queries"] and fb - st["uncovered_queries"] <= 10, (st, fb)
            for opts in ({"sr_kernel": 2}, {"sr_scan_split": 4, "sr_scan_depth": 8}, {"sr_kernel": 2, "sr_scan_split": 2, "sr_scan_depth": 8}):
                for n_, v_ in opts.items():
                    idx.set_option(n_, v_)
                b = run()
                for n_, v_ in (("sr_kernel", 1), ("sr_scan_split", 1), ("sr_scan_depth", 4)):
                    idx.set_option(n_, v_)
                assert all(np.array_equal(a[j], b[j]) for j in range(3)), (bv, bb, opts)
                assert int(idx.stats()["filter_fallbacks"]) == fb
    finally:
        idx.close()
