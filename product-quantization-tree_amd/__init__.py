"""product-quantization-tree_amd -- MI355X (gfx950) Product-Quantization-Tree query engine.

The product is the C-ABI shared library ``csrc/libpqt_hip.so`` (declared in ``include/pqt_hip.h``) plus the
C++ host layer in ``host/`` (``pqt::ProQuantization`` / ``ProTree`` / ``PerturbationProTree`` shims and the
``tool_query`` / ``tool_createdb`` front-ends).  This Python module is only the ctypes glue that tests, bench.py
and __graft_entry__ use to drive the same C-ABI with torch-owned device buffers; it contains no compute and NO
fallback: if the HIP library is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("PQT_LIB") or os.path.join(CSRC, "libpqt_hip.so")  # PQT_LIB: tuning builds only
_LIB = None

u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)


class PqtError(RuntimeError):
    pass


class pqt_params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("dim", "p", "c1", "c2", "w", "lp")]


class pqt_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("queries", "candidates", "bins_visited", "bins_nonempty", "ties_l1",
                                          "ties_l2", "ties_bins", "ties_final")] + \
               [(n, C.c_float) for n in ("ms_tables", "ms_bins", "ms_rerank", "ms_select", "ms_total")] + \
               [("max_bin", C.c_uint32), ("filter_fallbacks", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/pqt_hip.h declares (checked by the CPU test-suite against the built library)
EXPORTS = [
    "pqt_last_error", "pqt_device_count", "pqt_index_create", "pqt_index_destroy", "pqt_index_params", "pqt_index_create_view",
    "pqt_index_set_option", "pqt_debug_tstamps", "pqt_kmeans_assign", "pqt_debug_calibrate_gather", "pqt_rerank_exact",
    "pqt_index_set_codebooks", "pqt_index_get_coarse", "pqt_index_build_heuristic", "pqt_index_build_heuristic_cuda", "pqt_index_build_heuristic_2d", "pqt_index_set_heuristic",
    "pqt_index_get_heuristic", "pqt_index_set_bins", "pqt_index_set_bins_shard", "pqt_index_set_bins_local", "pqt_index_set_db_hashed",
    "pqt_index_set_lines_host", "pqt_index_set_lines_dev", "pqt_build_assign_encode", "pqt_query", "pqt_query_host",
    "pqt_merge_topk", "pqt_compact_results", "pqt_index_device_bytes", "pqt_query_shard", "pqt_query_candidates", "pqt_index_device_arrays", "pqt_debug_stride", "pqt_debug_read", "pqt_get_stats",
    "pqt_get_rerank_launch_ms", "pqt_get_stage_ms_history", "pqt_dev_triangle", "pqt_get_last_path", "pqt_get_shared_rows_stats", "pqt_debug_stream_read", "pqt_debug_sort_scan", "pqt_traverse_bins", "pqt_query_shard_bins",
    "pqt_multi_last_error", "pqt_multi_create", "pqt_multi_destroy", "pqt_multi_shards", "pqt_multi_shard", "pqt_multi_shard_range", "pqt_multi_set_option",
    "pqt_multi_set_codebooks", "pqt_multi_build_heuristic", "pqt_multi_build_heuristic_cuda", "pqt_multi_build_heuristic_2d", "pqt_multi_set_heuristic", "pqt_multi_set_bins", "pqt_multi_set_lines_host", "pqt_multi_query",
    "pqt_multi_query_host", "pqt_multi_query_lane",
]


def build(force=False):
    """Compile csrc/libpqt_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("pqt_hip.hip", "pqt_rerank_launch.hip", "pqt_traverse_launch.hip", "pqt_fused_launch.hip", "pqt_shared_launch.hip", "pqt_shared_rows.h", "pqt_internal.h", "pqt_kernels.h", "pqt_device.h", "pqt_wave.h",
                                           "pqt_multi.cpp", "Makefile")] + \
           [os.path.join(_HERE, "..", "include", "pqt_hip.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-j6", "-C", CSRC, "libpqt_hip.so"])
    return LIB_PATH


def lib():
    """Load libpqt_hip.so; raises (never falls back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PqtError("libpqt_hip.so is not built (%s); run __graft_entry__.build() / make -C %s" % (LIB_PATH, CSRC))
    L = C.CDLL(LIB_PATH)
    L.pqt_last_error.restype = C.c_char_p
    L.pqt_device_count.restype = C.c_int
    L.pqt_index_create.argtypes = [C.POINTER(pqt_params), C.c_int, C.POINTER(C.c_void_p)]
    L.pqt_index_destroy.argtypes = [C.c_void_p]
    L.pqt_index_destroy.restype = None
    L.pqt_index_params.argtypes = [C.c_void_p, C.POINTER(pqt_params)]
    L.pqt_index_create_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.pqt_index_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.pqt_debug_tstamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.pqt_index_set_codebooks.argtypes = [C.c_void_p, f32p, f32p]
    L.pqt_index_get_coarse.argtypes = [C.c_void_p, f32p]
    L.pqt_index_build_heuristic.argtypes = [C.c_void_p, C.c_uint64]
    L.pqt_index_build_heuristic_cuda.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    L.pqt_index_build_heuristic_2d.argtypes = [C.c_void_p, C.c_uint32]
    L.pqt_index_set_heuristic.argtypes = [C.c_void_p, u32p, C.c_uint64]
    L.pqt_index_get_heuristic.argtypes = [C.c_void_p, u32p, C.c_uint64]
    L.pqt_index_set_bins.argtypes = [C.c_void_p, C.c_uint64, u32p, u32p, u32p]
    L.pqt_index_set_bins_shard.argtypes = [C.c_void_p, C.c_uint64, u32p, u32p, u32p, C.c_uint32, C.c_uint32]
    L.pqt_index_set_bins_local.argtypes = [C.c_void_p, C.c_uint64, u32p, u32p, u32p, u32p, u32p, C.c_uint64]
    L.pqt_index_set_db_hashed.argtypes = [C.c_void_p, C.c_uint32, u32p, u32p, u32p, C.c_uint32]
    L.pqt_index_set_lines_host.argtypes = [C.c_void_p, u32p, C.c_uint64, C.c_uint64]
    L.pqt_index_set_lines_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    L.pqt_build_assign_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pqt_query.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_query_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_traverse_bins.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_query_shard_bins.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_query_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_index_device_arrays.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pqt_query_host.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, f32p, u32p]
    L.pqt_merge_topk.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_compact_results.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_index_device_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.pqt_rerank_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64,
                                   C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_debug_stride.argtypes = [C.c_void_p]
    L.pqt_debug_stride.restype = C.c_uint64
    L.pqt_debug_read.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, u32p, u32p, f32p, u32p]
    L.pqt_get_stats.argtypes = [C.c_void_p, C.POINTER(pqt_stats)]
    L.pqt_get_rerank_launch_ms.argtypes = [C.c_void_p, f32p, C.c_int]
    L.pqt_get_stage_ms_history.argtypes = [C.c_void_p, f32p, C.c_int]
    L.pqt_get_last_path.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.pqt_get_shared_rows_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.pqt_debug_stream_read.argtypes = [C.c_int, C.c_uint64, C.c_int, f32p]
    L.pqt_debug_sort_scan.argtypes = [C.c_int, C.c_uint32, C.c_uint32, u32p]
    L.pqt_multi_last_error.restype = C.c_char_p
    L.pqt_multi_create.argtypes = [C.POINTER(pqt_params), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    L.pqt_multi_destroy.argtypes = [C.c_void_p]
    L.pqt_multi_destroy.restype = None
    L.pqt_multi_shards.argtypes = [C.c_void_p]
    L.pqt_multi_shard.argtypes = [C.c_void_p, C.c_int]
    L.pqt_multi_shard.restype = C.c_void_p
    L.pqt_multi_shard_range.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pqt_multi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.pqt_multi_set_codebooks.argtypes = [C.c_void_p, f32p, f32p]
    L.pqt_multi_build_heuristic.argtypes = [C.c_void_p, C.c_uint64]
    L.pqt_multi_build_heuristic_cuda.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    L.pqt_multi_build_heuristic_2d.argtypes = [C.c_void_p, C.c_uint32]
    L.pqt_multi_set_heuristic.argtypes = [C.c_void_p, u32p, C.c_uint64]
    L.pqt_multi_set_bins.argtypes = [C.c_void_p, C.c_uint64, u32p, u32p, u32p, C.c_uint64]
    L.pqt_multi_set_lines_host.argtypes = [C.c_void_p, u32p, C.c_uint64]
    L.pqt_multi_query.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_multi_query_lane.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pqt_multi_query_host.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, f32p, u32p]
    L.pqt_dev_triangle.argtypes = [f32p, f32p, f32p, f32p, C.c_uint32, f32p, f32p, C.POINTER(C.c_uint16), f32p, C.c_int]
    _LIB = L
    return L


def _chk(rc):
    if rc < 0:
        raise PqtError("pqt error %d: %s" % (rc, lib().pqt_last_error().decode()))
    return rc


def _np(a, dt):
    return np.ascontiguousarray(a, dt)


def _p(a, t):
    return a.ctypes.data_as(t)


class PqtIndex:
    """Handle on a device-resident PQT index (thin wrapper over the C-ABI; see include/pqt_hip.h)."""

    def __init__(self, D, P, C1, C2, W, LP, device=0):
        self.L = lib()
        self.D, self.P, self.C1, self.C2, self.W, self.LP = D, P, C1, C2, W, LP
        self.device = device
        self.h = C.c_void_p()
        prm = pqt_params(D, P, C1, C2, W, LP)
        _chk(self.L.pqt_index_create(C.byref(prm), device, C.byref(self.h)))
        self._keep = []

    def close(self):
        for v in getattr(self, "_views", []):  # the library destroys the views with their owner: only drop the handles
            v.h = C.c_void_p()
        self._views = []
        if getattr(self, "h", None) and self.h.value:
            self.L.pqt_index_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def view(self):
        """A second handle on the same loaded index (pqt_index_create_view): own scratch / stream / statistics, for a second batch
        in flight.  Query entry points only; dies with this object."""
        v = object.__new__(PqtIndex)
        v.L, v.device, v._keep, v._views = self.L, self.device, [], []
        v.D, v.P, v.C1, v.C2, v.W, v.LP = self.D, self.P, self.C1, self.C2, self.W, self.LP
        v.h = C.c_void_p()
        _chk(self.L.pqt_index_create_view(self.h, C.byref(v.h)))
        if not hasattr(self, "_views"):
            self._views = []
        self._views.append(v)
        return v

    def set_option(self, name, value):
        _chk(self.L.pqt_index_set_option(self.h, name.encode(), int(value)))

    # ---- population ---------------------------------------------------------------------------
    def set_codebooks(self, cb1, cb2):
        cb1 = _np(cb1, np.float32).reshape(self.C1, self.D)
        cb2 = _np(cb2, np.float32).reshape(self.P, self.C1, self.C2, self.D // self.P)
        _chk(self.L.pqt_index_set_codebooks(self.h, _p(cb1, f32p), _p(cb2, f32p)))

    def coarse(self):
        out = np.zeros((self.LP, self.C1, self.C1), np.float32)
        _chk(self.L.pqt_index_get_coarse(self.h, _p(out, f32p)))
        return out

    def build_heuristic(self, rows):
        _chk(self.L.pqt_index_build_heuristic(self.h, rows))

    def build_heuristic_cuda(self, max_cluster, rows):
        """The CUDA library's prepareDistSequence order (sum of sqrt(digit), digits < min(16, max_cluster)) as the table."""
        _chk(self.L.pqt_index_build_heuristic_cuda(self.h, max_cluster, rows))

    def build_heuristic_2d(self, max_cluster):
        """The CUDA library's 2-D anisotropic sequences (prepare2DDistSequence + the 1B path's pairwise merges): per-query rows, p = 4."""
        _chk(self.L.pqt_index_build_heuristic_2d(self.h, max_cluster))

    def set_heuristic(self, tuples):
        t = _np(tuples, np.uint32).reshape(-1, self.P)
        _chk(self.L.pqt_index_set_heuristic(self.h, _p(t, u32p), t.shape[0]))

    def heuristic(self, rows):
        out = np.zeros((rows, self.P), np.uint32)
        _chk(self.L.pqt_index_get_heuristic(self.h, _p(out, u32p), rows))
        return out

    def set_bins(self, ids, sizes, members):
        ids, sizes, members = _np(ids, np.uint32), _np(sizes, np.uint32), _np(members, np.uint32)
        _chk(self.L.pqt_index_set_bins(self.h, ids.shape[0], _p(ids, u32p), _p(sizes, u32p), _p(members, u32p)))

    def set_bins_shard(self, ids, sizes, members, id_lo, id_hi):
        ids, sizes, members = _np(ids, np.uint32), _np(sizes, np.uint32), _np(members, np.uint32)
        _chk(self.L.pqt_index_set_bins_shard(self.h, ids.shape[0], _p(ids, u32p), _p(sizes, u32p), _p(members, u32p),
                                             id_lo, id_hi))

    def set_bins_local(self, ids, gsizes, lower, lsizes, local_members, n_total):
        """Range shard described from the shard's side (see include/pqt_hip.h: pqt_index_set_bins_local)."""
        ids, gsizes, lower, lsizes, local_members = (_np(a, np.uint32) for a in (ids, gsizes, lower, lsizes, local_members))
        assert ids.shape == gsizes.shape == lower.shape == lsizes.shape
        _chk(self.L.pqt_index_set_bins_local(self.h, ids.shape[0], _p(ids, u32p), _p(gsizes, u32p), _p(lower, u32p), _p(lsizes, u32p),
                                             _p(local_members, u32p), int(n_total)))

    def set_db_hashed(self, prefix, counts, dbidx, hash_size):
        prefix, counts, dbidx = _np(prefix, np.uint32), _np(counts, np.uint32), _np(dbidx, np.uint32)
        _chk(self.L.pqt_index_set_db_hashed(self.h, dbidx.shape[0], _p(prefix, u32p), _p(counts, u32p), _p(dbidx, u32p),
                                            hash_size))

    def set_lines(self, codes, id_base=0):
        codes = _np(codes, np.uint32).reshape(-1, self.LP)
        _chk(self.L.pqt_index_set_lines_host(self.h, _p(codes, u32p), codes.shape[0], id_base))

    def set_lines_dev(self, codes_tensor, id_base=0):
        """Adopt a torch int32/uint32 CUDA tensor [n, LP] as the line store (kept alive by this object)."""
        assert codes_tensor.is_cuda and codes_tensor.is_contiguous() and codes_tensor.element_size() == 4
        self._keep.append(codes_tensor)
        _chk(self.L.pqt_index_set_lines_dev(self.h, codes_tensor.data_ptr(), codes_tensor.shape[0], id_base))

    # ---- device-pointer entry points (torch tensors own the memory) ---------------------------------------
    def query_dev(self, q, Bv, Bb, k, out_idx, out_dist, out_count=None, stream=None, sync=False):
        _chk(self.L.pqt_query(self.h, q.data_ptr(), q.shape[0], Bv, Bb, k, out_idx.data_ptr(), out_dist.data_ptr(),
                              out_count.data_ptr() if out_count is not None else None, stream, int(sync)))

    def query_candidates_dev(self, q, Bv, Bb, cap, out_idx, out_dist, out_count, stream=None, sync=False):
        """The reference's whole sorted candidate list per query (treequantizer::query output), cap entries per row."""
        _chk(self.L.pqt_query_candidates(self.h, q.data_ptr(), q.shape[0], Bv, Bb, cap, out_idx.data_ptr(), out_dist.data_ptr(),
                                         out_count.data_ptr(), stream, int(sync)))

    def query_shard_dev(self, q, Bv, Bb, k, out_idx, out_dist, out_pos, out_count=None, stream=None, sync=False):
        _chk(self.L.pqt_query_shard(self.h, q.data_ptr(), q.shape[0], Bv, Bb, k, out_idx.data_ptr(), out_dist.data_ptr(),
                                    out_pos.data_ptr(), out_count.data_ptr() if out_count is not None else None,
                                    stream, int(sync)))

    def traverse_bins_dev(self, q, Bv, Bb, cap, out_bins, stream=None, sync=False):
        """Query-sharded traversal: out_bins int64 [qn][cap + 1] (bin id | global start << 32 ..., trailer = count | nCand << 32)."""
        assert out_bins.element_size() == 8 and out_bins.is_contiguous() and out_bins.numel() >= q.shape[0] * (cap + 1)
        _chk(self.L.pqt_traverse_bins(self.h, q.data_ptr(), q.shape[0], Bv, Bb, cap, out_bins.data_ptr(), stream, int(sync)))

    def query_shard_bins_dev(self, q, Bv, Bb, k, bins, cap, out_idx, out_dist, out_pos, out_count=None, stream=None, sync=False):
        _chk(self.L.pqt_query_shard_bins(self.h, q.data_ptr(), q.shape[0], Bv, Bb, k, bins.data_ptr(), cap, out_idx.data_ptr(), out_dist.data_ptr(),
                                         out_pos.data_ptr(), out_count.data_ptr() if out_count is not None else None, stream, int(sync)))

    def merge_topk_dev(self, nshards, qn, k, idx_all, dist_all, pos_all, out_idx, out_dist, stream=None, sync=False, shard_stride=0):
        _chk(self.L.pqt_merge_topk(self.h, nshards, qn, k, idx_all.data_ptr(), dist_all.data_ptr(), pos_all.data_ptr(), shard_stride,
                                   out_idx.data_ptr(), out_dist.data_ptr(), stream, int(sync)))

    def rerank_exact_dev(self, q, k, in_idx, raw, out_idx, out_dist, raw_id_base=0, stream=None, sync=False):
        """Exact re-rank of in_idx[QN][k] against raw vectors (torch CUDA tensor, float32 or uint8 rows)."""
        import torch
        _chk(self.L.pqt_rerank_exact(self.h, q.data_ptr(), q.shape[0], k, in_idx.data_ptr(), raw.data_ptr(),
                                     int(raw.dtype == torch.uint8), raw_id_base, raw.shape[0], out_idx.data_ptr(),
                                     out_dist.data_ptr(), stream, int(sync)))

    def assign_encode_dev(self, vecs, out_bin, out_codes, stream=None):
        _chk(self.L.pqt_build_assign_encode(self.h, vecs.data_ptr(), vecs.shape[0], out_bin.data_ptr(),
                                            out_codes.data_ptr(), stream))

    # ---- host-pointer convenience -----------------------------------------------------------------------------
    def query(self, Q, Bv, Bb, k):
        Q = _np(Q, np.float32).reshape(-1, self.D)
        qn = Q.shape[0]
        idx = np.zeros((qn, k), np.uint32)
        dist = np.zeros((qn, k), np.float32)
        cnt = np.zeros(qn, np.uint32)
        _chk(self.L.pqt_query_host(self.h, _p(Q, f32p), qn, Bv, Bb, k, _p(idx, u32p), _p(dist, f32p), _p(cnt, u32p)))
        return idx, dist, cnt

    def stats(self):
        s = pqt_stats()
        _chk(self.L.pqt_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def shared_rows_stats(self):
        """What the shared-row pass read and wrote for the last batch (option "sr_stats" = 1 before the call; include/pqt_hip.h)."""
        out = (C.c_uint64 * 8)()
        _chk(self.L.pqt_get_shared_rows_stats(self.h, out))
        names = ("bins", "pairs", "distinct_rows", "rows_read", "items", "uncovered_queries", "distances_written", "capacity_flag")
        return dict(zip(names, [int(v) for v in out]))

    def debug_read_dist(self, qn):
        """(ncand [qn], cand_dist [qn][stride]) of the last call where it left distances in HBM (staged path; shared-row pass: filter
        distances d1, exact ones for the queries handed back)."""
        stride = self.L.pqt_debug_stride(self.h)
        nc = np.zeros(qn, np.uint32)
        cd = np.zeros((qn, stride), np.float32)
        _chk(self.L.pqt_debug_read(self.h, qn, None, None, None, None, _p(cd, f32p), _p(nc, u32p)))
        return nc, cd

    def device_bytes(self):
        """Device memory of this handle by purpose (include/pqt_hip.h: pqt_index_device_bytes)."""
        out = (C.c_uint64 * 8)()
        _chk(self.L.pqt_index_device_bytes(self.h, out))
        names = ("lines_id_order_owned", "lines_bin_order", "lines_group_major", "lines_xcode", "bias_and_ids", "bin_table_and_bitmap", "tree_and_heuristic", "scratch")
        d = dict(zip(names, [int(v) for v in out]))
        d["total"] = sum(d.values())
        return d

    def last_path(self):
        """Kernel variants of the last query call, e.g. 'traverse=fused-shape2 rerank=mode2-nw12-runs chunks=1'."""
        buf = C.create_string_buffer(256)
        _chk(self.L.pqt_get_last_path(self.h, buf, 256))
        return buf.value.decode()

    def stage_ms_history(self, cap=32):
        """[n][5] per-call device ms of {tables, traversal, gap between the fused kernels, rerank(+select), select}, oldest first."""
        out = np.zeros((cap, 5), np.float32)
        n = _chk(self.L.pqt_get_stage_ms_history(self.h, _p(out, f32p), cap))
        return out[:n].copy()

    def rerank_launch_ms(self, cap=64):
        out = np.zeros(cap, np.float32)
        n = _chk(self.L.pqt_get_rerank_launch_ms(self.h, _p(out, f32p), cap))
        return out[:n].copy()

    def debug_read(self, qn, cands=True, segs=True, dists=True):
        stride = self.L.pqt_debug_stride(self.h)
        WC = self.W * self.C2
        l1 = np.zeros((qn, self.LP, self.C1), np.float32)
        sd = np.zeros((qn, self.P, WC), np.float32) if segs else None
        sb = np.zeros((qn, self.P, WC), np.uint32) if segs else None
        nc = np.zeros(qn, np.uint32)
        ci = np.zeros((qn, stride), np.uint32) if cands else None
        cd = np.zeros((qn, stride), np.float32) if (cands and dists) else None
        _chk(self.L.pqt_debug_read(self.h, qn, _p(l1, f32p), _p(sd, f32p) if segs else None, _p(sb, u32p) if segs else None,
                                   _p(ci, u32p) if cands else None, _p(cd, f32p) if (cands and dists) else None, _p(nc, u32p)))
        return dict(l1virt=l1, seg_d2=sd, seg_bin=sb, ncand=nc, cand_idx=ci, cand_dist=cd, stride=stride)


class PqtMulti:
    """One handle over a range-sharded database: N shards on N devices of one process (include/pqt_hip.h: pqt_multi_*)."""

    def __init__(self, D, P, C1, C2, W, LP, devices):
        self.L = lib()
        self.D, self.P, self.C1, self.C2, self.W, self.LP = D, P, C1, C2, W, LP
        self.h = C.c_void_p()
        prm = pqt_params(D, P, C1, C2, W, LP)
        devs = (C.c_int * len(devices))(*devices)
        self._chk(self.L.pqt_multi_create(C.byref(prm), len(devices), devs, C.byref(self.h)))
        self.n = len(devices)

    def _chk(self, rc):
        if rc < 0:
            raise PqtError("pqt_multi error %d: %s" % (rc, self.L.pqt_multi_last_error().decode()))
        return rc

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.pqt_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        self._chk(self.L.pqt_multi_set_option(self.h, name.encode(), int(value)))

    def set_codebooks(self, cb1, cb2):
        cb1 = _np(cb1, np.float32).reshape(self.C1, self.D)
        cb2 = _np(cb2, np.float32).reshape(self.P, self.C1, self.C2, self.D // self.P)
        self._chk(self.L.pqt_multi_set_codebooks(self.h, _p(cb1, f32p), _p(cb2, f32p)))

    def build_heuristic(self, rows):
        self._chk(self.L.pqt_multi_build_heuristic(self.h, rows))

    def build_heuristic_cuda(self, max_cluster, rows):
        self._chk(self.L.pqt_multi_build_heuristic_cuda(self.h, max_cluster, rows))

    def build_heuristic_2d(self, max_cluster):
        self._chk(self.L.pqt_multi_build_heuristic_2d(self.h, max_cluster))

    def set_heuristic(self, tuples):
        t = _np(tuples, np.uint32).reshape(-1, self.P)
        self._chk(self.L.pqt_multi_set_heuristic(self.h, _p(t, u32p), t.shape[0]))

    def set_bins(self, ids, sizes, members, n_total=0):
        ids, sizes, members = _np(ids, np.uint32), _np(sizes, np.uint32), _np(members, np.uint32)
        self._chk(self.L.pqt_multi_set_bins(self.h, ids.shape[0], _p(ids, u32p), _p(sizes, u32p), _p(members, u32p), n_total))

    def set_lines(self, codes):
        codes = _np(codes, np.uint32).reshape(-1, self.LP)
        self._chk(self.L.pqt_multi_set_lines_host(self.h, _p(codes, u32p), codes.shape[0]))

    def shard_range(self, s):
        lo, hi = C.c_uint64(), C.c_uint64()
        self._chk(self.L.pqt_multi_shard_range(self.h, s, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def shard_last_path(self, s):
        buf = C.create_string_buffer(256)
        _chk(self.L.pqt_get_last_path(self.L.pqt_multi_shard(self.h, s), buf, 256))
        return buf.value.decode()

    def query_dev(self, q, Bv, Bb, k, out_idx, out_dist, out_count=None, stream=None, sync=False):
        self._chk(self.L.pqt_multi_query(self.h, q.data_ptr(), q.shape[0], Bv, Bb, k, out_idx.data_ptr(), out_dist.data_ptr(),
                                         out_count.data_ptr() if out_count is not None else None, stream, int(sync)))

    def query_lane_dev(self, lane, q, Bv, Bb, k, out_idx, out_dist, out_count=None, stream=None, sync=False):
        """pqt_multi_query_lane: lane 0 = the shards, lane 1 = views of them (two batches in flight behind one handle)."""
        self._chk(self.L.pqt_multi_query_lane(self.h, int(lane), q.data_ptr(), q.shape[0], Bv, Bb, k, out_idx.data_ptr(), out_dist.data_ptr(),
                                              out_count.data_ptr() if out_count is not None else None, stream, int(sync)))

    def query(self, Q, Bv, Bb, k):
        Q = _np(Q, np.float32).reshape(-1, self.D)
        qn = Q.shape[0]
        idx = np.zeros((qn, k), np.uint32)
        dist = np.zeros((qn, k), np.float32)
        cnt = np.zeros(qn, np.uint32)
        self._chk(self.L.pqt_multi_query_host(self.h, _p(Q, f32p), qn, Bv, Bb, k, _p(idx, u32p), _p(dist, f32p), _p(cnt, u32p)))
        return idx, dist, cnt


def stream_read_GBps(nbytes=4 << 30, reps=5, device=0):
    """GB/s of a read-only streaming kernel over nbytes of device memory (measured beside the nominal HBM peak)."""
    ms = C.c_float(0)
    _chk(lib().pqt_debug_stream_read(device, nbytes, reps, C.byref(ms)))
    return nbytes / (ms.value * 1e-3) / 1e9


def debug_sort_scan(mode, n, device=0):
    """The reference's sort / scan self-check patterns through the library's primitives (include/pqt_hip.h: pqt_debug_sort_scan)."""
    out = np.zeros(max(n, 512) + 1, np.uint32)
    _chk(lib().pqt_debug_sort_scan(device, mode, n, _p(out, u32p)))
    return out


def dev_triangle(a, b, c, l, device=0):
    """extractDistance / calcRatio / lambda codec evaluated by a kernel on the device."""
    a, b, c, l = (_np(x, np.float32).ravel() for x in (a, b, c, l))
    n = a.shape[0]
    d = np.zeros(n, np.float32)
    r = np.zeros(n, np.float32)
    u = np.zeros(n, np.uint16)
    f = np.zeros(n, np.float32)
    _chk(lib().pqt_dev_triangle(_p(a, f32p), _p(b, f32p), _p(c, f32p), _p(l, f32p), n, _p(d, f32p), _p(r, f32p),
                                u.ctypes.data_as(C.POINTER(C.c_uint16)), _p(f, f32p), device))
    return d, r, u, f
