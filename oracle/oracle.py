"""ctypes binding of oracle/libpqt_oracle.so (TEST INFRASTRUCTURE ONLY -- see pqt_oracle.cpp header).

`Oracle` mirrors the reference's `treequantizer<T,D,C1,C2,P,W,LP>` surface
(cpu_version/quantizer/treequantizer.hpp: generate/insert/query/saveTree/loadTree/saveBins/loadBins)
with run-time parameters.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)


def build_oracle(force=False):
    """Compile the restatement (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "libpqt_oracle.so")
    src = os.path.join(_HERE, "pqt_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libpqt_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/cpu_version"):
        subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return so


def _ptr(a, t):
    return a.ctypes.data_as(t)


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = build_oracle()
    L = C.CDLL(so)
    L.pqo_create.restype = C.c_void_p
    L.pqo_create.argtypes = [C.c_uint] * 6 + [C.c_ulonglong, C.c_int]
    L.pqo_destroy.argtypes = [C.c_void_p]
    L.pqo_set_sort_mode.argtypes = [C.c_void_p, C.c_int]
    L.pqo_set_sum_mode.argtypes = [C.c_void_p, C.c_int]
    L.pqo_max_multi_index.restype = C.c_ulonglong
    L.pqo_max_multi_index.argtypes = [C.c_void_p]
    L.pqo_set_max_multi_index.argtypes = [C.c_void_p, C.c_ulonglong]
    L.pqo_heuristic_rows.restype = C.c_ulonglong
    L.pqo_heuristic_rows.argtypes = [C.c_void_p]
    L.pqo_get_heuristic.argtypes = [C.c_void_p, u32p, C.c_ulonglong]
    L.pqo_set_heuristic.argtypes = [C.c_void_p, u32p, C.c_ulonglong]
    L.pqo_build_heuristic_2d.restype = C.c_int
    L.pqo_build_heuristic_2d.argtypes = [C.c_void_p, C.c_uint]
    L.pqo_get_heuristic_2d.argtypes = [C.c_void_p, u32p]
    L.pqo_rows_2d.argtypes = [C.c_void_p, f32p, C.c_uint, u32p]
    L.pqo_set_codebooks.argtypes = [C.c_void_p, f32p, f32p]
    L.pqo_get_codebooks.argtypes = [C.c_void_p, f32p, f32p]
    L.pqo_get_coarse.argtypes = [C.c_void_p, f32p]
    L.pqo_train.argtypes = [C.c_void_p, f32p, C.c_ulonglong]
    L.pqo_insert.argtypes = [C.c_void_p, f32p, C.c_ulonglong]
    L.pqo_bin_id.restype = C.c_uint
    L.pqo_bin_id.argtypes = [C.c_void_p, f32p]
    L.pqo_num_vectors.restype = C.c_ulonglong
    L.pqo_num_vectors.argtypes = [C.c_void_p]
    L.pqo_num_bins.restype = C.c_ulonglong
    L.pqo_num_bins.argtypes = [C.c_void_p]
    L.pqo_export_bins.argtypes = [C.c_void_p, u32p, u32p, u32p]
    L.pqo_import_bins.argtypes = [C.c_void_p, C.c_ulonglong, u32p, u32p, u32p]
    L.pqo_export_codes.argtypes = [C.c_void_p, u32p]
    L.pqo_import_codes.argtypes = [C.c_void_p, u32p, C.c_ulonglong]
    for n in ("pqo_save_tree", "pqo_load_tree", "pqo_save_bins", "pqo_load_bins"):
        getattr(L, n).restype = C.c_int
        getattr(L, n).argtypes = [C.c_void_p, C.c_char_p]
    L.pqo_stage_l1.argtypes = [C.c_void_p, f32p, f32p, f32p, u32p]
    L.pqo_stage_segments.argtypes = [C.c_void_p, f32p, u32p, u32p, f32p, f32p, u32p]
    L.pqo_stage_bins.restype = C.c_ulonglong
    L.pqo_stage_bins.argtypes = [C.c_void_p, f32p, C.c_uint, C.c_int, u32p, f32p, u32p]
    L.pqo_query.restype = C.c_ulonglong
    L.pqo_query.argtypes = [C.c_void_p, f32p, C.c_uint, C.c_uint, C.c_int, u32p, f32p, C.c_ulonglong]
    L.pqo_query_unsorted.restype = C.c_ulonglong
    L.pqo_query_unsorted.argtypes = [C.c_void_p, f32p, C.c_uint, C.c_uint, u32p, f32p, C.c_ulonglong]
    L.pqo_query_batch.argtypes = [C.c_void_p, f32p, C.c_ulonglong, C.c_uint, C.c_uint, C.c_uint, u32p, f32p, u32p, C.c_int]
    L.pqo_extract_distance.restype = C.c_float
    L.pqo_extract_distance.argtypes = [C.c_float] * 4
    L.pqo_calc_ratio.restype = C.c_float
    L.pqo_calc_ratio.argtypes = [C.c_float] * 3
    L.pqo_lambda_encode.restype = C.c_ushort
    L.pqo_lambda_encode.argtypes = [C.c_float]
    L.pqo_lambda_decode.restype = C.c_float
    L.pqo_lambda_decode.argtypes = [C.c_ushort]
    L.pqo_code_pack.restype = C.c_uint32
    L.pqo_code_pack.argtypes = [C.c_uint, C.c_uint, C.c_float]
    L.pqo_upow.restype = C.c_uint
    L.pqo_upow.argtypes = [C.c_uint, C.c_uint]
    L.pqo_max_threads.restype = C.c_int
    _LIB = L
    return L


def ref_helper():
    """The genuine cpu_version/helper.hpp functions (oracle/_ref), or None if not built."""
    so = os.path.join(_HERE, "_ref", "libref_helper.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.ref_extract_distance.restype = C.c_float
    L.ref_extract_distance.argtypes = [C.c_float] * 4
    L.ref_calc_ratio.restype = C.c_float
    L.ref_calc_ratio.argtypes = [C.c_float] * 3
    L.ref_code_pack.restype = C.c_uint32
    L.ref_code_pack.argtypes = [C.c_uint, C.c_uint, C.c_float]
    for n in ("ref_code_a", "ref_code_b"):
        getattr(L, n).restype = C.c_uint
        getattr(L, n).argtypes = [C.c_uint32]
    L.ref_code_lambda.restype = C.c_float
    L.ref_code_lambda.argtypes = [C.c_uint32]
    L.ref_to_ushort.restype = C.c_ushort
    L.ref_to_ushort.argtypes = [C.c_float]
    L.ref_upow.restype = C.c_uint
    L.ref_upow.argtypes = [C.c_uint, C.c_uint]
    L.ref_sizeof_code.restype = C.c_uint
    return L


def ref_format(cpu_version=False):
    """The genuine file-format layer of the reference (oracle/_ref): convert/filehelper.hpp + utils/filereader.hpp, or with
    cpu_version=True cpu_version/filehelper.hpp.  None if not built.  Every function returns 0, or 1 when the reference threw."""
    so = os.path.join(_HERE, "_ref", "libref_format_cpu.so" if cpu_version else "libref_format.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    up = C.POINTER(C.c_uint)
    for t in ("u8", "f32", "i32"):
        getattr(L, "reffmt_write_" + t).argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_void_p, C.c_uint, C.c_uint]
        getattr(L, "reffmt_read_" + t).argtypes = [C.c_char_p, up, up, C.c_void_p, C.c_uint, C.c_uint]
        getattr(L, "reffmt_jegou_header_" + t).argtypes = [C.c_char_p, up, up]
        getattr(L, "reffmt_jegou_" + t).argtypes = [C.c_char_p, C.c_void_p, up, up]
        if not cpu_version:
            getattr(L, "reffmt_filereader_" + t).argtypes = [C.c_char_p, C.c_void_p, up, up, C.c_size_t, C.c_size_t]
    L.reffmt_header.argtypes = [C.c_char_p, up, up]
    L.reffmt_jegou_batch_u8.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_void_p]
    return L


def ref_triangle():
    """The genuine pqt/triangle.cuh host functions (oracle/_ref), or None if not built/loadable."""
    so = os.path.join(_HERE, "_ref", "libref_triangle.so")
    if not os.path.exists(so):
        return None
    try:
        L = C.CDLL(so)
    except OSError:
        return None
    L.reftri_to_ushort.restype = C.c_ushort
    L.reftri_to_ushort.argtypes = [C.c_float]
    L.reftri_to_float.restype = C.c_float
    L.reftri_to_float.argtypes = [C.c_ushort]
    L.reftri_dist.restype = C.c_float
    L.reftri_dist.argtypes = [C.c_float] * 4
    L.reftri_project.restype = C.c_float
    L.reftri_project.argtypes = [C.c_float] * 3
    L.reftri_project_d2.restype = C.c_float
    L.reftri_project_d2.argtypes = [C.c_float] * 3 + [f32p]
    L.reftri_equal.restype = C.c_int
    L.reftri_equal.argtypes = [C.c_float] * 2
    return L


class Oracle:
    """Run-time-parameter restatement of treequantizer<float,D,C1,C2,P,W,LP>."""

    def __init__(self, D, P, C1, C2, W, LP, heur_keep=1 << 16, sort_mode=0):
        self.L = _lib()
        self.D, self.P, self.C1, self.C2, self.W, self.LP = D, P, C1, C2, W, LP
        self.S, self.SS = D // P, D // LP
        self.h = self.L.pqo_create(D, P, C1, C2, W, LP, heur_keep, sort_mode)
        if not self.h:
            raise ValueError("invalid PQT parameters")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pqo_destroy(self.h)
            self.h = None

    # ---- configuration -----------------------------------------------------------
    def set_sort_mode(self, m):
        self.L.pqo_set_sort_mode(self.h, m)

    def set_sum_mode(self, m):
        """Sensitivity probe: order of the squared-norm sums (0 = sequential, the parity definition; 1..5 see pqt_oracle.cpp)."""
        self.L.pqo_set_sum_mode(self.h, m)

    @property
    def max_multi_index(self):
        return self.L.pqo_max_multi_index(self.h)

    def lift_tuple_wrap(self, rows):
        """NOT reference behaviour (the checker following the engine's throughput-only "enumerate_beyond_wrap" mode): allow `rows`
        heuristic rows to be enumerated although (W*C2)^P wraps in the reference's uint arithmetic."""
        self.L.pqo_set_max_multi_index(self.h, int(rows))

    def heuristic(self, rows=None):
        n = self.L.pqo_heuristic_rows(self.h)
        rows = n if rows is None else min(rows, n)
        out = np.zeros((rows, self.P), np.uint32)
        self.L.pqo_get_heuristic(self.h, _ptr(out, u32p), rows)
        return out

    def set_heuristic(self, rows):
        rows = np.ascontiguousarray(rows, np.uint32)
        self.L.pqo_set_heuristic(self.h, _ptr(rows, u32p), rows.shape[0])

    def build_heuristic_2d(self, max_cluster):
        """Optional mode: the CUDA library's 2-D anisotropic sequences (ProTree.cu:50-126) and their per-query use
        (PerturbationProTree.cu:2839-3100); p = 4.  Any other heuristic call switches it off again."""
        if self.L.pqo_build_heuristic_2d(self.h, int(max_cluster)) != 0:
            raise ValueError("2-D sequences need p = 4 and 2 <= max_cluster <= 4096")

    def heuristic_2d(self):
        out = np.zeros((10, 65536), np.uint32)
        self.L.pqo_get_heuristic_2d(self.h, _ptr(out, u32p))
        return out

    def rows_2d(self, vec, rows):
        """The rows one query enumerates in the 2-D mode: [rows][P] part ranks, digit 0 = 0xffffffff where the cell names no bin."""
        v = np.ascontiguousarray(vec, np.float32).reshape(self.D)
        out = np.zeros((rows, self.P), np.uint32)
        self.L.pqo_rows_2d(self.h, _ptr(v, f32p), rows, _ptr(out, u32p))
        return out

    def set_codebooks(self, cb1, cb2):
        cb1 = np.ascontiguousarray(cb1, np.float32).reshape(self.C1, self.D)
        cb2 = np.ascontiguousarray(cb2, np.float32).reshape(self.P, self.C1, self.C2, self.S)
        self.L.pqo_set_codebooks(self.h, _ptr(cb1, f32p), _ptr(cb2, f32p))

    def codebooks(self):
        cb1 = np.zeros((self.C1, self.D), np.float32)
        cb2 = np.zeros((self.P, self.C1, self.C2, self.S), np.float32)
        self.L.pqo_get_codebooks(self.h, _ptr(cb1, f32p), _ptr(cb2, f32p))
        return cb1, cb2

    def coarse(self):
        out = np.zeros((self.LP, self.C1, self.C1), np.float32)
        self.L.pqo_get_coarse(self.h, _ptr(out, f32p))
        return out

    # ---- offline ------------------------------------------------------------------
    def train(self, data):
        data = np.ascontiguousarray(data, np.float32)
        self.L.pqo_train(self.h, _ptr(data, f32p), data.shape[0])

    def insert(self, vecs):
        vecs = np.ascontiguousarray(vecs, np.float32).reshape(-1, self.D)
        self.L.pqo_insert(self.h, _ptr(vecs, f32p), vecs.shape[0])

    def bin_id(self, vec):
        vec = np.ascontiguousarray(vec, np.float32)
        return self.L.pqo_bin_id(self.h, _ptr(vec, f32p))

    @property
    def num_vectors(self):
        return self.L.pqo_num_vectors(self.h)

    @property
    def num_bins(self):
        return self.L.pqo_num_bins(self.h)

    def export_bins(self):
        nb, nv = self.num_bins, self.num_vectors
        ids = np.zeros(nb, np.uint32)
        sizes = np.zeros(nb, np.uint32)
        members = np.zeros(nv, np.uint32)
        self.L.pqo_export_bins(self.h, _ptr(ids, u32p), _ptr(sizes, u32p), _ptr(members, u32p))
        return ids, sizes, members

    def import_bins(self, ids, sizes, members):
        ids = np.ascontiguousarray(ids, np.uint32)
        sizes = np.ascontiguousarray(sizes, np.uint32)
        members = np.ascontiguousarray(members, np.uint32)
        self.L.pqo_import_bins(self.h, ids.shape[0], _ptr(ids, u32p), _ptr(sizes, u32p), _ptr(members, u32p))

    def export_codes(self):
        out = np.zeros((self.num_vectors, self.LP), np.uint32)
        self.L.pqo_export_codes(self.h, _ptr(out, u32p))
        return out

    def import_codes(self, codes):
        codes = np.ascontiguousarray(codes, np.uint32).reshape(-1, self.LP)
        self.L.pqo_import_codes(self.h, _ptr(codes, u32p), codes.shape[0])

    def _io(self, fn, path):
        rc = getattr(self.L, fn)(self.h, os.fsencode(path))
        if rc != 0:
            raise RuntimeError("%s(%s) failed: %d" % (fn, path, rc))

    def save_tree(self, p):
        self._io("pqo_save_tree", p)

    def load_tree(self, p):
        self._io("pqo_load_tree", p)

    def save_bins(self, p):
        self._io("pqo_save_bins", p)

    def load_bins(self, p):
        self._io("pqo_load_bins", p)

    # ---- query stages ---------------------------------------------------------------
    def stage_l1(self, vec):
        vec = np.ascontiguousarray(vec, np.float32)
        virt = np.zeros((self.LP, self.C1), np.float32)
        l1 = np.zeros((self.P, self.C1), np.float32)
        order = np.zeros((self.P, self.C1), np.uint32)
        self.L.pqo_stage_l1(self.h, _ptr(vec, f32p), _ptr(virt, f32p), _ptr(l1, f32p), _ptr(order, u32p))
        return virt, l1, order

    def stage_segments(self, vec):
        vec = np.ascontiguousarray(vec, np.float32)
        n = (self.P, self.W * self.C2)
        l1 = np.zeros(n, np.uint32)
        l2 = np.zeros(n, np.uint32)
        d1 = np.zeros(n, np.float32)
        d2 = np.zeros(n, np.float32)
        order = np.zeros(n, np.uint32)
        self.L.pqo_stage_segments(self.h, _ptr(vec, f32p), _ptr(l1, u32p), _ptr(l2, u32p), _ptr(d1, f32p), _ptr(d2, f32p), _ptr(order, u32p))
        return l1, l2, d1, d2, order

    def stage_bins(self, vec, Bb, sort_bins=True):
        vec = np.ascontiguousarray(vec, np.float32)
        ids = np.zeros(Bb, np.uint32)
        dist = np.zeros(Bb, np.float32)
        seq = np.zeros(Bb, np.uint32)
        n = self.L.pqo_stage_bins(self.h, _ptr(vec, f32p), Bb, int(sort_bins), _ptr(ids, u32p), _ptr(dist, f32p), _ptr(seq, u32p))
        return ids[:n], dist[:n], seq[:n]

    def query(self, vec, Bv, Bb, cap=None, sort_bins=True):
        """treequantizer::query(boundVectors, boundBins, vec, out): full sorted (id, dist) list."""
        vec = np.ascontiguousarray(vec, np.float32)
        cap = cap or (Bv + self.num_vectors + 1)
        cap = int(min(cap, 1 << 26))
        ids = np.zeros(cap, np.uint32)
        dist = np.zeros(cap, np.float32)
        n = self.L.pqo_query(self.h, _ptr(vec, f32p), Bv, Bb, int(sort_bins), _ptr(ids, u32p), _ptr(dist, f32p), cap)
        n = min(n, cap)
        return ids[:n].copy(), dist[:n].copy()

    def query_unsorted(self, vec, Bv, Bb, cap=None):
        vec = np.ascontiguousarray(vec, np.float32)
        cap = int(min(cap or (Bv + self.num_vectors + 1), 1 << 26))
        ids = np.zeros(cap, np.uint32)
        dist = np.zeros(cap, np.float32)
        n = self.L.pqo_query_unsorted(self.h, _ptr(vec, f32p), Bv, Bb, _ptr(ids, u32p), _ptr(dist, f32p), cap)
        n = min(n, cap)
        return ids[:n].copy(), dist[:n].copy()

    def query_batch(self, Q, Bv, Bb, k, nthreads=0):
        Q = np.ascontiguousarray(Q, np.float32).reshape(-1, self.D)
        qn = Q.shape[0]
        ids = np.zeros((qn, k), np.uint32)
        dist = np.zeros((qn, k), np.float32)
        cnt = np.zeros(qn, np.uint32)
        self.L.pqo_query_batch(self.h, _ptr(Q, f32p), qn, Bv, Bb, k, _ptr(ids, u32p), _ptr(dist, f32p), _ptr(cnt, u32p), nthreads)
        return ids, dist, cnt

    def max_threads(self):
        return self.L.pqo_max_threads()
