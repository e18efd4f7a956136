"""ctypes glue for host/libpqt_frontend.so: the kept C++ front-end (pqt::PerturbationProTree::queryKNN, the call of the reference's
tool_query.cpp:153-161) driven from Python so that bench.py and the tests can time and check it.  No compute here."""
import ctypes as C
import os
import subprocess

import numpy as np

_HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HOST, "libpqt_frontend.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HOST, "libpqt_frontend.so"])
        L = C.CDLL(path)
        L.pqtfe_last_error.restype = C.c_char_p
        L.pqtfe_create.restype = C.c_void_p
        L.pqtfe_create.argtypes = [C.c_uint32] * 6 + [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.pqtfe_destroy.argtypes = [C.c_void_p]
        L.pqtfe_destroy.restype = None
        L.pqtfe_queryKNN.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pqtfe_queryKNN_inflight.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pqtfe_set_keep_padding.argtypes = [C.c_void_p, C.c_int]
        L.pqtfe_set_keep_padding.restype = None
        L.pqtfe_set_legacy_copy.argtypes = [C.c_void_p, C.c_int]
        L.pqtfe_set_legacy_copy.restype = None
        L.pqtfe_scribble.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.pqtfe_scribble.restype = None
        _LIB = L
    return _LIB


class FrontEnd:
    """A pqt::PerturbationProTree holding a tree + database handed over as host arrays."""

    def __init__(self, D, P, C1, C2, W, LP, cb1, cb2, bin_ids, bin_sizes, members, codes, devices=(0,)):
        L = lib()
        a = [np.ascontiguousarray(x, t) for x, t in ((cb1, np.float32), (cb2, np.float32), (bin_ids, np.uint32), (bin_sizes, np.uint32), (members, np.uint32), (codes, np.uint32))]
        devs = (C.c_int * len(devices))(*devices)
        self.h = L.pqtfe_create(D, P, C1, C2, W, LP, a[0].ctypes.data, a[1].ctypes.data, a[2].shape[0], a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data,
                                a[5].ctypes.data, a[5].size // LP, devs, len(devices))
        if not self.h:
            raise RuntimeError("pqtfe_create: " + L.pqtfe_last_error().decode())

    def close(self):
        if self.h:
            lib().pqtfe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def queryKNN(self, q_dev_ptr, qn, nvec, bv, bb, reps=1, want_results=True):
        """reps calls of queryKNN on the same two std::vectors; returns (timing dict of per-call means, idx [qn][nvec] u32, dist f32)."""
        t = np.zeros(7, np.float64)
        oi = np.empty((qn, nvec), np.uint32) if want_results else None
        od = np.empty((qn, nvec), np.float32) if want_results else None
        rc = lib().pqtfe_queryKNN(self.h, q_dev_ptr, qn, nvec, bv, bb, reps, t.ctypes.data, oi.ctypes.data if want_results else None,
                                  od.ctypes.data if want_results else None)
        if rc:
            raise RuntimeError("pqtfe_queryKNN: " + lib().pqtfe_last_error().decode())
        tm = dict(zip(("total_ms", "kernels_ms", "d2h_ms", "host_ms", "d2h_bytes", "columns", "packed"), t.tolist()))
        return tm, oi, od

    def queryKNN_inflight(self, q_dev_ptr, q_dev_ptr2, qn, nvec, bv, bb, reps=2, keep_padding=True, want_results=True):
        """reps batches through queryKNNAsync / queryKNNCollect with two in flight (batches alternate between the two query arrays);
        returns (wall ms per batch, idx, dist of the last batch)."""
        t = np.zeros(1, np.float64)
        oi = np.empty((qn, nvec), np.uint32) if want_results else None
        od = np.empty((qn, nvec), np.float32) if want_results else None
        rc = lib().pqtfe_queryKNN_inflight(self.h, q_dev_ptr, q_dev_ptr2, qn, nvec, bv, bb, reps, int(keep_padding), t.ctypes.data,
                                           oi.ctypes.data if want_results else None, od.ctypes.data if want_results else None)
        if rc:
            raise RuntimeError("pqtfe_queryKNN_inflight: " + lib().pqtfe_last_error().decode())
        return float(t[0]), oi, od

    def set_keep_padding(self, on):
        lib().pqtfe_set_keep_padding(self.h, int(on))

    def set_legacy_copy(self, on):
        lib().pqtfe_set_legacy_copy(self.h, int(on))

    def scribble(self, nvec, col, value):
        lib().pqtfe_scribble(self.h, nvec, col, value)
