import ctypes, sys
sys.path.insert(0,'/root/repo')
import importlib
pkg = importlib.import_module('product-quantization-tree_amd')
L = pkg.lib()
L.pqt_debug_calibrate_gather.argtypes=[ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_float)]
for lg in (20, 22, 24):
    for rb in (64, 128):
        best=1e9
        for rep in range(5):
            ms=ctypes.c_float()
            rc=L.pqt_debug_calibrate_gather(0, lg, rb, 1<<lg, ctypes.byref(ms))
            best=min(best, ms.value)
        print("rows 2^%d x %d B (%.0f MB): %.4f ms for all rows once -> %.2f G rows/s, %.2f TB/s" % (lg, rb, (1<<lg)*rb/1e6, best, (1<<lg)/best/1e6, (1<<lg)*rb/best/1e9))
