import ctypes, sys
sys.path.insert(0,'/root/repo')
import importlib
pkg = importlib.import_module('product-quantization-tree_amd')
L = pkg.lib()
L.pqt_debug_calibrate_gather.argtypes=[ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_float)]
for lg in (20, 22, 24):
    for rb in (64, 128, 64 + 0x1000, 128 + 0x1000):
        best=1e9
        for rep in range(5):
            ms=ctypes.c_float()
            rc=L.pqt_debug_calibrate_gather(0, lg, rb, 1<<lg, ctypes.byref(ms))
            best=min(best, ms.value)
        coop = bool(rb & 0x1000); rbb = rb & 0xfff
        print("rows 2^%d x %d B (%.0f MB)%s: %.4f ms for all rows once -> %.2f G rows/s, %.2f TB/s" % (lg, rbb, (1<<lg)*rbb/1e6, " one lane per 16-B piece" if coop else "", best, (1<<lg)/best/1e6, (1<<lg)*rbb/best/1e9))
