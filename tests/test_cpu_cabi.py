"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol that
include/pqt_hip.h declares, and refuses to work without a gfx950 device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from common import ROOT, pqt_pkg


def header_functions():
    src = open(os.path.join(ROOT, "include", "pqt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pqt_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    pkg = pqt_pkg()
    pkg.build()
    L = ctypes.CDLL(pkg.LIB_PATH)
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libpqt_hip.so does not export %s" % n
    assert sorted(pkg.EXPORTS) == names, "python binding list and header disagree"


def test_struct_layouts_match_header():
    pkg = pqt_pkg()
    assert ctypes.sizeof(pkg.pqt_params) == 24
    assert ctypes.sizeof(pkg.pqt_stats) == 8 * 8 + 5 * 4 + 2 * 4 + 4  # 8-byte aligned tail padding
    

def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    pkg = pqt_pkg()
    assert pkg.lib().pqt_device_count() == 0
    with pytest.raises(pkg.PqtError):
        pkg.PqtIndex(128, 4, 32, 32, 2, 16)


def test_product_does_not_reference_the_oracle():
    """The product path must never import/link the checker."""
    pdir = os.path.join(ROOT, "product-quantization-tree_amd")
    for dp, _, files in os.walk(pdir):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "pqt_oracle" not in txt and "libpqt_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn
