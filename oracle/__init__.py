"""oracle/ -- TEST INFRASTRUCTURE ONLY (CPU checker for the HIP path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (product-quantization-tree_amd/) never does.
"""
from .oracle import Oracle, build_oracle, ref_helper, ref_triangle, ref_format  # noqa: F401
