"""Host-side helper: the first rows of the traversal heuristic WITHOUT enumerating the whole table.

prepareHeuristic (cpu_version/quantizer/treequantizer.hpp:75-127) sorts all (W*C2)^P tuples of {0..W*C2-1}^P by squared norm and
orderBins reads only the first boundBins rows.  At BASELINE configs[4] (P = 8, W*C2 = 64) the table would have 2^48 rows -- and the
reference's uint32 count of them wraps to 0, so it enumerates nothing there.  For the throughput-only leg of that shape
(option "enumerate_beyond_wrap", no reference counterpart) the first rows are produced best-first with a heap: keys are the
reference's (sum of squared digits), ties are broken by the tuple index (digit p has weight (W*C2)^p, like the reference's
decomposition `idx / base^p % base`), which is deterministic -- the reference's own tie order is an artefact of std::sort on the
full table and does not exist for a table that cannot be built.
"""
import heapq

import numpy as np


def heuristic_prefix_best_first(base, P, rows):
    """First `rows` tuples of {0..base-1}^P in (squared norm, tuple index) order, as uint32 [rows][P]."""
    rows = int(min(rows, base ** P))
    out = np.zeros((rows, P), np.uint32)
    start = (0,) * P
    heap = [(0, 0, start)]
    n = 0
    while heap and n < rows:
        norm, index, t = heapq.heappop(heap)
        out[n] = t
        n += 1
        # children: one more step in the last non-zero digit's position or any later one (every tuple has exactly one parent:
        # itself with its last non-zero digit decremented), so nothing is generated twice and the norm grows along every edge
        last = max([p for p in range(P) if t[p]], default=0)
        for p in range(last, P):
            if t[p] + 1 < base:
                c = t[:p] + (t[p] + 1,) + t[p + 1:]
                heapq.heappush(heap, (norm + 2 * t[p] + 1, index + base ** p, c))
    return out
