#!/usr/bin/env python3
"""many_arrays_probe.py [N] -- are there FAST arrays on a box whose first few
allocations are slow?  N (default 32) arrays of 4 GiB; per array the write-only
and the read-only sweep (the seeded kernel's work distribution, non-temporal);
the 0R2W pattern over every pair of the first 5 (what the placement sees at
first) and of the 8 fastest writers; the 1R2W pattern of the best pair with
every other array in the read role."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_probes import hbm_probe

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 1 << 30
dev = torch.device("cuda:0")
arr = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(N)]
for a in arr:
    a.zero_()
torch.cuda.synchronize()
P = [a.data_ptr() for a in arr]
GB = n * 4 / 1e9


def frac(ms, arrays):
    return arrays * GB / ms / 8.0      # GB per ms = TB/s, of 8 TB/s


w1 = [hbm_probe(None, None, p, None, n, 0, 1, 3, 3) for p in P]
r1 = [hbm_probe(p, None, None, None, n, 1, 0, 3, 3) for p in P]
print("# %d arrays of 4 GiB: address, write-only ms (of 8 TB/s), read-only ms" % N)
for i in range(N):
    print("  %2d %014x  %.3f (%.3f)  %.3f (%.3f)" % (i, P[i], w1[i], frac(w1[i], 1), r1[i], frac(r1[i], 1)))


def pairs(idx, label):
    res = []
    for i, j in itertools.combinations(idx, 2):
        res.append((hbm_probe(None, None, P[i], P[j], n, 0, 2, 3, 3), i, j))
    res.sort()
    print("# 0R2W over every pair of %s: best %.3f ms (%.3f) arrays %d,%d; median %.3f; worst %.3f"
          % (label, res[0][0], frac(res[0][0], 2), res[0][1], res[0][2],
             res[len(res) // 2][0], res[-1][0]))
    return res


if os.environ.get("MANY_MATRIX", "1") == "1":
    # the whole pair matrix (0R2W ms x 100), one launch pair per entry
    print("# 0R2W ms x 100 for every pair (row i, column j > i)")
    for i in range(N):
        row = []
        for j in range(N):
            if j <= i:
                row.append("   ")
            else:
                row.append("%3d" % round(100 * hbm_probe(None, None, P[i], P[j], n, 0, 2, 3, 2)))
        print("m %2d %s" % (i, " ".join(row)))
first = pairs(range(5), "the first 5 arrays")
order = sorted(range(N), key=lambda i: w1[i])
fast = pairs(order[:8], "the 8 fastest writers")
allp = pairs(range(min(N, 16)), "the first 16 arrays")
best = min(first + fast + allp)
i, j = best[1], best[2]
tri = sorted((hbm_probe(P[k], None, P[i], P[j], n, 1, 2, 3, 3), k) for k in range(N) if k not in (i, j))
print("# 1R2W with the best pair (%d,%d) written and each other array read: best %.3f ms (%.3f) array %d; "
      "median %.3f; worst %.3f" % (i, j, tri[0][0], frac(tri[0][0], 3), tri[0][1],
                                   tri[len(tri) // 2][0], tri[-1][0]))
fi, fj = first[0][1], first[0][2]
tri5 = sorted((hbm_probe(P[k], None, P[fi], P[fj], n, 1, 2, 3, 3), k) for k in range(5) if k not in (fi, fj))
print("# ... what the first 5 alone give: pair (%d,%d) %.3f ms, 1R2W best %.3f ms (%.3f)"
      % (fi, fj, first[0][0], tri5[0][0], frac(tri5[0][0], 3)))
