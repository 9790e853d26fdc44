#!/usr/bin/env python3
"""pair_rw_probe.py [N] -- do the allocation classes (profiles/r05/pair_matrix.txt)
also show when one array is READ and the other written (the table cores'
1R1W pattern)?  N arrays of 4 GiB; 0R2W and 1R1W (row read, column written)
over every pair, ms x 100."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_probes import hbm_probe

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << 30
arr = [torch.empty(n, dtype=torch.int32, device="cuda:0") for _ in range(N)]
for a in arr:
    a.zero_()
torch.cuda.synchronize()
P = [a.data_ptr() for a in arr]
print("# 0R2W ms x 100 (row i, column j > i)")
for i in range(N):
    print("w %2d %s" % (i, " ".join("   " if j <= i else "%3d" % round(
        100 * hbm_probe(None, None, P[i], P[j], n, 0, 2, 3, 2)) for j in range(N))))
print("# 1R1W ms x 100 (row read, column written)")
for i in range(N):
    print("r %2d %s" % (i, " ".join("   " if j == i else "%3d" % round(
        100 * hbm_probe(P[i], None, P[j], None, n, 1, 1, 3, 2)) for j in range(N))))
print("# 2R2W ms x 100 with arrays (0,1) read: written pair (row, column)")
for i in range(2, N):
    print("q %2d %s" % (i, " ".join("   " if j <= i else "%3d" % round(
        100 * hbm_probe(P[0], P[1], P[i], P[j], n, 2, 2, 3, 2)) for j in range(N))))
