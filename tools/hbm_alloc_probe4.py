#!/usr/bin/env python3
"""hbm_alloc_probe4.py -- 1R2W one-shot-tile copy over every (input; output
pair) choice out of N consecutively allocated 2^30-word arrays: which relative
placements are fast?  Prints the fraction of 8 TB/s per triple, grouped by the
index distances (consecutive allocations are 4 GiB + 2 MiB apart)."""
import collections
import ctypes
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libhbmprobe.so"))
lib.hbm_probe.restype = ctypes.c_float
lib.hbm_probe.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
n = 1 << 30
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
arr = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(N)]
for t in arr:
    t.zero_()
torch.cuda.synchronize()
P = [t.data_ptr() for t in arr]
print("# arrays at " + " ".join("%x" % p for p in P))
res = {}
for i in range(N):
    for a, b in itertools.combinations([k for k in range(N) if k != i], 2):
        ms = lib.hbm_probe(P[i], None, P[a], P[b], n, 1, 2, 0, 8, None)
        res[(i, a, b)] = 12.0 * n / (ms * 1e-3) / 8e12
by = collections.defaultdict(list)
for (i, a, b), f in res.items():
    by[(a - i, b - i)].append(f)
print("# (out0 - in, out1 - in) in allocation steps: min .. max fraction (count)")
for k in sorted(by):
    v = by[k]
    print("%-10s %.3f .. %.3f  (%d)" % (k, min(v), max(v), len(v)))
print("# per input array: best and worst pair")
for i in range(N):
    v = sorted(((f, a, b) for (ii, a, b), f in res.items() if ii == i))
    print("in %d: worst %.3f with %s, best %.3f with %s" % (i, v[0][0], v[0][1:], v[-1][0], v[-1][1:]))
