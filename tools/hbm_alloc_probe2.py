#!/usr/bin/env python3
"""hbm_alloc_probe2.py -- follow-up of hbm_alloc_probe.py: within ONE set of
three arrays (each allocated 64 MiB larger than needed), time the 1R2W
one-shot-tile copy for all six role assignments and for displacements of the
arrays inside their allocations."""
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
slack = 64 << 20
arr = [torch.empty(n + slack // 4, dtype=torch.int32, device="cuda") for _ in range(3)]
for t in arr:
    t.zero_()
torch.cuda.synchronize()
P = [t.data_ptr() for t in arr]
print("# arrays at %x %x %x" % tuple(P))


def run(i, o0, o1, reps=20):
    ms = lib.hbm_probe(i, None, o0, o1, n, 1, 2, 0, reps, None)
    return ms, 12.0 * n / (ms * 1e-3) / 8e12


print("# role assignments (in, out0, out1)")
for perm in itertools.permutations(range(3)):
    ms, f = run(P[perm[0]], P[perm[1]], P[perm[2]])
    print("roles %s  %.3f ms  %.3f" % (perm, ms, f))
print("# displacement of the INPUT array inside its allocation (others fixed)")
for d in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 48 << 20):
    ms, f = run(P[0] + d, P[1], P[2])
    print("in +%-9d %.3f ms  %.3f" % (d, ms, f))
print("# displacement of out0")
for d in (4096, 65536, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20):
    ms, f = run(P[0], P[1] + d, P[2])
    print("out0 +%-9d %.3f ms  %.3f" % (d, ms, f))
print("# displacement of out1")
for d in (4096, 65536, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20):
    ms, f = run(P[0], P[1], P[2] + d)
    print("out1 +%-9d %.3f ms  %.3f" % (d, ms, f))
print("# all three displaced together")
for d in (1 << 20, 2 << 20, 8 << 20, 32 << 20):
    ms, f = run(P[0] + d, P[1] + d, P[2] + d)
    print("all +%-9d %.3f ms  %.3f" % (d, ms, f))
