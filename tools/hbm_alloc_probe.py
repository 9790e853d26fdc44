#!/usr/bin/env python3
"""hbm_alloc_probe.py -- does the speed of the 1R2W stream depend on WHICH
memory the three arrays got?  One process allocates several sets of three
2^30-word arrays (all kept alive, so every set is different physical memory),
times the arithmetic-free one-shot-tile copy (tools/libhbmprobe.so, mode 0) on
each set, then goes over the sets again and again: a set whose time stays put
while sets differ means placement, a time that moves for all sets together
means chip state."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libhbmprobe.so"))
lib.hbm_probe.restype = ctypes.c_float
lib.hbm_probe.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t] + [ctypes.c_int] * 4 + [ctypes.c_void_p]

nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << 30
sets = []
for s in range(nsets):
    a = torch.empty(n, dtype=torch.int32, device="cuda")
    b = torch.empty(n, dtype=torch.int32, device="cuda")
    c = torch.empty(n, dtype=torch.int32, device="cuda")
    a.zero_(); b.zero_(); c.zero_()
    sets.append((a, b, c))
torch.cuda.synchronize()
print("# %d sets of 3 x 4 GiB; columns: round, then ms of the 1R2W one-shot-tile "
      "copy per set (20 launches each); 8 TB/s fraction below" % nsets)
for s, (a, b, c) in enumerate(sets):
    print("# set %d at %x %x %x" % (s, a.data_ptr(), b.data_ptr(), c.data_ptr()))
for r in range(rounds):
    row = []
    for a, b, c in sets:
        ms = lib.hbm_probe(a.data_ptr(), None, b.data_ptr(), c.data_ptr(), n, 1, 2, 0, 20, None)
        row.append(ms)
    print("round %d  " % r + "  ".join("%.3f" % m for m in row)
          + "   | " + "  ".join("%.3f" % (12.0 * n / (m * 1e-3) / 8e12) for m in row))
    sys.stdout.flush()
    time.sleep(0.5)
# the same with the role of the arrays permuted within set 0 and across sets
a0, b0, c0 = sets[0]
a1, b1, c1 = sets[-1]
for name, (x, y, z) in (("set0 rotated", (b0, c0, a0)), ("mixed 0/last", (a0, b1, c0)),
                        ("mixed last/0", (a1, b0, c1))):
    ms = lib.hbm_probe(x.data_ptr(), None, y.data_ptr(), z.data_ptr(), n, 1, 2, 0, 20, None)
    print("%-14s %.3f ms  %.3f" % (name, ms, 12.0 * n / (ms * 1e-3) / 8e12))
