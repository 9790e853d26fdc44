#!/usr/bin/env python3
"""hbm_alloc_probe3.py -- is an allocation's READ speed what decides the 1R2W
stream?  N arrays of 2^30 words; per array a read-only and a write-only
one-shot-tile sweep; then the 1R2W copy with the fastest / slowest reader as
the input, and the CORDIC kernel itself (cordic_plan_p2r_const) on the same
assignments."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cordic_amd as ca  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libhbmprobe.so"))
lib.hbm_probe.restype = ctypes.c_float
lib.hbm_probe.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
n = 1 << 30
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
arr = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(N)]
for t in arr:
    t.zero_()
torch.cuda.synchronize()
P = [t.data_ptr() for t in arr]
rd, wr = [], []
for i in range(N):
    r = lib.hbm_probe(P[i], None, None, None, n, 1, 0, 0, 20, None)
    w = lib.hbm_probe(None, None, P[i], None, n, 0, 1, 0, 20, None)
    rd.append(r); wr.append(w)
    print("array %d at %x  read %.3f ms (%.3f of 8 TB/s)  write %.3f ms (%.3f)"
          % (i, P[i], r, 4.0 * n / (r * 1e-3) / 8e12, w, 4.0 * n / (w * 1e-3) / 8e12))
order = sorted(range(N), key=lambda i: rd[i])
best, worst = order[0], order[-1]
others = [i for i in range(N) if i not in (best, worst)]
o0, o1 = others[0], others[1]
cfg = ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 16)
plan = ca.Plan(cfg)


def kernel_ms(i, a, b, reps=20):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        ca.lib().cordic_plan_p2r_const(plan._h, n, 2**31 - 1, 0, P[i], P[a], P[b], st)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ca.lib().cordic_plan_p2r_const(plan._h, n, 2**31 - 1, 0, P[i], P[a], P[b], st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for name, i in (("fastest reader", best), ("slowest reader", worst)):
    ms = lib.hbm_probe(P[i], None, P[o0], P[o1], n, 1, 2, 0, 20, None)
    km = kernel_ms(i, o0, o1)
    print("%s (array %d) as input, outputs %d %d: copy %.3f ms (%.3f)   CORDIC cfg2 %.3f ms (%.3f)"
          % (name, i, o0, o1, ms, 12.0 * n / (ms * 1e-3) / 8e12, km, 12.0 * n / (km * 1e-3) / 8e12))
wo = sorted(range(N), key=lambda i: wr[i])
print("fastest writers %s, slowest %s" % (wo[:2], wo[-2:]))
for name, (a, b) in (("fastest writers", (wo[0], wo[1])), ("slowest writers", (wo[-1], wo[-2]))):
    cand = [i for i in order if i not in (a, b)]
    i = cand[0]
    ms = lib.hbm_probe(P[i], None, P[a], P[b], n, 1, 2, 0, 20, None)
    km = kernel_ms(i, a, b)
    print("%s %d %d, input %d: copy %.3f ms (%.3f)   CORDIC cfg2 %.3f ms (%.3f)"
          % (name, a, b, i, ms, 12.0 * n / (ms * 1e-3) / 8e12, km, 12.0 * n / (km * 1e-3) / 8e12))
