#!/usr/bin/env python3
"""hbm_alloc_probe.py -- does the speed of a job's memory streams depend on
WHICH allocations its arrays are?  (It does: profiles/r02/hbm_placement.txt;
tools/hbm_vmm_probe.cpp then separates physical from virtual placement.)
All modes time the arithmetic-free one-shot-tile copy of tools/libhbmprobe.so
over 2^30-word arrays allocated by torch (hipMalloc underneath).

  sets   [NSETS=6] [ROUNDS=6]  several sets of three arrays, all alive; each
         set timed again and again: a set that keeps its time while sets
         differ means placement, a time that moves for all sets means chip state
  roles                        ONE set (allocated 64 MiB larger): all six role
         assignments, then displacements of each array inside its allocation
  rw     [N=6]                 per array a read-only and a write-only sweep,
         then the 1R2W copy and the CORDIC cfg2 kernel with the fastest /
         slowest reader as input and the fastest / slowest writers as outputs
  pairs  [N=8]                 every (input; output pair) out of N arrays,
         grouped by their distance in allocation order
  slab   [GIB=40]              ONE allocation; the three arrays at chosen
         offsets inside it: role permutations, displacements of one array,
         spacings, shifts of the whole triple -- is there a rule in the
         addresses?
"""
import collections
import ctypes
import itertools
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libhbmprobe.so"))
lib.hbm_probe.restype = ctypes.c_float
lib.hbm_probe.argtypes = ([ctypes.c_void_p] * 4 + [ctypes.c_size_t]
                          + [ctypes.c_int] * 4 + [ctypes.c_void_p])
n = 1 << 30


def arrays(count, extra_words=0):
    arr = [torch.empty(n + extra_words, dtype=torch.int32, device="cuda")
           for _ in range(count)]
    for t in arr:
        t.zero_()
    torch.cuda.synchronize()
    return arr


def copy(i, o0, o1, reps=20, r=1, w=2):
    ms = lib.hbm_probe(i, None, o0, o1, n, r, w, 0, reps, None)
    return ms, 4.0 * (r + w) * n / (ms * 1e-3) / 8e12


def mode_sets(nsets=6, rounds=6):
    sets = [arrays(3) for _ in range(nsets)]
    print("# %d sets of 3 x 4 GiB; columns: round, then ms of the 1R2W one-shot-"
          "tile copy per set (20 launches each); 8 TB/s fraction below" % nsets)
    for s, (a, b, c) in enumerate(sets):
        print("# set %d at %x %x %x" % (s, a.data_ptr(), b.data_ptr(), c.data_ptr()))
    for r in range(rounds):
        row = [copy(a.data_ptr(), b.data_ptr(), c.data_ptr()) for a, b, c in sets]
        print("round %d  " % r + "  ".join("%.3f" % m for m, _ in row)
              + "   | " + "  ".join("%.3f" % f for _, f in row))
        sys.stdout.flush()
        time.sleep(0.5)
    a0, b0, c0 = sets[0]
    a1, b1, c1 = sets[-1]
    for name, (x, y, z) in (("set0 rotated", (b0, c0, a0)),
                            ("mixed 0/last", (a0, b1, c0)),
                            ("mixed last/0", (a1, b0, c1))):
        ms, f = copy(x.data_ptr(), y.data_ptr(), z.data_ptr())
        print("%-14s %.3f ms  %.3f" % (name, ms, f))


def mode_roles():
    arr = arrays(3, (64 << 20) // 4)
    P = [t.data_ptr() for t in arr]
    print("# arrays at %x %x %x" % tuple(P))
    print("# role assignments (in, out0, out1)")
    for perm in itertools.permutations(range(3)):
        ms, f = copy(P[perm[0]], P[perm[1]], P[perm[2]])
        print("roles %s  %.3f ms  %.3f" % (perm, ms, f))
    steps = (4096, 65536, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20)
    for k, name in enumerate(("in", "out0", "out1")):
        print("# displacement of %s inside its allocation (others fixed)" % name)
        for d in steps:
            q = list(P)
            q[k] += d
            ms, f = copy(*q)
            print("%s +%-9d %.3f ms  %.3f" % (name, d, ms, f))
    print("# all three displaced together")
    for d in (1 << 20, 2 << 20, 8 << 20, 32 << 20):
        ms, f = copy(P[0] + d, P[1] + d, P[2] + d)
        print("all +%-9d %.3f ms  %.3f" % (d, ms, f))


def mode_rw(N=6):
    sys.path.insert(0, ROOT)
    import cordic_amd as ca
    arr = arrays(N)
    P = [t.data_ptr() for t in arr]
    rd, wr = [], []
    for i in range(N):
        r, fr = copy(P[i], None, None, r=1, w=0)
        w, fw = copy(None, P[i], None, r=0, w=1)
        rd.append(r); wr.append(w)
        print("array %d at %x  read %.3f ms (%.3f of 8 TB/s)  write %.3f ms (%.3f)"
              % (i, P[i], r, fr, w, fw))
    plan = ca.Plan(ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 16))

    def kernel(i, a, b, reps=20):
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: ca.lib().cordic_plan_p2r_const(  # noqa: E731
            plan._h, n, 2**31 - 1, 0, P[i], P[a], P[b], st)
        run(); run()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        return ms, 12.0 * n / (ms * 1e-3) / 8e12

    order = sorted(range(N), key=lambda i: rd[i])
    wo = sorted(range(N), key=lambda i: wr[i])
    rest = [i for i in range(N) if i not in (order[0], order[-1])]
    cases = [("fastest reader as input", order[0], rest[0], rest[1]),
             ("slowest reader as input", order[-1], rest[0], rest[1])]
    for name, (a, b) in (("fastest writers", wo[:2]), ("slowest writers", wo[-2:])):
        cases.append((name, [i for i in order if i not in (a, b)][0], a, b))
    for name, i, a, b in cases:
        print("%s (in %d, out %d %d): copy %.3f ms (%.3f)   CORDIC cfg2 %.3f ms (%.3f)"
              % ((name, i, a, b) + copy(P[i], P[a], P[b]) + kernel(i, a, b)))


def mode_slab(gib=40):
    G, M = 1 << 30, 1 << 20
    slab = torch.empty(gib * G, dtype=torch.uint8, device="cuda")
    slab.zero_()
    torch.cuda.synchronize()
    base = (slab.data_ptr() + (2 * M - 1)) & ~(2 * M - 1)
    room = slab.data_ptr() + gib * G - base
    print("# slab of %d GiB at %x (2 MiB-aligned base %x); arrays of 4 GiB at "
          "byte offsets from the base; 1R2W one-shot-tile copy, 20 launches"
          % (gib, slab.data_ptr(), base))

    def run(tag, oi, o0, o1):
        assert max(oi, o0, o1) + 4 * G <= room, (tag, oi, o0, o1)
        ms, f = copy(base + oi, base + o0, base + o1)
        print("%-34s in %6.0f M  out0 %6.0f M  out1 %6.0f M   %.3f ms  %.3f"
              % (tag, oi / M, o0 / M, o1 / M, ms, f))
        sys.stdout.flush()
        return f

    print("## roles over the ranges 0 / 4 / 8 GiB")
    for perm in itertools.permutations((0, 4 * G, 8 * G)):
        run("roles", *perm)
    print("## the input displaced (outputs at 0 and 4 GiB)")
    for k in range(0, 12):
        run("in at 8 GiB + 2^%d MiB" % (k + 1), 8 * G + (2 * M << k), 0, 4 * G)
    print("## out1 displaced (in at 10 GiB... , out0 at 0)")
    for k in range(0, 11):
        run("out1 at 4 GiB + 2^%d MiB" % (k + 1), 10 * G + 2 * G, 0, 4 * G + (2 * M << k))
    print("## out0 displaced (the read array LOWEST)")
    for k in range(0, 11):
        run("in 0, out0 at 4 GiB + 2^%d MiB" % (k + 1), 0, 4 * G + (2 * M << k), 10 * G)
    print("## spacing: arrays at 0, 4 GiB + d, 8 GiB + 2 d")
    for k in range(0, 11):
        d = 2 * M << k
        run("d = 2^%d MiB, in lowest" % (k + 1), 0, 4 * G + d, 8 * G + 2 * d)
        run("d = 2^%d MiB, in highest" % (k + 1), 8 * G + 2 * d, 0, 4 * G + d)
    print("## the whole triple (0 / 4 / 8 GiB, in lowest; then in highest) shifted")
    for k in range(0, 14):
        sft = 2 * M << k
        if sft + 12 * G > room:
            break
        run("shift 2^%d MiB, in lowest" % (k + 1), sft, sft + 4 * G, sft + 8 * G)
        run("shift 2^%d MiB, in highest" % (k + 1), sft + 8 * G, sft, sft + 4 * G)


def mode_pairs(N=8):
    arr = arrays(N)
    P = [t.data_ptr() for t in arr]
    print("# arrays at " + " ".join("%x" % p for p in P))
    res = {}
    for i in range(N):
        for a, b in itertools.combinations([k for k in range(N) if k != i], 2):
            res[(i, a, b)] = copy(P[i], P[a], P[b], reps=8)[1]
    by = collections.defaultdict(list)
    for (i, a, b), f in res.items():
        by[(a - i, b - i)].append(f)
    print("# (out0 - in, out1 - in) in allocation steps: min .. max fraction (count)")
    for k in sorted(by):
        v = by[k]
        print("%-10s %.3f .. %.3f  (%d)" % (k, min(v), max(v), len(v)))
    print("# per input array: best and worst pair")
    for i in range(N):
        v = sorted((f, a, b) for (ii, a, b), f in res.items() if ii == i)
        print("in %d: worst %.3f with %s, best %.3f with %s"
              % (i, v[0][0], v[0][1:], v[-1][0], v[-1][1:]))


if __name__ == "__main__":
    modes = {"sets": mode_sets, "roles": mode_roles, "rw": mode_rw,
             "pairs": mode_pairs, "slab": mode_slab}
    if len(sys.argv) < 2 or sys.argv[1] not in modes:
        sys.exit(__doc__)
    modes[sys.argv[1]](*[int(v) for v in sys.argv[2:]])
