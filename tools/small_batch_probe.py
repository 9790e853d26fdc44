#!/usr/bin/env python3
"""Per-launch time of the constant-vector rotator against the batch size: the
table-seeded kernel with its prologue served from the plan's image (round 5)
and with every block rebuilding the seed table (round 4), against the full
recurrence (no prologue), cfg2's and cfg4's cores.  Where is the crossover,
i.e. below which n should a plan launch the plain kernel?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
os.environ["CORDIC_SEED_MIN_SAMPLES"] = "0"     # both kernels at every size
import cordic_amd as ca

dev = torch.device("cuda:0")
for ns in (16, 24):
    cfg = ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, ns)
    plan = ca.Plan(cfg)
    os.environ["CORDIC_SEED_IMAGES"] = "0"      # round 4: every block computes
    noimg = ca.Plan(cfg)                        # its own prologue
    del os.environ["CORDIC_SEED_IMAGES"]
    plain = ca.Plan(cfg.with_flags(ca.FLAG_NO_SEED))
    print("p2r %d stages: n, seeded (plan image) us, seeded (prologue per block) us, "
          "plain us, the three in Gs/s" % ns)
    for lg in range(12, 28):
        n = 1 << lg
        ph = torch.empty(n, dtype=torch.int32, device=dev)
        a = torch.empty_like(ph); b = torch.empty_like(ph)
        ca.fill_phase_ramp(ph, 0, 2)
        res = []
        for p in (plan, noimg, plain):
            reps = 200 if lg < 22 else 40
            for _ in range(5):
                p.p2r_const(2**31 - 1, 0, ph, a, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                p.p2r_const(2**31 - 1, 0, ph, a, b)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / reps * 1e3)
        print("  2^%-2d %9.2f %9.2f %9.2f %8.1f %8.1f %8.1f" % (
            lg, res[0], res[1], res[2], n / res[0] / 1e3, n / res[1] / 1e3,
            n / res[2] / 1e3))
    plan.close(); noimg.close(); plain.close()
# many small jobs in ONE launch (cordic_jobset): 2^26 samples as 2^(26-lg) jobs
# of 2^lg samples, against the same jobs as one plan call each
for ns in (16, 24):
    cfg = ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, ns)
    plan = ca.Plan(cfg)
    plan.set_min_samples(-1)
    os.environ.pop("CORDIC_SEED_MIN_SAMPLES", None)
    total = 1 << 26
    ph = torch.empty(total, dtype=torch.int32, device=dev)
    a = torch.empty_like(ph); b = torch.empty_like(ph)
    ca.fill_phase_ramp(ph, 0, 2)
    print("p2r %d stages, 2^26 samples as jobs of 2^lg: lg, jobs, job set us "
          "(Gs/s), one call per job us (Gs/s)" % ns)
    for lg in (12, 14, 16, 18, 20, 22, 24, 26):
        n = 1 << lg
        nj = total // n
        jobs = [dict(phase=ph[k * n:(k + 1) * n], ox=a[k * n:(k + 1) * n],
                     oy=b[k * n:(k + 1) * n], n=n) for k in range(nj)]
        js = ca.Jobset(plan, ca.JOBS_PHASE_ARRAYS, jobs)
        res = []

        def one_by_one():
            for jb in jobs:
                plan.p2r_const(2**31 - 1, 0, jb["phase"], jb["ox"], jb["oy"])
        for fn, reps in ((lambda: js.run(2**31 - 1, 0), 20),
                         (one_by_one, 2 if nj > 1024 else 5)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / reps * 1e3)
        print("  2^%-2d %6d %10.1f (%6.1f) %12.1f (%6.1f)" % (
            lg, nj, res[0], total / res[0] / 1e3, res[1], total / res[1] / 1e3))
        js.close()
    plan.close()
os.environ["CORDIC_SEED_MIN_SAMPLES"] = "0"
# per-sample vectors: directions looked up (small tables) against the recurrence
cfg = ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 16)
plan = ca.Plan(cfg)
print("p2r 16 stages, per-sample vectors: n, directions us, recurrence us")
for lg in range(12, 26, 2):
    n = 1 << lg
    ph = torch.empty(n, dtype=torch.int32, device=dev)
    x = torch.empty_like(ph); y = torch.empty_like(ph)
    a = torch.empty_like(ph); b = torch.empty_like(ph)
    ca.fill_phase_ramp(ph, 0, 2)
    ca.fill_iq_ramp(x, y, 0, 0x9E3779B1, 0x85EBCA77, 32)
    res = []
    for fn in (lambda: plan.p2r(x, y, ph, a, b), lambda: ca.p2r(cfg, x, y, ph, a, b)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 100 * 1e3)
    print("  2^%-2d %9.2f %9.2f" % (lg, res[0], res[1]))
plan.close()
