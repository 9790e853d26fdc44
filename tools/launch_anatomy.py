#!/usr/bin/env python3
"""launch_anatomy.py -- where does the time of ONE plan call go at 2^12 ... 2^26
samples?  Run under rocprofv3 --kernel-trace; `--report <kernel_trace.csv>`
then splits the per-launch time of back-to-back calls on one stream into the
kernel's own duration (End - Start of the dispatch) and the gap to the next
dispatch (command processor + runtime), per batch size.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -- python tools/launch_anatomy.py
    python tools/launch_anatomy.py --report out/*/*kernel_trace.csv
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = (12, 16, 20, 22, 24, 25, 26)
REPS = 60


def run():
    sys.path.insert(0, ROOT)
    import torch
    import cordic_amd as ca
    os.environ["CORDIC_SEED_MIN_SAMPLES"] = "0"
    dev = torch.device("cuda:0")
    cfg = ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    for lg in SIZES:
        n = 1 << lg
        ph = torch.empty(n, dtype=torch.int32, device=dev)
        a = torch.empty_like(ph)
        b = torch.empty_like(ph)
        ca.fill_phase_ramp(ph, 0, 2)
        torch.cuda.synchronize()
        for _ in range(REPS):
            plan.p2r_const(2**31 - 1, 0, ph, a, b)
        torch.cuda.synchronize()
    plan.close()


def report(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "rotator_seeded" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                             int(r["Grid_Size"]), int(r["Workgroup_Size"])))
    rows.sort()
    rows = [r for r in rows if r[2] > r[3]]         # (not the one-block image build)
    print("# cfg2's plan, %d back-to-back calls per size on one stream (rocprofv3 "
          "--kernel-trace):" % REPS)
    print("# log2(n)  kernel us (median)  gap to the next dispatch us (median)  "
          "sum  Gsample/s at the sum  ... at the kernel alone")
    for i, lg in enumerate(SIZES):
        grp = rows[i * REPS:(i + 1) * REPS]
        if len(grp) < REPS:
            break
        dur = sorted(e - s for s, e, _, _ in grp)
        gap = sorted(grp[k + 1][0] - grp[k][1] for k in range(len(grp) - 1))
        d, g = dur[len(dur) // 2] / 1e3, gap[len(gap) // 2] / 1e3
        n = 1 << lg
        print("  2^%-2d %10.2f %10.2f %10.2f %10.1f %10.1f" % (
            lg, d, g, d + g, n / (d + g) / 1e3, n / d / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run()
