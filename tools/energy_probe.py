#!/usr/bin/env python3
"""energy_probe.py -- joules per VALU instruction under SUSTAINED load: each
single-opcode variant of tools/sched_probe for 6 s from 8 waves a SIMD, socket
power and shader clock of THIS device from hwmon (median of the last 4 s),
against the idle reading.  The CORDIC kernels are held by the power limiter
(DESIGN.md 4.1): what an instruction costs in joules, not cycles, decides."""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_power import PowerSampler

VARIANTS = sys.argv[1:] or ["onlybitop3", "onlyxor", "onlyashr", "onlyalign", "only32",
                            "onlymadv", "onlymads", "onlymad", "sample_major", "nop_32"]
SIMDS = 1024


def sample(seconds, skip):
    sp = PowerSampler(0, period=0.02)
    sp.start()
    t0 = time.perf_counter()
    time.sleep(seconds)
    sp.stop()
    return sp.window(t0 + skip, t0 + seconds)


time.sleep(3)
idle = sample(2.0, 0.0)
print("# idle: %.0f W, %.0f MHz" % (idle["socket_w_median"], idle["sclk_mhz_median"]))
print("# variant: wave-instructions/s/SIMD held (cycles each at the clock held); socket W; "
      "sclk MHz; nJ per wave-instruction over idle; relative to v_xor_b32")
rows = []
for v in VARIANTS:
    p = subprocess.Popen([os.path.join(ROOT, "tools", "sched_probe"), "--sustain", v, "6"],
                         stdout=subprocess.PIPE, text=True)
    w = sample(5.5, 2.0)
    out = p.communicate()[0]
    m = re.search(r"([\d.e+]+) wave-instructions/s/SIMD", out)
    rate = float(m.group(1))
    nj = (w["socket_w_median"] - idle["socket_w_median"]) / (rate * SIMDS) * 1e9
    rows.append((v, rate, w, nj))
    time.sleep(2)
ref = next((r[3] for r in rows if r[0] == "onlyxor"), rows[0][3])
for v, rate, w, nj in rows:
    print("%-14s %.4g (%.2f cyc)  %6.0f W  %5.0f MHz  %.3f nJ  %.2fx" % (
        v, rate, w["sclk_mhz_median"] * 1e6 / rate, w["socket_w_median"],
        w["sclk_mhz_median"], nj, nj / ref))
