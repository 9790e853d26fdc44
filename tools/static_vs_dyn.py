#!/usr/bin/env python3
"""static_vs_dyn.py -- what does a static (fully unrolled, fixed stage count)
instance buy over the dynamic-exit instance of the same kernel, core by core?

    python tools/static_vs_dyn.py            # runs itself twice, prints the table

VERDICT r04 item 6(a): "measure every static instance against the DYN instance
on its own core and drop those within 2 %".  Child runs differ only in
CORDIC_FORCE_DYN (cordic_inst_body.h: every launch of an instantiation unit
goes to its dynamic-exit instance); rates in Gsample/s, 2^28 samples, median
of 5 x 10 launches."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cases():
    out = []
    for n in range(13, 30):
        out.append(("rot_lj29 full recurrence", "p2r", (0, 32, 32, 2, 32, n), 0x8))
        out.append(("seed_lj29 ramp", "p2r", (0, 32, 32, 2, 32, n), 0))
        out.append(("rot_lj29 per-sample vectors", "p2rxy", (0, 32, 32, 2, 32, n), 0x40))
        out.append(("rot_lj30 full recurrence", "p2r", (0, 29, 29, 2, 32, n), 0x8))
        out.append(("seed_lj30 ramp", "p2r", (0, 29, 29, 2, 32, n), 0))
    for n in range(16, 30):
        out.append(("pol_lj", "r2p", (1, 24, 24, 2, 32, n), 0))
    for n in (16, 24):
        out.append(("rot_wide2 (NO_LJ, WW 35)", "p2r", (0, 32, 32, 2, 32, n), 0x8 | 0x4))
        out.append(("rot_narrow (NO_LJ, WW 32)", "p2r", (0, 29, 29, 2, 32, n), 0x8 | 0x4))
        out.append(("seed_narrow (NO_LJ, WW 32)", "p2r", (0, 29, 29, 2, 32, n), 0x4))
        out.append(("rot_wide8 (WW 41)", "p2r", (0, 32, 32, 8, 32, n), 0))
        out.append(("rot_wideall (WW 48)", "p2r", (0, 32, 32, 15, 32, n), 0))
        out.append(("pol_narrow (NO_LJ)", "r2p", (1, 24, 24, 2, 32, n), 0x4))
        out.append(("pol_wideall (WW 46)", "r2p", (1, 32, 32, 5, 32, n), 0))
    return out


def child():
    import torch
    import cordic_amd as ca
    os.environ["CORDIC_SEED_MIN_SAMPLES"] = "0"
    n = 1 << 28
    dev = torch.device("cuda:0")
    ph = torch.empty(n, dtype=torch.int32, device=dev)
    x = torch.empty_like(ph)
    y = torch.empty_like(ph)
    a = torch.empty_like(ph)
    b = torch.empty_like(ph)
    ca.fill_phase_ramp(ph, 0, 2)
    ca.fill_iq_ramp(x, y, 0, 0x9E3779B1, 0x85EBCA77, 24)
    res = []
    for name, kind, cli, flags in cases():
        cfg = ca.Config.from_cli(*cli)
        if flags:
            cfg = cfg.with_flags(flags)
        plan = ca.Plan(cfg)
        amp = (1 << (cfg.iw - 1)) - 1
        if kind == "p2r":
            fn = lambda: plan.p2r_const(amp, 0, ph, a, b)
        elif kind == "p2rxy":
            fn = lambda: ca.p2r(cfg, x, y, ph, a, b)
        else:
            fn = lambda: ca.r2p(cfg, x, y, a, b)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        ts.sort()
        res.append([name, cfg.nlive, n / ts[2] / 1e6, ca.last_kernel()])
        plan.close()
    print(json.dumps(res))


def main():
    if "--child" in sys.argv:
        return child()
    runs = {}
    for mode in ("static", "dyn", "static2"):
        env = dict(os.environ)
        env.pop("CORDIC_FORCE_DYN", None)
        if mode == "dyn":
            env["CORDIC_FORCE_DYN"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"],
                           env=env, capture_output=True, text=True, timeout=1800)
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("[")]
        if r.returncode != 0 or not rows:
            sys.exit("child %s failed: %s" % (mode, r.stderr[-2000:]))
        runs[mode] = json.loads(rows[-1])
    print("# unit / feed, live stages: static Gsample/s (two runs), dynamic-exit, "
          "static over dynamic")
    for s1, d, s2 in zip(runs["static"], runs["dyn"], runs["static2"]):
        best = max(s1[2], s2[2])
        print("%-32s %2d  %7.1f %7.1f  %7.1f  %+5.1f %%" % (
            s1[0], s1[1], s1[2], s2[2], d[2], (best / d[2] - 1) * 100))


if __name__ == "__main__":
    main()
