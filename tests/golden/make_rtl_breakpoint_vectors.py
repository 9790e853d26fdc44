#!/usr/bin/env python3
"""Regenerate tests/golden/rtl_breakpoint_vectors.json: the emitted RTL of
BASELINE's rotators EXECUTED (tests/vsim.py, as in make_rtl_vectors.py) on the
phases where a rotation direction flips -- +/- 1 around the partial sums
+/- a_0 +/- a_1 ... of the arctan table, in every quadrant -- with random
per-sample vectors.  Round 4's table-driven kernels (direction tables of
cordic_plan_p2r, the seeds and direction tails of cordic_plan_p2r_const) take
their rotation directions from tables whose break points are exactly these
phases; this fixture pins them to what the reference's RTL text does there,
not only to the oracle.  Only data is stored; no Verilog text.

Run:  python tests/golden/make_rtl_breakpoint_vectors.py
      (needs oracle/_ref/gencordic; under a minute)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import vsim  # noqa: E402
from make_rtl_vectors import GEN, emit_raw, repair_truncating_core  # noqa: E402

CORES = {
    # name: (gencordic args, break-point depth, samples)
    "cfg2_breaks": ("-a -t p2r -i 32 -o 32 -p 32 -n 16", 12, 4000),
    "cfg4_breaks": ("-a -t p2r -i 32 -o 32 -p 32 -n 24", 14, 3000),
    "nat24_breaks": ("-a -t p2r -i 24 -o 24", 12, 2000),
    "nat16_breaks": ("-a -t p2r -i 16 -o 16", 10, 1500),
}


def angles(v):
    """cordic_angle[] of the emitted core, PW-bit integers, read off the
    executed module's memory initialisation (no text is stored)"""
    import re
    vals = re.findall(r"cordic_angle\[\s*(\d+)\]\s*=\s*\d+'h([0-9a-fA-F_]+)", v)
    out = {}
    for i, h in vals:
        out[int(i)] = int(h.replace("_", ""), 16)
    return [out[i] for i in range(len(out))]


def main():
    if not os.path.exists(GEN):
        sys.exit("build oracle/_ref/gencordic first (make -C oracle ref)")
    out = {}
    rng = np.random.RandomState(20260929)
    for name, (args, depth, n) in CORES.items():
        raw, _ = emit_raw(args)
        v = repair_truncating_core(raw)
        assert v == raw
        m = vsim.Module(v)
        iw, pw = m.params["IW"], m.params["PW"]
        ang = angles(v)
        sums = {0}
        for a in ang[:depth]:
            sums = sums | {s + a for s in sums} | {s - a for s in sums}
        base = np.array(sorted(sums), dtype=np.int64)
        base = base[np.abs(base) <= (1 << (pw - 3))]
        pick = base[rng.choice(base.size, n // 3 + 1, replace=base.size < n // 3 + 1)]
        ph = (pick[:, None] + np.array([-1, 0, 1])[None, :]).ravel()[:n]
        ph = (ph + (rng.randint(0, 4, n).astype(np.int64) << (pw - 2))) & ((1 << pw) - 1)
        lo, hi = -(1 << (iw - 1)), (1 << (iw - 1))
        x = rng.randint(lo, hi, n)
        y = rng.randint(lo, hi, n)
        x[:4] = [hi - 1, lo, hi - 1, 0]
        y[:4] = [0, lo, hi - 1, lo]
        samples = [dict(i_xval=int(x[i]), i_yval=int(y[i]), i_phase=int(ph[i]))
                   for i in range(n)]
        res = vsim.run_pipelined(m, samples)
        out[name] = {"args": args.replace("-a ", ""), "IW": iw,
                     "OW": m.params["OW"], "WW": m.params["WW"], "PW": pw,
                     "depth": depth,
                     "x": [int(t) for t in x], "y": [int(t) for t in y],
                     "phase": [int(t) for t in ph],
                     "o_xval": [r["o_xval"] for r in res],
                     "o_yval": [r["o_yval"] for r in res]}
        print(name, n, "samples,", base.size, "break points to depth", depth)
    with open(os.path.join(HERE, "rtl_breakpoint_vectors.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
