#!/usr/bin/env python3
"""Regenerate tests/golden/stream_vectors.json: clock-by-clock port traces of
the PIPELINED cores under random i_ce / i_reset / i_aux activity, obtained by
EXECUTING the Verilog the real reference generator emits
(oracle/_ref/gencordic) with tests/vsim.py.  Every clock stores the inputs that
were applied and the outputs as they stand after that clock -- what a
Verilator bench reads after tick() (bench/cpp/testb.h:87-106).  Only data is
stored; no Verilog text.

Run:  python tests/golden/make_stream_vectors.py   (needs oracle/_ref/gencordic)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vsim  # noqa: E402

GEN = os.path.join(ROOT, "oracle", "_ref", "gencordic")

CORES = {
    "rtl_cordic":  ("-a -t p2r -i 13 -o 13 -x 2", 1600),
    "rtl_topolar": ("-a -t r2p -i 13 -o 13 -x 2", 1600),
    "cfg2":        ("-a -t p2r -i 32 -o 32 -p 32 -n 16", 900),
    "cfg3":        ("-a -t r2p -i 24 -o 24 -n 20", 900),
    "skip_p2r":    ("-a -t p2r -i 8 -o 8 -x 1 -p 32 -n 30", 700),   # i >= WW
    "skip_r2p":    ("-a -t r2p -i 6 -o 6 -x 0 -p 12 -n 14", 700),   # angle == 0
    "trunc_p2r":   ("-a -t p2r -i 12 -o 12 -x 0 -p 16", 500),       # WW == OW+1
}


def activity(rng, n, lat):
    """i_ce / i_reset patterns: full rate, throttled, long stalls, reset
    pulses (also together with i_ce), reset shortly after reset."""
    ce = np.ones(n, dtype=np.uint8)
    rs = np.zeros(n, dtype=np.uint8)
    a = n // 6
    ce[a:2 * a] = rng.randint(0, 2, a)
    ce[2 * a:3 * a] = (rng.randint(0, 5, a) == 0)
    ce[3 * a + 10:3 * a + 10 + 3 * lat] = 0            # long stall
    ce[4 * a:5 * a] = rng.randint(0, 2, a)
    for t in (a // 2, a + a // 2, 3 * a + 5, 4 * a + 7, 4 * a + 9,
              4 * a + 9 + lat // 2, 5 * a + 3):
        rs[t] = 1
    rs[2 * a + 20:2 * a + 24] = 1                      # held reset
    return ce, rs


def main():
    if not os.path.exists(GEN):
        sys.exit("build oracle/_ref/gencordic first (make -C oracle ref)")
    out = {}
    rng = np.random.RandomState(20240919)
    for name, (args, n) in CORES.items():
        with tempfile.TemporaryDirectory() as td:
            vf = os.path.join(td, "core.v")
            subprocess.run([GEN] + args.split() + ["-f", vf], check=True,
                           capture_output=True)
            v = open(vf).read().replace("// }}}\talways", "// }}}\n\talways")
        m = vsim.Module(v)
        iw, pw, ns = m.params["IW"], m.params["PW"], m.params["NSTAGES"]
        rot = "i_phase" in m.decl
        lo, hi = -(1 << (iw - 1)), (1 << (iw - 1))
        x = rng.randint(lo, hi, n)
        y = rng.randint(lo, hi, n)
        ph = rng.randint(0, 1 << pw, n, dtype=np.int64)
        aux = rng.randint(0, 2, n).astype(np.uint8)
        ce, rs = activity(rng, n, ns + 2)
        outs = ["o_xval", "o_yval"] if rot else ["o_mag", "o_phase"]
        tr = {k: [] for k in outs + ["o_aux"]}
        for t in range(n):
            pins = dict(i_xval=int(x[t]), i_yval=int(y[t]), i_ce=int(ce[t]),
                        i_reset=int(rs[t]), i_aux=int(aux[t]))
            if rot:
                pins["i_phase"] = int(ph[t])
            m.tick(**pins)
            for k in outs:
                val = m.out(k)
                tr[k].append(val & ((1 << pw) - 1) if k == "o_phase" else val)
            tr["o_aux"].append(int(m.get("o_aux")))
        e = {"args": args.replace("-a ", ""), "IW": iw, "PW": pw, "NSTAGES": ns,
             "x": x.tolist(), "y": y.tolist(), "ce": ce.tolist(),
             "reset": rs.tolist(), "aux": aux.tolist()}
        if rot:
            e["phase"] = ph.tolist()
        e.update(tr)
        out[name] = e
        print(name, n, "clocks,", int(ce.sum()), "enabled,", int(rs.sum()),
              "reset")
    with open(os.path.join(HERE, "stream_vectors.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
