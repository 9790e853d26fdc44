#!/usr/bin/env python3
"""Regenerate tests/golden/seq_traces.json: clock-by-clock port traces of the
SEQUENTIAL cores (i_stb / o_busy / o_done handshake) obtained by executing the
Verilog the real reference generator emits (oracle/_ref/gencordic) with
tests/vsim.py.  i_stb is random (also while the core is busy, where the RTL
ignores it) except on the clock that completes a sample, where the RTL's
behaviour is off protocol (see tests/seq_model.py); i_reset pulses at random.
Only data is stored.

Run:  python tests/golden/make_seq_traces.py   (needs oracle/_ref/gencordic)
"""
import json
import os
import re
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
    "rtl_seqcordic": ("-a -t sp2r -i 13 -o 13 -x 2", 2500),
    "rtl_seqpolar":  ("-a -t sr2p -i 13 -o 13 -x 2", 2500),
    "cfg5_seq":      ("-a -t sp2r -i 32 -o 32 -p 32 -n 16", 1500),
    "seq_cfg3":      ("-a -t sr2p -i 24 -o 24 -n 20", 1500),
    "sp2r_n12":      ("-a -t sp2r -i 10 -o 12 -x 1 -p 18 -n 12", 1200),
    "sr2p_n9":       ("-a -t sr2p -i 8 -o 8 -x 1 -p 14 -n 9", 1200),
}


def main():
    if not os.path.exists(GEN):
        sys.exit("build oracle/_ref/gencordic first (make -C oracle ref)")
    out = {}
    rng = np.random.RandomState(20240920)
    for name, (args, n) in CORES.items():
        with tempfile.TemporaryDirectory() as td:
            vf = os.path.join(td, "core.v")
            subprocess.run([GEN] + args.split() + ["-c", "-f", vf], check=True,
                           capture_output=True)
            v = open(vf).read()
            h = open(os.path.join(td, "core.h")).read()
        cpo = int(re.search(r"CLOCKS_PER_OUTPUT\t(\d+)", h).group(1))
        m = vsim.Module(v)
        iw, pw = m.params["IW"], m.params["PW"]
        rot = "i_phase" in m.decl
        lo, hi = -(1 << (iw - 1)), (1 << (iw - 1))
        x, y = rng.randint(lo, hi, n), rng.randint(lo, hi, n)
        ph = rng.randint(0, 1 << pw, n, dtype=np.int64)
        aux = rng.randint(0, 2, n).astype(np.uint8)
        # i_stb density varies along the trace: back to back, sparse, bursts
        dens = np.repeat(rng.choice([1.0, 0.6, 0.08, 0.02, 0.3], n // 100 + 1),
                         100)[:n]
        stb = (rng.rand(n) < dens).astype(np.uint8)
        rs = (rng.randint(0, 300, n) == 0).astype(np.uint8)
        rs[:3] = 0
        outs = ["o_xval", "o_yval"] if rot else ["o_mag", "o_phase"]
        tr = {k: [] for k in outs + ["o_aux", "o_busy", "o_done"]}
        left = 0                          # clocks until the sample completes
        for t in range(n):
            if left == 1 and rng.rand() < 0.06:
                rs[t] = 1                 # reset ON the completing clock
            if left == 1 and not rs[t]:
                stb[t] = 0                # keep to the protocol on this clock
            pins = dict(i_xval=int(x[t]), i_yval=int(y[t]), i_stb=int(stb[t]),
                        i_reset=int(rs[t]), i_aux=int(aux[t]))
            if rot:
                pins["i_phase"] = int(ph[t])
            was_idle = not m.get("o_busy")
            m.tick(**pins)
            if rs[t]:
                left = 0
            elif left == 0:
                left = cpo - 1 if (stb[t] and was_idle) else 0
            else:
                left -= 1
            for k in outs:
                val = m.out(k)
                tr[k].append(val & ((1 << pw) - 1) if k == "o_phase" else val)
            for k in ("o_aux", "o_busy", "o_done"):
                tr[k].append(int(m.get(k)))
        e = {"args": args.replace("-a ", ""), "IW": iw, "PW": pw,
             "CLOCKS_PER_OUTPUT": cpo, "x": x.tolist(), "y": y.tolist(),
             "stb": stb.tolist(), "reset": rs.tolist(), "aux": aux.tolist()}
        if rot:
            e["phase"] = ph.tolist()
        e.update(tr)
        out[name] = e
        print(name, n, "clocks,", int(sum(tr["o_done"])), "results,",
              int(rs.sum()), "resets")
    with open(os.path.join(HERE, "seq_traces.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
