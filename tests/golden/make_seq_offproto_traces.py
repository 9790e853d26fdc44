#!/usr/bin/env python3
"""Regenerate tests/golden/seq_offproto_traces.json: port traces of the
SEQUENTIAL cores under i_stb activity that does NOT keep to the protocol --
i_stb held high for long stretches, i_stb exactly on completing clocks (chains
of them), random i_stb and i_reset -- obtained by executing the Verilog the
real reference generator emits (oracle/_ref/gencordic) with tests/vsim.py.
On a completing clock an i_stb keeps `idle` low without loading a sample
(rtl/seqcordic.v:229-236,270-291), so the free-running datapath goes round
again over its own result; these traces pin that behaviour.  Only data is
stored.

Run:  python tests/golden/make_seq_offproto_traces.py
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
sys.path.insert(0, HERE)
import vsim  # noqa: E402
from make_seq_traces import CORES, GEN  # noqa: E402


def main():
    if not os.path.exists(GEN):
        sys.exit("build oracle/_ref/gencordic first (make -C oracle ref)")
    out = {}
    rng = np.random.RandomState(20240928)
    for name, (args, _) in CORES.items():
        n = 1400
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
        # stretches: held high / dense / sparse / "only on completing clocks"
        kind = np.repeat(rng.choice([0, 1, 2, 3], n // 140 + 1), 140)[:n]
        stb = np.zeros(n, dtype=np.uint8)
        rs = (rng.randint(0, 400, n) == 0).astype(np.uint8)
        rs[:3] = 0
        outs = ["o_xval", "o_yval"] if rot else ["o_mag", "o_phase"]
        tr = {k: [] for k in outs + ["o_aux", "o_busy", "o_done"]}
        reruns = 0
        for t in range(n):
            busy = bool(m.get("o_busy"))
            # the clock on which o_done will rise next is the one where the
            # state register shows its last value
            completing = busy and m.get("state") >= (
                m.params.get("NSTAGES", 0) - 1 if rot else 0) and False
            k = kind[t]
            if k == 0:
                stb[t] = 1
            elif k == 1:
                stb[t] = rng.rand() < 0.5
            elif k == 2:
                stb[t] = rng.rand() < 0.05
            else:
                stb[t] = (not busy) and rng.rand() < 0.2
            pins = dict(i_xval=int(x[t]), i_yval=int(y[t]), i_stb=int(stb[t]),
                        i_reset=int(rs[t]), i_aux=int(aux[t]))
            if rot:
                pins["i_phase"] = int(ph[t])
            m.tick(**pins)
            for kk in outs:
                val = m.out(kk)
                tr[kk].append(val & ((1 << pw) - 1) if kk == "o_phase" else val)
            for kk in ("o_aux", "o_busy", "o_done"):
                tr[kk].append(int(m.get(kk)))
        # count o_done events that were not preceded by an idle accept
        e = {"args": args.replace("-a ", ""), "IW": iw, "PW": pw,
             "CLOCKS_PER_OUTPUT": cpo, "x": x.tolist(), "y": y.tolist(),
             "stb": stb.tolist(), "reset": rs.tolist(), "aux": aux.tolist()}
        if rot:
            e["phase"] = ph.tolist()
        e.update(tr)
        out[name] = e
        print(name, n, "clocks,", int(sum(tr["o_done"])), "o_done,",
              int(rs.sum()), "resets,", int(stb.sum()), "strobes")
    with open(os.path.join(HERE, "seq_offproto_traces.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
