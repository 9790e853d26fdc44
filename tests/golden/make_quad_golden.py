#!/usr/bin/env python3
"""Regenerate tests/golden/quad_golden.json for the quadratically interpolated
sine core (gencordic -t qtbl).

For every parameter set the REAL reference generator (oracle/_ref/gencordic,
built from /root/reference/sw by oracle/Makefile) is run; stored are
  * the localparams it emits (LGTBL, QBITS, LBITS, CBITS, XTRA, PW, OW),
  * the constants of the header it writes,
  * the three coefficient tables it writes (<name>_{c,l,q}tbl.hex), as values,
  * per-sample vectors obtained by EXECUTING the emitted Verilog with
    tests/vsim.py (this project's own simulator, not Verilator) clock by clock
    the way bench/cpp/quadtbl_tb.cpp drives the core.
Only data is stored; no Verilog text.

Run:  python tests/golden/make_quad_golden.py   (needs oracle/_ref/gencordic)
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
    # name: (gencordic args, samples run through vsim)
    "rtl_quadtbl": ("-t qtbl -o 13 -p 18", 1500),      # the checked-in core
    "o16":         ("-t qtbl -o 16", 800),
    "o8p12":       ("-t qtbl -o 8 -p 12", 800),
    "o24":         ("-t qtbl -o 24", 500),              # PW 31, 512 entries
    "o24p32":      ("-t qtbl -o 24 -p 32", 500),
    "o20x4p24":    ("-t qtbl -o 20 -x 4 -p 24", 500),
    "o10x1":       ("-t qtbl -o 10 -x 1", 800),         # nxtra 2
    "i14o10p20":   ("-t qtbl -i 14 -o 10 -p 20", 800),
    "o28x1p30":    ("-t qtbl -o 28 -x 1 -p 30", 400),   # CBITS 30, 1024 entries
    # nxtra = 1: tables 1 bit narrower than WW = OW + 2, r_value[WW-1] does not
    # exist -- the emitted core cannot elaborate; the engine must refuse it
    "o12x0p16":    ("-t qtbl -o 12 -x 0 -p 16", 0),
}


def read_hex(path, bits):
    v = [int(t, 16) for t in open(path).read().split() if not t.startswith("@")]
    return [x - (1 << bits) if x >> (bits - 1) else x for x in v]


def main():
    if not os.path.exists(GEN):
        sys.exit("build oracle/_ref/gencordic first (make -C oracle ref)")
    out = {}
    rng = np.random.RandomState(20240918)
    for name, (args, n) in CORES.items():
        with tempfile.TemporaryDirectory() as td:
            vf = os.path.join(td, "core.v")
            subprocess.run([GEN, "-a", "-c"] + args.split() + ["-f", vf],
                           check=True, capture_output=True)
            vtext = open(vf).read()
            htext = open(os.path.join(td, "core.h")).read()
            lp = {k: int(v) for k, v in re.findall(
                r"\b(PW|OW|XTRA|LGTBL|QBITS|LBITS|CBITS)\s*=\s*(\d+)", vtext)}
            hdr = {k: v for k, v in re.findall(
                r"const\t\w+\t(\w+)\s*= ([^;]+);", htext)}
            e = {"args": args, "localparams": lp, "header": hdr}
            ww = lp["OW"] + lp["XTRA"]
            if lp["CBITS"] < ww:
                e["elaborates"] = False
                out[name] = e
                print(name, "does not elaborate (CBITS %d < WW %d)"
                      % (lp["CBITS"], ww))
                continue
            e["elaborates"] = True
            for t, b in (("c", "CBITS"), ("l", "LBITS"), ("q", "QBITS")):
                e[t + "tbl"] = read_hex(os.path.join(td, "core_%stbl.hex" % t),
                                        lp[b])
            m = vsim.Module(vtext, readmem_dir=td)
            pw = lp["PW"]
            dx = pw - lp["LGTBL"]
            ph = rng.randint(0, 1 << pw, n, dtype=np.int64)
            # interval edges, the peaks (where the no-overflow cases of the
            # rounding apply) and the zero crossings
            edges = [0, 1, (1 << pw) - 1, 1 << (pw - 1), (1 << (pw - 1)) - 1]
            for quarter in (1, 3):
                c = quarter << (pw - 2)
                edges += [c + d for d in range(-6, 7)]
            for k in (0, 1, 5, (1 << lp["LGTBL"]) - 1):
                edges += [(k << dx) + d for d in (0, 1, (1 << dx) - 1)]
            for j, v in enumerate(edges):
                ph[j] = v % (1 << pw)
            res = vsim.run_pipelined(m, [dict(i_phase=int(p)) for p in ph])
            e["phase"] = [int(p) for p in ph]
            e["o_sin"] = [r["o_sin"] for r in res]
        out[name] = e
        print(name, lp, n, "samples")
    with open(os.path.join(HERE, "quad_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
