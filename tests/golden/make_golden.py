#!/usr/bin/env python3
"""Regenerate tests/golden/gencordic_golden.json from the REAL reference
generator (oracle/_ref/gencordic, built by oracle/Makefile from
/root/reference/sw/*.cpp -- nothing of the reference is copied).

For every command line below the generator is run; from the Verilog it emits
we keep only DATA: the localparam values, the cordic_angle[] table and the
pre-rotation phase constants and the gain-annihilation multiplier of the
comment block (sw/cordiclib.cpp:205-209); from the C header it emits (-c) we keep the
constant lines between #ifndef/#endif.  These pin the parameter derivation
(sw/main.cpp:260-357), the angle table (sw/cordiclib.cpp:157-169) and the
header emission (sw/basiccordic.cpp:449-505, sw/topolar.cpp:412-451,
sw/seqcordic.cpp:446-500, sw/seqpolar.cpp:383-420).

Run:  python tests/golden/make_golden.py      (needs oracle/_ref/gencordic)
"""
import itertools
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GEN = os.path.join(ROOT, "oracle", "_ref", "gencordic")

NAMED = {
    # sw/Makefile:115,124,134,144 (NB=13, XTRA=2, CRDCARGS=-vca)
    "rtl_topolar":   "-vca -i 13 -o 13 -t r2p -x 2 -c",
    "rtl_seqpolar":  "-vca -i 13 -o 13 -t sr2p -x 2 -c",
    "rtl_cordic":    "-vca -v -i 13 -o 13 -t p2r -x 2 -c",
    "rtl_seqcordic": "-vca -v -i 13 -o 13 -t sp2r -x 2 -c",
    # BASELINE.json configs 1..5
    "cfg1": "-t p2r -i 16 -o 16 -p 16 -n 16 -c",
    "cfg2": "-t p2r -i 32 -o 32 -p 32 -n 16 -c",
    "cfg3": "-t r2p -i 24 -o 24 -n 20 -c",
    "cfg4": "-t p2r -i 32 -o 32 -p 32 -n 24 -c",
    "cfg5": "-t sp2r -i 32 -o 32 -p 32 -n 16 -c",
}


def sweep():
    cmds = dict(NAMED)
    k = 0
    for mode in ("p2r", "r2p", "sp2r", "sr2p"):
        for iw, ow in ((8, 8), (12, 16), (16, 12), (13, 13), (24, 24),
                       (18, 14), (5, 5), (3, 7), (28, 28)):
            for xtra in (0, 2, 5):
                for extra in ("", "-p 24", "-n 12", "-p 32 -n 30",
                              "-p 10 -n 18"):
                    k += 1
                    # keep the fixture small: a deterministic 1-in-3 subset
                    if k % 3:
                        continue
                    cmds["sweep%03d" % k] = "-t %s -i %d -o %d -x %d %s -c" % (
                        mode, iw, ow, xtra, extra)
    # defaults: -t with nothing else, only -o, only -i
    cmds["default_p2r"] = "-t p2r -c"
    cmds["default_r2p"] = "-t r2p -c"
    cmds["only_o_p2r"] = "-t p2r -o 17 -c"
    cmds["only_i_r2p"] = "-t r2p -i 11 -c"
    return cmds


def run_one(name, args):
    with tempfile.TemporaryDirectory() as td:
        vf = os.path.join(td, "core.v")
        cmd = [GEN] + args.split() + ["-f", vf]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(vf):
            return {"args": args, "failed": True}
        v = open(vf).read()
        hf = os.path.join(td, "core.h")
        h = open(hf).read() if os.path.exists(hf) else ""
    out = {"args": args}
    for key in ("IW", "OW", "NSTAGES", "XTRA", "WW", "PW"):
        m = re.search(r"\b%s=\s*(\d+)" % key, v)
        out[key] = int(m.group(1)) if m else None
    ang = re.findall(r"cordic_angle\[\s*(\d+)\]\s*=\s*\d+'h([0-9a-f_]+);", v)
    out["angles"] = [int(hx.replace("_", ""), 16) for _, hx in ang]
    out["angle_idx"] = [int(i) for i, _ in ang]
    # pre-rotation phase constants in emission order
    out["prerot_consts"] = [int(hx, 16) for hx in re.findall(
        r"(?:preph|ph\[0\])\s*<=\s*(?:i_phase - )?\d+'h([0-9a-f]+);", v)]
    out["rounds"] = ("Round our" in v)
    # "You can annihilate this gain by multiplying by 32'h%08x"
    # (sw/cordiclib.cpp:205-209)
    m = re.search(r"annihilate this gain by multiplying by 32'h([0-9a-f]+)", v)
    out["annihilate"] = int(m.group(1), 16) if m else None
    m = re.search(r"#ifndef.*#endif[^\n]*\n", h, re.S)
    out["header"] = m.group(0) if m else ""
    return out


def main():
    if not os.path.exists(GEN):
        sys.exit("build oracle/_ref/gencordic first (make -C oracle ref)")
    res = {}
    for name, args in sweep().items():
        res[name] = run_one(name, args)
    with open(os.path.join(HERE, "gencordic_golden.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    print("wrote %d entries" % len(res))


if __name__ == "__main__":
    main()
