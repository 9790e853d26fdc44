#!/usr/bin/env python3
"""Regenerate tests/golden/getopt_golden.json from the REAL reference generator
(oracle/_ref/gencordic): command lines that probe how getopt(3) and
sw/main.cpp:139-232 treat repeated -t, the default file name, words that are
not options, "--" and -h.  Each line runs in an empty directory WITHOUT -f
unless the line has one, so the file that appears is the generator's own
choice; from the emitted Verilog only DATA is kept (which ports exist, the
localparams).

Run:  python tests/golden/make_getopt_golden.py
"""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GEN = os.path.join(ROOT, "oracle", "_ref", "gencordic")

CASES = [
    "-t sr2p -t p2r -i 13 -o 13",        # sequential is sticky; first -t names the file
    "-t sp2r -t r2p -i 12 -o 12",
    "-t p2r -t r2p -i 12 -o 12",
    "-t r2p -t p2r -i 12 -o 14",
    "-i 13 junk -o 14 -t p2r",           # a word that is no option is skipped
    "-t p2r -i 13 - -o 15",              # a lone dash is no option either
    "-t p2r -i 11 -- -o 5",              # "--" ends the options
    "-f a.v -t sp2r -i 10 -o 10",
    "-t p2r -f b.v -i 10 -o 10",
    "-t r2p -f c.v -t sp2r -i 10 -o 10",
    "-h -t p2r -i 13",                   # usage, exit 0, no core
    "-t p2r -i 13 -h",
    "-i -3 -t p2r -o 13",                # "-3" is the VALUE of -i
    "-t p2r -x 1 -x 4 -i 13 -o 13",      # last value wins
    "-t p2r -r -R -i 13 -o 13",
    "-t p2r -R -A -i 13 -o 13",
    "-vcat p2r -i13 -o13",               # bundle ending in an option with a value
    "-t p2r -i 13 -q",                   # unknown option
    "-t p2r -i",                         # missing value
]


def run(args):
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([GEN] + args.split(), cwd=td, capture_output=True,
                           text=True)
        files = sorted(os.listdir(td))
        out = {"args": args, "rc": r.returncode, "files": files}
        vs = [f for f in files if f.endswith(".v")]
        if r.returncode == 0 and len(vs) == 1:
            v = open(os.path.join(td, vs[0])).read()
            core = {"kind": "p2r" if re.search(r"\bi_phase\b", v) else "r2p",
                    "sequential": bool(re.search(r"\bo_busy\b", v)),
                    "has_reset": bool(re.search(r"\bi_reset\b|\bi_areset_n\b", v)),
                    "async_reset": bool(re.search(r"\bi_areset_n\b", v)),
                    "has_aux": bool(re.search(r"\bi_aux\b", v))}
            for key in ("IW", "OW", "NSTAGES", "WW", "PW"):
                m = re.search(r"\b%s=\s*(\d+)" % key, v)
                core[key] = int(m.group(1)) if m else None
            out["core"] = core
        return out


def main():
    if not os.path.exists(GEN):
        sys.exit("build oracle/_ref/gencordic first (make -C oracle ref)")
    res = [run(a) for a in CASES]
    with open(os.path.join(HERE, "getopt_golden.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for r in res:
        print(r["rc"], r["files"], r.get("core", {}).get("kind"),
              r.get("core", {}).get("sequential"), "|", r["args"])


if __name__ == "__main__":
    main()
