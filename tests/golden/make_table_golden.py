#!/usr/bin/env python3
"""Regenerate tests/golden/table_golden.json from the REAL reference generator
(oracle/_ref/gencordic): for -t tbl / -t qtr command lines keep the PW / OW it
derived and the contents of the .hex table it wrote (full table when small,
otherwise its length, a SHA-256 of the words and the first/last entries).

Run:  python tests/golden/make_table_golden.py
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GEN = os.path.join(ROOT, "oracle", "_ref", "gencordic")

CASES = ["-t tbl -p 6 -o 8", "-t tbl -p 9 -o 10", "-t tbl -o 7", "-t tbl -o 13",
         "-t tbl -i 12", "-t tbl -p 17 -o 13", "-t tbl -p 14",
         "-t qtr -p 7 -o 8", "-t qtr -p 10 -o 12", "-t qtr -i 8",
         "-t qtr -o 13", "-t qtr -p 18 -o 24", "-t qtr -p 16", "-t qtr -o 20"]


def main():
    if not os.path.exists(GEN):
        sys.exit("build oracle/_ref/gencordic first (make -C oracle ref)")
    out = {}
    for args in CASES:
        with tempfile.TemporaryDirectory() as td:
            vf = os.path.join(td, "core.v")
            r = subprocess.run([GEN] + args.split() + ["-f", vf],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(vf):
                out[args] = {"failed": True}
                continue
            v = open(vf).read()
            words = []
            for tok in open(os.path.join(td, "core.hex")).read().split():
                if not tok.startswith("@"):
                    words.append(int(tok, 16))
        e = {"PW": int(re.search(r"PW\s*=\s*(\d+)", v).group(1)),
             "OW": int(re.search(r"OW\s*=\s*(\d+)", v).group(1)),
             "entries": len(words),
             "sha256": hashlib.sha256(
                 b"".join(w.to_bytes(4, "little") for w in words)).hexdigest(),
             "head": words[:8], "tail": words[-8:]}
        if len(words) <= 512:
            e["words"] = words
        out[args] = e
    with open(os.path.join(HERE, "table_golden.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote %d entries" % len(out))


if __name__ == "__main__":
    main()
