#!/usr/bin/env python3
"""bench_row.py -- one bench.py run condensed to one text row (the A/B scripts'
unit): rate, HBM fraction, clock, power and the oracle verdict, read from the
run's detail file (the printed line is a selection of it).

    python tools/bench_row.py LABEL [bench.py arguments ...]
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    label, args = sys.argv[1], sys.argv[2:]
    fd, path = tempfile.mkstemp(prefix="bench_row_", suffix=".json", dir="/tmp")
    os.close(fd)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                            "--detail", path] + args, text=True,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            print(label, "FAILED rc", r.returncode, r.stderr[-300:].replace("\n", " | "))
            return 1
        with open(path) as f:
            d = json.load(f)
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    roof = d["roofline"]
    p = (roof.get("power") or {}).get("sustained") or {}
    thr = (roof.get("power") or {}).get("throttle") or {}
    print(label, round(d["value"]), "Msamples/s  frac", round(roof["frac"], 4),
          "kernel_ms", round(roof["kernel_ms_avg"], 4),
          "copy", round(roof.get("copy_frac") or 0, 3),
          "sclk", p.get("sclk_mhz_median"), "W", p.get("socket_w_median"),
          "ppt", round(thr.get("ppt_frac", 0), 2),
          "instr/sample", round((roof.get("valu") or {}).get("instr_per_sample") or 0, 1),
          "bit_exact", d.get("bit_exact_vs_oracle"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
