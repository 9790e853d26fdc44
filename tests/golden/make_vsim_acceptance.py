#!/usr/bin/env python3
"""The reference's acceptance criteria on the reference's RTL, EXECUTED.

bench/cpp/cordic_tb.cpp (:127-139 sweep, :223-337 statistics and thresholds)
and bench/cpp/topolar_tb.cpp (:127-141 circle, :222-256, :303-315) run the
Verilated rtl/cordic.v / rtl/topolar.v over 2^PW samples.  Verilator is absent
here, so this script executes the same checked-in Verilog with tests/vsim.py
over the same complete sweeps -- 2^20 phases through rtl/cordic.v, 2^21 circle
points through rtl/topolar.v, one sample per clock as testb.h steps the model --
evaluates the benches' statistics on what the RTL text produced, and stores
  * the report numbers and verdicts,
  * a SHA-256 of the complete output arrays,
in tests/golden/vsim_acceptance.json.  tests/test_vsim_acceptance.py then
requires the oracle (CPU) and the engine (GPU) to reproduce both: the criteria
are thereby evaluated on vsim-executed reference RTL rather than on the
restatement.  vsim.py is this project's reading of Verilog, not Verilator:
the status of sample-level parity stays "unpinned by a reference executor".

~10 minutes on 8 cores:  python tests/golden/make_vsim_acceptance.py
"""
import hashlib
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import quality as Q  # noqa: E402
import vsim  # noqa: E402

REF_RTL = "/root/reference/rtl"
WORKERS = 8


def _chunk(job):
    core, lo, hi = job
    m = vsim.Module(open(os.path.join(REF_RTL, core + ".v")).read())
    if core == "cordic":
        ph, x0, y0 = Q.p2r_bench_inputs(m.params["IW"], m.params["PW"])
        samples = [dict(i_xval=x0, i_yval=y0, i_phase=int(p)) for p in ph[lo:hi]]
        res = vsim.run_pipelined(m, samples)
        return (np.array([r["o_xval"] for r in res], dtype=np.int32),
                np.array([r["o_yval"] for r in res], dtype=np.int32))
    x, y, _ = Q.r2p_bench_inputs(m.params["IW"], m.params["PW"])
    samples = [dict(i_xval=int(a), i_yval=int(b)) for a, b in zip(x[lo:hi], y[lo:hi])]
    res = vsim.run_pipelined(m, samples)
    pm = (1 << m.params["PW"]) - 1
    return (np.array([r["o_mag"] for r in res], dtype=np.int32),
            np.array([r["o_phase"] & pm for r in res], dtype=np.uint32))


def sweep(core, n):
    step = 1 << 13
    jobs = [(core, lo, min(n, lo + step)) for lo in range(0, n, step)]
    with Pool(WORKERS) as pool:
        parts = pool.map(_chunk, jobs, chunksize=1)
    return (np.concatenate([p[0] for p in parts]),
            np.concatenate([p[1] for p in parts]))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    out = {}
    c = O.config_cli(O.P2R, 13, 13, 2)
    ph, x0, y0 = Q.p2r_bench_inputs(c.iw, c.pw)
    ox, oy = sweep("cordic", ph.size)
    q = Q.p2r_quality(c, ph, x0, y0, ox, oy)
    rx, ry = O.rotate(c, x0, y0, ph)
    out["cordic"] = {
        "rtl": "rtl/cordic.v", "samples": int(ph.size),
        "avg_err": float(q["averr"]), "max_err": float(q["mxerr"]),
        "alpha": float(q["alpha"]), "cnr_db": float(q["cnr"]),
        "expected_err": float(q["sigma"]), "pass": bool(q["ok"]),
        "sfdr_dbc": float(Q.sfdr_dbc(ox, oy)),
        "sha256_outputs": sha(ox, oy),
        "oracle_equal_at_generation": bool(np.array_equal(ox, rx)
                                           and np.array_equal(oy, ry))}
    print(out["cordic"])
    c = O.config_cli(O.R2P, 13, 13, 2)
    x, y, mg = Q.r2p_bench_inputs(c.iw, c.pw)
    mag, oph = sweep("topolar", x.size)
    q = Q.r2p_quality(c, x, y, mg, mag, oph)
    rm, rp = O.topolar(c, x, y)
    out["topolar"] = {
        "rtl": "rtl/topolar.v", "samples": int(x.size),
        "max_phase_err": float(q["mxperr"]), "max_mag_err": float(q["mxverr"]),
        "phase_limit": float(q["phase_limit"]), "mag_limit": float(q["mag_limit"]),
        "pass": bool(q["ok"]),
        "sha256_inputs": sha(x, y),
        "sha256_outputs": sha(mag, oph),
        "oracle_equal_at_generation": bool(np.array_equal(mag, rm)
                                           and np.array_equal(oph, rp))}
    print(out["topolar"])
    with open(os.path.join(HERE, "vsim_acceptance.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
