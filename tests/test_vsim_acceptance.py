"""The reference's acceptance criteria evaluated on the reference's RTL as
EXECUTED by vsim (tests/golden/make_vsim_acceptance.py: all 2^20 phases through
rtl/cordic.v, the whole 2^21-point circle through rtl/topolar.v, the benches'
statistics on what the Verilog text produced).  The committed record holds
the report numbers, the verdicts and a SHA-256 of the complete output arrays;
the oracle (here) and the engine (-m gpu) must reproduce all of it.

vsim.py is this project's reading of Verilog, not Verilator: sample-level
parity stays "unpinned by a reference executor" (DESIGN.md section 2)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import quality as Q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def record():
    with open(os.path.join(ROOT, "tests", "golden", "vsim_acceptance.json")) as f:
        return json.load(f)


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_the_executed_rtl_passes_the_reference_criteria(record):
    c, t = record["cordic"], record["topolar"]
    assert c["samples"] == 1 << 20 and t["samples"] == 1 << 21
    assert c["pass"] and t["pass"]
    assert c["oracle_equal_at_generation"] and t["oracle_equal_at_generation"]
    # cordic_tb.cpp:315-337 thresholds, on the RTL's own outputs
    assert c["avg_err"] < 1.5 * c["expected_err"]
    assert c["max_err"] < 5.2 * c["expected_err"]
    assert abs(c["alpha"] - 1) < 0.01
    # topolar_tb.cpp:303-315
    assert t["max_phase_err"] < t["phase_limit"]
    assert t["max_mag_err"] < t["mag_limit"]


def test_oracle_reproduces_the_executed_rtl_sweeps(record):
    c = O.config_cli(O.P2R, 13, 13, 2)
    ph, x0, y0 = Q.p2r_bench_inputs(c.iw, c.pw)
    ox, oy = O.rotate(c, x0, y0, ph)
    assert sha(ox, oy) == record["cordic"]["sha256_outputs"]
    q = Q.p2r_quality(c, ph, x0, y0, ox, oy)
    assert q["averr"] == pytest.approx(record["cordic"]["avg_err"], rel=1e-12)
    assert q["mxerr"] == pytest.approx(record["cordic"]["max_err"], rel=1e-12)
    assert Q.sfdr_dbc(ox, oy) == pytest.approx(record["cordic"]["sfdr_dbc"], abs=1e-6)
    c = O.config_cli(O.R2P, 13, 13, 2)
    x, y, mg = Q.r2p_bench_inputs(c.iw, c.pw)
    assert sha(x, y) == record["topolar"]["sha256_inputs"]
    mag, p = O.topolar(c, x, y)
    assert sha(mag, p) == record["topolar"]["sha256_outputs"]
    q = Q.r2p_quality(c, x, y, mg, mag, p)
    assert q["mxperr"] == pytest.approx(record["topolar"]["max_phase_err"], rel=1e-12)
    assert q["mxverr"] == pytest.approx(record["topolar"]["max_mag_err"], rel=1e-12)


@pytest.mark.gpu
def test_gpu_reproduces_the_executed_rtl_sweeps(record):
    import cordic_amd as ca
    from gpu_util import gpu_p2r, gpu_r2p, gpu_plan_p2r
    c = O.config_cli(O.P2R, 13, 13, 2)
    cfg = ca.Config.from_cli(ca.P2R, 13, 13, 2)
    ph, x0, y0 = Q.p2r_bench_inputs(c.iw, c.pw)
    want = record["cordic"]["sha256_outputs"]
    assert sha(*gpu_p2r(cfg, x0, y0, ph)) == want                 # full recurrence
    plan = ca.Plan(cfg)
    assert sha(*gpu_plan_p2r(plan, x0, y0, ph)) == want            # seeded
    plan.close()
    assert sha(*gpu_p2r(cfg, np.full(ph.size, x0, np.int32),
                        np.zeros(ph.size, np.int32), ph)) == want  # vector feed
    cfg = ca.Config.from_cli(ca.R2P, 13, 13, 2)
    x, y, _ = Q.r2p_bench_inputs(13, 21)
    assert sha(x, y) == record["topolar"]["sha256_inputs"]
    assert sha(*gpu_r2p(cfg, x, y)) == record["topolar"]["sha256_outputs"]
