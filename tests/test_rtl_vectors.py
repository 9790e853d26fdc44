"""The oracle (and, on the GPU, the engine) against the reference's RTL text.

tests/golden/rtl_vectors.json holds per-sample vectors produced by executing
the Verilog emitted by the real reference generator with tests/vsim.py
(tests/golden/make_rtl_vectors.py).  The live tests below additionally run
vsim on the checked-in rtl/*.v (when /root/reference is mounted) and on fresh
parameter sets emitted by oracle/_ref/gencordic (when it is built)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import vsim
from test_oracle_golden import MODES, parse_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(O.ORACLE_DIR, "_ref", "gencordic")
REF_RTL = "/root/reference/rtl"


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(ROOT, "tests", "golden", "rtl_vectors.json")) as f:
        return json.load(f)


def oracle_cfg(args):
    d = parse_args(args)
    return O.config_cli(d["mode"], d["iw"], d["ow"], d["xtra"], d["pw"], d["n"])


def test_oracle_reproduces_every_rtl_vector(vectors):
    assert len(vectors) >= 15
    total = 0
    for name, e in vectors.items():
        c = oracle_cfg(e["args"])
        assert (c.iw, c.ow, c.ww, c.pw) == (e["IW"], e["OW"], e["WW"], e["PW"])
        x = np.array(e["x"], dtype=np.int32)
        y = np.array(e["y"], dtype=np.int32)
        if "phase" in e:
            ox, oy = O.rotate(c, x, y, np.array(e["phase"], dtype=np.uint32))
            assert ox.tolist() == e["o_xval"], name
            assert oy.tolist() == e["o_yval"], name
        else:
            mag, ph = O.topolar(c, x, y)
            assert mag.tolist() == e["o_mag"], name
            assert ph.tolist() == e["o_phase"], name
        total += len(e["x"])
    assert total >= 10000


@pytest.mark.skipif(not os.path.exists(GEN), reason="oracle/_ref/gencordic "
                    "not built (make -C oracle ref)")
def test_the_only_text_repair_is_the_truncating_core(vectors):
    """A finding about the reference, not a convenience: for WW == OW+1 the
    generator comments its own `always` header out (sw/basiccordic.cpp:418-419),
    so the core as emitted has no output registers.  vsim must reject or
    mis-elaborate the raw text (no o_xval assignment is ever executed), the
    repaired text must differ by exactly that one line break, and no other
    core's text may be touched."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_rtl_vectors as mk
    for name, (args, _) in mk.CORES.items():
        raw, _ = mk.emit_raw(args)
        fixed = mk.repair_truncating_core(raw)
        e = vectors[name]
        truncating = ("phase" in e) and e["WW"] == e["OW"] + 1
        assert e.get("repaired_text", False) == truncating, name
        if not truncating:
            assert fixed == raw, name
            continue
        assert raw.count(mk.BROKEN) == 1
        assert fixed == raw.replace(mk.BROKEN, mk.REPAIRED, 1)
        # as emitted: either unparsable, or a core whose outputs never change
        try:
            m = vsim.Module(raw)
        except (SyntaxError, KeyError, IndexError):
            continue
        res = vsim.run_pipelined(m, [dict(i_xval=1000, i_yval=-700,
                                          i_phase=12345)] * 4)
        assert all(r["o_xval"] == 0 and r["o_yval"] == 0 for r in res)


def test_vectors_cover_the_corner_semantics(vectors):
    """The fixture must really exercise what is hard: WW-bit overflow of a
    tiny core, the WW == OW+1 truncation branch, stages past WW."""
    t = vectors["tiny_wrap"]
    assert t["WW"] == 3 and max(abs(v) for v in t["o_xval"]) <= 2
    assert vectors["trunc_p2r"]["WW"] == vectors["trunc_p2r"]["OW"] + 1
    assert vectors["many_stages"]["WW"] < 30


def drive(m, x, y, ph, cpo=None):
    samples = []
    for i in range(len(x)):
        s = dict(i_xval=int(x[i]), i_yval=int(y[i]))
        if ph is not None:
            s["i_phase"] = int(ph[i])
        samples.append(s)
    if cpo:
        return vsim.run_sequential(m, samples, cpo)
    return vsim.run_pipelined(m, samples)


@pytest.mark.skipif(not os.path.isdir(REF_RTL), reason="reference not mounted")
@pytest.mark.parametrize("core,mode,cpo", [("cordic", O.P2R, None),
                                           ("topolar", O.R2P, None),
                                           ("seqcordic", O.SP2R, 17),
                                           ("seqpolar", O.SR2P, 21)])
def test_checked_in_rtl_executed_by_vsim_equals_oracle(core, mode, cpo):
    """rtl/{cordic,topolar,seqcordic,seqpolar}.v, read where they lie."""
    m = vsim.Module(open(os.path.join(REF_RTL, core + ".v")).read())
    c = O.config_cli(mode, 13, 13, 2)
    assert (m.params["IW"], m.params["WW"], m.params["PW"]) == (c.iw, c.ww, c.pw)
    rng = np.random.RandomState(5)
    n = 2500 if cpo is None else 400
    x = rng.randint(-4096, 4096, n)
    y = rng.randint(-4096, 4096, n)
    ph = rng.randint(0, 1 << c.pw, n)
    x[:4], y[:4] = [-4096, 4095, 0, -4096], [-4096, 4095, 0, 4095]
    rot = mode in (O.P2R, O.SP2R)
    res = drive(m, x, y, ph if rot else None, cpo)
    if rot:
        ox, oy = O.rotate(c, x.astype(np.int32), y.astype(np.int32),
                          ph.astype(np.uint32))
        assert [r["o_xval"] for r in res] == ox.tolist()
        assert [r["o_yval"] for r in res] == oy.tolist()
    else:
        mag, p = O.topolar(c, x.astype(np.int32), y.astype(np.int32))
        assert [r["o_mag"] for r in res] == mag.tolist()
        assert [r["o_phase"] & ((1 << c.pw) - 1) for r in res] == p.tolist()


@pytest.mark.skipif(not os.path.exists(GEN), reason="oracle/_ref not built")
def test_fresh_generator_output_executed_by_vsim_equals_oracle(tmp_path):
    """Random parameter sets: emit with the real generator, execute the text,
    compare with the oracle."""
    rng = np.random.RandomState(77)
    done = 0
    for trial in range(40):
        mode = ["p2r", "r2p", "sp2r", "sr2p"][rng.randint(4)]
        iw, ow = int(rng.randint(2, 25)), int(rng.randint(2, 25))
        xtra = int(rng.randint(0, 4))
        pw = int(rng.randint(6, 33))
        ns = int(rng.randint(3, 26))
        try:
            c = O.config_cli(MODES[mode], iw, ow, xtra, pw, ns)
        except ValueError:
            continue
        if mode == "p2r" and c.ww == c.ow + 1:
            continue        # emitted text does not elaborate (see make_rtl_vectors)
        vf = tmp_path / "core.v"
        subprocess.run([GEN, "-a", "-c", "-t", mode, "-i", str(iw), "-o",
                        str(ow), "-x", str(xtra), "-p", str(pw), "-n", str(ns),
                        "-f", str(vf)], check=True, capture_output=True)
        m = vsim.Module(vf.read_text())
        h = (tmp_path / "core.h").read_text()
        cpo = None
        if mode in ("sp2r", "sr2p"):
            cpo = int(re.search(r"CLOCKS_PER_OUTPUT\t(\d+)", h).group(1))
            assert cpo == c.clocks_per_output
        n = 250 if cpo is None else 60
        lo, hi = -(1 << (iw - 1)), 1 << (iw - 1)
        x = rng.randint(lo, hi, n)
        y = rng.randint(lo, hi, n)
        ph = rng.randint(0, 1 << pw, n, dtype=np.int64)
        x[:3], y[:3] = [lo, hi - 1, lo], [lo, hi - 1, hi - 1]
        rot = mode in ("p2r", "sp2r")
        res = drive(m, x, y, ph if rot else None, cpo)
        if rot:
            ox, oy = O.rotate(c, x.astype(np.int32), y.astype(np.int32),
                              ph.astype(np.uint32))
            assert [r["o_xval"] for r in res] == ox.tolist(), (mode, iw, ow, xtra, pw, ns)
            assert [r["o_yval"] for r in res] == oy.tolist(), (mode, iw, ow, xtra, pw, ns)
        else:
            mag, p = O.topolar(c, x.astype(np.int32), y.astype(np.int32))
            assert [r["o_mag"] for r in res] == mag.tolist(), (mode, iw, ow, xtra, pw, ns)
            assert [r["o_phase"] & ((1 << pw) - 1) for r in res] == p.tolist()
        done += 1
    assert done >= 20


@pytest.mark.gpu
def test_gpu_reproduces_every_rtl_vector(vectors):
    """The engine against the RTL-derived vectors directly (not via the
    oracle): every kernel path that these cores select."""
    import cordic_amd as ca
    from gpu_util import gpu_p2r, gpu_r2p
    for name, e in vectors.items():
        d = parse_args(e["args"])
        cfg = ca.Config.from_cli(d["mode"], d["iw"], d["ow"], d["xtra"],
                                 d["pw"], d["n"])
        x = np.array(e["x"], dtype=np.int32)
        y = np.array(e["y"], dtype=np.int32)
        if "phase" in e:
            gx, gy = gpu_p2r(cfg, x, y, np.array(e["phase"], dtype=np.uint32))
            assert gx.tolist() == e["o_xval"], name
            assert gy.tolist() == e["o_yval"], name
        else:
            gm, gp = gpu_r2p(cfg, x, y)
            assert gm.tolist() == e["o_mag"], name
            assert gp.tolist() == e["o_phase"], name


@pytest.mark.skipif(not os.path.exists(GEN), reason="oracle/_ref not built")
def test_worst_case_samples_of_the_acceptance_sweeps(tmp_path):
    """The full sweeps of tools/cordic_tb on the GPU (tests/test_acceptance.py,
    profiles/r03/acceptance/) put the 24-bit cores gencordic derives for
    itself just OUTSIDE two of the reference's own thresholds:
      p2r -i 24 -o 24 (PW 31, 27 stages): MAX err 3.1467 > 5.2 sigma = 3.0529
          at phase 0x5ffbf244 -- one phase of 2^31;
      r2p -i 24 -o 24 (PW 32, 29 stages): max phase error 12.36 > 9.31 at
          sample 643630894 of topolar_tb's circle.
    Is that the engine / oracle, or the reference's arithmetic?  The emitted
    RTL of exactly those cores, executed by vsim on exactly those samples
    (and their neighbours), gives the oracle's outputs bit for bit, and the
    error recomputed from the RTL's own outputs is the figure the sweep
    reported."""
    import quality as Q

    def emit(args):
        vf = tmp_path / "core.v"
        subprocess.run([GEN, "-a"] + args.split() + ["-f", str(vf)], check=True,
                       capture_output=True)
        return vsim.Module(vf.read_text())

    # ---- p2r, 24 bits
    m = emit("-t p2r -i 24 -o 24")
    c = O.config_cli(O.P2R, 24, 24, 2)
    assert (m.params["PW"], m.params["WW"]) == (c.pw, c.ww) == (31, 27)
    worst = 0x5ffbf244
    ph = np.array([worst - 2, worst - 1, worst, worst + 1, worst + 2], dtype=np.uint32)
    x0 = 2 ** 23 - 1
    res = drive(m, [x0] * ph.size, [0] * ph.size, ph)
    ox, oy = O.rotate(c, x0, 0, ph)
    assert [r["o_xval"] for r in res] == ox.tolist()
    assert [r["o_yval"] for r in res] == oy.tolist()
    q = Q.p2r_quality(c, ph[2:3], x0, 0, ox[2:3], oy[2:3])
    assert q["mxerr"] == pytest.approx(3.146680, abs=2e-6)
    assert q["mxerr"] > 5.2 * q["sigma"]            # the reference's threshold

    # ---- r2p, 24 bits: sample i of topolar_tb.cpp:127-141 (LGNSAMPLES = PW)
    m = emit("-t r2p -i 24 -o 24")
    c = O.config_cli(O.R2P, 24, 24, 2)
    assert (m.params["PW"], m.params["WW"], c.nstages) == (32, 32, 29)
    i = np.arange(643630894 - 2, 643630894 + 3, dtype=np.int64)
    ip = (i << 1).astype(np.int32).astype(np.float64)      # ipdata = (int)lv
    mg = float(2 ** 23 - 1)
    x = np.trunc(mg * np.cos(ip * np.pi / 2 ** 31)).astype(np.int32)
    y = np.trunc(mg * np.sin(ip * np.pi / 2 ** 31)).astype(np.int32)
    res = drive(m, x, y, None)
    mag, p = O.topolar(c, x, y)
    assert [r["o_mag"] for r in res] == mag.tolist()
    assert [r["o_phase"] & 0xffffffff for r in res] == p.tolist()
    q = Q.r2p_quality(c, x[2:3], y[2:3], int(mg), mag[2:3], p[2:3])
    assert q["mxperr"] == pytest.approx(12.36, abs=0.01)
    assert q["mxperr"] > q["phase_limit"]


# ---------------------------------------- where a rotation direction flips
#
# tests/golden/rtl_breakpoint_vectors.json (make_rtl_breakpoint_vectors.py):
# BASELINE's rotators and gencordic's own 24- / 16-bit cores, their emitted RTL
# EXECUTED by vsim on phases +/- 1 around the partial sums of the arctan table
# (the break points of the seed / direction tables), every quadrant, random
# per-sample vectors.

@pytest.fixture(scope="module")
def break_vectors():
    with open(os.path.join(ROOT, "tests", "golden",
                           "rtl_breakpoint_vectors.json")) as f:
        return json.load(f)


def test_oracle_reproduces_the_rtl_at_every_direction_break_point(break_vectors):
    total = 0
    for name, e in break_vectors.items():
        c = oracle_cfg(e["args"])
        assert (c.iw, c.ow, c.ww, c.pw) == (e["IW"], e["OW"], e["WW"], e["PW"])
        ox, oy = O.rotate(c, np.array(e["x"], dtype=np.int32),
                          np.array(e["y"], dtype=np.int32),
                          np.array(e["phase"], dtype=np.uint32))
        assert ox.tolist() == e["o_xval"], name
        assert oy.tolist() == e["o_yval"], name
        total += len(e["x"])
    assert total >= 10000


@pytest.mark.gpu
def test_table_driven_kernels_reproduce_the_rtl_at_every_break_point(break_vectors):
    """The engine against the executed RTL directly: per-sample vectors through
    the plan (directions looked up, cordic_xydir.h) and through cordic_p2r
    (phase recurrence); and, for the samples that share one vector, the
    table-seeded constant-vector kernel with its direction tails."""
    import torch
    import cordic_amd as ca
    from gpu_util import DEV, dev_i32, gpu_p2r, gpu_plan_p2r, to_np
    for name, e in break_vectors.items():
        d = parse_args(e["args"])
        cfg = ca.Config.from_cli(d["mode"], d["iw"], d["ow"], d["xtra"],
                                 d["pw"], d["n"])
        x = np.array(e["x"], dtype=np.int32)
        y = np.array(e["y"], dtype=np.int32)
        ph = np.array(e["phase"], dtype=np.uint32)
        gx, gy = gpu_p2r(cfg, x, y, ph)
        assert gx.tolist() == e["o_xval"] and gy.tolist() == e["o_yval"], name
        plan = ca.Plan(cfg)
        assert plan.dir_groups, name
        ox = torch.zeros(x.size, dtype=torch.int32, device=DEV)
        oy = torch.zeros(x.size, dtype=torch.int32, device=DEV)
        plan.p2r(dev_i32(x), dev_i32(y), dev_i32(ph), ox, oy)
        torch.cuda.synchronize()
        assert ca.last_kernel() == ca.KERNEL_DIRECTIONS
        assert to_np(ox).tolist() == e["o_xval"], name
        assert to_np(oy).tolist() == e["o_yval"], name
        # constant vector: the RTL's outputs for x[0], y[0] at EVERY phase of the
        # set come from the oracle (pinned to the RTL above); the seeded kernel
        # must agree, and at sample 0 with the RTL itself
        c = oracle_cfg(e["args"])
        rx, ry = O.rotate(c, int(x[0]), int(y[0]), ph)
        sx, sy = gpu_plan_p2r(plan, int(x[0]), int(y[0]), ph)
        assert ca.last_kernel() == ca.KERNEL_SEEDED
        assert np.array_equal(sx, rx) and np.array_equal(sy, ry), name
        assert int(sx[0]) == e["o_xval"][0] and int(sy[0]) == e["o_yval"][0]
        plan.close()
