"""The oracle against everything the reference offers to pin it (no GPU).

1. Table / parameter math against outputs of the REAL reference generator
   (tests/golden/gencordic_golden.json, made by tests/golden/make_golden.py
   from oracle/_ref/gencordic) and, live, against that binary when present.
2. The reference's own pass criteria (bench/cpp/cordic_tb.cpp:285-337,
   bench/cpp/topolar_tb.cpp:303-315) on the oracle's output at the checked-in
   configuration (exhaustive 2^PW sweep).
3. The clock-by-clock models of rtl/seqcordic.v / rtl/seqpolar.v against the
   closed forms, including the CLOCKS_PER_OUTPUT handshake count.
"""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import quality as Q

MODES = {"p2r": O.P2R, "r2p": O.R2P, "sp2r": O.SP2R, "sr2p": O.SR2P}


def parse_args(args):
    a = args.split()
    d = dict(mode=None, iw=-1, ow=-1, xtra=2, pw=-1, n=-1)
    i = 0
    while i < len(a):
        if a[i] == "-t":
            d["mode"] = MODES[a[i + 1]]; i += 2
        elif a[i] in ("-i", "-o", "-x", "-p", "-n"):
            d[{"-i": "iw", "-o": "ow", "-x": "xtra", "-p": "pw",
               "-n": "n"}[a[i]]] = int(a[i + 1]); i += 2
        else:
            i += 1
    return d


def header_consts(text):
    out = {}
    for k, v in re.findall(r"const \w+\s+(\w+)\s*=\s*([^;]+);", text):
        out[k] = v.strip()
    m = re.search(r"#define\tCLOCKS_PER_OUTPUT\t(\d+)", text)
    if m:
        out["CLOCKS_PER_OUTPUT"] = m.group(1)
    return out


def oracle_for(entry):
    d = parse_args(entry["args"])
    return O.config_cli(d["mode"], d["iw"], d["ow"], d["xtra"], d["pw"],
                        d["n"]), d


def test_golden_file_is_substantial(golden):
    assert len(golden) >= 150
    assert not any(v.get("failed") for v in golden.values())


def test_oracle_config_matches_reference_generator(golden):
    checked = 0
    for name, e in golden.items():
        d = parse_args(e["args"])
        try:
            cfg, d = oracle_for(e)
        except ValueError:
            # only the documented unsupported families may be refused
            assert e["PW"] > 32 or e["WW"] > 64 or d["mode"] in (
                O.SP2R, O.SR2P), (name, e["args"])
            continue
        assert (cfg.iw, cfg.ow, cfg.ww, cfg.pw, cfg.nstages, cfg.nxtra) == (
            e["IW"], e["OW"], e["WW"], e["PW"], e["NSTAGES"], e["XTRA"]), name
        hc = header_consts(e["header"])
        assert int(hc["NEXTRA"]) == cfg.nxtra
        # the pipelined cores list NSTAGES angles, the sequential ones a
        # power-of-two table; the first NSTAGES entries must agree
        ang = e["angles"][: cfg.nstages]
        assert list(cfg.angle[: cfg.nstages]) == ang, name
        # constants as the reference printed them
        assert "%.16f" % cfg.gain == hc["GAIN"], name
        if d["mode"] in (O.P2R, O.SP2R):
            assert "%.4e" % cfg.quantization_variance == \
                hc["QUANTIZATION_VARIANCE"], name
            assert "%.4e" % cfg.phase_variance_rad == \
                hc["PHASE_VARIANCE_RAD"], name
            assert "%.2f" % cfg.best_possible_cnr == \
                hc["BEST_POSSIBLE_CNR"], name
        else:
            assert "%.16f" % cfg.quantization_variance == \
                hc["QUANTIZATION_VARIANCE"], name
            assert "%.16f" % cfg.phase_variance_rad == \
                hc["PHASE_VARIANCE_RAD"], name
        if "CLOCKS_PER_OUTPUT" in hc:
            assert int(hc["CLOCKS_PER_OUTPUT"]) == cfg.clocks_per_output
        checked += 1
    assert checked >= 140


def test_prerotation_constants_match_emitted_verilog(golden):
    """rtl/cordic.v:131-188 subtracts k*2^(PW-2); rtl/topolar.v:122-152
    loads {7,3,5,1}*2^(PW-3): the oracle hard-codes those, the generator
    prints them."""
    for name in ("rtl_cordic", "cfg2", "cfg1"):
        e = golden[name]
        q = 1 << (e["PW"] - 2)
        assert e["prerot_consts"] == [q, q, 2 * q, 2 * q, 3 * q, 3 * q]
    for name in ("rtl_topolar", "cfg3"):
        e = golden[name]
        u = 1 << (e["PW"] - 3)
        assert e["prerot_consts"] == [7 * u, 3 * u, 5 * u, 1 * u]


def test_checked_in_rtl_values():
    """rtl/cordic.h:46-55, rtl/topolar.h:46-54, rtl/cordic.v:204-219."""
    c = O.config_cli(O.P2R, 13, 13, 2)
    assert (c.ww, c.pw, c.nstages, c.nxtra) == (16, 20, 16, 3)
    assert list(c.angle[:16]) == [
        0x12e40, 0x09fb3, 0x05111, 0x028b0, 0x0145d, 0x00a2f, 0x00517,
        0x0028b, 0x00145, 0x000a2, 0x00051, 0x00028, 0x00014, 0x0000a,
        0x00005, 0x00002]
    assert "%.16f" % c.gain == "1.1644353454607288"
    assert "%.2f" % c.best_possible_cnr == "78.92"
    t = O.config_cli(O.R2P, 13, 13, 2)
    assert (t.ww, t.pw, t.nstages, t.nxtra) == (21, 21, 18, 4)
    assert "%.16f" % t.gain == "0.8233801290585359"
    assert "%.16f" % t.quantization_variance == "0.1964179315931617"


def test_baseline_configs_derivation():
    """the five BASELINE configs."""
    c1 = O.config_cli(O.P2R, 16, 16, 2, 16, 16)
    assert (c1.ww, c1.pw) == (19, 16)
    assert list(c1.angle[13:16]) == [0, 0, 0] and c1.angle[12] != 0
    c2 = O.config_cli(O.P2R, 32, 32, 2, 32, 16)
    assert c2.ww == 35 and c2.angle[0] == 0x12e4051d
    c3 = O.config_cli(O.R2P, 24, 24, 2, -1, 20)
    assert (c3.nxtra, c3.ww, c3.pw) == (4, 32, 32)
    c4 = O.config_cli(O.P2R, 32, 32, 2, 32, 24)
    assert c4.ww == 35 and c4.angle[23] == 0x28
    c5 = O.config_cli(O.SP2R, 32, 32, 2, 32, 16)
    assert c5.ww == 35 and c5.clocks_per_output == 17


@pytest.mark.skipif(not os.path.exists(os.path.join(
    O.ORACLE_DIR, "_ref", "gencordic")), reason="oracle/_ref not built")
def test_live_against_reference_generator(tmp_path):
    """Run the real generator on fresh parameter sets (not in the fixture)."""
    gen = os.path.join(O.ORACLE_DIR, "_ref", "gencordic")
    rng = np.random.RandomState(7)
    for _ in range(40):
        mode = ["p2r", "r2p", "sp2r", "sr2p"][rng.randint(4)]
        iw, ow = int(rng.randint(4, 29)), int(rng.randint(4, 29))
        xtra = int(rng.randint(0, 5))
        args = ["-t", mode, "-i", str(iw), "-o", str(ow), "-x", str(xtra)]
        pw = ns = -1
        if rng.rand() < 0.5:
            pw = int(rng.randint(8, 33)); args += ["-p", str(pw)]
        if rng.rand() < 0.5:
            ns = int(rng.randint(4, 33)); args += ["-n", str(ns)]
        vf = tmp_path / "core.v"
        subprocess.run([gen] + args + ["-c", "-f", str(vf)], check=True,
                       capture_output=True)
        v = vf.read_text()
        h = (tmp_path / "core.h").read_text()
        try:
            cfg = O.config_cli(MODES[mode], iw, ow, xtra, pw, ns)
        except ValueError:
            continue
        got = {k: int(re.search(r"\b%s=\s*(\d+)" % k, v).group(1))
               for k in ("IW", "OW", "NSTAGES", "WW", "PW")}
        assert got == dict(IW=cfg.iw, OW=cfg.ow, NSTAGES=cfg.nstages,
                           WW=cfg.ww, PW=cfg.pw), args
        ang = [int(hx.replace("_", ""), 16) for hx in re.findall(
            r"cordic_angle\[\s*\d+\]\s*=\s*\d+'h([0-9a-f_]+);", v)]
        assert ang[: cfg.nstages] == list(cfg.angle[: cfg.nstages]), args
        assert "%.16f" % cfg.gain == header_consts(h)["GAIN"]


# ------------------------------------------------------------------ quality

def test_p2r_passes_reference_bench_criteria():
    c = O.config_cli(O.P2R, 13, 13, 2)
    ph, x0, y0 = Q.p2r_bench_inputs(c.iw, c.pw)
    ox, oy = O.rotate(c, x0, y0, ph)
    q = Q.p2r_quality(c, ph, x0, y0, ox, oy)
    assert q["ok"], q
    # the oracle sits where an independent restatement did
    assert abs(q["averr"] - 0.5583) < 1e-3 and abs(q["mxerr"] - 1.9247) < 1e-3
    assert abs(q["cnr"] - 78.63) < 0.02
    assert abs(Q.sfdr_dbc(ox, oy) - 93.88) < 0.05
    # phase 0, x = 4095: ~ 4095 * 1.16444 / 2
    assert ox[0] == 2385 and oy[0] == 0


def test_seq_p2r_passes_reference_bench_criteria():
    c = O.config_cli(O.SP2R, 13, 13, 2)
    ph, x0, y0 = Q.p2r_bench_inputs(c.iw, c.pw)
    ox, oy = O.rotate(c, x0, y0, ph)
    assert Q.p2r_quality(c, ph, x0, y0, ox, oy)["ok"]


@pytest.mark.parametrize("mode", [O.R2P, O.SR2P])
def test_r2p_passes_reference_bench_criteria(mode):
    c = O.config_cli(mode, 13, 13, 2)
    x, y, mg = Q.r2p_bench_inputs(c.iw, c.pw)
    m, p = O.topolar(c, x, y)
    q = Q.r2p_quality(c, x, y, mg, m, p)
    assert q["ok"], q
    assert abs(q["mxverr"] - 0.8708) < 1e-3     # the tight margin (0.8864)


def test_wrong_rounding_would_fail_the_reference_criteria():
    """The r2p magnitude margin is tight enough to reject round-half-up."""
    c = O.config_cli(O.R2P, 13, 13, 2)
    x, y, mg = Q.r2p_bench_inputs(c.iw, c.pw)
    m, p = O.topolar(c, x, y)
    q = Q.r2p_quality(c, x, y, mg, m + 1, p)    # off-by-one magnitude
    assert not q["ok"]


# ------------------------------------------------- sequential cycle models

def test_seq_p2r_cycle_model_equals_closed_form():
    rng = np.random.RandomState(3)
    for (iw, ow, xtra, pw, ns) in [(13, 13, 2, -1, -1), (16, 16, 2, 24, 20),
                                   (12, 10, 3, 18, 9), (32, 32, 2, 32, 16),
                                   (8, 8, 2, 12, 14)]:
        c = O.config_cli(O.SP2R, iw, ow, xtra, pw, ns)
        n = 300
        x = rng.randint(-(1 << (iw - 1)), 1 << (iw - 1), n).astype(np.int32)
        y = rng.randint(-(1 << (iw - 1)), 1 << (iw - 1), n).astype(np.int32)
        ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        ph &= np.uint32((1 << c.pw) - 1 if c.pw < 32 else 0xffffffff)
        ox, oy = O.rotate(c, x, y, ph)
        for i in range(n):
            t, cx, cy = O.seq_p2r_cycle(c, int(x[i]), int(y[i]), int(ph[i]))
            assert t == c.clocks_per_output == c.nstages + 1
            assert (cx, cy) == (int(ox[i]), int(oy[i]))


def test_seq_r2p_cycle_model_equals_closed_form():
    rng = np.random.RandomState(4)
    for (iw, ow, xtra, pw, ns) in [(13, 13, 2, -1, -1), (16, 16, 2, 24, 20),
                                   (12, 10, 3, 18, 9), (24, 24, 2, -1, 20),
                                   (8, 8, 2, 12, 14)]:
        c = O.config_cli(O.SR2P, iw, ow, xtra, pw, ns)
        n = 300
        x = rng.randint(-(1 << (iw - 1)), 1 << (iw - 1), n).astype(np.int32)
        y = rng.randint(-(1 << (iw - 1)), 1 << (iw - 1), n).astype(np.int32)
        m, p = O.topolar(c, x, y)
        for i in range(n):
            t, cm, cp = O.seq_r2p_cycle(c, int(x[i]), int(y[i]))
            assert t == c.clocks_per_output == c.nstages + 3
            assert (cm, cp) == (int(m[i]), int(p[i]))


def test_seq_differs_from_pipelined_as_surveyed():
    """seqcordic = NSTAGES-2 rotations."""
    c = O.config_cli(O.P2R, 13, 13, 2)
    cs = O.config_cli(O.SP2R, 13, 13, 2)
    c14 = O.config_cli(O.P2R, 13, 13, 2, 20, 14)
    ph = np.arange(0, 1 << 20, 37, dtype=np.uint32)
    a = O.rotate(c, 4095, 0, ph)
    s = O.rotate(cs, 4095, 0, ph)
    b = O.rotate(c14, 4095, 0, ph)
    assert np.array_equal(s[0], b[0]) and np.array_equal(s[1], b[1])
    assert not (np.array_equal(s[0], a[0]) and np.array_equal(s[1], a[1]))


def test_unsupported_parameter_sets_are_refused():
    with pytest.raises(ValueError):
        O.config_cli(O.P2R, 32, 32)            # default PW would be 39
    with pytest.raises(ValueError):
        O.config_cli(O.SR2P, 13, 13, 2, 20, 15)  # o_done never rises
    with pytest.raises(ValueError):
        O.config_core(O.SP2R, 16, 13, 13, 1, 20)  # emits an i_ce-less core
    with pytest.raises(ValueError):
        O.config_cli(7, 13, 13)
