"""Does the pin discriminate?  (CPU only.)

Sample-level parity of this project rests on three readings agreeing: the
oracle (oracle/cordic_oracle.c), tests/vsim.py executing the reference's
emitted Verilog, and the committed vectors (tests/golden/rtl_vectors.json).
Agreement only means something if a WRONG reading would have been caught, so:

 (a) RTL mutations: single-token edits of the reference's checked-in
     rtl/cordic.v, rtl/topolar.v, rtl/seqcordic.v (read where they lie, never
     copied) -- `>>>` -> `>>`, a shift off by one, `+` <-> `-` in one branch, the
     wrong sign bit, the `!` of the rounding replicate, an octant constant off
     by one LSB, a dropped sign extension ... -- each executed by vsim on the
     same samples as the unmutated text.  Every one must CHANGE vsim's
     outputs (so vsim's agreement with the oracle is not vacuous); the
     edits that provably cannot change anything (a `$signed` on an addend of a
     same-width sum, +-half-turn on a modular phase, testing a second sign bit
     of a value the fold has already bounded) must leave them EQUAL.
 (b) oracle mutations: single-token edits of oracle/cordic_oracle.c, compiled
     into a scratch library, run against the committed vectors: every one
     must FAIL to reproduce them.

Status note: none of this is a reference executor --
Verilator is absent -- so sample-level parity remains "unpinned by a
reference executor"; this shows the stand-in is sharp, nothing more.
"""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import vsim
from test_oracle_golden import parse_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_RTL = "/root/reference/rtl"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF_RTL),
                               reason="reference tree not mounted")


def nth_replace(text, old, new, nth=0):
    """replace the nth (0-based) occurrence of `old`; it must exist"""
    pos = -1
    for _ in range(nth + 1):
        pos = text.find(old, pos + 1)
        assert pos >= 0, "mutation site %r #%d not found" % (old, nth)
    return text[:pos] + new + text[pos + len(old):]


# (core, old, new, occurrence, changes_outputs)
RTL_MUTATIONS = [
    # ---- rtl/cordic.v
    ("cordic", "xv[i+1] <= xv[i] + (yv[i]>>>(i+1));",
     "xv[i+1] <= xv[i] + (yv[i]>>(i+1));", 0, True),          # logical shift
    ("cordic", "yv[i+1] <= yv[i] - (xv[i]>>>(i+1));",
     "yv[i+1] <= yv[i] - (xv[i]>>>(i));", 0, True),           # shift off by one
    ("cordic", "ph[i+1] <= ph[i] + cordic_angle[i];",
     "ph[i+1] <= ph[i] - cordic_angle[i];", 0, True),
    ("cordic", "xv[i+1] <= xv[i] - (yv[i]>>>(i+1));",
     "xv[i+1] <= xv[i] + (yv[i]>>>(i+1));", 0, True),
    ("cordic", "else if (ph[i][(PW-1)])", "else if (ph[i][(PW-4)])", 0, True),
    ("cordic", "{(WW-OW-1){!xv[NSTAGES][WW-OW]}}",
     "{(WW-OW-1){xv[NSTAGES][WW-OW]}}", 0, True),             # rounding `!`
    ("cordic", "xv[NSTAGES][(WW-OW)],", "xv[NSTAGES][(WW-OW-1)],", 0, True),
    ("cordic", "ph[0] <= i_phase - 20'h40000;",
     "ph[0] <= i_phase - 20'h40001;", 0, True),               # octant constant
    ("cordic", "xv[0] <= -e_yval;", "xv[0] <= e_yval;", 0, True),
    ("cordic", "yv[0] <= e_xval;", "yv[0] <= e_yval;", 1, True),
    ("cordic", "assign\te_xval = { {i_xval[(IW-1)]}, i_xval,",
     "assign\te_xval = { 1'b0, i_xval,", 0, True),            # no sign extension
    ("cordic", "o_xval <= pre_xval[(WW-1):(WW-OW)];",
     "o_xval <= pre_xval[(WW-2):(WW-OW-1)];", 0, True),
    ("cordic", "if ((cordic_angle[i] == 0)||(i >= WW))",
     "if ((cordic_angle[i] == 0)||(i >= 9))", 0, True),       # stops early
    ("cordic", "ph[0] <= i_phase - 20'hc0000;",
     "ph[0] <= i_phase + 20'hc0000;", 0, True),
    ("cordic", "case(i_phase[(PW-1):(PW-3)])",
     "case(i_phase[(PW-2):(PW-4)])", 0, True),                # wrong octant bits
    # equivalent mutants: must NOT change anything
    ("cordic", "assign\tpre_xval = xv[NSTAGES] + $signed({",
     "assign\tpre_xval = xv[NSTAGES] + ({", 0, False),        # same-width sum
    ("cordic", "ph[0] <= i_phase - 20'h80000;",
     "ph[0] <= i_phase + 20'h80000;", 0, False),              # half turn mod 2^PW
    # after the octant fold |ph| <= 2^(PW-3) and it only shrinks: bits PW-1
    # and PW-2 are both sign bits (found by this test: vsim did not react)
    ("cordic", "else if (ph[i][(PW-1)])", "else if (ph[i][(PW-2)])", 0, False),
    # ---- rtl/topolar.v
    ("topolar", "xv[0] <=  e_xval - e_yval;", "xv[0] <=  e_xval + e_yval;", 0, True),
    ("topolar", "ph[0] <= 21'h1c0000;", "ph[0] <= 21'h1c0001;", 0, True),
    ("topolar", "else if (yv[i][(WW-1)])", "else if (yv[i][(WW-4)])", 0, True),
    # the quadrant fold leaves |y| <= 2^(WW-3): bit WW-2 is a sign bit too
    ("topolar", "else if (yv[i][(WW-1)])", "else if (yv[i][(WW-2)])", 0, False),
    ("topolar", "xv[i+1] <= xv[i] - (yv[i]>>>(i+1));",
     "xv[i+1] <= xv[i] - (yv[i]>>(i+1));", 0, True),
    ("topolar", "ph[i+1] <= ph[i] - cordic_angle[i];",
     "ph[i+1] <= ph[i] + cordic_angle[i];", 0, True),
    ("topolar", "o_phase <= ph[NSTAGES];", "o_phase <= ph[NSTAGES-1];", 0, True),
    ("topolar", "{(WW-OW-1){!xv[NSTAGES][WW-OW]}}",
     "{(WW-OW-1){xv[NSTAGES][WW-OW]}}", 0, True),
    ("topolar", "yv[0] <= -e_xval + e_yval;", "yv[0] <=  e_xval + e_yval;", 0, True),
    ("topolar", "yv[i+1] <= yv[i] + (xv[i]>>>(i+1));",
     "yv[i+1] <= yv[i] + (xv[i]>>>(i+2));", 0, True),
    ("topolar", "case({i_xval[IW-1], i_yval[IW-1]})",
     "case({i_yval[IW-1], i_xval[IW-1]})", 0, True),
]


def _samples(iw, pw, n, rot, seed):
    rng = np.random.RandomState(seed)
    lo, hi = -(1 << (iw - 1)), 1 << (iw - 1)
    x = rng.randint(lo, hi, n)
    y = rng.randint(lo, hi, n)
    ext = [lo, hi - 1, 0, -1, 1, lo + 1]
    k = 0
    for a in ext:
        for b in ext:
            x[k], y[k] = a, b
            k += 1
    ph = rng.randint(0, 1 << pw, n, dtype=np.int64) if rot else None
    if rot:
        q = 1 << (pw - 3)
        for j, e in enumerate([(j * q + d) % (1 << pw)
                               for j in range(9) for d in (-1, 0, 1)]):
            ph[40 + j] = e
    return x, y, ph


def _run(text, x, y, ph, cpo=None):
    m = vsim.Module(text)
    samples = []
    for i in range(len(x)):
        s = dict(i_xval=int(x[i]), i_yval=int(y[i]))
        if ph is not None:
            s["i_phase"] = int(ph[i])
        samples.append(s)
    res = (vsim.run_sequential(m, samples, cpo) if cpo
           else vsim.run_pipelined(m, samples))
    keys = sorted(k for k in res[0])
    return [tuple(r[k] for k in keys) for r in res]


@needs_ref
def test_rtl_mutants_are_told_apart():
    texts, base, inputs = {}, {}, {}
    for core, rot in (("cordic", True), ("topolar", False)):
        texts[core] = open(os.path.join(REF_RTL, core + ".v")).read()
        m = vsim.Module(texts[core])
        inputs[core] = _samples(m.params["IW"], m.params["PW"], 260, rot, 9)
        base[core] = _run(texts[core], *inputs[core])
    # the unmutated text is what the oracle computes
    c = O.config_cli(O.P2R, 13, 13, 2)
    x, y, ph = inputs["cordic"]
    ox, oy = O.rotate(c, x.astype(np.int32), y.astype(np.int32),
                      ph.astype(np.uint32))
    assert base["cordic"] == list(zip(ox.tolist(), oy.tolist()))
    killed = equal = 0
    for core, old, new, nth, changes in RTL_MUTATIONS:
        mutant = nth_replace(texts[core], old, new, nth)
        assert mutant != texts[core]
        got = _run(mutant, *inputs[core])
        if changes:
            assert got != base[core], "vsim blind to: %s -> %s" % (old, new)
            killed += 1
        else:
            assert got == base[core], "not equivalent after all: %s" % old
            equal += 1
    assert killed >= 24 and equal == 4


@needs_ref
def test_sequential_rtl_mutants_are_told_apart():
    """rtl/seqcordic.v: the arithmetic the closed form NSTAGES-2 rests on."""
    text = open(os.path.join(REF_RTL, "seqcordic.v")).read()
    m = vsim.Module(text)
    x, y, ph = _samples(m.params["IW"], m.params["PW"], 90, True, 10)
    base = _run(text, x, y, ph, cpo=17)
    c = O.config_cli(O.SP2R, 13, 13, 2)
    ox, oy = O.rotate(c, x.astype(np.int32), y.astype(np.int32),
                      ph.astype(np.uint32))
    assert base == list(zip(ox.tolist(), oy.tolist()))
    sites = [s for s in (
        ("xv <= xv + (yv >>> state);", "xv <= xv + (yv >> state);"),
        ("yv <= yv - (xv >>> state);", "yv <= yv + (xv >>> state);"),
        ("ph <= ph + cangle;", "ph <= ph - cangle;"),
    ) if s[0] in text]
    assert sites, "seqcordic.v: no known mutation site found"
    for old, new in sites:
        got = _run(nth_replace(text, old, new), x, y, ph, cpo=17)
        assert got != base, "vsim blind to: %s -> %s" % (old, new)


# ------------------------------------------------------------ oracle mutants

ORACLE_MUTATIONS = [
    ("*x = sx(xo + asr(yo, shift), ww);", "*x = sx(xo - asr(yo, shift), ww);", 0),
    ("*p = (*p + ang) & pm;", "*p = (*p - ang) & pm;", 0),       # p2r_rotate
    ("*p = (*p + ang) & pm;", "*p = (*p - ang) & pm;", 1),       # r2p_rotate
    ("p2r_rotate(c->ww, pm, c->pw, (unsigned)i + 1,",
     "p2r_rotate(c->ww, pm, c->pw, (unsigned)i,", 0),
    ("if ((*p >> (pw - 1)) & 1) {", "if ((*p >> (pw - 4)) & 1) {", 0),
    ("| (b ? 0 : (((int64_t)1 << (r - 1)) - 1));",
     "| (b ? (((int64_t)1 << (r - 1)) - 1) : 0);", 0),           # rounding `!`
    ("*x = sx(-ey, ww); *y = ex; *p = (ph - q) & pm;",
     "*x = sx(-ey, ww); *y = ey; *p = (ph - q) & pm;", 0),
    ("*p = (ph - 3 * q) & pm;", "*p = (ph - 2 * q) & pm;", 0),
    ("ex = sx((int64_t)((uint64_t)sx(ix, c->iw) << (ww - c->iw - 1)), ww);",
     "ex = sx((int64_t)((uint64_t)sx(ix, c->iw) << (ww - c->iw - 2)), ww);", 0),
    ("if (yo < 0) {", "if (yo <= 0) {", 0),                     # tie at y == 0
    ("*y = sx(ex + ey, ww); *p = 7 * e;", "*y = sx(ex + ey, ww); *p = 6 * e;", 0),
    ("if ((c->angle[i] == 0) || (i >= c->ww))",
     "if ((c->angle[i] == 0) || (i > c->ww))", 0),               # skip rule
    ("for (int i = 0; i < c->nstages - 2; i++)",
     "for (int i = 0; i < c->nstages - 1; i++)", 0),             # seqcordic count
    # (a LOGICAL shift inside asr() is an equivalent mutant: values are kept
    # sign extended in 64 bits and wrapped to WW bits after every operation,
    # so the bits a logical shift would zero never reach a result)
    ("return (int32_t)sx(v >> r, ow);", "return (int32_t)sx(v >> (r - 1), ow);", 0),
    ("int64_t b = (v >> r) & 1;", "int64_t b = (v >> (r - 1)) & 1;", 0),
]


def _reproduces(lib, vectors):
    """names of the committed cores this library fails to reproduce"""
    i32p, u32p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    cfgp = C.POINTER(O.OrcConfig)
    lib.orc_config_cli.argtypes = [cfgp] + [C.c_int] * 6
    lib.orc_rotate.argtypes = [cfgp, C.c_size_t, i32p, i32p, C.c_int, u32p,
                               i32p, i32p]
    lib.orc_rotate.restype = None
    lib.orc_topolar.argtypes = [cfgp, C.c_size_t, i32p, i32p, i32p, u32p]
    lib.orc_topolar.restype = None
    bad = []
    for name, e in vectors.items():
        d = parse_args(e["args"])
        cfg = O.OrcConfig()
        assert lib.orc_config_cli(C.byref(cfg), d["mode"], d["iw"], d["ow"],
                                  d["xtra"], d["pw"], d["n"]) == 0
        x = np.array(e["x"], dtype=np.int32)
        y = np.array(e["y"], dtype=np.int32)
        n = x.size
        a = np.empty(n, dtype=np.int32)
        if "phase" in e:
            ph = np.array(e["phase"], dtype=np.uint32)
            b = np.empty(n, dtype=np.int32)
            lib.orc_rotate(C.byref(cfg), n, O._i32(x), O._i32(y), 1, O._u32(ph),
                           O._i32(a), O._i32(b))
            same = a.tolist() == e["o_xval"] and b.tolist() == e["o_yval"]
        else:
            b = np.empty(n, dtype=np.uint32)
            lib.orc_topolar(C.byref(cfg), n, O._i32(x), O._i32(y), O._i32(a),
                            O._u32(b))
            same = a.tolist() == e["o_mag"] and b.tolist() == e["o_phase"]
        if not same:
            bad.append(name)
    return bad


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_mutants_fail_the_committed_vectors(tmp_path):
    with open(os.path.join(ROOT, "tests", "golden", "rtl_vectors.json")) as f:
        vectors = json.load(f)
    src = open(os.path.join(O.ORACLE_DIR, "cordic_oracle.c")).read()

    def build(text, tag):
        c = tmp_path / (tag + ".c")
        so = tmp_path / (tag + ".so")
        c.write_text(text)
        subprocess.run(["gcc", "-O1", "-fPIC", "-shared", "-fno-strict-aliasing",
                        "-ffp-contract=off", "-I", O.ORACLE_DIR, "-o", str(so),
                        str(c), "-lm", "-lpthread"], check=True,
                       capture_output=True)
        return C.CDLL(str(so))

    assert _reproduces(build(src, "pristine"), vectors) == []
    for k, (old, new, nth) in enumerate(ORACLE_MUTATIONS):
        mutant = nth_replace(src, old, new, nth)
        bad = _reproduces(build(mutant, "m%d" % k), vectors)
        assert bad, "the vectors do not notice: %s -> %s" % (old, new)
    assert len(ORACLE_MUTATIONS) >= 10
