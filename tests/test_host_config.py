"""Host side of the product (cordic_config.cpp through the C ABI), no GPU:
parameter derivation, arctan table, constants-header text and argv parsing
against the real reference generator's output (tests/golden/), plus the ABI
surface itself."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import cordic_amd as ca
from cordic_amd import _native
import oracle_lib as O
from test_oracle_golden import MODES, header_consts, parse_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Every function include/cordic_amd.h declares must be exported by
    libcordic_amd.so and bound in _native.ABI (and nothing else is bound)."""
    hdr = open(os.path.join(ROOT, "include", "cordic_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cordic_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert declared == set(_native.ABI), declared ^ set(_native.ABI)
    out = subprocess.run(["nm", "-D", "--defined-only", ca.lib_path()],
                         capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (cordic_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert ca.lib().cordic_abi_version() == 1


def test_config_struct_layout_matches_header():
    """The ctypes mirror and the C struct must agree field by field."""
    hdr = open(os.path.join(ROOT, "include", "cordic_amd.h")).read()
    body = re.search(r"typedef struct cordic_config \{(.*?)\} cordic_config;",
                     hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(?:int32_t|uint32_t|double)\s+(\w+)", body)
    assert names == [f[0] for f in _native._CConfig._fields_]
    assert C.sizeof(_native._CConfig) == 8 * 4 + 4 * 8 + 6 * 4 + 64 * 4


def test_config_matches_reference_generator(golden):
    checked = 0
    for name, e in golden.items():
        d = parse_args(e["args"])
        try:
            cfg = ca.Config.from_cli(d["mode"], d["iw"], d["ow"], d["xtra"],
                                     d["pw"], d["n"])
        except ca.CordicError as err:
            assert err.status in (-3, -4, -6), (name, err)
            # the oracle refuses exactly the same sets
            with pytest.raises(ValueError):
                O.config_cli(d["mode"], d["iw"], d["ow"], d["xtra"], d["pw"],
                             d["n"])
            continue
        assert (cfg.iw, cfg.ow, cfg.ww, cfg.pw, cfg.nstages, cfg.nextra) == (
            e["IW"], e["OW"], e["WW"], e["PW"], e["NSTAGES"], e["XTRA"]), name
        assert cfg.angles == e["angles"][: cfg.nstages], name
        # the multiplier of the "annihilate this gain" comment
        # (sw/cordiclib.cpp:205-209), as printed into this very core
        assert ca.lib().cordic_config_gain_annihilator(cfg.ref) == \
            e["annihilate"], name
        checked += 1
    assert checked >= 140
    assert ca.lib().cordic_gain_annihilator(16) == 0xdbd95b16   # rtl/cordic.v:224


def test_header_text_is_byte_identical(golden):
    """cordic_config_write_header vs the text the reference wrote with -c
    (licence banner excluded)."""
    checked = 0
    for name, e in golden.items():
        args = e["args"].replace(" -c", "")
        try:
            cfg = ca.Config.from_args(args)
        except ca.CordicError:
            continue
        assert cfg.header_text("core") == e["header"], name
        checked += 1
    assert checked >= 140


def test_from_args_flags_and_defaults():
    c = ca.Config.from_args("-vca -f ../rtl/cordic.v -v -i 13 -o 13 -t p2r "
                            "-x 2 -c")
    assert (c.mode, c.ww, c.pw, c.nstages) == (ca.P2R, 16, 20, 16)
    assert c.fname == "../rtl/cordic.v" and c.c_header and c.has_aux
    assert c.has_reset and not c.async_reset
    # glued values, -R, -A
    c = ca.Config.from_args(["-tsp2r", "-i13", "-o13", "-R"])
    assert c.mode == ca.SP2R and not c.has_reset
    assert c.fname == "seqcordic.v" and not c.c_header
    c = ca.Config.from_args("-t r2p -A")
    assert c.async_reset and c.has_reset and (c.iw, c.ow) == (24, 24)
    # no -t at all: the reference defaults to r2p (sw/main.cpp:100)
    assert ca.Config.from_args("-i 12").mode == ca.R2P
    for bad, status in (("-t tbl", -1), ("-t nope", -1), ("-q", -7),
                        ("-i", -7)):
        with pytest.raises(ca.CordicError) as ei:
            ca.Config.from_args(bad)
        assert ei.value.status == status


def test_getopt_semantics_match_the_generator():
    """Repeated -t (sticky `sequential`, the FIRST -t names the default
    file), words that are not options, "--", -h: cordic_config_from_args
    against what the real gencordic did with the same command lines
    (tests/golden/getopt_golden.json, made by make_getopt_golden.py)."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden",
                           "getopt_golden.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 15
    for e in cases:
        core = e.get("core")
        if core is None:            # the generator wrote no core (-h, errors)
            with pytest.raises(ca.CordicError):
                ca.Config.from_args(e["args"])
            continue
        c = ca.Config.from_args(e["args"])
        want_mode = {("p2r", False): ca.P2R, ("p2r", True): ca.SP2R,
                     ("r2p", False): ca.R2P, ("r2p", True): ca.SR2P}[
                         (core["kind"], core["sequential"])]
        assert c.mode == want_mode, e["args"]
        vfile = [f for f in e["files"] if f.endswith(".v")][0]
        assert c.fname == vfile, e["args"]
        assert (c.iw, c.ow, c.ww, c.pw, c.nstages) == (
            core["IW"], core["OW"], core["WW"], core["PW"],
            core["NSTAGES"]), e["args"]
        assert bool(c.has_reset) == core["has_reset"], e["args"]
        assert bool(c.async_reset) == core["async_reset"], e["args"]
        assert bool(c.has_aux) == core["has_aux"], e["args"]
        assert c.c_header == any(f.endswith(".h") for f in e["files"]), e["args"]


def test_core_level_constructor_equals_cli_level():
    """sw/main.cpp hands the emitters nxtra = xtra+1 (p2r) / xtra+2 (r2p)."""
    a = ca.Config.from_cli(ca.P2R, 13, 13, 2)
    b = ca.Config.from_core(ca.P2R, a.nstages, 13, 13, 3, a.pw)
    assert bytes(a.c) == bytes(b.c)
    a = ca.Config.from_cli(ca.R2P, 24, 24, 2, -1, 20)
    b = ca.Config.from_core(ca.R2P, 20, 24, 24, 4, a.pw)
    assert bytes(a.c) == bytes(b.c)
    # the emitters clamp nxtra (sw/basiccordic.cpp:67, sw/topolar.cpp:67)
    assert ca.Config.from_core(ca.P2R, 12, 10, 10, 0, 16).ww == 11
    assert ca.Config.from_core(ca.R2P, 12, 10, 10, 0, 16).ww == 14


def test_error_codes():
    L = ca.lib()
    cfg = _native._CConfig()
    r = C.byref(cfg)
    assert L.cordic_config_init_core(r, 9, 16, 13, 13, 3, 20) == -1
    assert L.cordic_config_init_core(r, 0, 16, 0, 13, 3, 20) == -2
    assert L.cordic_config_init_core(r, 0, 16, 33, 13, 3, 20) == -2
    assert L.cordic_config_init_core(r, 0, 16, 13, 13, 3, 2) == -3
    assert L.cordic_config_init_core(r, 0, 16, 13, 13, 3, 33) == -3
    assert L.cordic_config_init_core(r, 0, 16, 32, 32, 40, 20) == -4
    assert L.cordic_config_init_core(r, 0, 0, 13, 13, 3, 20) == -5
    assert L.cordic_config_init_core(r, 0, 65, 13, 13, 3, 20) == -5
    assert L.cordic_config_init_core(r, 3, 15, 13, 13, 4, 20) == -6
    assert L.cordic_config_init_core(None, 0, 16, 13, 13, 3, 20) == -7
    assert L.cordic_strerror(-3).decode().startswith("phase bits")
    # device entry points validate before touching HIP
    good = ca.Config.from_cli(ca.P2R, 13, 13, 2)
    assert L.cordic_p2r_const(good.ref, 0, 1, 0, None, None, None, None) == 0
    assert L.cordic_p2r_const(good.ref, 8, 1, 0, None, None, None, None) == -7
    assert L.cordic_r2p(good.ref, 8, None, None, None, None, None) == -1


def test_cordiclib_functions_exported():
    """sw/cordiclib.h:45-52 equivalents."""
    L = ca.lib()
    assert [L.cordic_nextlg(v) for v in (1, 3, 4, 5, 8, 9)] == [0, 2, 2, 3, 3, 4]
    assert "%.16f" % L.cordic_gain(16) == "1.1644353454607288"
    assert L.cordic_calc_phase_bits(16) == 20     # WW=16 -> PW=20 (rtl/cordic.h)
    assert L.cordic_calc_stages_ww(16, 20) == 16
    assert L.cordic_calc_stages(21) == 18         # rtl/topolar.h
    out = (C.c_uint32 * 4)()
    assert L.cordic_angles(4, 20, out) == 0
    assert list(out) == [0x12e40, 0x09fb3, 0x05111, 0x028b0]
    o = O.lib()
    for ns, pw in ((16, 20), (18, 21), (24, 32)):
        assert L.cordic_phase_variance(ns, pw) == o.orc_phase_variance(ns, pw)
    assert L.cordic_transform_quantization_variance(16, 3, 3) == \
        o.orc_transform_quantization_variance(16, 3, 3)


def test_live_stage_counts_and_wrap_analysis():
    c1 = ca.Config.from_cli(ca.P2R, 16, 16, 2, 16, 16)
    assert c1.nlive == 13                       # angles 13..15 are zero
    assert ca.Config.from_cli(ca.SP2R, 32, 32, 2, 32, 16).nlive == 14
    assert ca.Config.from_cli(ca.SR2P, 13, 13, 2).nlive == 18
    # i >= WW cut-off: WW = 9, 30 stages requested, PW = 32
    c = ca.Config.from_cli(ca.P2R, 6, 6, 2, 32, 30)
    assert c.ww == 9 and c.nlive == 9
    # realistic cores never overflow their WW-bit registers ...
    for cfg in (c1, ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 24),
                ca.Config.from_cli(ca.R2P, 24, 24, 2, -1, 20)):
        assert cfg.needs_wrap == 0
    # ... tiny ones can (truncation noise is comparable to full scale)
    assert ca.Config.from_core(ca.P2R, 6, 1, 1, 1, 8).needs_wrap == 1


def test_device_entry_points_fail_loudly_without_a_gpu():
    """No CPU fallback: on a machine without a GPU every entry point that
    needs the device returns CORDIC_ERR_DEVICE instead of computing anything
    on the host.  (Skipped where a GPU is present.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = ca.Config.from_cli(ca.P2R, 13, 13)
    with pytest.raises(ca.CordicError) as e:
        ca.Plan(cfg)
    assert e.value.status == ca.ERR_DEVICE
    with pytest.raises(ca.CordicError) as e:
        ca.Stream(cfg)
    assert e.value.status == ca.ERR_DEVICE
    with pytest.raises(ca.CordicError) as e:
        ca.Seq(ca.Config.from_cli(ca.SP2R, 13, 13))
    assert e.value.status == ca.ERR_DEVICE
    with pytest.raises(ca.CordicError) as e:
        ca.Quad(ow=13, pw=18)
    assert e.value.status == ca.ERR_DEVICE
    with pytest.raises(ca.CordicError) as e:
        ca.Table(ca.TBL, -1, 13, 17)
    assert e.value.status == ca.ERR_DEVICE
    x = np.zeros(8, dtype=np.int32)
    with pytest.raises(ca.CordicError) as e:
        ca.p2r_host(cfg, x, x, x.view(np.uint32))
    assert e.value.status == ca.ERR_DEVICE
    # a launch on (null) device pointers must not pretend to succeed either
    rc = ca.lib().cordic_p2r_const(cfg.ref, 8, 1, 0, 16, 16, 16, None)
    assert rc == ca.ERR_DEVICE


def test_product_and_oracle_accept_and_refuse_the_same_parameters():
    """Random parameter sets, valid and not: the host layer and the oracle
    must agree on which ones exist, and on what they derive from those."""
    rng = np.random.RandomState(77)
    accepted = 0
    for _ in range(2500):
        mode = int(rng.randint(-1, 6))
        iw, ow = int(rng.randint(-3, 40)), int(rng.randint(-3, 40))
        xtra, pw = int(rng.randint(-2, 40)), int(rng.randint(-2, 40))
        ns = int(rng.randint(-2, 70))
        try:
            c = ca.Config.from_cli(mode, iw, ow, xtra, pw, ns)
        except ca.CordicError:
            with pytest.raises(ValueError):
                O.config_cli(mode, iw, ow, xtra, pw, ns)
        else:
            o = O.config_cli(mode, iw, ow, xtra, pw, ns)
            assert (c.iw, c.ow, c.ww, c.pw, c.nstages) == (o.iw, o.ow, o.ww,
                                                            o.pw, o.nstages)
            assert c.angles == list(o.angle[: o.nstages])
            accepted += 1
        try:
            q = ca.Quad(iw, ow, xtra, pw, device=False)
        except ca.CordicError:
            with pytest.raises(ValueError):
                O.quad_cli(iw, ow, xtra, pw)
        else:
            oq = O.quad_cli(iw, ow, xtra, pw)
            assert (q.pw, q.lgtbl, q.cbits, q.lbits, q.qbits) == (
                oq.pw, oq.lgtbl, oq.cbits, oq.lbits, oq.qbits)
        for kind in (ca.TBL, ca.QTR):
            try:
                t = ca.Table(kind, iw, ow, pw, device=False)
            except ca.CordicError:
                with pytest.raises(ValueError):
                    O.table_config(kind, iw, ow, pw)
            else:
                assert (t.pw, t.ow) == O.table_config(kind, iw, ow, pw)
        # the exported helpers take any integers
        L = ca.lib()
        L.cordic_calc_stages(pw)
        L.cordic_calc_stages_ww(iw, pw)
        L.cordic_calc_phase_bits(ow)
        L.cordic_phase_variance(ns, pw)
        L.cordic_gain_annihilator(ns)
    assert accepted > 300


def test_corrupt_config_is_refused_before_any_launch():
    """The config is a POD the caller owns; fields that cannot have come out
    of cordic_config_init* are refused (CORDIC_ERR_ARGS) ahead of the device."""
    good = ca.Config.from_cli(ca.P2R, 13, 13)
    for field, bad in (("nstages", 100), ("nlive", 65), ("ww", 70), ("pw", 40),
                       ("iw", 0), ("ow", 33), ("ww", 12), ("nlive", -1)):
        c = good.with_flags(0)
        setattr(c.c, field, bad)
        rc = ca.lib().cordic_p2r_const(c.ref, 8, 1, 0, 16, 16, 16, None)
        assert rc == ca.ERR_ARGS, (field, bad, rc)
    r = ca.Config.from_cli(ca.R2P, 13, 13).with_flags(0)
    r.c.nstages = 0
    assert ca.lib().cordic_r2p(r.ref, 8, 16, 16, 16, 16, None) == ca.ERR_ARGS
