"""The C++ host tools that sit on the C ABI: gencordic_amd (the reference's
command line) and cordic_tb (the reference's acceptance benches)."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import quality as Q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tools", "gencordic_amd")
TB = os.path.join(ROOT, "tools", "cordic_tb")


def test_gencordic_amd_writes_the_reference_header(tmp_path, golden):
    """sw/Makefile:115-144 command lines, pointed at gencordic_amd."""
    for name in ("rtl_cordic", "rtl_topolar", "rtl_seqcordic", "rtl_seqpolar",
                 "cfg2", "cfg3"):
        e = golden[name]
        vf = tmp_path / "core.v"
        r = subprocess.run([GEN] + e["args"].split() + ["-f", str(vf)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        text = (tmp_path / "core.h").read_text()
        body = re.search(r"#ifndef.*#endif[^\n]*\n", text, re.S).group(0)
        assert body == e["header"], name
        os.remove(tmp_path / "core.h")


def test_gencordic_amd_errors_like_the_reference():
    r = subprocess.run([GEN, "-t", "bogus"], capture_output=True, text=True)
    assert r.returncode != 0 and "ERR" in r.stderr
    r = subprocess.run([GEN, "-q"], capture_output=True, text=True)
    assert r.returncode != 0
    r = subprocess.run([GEN], capture_output=True, text=True)
    assert r.returncode == 0 and "USAGE" in r.stderr
    r = subprocess.run([GEN, "-v", "-t", "r2p", "-i", "13", "-o", "13"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "Phase  bits     : 21" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["p2r", "sp2r"])
def test_cordic_tb_p2r_report(mode):
    """Same report as bench/cpp/cordic_tb.cpp on the checked-in core."""
    r = subprocess.run([TB, "-t", mode, "-i", "13", "-o", "13", "-x", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "SUCCESS!!" in r.stdout
    m = dict(avg=float(re.search(r"AVG Err: ([\d.]+)", r.stdout).group(1)),
             mx=float(re.search(r"MAX Err: ([\d.]+)", r.stdout).group(1)),
             cnr=float(re.search(r"CNR    : ([\d.]+)", r.stdout).group(1)),
             sfdr=float(re.search(r"SFDR = +([\d.]+)", r.stdout).group(1)))
    c = O.config_cli(O.P2R if mode == "p2r" else O.SP2R, 13, 13, 2)
    ph, x0, y0 = Q.p2r_bench_inputs(c.iw, c.pw)
    ox, oy = O.rotate(c, x0, y0, ph)
    q = Q.p2r_quality(c, ph, x0, y0, ox, oy)
    assert abs(m["avg"] - q["averr"]) < 1e-5 and abs(m["mx"] - q["mxerr"]) < 1e-5
    assert abs(m["cnr"] - q["cnr"]) < 0.01
    assert abs(m["sfdr"] - Q.sfdr_dbc(ox, oy)) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["r2p", "sr2p"])
def test_cordic_tb_r2p_report(mode):
    r = subprocess.run([TB, "-t", mode, "-i", "13", "-o", "13", "-x", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "SUCCESS" in r.stdout
    mxp = float(re.search(r"Max phase     error: ([\d.]+)", r.stdout).group(1))
    mxv = float(re.search(r"Max magnitude error:\s+([\d.]+)", r.stdout).group(1))
    c = O.config_cli(O.R2P if mode == "r2p" else O.SR2P, 13, 13, 2)
    x, y, mg = Q.r2p_bench_inputs(c.iw, c.pw)
    mag, p = O.topolar(c, x, y)
    q = Q.r2p_quality(c, x, y, mg, mag, p)
    assert abs(mxp - q["mxperr"]) < 0.01 and abs(mxv - q["mxverr"]) < 1e-5


def test_gencordic_amd_writes_the_reference_hex_tables(tmp_path):
    """-t tbl / -t qtr: <fname>.hex byte for byte what the real generator
    writes (when it is available), and always what the golden hash says."""
    import hashlib
    import json
    gen = os.path.join(O.ORACLE_DIR, "_ref", "gencordic")
    with open(os.path.join(ROOT, "tests", "golden", "table_golden.json")) as f:
        golden = json.load(f)
    for args, e in golden.items():
        mine = tmp_path / "mine.v"
        r = subprocess.run([GEN] + args.split() + ["-f", str(mine)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        text = (tmp_path / "mine.hex").read_text()
        words = [int(t, 16) for t in text.split() if not t.startswith("@")]
        h = hashlib.sha256(b"".join(w.to_bytes(4, "little")
                                    for w in words)).hexdigest()
        assert h == e["sha256"], args
        if os.path.exists(gen):
            ref = tmp_path / "ref.v"
            subprocess.run([gen] + args.split() + ["-f", str(ref)], check=True,
                           capture_output=True)
            assert text == (tmp_path / "ref.hex").read_text(), args


def test_gencordic_amd_writes_the_reference_quadtbl_files(tmp_path):
    """-t qtbl: the three coefficient .hex files and the constants header
    carry what the real generator wrote (tests/golden/quad_golden.json)."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden",
                                       "quad_golden.json")))
    for name in ("rtl_quadtbl", "o16", "o20x4p24"):
        e = gold[name]
        vf = tmp_path / "core.v"
        r = subprocess.run([GEN, "-a", "-c"] + e["args"].split() +
                           ["-f", str(vf)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for t, bits in (("c", "CBITS"), ("l", "LBITS"), ("q", "QBITS")):
            b = e["localparams"][bits]
            toks = (tmp_path / ("core_%stbl.hex" % t)).read_text().split()
            assert toks[0] == "@00000000" and toks[9] == "@00000008"
            vals = [int(x, 16) for x in toks if not x.startswith("@")]
            assert all(len(x) == (b + 3) // 4 for x in toks
                       if not x.startswith("@"))
            vals = [v - (1 << b) if v >> (b - 1) else v for v in vals]
            assert vals == e[t + "tbl"], (name, t)
        hdr = dict(re.findall(r"const\t\w+\t(\w+)\s*= ([^;]+);",
                              (tmp_path / "core.h").read_text()))
        assert hdr == e["header"], name
    r = subprocess.run([GEN, "-t", "qtbl", "-o", "12", "-x", "0", "-p", "16",
                        "-f", str(tmp_path / "bad.v")], capture_output=True,
                       text=True)
    assert r.returncode != 0 and "ERR" in r.stderr


@pytest.mark.gpu
def test_cordic_tb_quadtbl_report():
    """bench/cpp/quadtbl_tb.cpp on the checked-in core (-p 18 -o 13)."""
    r = subprocess.run([TB, "-t", "qtbl", "-o", "13", "-p", "18"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "SUCCESS!!" in r.stdout
    mx = float(re.search(r"MXERR: ([\d.]+) \(Expected (-?[\d.]+)\)",
                         r.stdout).group(1))
    q = O.quad_cli(ow=13, pw=18)
    ph = np.arange(1 << 18, dtype=np.uint32)
    out = O.quad_lookup(q, O.quad_tables(q), ph)
    want = np.sin(ph * (2 * np.pi / (1 << 18))) * 4095
    assert abs(mx - np.abs(want - out).max()) < 1e-5
    assert "MXVAL: 0x%08x" % out.max() in r.stdout
    assert "MNVAL: 0x%08x" % (int(out.min()) & 0xffffffff) in r.stdout
    sfdr = float(re.search(r"SFDR = +([\d.]+)", r.stdout).group(1))
    assert sfdr > 80.0


def test_header_is_plain_c99(tmp_path):
    """include/cordic_amd.h must compile as C, not only as C++: the boundary
    is a C ABI (examples/sincos.c is a C99 caller)."""
    src = tmp_path / "t.c"
    src.write_text('#include "cordic_amd.h"\nint main(void) { cordic_config c; '
                   'return cordic_config_init(&c, CORDIC_P2R, 13, 13, 2, -1, -1); }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic",
                        "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_c_example_runs():
    exe = os.path.join(ROOT, "tools", "sincos")
    if not os.path.exists(exe):
        pytest.skip("tools/sincos not built")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "phase 0: (2385, 0)" in r.stdout        # SURVEY 8c structural value
    r = subprocess.run([exe, "-i", "32", "-o", "32", "-p", "32", "-n", "16"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_channeliser_example_runs_job_sets():
    """examples/channeliser.c: CORDIC_JOBS_MIX + CORDIC_JOBS_R2P from plain
    C99 -- 256 ragged blocks through mixer and converter as two launches,
    word for word what 512 per-block calls deliver."""
    exe = os.path.join(ROOT, "tools", "channeliser")
    if not os.path.exists(exe):
        pytest.skip("tools/channeliser not built")
    r = subprocess.run([exe, "-c", "256", "-l", "14"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("job sets vs per-block calls: equal") == 4
    assert "DIFFER" not in r.stdout
    sets = float(re.search(r"two job sets :\s+([\d.]+) ms", r.stdout).group(1))
    calls = float(re.search(r"512 calls  :\s+([\d.]+) ms", r.stdout).group(1))
    assert sets < calls / 3.0, r.stdout


def test_host_layer_under_address_and_ub_sanitizers(tmp_path):
    """tools/host_selftest.cpp: the host-only sources compiled with
    -fsanitize=address,undefined and driven with random / hostile parameters,
    undersized buffers and random command lines."""
    exe = tmp_path / "host_selftest"
    csrc = os.path.join(ROOT, "cordic_amd", "csrc")
    r = subprocess.run(
        ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined",
         "-fno-sanitize-recover=all", "-fwrapv", "-ffp-contract=off",
         "-I", os.path.join(ROOT, "include"), "-I", csrc,
         os.path.join(ROOT, "tools", "host_selftest.cpp"),
         os.path.join(csrc, "cordic_config.cpp"),
         os.path.join(csrc, "cordic_plan.cpp"),
         os.path.join(csrc, "cordic_quadtbl.cpp"), "-o", str(exe)],
        capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), "7"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "host selftest ok" in r.stdout


def test_ctypes_structures_have_the_headers_layout(tmp_path):
    """The ctypes view (cordic_amd/_native.py) against include/cordic_amd.h as
    a C compiler lays it out: size of every struct that crosses the boundary
    and the offset of every field -- a field added on one side only (round 6
    added d_xval / d_yval to cordic_job) fails here, on a CPU box."""
    import ctypes as C
    import cordic_amd._native as N
    pairs = [("cordic_config", N._CConfig), ("cordic_job", N._CJob),
             ("cordic_table_config", N._CTableConfig),
             ("cordic_queue_info", N._CQueueInfo),
             ("cordic_quad_config", N._CQuadConfig),
             ("cordic_p2r_quality", N._CP2RQuality),
             ("cordic_r2p_quality", N._CR2PQuality),
             ("cordic_host_stats", N._CHostStats)]
    body = ['#include <stddef.h>', '#include <stdio.h>', '#include "cordic_amd.h"',
            'int main(void) {']
    for cname, cls in pairs:
        body.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            body.append('printf("%s %s %%zu\\n", offsetof(%s, %s));'
                        % (cname, fname, cname, fname))
    body.append('return 0; }')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(body) + "\n")
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr          # (a field the header lacks)
    out = subprocess.run([str(exe)], capture_output=True, text=True).stdout
    want = {}
    for ln in out.splitlines():
        a, b, c = ln.split()
        want[(a, b)] = int(c)
    for cname, cls in pairs:
        assert C.sizeof(cls) == want[(cname, "size")], cname
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == want[(cname, fname)], (cname, fname)
