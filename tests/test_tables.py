"""Table cores (row F4: -t tbl / -t qtr, sw/sintable.cpp): parameters and table
contents against the real generator's .hex files, the emitted RTL executed by
vsim against the oracle, and the GPU gather against the oracle."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import cordic_amd as ca
import oracle_lib as O
import vsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(O.ORACLE_DIR, "_ref", "gencordic")


def parse(args):
    a = args.split()
    d = {"-t": None, "-i": -1, "-o": -1, "-p": -1}
    for i in range(0, len(a), 2):
        d[a[i]] = a[i + 1] if a[i] == "-t" else int(a[i + 1])
    kind = O.TBL if d["-t"] == "tbl" else O.QTR
    return kind, d["-i"], d["-o"], d["-p"]


def sha(words, ow):
    w = (np.asarray(words).astype(np.int64) & ((1 << ow) - 1)).astype("<u4")
    return hashlib.sha256(w.tobytes()).hexdigest()


def test_tables_match_the_generators_hex_files():
    with open(os.path.join(ROOT, "tests", "golden", "table_golden.json")) as f:
        golden = json.load(f)
    assert len(golden) >= 12
    for args, e in golden.items():
        assert not e.get("failed"), args
        kind, iw, ow, pw = parse(args)
        # oracle
        opw, oow = O.table_config(kind, iw, ow, pw)
        assert (opw, oow) == (e["PW"], e["OW"]), args
        tv = O.table_values(kind, opw, oow)
        assert tv.size == e["entries"] and sha(tv, oow) == e["sha256"], args
        # product (host layer through the C ABI)
        t = ca.Table(kind, iw, ow, pw, device=False)
        assert (t.pw, t.ow, t.entries) == (e["PW"], e["OW"], e["entries"]), args
        pv = t.values()
        assert sha(pv, t.ow) == e["sha256"], args
        if "words" in e:
            assert ((pv.astype(np.int64) & ((1 << t.ow) - 1)).tolist()
                    == e["words"]), args


def test_table_parameter_limits():
    for kind in (ca.TBL, ca.QTR):
        with pytest.raises(ca.CordicError):
            ca.Table(kind, -1, 31, 12, device=False)    # hexfile.cpp: ow < 31
        with pytest.raises(ca.CordicError):
            ca.Table(kind, -1, 13, 26, device=False)    # sintable.cpp: < 2^26
    with pytest.raises(ca.CordicError):
        ca.Table(9, -1, 13, 12, device=False)
    with pytest.raises(ValueError):
        O.table_config(O.TBL, -1, 24, -1)       # default OW 24 -> PW 28: refused


@pytest.mark.skipif(not os.path.exists(GEN), reason="oracle/_ref not built")
@pytest.mark.parametrize("args", ["-t tbl -p 9 -o 10", "-t qtr -p 10 -o 12",
                                  "-t qtr -i 8", "-t tbl -o 7"])
def test_emitted_table_rtl_executed_by_vsim_equals_oracle(args, tmp_path):
    vf = tmp_path / "core.v"
    subprocess.run([GEN, "-a"] + args.split() + ["-f", str(vf)], check=True,
                   capture_output=True)
    m = vsim.Module(vf.read_text(), readmem_dir=str(tmp_path))
    kind, iw, ow, pw = parse(args)
    pw, ow = O.table_config(kind, iw, ow, pw)
    assert (m.params["PW"], m.params["OW"]) == (pw, ow)
    tbl = O.table_values(kind, pw, ow)
    ph = np.arange(1 << pw, dtype=np.uint32)
    res = vsim.run_pipelined(m, [dict(i_phase=int(p)) for p in ph])
    exp = O.table_lookup(kind, pw, ow, tbl, ph)
    assert [r["o_val"] for r in res] == [int(v) & ((1 << ow) - 1) for v in exp]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,iw,ow,pw", [(ca.TBL, -1, 13, 17),
                                           (ca.QTR, -1, 24, 18),
                                           (ca.TBL, -1, 8, 6),
                                           (ca.QTR, 8, -1, -1),
                                           (ca.QTR, -1, 30, 20),
                                           (ca.QTR, -1, 16, 17),
                                           (ca.TBL, -1, 16, 16),
                                           (ca.TBL, -1, 12, 15),
                                           (ca.QTR, -1, 9, 5),
                                           (ca.QTR, -1, 24, 17),
                                           (ca.QTR, -1, 20, 14),
                                           (ca.TBL, -1, 24, 17),
                                           (ca.TBL, -1, 18, 10)])
def test_gpu_table_lookup_equals_oracle(kind, iw, ow, pw):
    import torch
    from gpu_util import DEV, dev_i32, to_np
    t = ca.Table(kind, iw, ow, pw)
    tbl = O.table_values(kind, t.pw, t.ow)
    rng = np.random.RandomState(2)
    for n, off in ((0, 0), (5, 0), (1 << 20, 0), ((1 << 16) + 3, 1)):
        ph = rng.randint(0, 1 << 32, n + off, dtype=np.uint64).astype(np.uint32)
        if n >= (1 << t.pw):
            ph[: 1 << t.pw] = np.arange(1 << t.pw, dtype=np.uint32)  # exhaustive
        dph = dev_i32(ph) if ph.size else torch.zeros(4, dtype=torch.int32,
                                                      device=DEV)
        out = torch.zeros(max(n + off, 4), dtype=torch.int32, device=DEV)
        t.lookup(dph[off:], out[off:], n=n)
        torch.cuda.synchronize()
        exp = O.table_lookup(kind, t.pw, t.ow, tbl, ph[off:off + n])
        assert np.array_equal(to_np(out)[off:off + n], exp)
    # small 16-bit tables are served from LDS (a full-wave table only if the
    # generated entries really are symmetric); the rest gather from L2
    # (wider outputs: the 32-bit entries in LDS, modes 3 / 4)
    fits = (1 << (t.pw - 2)) <= 32768 and t.pw >= 4
    small = t.ow <= 16 and fits
    wide = t.ow > 16 and fits
    assert t.lds_mode in ((1,) if (small and kind == ca.QTR) else
                          (0, 2) if small else
                          (3,) if (wide and kind == ca.QTR) else
                          (0, 4) if wide else (0,))
    t.close()
