"""Clocked view of the pipelined cores (cordic_stream).

tests/golden/stream_vectors.json holds clock-by-clock port traces produced by
executing the reference generator's Verilog (vsim.py) under random i_ce /
i_reset / i_aux activity.  CPU: the register-level model (stream_model.py, on
top of the oracle) reproduces them.  GPU: cordic_stream_ticks reproduces them
when the trace is fed in blocks of arbitrary length, and equals the model on
long random traces."""
import json
import os
import shlex

import numpy as np
import pytest

import cordic_amd as ca
import oracle_lib as O
from stream_model import PipeModel

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                   "stream_vectors.json")))
MODE = {"p2r": ca.P2R, "r2p": ca.R2P}


def configs(args):
    a = shlex.split(args)
    get = lambda f, d: int(a[a.index(f) + 1]) if f in a else d  # noqa: E731
    t = (MODE[a[a.index("-t") + 1]], get("-i", -1), get("-o", -1),
         get("-x", 2), get("-p", -1), get("-n", -1))
    return t


@pytest.mark.parametrize("name", sorted(GOLD))
def test_register_model_reproduces_rtl_traces(name):
    g = GOLD[name]
    ocfg = O.config_cli(*configs(g["args"]))
    rot = "phase" in g
    m = PipeModel(ocfg, rot)
    assert m.ns == g["NSTAGES"]
    o0, o1, oa = m.run(g["x"], g["y"], g.get("phase"), g["ce"], g["reset"],
                       g["aux"])
    k0, k1 = ("o_xval", "o_yval") if rot else ("o_mag", "o_phase")
    assert o0.tolist() == g[k0]
    assert o1.tolist() == g[k1]
    assert oa.tolist() == g["o_aux"]


def _gpu_run(stream, rot, x, y, ph, ce, rs, aux, cuts):
    import torch
    dev = "cuda:0"

    def d32(a):
        return torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)
                                .astype(np.uint32).view(np.int32)).to(dev)

    def d8(a):
        return None if a is None else torch.from_numpy(
            np.ascontiguousarray(a, dtype=np.uint8)).to(dev)
    n = len(x)
    dx, dy = d32(x), d32(y)
    dph = d32(ph) if rot else None
    dce, drs, dax = d8(ce), d8(rs), d8(aux)
    o0 = torch.zeros(n, dtype=torch.int32, device=dev)
    o1 = torch.zeros(n, dtype=torch.int32, device=dev)
    oa = torch.zeros(n, dtype=torch.uint8, device=dev)
    edges = [0] + list(cuts) + [n]
    for a, b in zip(edges[:-1], edges[1:]):
        if b <= a:
            continue
        sl = slice(a, b)
        stream.ticks(dx[sl], dy[sl], dph[sl] if rot else None, o0[sl], o1[sl],
                     oa[sl], ce=None if dce is None else dce[sl],
                     reset=None if drs is None else drs[sl],
                     aux=None if dax is None else dax[sl], n=b - a)
    torch.cuda.synchronize()
    r0 = o0.cpu().numpy().astype(np.int64)
    r1 = o1.cpu().numpy()
    r1 = r1.astype(np.int64) if rot else r1.view(np.uint32).astype(np.int64)
    return r0, r1, oa.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_gpu_stream_reproduces_rtl_traces(name):
    g = GOLD[name]
    cfg = ca.Config.from_cli(*configs(g["args"]))
    rot = "phase" in g
    k0, k1 = ("o_xval", "o_yval") if rot else ("o_mag", "o_phase")
    n = len(g["x"])
    # one call, then the same trace cut into blocks at awkward places
    # (shorter than the pipeline, inside stalls, right after resets)
    for cuts in ([], [1, 2, 3, 7, 8, 40, 41, n // 6 + 3, n // 2, n // 2 + 5,
                      n - 9, n - 1]):
        s = ca.Stream(cfg)
        assert s.latency == g["NSTAGES"] + 2
        r0, r1, ra = _gpu_run(s, rot, g["x"], g["y"], g.get("phase"), g["ce"],
                              g["reset"], g["aux"], cuts)
        assert r0.tolist() == g[k0]
        assert r1.tolist() == g[k1]
        assert ra.tolist() == g["o_aux"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,iw,ow,pw,ns", [
    (ca.P2R, 13, 13, -1, -1), (ca.R2P, 13, 13, -1, -1),
    (ca.P2R, 32, 32, 32, 16), (ca.R2P, 24, 24, -1, 20)])
@pytest.mark.parametrize("flags", ["all", "ce", "reset", "none"])
def test_gpu_stream_long_random_trace_equals_model(mode, iw, ow, pw, ns, flags):
    cfg = ca.Config.from_cli(mode, iw, ow, 2, pw, ns)
    ocfg = O.config_cli(mode, iw, ow, 2, pw, ns)
    rot = mode == ca.P2R
    rng = np.random.RandomState(5)
    n = 40000                       # ~20 scan tiles, several spine rounds
    lo, hi = -(1 << (iw - 1)), (1 << (iw - 1))
    x, y = rng.randint(lo, hi, n), rng.randint(lo, hi, n)
    ph = rng.randint(0, 1 << cfg.pw, n, dtype=np.int64)
    aux = rng.randint(0, 2, n).astype(np.uint8)
    ce = (rng.randint(0, 4, n) != 0).astype(np.uint8) \
        if flags in ("all", "ce") else None
    rs = (rng.randint(0, 3000, n) == 0).astype(np.uint8) \
        if flags in ("all", "reset") else None
    if ce is not None:
        ce[5000:5000 + 3 * (cfg.nstages + 2)] = 0
    cuts = [17, 2048, 2049, 4096 + 5, 20000, 20001, 39990]
    s = ca.Stream(cfg)
    r0, r1, ra = _gpu_run(s, rot, x, y, ph, ce, rs, aux, cuts)
    m0, m1, ma = PipeModel(ocfg, rot).run(x, y, ph, ce, rs, aux)
    assert np.array_equal(r0, m0)
    assert np.array_equal(r1, m1)
    assert np.array_equal(ra, ma)
    # cordic_stream_reset == a clock with i_reset: outputs restart from zero
    s.reset()
    r0, r1, ra = _gpu_run(s, rot, x[:200], y[:200], ph[:200], None, None,
                          aux[:200], [])
    m0, m1, ma = PipeModel(ocfg, rot).run(x[:200], y[:200], ph[:200], None,
                                          None, aux[:200])
    assert np.array_equal(r0, m0) and np.array_equal(r1, m1)
    assert np.array_equal(ra, ma)


@pytest.mark.gpu
def test_stream_refuses_sequential_cores():
    with pytest.raises(ca.CordicError):
        ca.Stream(ca.Config.from_cli(ca.SP2R, 13, 13))


@pytest.mark.gpu
def test_clocked_views_replay_from_a_captured_graph():
    """cordic_stream_ticks / cordic_seq_ticks only enqueue once their scratch
    is reserved, and their state lives in one set of device buffers updated in
    place: a captured graph can be replayed, block after block."""
    import torch
    from seq_model import SeqModel
    dev = "cuda:0"
    n = 5000
    rng = np.random.RandomState(12)
    cfg = ca.Config.from_cli(ca.P2R, 13, 13)
    ocfg = O.config_cli(ca.P2R, 13, 13)
    scfg = ca.Config.from_cli(ca.SP2R, 13, 13)
    socfg = O.config_cli(ca.SP2R, 13, 13)
    t32 = lambda: torch.zeros(n, dtype=torch.int32, device=dev)   # noqa: E731
    t8 = lambda: torch.zeros(n, dtype=torch.uint8, device=dev)    # noqa: E731
    x, y, ph, o0, o1, q0, q1 = (t32() for _ in range(7))
    ce, rs, ax, oa, stb, qb, qd, qa = (t8() for _ in range(8))
    s, q = ca.Stream(cfg), ca.Seq(scfg)
    s.reserve(n)
    q.ticks(stb, x, y, ph, q0, q1, qb, qd, qa, reset=rs, aux=ax)   # reserves
    s.ticks(x, y, ph, o0, o1, oa, ce=ce, reset=rs, aux=ax)
    torch.cuda.synchronize()
    s.reset()
    q = ca.Seq(scfg)
    q.ticks(stb, x, y, ph, q0, q1, qb, qd, qa, reset=rs, aux=ax)
    torch.cuda.synchronize()
    q = ca.Seq(scfg)
    lib_reserve = ca.lib().cordic_seq_reserve
    assert lib_reserve(q._h, n) == 0
    s.reset()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s.ticks(x, y, ph, o0, o1, oa, ce=ce, reset=rs, aux=ax)
        q.ticks(stb, x, y, ph, q0, q1, qb, qd, qa, reset=rs, aux=ax)
    # capture does not execute: both objects are still in their reset state
    # the sequential view against the oracle's register-level model: the
    # random strobes below also land on completing clocks (re-runs)
    pm, sregs = PipeModel(ocfg, True), O.seq_regs()
    for block in range(3):
        hx = rng.randint(-4096, 4096, n)
        hy = rng.randint(-4096, 4096, n)
        hp = rng.randint(0, 1 << cfg.pw, n)
        hce = (rng.randint(0, 3, n) != 0).astype(np.uint8)
        hrs = (rng.randint(0, 900, n) == 0).astype(np.uint8)
        hax = rng.randint(0, 2, n).astype(np.uint8)
        hstb = (rng.randint(0, 4, n) == 0).astype(np.uint8)
        for t, h in ((x, hx), (y, hy), (ph, hp)):
            t.copy_(torch.from_numpy(h.astype(np.int32)))
        for t, h in ((ce, hce), (rs, hrs), (ax, hax), (stb, hstb)):
            t.copy_(torch.from_numpy(h))
        g.replay()
        torch.cuda.synchronize()
        m0, m1, ma = pm.run(hx, hy, hp, hce, hrs, hax)
        assert np.array_equal(o0.cpu().numpy(), m0)
        assert np.array_equal(o1.cpu().numpy(), m1)
        assert np.array_equal(oa.cpu().numpy(), ma)
        w0, w1, wa, wb, wd = O.seq_trace(socfg, hstb, hx, hy, hp, hrs, hax,
                                         regs=sregs)
        assert np.array_equal(q0.cpu().numpy(), w0)
        assert np.array_equal(q1.cpu().numpy(), w1)
        assert np.array_equal(qa.cpu().numpy(), wa)
        assert np.array_equal(qb.cpu().numpy(), wb)
        assert np.array_equal(qd.cpu().numpy(), wd)


GEN = os.path.join(O.ORACLE_DIR, "_ref", "gencordic")


@pytest.mark.skipif(not os.path.exists(GEN), reason="oracle/_ref not built")
def test_fresh_random_cores_clocked_by_vsim_match_the_models(tmp_path):
    """Beyond the committed traces: random pipelined AND sequential cores from
    the live generator, clocked by vsim.py under random activity, against the
    register model / the handshake model."""
    import re
    import subprocess
    import vsim
    from seq_model import SeqModel
    rng = np.random.RandomState(int.from_bytes(os.urandom(4), "little"))
    done = {"pipe": 0, "seq": 0}
    names = {ca.P2R: "p2r", ca.R2P: "r2p", ca.SP2R: "sp2r", ca.SR2P: "sr2p"}
    for trial in range(60):
        mode = int(rng.choice([ca.P2R, ca.R2P, ca.SP2R, ca.SR2P]))
        kind = "pipe" if mode in (ca.P2R, ca.R2P) else "seq"
        if done[kind] >= 2:
            if all(v >= 2 for v in done.values()):
                break
            continue
        iw, ow = int(rng.randint(4, 20)), int(rng.randint(4, 20))
        xtra, pw = int(rng.randint(1, 4)), int(rng.randint(8, 25))
        ns = int(rng.randint(4, 22))
        try:
            ocfg = O.config_cli(mode, iw, ow, xtra, pw, ns)
        except ValueError:
            continue
        args = ["-a", "-t", names[mode], "-i", str(iw), "-o", str(ow), "-x",
                str(xtra), "-p", str(pw), "-n", str(ns)]
        vf = tmp_path / ("c%d.v" % trial)
        subprocess.run([GEN] + args + ["-c", "-f", str(vf)], check=True,
                       capture_output=True)
        text = vf.read_text().replace("// }}}\talways", "// }}}\n\talways")
        m = vsim.Module(text)
        rot = mode in (ca.P2R, ca.SP2R)
        n = 500
        lo, hi = -(1 << (iw - 1)), (1 << (iw - 1))
        x, y = rng.randint(lo, hi, n), rng.randint(lo, hi, n)
        ph = rng.randint(0, 1 << pw, n, dtype=np.int64)
        aux = rng.randint(0, 2, n).astype(np.uint8)
        rs = (rng.randint(0, 120, n) == 0).astype(np.uint8)
        outs = ["o_xval", "o_yval"] if rot else ["o_mag", "o_phase"]
        got = {k: [] for k in outs + ["o_aux"]}
        if kind == "pipe":
            ce = (rng.randint(0, 3, n) != 0).astype(np.uint8)
            for t in range(n):
                pins = dict(i_xval=int(x[t]), i_yval=int(y[t]), i_ce=int(ce[t]),
                            i_reset=int(rs[t]), i_aux=int(aux[t]))
                if rot:
                    pins["i_phase"] = int(ph[t])
                m.tick(**pins)
                for k in outs:
                    v = m.out(k)
                    got[k].append(v & ((1 << pw) - 1) if k == "o_phase" else v)
                got["o_aux"].append(int(m.get("o_aux")))
            w0, w1, wa = PipeModel(ocfg, rot).run(x, y, ph, ce, rs, aux)
        else:
            cpo = ocfg.clocks_per_output
            stb = (rng.rand(n) < rng.choice([1.0, 0.3, 0.05])).astype(np.uint8)
            got.update(o_busy=[], o_done=[])
            left = 0
            for t in range(n):
                if left == 1 and not rs[t]:
                    stb[t] = 0          # keep to the protocol on this clock
                pins = dict(i_xval=int(x[t]), i_yval=int(y[t]),
                            i_stb=int(stb[t]), i_reset=int(rs[t]),
                            i_aux=int(aux[t]))
                if rot:
                    pins["i_phase"] = int(ph[t])
                idle = not m.get("o_busy")
                m.tick(**pins)
                left = 0 if rs[t] else (
                    (cpo - 1 if (stb[t] and idle) else 0) if left == 0
                    else left - 1)
                for k in outs:
                    v = m.out(k)
                    got[k].append(v & ((1 << pw) - 1) if k == "o_phase" else v)
                for k in ("o_aux", "o_busy", "o_done"):
                    got[k].append(int(m.get(k)))
            w0, w1, wa, wb, wd = SeqModel(ocfg, rot).run(stb, x, y, ph, rs, aux)
            assert wb.tolist() == got["o_busy"], args
            assert wd.tolist() == got["o_done"], args
        assert w0.tolist() == got[outs[0]], args
        assert w1.tolist() == got[outs[1]], args
        assert wa.tolist() == got["o_aux"], args
        done[kind] += 1
    assert done["pipe"] >= 1 and done["seq"] >= 1
