"""No entry point writes outside [0, n) of its output arrays: every output is
a window of a larger buffer filled with a canary, for sizes around the vector
width, the block sizes and the chunking of the kernels."""
import numpy as np
import pytest

import cordic_amd as ca

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
DEV = "cuda:0"
PAD = 64
CANARY = {torch.int32: 0x5A5A5A5A, torch.int16: 0x5A5A, torch.uint8: 0x5A}
SIZES = [1, 3, 4, 5, 255, 1023, 1025, 4096, 4099, 70001]


class Guarded:
    def __init__(self, n, dtype=torch.int32, off=0):
        self.n, self.off = n, off
        self.buf = torch.full((n + 2 * PAD + off,), CANARY[dtype], dtype=dtype,
                              device=DEV)
        self.win = self.buf[PAD + off:PAD + off + n]

    def intact(self):
        lo = self.buf[:PAD + self.off]
        hi = self.buf[PAD + self.off + self.n:]
        c = CANARY[self.buf.dtype]
        return bool((lo == c).all() and (hi == c).all())


def rnd(n, dtype=torch.int32, lim=None):
    lim = lim or (2**31 - 1 if dtype == torch.int32 else 2**15 - 1)
    return torch.randint(-lim, lim, (max(n, 1),), dtype=torch.int64,
                         device=DEV).to(dtype)[:n]


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("off", [0, 1])
def test_rotators_and_converters_stay_inside(n, off):
    for cli, flags in (((ca.P2R, 32, 32, 2, 32, 16), 0),
                       ((ca.P2R, 13, 13, 2, -1, -1), 0),
                       ((ca.P2R, 32, 32, 3, 32, 19), 0),
                       ((ca.P2R, 13, 13, 2, -1, -1), ca.FLAG_FORCE_GENERIC)):
        cfg = ca.Config.from_cli(*cli)
        if flags:
            cfg = cfg.with_flags(flags)
        ox, oy = Guarded(n, off=off), Guarded(n, off=off)
        ca.p2r(cfg, rnd(n), rnd(n), rnd(n), ox.win, oy.win, n=n)
        plan = ca.Plan(cfg)
        px, py = Guarded(n, off=off), Guarded(n, off=off)
        plan.p2r_const(1000, -7, rnd(n), px.win, py.win, n=n)
        qx, qy = Guarded(n, off=off), Guarded(n, off=off)
        plan.nco(n, 5, 0x01234567, 1 << 35, 1000, 0, qx.win, qy.win)
        torch.cuda.synchronize()
        assert all(g.intact() for g in (ox, oy, px, py, qx, qy)), cli
    for cli in ((ca.R2P, 24, 24, 2, -1, 20), (ca.R2P, 32, 32, 2, 32, 24),
                (ca.SR2P, 13, 13, 2, -1, -1)):
        cfg = ca.Config.from_cli(*cli)
        m, p = Guarded(n, off=off), Guarded(n, off=off)
        ca.r2p(cfg, rnd(n), rnd(n), m.win, p.win, n=n)
        torch.cuda.synchronize()
        assert m.intact() and p.intact(), cli


@pytest.mark.parametrize("n", SIZES)
def test_16bit_tables_and_quad_stay_inside(n):
    cfg = ca.Config.from_cli(ca.P2R, 16, 16, 2, 16, 16)
    for off in (0, 1, 3):
        ox, oy = (Guarded(n, torch.int16, off) for _ in range(2))
        ca.p2r(cfg, rnd(n, torch.int16), rnd(n, torch.int16),
               rnd(n, torch.int16), ox.win, oy.win, n=n)
        px, py = (Guarded(n, torch.int16, off) for _ in range(2))
        ca.Plan(cfg).p2r_const(32767, 0, rnd(n, torch.int16), px.win, py.win,
                               n=n)
        torch.cuda.synchronize()
        assert all(g.intact() for g in (ox, oy, px, py))
    rcfg = ca.Config.from_cli(ca.R2P, 16, 16, 2, 16, -1)
    m, p = Guarded(n, torch.int16), Guarded(n, torch.int16)
    ca.r2p(rcfg, rnd(n, torch.int16), rnd(n, torch.int16), m.win, p.win, n=n)
    for tab in (ca.Table(ca.TBL, -1, 13, 17), ca.Table(ca.QTR, -1, 24, 18),
                ca.Table(ca.QTR, -1, 16, 17)):
        for off in (0, 1):
            o = Guarded(n, off=off)
            tab.lookup(rnd(n), o.win, n=n)
            torch.cuda.synchronize()
            assert o.intact()
    quad = ca.Quad(ow=13, pw=18)
    for off in (0, 1):
        o = Guarded(n, off=off)
        quad.lookup(rnd(n), o.win, n=n)
        torch.cuda.synchronize()
        assert o.intact()
    assert m.intact() and p.intact()


@pytest.mark.parametrize("n", [1, 17, 18, 19, 2047, 2049, 5000])
def test_clocked_views_stay_inside(n):
    cfg = ca.Config.from_cli(ca.P2R, 13, 13)
    scfg = ca.Config.from_cli(ca.SP2R, 13, 13)
    flags = lambda p: (torch.rand(n, device=DEV) < p).to(torch.uint8)  # noqa
    for kw in ({}, dict(ce=flags(0.7), reset=flags(0.01), aux=flags(0.5))):
        o0, o1 = Guarded(n), Guarded(n)
        oa = Guarded(n, torch.uint8)
        ca.Stream(cfg).ticks(rnd(n, lim=4000), rnd(n, lim=4000), rnd(n),
                             o0.win, o1.win, oa.win, n=n, **kw)
        torch.cuda.synchronize()
        assert o0.intact() and o1.intact() and oa.intact()
    o0, o1 = Guarded(n), Guarded(n)
    ob, od, oa = (Guarded(n, torch.uint8) for _ in range(3))
    ca.Seq(scfg).ticks(flags(0.3), rnd(n, lim=4000), rnd(n, lim=4000), rnd(n),
                       o0.win, o1.win, ob.win, od.win, oa.win,
                       reset=flags(0.01), aux=flags(0.5), n=n)
    torch.cuda.synchronize()
    assert all(g.intact() for g in (o0, o1, ob, od, oa))
