"""16-bit sample containers (cordic_*16): the same cores on int16 / uint16
arrays must return exactly the low 16 bits of what the oracle computes -- which
is the whole value, the ports being at most 16 bits wide.  SURVEY.md 8(d)
config 1 (-t p2r -i 16 -o 16 -p 16 -n 16) is the case these exist for."""
import numpy as np
import pytest

import cordic_amd as ca
import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
DEV = "cuda:0"


def dev16(a, offset=0):
    """int16 device view `offset` elements past an aligned allocation"""
    a = np.ascontiguousarray(a).view(np.int16)
    t = torch.zeros(a.size + offset + 8, dtype=torch.int16, device=DEV)
    v = t[offset:offset + a.size]
    if a.size:
        v.copy_(torch.from_numpy(a).to(DEV))
    return v


def out16(n, offset=0):
    return torch.zeros(n + offset + 8, dtype=torch.int16,
                       device=DEV)[offset:offset + n]


def both(mode, iw, ow, xtra=2, pw=-1, ns=-1):
    return (ca.Config.from_cli(mode, iw, ow, xtra, pw, ns),
            O.config_cli(mode, iw, ow, xtra, pw, ns))


def rand16(rng, iw, pw, n):
    lo, hi = -(1 << (iw - 1)), (1 << (iw - 1))
    x = rng.randint(lo, hi, n).astype(np.int16)
    y = rng.randint(lo, hi, n).astype(np.int16)
    ph = rng.randint(0, 1 << 16, n).astype(np.uint16)
    ext = [lo, hi - 1, 0, -1, 1]
    k = 0
    for a in ext:
        for b in ext:
            if k < n:
                x[k], y[k] = a, b
                k += 1
    q = 1 << max(pw - 3, 0)
    for j, e in enumerate([(i * q + d) & 0xffff for i in range(9)
                           for d in (-1, 0, 1)]):
        if 30 + j < n:
            ph[30 + j] = e
    return x, y, ph


P2R16 = {
    "cfg1": (ca.P2R, 16, 16, 2, 16, 16),
    "cfg1_seq": (ca.SP2R, 16, 16, 2, 16, 16),
    "i12o14": (ca.P2R, 12, 14, 3, 15, -1),
    "i16o8": (ca.P2R, 16, 8, 2, 12, 10),
    "i8o16": (ca.P2R, 8, 16, 4, 16, 18),
}


@pytest.mark.parametrize("offset", [0, 1, 4])
@pytest.mark.parametrize("name", sorted(P2R16))
def test_p2r16_matches_oracle(name, offset):
    cfg, ocfg = both(*P2R16[name])
    rng = np.random.RandomState(31)
    n = (1 << 18) + 3
    x, y, ph = rand16(rng, cfg.iw, cfg.pw, n)
    rx, ry = O.rotate(ocfg, x.astype(np.int32), y.astype(np.int32),
                      ph.astype(np.uint32))
    assert rx.min() >= -32768 and rx.max() <= 32767
    ox, oy = out16(n, offset), out16(n, offset)
    ca.p2r(cfg, dev16(x, offset), dev16(y, offset), dev16(ph, offset), ox, oy,
           n=n)
    torch.cuda.synchronize()
    assert np.array_equal(ox.cpu().numpy(), rx.astype(np.int16))
    assert np.array_equal(oy.cpu().numpy(), ry.astype(np.int16))

    # constant vector, with and without a plan (seeded when eligible)
    x0 = (1 << (cfg.iw - 1)) - 1
    rx, ry = O.rotate(ocfg, x0, 0, ph.astype(np.uint32))
    dph = dev16(ph, offset)
    ca.p2r_const(cfg, x0, 0, dph, ox, oy, n=n)
    torch.cuda.synchronize()
    assert np.array_equal(ox.cpu().numpy(), rx.astype(np.int16))
    assert np.array_equal(oy.cpu().numpy(), ry.astype(np.int16))
    plan = ca.Plan(cfg)
    ox.zero_(); oy.zero_()
    plan.p2r_const(x0, 0, dph, ox, oy, n=n)
    torch.cuda.synchronize()
    assert np.array_equal(ox.cpu().numpy(), rx.astype(np.int16))
    assert np.array_equal(oy.cpu().numpy(), ry.astype(np.int16))


def test_cfg1_exhaustive_ramp_is_seeded_and_exact():
    """N = 2^20, phase[n] = n mod 2^16, x = 32767."""
    cfg, ocfg = both(ca.P2R, 16, 16, 2, 16, 16)
    n = 1 << 20
    ph = (np.arange(n) & 0xffff).astype(np.uint16)
    rx, ry = O.rotate(ocfg, 32767, 0, ph.astype(np.uint32))
    plan = ca.Plan(cfg)
    assert plan.seed_info["stages"] > 0
    ox, oy = out16(n), out16(n)
    plan.p2r_const(32767, 0, dev16(ph), ox, oy)
    torch.cuda.synchronize()
    assert np.array_equal(ox.cpu().numpy(), rx.astype(np.int16))
    assert np.array_equal(oy.cpu().numpy(), ry.astype(np.int16))
    # same samples through the 32-bit containers
    o32x = torch.zeros(n, dtype=torch.int32, device=DEV)
    o32y = torch.zeros(n, dtype=torch.int32, device=DEV)
    p32 = torch.from_numpy(ph.astype(np.int32)).to(DEV)
    plan.p2r_const(32767, 0, p32, o32x, o32y)
    torch.cuda.synchronize()
    assert torch.equal(o32x.to(torch.int16), ox)
    assert torch.equal(o32y.to(torch.int16), oy)


@pytest.mark.parametrize("pw", [16, 24, 32])
def test_nco16(pw):
    cfg, ocfg = both(ca.P2R, 16, 16, 2, pw, 16)
    n = (1 << 18) + 2
    fcw, ph0, idx0 = 0x01234567 >> (32 - pw), 5, (1 << 33) + 7
    ph = ((ph0 + (np.arange(n, dtype=np.uint64) + np.uint64(idx0))
           * np.uint64(fcw)) & np.uint64((1 << pw) - 1)).astype(np.uint32)
    rx, ry = O.rotate(ocfg, 32767, 0, ph)
    for runner in (lambda *a: ca.nco(cfg, *a), ca.Plan(cfg).nco):
        ox, oy = out16(n), out16(n)
        runner(n, ph0, fcw, idx0, 32767, 0, ox, oy)
        torch.cuda.synchronize()
        assert np.array_equal(ox.cpu().numpy(), rx.astype(np.int16))
        assert np.array_equal(oy.cpu().numpy(), ry.astype(np.int16))


R2P16 = {
    "i16o16p16": (ca.R2P, 16, 16, 2, 16, -1),
    "i13o13p16": (ca.R2P, 13, 13, 2, 16, 14),
    "seq_i16o12": (ca.SR2P, 16, 12, 2, 14, -1),
}


@pytest.mark.parametrize("offset", [0, 2])
@pytest.mark.parametrize("name", sorted(R2P16))
def test_r2p16_matches_oracle(name, offset):
    cfg, ocfg = both(*R2P16[name])
    rng = np.random.RandomState(32)
    n = (1 << 18) + 1
    x, y, _ = rand16(rng, cfg.iw, cfg.pw, n)
    rm, rp = O.topolar(ocfg, x.astype(np.int32), y.astype(np.int32))
    mag, oph = out16(n, offset), out16(n, offset)
    ca.r2p(cfg, dev16(x, offset), dev16(y, offset), mag, oph, n=n)
    torch.cuda.synchronize()
    assert np.array_equal(mag.cpu().numpy(), rm.astype(np.int16))
    assert np.array_equal(oph.cpu().numpy().view(np.uint16),
                          rp.astype(np.uint16))


def test_container_errors():
    """A port that does not fit the container is refused, not truncated."""
    t = out16(16)
    with pytest.raises(ca.CordicError) as e:
        ca.p2r_const(ca.Config.from_cli(ca.P2R, 17, 16, 2, 16, 16), 1, 0, t, t, t)
    assert e.value.status == ca.ERR_CONTAINER
    with pytest.raises(ca.CordicError) as e:
        ca.p2r_const(ca.Config.from_cli(ca.P2R, 16, 16, 2, 17, 16), 1, 0, t, t, t)
    assert e.value.status == ca.ERR_CONTAINER
    with pytest.raises(ca.CordicError) as e:
        ca.r2p(ca.Config.from_cli(ca.R2P, 16, 16, 2, -1, -1), t, t, t, t)
    assert e.value.status == ca.ERR_CONTAINER      # default PW is 23 here
    # the NCO takes scalar phases: any PW
    ca.nco(ca.Config.from_cli(ca.P2R, 16, 16, 2, 32, 16), 16, 0, 1, 0, 1, 0, t, t)
    torch.cuda.synchronize()


def test_unit_gain_on_16bit_containers():
    """CORDIC_FLAG_UNIT_GAIN through the io16 instances (rotator, seeded,
    converter)."""
    base, ocfg = both(ca.P2R, 16, 16, 2, 16, 16)
    cfg = base.with_flags(ca.FLAG_UNIT_GAIN)
    k = ca.lib().cordic_config_gain_annihilator(cfg.ref)
    rng = np.random.RandomState(33)
    n = (1 << 16) + 2
    x, y, ph = rand16(rng, 16, 16, n)

    def scaled(a):
        return ((a.astype(np.int64) * k) >> 32).astype(np.int16)
    rx, ry = O.rotate(ocfg, x.astype(np.int32), y.astype(np.int32),
                      ph.astype(np.uint32))
    ox, oy = out16(n), out16(n)
    ca.p2r(cfg, dev16(x), dev16(y), dev16(ph), ox, oy, n=n)
    torch.cuda.synchronize()
    assert np.array_equal(ox.cpu().numpy(), scaled(rx))
    assert np.array_equal(oy.cpu().numpy(), scaled(ry))
    rx, ry = O.rotate(ocfg, 32767, 0, ph.astype(np.uint32))
    ca.Plan(cfg).p2r_const(32767, 0, dev16(ph), ox, oy, n=n)
    torch.cuda.synchronize()
    assert np.array_equal(ox.cpu().numpy(), scaled(rx))
    assert np.array_equal(oy.cpu().numpy(), scaled(ry))

    base, ocfg = both(ca.R2P, 16, 16, 2, 16, -1)
    cfg = base.with_flags(ca.FLAG_UNIT_GAIN)
    k = ca.lib().cordic_config_gain_annihilator(cfg.ref)
    rm, rp = O.topolar(ocfg, x.astype(np.int32), y.astype(np.int32))
    mag, oph = out16(n), out16(n)
    ca.r2p(cfg, dev16(x), dev16(y), mag, oph, n=n)
    torch.cuda.synchronize()
    assert np.array_equal(mag.cpu().numpy(), scaled(rm))
    assert np.array_equal(oph.cpu().numpy().view(np.uint16), rp.astype(np.uint16))
