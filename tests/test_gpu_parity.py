"""Parity tests proper: the HIP path through the C ABI against the oracle on
the same inputs -- bit-exact (integer arithmetic: tolerance is zero)."""
import numpy as np
import pytest

import cordic_amd as ca
import oracle_lib as O
import quality as Q

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if torch.cuda.is_available():
    from gpu_util import (DEV, cpu_digest, dev_i32, gpu_digest, gpu_nco,
                          gpu_p2r, gpu_plan_nco, gpu_plan_p2r, gpu_r2p, to_np)


def both(mode, iw=-1, ow=-1, xtra=2, pw=-1, ns=-1):
    return (ca.Config.from_cli(mode, iw, ow, xtra, pw, ns),
            O.config_cli(mode, iw, ow, xtra, pw, ns))


def rand_inputs(rng, iw, pw, n):
    lo, hi = -(1 << (iw - 1)), (1 << (iw - 1))
    x = rng.randint(lo, hi, n).astype(np.int32)
    y = rng.randint(lo, hi, n).astype(np.int32)
    ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    # adversarial head: extremes and octant / quadrant boundaries
    ext = [lo, hi - 1, 0, -1, 1, lo + 1]
    k = 0
    for a in ext:
        for b in ext:
            if k < n:
                x[k], y[k] = a, b
                k += 1
    q = 1 << max(pw - 3, 0)
    edges = [(j * q + d) & 0xffffffff for j in range(9) for d in (-1, 0, 1)]
    for j, e in enumerate(edges):
        if 40 + j < n:
            ph[40 + j] = e
    return x, y, ph


def assert_p2r(cfg, ocfg, x, y, ph, **kw):
    gx, gy = gpu_p2r(cfg, x, y, ph, **kw)
    rx, ry = O.rotate(ocfg, x, y, ph)
    assert np.array_equal(gx, rx) and np.array_equal(gy, ry)


def assert_r2p(cfg, ocfg, x, y, **kw):
    gm, gp = gpu_r2p(cfg, x, y, **kw)
    rm, rp = O.topolar(ocfg, x, y)
    assert np.array_equal(gm, rm) and np.array_equal(gp, rp)


# ------------------------------------------------------- BASELINE configs

BASELINE_P2R = {
    # WW = 33 / 34 / 35 take the left-justified wide kernels (LJ 30 / 30 / 29)
    "ww33": (ca.P2R, 30, 30, 2, 32, 16),
    "ww34": (ca.P2R, 31, 31, 2, 30, 20),
    "ww34_seq": (ca.SP2R, 31, 28, 2, 32, 24),
    "ww35_30st": (ca.P2R, 32, 32, 2, 32, 30),
    "ww36": (ca.P2R, 32, 32, 3, 32, 16),
    "ww41": (ca.P2R, 32, 20, 8, 32, 24),
    "ww48": (ca.P2R, 32, 32, 15, 32, 30),
    "cfg1": (ca.P2R, 16, 16, 2, 16, 16),
    "cfg2": (ca.P2R, 32, 32, 2, 32, 16),
    "cfg4": (ca.P2R, 32, 32, 2, 32, 24),
    "cfg5_seq": (ca.SP2R, 32, 32, 2, 32, 16),
    # PW = 3: angle table degenerates (found by tools/fuzz_gpu.py); no seeds
    "degenerate_pw3": (ca.SP2R, 9, 31, 3, 3, 33),
    "rtl_cordic": (ca.P2R, 13, 13, 2, -1, -1),
    "rtl_seqcordic": (ca.SP2R, 13, 13, 2, -1, -1),
}


@pytest.mark.parametrize("name", sorted(BASELINE_P2R))
def test_p2r_baseline_configs(name):
    cfg, ocfg = both(*BASELINE_P2R[name])
    rng = np.random.RandomState(11)
    n = (1 << 20) + 3
    x, y, ph = rand_inputs(rng, cfg.iw, cfg.pw, n)
    assert_p2r(cfg, ocfg, x, y, ph)                     # vector inputs
    x0 = (1 << (cfg.iw - 1)) - 1
    assert_p2r(cfg, ocfg, x0, 0, ph)                    # bench-style const
    assert_p2r(cfg, ocfg, -(1 << (cfg.iw - 1)), -(1 << (cfg.iw - 1)), ph)
    # the bench's ramp (cordic_tb.cpp:138)
    ramp = (np.arange(n, dtype=np.uint64) << 2).astype(np.uint32)
    assert_p2r(cfg, ocfg, x0, 0, ramp)


BASELINE_R2P = {
    "cfg3": (ca.R2P, 24, 24, 2, -1, 20),
    "rtl_topolar": (ca.R2P, 13, 13, 2, -1, -1),
    "rtl_seqpolar": (ca.SR2P, 13, 13, 2, -1, -1),
    "seq_cfg3": (ca.SR2P, 24, 24, 2, -1, 20),
}


@pytest.mark.parametrize("name", sorted(BASELINE_R2P))
def test_r2p_baseline_configs(name):
    cfg, ocfg = both(*BASELINE_R2P[name])
    rng = np.random.RandomState(12)
    n = (1 << 20) + 1
    x, y, _ = rand_inputs(rng, cfg.iw, cfg.pw, n)
    assert_r2p(cfg, ocfg, x, y)
    # SURVEY 8d config 3 throughput ramps
    g = np.arange(n, dtype=np.uint64)
    sh = 32 - cfg.iw

    def ramp(mul):
        v = (((g * mul) & 0xffffffff) >> 8).astype(np.uint32)
        return ((v << sh).astype(np.int32) >> sh)
    assert_r2p(cfg, ocfg, ramp(0x9E3779B1), ramp(0x85EBCA77))


# ------------------------------------------ exhaustive, reference criteria

def test_exhaustive_p2r_checked_in_core_and_bench_criteria():
    """All 2^20 phases of the checked-in core (rtl/cordic.v): equal to the
    oracle, and the GPU output passes cordic_tb's thresholds."""
    cfg, ocfg = both(ca.P2R, 13, 13, 2)
    ph, x0, y0 = Q.p2r_bench_inputs(cfg.iw, cfg.pw)
    gx, gy = gpu_p2r(cfg, x0, y0, ph)
    rx, ry = O.rotate(ocfg, x0, y0, ph)
    assert np.array_equal(gx, rx) and np.array_equal(gy, ry)
    q = Q.p2r_quality(cfg, ph, x0, y0, gx, gy)
    assert q["ok"], q
    assert abs(q["cnr"] - cfg.best_possible_cnr) < 0.5


POL_LJ_CASES = [   # (iw, ow, xtra, pw, nstages): WW <= 34 (the last rows: 33, 34)
    (24, 24, 2, -1, 20), (24, 24, 2, -1, 16), (24, 24, 2, -1, 18),
    (24, 24, 2, -1, 24), (24, 24, 2, -1, 22), (13, 13, 2, -1, -1),
    (16, 16, 2, -1, -1), (20, 24, 1, 30, 19), (26, 26, 0, -1, 24),
    (8, 8, 2, -1, -1), (12, 16, 3, -1, 12), (24, 24, 2, 32, 3),
    (24, 24, 2, 32, 2), (24, 24, 2, 32, 11), (24, 24, 2, 32, 1),
    (24, 24, 2, 32, 30), (22, 22, 3, 32, 37), (10, 10, 2, 32, 28),
    (25, 25, 2, 32, 20), (26, 26, 2, 32, 20), (27, 27, 1, 32, 24),
    (26, 24, 2, 32, 1), (28, 28, 0, 32, 31), (26, 13, 2, 24, 16),
    # WW - OW >= 32: the rounding increment no longer fits a signed
    # multiplicand (found by tools/fuzz_gpu.py: r2p -i 28 -o 2 -x 1 -p 7 -n 55)
    (28, 2, 1, 7, 55), (26, 1, 2, 24, 16), (26, 2, 2, 24, 16), (25, 2, 2, 24, 9),
]


@pytest.mark.parametrize("mode", [ca.R2P, ca.SR2P])
@pytest.mark.parametrize("args", POL_LJ_CASES)
def test_r2p_left_justified_form(args, mode):
    """topolar_lj (x / y left-justified by 30 bits, multipliers +/-2^30 off the
    sign bit, phase accumulated at the same scale) against the oracle AND
    against the plain unrolled kernel (CORDIC_FLAG_NO_LJ), on random vectors,
    the axes / diagonals / extremes and vectors a few LSBs long."""
    try:
        cfg, ocfg = both(mode, *args)
    except ca.CordicError:
        pytest.skip("core refused (sequential corner case)")
    if cfg.needs_wrap:
        pytest.skip("registers can overflow: the explicit-wrap kernels serve this core")
    assert cfg.ww <= 34 and cfg.nlive >= 1
    rng = np.random.RandomState(7)
    n = (1 << 17) + 3
    lim = 1 << (cfg.iw - 1)
    x = rng.randint(-lim, lim, size=n).astype(np.int32)
    y = rng.randint(-lim, lim, size=n).astype(np.int32)
    sp = np.array([0, 1, -1, lim - 1, -lim, lim // 2, -(lim // 2), 2, -2, 3],
                  dtype=np.int32)
    gx, gy = np.meshgrid(sp, sp)
    x[:100], y[:100] = gx.ravel(), gy.ravel()
    small = rng.randint(-8, 8, size=(2, 4096)).astype(np.int32)
    x[100:4196], y[100:4196] = small[0], small[1]
    # |y| == x after the fold (the 45 degree edge of the convergence range)
    x[4196:8292] = rng.randint(-lim, lim, size=4096)
    y[4196:8292] = 0
    m, p = gpu_r2p(cfg, x, y)
    rm, rp = O.topolar(ocfg, x, y)
    assert np.array_equal(m, rm) and np.array_equal(p, rp)
    m2, p2 = gpu_r2p(cfg.with_flags(ca.FLAG_NO_LJ), x, y)
    assert np.array_equal(m, m2) and np.array_equal(p, p2)


POL_LJW_CASES = [  # (iw, ow, xtra, pw, nstages): WW 35 .. 40, run-time justification
    (27, 27, 2, 32, 20), (28, 28, 2, 32, 20), (29, 29, 2, 32, 24),
    (30, 30, 2, 32, 18), (31, 31, 1, 32, 22), (32, 32, 2, 32, 20),
    (32, 32, 2, 32, 32), (32, 32, 0, 32, 16), (32, 16, 2, 32, 12),
    (32, 32, 2, 32, 1), (32, 32, 2, 32, 2), (32, 32, 2, 32, 7),
    (32, 32, 2, 32, 8), (32, 32, 2, 32, 9), (30, 2, 2, 24, 16),
    (32, 1, 2, 16, 11), (32, 32, 2, 32, 45), (29, 32, 1, 30, 26),
]


@pytest.mark.parametrize("mode", [ca.R2P, ca.SR2P])
@pytest.mark.parametrize("args", POL_LJW_CASES)
def test_r2p_wide_left_justified_form(args, mode):
    """topolar_ljw (WW 35..40: justification 64 - WW at run time, early stages
    as extra multiply-adds) against the oracle and against the round-1 wide
    kernels (CORDIC_FLAG_NO_LJ)."""
    try:
        cfg, ocfg = both(mode, *args)
    except ca.CordicError:
        pytest.skip("core refused (sequential corner case)")
    if cfg.needs_wrap:
        pytest.skip("registers can overflow: the explicit-wrap kernels serve this core")
    assert 35 <= cfg.ww <= 40 and cfg.nlive >= 1
    rng = np.random.RandomState(11)
    n = (1 << 17) + 3
    lim = 1 << (cfg.iw - 1)
    x = rng.randint(-lim, lim, size=n, dtype=np.int64).astype(np.int32)
    y = rng.randint(-lim, lim, size=n, dtype=np.int64).astype(np.int32)
    sp = np.array([0, 1, -1, lim - 1, -lim, lim // 2, -(lim // 2), 2, -2, 3],
                  dtype=np.int64).astype(np.int32)
    gx, gy = np.meshgrid(sp, sp)
    x[:100], y[:100] = gx.ravel(), gy.ravel()
    small = rng.randint(-8, 8, size=(2, 4096)).astype(np.int32)
    x[100:4196], y[100:4196] = small[0], small[1]
    x[4196:8292] = rng.randint(-lim, lim, size=4096, dtype=np.int64).astype(np.int32)
    y[4196:8292] = 0
    m, p = gpu_r2p(cfg, x, y)
    rm, rp = O.topolar(ocfg, x, y)
    assert np.array_equal(m, rm) and np.array_equal(p, rp)
    m2, p2 = gpu_r2p(cfg.with_flags(ca.FLAG_NO_LJ), x, y)
    assert np.array_equal(m, m2) and np.array_equal(p, p2)


def test_exhaustive_r2p_checked_in_core_and_bench_criteria():
    cfg, ocfg = both(ca.R2P, 13, 13, 2)
    x, y, mg = Q.r2p_bench_inputs(cfg.iw, cfg.pw)
    gm, gp = gpu_r2p(cfg, x, y)
    rm, rp = O.topolar(ocfg, x, y)
    assert np.array_equal(gm, rm) and np.array_equal(gp, rp)
    assert Q.r2p_quality(cfg, x, y, mg, gm, gp)["ok"]


def test_exhaustive_small_core_all_inputs():
    """IW=6: every (x, y, phase) triple of a small core, 2^6*2^6*2^9."""
    cfg, ocfg = both(ca.P2R, 6, 6, 2, 9, -1)
    v = np.arange(-32, 32, dtype=np.int32)
    ph = np.arange(1 << 9, dtype=np.uint32)
    X, Y, P = np.meshgrid(v, v, ph, indexing="ij")
    assert_p2r(cfg, ocfg, X.ravel(), Y.ravel(), P.ravel().astype(np.uint32))
    cfg, ocfg = both(ca.R2P, 9, 9, 2)
    v = np.arange(-256, 256, dtype=np.int32)
    X, Y = np.meshgrid(v, v, indexing="ij")
    assert_r2p(cfg, ocfg, X.ravel(), Y.ravel())


# ---------------------------------------------------------- config sweeps

def test_parameter_sweep_all_kernel_paths():
    """Random parameter sets: unrolled 32- and 64-bit kernels, the generic
    kernel (stage counts without an unrolled instance) and cores that need
    explicit WW-bit wrapping."""
    rng = np.random.RandomState(5)
    seen = {"wrap": 0, "wide": 0, "narrow": 0}
    for trial in range(80):
        mode = int(rng.randint(4))
        iw, ow = int(rng.randint(1, 33)), int(rng.randint(1, 33))
        xtra = int(rng.randint(0, 6))
        pw = int(rng.randint(3, 33))
        ns = int(rng.randint(1, 41))
        try:
            cfg = ca.Config.from_cli(mode, iw, ow, xtra, pw, ns)
        except ca.CordicError:
            with pytest.raises(ValueError):
                O.config_cli(mode, iw, ow, xtra, pw, ns)
            continue
        ocfg = O.config_cli(mode, iw, ow, xtra, pw, ns)
        seen["wrap"] += cfg.needs_wrap
        seen["wide" if cfg.ww > 32 else "narrow"] += 1
        x, y, ph = rand_inputs(rng, iw, pw, 4099)
        if mode in (ca.P2R, ca.SP2R):
            assert_p2r(cfg, ocfg, x, y, ph)
        else:
            assert_r2p(cfg, ocfg, x, y)
    assert seen["wrap"] >= 2 and seen["wide"] >= 10 and seen["narrow"] >= 10


@pytest.mark.parametrize("kind", ["lj29", "lj30", "narrow", "wide", "r2p",
                                  "r2p_wide", "seq"])
def test_every_stage_count(kind):
    """Stage counts 1..40: static instances where they exist, the dynamic-exit
    instance elsewhere (plain and seeded), the generic kernel beyond."""
    rng = np.random.RandomState(31)
    for ns in list(range(1, 41)) + [47, 64]:
        if kind == "lj29":
            args = (ca.P2R, 32, 32, 2, 32, ns)
        elif kind == "lj30":
            args = (ca.P2R, 31, 30, 2, 32, ns)
        elif kind == "narrow":
            args = (ca.P2R, 24, 24, 2, 32, ns)
        elif kind == "wide":
            args = (ca.P2R, 32, 32, 7, 32, ns)
        elif kind == "r2p":
            args = (ca.R2P, 24, 24, 2, 32, ns)
        elif kind == "r2p_wide":
            args = (ca.R2P, 32, 32, 2, 32, ns)
        else:
            args = (ca.SP2R, 32, 32, 2, 32, ns)
        try:
            cfg, ocfg = both(*args)
        except ca.CordicError:
            continue
        x, y, ph = rand_inputs(rng, cfg.iw, cfg.pw, 3001)
        if args[0] in (ca.P2R, ca.SP2R):
            assert_p2r(cfg, ocfg, x, y, ph)
            plan = ca.Plan(cfg)
            x0 = (1 << (cfg.iw - 1)) - 1
            gx, gy = gpu_plan_p2r(plan, x0, -5, ph)
            rx, ry = O.rotate(ocfg, x0, -5, ph)
            assert np.array_equal(gx, rx) and np.array_equal(gy, ry), ns
            plan.close()
        else:
            assert_r2p(cfg, ocfg, x, y)


def test_tiny_cores_wrap_like_the_registers():
    """WW of a few bits: truncation noise reaches full scale and the WW-bit
    registers really overflow; the kernel must wrap exactly as they do."""
    rng = np.random.RandomState(6)
    hit = 0
    for iw, ow, nx, pw, ns in [(1, 1, 1, 8, 6), (2, 1, 1, 6, 5), (1, 2, 1, 5, 8),
                               (2, 2, 2, 7, 7), (3, 3, 1, 9, 12),
                               (1, 1, 2, 3, 2)]:
        for mode in (ca.P2R, ca.R2P):
            try:
                cfg = ca.Config.from_core(mode, ns, iw, ow, nx, pw)
            except ca.CordicError:
                continue
            ocfg = O.config_core(mode, ns, iw, ow, nx, pw)
            hit += cfg.needs_wrap
            x, y, ph = rand_inputs(rng, iw, pw, 2048)
            if mode == ca.P2R:
                assert_p2r(cfg, ocfg, x, y, ph)
            else:
                assert_r2p(cfg, ocfg, x, y)
    assert hit >= 3


def test_left_justified_kernel_equals_right_justified_kernel():
    """WW 33..40: the LJ kernels (early stages as extra multiply-adds; WW 36..40
    with the justification 64 - WW and the phase at 2^(LJ+2)) against the plain
    64-bit unrolled kernel and the oracle: per-sample vectors, constant vectors
    with and without the seed table, the fused NCO."""
    for args in [(ca.P2R, 32, 32, 2, 32, 16), (ca.P2R, 32, 32, 2, 32, 24),
                 (ca.SP2R, 32, 32, 2, 32, 16), (ca.P2R, 30, 30, 2, 32, 16),
                 (ca.P2R, 31, 31, 2, 30, 20),
                 (ca.P2R, 32, 32, 3, 32, 16), (ca.P2R, 32, 32, 4, 32, 20),
                 (ca.P2R, 32, 32, 5, 32, 24), (ca.P2R, 32, 32, 6, 32, 20),
                 (ca.P2R, 32, 32, 7, 32, 20), (ca.SP2R, 32, 32, 7, 32, 32),
                 (ca.P2R, 32, 24, 7, 32, 1), (ca.P2R, 32, 32, 7, 32, 2),
                 (ca.P2R, 32, 32, 7, 32, 7), (ca.P2R, 32, 32, 7, 32, 8),
                 (ca.P2R, 30, 32, 6, 28, 19), (ca.P2R, 32, 8, 4, 24, 12),
                 (ca.P2R, 32, 32, 7, 32, 44),
                 # WW <= 32: left-justified by 30 against the 32-bit container
                 # (static instances with and without direction tails)
                 (ca.P2R, 24, 24, 2, -1, -1), (ca.P2R, 16, 16, 2, -1, -1),
                 (ca.SP2R, 20, 20, 2, -1, -1), (ca.P2R, 13, 13, 2, -1, -1),
                 (ca.P2R, 12, 20, 1, 17, 15), (ca.P2R, 29, 29, 2, 32, 26)]:
        cfg, ocfg = both(*args)
        rj = cfg.with_flags(ca.FLAG_NO_LJ)
        rng = np.random.RandomState(18)
        x, y, ph = rand_inputs(rng, cfg.iw, cfg.pw, 50001)
        a = gpu_p2r(cfg, x, y, ph)
        b = gpu_p2r(rj, x, y, ph)
        c = O.rotate(ocfg, x, y, ph)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), args
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]), args
        lim = 1 << (cfg.iw - 1)
        for x0, y0 in ((lim - 1, 0), (-lim, -lim), (12345 % lim, -(777 % lim))):
            for flags in (0, ca.FLAG_NO_SEED, ca.FLAG_NO_LJ):
                plan = ca.Plan(cfg.with_flags(flags))
                a = gpu_plan_p2r(plan, x0, y0, ph)
                c = O.rotate(ocfg, x0, y0, ph)
                assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]), args
                a = gpu_plan_nco(plan, 30001, 0x1234, 0x01234567, 1 << 33, x0, y0)
                c = O.nco(ocfg, 30001, 0x1234, 0x01234567, 1 << 33, x0, y0)
                assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]), args
                plan.close()


def test_generic_kernel_equals_unrolled_kernel():
    for args in [(ca.P2R, 32, 32, 2, 32, 16), (ca.P2R, 16, 16, 2, 16, 16),
                 (ca.R2P, 24, 24, 2, -1, 20)]:
        cfg, ocfg = both(*args)
        gen = cfg.with_flags(ca.FLAG_FORCE_GENERIC)
        rng = np.random.RandomState(8)
        x, y, ph = rand_inputs(rng, cfg.iw, cfg.pw, 70001)
        if args[0] == ca.P2R:
            a = gpu_p2r(cfg, x, y, ph)
            b = gpu_p2r(gen, x, y, ph)
        else:
            a = gpu_r2p(cfg, x, y)
            b = gpu_r2p(gen, x, y)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ------------------------------------------------------------- edge cases

@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 7, 255, 1023, 1024, 1025,
                               4096 * 3 + 2])
def test_ragged_sizes(n):
    rng = np.random.RandomState(n + 1)
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    x, y, ph = rand_inputs(rng, 32, 32, n)
    assert_p2r(cfg, ocfg, x, y, ph)
    assert_p2r(cfg, ocfg, 12345, -777, ph)
    cfg, ocfg = both(ca.R2P, 24, 24, 2, -1, 20)
    x, y, _ = rand_inputs(rng, 24, 32, n)
    assert_r2p(cfg, ocfg, x, y)


@pytest.mark.parametrize("offset", [1, 2, 3])
def test_unaligned_buffers(offset):
    rng = np.random.RandomState(offset)
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    x, y, ph = rand_inputs(rng, 32, 32, 5000)
    assert_p2r(cfg, ocfg, x, y, ph, offset=offset)
    cfg, ocfg = both(ca.R2P, 24, 24, 2, -1, 20)
    x, y, _ = rand_inputs(rng, 24, 32, 5000)
    assert_r2p(cfg, ocfg, x, y, offset=offset)


def test_inputs_are_taken_modulo_their_port_width():
    """The ports are IW / PW bits wide (rtl/cordic.v:60-61): bits above them
    in the 32-bit containers must not matter."""
    rng = np.random.RandomState(9)
    cfg, ocfg = both(ca.P2R, 13, 13, 2)
    n = 10000
    x, y, ph = rand_inputs(rng, 13, 20, n)
    junk = rng.randint(0, 1 << 19, n).astype(np.int32)
    xj = (x & 0x1fff) | (junk << 13)
    yj = (y & 0x1fff) | ((junk ^ 0x5555) << 13)
    phj = (ph & np.uint32(0xfffff)) | (junk.astype(np.uint32) << 20)
    a = gpu_p2r(cfg, xj, yj, phj)
    b = O.rotate(ocfg, x, y, ph & np.uint32(0xfffff))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    cfg, ocfg = both(ca.R2P, 13, 13, 2)
    a = gpu_r2p(cfg, xj, yj)
    b = O.topolar(ocfg, x, y)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_host_buffer_entry_points():
    rng = np.random.RandomState(10)
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    x, y, ph = rand_inputs(rng, 32, 32, 33333)
    a = ca.p2r_host(cfg, x, y, ph)
    b = O.rotate(ocfg, x, y, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a = ca.p2r_host(cfg, 2**31 - 1, 0, ph)
    b = O.rotate(ocfg, 2**31 - 1, 0, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    cfg, ocfg = both(ca.R2P, 24, 24, 2, -1, 20)
    x, y, _ = rand_inputs(rng, 24, 32, 33333)
    a = ca.r2p_host(cfg, x, y)
    b = O.topolar(ocfg, x, y)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_wrong_mode_is_refused_on_device_calls():
    cfg = ca.Config.from_cli(ca.R2P, 13, 13, 2)
    t = torch.zeros(16, dtype=torch.int32, device=DEV)
    with pytest.raises(ca.CordicError):
        ca.p2r_const(cfg, 1, 0, t, t, t)
    cfg = ca.Config.from_cli(ca.P2R, 13, 13, 2)
    with pytest.raises(ca.CordicError):
        ca.r2p(cfg, t, t, t, t)


# --------------------------------------------------------------------- NCO

@pytest.mark.parametrize("args", [(ca.P2R, 32, 32, 2, 32, 16),
                                  (ca.SP2R, 32, 32, 2, 32, 16),
                                  (ca.P2R, 13, 13, 2, -1, -1),
                                  (ca.P2R, 16, 16, 2, 16, 16)])
def test_nco_equals_oracle(args):
    cfg, ocfg = both(*args)
    x0 = (1 << (cfg.iw - 1)) - 1
    for n, phase0, fcw, index0 in [(100003, 0, 0x01234567, 0),
                                   (4097, 0xdeadbeef, 0x9e3779b9, 12345),
                                   (65536, 5, 1, (7 << 32) + 99)]:
        a = gpu_nco(cfg, n, phase0, fcw, index0, x0, 0)
        b = O.nco(ocfg, n, phase0, fcw, index0, x0, 0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ------------------------------------------------------------ input fills

def test_fill_kernels_and_digest_twin():
    n = 100000
    t = torch.empty(n, dtype=torch.int32, device=DEV)
    ca.fill_phase_ramp(t, (3 << 32) + 17, 2)
    torch.cuda.synchronize()
    g = np.arange(n, dtype=np.uint64) + np.uint64((3 << 32) + 17)
    assert np.array_equal(to_np(t, np.uint32),
                          ((g << np.uint64(2)) & np.uint64(0xffffffff))
                          .astype(np.uint32))
    x = torch.empty(n, dtype=torch.int32, device=DEV)
    y = torch.empty(n, dtype=torch.int32, device=DEV)
    ca.fill_iq_ramp(x, y, 5, 0x9E3779B1, 0x85EBCA77, 24)
    torch.cuda.synchronize()
    g = (np.arange(n, dtype=np.uint64) + np.uint64(5)) & np.uint64(0xffffffff)
    ex = ((((g * np.uint64(0x9E3779B1)) & np.uint64(0xffffffff))
           >> np.uint64(8)).astype(np.uint32) << 8).astype(np.int32) >> 8
    assert np.array_equal(to_np(x), ex)
    assert gpu_digest(x, index0=77) == cpu_digest(to_np(x), index0=77)
    # digests of shards add up to the digest of the whole
    whole = gpu_digest(x, 0)
    parts = (gpu_digest(x[:40000], 0) + gpu_digest(x[40000:], 40000)) % 2**64
    assert whole == parts


# ------------------------------------------------- plans / seeded kernels

SEED_CASES = {
    "cfg2": (ca.P2R, 32, 32, 2, 32, 16),
    "cfg4": (ca.P2R, 32, 32, 2, 32, 24),
    "cfg5_seq": (ca.SP2R, 32, 32, 2, 32, 16),
    "rtl_cordic": (ca.P2R, 13, 13, 2, -1, -1),
    "rtl_seqcordic": (ca.SP2R, 13, 13, 2, -1, -1),
    "cfg1": (ca.P2R, 16, 16, 2, 16, 16),
    "ww33": (ca.P2R, 30, 30, 2, 32, 16),
    "ww34": (ca.P2R, 31, 31, 2, 30, 20),
    "ww32": (ca.P2R, 29, 29, 2, 32, 18),
    "pw14": (ca.P2R, 12, 12, 2, 14, 14),
}


def breakpoint_phases(cfg):
    """Every phase within +/-2 LSB of a leaf boundary, in all four octant
    pairs, as PW-bit phase values."""
    words = ca.seed_table(cfg)
    m, S, nb, L = (int(v) for v in words[:4])
    buckets = words[4:4 + nb * 2].reshape(nb, 2).astype(np.int64)
    lsb = 1 << (32 - cfg.pw)
    bounds = buckets[:, 0]
    bounds = bounds[bounds != 0x7fffffff] + 1           # r-domain boundaries
    starts = (np.arange(nb, dtype=np.int64) << S)        # bucket edges too
    r = np.concatenate([bounds, starts, [0, (1 << 30) - lsb]])
    r = np.concatenate([r + d * lsb for d in (-2, -1, 0, 1, 2)])
    r = r[(r >= 0) & (r < (1 << 30))]
    out = []
    for q in range(4):
        P = (r - (1 << 29) + (q << 30)) & 0xffffffff
        out.append(P >> (32 - cfg.pw))
    return np.concatenate(out).astype(np.uint32)


@pytest.mark.parametrize("name", sorted(SEED_CASES))
def test_seeded_plan_is_bit_exact(name):
    cfg, ocfg = both(*SEED_CASES[name])
    plan = ca.Plan(cfg)
    info = plan.seed_info
    assert info["stages"] == 11 and info["nleaves"] >= 100
    rng = np.random.RandomState(21)
    n = (1 << 19) + 5
    _, _, ph = rand_inputs(rng, cfg.iw, cfg.pw, n)
    ph = np.concatenate([ph, breakpoint_phases(cfg)])
    lo, hi = -(1 << (cfg.iw - 1)), (1 << (cfg.iw - 1)) - 1
    for x0, y0 in [(hi, 0), (lo, lo), (0, hi), (-12345 % (hi + 1), 777 % hi),
                   (hi, lo), (0, 0), (1, -1)]:
        gx, gy = gpu_plan_p2r(plan, x0, y0, ph)
        rx, ry = O.rotate(ocfg, x0, y0, ph)
        assert np.array_equal(gx, rx) and np.array_equal(gy, ry), (x0, y0)
    # and it really is the seeded kernel: same results with it switched off
    plain = ca.Plan(cfg.with_flags(ca.FLAG_NO_SEED))
    a = gpu_plan_p2r(plan, hi, 0, ph)
    b = gpu_plan_p2r(plain, hi, 0, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    plan.close()
    plain.close()


def test_seeded_plan_exhaustive_checked_in_core():
    """All 2^20 phases of rtl/cordic.v through the seeded kernel."""
    cfg, ocfg = both(ca.P2R, 13, 13, 2)
    plan = ca.Plan(cfg)
    ph, x0, y0 = Q.p2r_bench_inputs(cfg.iw, cfg.pw)
    gx, gy = gpu_plan_p2r(plan, x0, y0, ph)
    rx, ry = O.rotate(ocfg, x0, y0, ph)
    assert np.array_equal(gx, rx) and np.array_equal(gy, ry)
    assert Q.p2r_quality(cfg, ph, x0, y0, gx, gy)["ok"]


@pytest.mark.parametrize("args", [(ca.P2R, 32, 32, 2, 32, 16),
                                  (ca.SP2R, 32, 32, 2, 32, 16),
                                  (ca.P2R, 13, 13, 2, -1, -1)])
def test_seeded_nco(args):
    cfg, ocfg = both(*args)
    plan = ca.Plan(cfg)
    x0 = (1 << (cfg.iw - 1)) - 1
    for n, phase0, fcw, index0 in [(200003, 0, 0x01234567, 0),
                                   (4097, 0xdeadbeef, 0x9e3779b9, 12345),
                                   (65536, 5, 1, (7 << 32) + 99)]:
        a = gpu_plan_nco(plan, n, phase0, fcw, index0, x0, 0)
        b = O.nco(ocfg, n, phase0, fcw, index0, x0, 0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_plan_for_ineligible_core_still_works():
    """r2p / wide cores: the plan carries no table and runs the plain path."""
    cfg, ocfg = both(ca.P2R, 32, 32, 3, 32, 16)          # WW 36
    plan = ca.Plan(cfg)
    assert plan.seed_info["stages"] == 0
    rng = np.random.RandomState(3)
    _, _, ph = rand_inputs(rng, 32, 32, 30001)
    a = gpu_plan_p2r(plan, 2**31 - 1, 0, ph)
    b = O.rotate(ocfg, 2**31 - 1, 0, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_plan_for_degenerate_angle_table():
    """sp2r -i 9 -o 31 -x 3 -p 3 -n 33 (found by tools/fuzz_gpu.py): with a
    3-bit phase every angle after the first is zero, the residual after the
    seed stages does not shrink, and the core must not be seeded."""
    cfg, ocfg = both(ca.SP2R, 9, 31, 3, 3, 33)
    plan = ca.Plan(cfg)
    assert plan.seed_info["stages"] == 0
    rng = np.random.RandomState(4)
    _, _, ph = rand_inputs(rng, 9, 3, 20003)
    for x0, y0 in ((55, -256), (255, 0), (-256, -256)):
        a = gpu_plan_p2r(plan, x0, y0, ph)
        b = O.rotate(ocfg, x0, y0, ph)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ------------------------------------------------ BASELINE sizes, properties

def _digest2(a, b, index0=0):
    return (gpu_digest(a, index0) + gpu_digest(b, index0 + (1 << 40))) % 2**64


def _full_size_p2r(nstages, shift):
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, nstages)
    n = 1 << 30
    x0 = 2**31 - 1
    phase = torch.empty(n, dtype=torch.int32, device=DEV)
    a = torch.empty(n, dtype=torch.int32, device=DEV)
    b = torch.empty(n, dtype=torch.int32, device=DEV)
    ca.fill_phase_ramp(phase, 0, shift)
    plan = ca.Plan(cfg)
    plan.p2r_const(x0, 0, phase, a, b)
    torch.cuda.synchronize()
    assert ca.last_kernel() == ca.KERNEL_SEEDED
    whole = _digest2(a, b)
    # PRIMARY: every one of the 2^30 output pairs against the oracle (threaded
    # scalar restatement of rtl/cordic.v:131-188,231-314, condensed by the same
    # position-aware digest; oracle/cordic_oracle.c: orc_digest)
    want, _ = O.job_digest(ocfg, "p2r", 0, n, 0, 1 << shift, x0, 0)
    assert whole == want, ("%016x" % whole, "%016x" % want)
    # same work as 64 shards through the full-recurrence kernel
    plain = ca.Plan(cfg.with_flags(ca.FLAG_NO_SEED))
    a2 = torch.empty(1 << 24, dtype=torch.int32, device=DEV)
    b2 = torch.empty(1 << 24, dtype=torch.int32, device=DEV)
    parts = 0
    for s in range(64):
        lo = s << 24
        plain.p2r_const(x0, 0, phase[lo:lo + (1 << 24)], a2, b2)
        torch.cuda.synchronize()
        parts = (parts + _digest2(a2, b2, lo)) % 2**64
    assert whole == parts
    idx = np.unique(np.concatenate([np.arange(4096), np.arange(n - 4096, n),
                                    np.arange(0, n, 65521)])).astype(np.int64)
    ti = torch.from_numpy(idx).to(DEV)
    rx, ry = O.rotate(ocfg, x0, 0, ((idx << shift) & 0xffffffff).astype(np.uint32))
    assert np.array_equal(a[ti].cpu().numpy(), rx)
    assert np.array_equal(b[ti].cpu().numpy(), ry)
    # linearity in the index: the NCO with fcw = 2^shift regenerates the same ramp
    a.zero_(); b.zero_()
    plan.nco(n, 0, 1 << shift, 0, x0, 0, a, b)
    torch.cuda.synchronize()
    assert _digest2(a, b) == whole
    # and the seeds with the phase recurrence behind them (no direction tails)
    a.zero_(); b.zero_()
    ca.Plan(cfg.with_flags(ca.FLAG_NO_TAILS)).p2r_const(x0, 0, phase, a, b)
    torch.cuda.synchronize()
    assert _digest2(a, b) == whole


def test_full_size_cfg2_checksum_of_checksums_and_spot_check():
    """2^30 samples (BASELINE config 2): one launch vs 64 launches of 2^24
    with the shard's global index -- digests must add up -- plus a strided
    subset and both ends against the oracle."""
    _full_size_p2r(16, 2)


def test_full_size_cfg4_direction_tails_checksum_of_checksums_and_spot_check():
    """The same for one GPU's share of BASELINE config 4 (24 stages, phase =
    index): the seeded kernel with its direction tails (array feed and NCO
    instance) against the full recurrence and the oracle."""
    _full_size_p2r(24, 0)


def test_full_size_cfg5_nco_4g_samples():
    """2^32 samples (BASELINE config 5), store only: the fused NCO against the
    phase-array path in four 2^30 quarters, and oracle spot checks across the
    32-bit index wrap."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    n = 1 << 32
    x0, fcw = 2**31 - 1, 0x01234567
    a = torch.empty(n, dtype=torch.int32, device=DEV)
    b = torch.empty(n, dtype=torch.int32, device=DEV)
    plan = ca.Plan(cfg)
    plan.nco(n, 0, fcw, 0, x0, 0, a, b)
    torch.cuda.synchronize()
    # PRIMARY: all 2^32 output pairs against the oracle
    want, _ = O.job_digest(ocfg, "nco", 0, n, 0, fcw, x0, 0)
    got = _digest2(a, b)
    assert got == want, ("%016x" % got, "%016x" % want)
    q = 1 << 30
    a2 = torch.empty(q, dtype=torch.int32, device=DEV)
    b2 = torch.empty(q, dtype=torch.int32, device=DEV)
    for s in range(4):
        ca.nco(cfg.with_flags(ca.FLAG_NO_SEED), q, 0, fcw, s * q, x0, 0, a2, b2)
        torch.cuda.synchronize()
        assert _digest2(a2, b2, s * q) == _digest2(a[s * q:(s + 1) * q],
                                                   b[s * q:(s + 1) * q], s * q)
    idx = np.unique(np.concatenate([np.arange(2048), np.arange(n - 2048, n),
                                    np.arange(0, n, 1048573)])).astype(np.int64)
    ti = torch.from_numpy(idx).to(DEV)
    ph = ((idx.astype(np.uint64) * np.uint64(fcw)) & np.uint64(0xffffffff))
    rx, ry = O.rotate(ocfg, x0, 0, ph.astype(np.uint32))
    assert np.array_equal(a[ti].cpu().numpy(), rx)
    assert np.array_equal(b[ti].cpu().numpy(), ry)


def test_full_size_cfg3_r2p_round_trip_and_spot_check():
    """2^30 I/Q pairs (BASELINE config 3): oracle spot check and the
    polar -> rect round trip of the result (phase in, angle out)."""
    cfg, ocfg = both(ca.R2P, 24, 24, 2, -1, 20)
    n = 1 << 30
    x = torch.empty(n, dtype=torch.int32, device=DEV)
    y = torch.empty(n, dtype=torch.int32, device=DEV)
    mag = torch.empty(n, dtype=torch.int32, device=DEV)
    ph = torch.empty(n, dtype=torch.int32, device=DEV)
    ca.fill_iq_ramp(x, y, 0, 0x9E3779B1, 0x85EBCA77, 24)
    ca.r2p(cfg, x, y, mag, ph)
    torch.cuda.synchronize()
    # PRIMARY: all 2^30 (magnitude, phase) pairs against the oracle
    # (rtl/topolar.v:122-152,195-271), which regenerates the I/Q ramps itself
    want, _ = O.job_digest(ocfg, "r2p", 0, n)
    got = _digest2(mag, ph)
    assert got == want, ("%016x" % got, "%016x" % want)
    idx = np.unique(np.concatenate([np.arange(4096), np.arange(n - 4096, n),
                                    np.arange(0, n, 65521)])).astype(np.int64)
    ti = torch.from_numpy(idx).to(DEV)
    rm, rp = O.topolar(ocfg, x[ti].cpu().numpy(), y[ti].cpu().numpy())
    assert np.array_equal(mag[ti].cpu().numpy(), rm)
    assert np.array_equal(ph[ti].cpu().numpy().view(np.uint32), rp)
    # launch-geometry independence: 16 shards of 2^26 give the same digests
    whole = _digest2(mag, ph)
    m2 = torch.empty(1 << 26, dtype=torch.int32, device=DEV)
    p2 = torch.empty(1 << 26, dtype=torch.int32, device=DEV)
    parts = 0
    for s in range(16):
        lo = s << 26
        ca.r2p(cfg, x[lo:lo + (1 << 26)], y[lo:lo + (1 << 26)], m2, p2)
        torch.cuda.synchronize()
        parts = (parts + _digest2(m2, p2, lo)) % 2**64
    assert whole == parts
    # round trip: rotating (mag, 0) by the measured angle points back at (x, y)
    rot = ca.Config.from_cli(ca.P2R, 24, 24, 2, 32, 20)
    ox = torch.empty(1 << 20, dtype=torch.int32, device=DEV)
    oy = torch.empty(1 << 20, dtype=torch.int32, device=DEV)
    zero = torch.zeros(1 << 20, dtype=torch.int32, device=DEV)
    ca.p2r(rot, mag[:1 << 20].contiguous(), zero, ph[:1 << 20].contiguous(),
           ox, oy)
    torch.cuda.synchronize()
    xs = x[:1 << 20].cpu().numpy().astype(np.float64)
    ys = y[:1 << 20].cpu().numpy().astype(np.float64)
    # gains: r2p 0.8234 * 2^(OW-IW-1), p2r 1.1644 * 2^(OW-IW-1)
    g = cfg.gain * 0.5 * rot.gain * 0.5
    err = np.hypot(ox.cpu().numpy() - xs * g, oy.cpu().numpy() - ys * g)
    # statistical property, not a parity check: two quantised conversions in
    # a row stay within a few output LSBs of the ideal (values are ~2^21)
    assert err.max() < 16.0


def test_entry_points_are_stream_ordered_and_graph_capturable():
    """The device entry points only enqueue on the given stream: they can be
    captured into a HIP graph and replayed on new data."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    rcfg, rocfg = both(ca.R2P, 24, 24, 2, -1, 20)
    plan = ca.Plan(cfg)
    n = 1 << 18
    rng = np.random.RandomState(4)
    phase = torch.zeros(n, dtype=torch.int32, device=DEV)
    ox = torch.zeros(n, dtype=torch.int32, device=DEV)
    oy = torch.zeros(n, dtype=torch.int32, device=DEV)
    mag = torch.zeros(n, dtype=torch.int32, device=DEV)
    ang = torch.zeros(n, dtype=torch.int32, device=DEV)
    x0 = 2**31 - 1
    # warm up outside the capture (first call queries the device once)
    plan.p2r_const(x0, 0, phase, ox, oy)
    ca.r2p(rcfg, ox, oy, mag, ang)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.p2r_const(x0, 0, phase, ox, oy)     # sin/cos ...
        ca.r2p(rcfg, ox, oy, mag, ang)           # ... and back to polar
    for trial in range(3):
        ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        phase.copy_(dev_i32(ph))
        g.replay()
        torch.cuda.synchronize()
        rx, ry = O.rotate(ocfg, x0, 0, ph)
        assert np.array_equal(to_np(ox), rx) and np.array_equal(to_np(oy), ry)
        rm, rp = O.topolar(rocfg, rx, ry)
        assert np.array_equal(to_np(mag), rm)
        assert np.array_equal(to_np(ang, np.uint32), rp)


# ------------------------------------------------ fused gain annihilation

@pytest.mark.parametrize("mode,iw,ow,pw,ns,flags", [
    (ca.P2R, 32, 32, 32, 16, 0),                 # seeded / LJ unrolled
    (ca.P2R, 13, 13, -1, -1, 0),                 # narrow
    (ca.P2R, 32, 32, 32, 19, 0),                 # dynamic-exit instance
    (ca.SP2R, 13, 13, -1, -1, 0),                # padded-table constant
    (ca.P2R, 13, 13, -1, -1, ca.FLAG_FORCE_GENERIC),
    (ca.R2P, 24, 24, -1, 20, 0),
    (ca.R2P, 13, 13, -1, -1, ca.FLAG_FORCE_GENERIC),
])
def test_unit_gain_flag_is_output_times_k_shift_32(mode, iw, ow, pw, ns, flags):
    """CORDIC_FLAG_UNIT_GAIN: every output is (o * K) >> 32 with the K the
    generator prints into the core (sw/cordiclib.cpp:205-209)."""
    base, ocfg = both(mode, iw, ow, 2, pw, ns)
    cfg = base.with_flags(flags | ca.FLAG_UNIT_GAIN)
    k = ca.lib().cordic_config_gain_annihilator(cfg.ref)
    assert 0xd0000000 < k < 0xe0000000
    rng = np.random.RandomState(21)
    n = (1 << 16) + 3
    x, y, ph = rand_inputs(rng, cfg.iw, cfg.pw, n)

    def scaled(a):
        return ((a.astype(np.int64) * k) >> 32).astype(np.int32)
    if mode in (ca.P2R, ca.SP2R):
        gx, gy = gpu_p2r(cfg, x, y, ph)
        rx, ry = O.rotate(ocfg, x, y, ph)
        assert np.array_equal(gx, scaled(rx)) and np.array_equal(gy, scaled(ry))
        x0 = (1 << (cfg.iw - 1)) - 1
        gx, gy = gpu_plan_p2r(ca.Plan(cfg), x0, 0, ph)
        rx, ry = O.rotate(ocfg, x0, 0, ph)
        assert np.array_equal(gx, scaled(rx)) and np.array_equal(gy, scaled(ry))
        # amplitude really is ~ unity gain now: |out| ~ x0 * 2^(OW-IW-1)... * 1
        if mode == ca.P2R:
            mag = np.hypot(gx.astype(float), gy.astype(float))
            want = x0 * 2.0 ** (cfg.ow - cfg.iw - 1) * \
                (ca.lib().cordic_gain(cfg.nstages) * k / 2.0 ** 32)
            assert abs(mag.mean() / want - 1) < 1e-3
            assert abs(ca.lib().cordic_gain(cfg.nstages) * k / 2.0 ** 32 - 1) < 1e-6
    else:
        gm, gp = gpu_r2p(cfg, x, y)
        rm, rp = O.topolar(ocfg, x, y)
        assert np.array_equal(gm, scaled(rm)) and np.array_equal(gp, rp)


def test_seeded_plan_in_a_hip_graph_and_on_two_streams():
    """The tile queue of the seeded kernel takes a fresh counter block per
    launch and zeroes it on the job's stream: (a) a launch captured into a HIP
    graph replays correctly any number of times, (b) launches of ONE plan that
    overlap on two streams never share counters."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    n = (1 << 20) + 4096 * 3 + 8
    rng = np.random.RandomState(3)
    ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    x0 = (1 << 31) - 1
    rx, ry = O.rotate(ocfg, x0, 0, ph)
    dph = dev_i32(ph)
    ox = torch.zeros(n, dtype=torch.int32, device=DEV)
    oy = torch.zeros(n, dtype=torch.int32, device=DEV)
    # (a) capture once, replay
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.p2r_const(x0, 0, dph, ox, oy, n=n - n % 4)   # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.p2r_const(x0, 0, dph, ox, oy, n=n - n % 4)
    for _ in range(5):
        ox.zero_(); oy.zero_()
        g.replay()
        torch.cuda.synchronize()
        m = n - n % 4
        assert np.array_equal(to_np(ox)[:m], rx[:m])
        assert np.array_equal(to_np(oy)[:m], ry[:m])
    # (b) two streams, interleaved launches of the same plan
    s1, s2 = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    outs = [[torch.zeros(n, dtype=torch.int32, device=DEV) for _ in range(2)]
            for _ in range(8)]
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(outs):
        plan.p2r_const(x0, 0, dph, a, b, n=n, stream=(s1, s2)[k & 1])
    torch.cuda.synchronize()
    for a, b in outs:
        assert np.array_equal(to_np(a), rx) and np.array_equal(to_np(b), ry)
    plan.close()


@pytest.mark.gpu
def test_more_launches_in_flight_than_the_plan_has_tile_queues():
    """A tile queue is handed out again only when the launch that used it has
    completed; when every one is busy a launch runs the static sweep.  160
    launches of ONE plan over four streams (far more than the 48 queues),
    every output array pre-filled with a sentinel: no sample may be missing,
    none may come from another launch."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    n = (1 << 22) + 4096 * 5
    x0 = (1 << 31) - 1
    streams = [torch.cuda.Stream(device=DEV) for _ in range(4)]
    rng = np.random.RandomState(11)
    phases, want, outs = [], [], []
    for k in range(4):
        ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        phases.append(dev_i32(ph))
        want.append(O.rotate(ocfg, x0, 0, ph))
        outs.append([[torch.empty(n, dtype=torch.int32, device=DEV)
                      for _ in range(2)] for _ in range(2)])
    torch.cuda.synchronize()
    for rep in range(40):
        for k, st in enumerate(streams):
            a, b = outs[k][rep & 1]
            with torch.cuda.stream(st):
                a.fill_(0x5a5a5a5a); b.fill_(0x5a5a5a5a)
            plan.p2r_const(x0, 0, phases[k], a, b, n=n, stream=st)
            assert ca.last_kernel() == ca.KERNEL_SEEDED
    torch.cuda.synchronize()
    for k in range(4):
        for a, b in outs[k]:
            assert np.array_equal(to_np(a), want[k][0])
            assert np.array_equal(to_np(b), want[k][1])
    plan.close()


@pytest.mark.gpu
def test_graph_replays_overlap_eager_launches_of_the_same_plan():
    """A captured launch keeps its tile queue for good (replayable at any
    time); eager launches on another stream never get that one."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    n = 1 << 22
    x0 = (1 << 31) - 1
    rng = np.random.RandomState(12)
    ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    rx, ry = O.rotate(ocfg, x0, 0, ph)
    dph = dev_i32(ph)
    gx = torch.zeros(n, dtype=torch.int32, device=DEV)
    gy = torch.zeros(n, dtype=torch.int32, device=DEV)
    ex = [torch.zeros(n, dtype=torch.int32, device=DEV) for _ in range(2)]
    side = torch.cuda.Stream(device=DEV)
    plan.p2r_const(x0, 0, dph, gx, gy)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.p2r_const(x0, 0, dph, gx, gy)
    for rep in range(60):           # more eager launches than queues
        if rep % 3 == 0:
            gx.fill_(-1); gy.fill_(-1)
            g.replay()
        with torch.cuda.stream(side):
            ex[0].fill_(-1); ex[1].fill_(-1)
        plan.p2r_const(x0, 0, dph, ex[0], ex[1], stream=side)
        if rep % 3 == 2:
            torch.cuda.synchronize()
            assert np.array_equal(to_np(gx), rx) and np.array_equal(to_np(gy), ry)
            assert np.array_equal(to_np(ex[0]), rx)
            assert np.array_equal(to_np(ex[1]), ry)
    plan.close()


def _gpu_plan_p2r_xy(plan, x, y, ph):
    n = ph.size
    dx, dy, dph = dev_i32(x), dev_i32(y), dev_i32(ph)
    ox = torch.zeros(n, dtype=torch.int32, device=DEV)
    oy = torch.zeros(n, dtype=torch.int32, device=DEV)
    plan.p2r(dx, dy, dph, ox, oy)
    torch.cuda.synchronize()
    return to_np(ox), to_np(oy)


@pytest.mark.gpu
@pytest.mark.parametrize("args,groups", [
    ((ca.P2R, 32, 32, 2, 32, 16), [5, 5, 5]),          # p2rxy: WW35, LJ 29
    ((ca.P2R, 32, 32, 2, 32, 24), [4, 4, 5, 5, 5]),    # cfg4's core
    ((ca.P2R, 32, 32, 2, 32, -1), [4, 5, 5, 5, 5]),    # gencordic's own: 29 st.
    ((ca.P2R, 24, 24, 2, -1, -1), [4, 5, 5, 5, 5]),    # WW27 PW31 27 st., LJ 30
    ((ca.P2R, 16, 16, 2, -1, -1), [4, 4, 5, 5]),       # WW19 PW23 19 st.
    ((ca.P2R, 30, 30, 2, 32, 20), [4, 5, 5, 5]),       # WW33, LJ 30
])
def test_plan_p2r_per_sample_vectors_with_looked_up_directions(args, groups):
    """cordic_plan_p2r: per-sample x / y / phase with the stage directions
    read from the plan's tables (cordic_xydir.h) -- random inputs, the corners
    of the ports, and unit-step phase ramps across EVERY leaf boundary of
    every group, bit for bit against the oracle and against cordic_p2r."""
    cfg, ocfg = both(*args)
    plan = ca.Plan(cfg)
    assert plan.dir_groups == groups
    rng = np.random.RandomState(sum(groups) + cfg.ww)
    x, y, ph = rand_inputs(rng, cfg.iw, cfg.pw, 100003)
    a = _gpu_plan_p2r_xy(plan, x, y, ph)
    assert ca.last_kernel() == ca.KERNEL_DIRECTIONS
    b = O.rotate(ocfg, x, y, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # extreme vectors
    lo, hi = -(1 << (cfg.iw - 1)), (1 << (cfg.iw - 1)) - 1
    for xv, yv in ((hi, hi), (lo, lo), (lo, hi), (hi, 0), (0, lo), (0, 0), (1, -1)):
        xs = np.full(ph.size, xv, dtype=np.int32)
        ys = np.full(ph.size, yv, dtype=np.int32)
        a = _gpu_plan_p2r_xy(plan, xs, ys, ph)
        b = O.rotate(ocfg, xs, ys, ph)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (xv, yv)
    # every direction boundary of the first stages: the phases whose residual
    # changes sign at stage k are +/- partial sums of the arctan table; walk
    # +/- 40 units around each of them in every quadrant
    ang = [int(ocfg.angle[i]) for i in range(min(cfg.nlive, 12))]
    sums = {0}
    for a_ in ang[:10]:
        sums |= {s + a_ for s in sums} | {s - a_ for s in sums}
    base = np.array(sorted(sums), dtype=np.int64)
    base = base[np.abs(base) < (1 << (cfg.pw - 3))]
    if base.size > 4000:
        base = base[rng.choice(base.size, 4000, replace=False)]
    around = (base[:, None] + np.arange(-40, 41)[None, :]).ravel()
    allq = (around[None, :] + (np.arange(4, dtype=np.int64)
                               << (cfg.pw - 2))[:, None]).ravel()
    phb = (allq & ((1 << cfg.pw) - 1)).astype(np.uint32)
    xs, ys, _ = rand_inputs(rng, cfg.iw, cfg.pw, phb.size)
    a = _gpu_plan_p2r_xy(plan, xs, ys, phb)
    b = O.rotate(ocfg, xs, ys, phb)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # ragged length: the tail goes through the generic kernel
    a = _gpu_plan_p2r_xy(plan, x[:1003], y[:1003], ph[:1003])
    assert np.array_equal(a[0], b_ := O.rotate(ocfg, x[:1003], y[:1003], ph[:1003])[0])
    plan.close()


@pytest.mark.gpu
def test_plan_p2r_without_a_direction_table_runs_the_plain_kernel():
    for args in ((ca.P2R, 32, 32, 3, 32, 16),          # WW 36
                 (ca.P2R, 32, 32, 2, 32, 12)):          # 12 stages: no instance
        cfg, ocfg = both(*args)
        plan = ca.Plan(cfg)
        rng = np.random.RandomState(5)
        x, y, ph = rand_inputs(rng, cfg.iw, cfg.pw, 20001)
        a = _gpu_plan_p2r_xy(plan, x, y, ph)
        assert ca.last_kernel() == ca.KERNEL_UNROLLED
        b = O.rotate(ocfg, x, y, ph)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        plan.close()
    # and with the A/B flag the table is ignored
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg.with_flags(ca.FLAG_NO_TAILS))
    a = _gpu_plan_p2r_xy(plan, x, y, ph)
    assert ca.last_kernel() == ca.KERNEL_UNROLLED
    b = O.rotate(ocfg, x, y, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    plan.close()


@pytest.mark.gpu
def test_queue_ring_bookkeeping_is_visible_and_survives_failed_launches():
    """cordic_plan_queue_info: captured launches take a block each for the
    life of the handle (208 of them), later ones sweep static chunks and are
    counted; a launch that FAILS (null output: CORDIC_ERR_ARGS) must leave its
    block's pending event in place (round-3 advice), so results stay complete
    when many streams share the plan right after failures."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    q = plan.queue_info
    assert q == {"eager_slots": 48, "captured_capacity": 208,
                 "captured_used": 0, "fallback_launches": 0}
    n = 1 << 22
    x0 = (1 << 31) - 1
    rng = np.random.RandomState(21)
    ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    rx, ry = O.rotate(ocfg, x0, 0, ph)
    dph = dev_i32(ph)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(4)]
    outs = [[torch.zeros(n, dtype=torch.int32, device=DEV) for _ in range(2)]
            for _ in range(4)]
    for rep in range(40):           # the ring wraps three times
        for st, (a, b) in zip(streams, outs):
            plan.p2r_const(x0, 0, dph, a, b, stream=st)
            if rep % 5 == 0:        # a failing launch between good ones
                rc = ca.lib().cordic_plan_p2r_const(
                    plan._h, n, x0, 0, dph.data_ptr(), None, None,
                    st.cuda_stream)
                assert rc == ca.ERR_ARGS
    torch.cuda.synchronize()
    for a, b in outs:
        assert np.array_equal(to_np(a), rx) and np.array_equal(to_np(b), ry)
    assert plan.queue_info["fallback_launches"] == 0
    # captures: 3 nodes in one graph
    gx = torch.zeros(n, dtype=torch.int32, device=DEV)
    gy = torch.zeros(n, dtype=torch.int32, device=DEV)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(3):
            plan.p2r_const(x0, 0, dph, gx, gy)
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(to_np(gx), rx) and np.array_equal(to_np(gy), ry)
    q = plan.queue_info
    assert q["captured_used"] == 3 and q["fallback_launches"] == 0
    plan.close()


@pytest.mark.gpu
def test_the_fast_paths_are_the_ones_that_run():
    """cordic_last_kernel: BASELINE's cores land on the kernels the bench
    reports (a silently slower fallback would still be bit-exact)."""
    n = 1 << 16
    ph = torch.zeros(n, dtype=torch.int32, device=DEV)
    a = torch.zeros_like(ph); b = torch.zeros_like(ph)
    for args in ((ca.P2R, 32, 32, 2, 32, 16), (ca.P2R, 32, 32, 2, 32, 24),
                 (ca.SP2R, 32, 32, 2, 32, 16)):
        cfg = ca.Config.from_cli(*args)
        plan = ca.Plan(cfg)
        plan.p2r_const(1, 0, ph, a, b)
        assert ca.last_kernel() == ca.KERNEL_SEEDED
        plan.nco(n, 0, 1, 0, 1, 0, a, b)
        assert ca.last_kernel() == ca.KERNEL_SEEDED
        ca.p2r_const(cfg, 1, 0, ph, a, b)
        assert ca.last_kernel() == ca.KERNEL_UNROLLED
        ca.p2r(cfg, a, b, ph, a, b)
        assert ca.last_kernel() == ca.KERNEL_UNROLLED
        plan.close()
    cfg = ca.Config.from_cli(ca.R2P, 24, 24, 2, -1, 20)
    ca.r2p(cfg, a, b, a, b)
    assert ca.last_kernel() == ca.KERNEL_LEFT_JUSTIFIED
    ca.r2p(cfg.with_flags(ca.FLAG_FORCE_GENERIC), a, b, a, b)
    assert ca.last_kernel() == ca.KERNEL_GENERIC
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_one_plan_launched_from_four_host_threads():
    """Plan handles may be shared between threads (include/cordic_amd.h): four
    host threads, one stream each, 30 launches each of ONE plan (ctypes drops
    the GIL inside the calls, so the queue-slot bookkeeping really is entered
    concurrently); every output complete and its own."""
    import threading
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    n = (1 << 21) + 4096 + 12
    x0 = (1 << 31) - 1
    rng = np.random.RandomState(21)
    jobs = []
    for k in range(4):
        ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        jobs.append(dict(ph=dev_i32(ph), want=O.rotate(ocfg, x0, 0, ph),
                         st=torch.cuda.Stream(device=DEV),
                         a=torch.empty(n, dtype=torch.int32, device=DEV),
                         b=torch.empty(n, dtype=torch.int32, device=DEV)))
    torch.cuda.synchronize()
    errors = []

    def work(j):
        try:
            for _ in range(30):
                with torch.cuda.stream(j["st"]):
                    j["a"].fill_(0x5a5a5a5a); j["b"].fill_(0x5a5a5a5a)
                plan.p2r_const(x0, 0, j["ph"], j["a"], j["b"], n=n, stream=j["st"])
        except Exception as e:      # surfaced in the main thread below
            errors.append(e)
    th = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for j in jobs:
        assert np.array_equal(to_np(j["a"]), j["want"][0])
        assert np.array_equal(to_np(j["b"]), j["want"][1])
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("args", [(ca.P2R, 32, 32, 2, 32, 16),      # cfg2: one group of 5
                                  (ca.P2R, 32, 32, 2, 32, 17),
                                  (ca.P2R, 32, 32, 2, 32, 18),
                                  (ca.P2R, 32, 32, 2, 32, 30),
                                  (ca.SP2R, 32, 32, 2, 32, -1),
                                  (ca.P2R, 31, 31, 2, 30, 28),
                                  (ca.P2R, 32, 32, 2, 32, 19),
                                  (ca.P2R, 24, 24, 2, -1, -1),     # WW 27: 32-bit container
                                  (ca.P2R, 16, 16, 2, -1, -1),     # WW 19, PW 23
                                  (ca.SP2R, 20, 20, 2, -1, -1),
                                  (ca.P2R, 32, 32, 2, 32, 20),
                                  (ca.P2R, 32, 32, 2, 32, 24),
                                  (ca.SP2R, 32, 32, 2, 32, 22),
                                  (ca.P2R, 31, 31, 2, 30, 21),
                                  (ca.P2R, 32, 32, 2, 32, 26)])
def test_direction_tails_on_every_group_boundary(args):
    """The stages behind the seeds take their multipliers from tables
    indexed by the residual phase (direction tails) wherever a wave's row of
    256 phases is coherent (first and last less than 2^24 apart), and run the
    phase recurrence elsewhere.  (a) Unit-step ramps of 2^21 phases from
    random starts in every quadrant: coherent rows that put the residual on
    EVERY integer of several seed leaves, hence on every boundary of every
    group; (b) 256-sample blocks around each boundary of the first group for
    a spread of seed leaves; (c) unrelated phases (the recurrence path of the
    same kernel); (d) the same batch with CORDIC_FLAG_NO_TAILS; (e) an NCO
    with a small increment (tails) and a large one (plain instance).  All
    equal the oracle."""
    from test_seed_table import parse, parse_tails
    cfg, ocfg = both(*args)
    words = ca.seed_table(cfg)
    m, S, nb, L, buckets, leaves = parse(words)
    tails = parse_tails(words)
    assert tails is not None
    plan = ca.Plan(cfg)
    assert plan.seed_info["stages"] == 11
    assert plan.tail_groups == [g["t"] for g in tails["groups"]]
    lsh = 32 - cfg.pw
    step = 1 << lsh                     # one LSB of the PW-bit phase, P units
    rng = np.random.RandomState(5)
    parts = []
    for q in range(4):                  # (a)
        start = int(rng.randint(0, 1 << 30)) + (q << 30)
        parts.append((start + step * np.arange(1 << 21, dtype=np.int64)) & 0xffffffff)
    g0 = tails["groups"][0]             # (b)
    b0 = g0["buckets"]
    bounds = b0[b0[:, 0] != 0x7fffffff, 0] + 1 - tails["bias0"]    # residuals
    blk = step * (np.arange(256, dtype=np.int64) - 128)
    for j in rng.choice(L, size=min(L, 24), replace=False):
        off = int(leaves[j, 1]) - (1 << 29)
        for b in bounds:
            base = (off + int(b)) // step * step
            parts.append((base + blk + (int(rng.randint(4)) << 30)) & 0xffffffff)
    parts.append(rng.randint(0, 1 << 32, 1 << 20, dtype=np.uint64).astype(np.int64))  # (c)
    ph = (np.concatenate(parts).astype(np.uint64) >> np.uint64(lsh)).astype(np.uint32)
    ph = ph[: ph.size - ph.size % 4]
    x0 = (1 << (cfg.iw - 1)) - 1
    want = O.rotate(ocfg, x0, -(x0 // 3), ph)
    got = gpu_plan_p2r(plan, x0, -(x0 // 3), ph)
    assert ca.last_kernel() == ca.KERNEL_SEEDED
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    plain = ca.Plan(cfg.with_flags(ca.FLAG_NO_TAILS))   # (d)
    got = gpu_plan_p2r(plain, x0, -(x0 // 3), ph)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    for fcw in (1, 3, 0x01234567 >> lsh):               # (e)
        a, b = gpu_plan_nco(plan, 1 << 21, 0x1234, fcw, 77, x0, 0)
        idx = (np.arange(1 << 21, dtype=np.uint64) + np.uint64(77))
        pn = ((np.uint64(0x1234) + idx * np.uint64(fcw))
              & np.uint64((1 << cfg.pw) - 1)).astype(np.uint32)
        wa, wb = O.rotate(ocfg, x0, 0, pn)
        assert np.array_equal(a, wa) and np.array_equal(b, wb), fcw
    plan.close(); plain.close()


@pytest.mark.gpu
def test_plan_p2r_on_two_streams_and_in_a_hip_graph():
    """cordic_plan_p2r only enqueues: launches of one plan on two streams at
    once, and a launch captured into a HIP graph and replayed on new data."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    n = 1 << 20
    rng = np.random.RandomState(31)
    sets = [rand_inputs(rng, 32, 32, n) for _ in range(2)]
    dev = [[dev_i32(a) for a in s] for s in sets]
    outs = [[torch.zeros(n, dtype=torch.int32, device=DEV) for _ in range(2)]
            for _ in range(2)]
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(5):
        for k in range(2):
            plan.p2r(dev[k][0], dev[k][1], dev[k][2], outs[k][0], outs[k][1],
                     stream=streams[k])
    torch.cuda.synchronize()
    for k in range(2):
        rx, ry = O.rotate(ocfg, *sets[k])
        assert np.array_equal(to_np(outs[k][0]), rx)
        assert np.array_equal(to_np(outs[k][1]), ry)
    # capture once, replay on other inputs
    gx, gy, gp = (torch.zeros(n, dtype=torch.int32, device=DEV) for _ in range(3))
    ox = torch.zeros(n, dtype=torch.int32, device=DEV)
    oy = torch.zeros(n, dtype=torch.int32, device=DEV)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.p2r(gx, gy, gp, ox, oy)
    for k in range(2):
        gx.copy_(dev[k][0]); gy.copy_(dev[k][1]); gp.copy_(dev[k][2])
        g.replay()
        torch.cuda.synchronize()
        rx, ry = O.rotate(ocfg, *sets[k])
        assert np.array_equal(to_np(ox), rx) and np.array_equal(to_np(oy), ry)
    plan.close()


@pytest.mark.gpu
def test_a_plan_serves_small_batches_with_the_plain_kernel():
    """Below ~2^22 samples staging the seed table costs more than it saves
    (profiles/r05/small_batch.txt; 2^23 before the plan kept the prologue's
    image): without CORDIC_SEED_MIN_SAMPLES in the environment (the test suite
    sets it to 0) a plan picks the kernel by the batch size -- same bits
    either way."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np, torch
import cordic_amd as ca, oracle_lib as O
from gpu_util import gpu_digest
for args, small, large in (((ca.P2R, 32, 32, 2, 32, 16), 1 << 21, 1 << 22),
                           ((ca.P2R, 32, 32, 2, 32, 24), 1 << 21, 1 << 22)):
    cfg, ocfg = ca.Config.from_cli(*args), O.config_cli(*args)
    plan = ca.Plan(cfg)
    for n, want in ((small, ca.KERNEL_UNROLLED), (large, ca.KERNEL_SEEDED),
                    (1 << 16, ca.KERNEL_UNROLLED)):
        ph = torch.empty(n, dtype=torch.int32, device="cuda")
        a = torch.empty_like(ph); b = torch.empty_like(ph)
        ca.fill_phase_ramp(ph, 0, 3)
        plan.p2r_const(2**31 - 1, 0, ph, a, b)
        torch.cuda.synchronize()
        assert ca.last_kernel() == want, (n, ca.last_kernel())
        d = (gpu_digest(a, 0) + gpu_digest(b, 1 << 40)) %% 2**64
        assert d == O.job_digest(ocfg, "p2r", 0, n, 0, 8, 2**31 - 1, 0)[0], n
        plan.nco(n, 5, 7, 0, 2**31 - 1, 0, a, b)
        torch.cuda.synchronize()
        assert ca.last_kernel() == want
    # per-sample vectors: directions looked up from 2^23 samples on
    for n, want in ((1 << 20, ca.KERNEL_UNROLLED), (1 << 23, ca.KERNEL_DIRECTIONS)):
        ph = torch.empty(n, dtype=torch.int32, device="cuda")
        x = torch.empty_like(ph); y = torch.empty_like(ph)
        a = torch.empty_like(ph); b = torch.empty_like(ph)
        ca.fill_phase_ramp(ph, 0, 3)
        ca.fill_iq_ramp(x, y, 0, O.IQ_MULX, O.IQ_MULY, 32)
        plan.p2r(x, y, ph, a, b)
        torch.cuda.synchronize()
        assert ca.last_kernel() == want, (n, ca.last_kernel())
        d = (gpu_digest(a, 0) + gpu_digest(b, 1 << 40)) %% 2**64
        assert d == O.job_digest(ocfg, "p2rxy", 0, n, 0, 8)[0], n
print("ok")
""" % (root, root)
    env = {k: v for k, v in os.environ.items() if k != "CORDIC_SEED_MIN_SAMPLES"}
    r = subprocess.run([sys.executable, "-c", script], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]
