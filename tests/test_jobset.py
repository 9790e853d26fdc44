"""Job sets (round 5, include/cordic_amd.h "job sets"): many small jobs in ONE
launch of the seeded kernel.  Per job the results must be, bit for bit, what
the oracle computes for that job alone -- ragged lengths, unaligned addresses,
empty jobs, jobs that end inside a tile, NCO jobs whose sample index wraps."""
import numpy as np
import pytest

import cordic_amd as ca
import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if torch.cuda.is_available():
    from gpu_util import DEV, dev_i32, to_np


def both(mode, iw=-1, ow=-1, xtra=2, pw=-1, ns=-1, flags=0):
    cfg = ca.Config.from_cli(mode, iw, ow, xtra, pw, ns)
    if flags:
        cfg = cfg.with_flags(flags)
    return cfg, O.config_cli(mode, iw, ow, xtra, pw, ns)


RAGGED = [0, 1, 3, 4, 5, 8191, 8192, 8193, 4096 * 4, 4096 * 4 + 2, 65536,
          65536 + 7, 100003, 1 << 17, (1 << 15) - 1, 12, 0, 2, 40000, 8192 * 4]


def carve(total_words, sizes, offsets):
    """sizes[k] words at a 4-byte offset offsets[k] past job k's own 64-byte
    aligned start inside one big device array: (views, flat array)"""
    big = torch.zeros(total_words, dtype=torch.int32, device=DEV)
    views, at = [], 0
    for n, off in zip(sizes, offsets):
        at = (at + 15) // 16 * 16 + off
        views.append(big[at:at + n])
        at += n
    assert at <= total_words
    return views, big


CORES = {
    "cfg2": ((ca.P2R, 32, 32, 2, 32, 16), 0),            # lj29
    "cfg4": ((ca.P2R, 32, 32, 2, 32, 24), 0),
    "nat16": ((ca.P2R, 16, 16, 2, -1, -1), 0),           # lj30
    "nat16_narrow": ((ca.P2R, 16, 16, 2, -1, -1), ca.FLAG_NO_LJ),
    "pw20": ((ca.P2R, 13, 13, 2, -1, -1), 0),            # PW 20: phase words scaled
    "cfg5seq": ((ca.SP2R, 32, 32, 2, 32, 16), 0),
}


@pytest.mark.parametrize("name", sorted(CORES))
def test_phase_array_jobs_equal_the_oracle_job_by_job(name):
    args, flags = CORES[name]
    cfg, ocfg = both(*args, flags=flags)
    plan = ca.Plan(cfg)
    rng = np.random.RandomState(31)
    sizes = RAGGED
    offs = [int(v) for v in rng.randint(0, 4, len(sizes))]
    total = sum(sizes) + 32 * len(sizes)
    phv, _ = carve(total, sizes, offs)
    oxv, oxbig = carve(total, sizes, offs[::-1])
    oyv, oybig = carve(total, sizes, [(o + 1) % 4 for o in offs])
    mask = (1 << cfg.pw) - 1
    hp = []
    for v in phv:
        h = rng.randint(0, 1 << 32, v.numel(), dtype=np.uint64).astype(np.uint32)
        hp.append(h & np.uint32(mask))
        if v.numel():
            v.copy_(dev_i32(hp[-1]))
    jobs = [dict(phase=p, ox=a, oy=b, n=p.numel())
            for p, a, b in zip(phv, oxv, oyv)]
    hi = (1 << (cfg.iw - 1)) - 1
    js = ca.Jobset(plan, ca.JOBS_PHASE_ARRAYS, jobs)
    info = js.info
    assert info["samples"] == sum(sizes)
    assert info["tiles"] == sum(-(-(n // 4) // 2048) for n in sizes)
    assert info["tail_samples"] == sum(n % 4 for n in sizes)
    for x0, y0 in [(hi, 0), (-hi // 3, hi // 5)]:
        oxbig.fill_(0x5a5a5a5a); oybig.fill_(0x5a5a5a5a)
        js.run(x0, y0)
        torch.cuda.synchronize()
        assert ca.last_kernel() == ca.KERNEL_SEEDED
        for k, n in enumerate(sizes):
            rx, ry = O.rotate(ocfg, x0, y0, hp[k])
            assert np.array_equal(to_np(oxv[k]), rx), (k, n)
            assert np.array_equal(to_np(oyv[k]), ry), (k, n)
        # nothing outside the jobs' arrays was touched
        untouched = int((oxbig == 0x5a5a5a5a).sum().item())
        assert untouched == oxbig.numel() - sum(sizes)
    # the one-shot form: same bits
    oxbig.zero_(); oybig.zero_()
    plan.p2r_const_batch(jobs, hi, 0)
    torch.cuda.synchronize()
    for k in range(len(sizes)):
        rx, ry = O.rotate(ocfg, hi, 0, hp[k])
        assert np.array_equal(to_np(oxv[k]), rx) and np.array_equal(to_np(oyv[k]), ry)
    ca.jobset_reap()
    js.close(); plan.close()


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "pw20", "nat16"])
def test_nco_jobs_equal_the_oracle_job_by_job(name):
    args, flags = CORES[name]
    cfg, ocfg = both(*args, flags=flags)
    plan = ca.Plan(cfg)
    rng = np.random.RandomState(37)
    sizes = [n for n in RAGGED if n]
    total = sum(sizes) + 32 * len(sizes)
    oxv, oxbig = carve(total, sizes, [k % 4 for k in range(len(sizes))])
    oyv, oybig = carve(total, sizes, [(k + 2) % 4 for k in range(len(sizes))])
    jobs = []
    for k, n in enumerate(sizes):
        jobs.append(dict(ox=oxv[k], oy=oyv[k], n=n,
                         phase0=int(rng.randint(0, 1 << 32, dtype=np.uint64)),
                         fcw=int(rng.randint(0, 1 << 32, dtype=np.uint64)) | 1,
                         # some jobs straddle the 2^32 wrap of the sample index
                         index0=(1 << 32) - n // 2 if k % 3 == 0 else
                         int(rng.randint(0, 1 << 40, dtype=np.uint64))))
    hi = (1 << (cfg.iw - 1)) - 1
    js = ca.Jobset(plan, ca.JOBS_NCO, jobs)
    for rep in range(2):
        oxbig.zero_(); oybig.zero_()
        js.run(hi, -5)
        torch.cuda.synchronize()
        for k, jb in enumerate(jobs):
            rx, ry = O.nco(ocfg, jb["n"], jb["phase0"], jb["fcw"], jb["index0"], hi, -5)
            assert np.array_equal(to_np(oxv[k]), rx), (rep, k)
            assert np.array_equal(to_np(oyv[k]), ry), (rep, k)
    oxbig.zero_(); oybig.zero_()
    plan.nco_batch(jobs, hi, -5)
    torch.cuda.synchronize()
    for k, jb in enumerate(jobs):
        rx, ry = O.nco(ocfg, jb["n"], jb["phase0"], jb["fcw"], jb["index0"], hi, -5)
        assert np.array_equal(to_np(oxv[k]), rx) and np.array_equal(to_np(oyv[k]), ry)
    ca.jobset_reap()
    js.close(); plan.close()


def test_a_thousand_jobs_in_one_launch_and_in_a_graph():
    """1024 jobs x 2^14 samples: one launch; the same set replayed from a HIP
    graph on new phases; digest of all outputs == oracle's, job by job."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    nj, n = 1024, 1 << 14
    ph = torch.empty(nj * n, dtype=torch.int32, device=DEV)
    ox = torch.zeros_like(ph)
    oy = torch.zeros_like(ph)
    ca.fill_phase_ramp(ph, 0, 5)
    jobs = [dict(phase=ph[k * n:(k + 1) * n], ox=ox[k * n:(k + 1) * n],
                 oy=oy[k * n:(k + 1) * n], n=n) for k in range(nj)]
    js = ca.Jobset(plan, ca.JOBS_PHASE_ARRAYS, jobs)
    x0 = (1 << 31) - 1
    plan.prepare(x0, 0)
    js.run(x0, 0)
    torch.cuda.synchronize()
    # contiguous jobs: the whole thing is cfg2's ramp job
    want = O.job_digest(ocfg, "p2r", 0, nj * n, 0, 32, x0, 0)[0]
    from gpu_util import gpu_digest
    assert (gpu_digest(ox, 0) + gpu_digest(oy, 1 << 40)) % 2**64 == want
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        js.run(x0, 0)
    ca.fill_phase_ramp(ph, 12345, 5)
    ox.zero_(); oy.zero_()
    g.replay()
    torch.cuda.synchronize()
    want = O.job_digest(ocfg, "p2r", 12345, nj * n, 0, 32, x0, 0)[0]
    assert (gpu_digest(ox, 12345) + gpu_digest(oy, 12345 + (1 << 40))) % 2**64 == want
    js.close(); plan.close()


def test_cores_without_a_seeded_kernel_run_the_jobs_one_by_one():
    for args, flags in (((ca.P2R, 32, 32, 8, 32, 24), 0),        # WW 41
                        ((ca.P2R, 32, 32, 2, 32, 16), ca.FLAG_NO_SEED),
                        ((ca.P2R, 12, 12, 2, 14, 10), 0)):        # 10 live stages
        cfg, ocfg = both(*args, flags=flags)
        plan = ca.Plan(cfg)
        sizes = [5, 4096, 0, 10001]
        phv, _ = carve(20000, sizes, [0, 1, 2, 3])
        oxv, _ = carve(20000, sizes, [1, 1, 1, 1])
        oyv, _ = carve(20000, sizes, [0, 0, 0, 0])
        rng = np.random.RandomState(41)
        hp = []
        for v in phv:
            hp.append(rng.randint(0, 1 << cfg.pw, v.numel(), dtype=np.uint64)
                      .astype(np.uint32))
            if v.numel():
                v.copy_(dev_i32(hp[-1]))
        jobs = [dict(phase=p, ox=a, oy=b, n=p.numel())
                for p, a, b in zip(phv, oxv, oyv)]
        js = ca.Jobset(plan, ca.JOBS_PHASE_ARRAYS, jobs)
        hi = (1 << (cfg.iw - 1)) - 1
        js.run(hi, 3)
        torch.cuda.synchronize()
        assert ca.last_kernel() != ca.KERNEL_SEEDED
        for k in range(len(sizes)):
            rx, ry = O.rotate(ocfg, hi, 3, hp[k])
            assert np.array_equal(to_np(oxv[k]), rx) and np.array_equal(to_np(oyv[k]), ry)
        js.close(); plan.close()


def test_bad_jobs_are_refused():
    cfg, _ = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    t = torch.zeros(64, dtype=torch.int32, device=DEV)
    with pytest.raises(ca.CordicError) as e:
        ca.Jobset(plan, ca.JOBS_PHASE_ARRAYS, [dict(phase=None, ox=t, oy=t, n=8)])
    assert e.value.status == ca.ERR_ARGS
    with pytest.raises(ca.CordicError):
        ca.Jobset(plan, 7, [dict(phase=t, ox=t, oy=t, n=8)])
    with pytest.raises(ca.CordicError):                 # 2-byte aligned output
        ca.Jobset(plan, ca.JOBS_NCO, [dict(ox=t.data_ptr() + 2, oy=t, n=8)])
    # a set cut for one core does not run on another
    js = ca.Jobset(plan, ca.JOBS_NCO, [dict(ox=t, oy=t[32:], n=8)])
    other = ca.Plan(ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 24))
    with pytest.raises(ca.CordicError):
        js.run(1, 0, plan=other)
    r2p = ca.Plan(ca.Config.from_cli(ca.R2P, 24, 24, 2, -1, 20))
    with pytest.raises(ca.CordicError):
        ca.Jobset(r2p, ca.JOBS_NCO, [dict(ox=t, oy=t[32:], n=8)])
    js.close(); plan.close(); other.close(); r2p.close()


# ------------------------------------------------- data-fed kinds (round 6)
#
# CORDIC_JOBS_R2P / _P2R_XY / _MIX: the converter (rtl/topolar.v:59-64), the
# rotator with all three ports live (rtl/cordic.v:58-63) and the fused mixer as
# many small jobs in ONE launch of the call's own kernel reading tile
# descriptors.  Same rules as above: ragged, unaligned, empty jobs; per job the
# oracle's bits.

def xy_tile_vecs(total_vecs):
    """cordic_abi.cpp: xy_tile_vecs"""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    per = total_vecs // (cus * 8 * 4)
    return min(2048, max(256, per // 256 * 256))


R2P_CORES = {
    "cfg3": ((ca.R2P, 24, 24, 2, -1, 20), 0, True),          # static 20
    "natr2p24": ((ca.R2P, 24, 24, 2, -1, -1), 0, True),      # static 29
    "r2p16": ((ca.R2P, 16, 16, 2, -1, -1), 0, True),         # dynamic exit
    "sr2p": ((ca.SR2P, 24, 24, 2, -1, 20), 0, True),
    "r2p32": ((ca.R2P, 32, 32, 2, 32, 24), 0, False),        # WW 40: one by one
    "unit_gain": ((ca.R2P, 24, 24, 2, -1, 20), ca.FLAG_UNIT_GAIN, False),
}


def _iq(rng, n, iw, full=False):
    lo, hi = -(1 << (iw - 1)), (1 << (iw - 1)) - 1
    if full:        # the ports take the low IW bits of whatever the word holds
        return rng.randint(-2**31, 2**31 - 1, n).astype(np.int32)
    return rng.randint(lo, hi + 1, n).astype(np.int32)


@pytest.mark.parametrize("name", sorted(R2P_CORES))
def test_r2p_jobs_equal_the_oracle_job_by_job(name):
    args, flags, one_launch = R2P_CORES[name]
    cfg, ocfg = both(*args, flags=flags)
    plan = ca.Plan(cfg)
    gain_k = (ca.lib().cordic_config_gain_annihilator(cfg.ref)
              if flags & ca.FLAG_UNIT_GAIN else None)

    def oracle(k):
        rm, rp = O.topolar(ocfg, hx[k], hy[k])
        if gain_k is not None:          # o = (o * K) >> 32 (include/cordic_amd.h)
            rm = ((rm.astype(np.int64) * gain_k) >> 32).astype(np.int32)
        return rm, rp
    rng = np.random.RandomState(43)
    sizes = RAGGED
    offs = [int(v) for v in rng.randint(0, 4, len(sizes))]
    total = sum(sizes) + 32 * len(sizes)
    xv, _ = carve(total, sizes, offs)
    yv, _ = carve(total, sizes, [(o + 3) % 4 for o in offs])
    mv, mbig = carve(total, sizes, offs[::-1])
    pv, pbig = carve(total, sizes, [(o + 1) % 4 for o in offs])
    hx, hy = [], []
    for k, n in enumerate(sizes):
        hx.append(_iq(rng, n, cfg.iw, full=(k % 5 == 0)))
        hy.append(_iq(rng, n, cfg.iw, full=(k % 5 == 0)))
        if n:
            xv[k].copy_(dev_i32(hx[-1]))
            yv[k].copy_(dev_i32(hy[-1]))
    jobs = [dict(x=a, y=b, ox=m, oy=p, n=a.numel())
            for a, b, m, p in zip(xv, yv, mv, pv)]
    js = ca.Jobset(plan, ca.JOBS_R2P, jobs)
    info = js.info
    tv = xy_tile_vecs(sum(n // 4 for n in sizes))
    assert info["samples"] == sum(sizes)
    assert info["tiles"] == sum(-(-(n // 4) // tv) for n in sizes)
    assert info["tail_samples"] == sum(n % 4 for n in sizes)
    for rep in range(2):
        mbig.fill_(0x5a5a5a5a); pbig.fill_(0x5a5a5a5a)
        js.run()
        torch.cuda.synchronize()
        if one_launch:
            assert ca.last_kernel() == ca.KERNEL_LEFT_JUSTIFIED
        for k, n in enumerate(sizes):
            rm, rp = oracle(k)
            assert np.array_equal(to_np(mv[k]), rm), (k, n)
            assert np.array_equal(to_np(pv[k], np.uint32), rp), (k, n)
        assert int((mbig == 0x5a5a5a5a).sum().item()) == mbig.numel() - sum(sizes)
        assert int((pbig == 0x5a5a5a5a).sum().item()) == pbig.numel() - sum(sizes)
    mbig.zero_(); pbig.zero_()
    plan.xy_batch(ca.JOBS_R2P, jobs)            # the one-shot form
    torch.cuda.synchronize()
    for k in range(len(sizes)):
        rm, rp = oracle(k)
        assert np.array_equal(to_np(mv[k]), rm)
        assert np.array_equal(to_np(pv[k], np.uint32), rp)
    ca.jobset_reap()
    js.close(); plan.close()


XY_CORES = {
    "cfg2": ((ca.P2R, 32, 32, 2, 32, 16), True),            # lj29, 16
    "cfg4": ((ca.P2R, 32, 32, 2, 32, 24), True),            # lj29, 24
    "nat32": ((ca.P2R, 32, 32, 2, 32, -1), True),           # lj29, 29
    "nat16": ((ca.P2R, 16, 16, 2, -1, -1), True),           # lj30, 19
    "nat24": ((ca.P2R, 24, 24, 2, -1, -1), True),           # lj30, 27
    "pw20": ((ca.P2R, 13, 13, 2, -1, -1), False),           # no instance: one by one
    "n20": ((ca.P2R, 32, 32, 2, 32, 20), False),
    "cfg5seq": ((ca.SP2R, 32, 32, 2, 32, 16), None),        # whichever serves it
}


@pytest.mark.parametrize("kind", ["p2rxy", "mix"])
@pytest.mark.parametrize("name", sorted(XY_CORES))
def test_vector_jobs_equal_the_oracle_job_by_job(name, kind):
    args, one_launch = XY_CORES[name]
    cfg, ocfg = both(*args)
    plan = ca.Plan(cfg)
    rng = np.random.RandomState(47)
    sizes = RAGGED
    offs = [int(v) for v in rng.randint(0, 4, len(sizes))]
    total = sum(sizes) + 32 * len(sizes)
    xv, _ = carve(total, sizes, offs)
    yv, _ = carve(total, sizes, [(o + 3) % 4 for o in offs])
    phv, _ = carve(total, sizes, [(o + 2) % 4 for o in offs])
    av, abig = carve(total, sizes, offs[::-1])
    bv, bbig = carve(total, sizes, [(o + 1) % 4 for o in offs])
    mask = (1 << cfg.pw) - 1
    hx, hy, hp, jobs = [], [], [], []
    for k, n in enumerate(sizes):
        hx.append(_iq(rng, n, cfg.iw, full=(k % 5 == 0)))
        hy.append(_iq(rng, n, cfg.iw, full=(k % 5 == 0)))
        hp.append(rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
                  & np.uint32(mask))
        if n:
            xv[k].copy_(dev_i32(hx[-1]))
            yv[k].copy_(dev_i32(hy[-1]))
            phv[k].copy_(dev_i32(hp[-1]))
        jb = dict(x=xv[k], y=yv[k], ox=av[k], oy=bv[k], n=n)
        if kind == "p2rxy":
            jb["phase"] = phv[k]
        else:
            jb.update(phase0=int(rng.randint(0, 1 << 32, dtype=np.uint64)),
                      fcw=int(rng.randint(0, 1 << 32, dtype=np.uint64)) | 1,
                      index0=(1 << 32) - n // 2 if k % 3 == 0 else
                      int(rng.randint(0, 1 << 40, dtype=np.uint64)))
        jobs.append(jb)
    K = ca.JOBS_P2R_XY if kind == "p2rxy" else ca.JOBS_MIX
    js = ca.Jobset(plan, K, jobs)
    tv = xy_tile_vecs(sum(n // 4 for n in sizes))
    assert js.info["tiles"] == sum(-(-(n // 4) // tv) for n in sizes)
    assert js.info["tail_samples"] == sum(n % 4 for n in sizes)

    def want(k):
        if kind == "p2rxy":
            return O.rotate(ocfg, hx[k], hy[k], hp[k])
        jb = jobs[k]
        return O.mix(ocfg, jb["phase0"], jb["fcw"], jb["index0"], hx[k], hy[k])
    for rep in range(2):
        abig.fill_(0x5a5a5a5a); bbig.fill_(0x5a5a5a5a)
        js.run()
        torch.cuda.synchronize()
        if one_launch is True:
            assert ca.last_kernel() == ca.KERNEL_DIRECTIONS
        for k, n in enumerate(sizes):
            if not n:
                continue
            ra, rb = want(k)
            assert np.array_equal(to_np(av[k]), ra), (rep, k, n)
            assert np.array_equal(to_np(bv[k]), rb), (rep, k, n)
        assert int((abig == 0x5a5a5a5a).sum().item()) == abig.numel() - sum(sizes)
        assert int((bbig == 0x5a5a5a5a).sum().item()) == bbig.numel() - sum(sizes)
    abig.zero_(); bbig.zero_()
    plan.xy_batch(K, jobs)
    torch.cuda.synchronize()
    for k, n in enumerate(sizes):
        if n:
            ra, rb = want(k)
            assert np.array_equal(to_np(av[k]), ra) and np.array_equal(to_np(bv[k]), rb)
    ca.jobset_reap()
    js.close(); plan.close()


@pytest.mark.parametrize("kind", ["r2p", "mix", "p2rxy"])
def test_a_thousand_data_fed_jobs_in_one_launch_and_in_a_graph(kind):
    """1024 contiguous jobs x 2^14 samples of the bench's I/Q ramps: ONE launch
    (+ none for tails), replayed from a HIP graph on new data; the digest of
    all outputs equals the oracle's digest of the same job in one piece."""
    from gpu_util import gpu_digest
    r2p = kind == "r2p"
    args = (ca.R2P, 24, 24, 2, -1, 20) if r2p else (ca.P2R, 32, 32, 2, 32, 16)
    cfg, ocfg = both(*args)
    plan = ca.Plan(cfg)
    nj, n = 1024, 1 << 14
    N = nj * n
    x = torch.empty(N, dtype=torch.int32, device=DEV)
    y = torch.empty_like(x)
    ph = torch.empty_like(x)
    a = torch.zeros_like(x)
    b = torch.zeros_like(x)
    fcw = 4 if kind == "p2rxy" else 0x01234567

    def fill(start):
        ca.fill_iq_ramp(x, y, start, O.IQ_MULX, O.IQ_MULY, cfg.iw)
        ca.fill_phase_ramp(ph, start, 2)
    fill(0)

    def job(k, start):
        s = slice(k * n, (k + 1) * n)
        jb = dict(x=x[s], y=y[s], ox=a[s], oy=b[s], n=n)
        if kind == "p2rxy":
            jb["phase"] = ph[s]
        elif kind == "mix":
            jb.update(phase0=0, fcw=fcw, index0=start + k * n)
        return jb
    K = {"r2p": ca.JOBS_R2P, "mix": ca.JOBS_MIX, "p2rxy": ca.JOBS_P2R_XY}[kind]
    okind = {"r2p": "r2p", "mix": "mix", "p2rxy": "p2rxy"}[kind]
    js = ca.Jobset(plan, K, [job(k, 0) for k in range(nj)])
    assert js.info["tail_samples"] == 0
    js.run()
    torch.cuda.synchronize()
    assert ca.last_kernel() == (ca.KERNEL_LEFT_JUSTIFIED if r2p else ca.KERNEL_DIRECTIONS)
    want = O.job_digest(ocfg, okind, 0, N, 0, fcw)[0]
    assert (gpu_digest(a, 0) + gpu_digest(b, 1 << 40)) % 2**64 == want
    # the same set from a HIP graph, on new data in the same arrays (a mixer's
    # accumulators are part of the set: its second run is a second set)
    start = 0 if kind == "mix" else 777 * 4
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        js.run()
    fill(start)
    a.zero_(); b.zero_()
    g.replay()
    torch.cuda.synchronize()
    want = O.job_digest(ocfg, okind, start, N, 0, fcw)[0]
    assert (gpu_digest(a, start) + gpu_digest(b, start + (1 << 40))) % 2**64 == want
    # the one-shot forms refuse a capturing stream (they allocate and copy)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g2 = torch.cuda.CUDAGraph()
        with pytest.raises(ca.CordicError) as e:
            with torch.cuda.graph(g2, stream=s):
                plan.xy_batch(K, [job(0, 0)], stream=s.cuda_stream)
        assert e.value.status == ca.ERR_UNSUPPORTED
    js.close(); plan.close()


def test_bad_data_fed_jobs_are_refused():
    t = torch.zeros(64, dtype=torch.int32, device=DEV)
    rot = ca.Plan(ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 16))
    pol = ca.Plan(ca.Config.from_cli(ca.R2P, 24, 24, 2, -1, 20))
    ok = dict(x=t[:8], y=t[8:16], phase=t[16:24], ox=t[32:40], oy=t[40:48], n=8)
    with pytest.raises(ca.CordicError) as e:            # a converter's kind
        ca.Jobset(rot, ca.JOBS_R2P, [ok])
    assert e.value.status == ca.ERR_MODE
    for K in (ca.JOBS_P2R_XY, ca.JOBS_MIX, ca.JOBS_PHASE_ARRAYS):
        with pytest.raises(ca.CordicError) as e:        # a rotator's kinds
            ca.Jobset(pol, K, [ok])
        assert e.value.status == ca.ERR_MODE
    for K, plan in ((ca.JOBS_R2P, pol), (ca.JOBS_P2R_XY, rot), (ca.JOBS_MIX, rot)):
        with pytest.raises(ca.CordicError) as e:        # no i_yval array
            ca.Jobset(plan, K, [dict(ok, y=None)])
        assert e.value.status == ca.ERR_ARGS
        with pytest.raises(ca.CordicError):             # 2-byte aligned input
            ca.Jobset(plan, K, [dict(ok, x=t.data_ptr() + 2)])
    with pytest.raises(ca.CordicError):                 # phase array missing
        ca.Jobset(rot, ca.JOBS_P2R_XY, [dict(ok, phase=None)])
    js = ca.Jobset(rot, ca.JOBS_MIX, [dict(ok, phase=None, fcw=3)])  # needs none
    other = ca.Plan(ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 24))
    with pytest.raises(ca.CordicError):                 # another core's plan
        js.run(plan=other)
    js.close(); rot.close(); pol.close(); other.close()


@pytest.mark.parametrize("kind", ["r2p", "mix"])
def test_a_thousand_ragged_data_fed_jobs(kind):
    """1024 jobs of 0 .. 9000 samples each at odd word offsets of shared arrays
    (most of them shorter than a tile, many not a multiple of four samples):
    one launch + one for the trailing samples; per job the oracle's bits, and
    not a word outside the jobs' outputs touched."""
    r2p = kind == "r2p"
    args = (ca.R2P, 24, 24, 2, -1, 20) if r2p else (ca.P2R, 32, 32, 2, 32, 24)
    cfg, ocfg = both(*args)
    plan = ca.Plan(cfg)
    rng = np.random.RandomState(53)
    nj = 1024
    sizes = [int(v) for v in rng.randint(0, 9001, nj)]
    sizes[5] = sizes[700] = 0
    offs = [int(v) for v in rng.randint(0, 4, nj)]
    total = sum(sizes) + 32 * nj
    xv, xbig = carve(total, sizes, offs)
    yv, ybig = carve(total, sizes, offs[::-1])
    av, abig = carve(total, sizes, [(o + 1) % 4 for o in offs])
    bv, bbig = carve(total, sizes, [(o + 2) % 4 for o in offs])
    hx = _iq(rng, total, cfg.iw)
    hy = _iq(rng, total, cfg.iw)
    xbig.copy_(dev_i32(hx))
    ybig.copy_(dev_i32(hy))
    jobs = []
    for k in range(nj):
        jb = dict(x=xv[k], y=yv[k], ox=av[k], oy=bv[k], n=sizes[k])
        if not r2p:
            jb.update(phase0=int(rng.randint(0, 1 << 32, dtype=np.uint64)),
                      fcw=int(rng.choice([1, 257, int(rng.randint(0, 1 << 32,
                                                                   dtype=np.uint64))])),
                      index0=int(rng.randint(0, 1 << 40, dtype=np.uint64)))
        jobs.append(jb)
    js = ca.Jobset(plan, ca.JOBS_R2P if r2p else ca.JOBS_MIX, jobs)
    assert js.info["samples"] == sum(sizes)
    assert js.info["tail_samples"] == sum(n % 4 for n in sizes)
    abig.fill_(0x5a5a5a5a); bbig.fill_(0x5a5a5a5a)
    js.run()
    torch.cuda.synchronize()
    ga, gb = to_np(abig), to_np(bbig)
    xh, yh = to_np(xbig), to_np(ybig)
    base = abig.data_ptr()

    def span(view):
        lo = (view.data_ptr() - base) // 4
        return lo, lo + view.numel()
    touched_a = np.zeros(total, dtype=bool)
    for k in range(nj):
        if not sizes[k]:
            continue
        lo, hi = span(av[k])
        xl = (xv[k].data_ptr() - xbig.data_ptr()) // 4
        yl = (yv[k].data_ptr() - ybig.data_ptr()) // 4
        jx, jy = xh[xl:xl + sizes[k]], yh[yl:yl + sizes[k]]
        if r2p:
            wa, wb = O.topolar(ocfg, jx, jy)
            wb = wb.view(np.int32)
        else:
            wa, wb = O.mix(ocfg, jobs[k]["phase0"], jobs[k]["fcw"], jobs[k]["index0"], jx, jy)
        bl = (bv[k].data_ptr() - bbig.data_ptr()) // 4
        assert np.array_equal(ga[lo:hi], wa), k
        assert np.array_equal(gb[bl:bl + sizes[k]], wb), k
        touched_a[lo:hi] = True
    assert np.all(ga[~touched_a] == 0x5a5a5a5a)
    assert int((bbig == 0x5a5a5a5a).sum().item()) >= bbig.numel() - sum(sizes)
    js.close(); plan.close()
