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
