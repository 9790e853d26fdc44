#!/usr/bin/env python3
"""One-off differential fuzz: random cores x random samples, GPU (plain entry
points and plans) against the oracle.  Not part of the test suite (minutes);
run on a GPU box:  python tools/fuzz_gpu.py [configs] [seed]"""
import os
import sys

# small batches: a plan would serve them with the full recurrence by itself
# (cordic_kernels.hip: seed_min_samples); the fuzz wants the seeded kernels
os.environ.setdefault("CORDIC_SEED_MIN_SAMPLES", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cordic_amd as ca  # noqa: E402
import oracle_lib as O  # noqa: E402
from gpu_util import gpu_nco, gpu_p2r, gpu_plan_nco, gpu_plan_p2r, gpu_r2p  # noqa: E402
from test_gpu_parity import rand_inputs  # noqa: E402



def cut(rng, n):
    """[0, n) as 1-6 consecutive ragged pieces (empty ones included)"""
    k = int(rng.randint(1, 7))
    at = sorted(int(v) for v in rng.randint(0, n + 1, k - 1))
    return list(zip([0] + at, at + [n]))


def jobset_fuzz(rng, plan, cfg, ocfg, x, y, ph, paths):
    """Round 6: the same samples as a JOB SET of every kind the core's mode
    has -- ragged pieces at odd offsets of the same arrays -- against the
    oracle, piece by piece."""
    import torch
    n = len(x)
    dev = [torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).cuda()
           for v in (x, y, ph)]
    oa = torch.empty(n, dtype=torch.int32, device="cuda")
    ob = torch.empty(n, dtype=torch.int32, device="cuda")
    pieces = cut(rng, n)
    rot = cfg.mode in (ca.P2R, ca.SP2R)
    kinds = ([ca.JOBS_P2R_XY, ca.JOBS_MIX, ca.JOBS_PHASE_ARRAYS, ca.JOBS_NCO]
             if rot else [ca.JOBS_R2P])
    x0, y0 = int(x[0]), int(y[n - 1])
    for kind in kinds:
        jobs, want = [], []
        for lo, hi in pieces:
            jb = dict(ox=oa[lo:hi], oy=ob[lo:hi], n=hi - lo)
            if kind in (ca.JOBS_P2R_XY, ca.JOBS_MIX, ca.JOBS_R2P):
                jb.update(x=dev[0][lo:hi], y=dev[1][lo:hi])
            if kind in (ca.JOBS_P2R_XY, ca.JOBS_PHASE_ARRAYS):
                jb["phase"] = dev[2][lo:hi]
            if kind in (ca.JOBS_MIX, ca.JOBS_NCO):
                jb.update(phase0=int(rng.randint(1 << 32)),
                          fcw=int(rng.choice([1, 3, int(rng.randint(1 << 32))])),
                          index0=int(rng.randint(1 << 40)))
            jobs.append(jb)
            if hi == lo:
                want.append(None)
            elif kind == ca.JOBS_R2P:
                want.append(O.topolar(ocfg, x[lo:hi], y[lo:hi]))
            elif kind == ca.JOBS_P2R_XY:
                want.append(O.rotate(ocfg, x[lo:hi], y[lo:hi], ph[lo:hi]))
            elif kind == ca.JOBS_PHASE_ARRAYS:
                want.append(O.rotate(ocfg, x0, y0, ph[lo:hi]))
            elif kind == ca.JOBS_MIX:
                want.append(O.mix(ocfg, jb["phase0"], jb["fcw"], jb["index0"],
                                  x[lo:hi], y[lo:hi]))
            else:
                want.append(O.nco(ocfg, hi - lo, jb["phase0"], jb["fcw"],
                                  jb["index0"], x0, y0))
        oa.fill_(0x5a5a5a5a)
        ob.fill_(0x5a5a5a5a)
        js = ca.Jobset(plan, kind, jobs)
        js.run(x0, y0)
        torch.cuda.synchronize()
        fam = ca.last_kernel()
        js.close()
        ga, gb = oa.cpu().numpy(), ob.cpu().numpy()
        for (lo, hi), w in zip(pieces, want):
            if w is None:
                continue
            if not (np.array_equal(ga[lo:hi], w[0])
                    and np.array_equal(gb[lo:hi].view(w[1].dtype), w[1])):
                raise SystemExit("job set mismatch: kind %d core mode %d iw %d ow %d ww %d "
                                 "pw %d nlive %d piece [%d, %d) of %r"
                                 % (kind, cfg.mode, cfg.iw, cfg.ow, cfg.ww, cfg.pw,
                                    cfg.nlive, lo, hi, pieces))
        key = "jobs%d:%s" % (kind, {ca.KERNEL_SEEDED: "seeded",
                                    ca.KERNEL_DIRECTIONS: "dirs",
                                    ca.KERNEL_LEFT_JUSTIFIED: "lj"}.get(fam, "one-by-one"))
        paths[key] = paths.get(key, 0) + 1


ncfg = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
done = refused = 0
paths = {}
for t in range(ncfg):
    mode = int(rng.randint(4))
    iw, ow = int(rng.randint(1, 33)), int(rng.randint(1, 33))
    xtra = int(rng.randint(0, 8))
    pw = int(rng.randint(3, 33))
    ns = int(rng.randint(1, 65))
    try:
        cfg = ca.Config.from_cli(mode, iw, ow, xtra, pw, ns)
    except ca.CordicError:
        try:
            O.config_cli(mode, iw, ow, xtra, pw, ns)
            raise SystemExit("oracle accepts what the product refuses: %r"
                             % ((mode, iw, ow, xtra, pw, ns),))
        except ValueError:
            refused += 1
            continue
    ocfg = O.config_cli(mode, iw, ow, xtra, pw, ns)
    n = int(rng.choice([1, 3, 4, 5, 257, 1024, 4099, 8192, 12293, 65541]))
    x, y, ph = rand_inputs(rng, iw, pw, n)
    if rng.randint(2):
        # slow ramps: the rows on which the seeded kernel takes its stage
        # multipliers from the direction tails instead of the recurrence
        h = n // 2
        ramp = int(rng.randint(1 << pw)) + int(rng.choice([1, 2, 3, 17])) * np.arange(n - h)
        ph = ph.copy()
        ph[h:] = (ramp & ((1 << pw) - 1)).astype(ph.dtype)
    key = (mode, "wrap" if cfg.needs_wrap else ("wide" if cfg.ww > 35 else
           "lj" if cfg.ww > 32 else "narrow"))
    paths[key] = paths.get(key, 0) + 1
    if mode in (ca.P2R, ca.SP2R):
        a = gpu_p2r(cfg, x, y, ph)
        b = O.rotate(ocfg, x, y, ph)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (t, "p2r")
        x0, y0 = int(x[n // 2]), int(y[0])
        plan = ca.Plan(cfg)
        a = gpu_plan_p2r(plan, x0, y0, ph)
        b = O.rotate(ocfg, x0, y0, ph)
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
            bad = np.nonzero((a[0] != b[0]) | (a[1] != b[1]))[0]
            raise SystemExit("plan mismatch: cli=%r ww=%d nlive=%d x0=%d y0=%d n=%d "
                             "first bad %d phase=%#x gpu=(%d,%d) oracle=(%d,%d) "
                             "(%d bad)" % ((mode, iw, ow, xtra, pw, ns), cfg.ww,
                                           cfg.nlive, x0, y0, n, bad[0],
                                           int(ph[bad[0]]), a[0][bad[0]],
                                           a[1][bad[0]], b[0][bad[0]],
                                           b[1][bad[0]], bad.size))
        # per-sample vectors through the plan: directions looked up where the
        # core has the tables (cordic_xydir.h), cordic_p2r's kernel elsewhere
        import torch  # noqa: E402
        dx_, dy_, dp_ = (torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).cuda()
                         for v in (x, y, ph))
        oa = torch.zeros(n, dtype=torch.int32, device="cuda")
        ob = torch.zeros(n, dtype=torch.int32, device="cuda")
        plan.p2r(dx_, dy_, dp_, oa, ob)
        torch.cuda.synchronize()
        if plan.dir_groups:
            paths["dirs%d" % len(plan.dir_groups)] = paths.get(
                "dirs%d" % len(plan.dir_groups), 0) + 1
        b = O.rotate(ocfg, x, y, ph)
        if not (np.array_equal(oa.cpu().numpy(), b[0])
                and np.array_equal(ob.cpu().numpy(), b[1])):
            bad = np.nonzero((oa.cpu().numpy() != b[0]) | (ob.cpu().numpy() != b[1]))[0]
            raise SystemExit("plan.p2r mismatch: cli=%r ww=%d nlive=%d groups=%r n=%d "
                             "first bad %d x=%d y=%d phase=%#x (%d bad)"
                             % ((mode, iw, ow, xtra, pw, ns), cfg.ww, cfg.nlive,
                                plan.dir_groups, n, bad[0], x[bad[0]], y[bad[0]],
                                int(ph[bad[0]]), bad.size))
        fcw, p0, i0 = int(rng.randint(1 << 32)), int(rng.randint(1 << 32)), int(rng.randint(1 << 40))
        if rng.randint(2):
            fcw = int(rng.choice([1, 2, 5, (1 << pw) - 3])) & 0xffffffff   # slow NCO: tails
        a = gpu_plan_nco(plan, n, p0, fcw, i0, x0, y0)
        b = O.nco(ocfg, n, p0 & 0xffffffff, fcw, i0, x0, y0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (t, "nco")
        jobset_fuzz(rng, plan, cfg, ocfg, x, y, ph, paths)
        plan.close()
    else:
        a = gpu_r2p(cfg, x, y)
        b = O.topolar(ocfg, x, y)
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
            bad = np.nonzero((a[0] != b[0]) | (a[1] != b[1]))[0]
            raise SystemExit("r2p mismatch: cli=%r ww=%d nlive=%d n=%d first bad %d "
                             "x=%d y=%d gpu=(%d,%#x) oracle=(%d,%#x) (%d bad)"
                             % ((mode, iw, ow, xtra, pw, ns), cfg.ww, cfg.nlive, n,
                                bad[0], x[bad[0]], y[bad[0]], a[0][bad[0]],
                                int(a[1][bad[0]]) & 0xffffffff, b[0][bad[0]],
                                int(b[1][bad[0]]) & 0xffffffff, bad.size))
        plan = ca.Plan(cfg)
        jobset_fuzz(rng, plan, cfg, ocfg, x, y, ph, paths)
        plan.close()
    done += 1
print("fuzz ok: %d cores checked, %d refused by both; paths %s" % (done, refused, paths))

# ---- second part: the options around the path, on random cores ------------
import torch  # noqa: E402
from stream_model import PipeModel  # noqa: E402
from test_stream import _gpu_run  # noqa: E402

DEV = "cuda:0"
extra = {"unit_gain": 0, "io16": 0, "quad": 0, "quad_refused": 0, "stream": 0}

for t in range(max(ncfg // 4, 20)):
    # unit gain on a random (accepted) core
    mode = int(rng.randint(4))
    iw, ow = int(rng.randint(2, 33)), int(rng.randint(2, 33))
    xtra, pw, ns = int(rng.randint(0, 6)), int(rng.randint(6, 33)), int(rng.randint(2, 41))
    try:
        cfg = ca.Config.from_cli(mode, iw, ow, xtra, pw, ns)
    except ca.CordicError:
        continue
    ocfg = O.config_cli(mode, iw, ow, xtra, pw, ns)
    ug = cfg.with_flags(ca.FLAG_UNIT_GAIN)
    k = ca.lib().cordic_config_gain_annihilator(ug.ref)
    n = int(rng.choice([5, 1024, 4099]))
    x, y, ph = rand_inputs(rng, iw, pw, n)

    def sc(a):
        return ((a.astype(np.int64) * k) >> 32).astype(np.int32)
    if mode in (ca.P2R, ca.SP2R):
        a = gpu_p2r(ug, x, y, ph)
        b = O.rotate(ocfg, x, y, ph)
        assert np.array_equal(a[0], sc(b[0])) and np.array_equal(a[1], sc(b[1])), (t, "ug")
        plan = ca.Plan(ug)
        a = gpu_plan_p2r(plan, int(x[1]), int(y[2]), ph)
        b = O.rotate(ocfg, int(x[1]), int(y[2]), ph)
        assert np.array_equal(a[0], sc(b[0])) and np.array_equal(a[1], sc(b[1])), (t, "ugplan")
    else:
        a = gpu_r2p(ug, x, y)
        b = O.topolar(ocfg, x, y)
        assert np.array_equal(a[0], sc(b[0])) and np.array_equal(a[1], b[1]), (t, "ugr2p")
    extra["unit_gain"] += 1

    # 16-bit containers
    if iw <= 16 and ow <= 16 and pw <= 16:
        def d16(a):
            return torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)
                                    .astype(np.uint16).view(np.int16)).to(DEV)
        o0 = torch.zeros(n, dtype=torch.int16, device=DEV)
        o1 = torch.zeros(n, dtype=torch.int16, device=DEV)
        if mode in (ca.P2R, ca.SP2R):
            ca.p2r(cfg, d16(x), d16(y), d16(ph), o0, o1, n=n)
            b = O.rotate(ocfg, x, y, ph)
            torch.cuda.synchronize()
            assert np.array_equal(o0.cpu().numpy(), b[0].astype(np.int16)), (t, "io16")
            assert np.array_equal(o1.cpu().numpy(), b[1].astype(np.int16)), (t, "io16")
        else:
            ca.r2p(cfg, d16(x), d16(y), o0, o1, n=n)
            b = O.topolar(ocfg, x, y)
            torch.cuda.synchronize()
            assert np.array_equal(o0.cpu().numpy(), b[0].astype(np.int16)), (t, "io16")
            assert np.array_equal(o1.cpu().numpy().view(np.uint16),
                                  b[1].astype(np.uint16)), (t, "io16")
        extra["io16"] += 1

    # clocked view on the pipelined cores
    if mode in (ca.P2R, ca.R2P) and t % 3 == 0:
        rot = mode == ca.P2R
        nn = 3000
        lo, hi = -(1 << (iw - 1)), (1 << (iw - 1))
        sx, sy = rng.randint(lo, hi, nn), rng.randint(lo, hi, nn)
        sp = rng.randint(0, 1 << cfg.pw, nn, dtype=np.int64)
        ce = (rng.randint(0, 3, nn) != 0).astype(np.uint8)
        rs = (rng.randint(0, 400, nn) == 0).astype(np.uint8)
        ax = rng.randint(0, 2, nn).astype(np.uint8)
        s = ca.Stream(cfg)
        r = _gpu_run(s, rot, sx, sy, sp, ce, rs, ax, [7, 1000, 1001, 2500])
        m = PipeModel(ocfg, rot).run(sx, sy, sp, ce, rs, ax)
        assert all(np.array_equal(a, b) for a, b in zip(r, m)), (t, "stream")
        s.close()
        extra["stream"] += 1

for t in range(max(ncfg // 8, 20)):
    ow, xtra = int(rng.randint(3, 29)), int(rng.randint(0, 5))
    pw = int(rng.choice([-1, int(rng.randint(6, 33))]))
    try:
        q = ca.Quad(-1, ow, xtra, pw)
    except ca.CordicError:
        try:
            O.quad_cli(-1, ow, xtra, pw)
            raise SystemExit("oracle accepts a quadtbl core the product refuses: %r"
                             % ((ow, xtra, pw),))
        except ValueError:
            extra["quad_refused"] += 1
            continue
    oq = O.quad_cli(-1, ow, xtra, pw)
    ot = O.quad_tables(oq)
    assert all(np.array_equal(a, b) for a, b in zip(q.tables(), ot)), (t, "quadtables")
    n = 20000
    ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    d = torch.from_numpy(ph.view(np.int32)).to(DEV)
    o = torch.empty_like(d)
    q.lookup(d, o)
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy(), O.quad_lookup(oq, ot, ph)), (t, "quad", ow, xtra, pw)
    q.close()
    extra["quad"] += 1
print("fuzz ok (options):", extra)
