"""bench_paths.py -- the informational lines behind the default run: the other
BASELINE configurations at their sizes (child runs of bench.py) and the
host-array entry points beside the raw PCIe rates.  Never `value`."""
import json
import os
import sys
import time

import numpy as np
import torch

from bench_common import MODE, ROOT, WORKLOADS

BENCH = os.path.join(ROOT, "bench.py")

# BASELINE.json's other GPU configurations at THEIR sizes (configs[2..4]) plus
# the per-sample-vector rotator: (workload, log2 samples per launch)
OTHER_PATHS = (("cfg3", 30), ("cfg4", 30), ("cfg5", 32), ("p2rxy", 30),
               ("ddc", 30))


def other_paths(args, steps=24, warmup=4):
    """Driver-timed lines of the other configurations (single-GPU default run
    only; informational, never `value`): each one is this script run on that
    workload -- same timing discipline, HIP events around every launch, oracle
    spot checks and digest, hwmon clock, one SQ_INSTS_VALU pass -- as a child
    process once the main measurement is finished, condensed to its rate,
    roofline (HBM fraction AND valu_fraction, bound) and checks."""
    import subprocess
    import tempfile
    res = {}
    fd, dpath = tempfile.mkstemp(prefix="bench_other_", suffix=".json", dir="/tmp")
    os.close(fd)
    for wl, log2n in OTHER_PATHS:
        cmd = [sys.executable, BENCH, "--workload", wl, "--detail", dpath,
               "--steps", str(steps), "--warmup", str(warmup),
               "--log2-samples", str(log2n), "--input", args.input,
               "--no-cpu-baseline",
               "--pmc-counters", "SQ_INSTS_VALU+SQ_INSTS_VALU_INT64"]
        if args.no_pmc:
            cmd.append("--no-pmc")
        if args.no_power:
            cmd.append("--no-power")
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            json.loads(line[-1])              # it printed its line ...
            with open(dpath) as f:            # ... and this is what it selects from
                d = json.load(f)
        except Exception as e:                # never lose the main line
            res[wl] = {"error": repr(e)}
            continue
        roof = d["roofline"]
        e = {"Msamples_per_s": d["value"], "ms_per_step": d["ms_per_step"],
             "steps": d["steps"], "samples_per_launch": 1 << log2n,
             "bytes_per_sample": roof["bytes_per_sample"],
             "kernel": d["config"]["kernel"],
             "bit_exact_vs_oracle": d["bit_exact_vs_oracle"],
             "digest": d["digest"],
             "digest_check": {k: (d.get("digest_check") or {}).get(k) for k in (
                 "samples", "equal", "oracle", "oracle_seconds")},
             "roofline": {k: roof[k] for k in (
                 "bound", "limiter", "achieved", "peak", "unit", "frac",
                 "valu_fraction", "valu_issue_fraction", "valu", "kernel_ms_avg",
                 "kernel_ms_min") if k in roof},
             "wall_s": time.perf_counter() - t0}
        pw = (roof.get("power") or {}).get("sustained")
        if pw:
            e["sustained"] = {k: pw[k] for k in (
                "socket_w_median", "sclk_mhz_median") if k in pw}
        if "full_recurrence_kernel" in d:
            f = d["full_recurrence_kernel"]
            e["full_recurrence_kernel"] = {
                "Msamples_per_s": f["value_per_gpu"], "hbm_frac": f["hbm_frac"],
                "outputs_identical_to_seeded_kernel":
                    f["outputs_identical_to_seeded_kernel"]}
        res[wl] = e
    try:
        os.unlink(dpath)
    except OSError:
        pass
    return res


def host_paths(log2n=28, reps=3):
    """The host-array entry points (cordic_p2r_host / cordic_r2p_host: what a
    caller holding the reference bench's plain `int` arrays uses,
    bench/cpp/cordic_tb.cpp:94-178) timed beside the raw PCIe rates of this
    box: pinned 1 GiB hipMemcpy each way, then BASELINE config 2's core on
    2^log2n host samples -- pinned arrays (DMA'd in place) and pageable numpy
    arrays (staged by the library's copy threads) -- and config 3's converter.
    Informational, never `value`: inputs start in HOST memory here.  Outputs
    are checked against the oracle's digest of every sample."""
    import ctypes as C
    import cordic_amd as ca
    import oracle_lib as O
    n = 1 << log2n
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    dev = torch.empty(n, dtype=torch.int32, device="cuda")
    pin = [ca.HostArray(n, "int32") for _ in range(4)]
    res = {"samples": n, "reps": reps}

    def best(fn):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return min(ts)
    pin[0].array[:] = 1
    h2d = best(lambda: hip.hipMemcpy(dev.data_ptr(), pin[0].array.ctypes.data,
                                     n * 4, 1))
    d2h = best(lambda: hip.hipMemcpy(pin[0].array.ctypes.data, dev.data_ptr(),
                                     n * 4, 2))
    res["pcie"] = {"h2d_GBps": n * 4 / h2d / 1e9, "d2h_GBps": n * 4 / d2h / 1e9,
                   "what": "hipMemcpy of %d MiB, pinned host memory, best of "
                           "%d" % (n * 4 >> 20, reps)}
    del dev

    def line(seconds, up, down, want, got):
        # the direction that takes longer at the raw rates is the bound
        t_up = up * n / (res["pcie"]["h2d_GBps"] * 1e9)
        t_down = down * n / (res["pcie"]["d2h_GBps"] * 1e9)
        return {"Msamples_per_s": n / seconds / 1e6, "seconds": seconds,
                "up_GBps": up * n / seconds / 1e9,
                "down_GBps": down * n / seconds / 1e9,
                "frac_of_slower_pcie_direction": max(t_up, t_down) / seconds,
                "stats": {k: v for k, v in ca.host_last_stats().items()
                          if k != "seconds"},
                "digest_equals_oracle": want == got}

    def dig(a, b):
        return (O.digest_words(a, 0) + O.digest_words(b, 1 << 40)) % (1 << 64)
    # config 2: constant vector, phase ramp n << 2
    m, iw, ow, xtra, pw, ns = WORKLOADS["cfg2"]["cli"]
    cfg = ca.Config.from_cli(MODE[m], iw, ow, xtra, pw, ns)
    ocfg = O.config_cli(MODE[m], iw, ow, xtra, pw, ns)
    x0 = (1 << (iw - 1)) - 1
    ramp = (np.arange(n, dtype=np.uint32) << np.uint32(2))
    want, _ = O.job_digest(ocfg, "p2r", 0, n, 0, 4, x0, 0)
    pin[0].array.view(np.uint32)[:] = ramp
    out = (pin[1].array, pin[2].array)
    ca.p2r_host(cfg, x0, 0, pin[0].array.view(np.uint32), out=out)   # set-up
    t = best(lambda: ca.p2r_host(cfg, x0, 0, pin[0].array.view(np.uint32),
                                 out=out))
    res["p2r_const_pinned"] = line(t, 4, 8, want, dig(*out))
    # the same call spread over every device of the node (round 5:
    # cordic_host_set_devices -- one pipeline, one PCIe link per device); on a
    # one-GPU box two lanes SHARE the device and its link, which exercises the
    # code and should cost nothing
    ndev = ca.device_count()
    devs = list(range(ndev)) if ndev > 1 else [0, 0]
    ca.host_set_devices(devs)
    try:
        ca.p2r_host(cfg, x0, 0, pin[0].array.view(np.uint32), out=out)
        t = best(lambda: ca.p2r_host(cfg, x0, 0, pin[0].array.view(np.uint32),
                                     out=out))
        e = line(t, 4, 8, want, dig(*out))
        e["devices"] = devs
        e["per_lane"] = []
        for k in range(len(devs)):
            ls = ca.host_lane_stats(k)
            e["per_lane"].append({
                "device": devs[k], "samples": ls["samples"],
                "Msamples_per_s": (ls["samples"] / ls["seconds"] / 1e6
                                   if ls["seconds"] else None)})
        res["p2r_const_pinned_all_devices" if ndev > 1 else
            "p2r_const_pinned_two_lanes_one_device"] = e
    finally:
        ca.host_set_devices([])
    pa, pb = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    ca.p2r_host(cfg, x0, 0, ramp, out=(pa, pb))
    t = best(lambda: ca.p2r_host(cfg, x0, 0, ramp, out=(pa, pb)))
    res["p2r_const_pageable"] = line(t, 4, 8, want, dig(pa, pb))
    # config 3: converter on the I/Q ramps (8 B up, 8 B down)
    m, iw, ow, xtra, pw, ns = WORKLOADS["cfg3"]["cli"]
    cfg = ca.Config.from_cli(MODE[m], iw, ow, xtra, pw, ns)
    ocfg = O.config_cli(MODE[m], iw, ow, xtra, pw, ns)
    g = np.arange(n, dtype=np.uint32)
    sh = 32 - iw
    for k, mul in ((0, O.IQ_MULX), (1, O.IQ_MULY)):
        with np.errstate(over="ignore"):
            v = ((g * np.uint32(mul)) >> np.uint32(8)) << np.uint32(sh)
        pin[k].array[:] = v.view(np.int32) >> sh
    want, _ = O.job_digest(ocfg, "r2p", 0, n)
    out = (pin[2].array, pin[3].array.view(np.uint32))
    ca.r2p_host(cfg, pin[0].array, pin[1].array, out=out)
    t = best(lambda: ca.r2p_host(cfg, pin[0].array, pin[1].array, out=out))
    res["r2p_pinned"] = line(t, 8, 8, want, dig(*out))
    for h in pin:
        h.close()
    ca.host_release()
    return res


def small_batches():
    """What a caller with SMALL batches gets: BASELINE
    config 2's core on 2^26 resident samples handed over (a) as one plan call
    per job of 2^16 / 2^20 / 2^24 samples, back to back on one stream, and (b)
    as ONE job set (cordic_jobset: a single launch walks every job's tiles).
    Rates are aggregate over the 2^26 samples; the job sets' outputs are
    checked against the oracle's digest of the whole ramp.  Informational,
    never `value`."""
    import cordic_amd as ca
    import oracle_lib as O
    from gpu_util import gpu_digest
    m, iw, ow, xtra, pw, ns = WORKLOADS["cfg2"]["cli"]
    cfg = ca.Config.from_cli(MODE[m], iw, ow, xtra, pw, ns)
    ocfg = O.config_cli(MODE[m], iw, ow, xtra, pw, ns)
    plan = ca.Plan(cfg)
    plan.set_min_samples(-1)            # the library's own choice of kernel
    total = 1 << 26
    x0 = (1 << (iw - 1)) - 1
    ph = torch.empty(total, dtype=torch.int32, device="cuda")
    a = torch.empty_like(ph)
    b = torch.empty_like(ph)
    ca.fill_phase_ramp(ph, 0, 2)
    want, _ = O.job_digest(ocfg, "p2r", 0, total, 0, 4, x0, 0)
    res = {"samples": total, "core": "cfg2", "rows": {}}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    for lg in (16, 20, 24):
        n = 1 << lg
        nj = total // n
        jobs = [dict(phase=ph[k * n:(k + 1) * n], ox=a[k * n:(k + 1) * n],
                     oy=b[k * n:(k + 1) * n], n=n) for k in range(nj)]

        def one_by_one():
            for jb in jobs:
                plan.p2r_const(x0, 0, jb["phase"], jb["ox"], jb["oy"])
        ms_calls = timed(one_by_one, 3)
        js = ca.Jobset(plan, ca.JOBS_PHASE_ARRAYS, jobs)
        a.zero_()
        b.zero_()
        ms_set = timed(lambda: js.run(x0, 0), 10)
        got = (gpu_digest(a, 0) + gpu_digest(b, 1 << 40)) % (1 << 64)
        js.close()
        res["rows"]["2^%d" % lg] = {
            "jobs": nj, "samples_per_job": n,
            "one_call_per_job_Msamples_per_s": total / ms_calls / 1e3,
            "one_call_per_job_us_per_call": ms_calls * 1e3 / nj,
            "job_set_Msamples_per_s": total / ms_set / 1e3,
            "job_set_ms": ms_set,
            "job_set_digest_equals_oracle": got == want}
    plan.close()
    return res


def small_batches_xy(log2_total=26, log2_job=16):
    """The same question for the DATA-FED calls (round 6; VERDICT r05 missing
    4): BASELINE config 3's converter, the per-sample-vector rotator and the
    fused mixer on 2^26 resident samples of the bench's I/Q ramps handed over
    as 1024 jobs of 2^16 -- (a) one call per job, (b) ONE job set
    (cordic_jobset: CORDIC_JOBS_R2P / _P2R_XY / _MIX), beside (c) the same
    samples as ONE long call.  Job-set outputs are checked against the
    oracle's digest of the whole ramp.  Informational, never `value`."""
    import cordic_amd as ca
    import oracle_lib as O
    from gpu_util import gpu_digest
    total, n = 1 << log2_total, 1 << log2_job
    nj = total // n
    x = torch.empty(total, dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    ph = torch.empty_like(x)
    a = torch.empty_like(x)
    b = torch.empty_like(x)
    fcw = 0x01234567
    res = {"samples": total, "jobs": nj, "samples_per_job": n, "rows": {}}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    for kind, wl in (("r2p", "cfg3"), ("mix", "cfg2"), ("p2rxy", "cfg2")):
        m, iw, ow, xtra, pw, ns = WORKLOADS[wl]["cli"]
        cfg = ca.Config.from_cli(MODE[m], iw, ow, xtra, pw, ns)
        ocfg = O.config_cli(MODE[m], iw, ow, xtra, pw, ns)
        plan = ca.Plan(cfg)
        plan.set_min_samples(-1)        # the library's own choice of kernel
        ca.fill_iq_ramp(x, y, 0, O.IQ_MULX, O.IQ_MULY, iw)
        ca.fill_phase_ramp(ph, 0, 2)
        K = {"r2p": ca.JOBS_R2P, "mix": ca.JOBS_MIX, "p2rxy": ca.JOBS_P2R_XY}[kind]
        jobs = []
        for k in range(nj):
            sl = slice(k * n, (k + 1) * n)
            jb = dict(x=x[sl], y=y[sl], ox=a[sl], oy=b[sl], n=n)
            if kind == "p2rxy":
                jb["phase"] = ph[sl]
            elif kind == "mix":
                jb.update(phase0=0, fcw=fcw, index0=k * n)
            jobs.append(jb)

        def call(jb):
            if kind == "r2p":
                ca.r2p(cfg, jb["x"], jb["y"], jb["ox"], jb["oy"])
            elif kind == "mix":
                plan.mix(jb["phase0"], jb["fcw"], jb["index0"], jb["x"], jb["y"],
                         jb["ox"], jb["oy"])
            else:
                plan.p2r(jb["x"], jb["y"], jb["phase"], jb["ox"], jb["oy"])

        def one_by_one():
            for jb in jobs:
                call(jb)
        whole = dict(x=x, y=y, ox=a, oy=b, n=total, phase=ph, phase0=0, fcw=fcw,
                     index0=0)
        ms_calls = timed(one_by_one, 2)
        ms_long = timed(lambda: call(whole), 10)
        js = ca.Jobset(plan, K, jobs)
        a.zero_()
        b.zero_()
        ms_set = timed(lambda: js.run(), 10)
        got = (gpu_digest(a, 0) + gpu_digest(b, 1 << 40)) % (1 << 64)
        want, _ = O.job_digest(ocfg, kind, 0, total, 0, 4 if kind == "p2rxy" else fcw)
        info = js.info
        js.close()
        plan.close()
        res["rows"][kind] = {
            "core": wl, "tiles": info["tiles"],
            "one_call_per_job_Msamples_per_s": total / ms_calls / 1e3,
            "one_call_per_job_us_per_call": ms_calls * 1e3 / nj,
            "job_set_Msamples_per_s": total / ms_set / 1e3,
            "job_set_ms": ms_set,
            "one_long_call_Msamples_per_s": total / ms_long / 1e3,
            "job_set_over_one_long_call": ms_long / ms_set,
            "job_set_digest_equals_oracle": got == want}
    return res


def small_batches_nco(log2_total=26, log2_job=16):
    """NCO BANKS as job sets: 1024 NCO jobs of 2^16
    samples on BASELINE's 16- and 24-stage cores -- one job set against ONE
    long cordic_plan_nco call over the same 2^26 samples, with a slow increment
    (rows take the direction tails), with 0x01234567 (cfg5's; on 24 stages the
    rows run the recurrence behind the seeds), and with a DIFFERENT increment
    per job (what a bank is; no single long call to compare with).  Since
    round 6 the 24-stage sets run the static descriptor instance: the kernel's
    per-row test chooses the tails job by job.  Equal-increment sets are
    checked against the oracle's digest of the one long job."""
    import cordic_amd as ca
    import oracle_lib as O
    from gpu_util import gpu_digest
    total, n = 1 << log2_total, 1 << log2_job
    nj = total // n
    a = torch.empty(total, dtype=torch.int32, device="cuda")
    b = torch.empty_like(a)
    res = {"samples": total, "jobs": nj, "samples_per_job": n, "rows": {}}

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    for wl in ("cfg2", "cfg4"):
        m, iw, ow, xtra, pw, ns = WORKLOADS[wl]["cli"]
        cfg = ca.Config.from_cli(MODE[m], iw, ow, xtra, pw, ns)
        ocfg = O.config_cli(MODE[m], iw, ow, xtra, pw, ns)
        plan = ca.Plan(cfg)
        plan.set_min_samples(-1)
        x0 = (1 << (iw - 1)) - 1
        for name, fcw in (("slow", 0x00000123), ("cfg5", 0x01234567), ("bank", None)):
            jobs = [dict(ox=a[k * n:(k + 1) * n], oy=b[k * n:(k + 1) * n], n=n,
                         phase0=0, index0=k * n if fcw is not None else 0,
                         fcw=fcw if fcw is not None else
                         (0x00000101 * (k + 1)) & 0xffffffff)
                    for k in range(nj)]
            js = ca.Jobset(plan, ca.JOBS_NCO, jobs)
            a.zero_()
            b.zero_()
            ms_set = timed(lambda: js.run(x0, 0))
            row = {"core": wl, "stages": ns, "job_set_ms": ms_set,
                   "job_set_Msamples_per_s": total / ms_set / 1e3}
            if fcw is not None:
                got = (gpu_digest(a, 0) + gpu_digest(b, 1 << 40)) % (1 << 64)
                want, _ = O.job_digest(ocfg, "nco", 0, total, 0, fcw, x0, 0)
                ms_long = timed(lambda: plan.nco(total, 0, fcw, 0, x0, 0, a, b))
                row.update({"one_long_call_Msamples_per_s": total / ms_long / 1e3,
                            "job_set_over_one_long_call": ms_long / ms_set,
                            "job_set_digest_equals_oracle": got == want})
            js.close()
            res["rows"]["%s_%s" % (wl, name)] = row
        plan.close()
    return res

