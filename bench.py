#!/usr/bin/env python3
"""bench.py -- throughput of the CORDIC rotation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM.  The default workload is BASELINE.json configs[1]:
basiccordic 16-stage, 32-bit phase -> 32-bit sin/cos, 2^30 samples per GPU,
phase[n] = (uint32)(n << 2) (the reference bench's ramp, cordic_tb.cpp:138),
x = 2^31-1, y = 0.  Multi-GPU: independent shards by global sample index,
no data-path collective (weak scaling), driven through the C++ cordic_group
layer of the C ABI; a digest all-reduce after the timed region checks the
shards, `--gather` additionally times collecting the outputs on one GPU.

`--gpus N` always means N GPUs: started by torch.distributed.run the world
size must equal N; started plainly with N > 1 the script re-executes itself
under torch.distributed.run with N ranks (one GPU each); fewer than N visible
GPUs is an error, never a silent 1-GPU run.  `--single-process` instead
drives all N devices from one host process (cordic_group, no process group).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
import build_stamp  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# TEST SWITCH (tests/test_bench_launch.py): BENCH_TEST_SHARE_GPU=1 lets N ranks
# share device 0 so that the N > 1 code path of this script -- rank / world
# bookkeeping, shard ranges, max-over-ranks timing, digest reduction, the
# one-process block -- can execute on a one-GPU box.  RCCL refuses two ranks on
# one device, so the process group is gloo and its tensors live on the host.
# Never set by the driver; a line produced this way says so in `launch.mode`.
SHARE_GPU = os.environ.get("BENCH_TEST_SHARE_GPU") == "1"


def dist_init(dist, rank, world, local):
    if SHARE_GPU:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))


def coll_device(dev):
    """where the tensors of the (tiny) collectives live"""
    return torch.device("cpu") if SHARE_GPU else dev

# VALU side of the roofline ("report roofline.achieved (HBM) AND
# valu_fraction").  1024 SIMDs x 64 lanes; in the mixed integer stream of a
# micro-rotation a SIMD with 8 resident waves issues one VALU wave-instruction
# every 3.65 cycles whatever the opcode (tools/stage_microbench.hip,
# profiles/r02/stage_microbench.txt), at the shader clock the kernel actually
# held (the hot kernels sit at the 1400 W socket limit below 2.4 GHz).
N_SIMD, WAVE_LANES = 1024, 64
VALU_ISSUE_CYCLES = 3.65
SCLK_MAX_GHZ = 2.4


def valu_block(samples_per_s, instr_per_sample, sclk_ghz, instr_source,
               sclk_source):
    """valu_fraction = lane-instructions/s the kernel retired / what the
    SIMDs can issue for this instruction mix at the clock it ran at."""
    if not instr_per_sample:
        return None
    clk = sclk_ghz or SCLK_MAX_GHZ
    peak = N_SIMD * WAVE_LANES * clk * 1e9 / VALU_ISSUE_CYCLES
    ach = samples_per_s * instr_per_sample
    return {"instr_per_sample": instr_per_sample,
            "instr_source": instr_source,
            "sclk_ghz": clk, "sclk_source": sclk_source if sclk_ghz else
            "nominal maximum (no hwmon samples)",
            "issue_cycles_per_wave_instr": VALU_ISSUE_CYCLES,
            "achieved_Tinstr_per_s": ach / 1e12,
            "peak_Tinstr_per_s": peak / 1e12,
            "frac": ach / peak,
            "frac_at_2.4GHz": ach / (peak * SCLK_MAX_GHZ / clk)}


def add_valu(roof, samples_per_s, pm, power, prof):
    """roofline.valu / valu_fraction / bound from this run's SQ_INSTS_VALU
    pass (or, without one, the committed profile) and this run's clock."""
    instr = src = None
    if pm and pm.get("valu_instr_per_sample"):
        instr, src = pm["valu_instr_per_sample"], (
            "SQ_INSTS_VALU x 64 / samples, rocprofv3 --pmc pass of this run")
    elif prof and prof.get("valu_instr_per_sample"):
        instr, src = prof["valu_instr_per_sample"], (
            "committed profile (%s), not re-measured" % prof.get("source"))
    sclk = ssrc = None
    for key in ("sustained", "timed_region"):
        if power and power.get(key) and power[key].get("sclk_mhz_median"):
            sclk = power[key]["sclk_mhz_median"] / 1e3
            ssrc = "hwmon freq1_input median, %s window of this run" % key
            break
    vb = valu_block(samples_per_s, instr, sclk, src, ssrc)
    if vb:
        roof["valu"] = vb
        roof["valu_fraction"] = vb["frac"]
        # whichever ceiling the kernel sits nearer to
        roof["bound"] = "hbm" if roof["frac"] >= vb["frac"] else "valu"
        roof["bound_note"] = ("hbm frac %.3f vs valu_fraction %.3f; the VALU "
                              "ceiling is at the clock the 1400 W socket limit "
                              "allowed" % (roof["frac"], vb["frac"]))
    return roof

# name -> (gencordic-style parameters, bytes/sample, VALU ops/sample counted
# in the ISA of the kernel that runs it, description)
WORKLOADS = {
    "cfg2": dict(kind="p2r", cli=("p2r", 32, 32, 2, 32, 16), bytes=12,
                 shift=2, desc="basiccordic 16-stage, 32-bit phase -> 32-bit "
                 "sin/cos, phase ramp n<<2, x=2^31-1, y=0"),
    "cfg1": dict(kind="p2r", cli=("p2r", 16, 16, 2, 16, 16), bytes=6,
                 shift=0, io16=True, desc="basiccordic 16-bit (WW19 PW16, 13 "
                 "live stages), int16/uint16 sample arrays, phase ramp "
                 "n mod 2^16, x=32767, y=0"),
    "cfg4": dict(kind="p2r", cli=("p2r", 32, 32, 2, 32, 24), bytes=12,
                 shift=0, desc="basiccordic 24-stage, 32-bit, phase ramp n"),
    "p2rxy": dict(kind="p2rxy", cli=("p2r", 32, 32, 2, 32, 16), bytes=20,
                  shift=2, desc="basiccordic 16-stage, 32-bit, per-sample x, y "
                  "and phase vectors (cordic_p2r)"),
    "sintbl": dict(kind="tbl", table=(4, -1, 13, 17), bytes=8, shift=0,
                   desc="sintable PW=17 OW=13 (rtl/sintable.v), phase ramp n"),
    "qtrtbl": dict(kind="tbl", table=(5, -1, 24, 18), bytes=8, shift=0,
                   desc="quarterwav PW=18 OW=24 (rtl/quarterwav.v), phase "
                   "ramp n"),
    "qtrtbl24": dict(kind="tbl", table=(5, -1, 24, 17), bytes=8, shift=0,
                     desc="quarterwav PW=17 OW=24 (32-bit entries in LDS, "
                     "128 KiB), phase ramp n"),
    "qtrtbl16": dict(kind="tbl", table=(5, -1, 16, 17), bytes=8, shift=0,
                     desc="quarterwav PW=17 OW=16 (int16 copy in LDS), phase "
                     "ramp n"),
    "quadtbl": dict(kind="tbl", quad=(-1, 13, 2, 18), bytes=8, shift=0,
                    desc="quadtbl PW=18 OW=13 (rtl/quadtbl.v: 64-entry C/L/Q "
                    "tables + quadratic interpolation), phase ramp n"),
    "quadtbl24": dict(kind="tbl", quad=(-1, 24, 2, 32), bytes=8, shift=0,
                      desc="quadtbl PW=32 OW=24 (512-entry tables), phase "
                      "ramp n"),
    "cfg3": dict(kind="r2p", cli=("r2p", 24, 24, 2, -1, 20), bytes=16,
                 desc="topolar 20-stage, 24-bit I/Q ramps -> mag + phase"),
    # the cores gencordic derives when -p / -n are left to it (the ones that
    # pass the reference's acceptance criteria)
    "nat32": dict(kind="p2r", cli=("p2r", 32, 32, 2, 32, -1), bytes=12, shift=2,
                  desc="gencordic -t p2r -i 32 -o 32 -p 32: 29 stages, phase "
                  "ramp n<<2"),
    "nat24": dict(kind="p2r", cli=("p2r", 24, 24, 2, -1, -1), bytes=12, shift=0,
                  desc="gencordic -t p2r -i 24 -o 24: WW27 PW31, 27 stages, "
                  "phase ramp n"),
    "nat16": dict(kind="p2r", cli=("p2r", 16, 16, 2, -1, -1), bytes=12, shift=0,
                  desc="gencordic -t p2r -i 16 -o 16: WW19 PW23, 19 stages, "
                  "32-bit containers, phase ramp n"),
    "natr2p24": dict(kind="r2p", cli=("r2p", 24, 24, 2, -1, -1), bytes=16,
                     desc="gencordic -t r2p -i 24 -o 24: WW32 PW32, 29 stages"),
    "cfg5": dict(kind="nco", cli=("p2r", 32, 32, 2, 32, 16), bytes=8,
                 desc="fused NCO (phase = n*0x01234567) + 16-stage p2r, "
                 "store only"),
    "cfg5seq": dict(kind="nco", cli=("sp2r", 32, 32, 2, 32, 16), bytes=8,
                    desc="fused NCO + seqcordic arithmetic (NSTAGES-2)"),
}
MODE = {"p2r": 0, "r2p": 1, "sp2r": 2, "sr2p": 3}


def _usable_cpus():
    """Hardware threads this process may actually use: the affinity mask, cut
    down to the cgroup CPU quota (the gpurun boxes show 256 CPUs but grant
    16 CPU-seconds per second; 256 busy threads under that quota measured
    half the rate of 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def ranks_on_this_node(world):
    """processes that share this node's host cores with us"""
    try:
        return max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    except ValueError:
        return max(1, world)


def oracle_digest_leg(args, w, ocfg, start, n, x0, y0, threads=None):
    """The oracle's digest of ALL n samples this rank computed (threaded
    orc_digest: every sample through the scalar restatement, condensed by the
    device's position-aware digest).  This CPU work is also the cpu_baseline
    sample, so it is done once.  None for inputs the oracle cannot regenerate
    (--input random) and for 16-bit containers (two samples per word)."""
    import oracle_lib as O
    kind = w["kind"]
    if (args.input != "ramp" or w.get("io16") or kind == "tbl"
            or getattr(args, "no_full_digest", False)):
        return None
    fcw = 0x01234567 if kind == "nco" else (1 << w.get("shift", 0))
    cores = threads or _usable_cpus()
    d, secs = O.job_digest(ocfg, kind, start, n, 0, fcw, x0, y0,
                           threads=cores)
    return {"digest": d, "samples": n, "seconds": secs, "cores": cores}


def reduce_digest_legs(dist, dev, world, local_ok, leg):
    """Every rank has compared ITS shards with the oracle: (all equal?, samples
    compared in all, sum of the oracle's digests mod 2^64, slowest leg)."""
    if dist is None or world == 1:
        return local_ok, leg["samples"], leg["digest"], leg["seconds"]
    od = leg["digest"]
    t = torch.tensor([1 if local_ok else 0, leg["samples"],
                      od - (1 << 64) if od >= 1 << 63 else od],
                     dtype=torch.int64, device=coll_device(dev))
    mn = t[:1].clone()
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    sm = t[1:].clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    sec = torch.tensor([leg["seconds"]], dtype=torch.float64,
                       device=coll_device(dev))
    dist.all_reduce(sec, op=dist.ReduceOp.MAX)
    return (bool(mn.item()), int(sm[0].item()),
            int(sm[1].item()) & 0xFFFFFFFFFFFFFFFF, float(sec.item()))


def cpu_baseline(workload, seconds=12.0, leg=None):
    """The oracle (a restatement of the reference RTL, NOT reference code:
    the reference has no CPU compute path) timed on
    the host cores of this box on a bounded sample of the same workload:
    oracle/cordic_oracle.c:orc_throughput runs one POSIX thread per hardware
    thread, each pushing 2^16-sample blocks through the scalar oracle until
    `seconds` have passed."""
    import ctypes as C
    import oracle_lib as O
    w = WORKLOADS[workload]
    if w["kind"] == "tbl":
        return None
    m, iw, ow, xtra, pw, ns = w["cli"]
    ocfg = O.config_cli(MODE[m], iw, ow, xtra, pw, ns)
    L = O.lib()
    cores = _usable_cpus()
    kind = 1 if w["kind"] == "r2p" else 0
    mul = 0x01234567 if w["kind"] == "nco" else (1 << w.get("shift", 0))
    x0 = (1 << (iw - 1)) - 1
    t0 = time.perf_counter()
    n1 = L.orc_throughput(C.byref(ocfg), kind, 1, 1.0, mul, x0, 0)
    one = n1 / (time.perf_counter() - t0)
    if leg is not None and leg["seconds"] >= 1.0:
        # the digest leg already pushed every sample of this run through the
        # oracle on all cores: that IS the bounded sample (not done twice)
        total, wall, cores = leg["samples"], leg["seconds"], leg["cores"]
        sample = ("all %d samples of this run's %s workload (%d threads "
                  "drawing 2^16-sample blocks, %.1f s), whose outputs' digest "
                  "is what digest_check compares with the device's; includes "
                  "making the inputs and the digest (~4 %% of the work)"
                  % (total, workload, cores, wall))
    else:
        t0 = time.perf_counter()
        total = L.orc_throughput(C.byref(ocfg), kind, cores, seconds, mul,
                                 x0, 0)
        wall = time.perf_counter() - t0
        sample = ("%d samples of the %s workload (%d threads x 2^16-sample "
                  "blocks for %.0f s)" % (total, workload, cores, seconds))
    return {
        "value": total / wall / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": sample + " through oracle/liboracle.so: gcc -O2 scalar "
                  "restatement of the reference RTL -- the reference itself "
                  "has no CPU compute path",
        "value_1thread": one / 1e6,
        "cpu": _cpu_model(),
        "cpus_visible": os.cpu_count(),
    }


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# BASELINE.json's other GPU configurations at THEIR sizes (configs[2..4]) plus
# the per-sample-vector rotator: (workload, log2 samples per launch)
OTHER_PATHS = (("cfg3", 30), ("cfg4", 30), ("cfg5", 32), ("p2rxy", 30))


def other_paths(args, steps=24, warmup=4):
    """Driver-timed lines of the other configurations (single-GPU default run
    only; informational, never `value`): each one is this script run on that
    workload -- same timing discipline, HIP events around every launch, oracle
    spot checks and digest, hwmon clock, one SQ_INSTS_VALU pass -- as a child
    process once the main measurement is finished, condensed to its rate,
    roofline (HBM fraction AND valu_fraction, bound) and checks."""
    import subprocess
    res = {}
    for wl, log2n in OTHER_PATHS:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl,
               "--steps", str(steps), "--warmup", str(warmup),
               "--log2-samples", str(log2n), "--input", args.input,
               "--no-cpu-baseline", "--no-other-paths", "--no-copy-probe",
               "--pmc-counters", "SQ_INSTS_VALU"]
        if args.no_pmc:
            cmd.append("--no-pmc")
        if args.no_power:
            cmd.append("--no-power")
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            d = json.loads(line[-1])
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
                 "bound", "achieved", "peak", "unit", "frac", "valu_fraction",
                 "valu", "kernel_ms_avg", "kernel_ms_min") if k in roof},
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


def _profile_entry(key):
    """Counters of a workload from the COMMITTED rocprofv3 passes
    (profiles/pmc_latest.json; tools/profile_workload.sh produced them in an
    earlier gpurun session) -- not measured by this run, and labelled so."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            db = json.load(f)
        return db.get(key), db.get("_source", "profiles/pmc_latest.json")
    except (OSError, ValueError):
        return None, None


def from_profile(key, samples_per_launch=1 << 30):
    """{"source": ..., "hbm_bytes_per_launch": ..., "valu_instr_per_sample": ...}
    for the bench line's `from_profile` block (FETCH_SIZE x 2 +
    WRITE_SIZE, SQ_INSTS_VALU x 64 lanes / samples), or None."""
    e, src = _profile_entry(key)
    if not e:
        return None
    out = {"source": src, "note": "rocprofv3 PMC passes of an earlier session "
           "on this kernel, not re-measured by this run"}
    if "hbm_bytes_per_launch" in e:
        out["hbm_bytes_per_launch"] = e["hbm_bytes_per_launch"]
    if "SQ_INSTS_VALU" in e:
        out["valu_instr_per_sample"] = (e["SQ_INSTS_VALU"] * 64.0
                                        / samples_per_launch)
    # clock the chip sustained under this kernel (GRBM_GUI_ACTIVE / 8 XCDs /
    # dispatch duration of the PMC pass; the kernels run at the 1400 W socket
    # limit) and VALU issue interval per SIMD at that clock
    for k in ("shader_clock_ghz", "valu_cycles_per_inst"):
        if k in e:
            out[k] = round(e[k], 3)
    return out


KERNEL_OF = {"cfg2": "rotator_seeded", "cfg4": "rotator_seeded",
             "cfg5": "rotator_seeded", "cfg5seq": "rotator_seeded",
             "cfg1": "rotator_seeded", "cfg3": "topolar_lj",
             "nat32": "rotator_seeded", "nat24": "rotator_seeded",
             "nat16": "rotator_seeded", "natr2p24": "topolar_lj",
             "p2rxy": "rotator_xydir", "quadtbl": "quad_lookup",
             "quadtbl24": "quad_lookup", "sintbl": "table_lookup",
             "qtrtbl": "table_lookup", "qtrtbl16": "table_lookup",
             "qtrtbl24": "table_lookup"}


def measure_pmc(args):
    """HBM bytes per launch and VALU instructions per sample of the workload's
    kernel, MEASURED now: separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE,
    SQ_INSTS_VALU -- the first two do not fit one pass, and PMC is never
    combined with tracing) over a 3-step run of this same script, corrected as MI355X_MICROARCH.md prescribes for gfx950
    (FETCH_SIZE counts half of a wide coalesced read; both are in KiB)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not found"}
    kern = KERNEL_OF.get(args.workload)
    if args.no_seed and kern == "rotator_seeded":
        kern = "rotator_unrolled"
    if args.no_tails and kern == "rotator_xydir":
        kern = "rotator_unrolled"
    base = [sys.executable, os.path.abspath(__file__), "--workload",
            args.workload, "--steps", "3", "--warmup", "1", "--log2-samples",
            str(args.log2_samples), "--input", args.input, "--no-cpu-baseline",
            "--no-other-paths", "--no-copy-probe", "--no-pmc", "--no-power",
            "--no-full-digest"]
    for flag, on in (("--no-seed", args.no_seed), ("--generic", args.generic),
                     ("--static-chunks", args.static_chunks),
                     ("--no-tails", args.no_tails)):
        if on:
            base.append(flag)
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    counters = [c for c in args.pmc_counters.split(",") if c]
    for ctr in counters:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            r = subprocess.run(["rocprofv3", "--pmc", ctr, "--output-format",
                                "csv", "-d", td, "--"] + base, cwd="/tmp",
                               env=env, capture_output=True, text=True,
                               timeout=600)
            rows = []
            for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"),
                               recursive=True):
                for row in csv.DictReader(open(f)):
                    if (row["Counter_Name"] == ctr and kern
                            and kern in row["Kernel_Name"]):
                        rows.append(float(row["Counter_Value"]))
            if not rows:
                return {"error": "no %s rows for %s (rocprofv3 rc %d)"
                        % (ctr, kern, r.returncode)}
            vals[ctr] = (sum(rows) / len(rows), len(rows))
    out = {"kernel": kern, "passes": counters}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        fetch, write = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
        out.update({
            "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
            "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
            "launches_averaged": vals["FETCH_SIZE"][1],
            "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM), "
                          "WRITE_SIZE as reported, KiB -> B"})
    if "SQ_INSTS_VALU" in vals:
        out["SQ_INSTS_VALU_per_launch"] = vals["SQ_INSTS_VALU"][0]
        out["valu_instr_per_sample"] = (vals["SQ_INSTS_VALU"][0] * 64.0
                                        / float(1 << args.log2_samples))
    return out


_probe_lib = None


# The contract is ONE JSON line on stdout.  Libraries write there too (RCCL
# prints a five-line version banner when its first communicator comes up), so
# main() points file descriptor 1 at stderr for the life of the process and
# emit() writes the line to the original stdout.
_STDOUT_FD = None


def claim_stdout():
    global _STDOUT_FD
    if _STDOUT_FD is None:
        sys.stdout.flush()
        _STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    if _STDOUT_FD is None:
        print(line)
        return
    data = (line + "\n").encode()
    while data:
        data = data[os.write(_STDOUT_FD, data):]


class PowerSampler(threading.Thread):
    """Socket power and shader clock of one GPU while it works, read by a host
    thread from the amdgpu hwmon files of THAT device (matched by PCI bus id):
    power1_input (microwatts), freq1_input (sclk, Hz), power1_cap (the limit).
    Plain file reads: nothing is launched on the GPU and no tool is started,
    so the timed region is not disturbed.  The CORDIC kernels turn out to run
    at the power limit with the clock below its 2.4 GHz maximum (DESIGN.md
    4.7); this puts the evidence into the bench line itself."""

    def __init__(self, device, period=0.002):
        super().__init__(daemon=True)
        self.period = period
        self.dir = self._find(device)
        self.rows = []                  # (t, watts, sclk MHz)
        self._halt = threading.Event()

    @staticmethod
    def _bus_id(device):
        import ctypes
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
                return None
            return buf.value.decode().lower()
        except OSError:
            return None

    @classmethod
    def _find(cls, device):
        import glob
        cands = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
        cands = [c for c in cands
                 if os.path.exists(os.path.join(c, "power1_input"))
                 and os.path.exists(os.path.join(c, "freq1_input"))]
        bus = cls._bus_id(device)
        for c in cands:
            real = os.path.realpath(os.path.dirname(os.path.dirname(c))).lower()
            if bus and real.endswith(bus):
                return c
        return cands[0] if len(cands) == 1 else None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def run(self):
        while not self._halt.is_set():
            try:
                self.rows.append((time.perf_counter(),
                                  self._read("power1_input") / 1e6,
                                  self._read("freq1_input") / 1e6))
            except (OSError, ValueError):
                pass
            time.sleep(self.period)

    def stop(self):
        self._halt.set()
        self.join()

    def window(self, t0, t1):
        """Statistics of the samples taken in [t0, t1]."""
        w = [r for r in self.rows if t0 <= r[0] <= t1]
        if not w:
            return None
        pw = sorted(r[1] for r in w)
        ck = sorted(r[2] for r in w)
        return {"samples": len(w), "seconds": t1 - t0,
                "socket_w_median": pw[len(pw) // 2], "socket_w_max": pw[-1],
                "sclk_mhz_median": ck[len(ck) // 2], "sclk_mhz_min": ck[0]}

    def limit_w(self):
        try:
            return self._read("power1_cap") / 1e6
        except (OSError, ValueError):
            return None


def start_power(device, enabled):
    if not enabled:
        return None
    sp = PowerSampler(device)
    if sp.dir is None:
        return None
    sp.start()
    return sp


def finish_power(sampler, step, sync, t0, elapsed, steps, samples_per_step):
    """The timed region is short (the governor is still settling): keep the
    same kernel going for two more seconds and sample that as well."""
    if sampler is None:
        return None
    t1 = time.perf_counter()
    ms_step = elapsed / steps
    more = max(1, min(4000, int(2.0 / max(ms_step, 1e-6))))
    for _ in range(more):
        step()
    sync()
    t2 = time.perf_counter()
    sampler.stop()
    power = {"source": "amdgpu hwmon of the device (power1_input, "
                       "freq1_input), host thread, every 2 ms",
             "limit_w": sampler.limit_w(),
             "timed_region": sampler.window(t0, t1),
             "sustained": sampler.window(t1 + (t2 - t1) / 2, t2)}
    if power["sustained"]:
        # for information only: `value` is the K timed steps
        power["sustained"]["steps"] = more
        power["sustained"]["msamples_per_s_local_shards"] = (
            samples_per_step * more / (t2 - t1) / 1e6)
        # At the cap the clock is whatever the power budget allows: a kernel
        # that stalls less then runs at a lower clock, and what raises the
        # rate is less ENERGY per sample (fewer / cheaper instructions, fewer
        # LDS and HBM bytes), not fewer stalls.
        w = power["sustained"].get("socket_w_median")
        if w and power["limit_w"]:
            power["at_cap"] = bool(w >= 0.985 * power["limit_w"])
            if samples_per_step:
                power["nj_per_sample"] = (
                    w / (power["sustained"]["msamples_per_s_local_shards"] * 1e6) * 1e9)
    return power


def hbm_probe(in0, in1, out0, out1, nwords, r, w, mode, reps, stream=0):
    """tools/libhbmprobe.so: average ms per launch of an arithmetic-free
    kernel reading r and writing w arrays of nwords 32-bit words -- the same
    traffic as the CORDIC kernel, on the bench's own buffers (which it
    OVERWRITES).  None if the library is not built."""
    global _probe_lib
    import ctypes as C
    if _probe_lib is None:
        path = os.path.join(ROOT, "tools", "libhbmprobe.so")
        if not os.path.exists(path):
            _probe_lib = False
        else:
            _probe_lib = C.CDLL(path)
            _probe_lib.hbm_probe.restype = C.c_float
            _probe_lib.hbm_probe.argtypes = [C.c_void_p] * 4 + [
                C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    if not _probe_lib:
        return None
    ms = _probe_lib.hbm_probe(in0, in1, out0, out1, nwords, r, w, mode, reps,
                              stream)
    return float(ms) if ms > 0 else None


def copy_probe(ptrs, n, rw, reps=10):
    """The copy patterns over the arrays in `ptrs` = [in0, in1, out0, out1]:
    `tiles` = the best streaming pattern found on this chip (one-shot 4 KiB
    tiles), `queued` = the seeded kernel's own work distribution; `_nt` = the
    same with non-temporal loads and stores."""
    r, w = rw
    res = {}
    for name, mode in (("tiles", 0), ("queued", 1), ("tiles_nt", 2),
                       ("queued_nt", 3)):
        ms = hbm_probe(ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, r, w, mode, reps)
        if ms is not None:
            res[name + "_ms"] = ms
    return res


def bench_table(args, w, ca, dist, dev, world, rank):
    """Table cores (row F4): same timing discipline, gather kernel."""
    import oracle_lib as O
    quad = "quad" in w
    if quad:
        tab = ca.Quad(*w["quad"])
        oq = O.quad_cli(*w["quad"])
    else:
        kind, iw, ow, pw = w["table"]
        tab = ca.Table(kind, iw, ow, pw)
    n = 1 << args.log2_samples
    index0 = rank * n
    # one read + one written array, placed by measurement (cordic_arrays_alloc)
    arrays = ca.Arrays(4 * n, 1, 1)
    phase = arrays.tensor(0, torch.int32)
    out = arrays.tensor(1, torch.int32)
    ca.fill_phase_ramp(phase, index0, w["shift"])
    if args.input == "random":
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        phase.random_(-2**31, 2**31 - 1, generator=gen)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    sampler = start_power(dev.index or 0, rank == 0 and not args.no_power)
    for _ in range(args.warmup):
        tab.lookup(phase, out)
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(args.steps):
        tab.lookup(phase, out)
        ev[k + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    power = finish_power(sampler, lambda: tab.lookup(phase, out),
                         torch.cuda.synchronize, t0, elapsed, args.steps,
                         float(n))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device(dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    if rank == 0:
        idx = np.unique(np.concatenate([
            np.arange(0, min(n, 4096)), np.arange(max(0, n - 4096), n),
            np.arange(0, n, 65521)])).astype(np.int64)
        ti = torch.from_numpy(idx).to(dev)
        sel = phase[ti].cpu().numpy().view(np.uint32)
        if quad:
            exp = O.quad_lookup(oq, O.quad_tables(oq), sel)
        else:
            tv = O.table_values(kind, tab.pw, tab.ow)
            exp = O.table_lookup(kind, tab.pw, tab.ow, tv, sel)
        ok = bool(np.array_equal(out[ti].cpu().numpy(), exp))
        avg = float(np.mean(kern_ms)) / 1e3
        achieved = w["bytes"] * n / avg / 1e9
        line = {
            "metric": "Msamples/sec (%s)" % args.workload,
            "value": float(world) * n * args.steps / elapsed / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "build": build_stamp.stamp(),
            "config": {"workload": "%s: %s" % (args.workload, w["desc"]),
                       "samples_per_gpu": n, "pw": tab.pw, "ow": tab.ow,
                       "entries": tab.entries,
                       "kernel": "quad_lookup" if quad else
                       "table_lookup (lds mode %d)" % tab.lds_mode,
                       "input": args.input, "parallelism": "shard%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None,
                         "bytes_per_sample": w["bytes"],
                         "kernel_ms_avg": avg * 1e3},
            "from_profile": from_profile(args.workload),
            "bit_exact_vs_oracle": ok}
        if power is not None:
            line["roofline"]["power"] = power
        emit(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


RW = {"p2r": (1, 2), "nco": (0, 2), "r2p": (2, 2)}   # arrays read / written


def visible_gpus():
    """HIP devices this process can see, through the C ABI (no torch.cuda
    initialisation in a parent that is about to exec)."""
    import cordic_amd as ca
    n = ca.device_count()
    return n if n > 0 else 0


def resolve_launch(args):
    """How this invocation runs, and the checks that make `n_gpus` in the
    output impossible to disagree with --gpus:
      "torchrun"        started by torch.distributed.run: WORLD_SIZE == --gpus
      "single-process"  --single-process: one host process, --gpus devices,
                        the C++ cordic_group layer, no process group
      "spawn"           plain `python bench.py --gpus N` with N > 1 (or
                        --spawn): re-exec under torch.distributed.run with N
                        ranks, one GPU each
      "direct"          N = 1, this process"""
    need = args.gpus
    if need < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" in os.environ:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != need:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- start "
                             "one rank per GPU (--nproc-per-node %d)"
                             % (need, world, need))
        have = visible_gpus()
        if have < need and not SHARE_GPU:
            raise SystemExit("bench.py: --gpus %d needs %d visible GPUs, "
                             "found %d" % (need, need, have))
        return "torchrun"
    have = visible_gpus()
    if have < need and not SHARE_GPU:
        raise SystemExit("bench.py: --gpus %d needs %d visible GPUs, found %d"
                         % (need, need, have))
    if args.single_process:
        return "single-process"
    if need > 1 or args.spawn:
        return "spawn"
    return "direct"


def respawn(args):
    """Replace this process by `python -m torch.distributed.run` with --gpus
    ranks running this same command line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, BENCH_SELF_SPAWNED="1",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                   "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def spot_indices(n):
    """(offset, count) windows of a shard checked against the oracle: both
    ends and 61 windows spread through the middle."""
    win = min(n, 4096)
    offs = {0, n - win}
    for k in range(1, 62):
        offs.add(min(n - win, (k * (n // 62)) // 4 * 4))
    return sorted((o, win) for o in offs)


def run_group(args, w, launch):
    """p2r / nco / r2p workloads through the C++ multi-GPU layer of the C ABI
    (cordic_group_*): this process drives `nlocal` shards of `total`."""
    import cordic_amd as ca
    import oracle_lib as O
    from gpu_util import cpu_digest

    m, iw, ow, xtra, pw, ns = w["cli"]
    cfg = ca.Config.from_cli(MODE[m], iw, ow, xtra, pw, ns)
    if args.generic:
        cfg = cfg.with_flags(ca.FLAG_FORCE_GENERIC)
    if args.no_seed:
        cfg = cfg.with_flags(ca.FLAG_NO_SEED)
    if args.static_chunks:
        cfg = cfg.with_flags(ca.FLAG_STATIC_CHUNKS)
    if args.no_tails:
        cfg = cfg.with_flags(ca.FLAG_NO_TAILS)

    dist = None
    rank, world, local = 0, 1, 0
    if launch == "torchrun":
        import torch.distributed as dist
        rank = int(os.environ["RANK"])
        world = int(os.environ["WORLD_SIZE"])
        local = 0 if SHARE_GPU else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist_init(dist, rank, world, local)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if launch == "single-process":
        total, nlocal, first = args.gpus, args.gpus, 0
        devices = [0] * args.gpus if SHARE_GPU else list(range(args.gpus))
    else:
        total, nlocal, first, devices = world, 1, rank, [local]

    n = 1 << args.log2_samples
    n_total = n * total
    kind = w["kind"]
    x0, y0 = (1 << (iw - 1)) - 1, 0
    grp = ca.Group(cfg, devices=devices, first_shard=first, total_shards=total)
    seeded, seed_stages, tails = False, 0, []
    if kind in ("p2r", "nco") and not args.generic and not args.no_seed:
        probe_plan = ca.Plan(cfg)
        seed_stages = probe_plan.seed_info["stages"]
        seeded = seed_stages > 0
        # (what the plan carries; an NCO with a large increment and rows of
        # unrelated phases still run the recurrence: DESIGN.md section 4.3)
        tails = [] if args.no_tails else probe_plan.tail_groups
        probe_plan.close()

    def fill(g):
        if kind == "p2r":
            g.fill_phase_ramp(n_total, w["shift"])
        elif kind == "r2p":
            g.fill_iq_ramp(n_total, 0x9E3779B1, 0x85EBCA77, iw)
        else:
            g.reserve(n_total, 0)
        if args.input == "random" and kind != "nco":
            for sh in range(nlocal):
                with torch.cuda.device(devices[sh]):
                    gen = torch.Generator(device="cuda").manual_seed(
                        1234 + first + sh)
                    lo, hi = ((-2**31, 2**31 - 1) if kind == "p2r" else
                              (-2**(iw - 1), 2**(iw - 1) - 1))
                    for arr in range(1 if kind == "p2r" else 2):
                        t = torch.empty(n, dtype=torch.int32, device="cuda")
                        t.random_(lo, hi, generator=gen)
                        torch.cuda.synchronize()
                        g.write(sh, arr, 0, t)
        g.sync()

    def step(g):
        if kind == "p2r":
            g.p2r_const(n_total, x0, y0)
        elif kind == "r2p":
            g.r2p(n_total)
        else:
            g.nco(n_total, 0, 0x01234567, x0, y0)

    def barrier():
        grp.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fill(grp)
    step(grp)                   # allocates the outputs; first-launch costs
    grp.sync()
    if seeded and ca.last_kernel() != ca.KERNEL_SEEDED:
        # a batch below the size from which a plan takes the table-driven
        # kernel by itself (cordic_kernels.hip: seed_min_samples; --log2-samples
        # under 23 without CORDIC_SEED_MIN_SAMPLES=0): the line must name the
        # kernel that ran
        seeded, seed_stages, tails = False, 0, []

    # ---- same-run copy probes on the very arrays of shard 0 (before)
    _, ptrs, _ = grp.buffers(0)
    probes = []
    if not args.no_copy_probe:
        with torch.cuda.device(devices[0]):
            probes.append(copy_probe(ptrs, n, RW[kind]))

    sampler = start_power(devices[0], rank == 0 and not args.no_power)

    for _ in range(args.warmup):
        step(grp)
    barrier()

    # ---- timed region: exactly K steps; HIP events on the streams the
    # kernels are launched on (the shards' compute streams, inside the C ABI)
    every = max(1, -(-args.steps // 254))
    marks = [0]
    t0 = time.perf_counter()
    grp.mark(0)
    for k in range(args.steps):
        step(grp)
        if (k + 1) % every == 0 or k + 1 == args.steps:
            grp.mark(len(marks))
            marks.append(k + 1)
    barrier()
    elapsed = time.perf_counter() - t0
    power = finish_power(sampler, lambda: step(grp), grp.sync, t0, elapsed,
                         args.steps, float(nlocal) * n)
    per_rank = [elapsed]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device(dev))
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(v.item()) for v in allt]
        elapsed = max(per_rank)
    span_ms = [grp.elapsed(i, i + 1)[0] / (marks[i + 1] - marks[i])
               for i in range(len(marks) - 1)]
    kern_total_ms, per_shard_ms = grp.elapsed(0, len(marks) - 1)
    kern_avg_s = kern_total_ms / args.steps / 1e3

    # ---- after the timed region: correctness of what was just computed
    digest = local_digest = grp.digest(n_total)
    if dist is not None:
        d = torch.tensor([digest - (1 << 64) if digest >= 1 << 63 else digest],
                         dtype=torch.int64, device=coll_device(dev))
        dist.all_reduce(d, op=dist.ReduceOp.SUM)     # digests of shards add
        digest = int(d.item()) & 0xFFFFFFFFFFFFFFFF

    check = digest_check = None
    ocfg = O.config_cli(MODE[m], iw, ow, xtra, pw, ns)
    # EVERY output of EVERY rank against the oracle: each rank pushes its own
    # shards' samples through the scalar oracle (its share of the node's host
    # cores) and compares digests; the verdicts are reduced
    lo = grp.range(n_total, first)[0]
    cnt_all = sum(grp.range(n_total, first + sh)[1] for sh in range(nlocal))
    leg = oracle_digest_leg(args, w, ocfg, lo, cnt_all, x0, y0,
                            threads=max(1, _usable_cpus()
                                        // ranks_on_this_node(world)))
    legs = None
    if leg is not None:
        legs = reduce_digest_legs(dist, dev, world,
                                  local_digest == leg["digest"], leg)
    if rank == 0:
        start0 = grp.range(n_total, first)[0]

        def oracle(off, cnt):
            if kind == "r2p":
                xi = grp.read(0, 0, off, cnt)
                yi = grp.read(0, 1, off, cnt)
                ra, rb = O.topolar(ocfg, xi, yi)
                return ra, rb.view(np.int32)
            if kind == "p2r":
                ph = grp.read(0, 0, off, cnt).view(np.uint32)
            else:
                idx = np.arange(cnt, dtype=np.uint64) + np.uint64(start0 + off)
                ph = ((idx * np.uint64(0x01234567))
                      & np.uint64(0xffffffff)).astype(np.uint32)
            return O.rotate(ocfg, x0, y0, ph)
        check = True
        for off, cnt in spot_indices(n):
            ra, rb = oracle(off, cnt)
            check = check and bool(
                np.array_equal(grp.read(0, 2, off, cnt), ra)
                and np.array_equal(grp.read(0, 3, off, cnt), rb))
        # the device digest kernel against the oracle's outputs over the same
        # leading 2^20 samples of shard 0
        cnt = min(n, 1 << 20)
        ra, rb = oracle(0, cnt)
        want = (cpu_digest(ra, start0)
                + cpu_digest(rb, start0 + (1 << 40))) % (1 << 64)
        with torch.cuda.device(devices[0]):
            dd = torch.zeros(1, dtype=torch.int64, device="cuda")
            ca.digest_u32(ptrs[2], start0, dd, n=cnt)
            ca.digest_u32(ptrs[3], start0 + (1 << 40), dd, n=cnt)
            torch.cuda.synchronize()
            got = int(dd.cpu().numpy().view(np.uint64)[0])
        digest_check = {"samples": cnt, "device": "%016x" % got,
                        "oracle": "%016x" % want, "equal": got == want}
        if legs is not None:
            all_ok, samples_all, oracle_sum, slowest = legs
            digest_check = {
                "samples": samples_all, "device": "%016x" % digest,
                "oracle": "%016x" % oracle_sum,
                "equal": all_ok and digest == oracle_sum,
                "oracle_seconds": slowest, "oracle_threads": leg["cores"],
                "ranks": world,
                "what": "position-aware 64-bit digest of ALL outputs of ALL "
                        "ranks: each rank's device digest over what its timed "
                        "kernels wrote vs oracle/cordic_oracle.c: orc_digest of "
                        "its own shards (every sample through the scalar "
                        "oracle); verdicts AND digests reduced over the ranks",
                "leading_2^20_also_equal": got == want}
            check = check and digest_check["equal"]

    # ---- constant-vector feeds: also time the full-recurrence kernel (every
    # sample runs all micro-rotations) so both numbers are on record
    full = None
    if seeded:
        grp2 = ca.Group(cfg.with_flags(ca.FLAG_NO_SEED), devices=devices,
                        first_shard=first, total_shards=total)
        for sh in range(nlocal):        # same inputs, bit for bit
            grp2.reserve(n_total, 1 if kind == "p2r" else 0)
            if kind == "p2r":
                _, p1, _ = grp.buffers(sh)
                grp2.write(sh, 0, 0, _RawWords(p1[0], n))
        k2 = max(3, min(args.steps, 10))
        step(grp2)
        grp2.sync()
        grp2.mark(0)
        for _ in range(k2):
            step(grp2)
        grp2.mark(1)
        ms2 = grp2.elapsed(0, 1)[0] / k2
        d2 = grp2.digest(n_total)
        if dist is not None:
            d = torch.tensor([d2 - (1 << 64) if d2 >= 1 << 63 else d2],
                             dtype=torch.int64, device=coll_device(dev))
            dist.all_reduce(d, op=dist.ReduceOp.SUM)
            d2 = int(d.item()) & 0xFFFFFFFFFFFFFFFF
        full = {"ms_per_step": ms2, "steps": k2,
                "value_per_gpu": n / ms2 / 1e3,
                "hbm_frac": w["bytes"] * n / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "outputs_identical_to_seeded_kernel": d2 == digest,
                "compared_by": "64-bit position-aware digest of all outputs",
                "from_profile": from_profile(args.workload + "_noseed")}
        grp2.close()

    gather = None
    if args.gather:
        gather = time_gather(args, grp, step, launch, dist, dev, devices, n,
                             n_total, rank, world, barrier)

    # ---- same-run copy probes again (after): the memory system may have
    # changed state under sustained load
    if not args.no_copy_probe:
        with torch.cuda.device(devices[0]):
            probes.append(copy_probe(ptrs, n, RW[kind]))

    grp_placement = grp.placement(0)
    grp.close()
    single = None
    if (launch == "torchrun" and not args.no_single_process_check
            and (world > 1 or os.environ.get("BENCH_FORCE_SINGLE_CHECK"))):
        # the C++ one-process layer on the same GPUs, for the record: rank 0
        # drives every device while the other ranks wait on the HOST (a gloo
        # barrier: no GPU kernel spins meanwhile)
        host = dist.new_group(backend="gloo")
        torch.cuda.empty_cache()
        if rank == 0:
            try:
                single = single_process_block(args, world, digest)
            except Exception as e:            # never lose the main line
                single = {"error": repr(e)}
        dist.barrier(group=host)

    if rank == 0:
        value = float(total) * n * args.steps / elapsed / 1e6
        achieved = w["bytes"] * n / kern_avg_s / 1e9
        roof = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "bytes_per_sample": w["bytes"],
            "kernel_ms_avg": kern_avg_s * 1e3,
            "kernel_ms_min": float(min(span_ms)),
            "kernel_ms_max": float(max(span_ms)),
        }
        if power is not None:
            roof["power"] = power
        # which of its candidate allocations the group gave which role
        # (cordic_group placement: include/cordic_amd.h)
        roof["placement"] = dict(
            grp_placement,
            what="arrays allocated +2 spare, arithmetic-free probes of the "
                 "job's traffic over the role assignments, best kept "
                 "(--no-placement: as hipMalloc hands them out)")
        pm = None
        if not args.no_pmc and total == 1:
            try:
                pm = measure_pmc(args)
            except Exception as e:            # never lose the main line
                pm = {"error": repr(e)}
            roof["pmc"] = pm
            if "hbm_bytes_per_launch" in pm:
                roof["traffic"] = pm["hbm_bytes_per_launch"] * (
                    n / float(1 << args.log2_samples))
                roof["traffic_over_algorithmic"] = roof["traffic"] / (
                    w["bytes"] * n)
        add_valu(roof, n / kern_avg_s, pm, power, from_profile(
            args.workload + ("_noseed" if args.no_seed else "")))
        if probes and probes[0]:
            # the plain-copy ceiling of THIS run on THESE arrays: best of the
            # probes before and after the timed region
            best = {}
            for pr in probes:
                for k, v in pr.items():
                    best[k] = min(v, best.get(k, v))
            roof["copy"] = {
                "what": "arithmetic-free kernels with this kernel's traffic "
                        "(%dR%dW x 4 B/sample) on the same arrays, 10 "
                        "launches each before and after the timed region: "
                        "one-shot 4 KiB tiles and the seeded kernel's tile "
                        "queue, plain and non-temporal accesses "
                        "(tools/hbm_probe_lib.hip); copy_frac = the fastest"
                        % RW[kind],
                "before": probes[0], "after": probes[-1]}
            # the fastest arithmetic-free copy of this traffic seen in this
            # run, whatever its distribution and cache policy
            allp = [best[k] for k in ("tiles_ms", "tiles_nt_ms", "queued_ms",
                                      "queued_nt_ms") if k in best]
            if allp:
                cf = w["bytes"] * n / (min(allp) * 1e-3) / 1e9 / HBM_PEAK_GBS
                roof["copy_frac"] = cf
                roof["frac_over_copy"] = roof["frac"] / cf
            queued = [best[k] for k in ("queued_ms", "queued_nt_ms") if k in best]
            if queued:
                roof["copy_frac_same_distribution"] = (
                    w["bytes"] * n / (min(queued) * 1e-3) / 1e9 / HBM_PEAK_GBS)
        out = {
            "metric": "Msamples/sec (sin+cos pairs) at 16-stage/32-bit"
                      if args.workload == "cfg2" else
                      "Msamples/sec (%s)" % args.workload,
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": total,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64" if cfg.ww > 32 else "int32",
            "data": "synthetic",
            "build": build_stamp.stamp(),
            "config": {
                "workload": "%s: %s" % (args.workload, w["desc"]),
                "samples_per_gpu": n,
                "iw": cfg.iw, "ow": cfg.ow, "ww": cfg.ww, "pw": cfg.pw,
                "nstages": cfg.nstages, "rotations": cfg.nlive,
                "kernel": "generic" if args.generic else (
                    "seeded(%d)%s+unrolled, %s" % (
                        seed_stages,
                        "+tails(%s)" % "+".join(map(str, tails)) if tails else "",
                        "static chunks" if args.static_chunks
                        else "address-ordered tile queue")
                    if seeded else ("topolar_lj / topolar_unrolled"
                                    if kind == "r2p" else "unrolled")),
                "input": args.input,
                "parallelism": "shard%d" % total,
            },
            "launch": {
                "mode": ("TEST: %d ranks sharing device 0, gloo process group; "
                         % world if SHARE_GPU else "") +
                        {"torchrun": "one process per GPU (torch.distributed"
                         ".run%s), RCCL only for the digest all-reduce" % (
                             ", self-spawned by bench.py" if os.environ.get(
                                 "BENCH_SELF_SPAWNED") else ""),
                         "single-process": "one host process, %d devices, C++ "
                         "cordic_group layer, no process group" % total,
                         "direct": "one process, one device"}[launch],
                "world_size": world,
                "shards_per_process": nlocal,
                "per_rank_Msamples_per_s": [
                    nlocal * n * args.steps / t / 1e6 for t in per_rank],
                "per_shard_kernel_ms": [v / args.steps for v in per_shard_ms],
            },
            "roofline": roof,
            "from_profile": from_profile(
                args.workload + ("_noseed" if args.no_seed else "")),
            "bit_exact_vs_oracle": check,
            "digest": "%016x" % digest,
            "digest_check": digest_check,
        }
        if full is not None:
            out["full_recurrence_kernel"] = full
        if gather is not None:
            out["gather"] = gather
        if single is not None:
            out["single_process_cordic_group"] = single
        if not args.no_cpu_baseline and total == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, leg=leg)
        if (total == 1 and args.workload == "cfg2" and not args.no_other_paths):
            torch.cuda.empty_cache()
            out["other_paths"] = other_paths(args)
            try:
                out["other_paths"]["host_arrays"] = host_paths()
            except Exception as e:            # never lose the main line
                out["other_paths"]["host_arrays"] = {"error": repr(e)}
        emit(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


class _RawWords:
    """A device address + word count, shaped like what Group.write accepts."""

    def __init__(self, ptr, n):
        self._p, self._n = ptr, n

    def data_ptr(self):
        return self._p

    def numel(self):
        return self._n


def single_process_block(args, ndev, expect):
    """The C++ one-process layer (cordic_group over every GPU, no process
    group) measured on the same node: a SEPARATE `bench.py --single-process`
    process with a time limit, so that nothing it does can take the main
    result down with it.  Returns a digest of its line."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE",
                        "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "BENCH_SELF_SPAWNED")}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(ndev),
           "--single-process", "--workload", args.workload, "--steps",
           str(args.steps), "--warmup", str(args.warmup), "--log2-samples",
           str(args.log2_samples), "--input", args.input, "--gather",
           "--no-cpu-baseline", "--no-other-paths", "--no-copy-probe",
           "--no-pmc", "--no-power"]
    for flag, on in (("--no-seed", args.no_seed), ("--generic", args.generic),
                     ("--static-chunks", args.static_chunks)):
        if on:
            cmd.append(flag)
    r = subprocess.run(cmd, env=env, text=True, capture_output=True,
                       timeout=300)
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not rows:
        return {"error": "rc %d: %s" % (r.returncode, r.stderr[-400:])}
    d = json.loads(rows[-1])
    res = {"n_gpus": d["n_gpus"], "value": d["value"], "unit": d["unit"],
           "ms_per_step": d["ms_per_step"], "steps": d["steps"],
           "mode": d["launch"]["mode"],
           "per_shard_kernel_ms": d["launch"]["per_shard_kernel_ms"],
           "bit_exact_vs_oracle": d["bit_exact_vs_oracle"],
           "digest_equals_multi_process_run":
               int(d["digest"], 16) == expect if args.input == "ramp" else None}
    if "gather" in d:
        res["gather"] = d["gather"]
    return res


def time_gather(args, grp, step, launch, dist, dev, devices, n, n_total, rank,
                world, barrier):
    """Collecting the outputs on one GPU, timed separately (never part of
    `value`): C++ peer copies in the single-process layout, RCCL gather in the
    process-per-GPU layout."""
    import cordic_amd as ca
    if launch == "single-process":
        root = ca.Group(grp.cfg, devices=[devices[0]], first_shard=0,
                        total_shards=1)
        root.reserve(n_total, 0)
        _, rp, _ = root.buffers(0)
        grp.set_gather(devices[0], rp[2], rp[3], 8)
        step(grp)
        grp.sync()
        k = max(3, min(args.steps, 10))
        t1 = time.perf_counter()
        for _ in range(k):
            step(grp)
        grp.sync()
        ms = (time.perf_counter() - t1) / k * 1e3
        ok = root.digest(n_total) == grp.digest(n_total)
        grp.set_gather(-1)
        root.close()
        return {"mode": "hipMemcpyPeerAsync, 8 pieces per shard behind the "
                "compute (cordic_group_set_gather)",
                "ms_compute_and_gather": ms, "outputs_identical": ok}
    if dist is None:
        return None
    # process-per-GPU: the C++ layer's own RCCL forwarding (ncclSend/ncclRecv
    # piece by piece behind the compute); torch.distributed only carries the
    # 128-byte RCCL id and the barrier
    uid = [ca.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    grp.rccl_init(uid[0])
    root = rp = None
    if rank == 0:
        root = ca.Group(grp.cfg, devices=[devices[0]], first_shard=0,
                        total_shards=1)
        root.reserve(n_total, 0)
        _, rp, _ = root.buffers(0)
    grp.set_gather_rccl(0, rp[2] if rp else None, rp[3] if rp else None, 8)
    step(grp)
    barrier()
    k = max(3, min(args.steps, 10))
    t1 = time.perf_counter()
    for _ in range(k):
        step(grp)
    barrier()
    ms = (time.perf_counter() - t1) / k * 1e3
    grp.set_gather_rccl(-1)
    d = grp.digest(n_total)
    t = torch.tensor([d - (1 << 64) if d >= 1 << 63 else d],
                     dtype=torch.int64, device=coll_device(dev))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    ok = None
    if rank == 0:
        ok = root.digest(n_total) == (int(t.item()) & 0xFFFFFFFFFFFFFFFF)
        root.close()
    return {"mode": "RCCL ncclSend/ncclRecv to shard 0, 8 pieces per shard "
            "behind the compute (cordic_group_set_gather_rccl)",
            "ms_compute_and_gather": ms, "outputs_identical": ok}


def run_direct(args, w, launch):
    """Workloads outside the cordic_group layer -- 16-bit sample containers
    (cfg1), per-sample x/y vectors (p2rxy) and the table cores -- on torch
    tensors through the stateless entry points; one process per GPU."""
    import cordic_amd as ca

    world = int(os.environ.get("WORLD_SIZE", "1")) if launch == "torchrun" else 1
    rank = int(os.environ.get("RANK", "0")) if launch == "torchrun" else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if launch == "torchrun" else 0
    if SHARE_GPU:
        local = 0
    dist = None
    if launch == "torchrun":
        # launched by torch.distributed.run: RCCL process group (also for a
        # single rank, so that the collective path can be exercised on 1 GPU)
        import torch.distributed as dist
        dist_init(dist, rank, world, local)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    w = WORKLOADS[args.workload]
    if w["kind"] == "tbl":
        return bench_table(args, w, ca, dist, dev, world, rank)
    m, iw, ow, xtra, pw, ns = w["cli"]
    cfg = ca.Config.from_cli(MODE[m], iw, ow, xtra, pw, ns)
    if args.generic:
        cfg = cfg.with_flags(ca.FLAG_FORCE_GENERIC)
    if args.no_seed:
        cfg = cfg.with_flags(ca.FLAG_NO_SEED)
    if args.static_chunks:
        cfg = cfg.with_flags(ca.FLAG_STATIC_CHUNKS)
    n = 1 << args.log2_samples
    index0 = rank * n                   # shard by global sample index
    x0, y0 = (1 << (iw - 1)) - 1, 0

    # ---- resident inputs / outputs
    io16 = bool(w.get("io16"))
    sdt = torch.int16 if io16 else torch.int32
    # the arrays of the job, placed by measurement (cordic_arrays_alloc: up to
    # two read + two written arrays; a third input is taken as it comes)
    nread = {"p2r": 1, "p2rxy": 2, "r2p": 2}[w["kind"]]
    arrays = ca.Arrays((2 if io16 else 4) * n, nread, 2)
    a = arrays.tensor(nread, sdt)
    b = arrays.tensor(nread + 1, sdt)
    if w["kind"] == "p2r":
        p32 = torch.empty(n, dtype=torch.int32, device=dev)
        ca.fill_phase_ramp(p32, index0, w["shift"])
        if args.input == "random":
            gen = torch.Generator(device=dev).manual_seed(1234 + rank)
            p32.random_(-2**31, 2**31 - 1, generator=gen)
        phase = arrays.tensor(0, sdt)
        phase.copy_(p32.to(sdt))            # io16: the low 16 bits, n mod 2^16
        del p32
        torch.cuda.empty_cache()

        plan = ca.Plan(cfg)

        def step():
            plan.p2r_const(x0, y0, phase, a, b)
    elif w["kind"] == "p2rxy":
        phase = arrays.tensor(0, torch.int32)
        xin = arrays.tensor(1, torch.int32)
        yin = torch.empty(n, dtype=torch.int32, device=dev)
        ca.fill_phase_ramp(phase, index0, w["shift"])
        ca.fill_iq_ramp(xin, yin, index0, 0x9E3779B1, 0x85EBCA77, iw)
        if args.input == "random":
            gen = torch.Generator(device=dev).manual_seed(1234 + rank)
            phase.random_(-2**31, 2**31 - 1, generator=gen)

        # through a plan: the stage directions are looked up (cordic_xydir.h);
        # --no-tails (CORDIC_FLAG_NO_TAILS) keeps cordic_p2r's kernel for A/B
        plan = ca.Plan(cfg.with_flags(ca.FLAG_NO_TAILS) if args.no_tails else cfg)

        def step():
            plan.p2r(xin, yin, phase, a, b)
    elif w["kind"] == "r2p":
        xin = arrays.tensor(0, torch.int32)
        yin = arrays.tensor(1, torch.int32)
        ca.fill_iq_ramp(xin, yin, index0, 0x9E3779B1, 0x85EBCA77, iw)
        if args.input == "random":
            gen = torch.Generator(device=dev).manual_seed(1234 + rank)
            xin.random_(-2**(iw - 1), 2**(iw - 1) - 1, generator=gen)
            yin.random_(-2**(iw - 1), 2**(iw - 1) - 1, generator=gen)

        def step():
            ca.r2p(cfg, xin, yin, a, b)
    else:
        plan = ca.Plan(cfg)

        def step():
            plan.nco(n, 0, 0x01234567, index0, x0, y0, a, b)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sampler = start_power(local, rank == 0 and not args.no_power)
    for _ in range(max(1, args.warmup)):
        step()
    barrier()
    ran = ca.last_kernel()      # the family that really serves this batch size

    # ---- timed region: exactly K steps; HIP events (on the stream the
    # kernels are launched on: torch's current stream) bracket every launch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(args.steps):
        step()
        ev[k + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    power = finish_power(sampler, step, torch.cuda.synchronize, t0, elapsed,
                         args.steps, float(n))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device(dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    kern_avg_s = float(np.mean(kern_ms)) / 1e3

    # ---- after the timed region: correctness of what was just computed
    d = torch.zeros(1, dtype=torch.int64, device=dev)
    ca.digest_u32(a.view(torch.int32), index0 // (2 if io16 else 1), d)
    ca.digest_u32(b.view(torch.int32),
                  index0 // (2 if io16 else 1) + (1 << 40), d)
    torch.cuda.synchronize()
    local_digest = int(d.cpu().numpy().view(np.uint64)[0])
    if dist is not None:
        d = d.to(coll_device(dev))
        dist.all_reduce(d, op=dist.ReduceOp.SUM)     # digests of shards add
    torch.cuda.synchronize()
    digest = int(d.cpu().numpy().view(np.uint64)[0])

    check = digest_check = None
    import oracle_lib as O
    ocfg = O.config_cli(MODE[m], iw, ow, xtra, pw, ns)
    # EVERY output of EVERY rank against the oracle (see run_group)
    leg = oracle_digest_leg(args, w, ocfg, index0, n, x0, y0,
                            threads=max(1, _usable_cpus()
                                        // ranks_on_this_node(world)))
    legs = None
    if leg is not None:
        legs = reduce_digest_legs(dist, dev, world,
                                  local_digest == leg["digest"], leg)
    if rank == 0:
        idx = np.unique(np.concatenate([
            np.arange(0, min(n, 4096)), np.arange(max(0, n - 4096), n),
            np.arange(0, n, 65521)])).astype(np.int64)
        ti = torch.from_numpy(idx).to(dev)
        ga, gb = a[ti].cpu().numpy(), b[ti].cpu().numpy()
        if w["kind"] == "r2p":
            ra, rb = O.topolar(ocfg, xin[ti].cpu().numpy(),
                               yin[ti].cpu().numpy())
            rb = rb.view(np.int32)
        elif w["kind"] == "p2rxy":
            ra, rb = O.rotate(ocfg, xin[ti].cpu().numpy(),
                              yin[ti].cpu().numpy(),
                              phase[ti].cpu().numpy().view(np.uint32))
        elif w["kind"] == "p2r" and io16:
            ra, rb = O.rotate(ocfg, x0, y0, phase[ti].cpu().numpy()
                              .view(np.uint16).astype(np.uint32))
            ra, rb = ra.astype(np.int16), rb.astype(np.int16)
        elif w["kind"] == "p2r":
            ra, rb = O.rotate(ocfg, x0, y0,
                              phase[ti].cpu().numpy().view(np.uint32))
        else:
            ph = ((idx.astype(np.uint64) + np.uint64(index0))
                  * np.uint64(0x01234567) & np.uint64(0xffffffff))
            ra, rb = O.rotate(ocfg, x0, y0, ph.astype(np.uint32))
        check = bool(np.array_equal(ga, ra) and np.array_equal(gb, rb))
        if legs is not None:
            all_ok, samples_all, oracle_sum, slowest = legs
            digest_check = {
                "samples": samples_all, "device": "%016x" % digest,
                "oracle": "%016x" % oracle_sum,
                "equal": all_ok and digest == oracle_sum,
                "oracle_seconds": slowest, "oracle_threads": leg["cores"],
                "ranks": world,
                "what": "position-aware 64-bit digest of ALL outputs of ALL "
                        "ranks: device digest kernel vs oracle/cordic_oracle.c: "
                        "orc_digest, rank by rank, verdicts and digests reduced"}
            check = check and digest_check["equal"]

    # ---- constant-vector feeds: also time the full-recurrence kernel (every
    # sample runs all micro-rotations) so both numbers are on record
    full = None
    if (w["kind"] in ("p2r", "nco") and not args.no_seed and not args.generic
            and plan.seed_info["stages"] > 0 and ran == ca.KERNEL_SEEDED):
        plan2 = ca.Plan(cfg.with_flags(ca.FLAG_NO_SEED))
        a2 = torch.empty_like(a)
        b2 = torch.empty_like(b)

        def step2():
            if w["kind"] == "p2r":
                plan2.p2r_const(x0, y0, phase, a2, b2)
            else:
                plan2.nco(n, 0, 0x01234567, index0, x0, y0, a2, b2)
        k2 = max(3, min(args.steps, 10))
        step2()
        barrier()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k2):
            step2()
        e1.record()
        barrier()
        ms2 = e0.elapsed_time(e1) / k2
        same = bool(torch.equal(a, a2) and torch.equal(b, b2))
        full = {"ms_per_step": ms2, "steps": k2,
                "value_per_gpu": n / ms2 / 1e3,
                "hbm_frac": w["bytes"] * n / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "outputs_identical_to_seeded_kernel": same}
        del a2, b2

    if rank == 0:
        total = float(world) * n * args.steps
        value = total / elapsed / 1e6
        achieved = w["bytes"] * n / kern_avg_s / 1e9
        out = {
            "metric": "Msamples/sec (sin+cos pairs) at 16-stage/32-bit"
                      if args.workload == "cfg2" else
                      "Msamples/sec (%s)" % args.workload,
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64" if cfg.ww > 32 else "int32",
            "data": "synthetic",
            "build": build_stamp.stamp(),
            "config": {
                "workload": "%s: %s" % (args.workload, w["desc"]),
                "samples_per_gpu": n,
                "iw": cfg.iw, "ow": cfg.ow, "ww": cfg.ww, "pw": cfg.pw,
                "nstages": cfg.nstages, "rotations": cfg.nlive,
                "kernel": "generic" if args.generic else (
                    ("directions(%s)" % "+".join(map(str, plan.dir_groups))
                     if ran == ca.KERNEL_DIRECTIONS else "unrolled")
                    if w["kind"] == "p2rxy" else
                    "unrolled" if (args.no_seed or w["kind"] == "r2p"
                                   or ran != ca.KERNEL_SEEDED)
                    else "seeded(%d)+unrolled" % plan.seed_info["stages"]),
                "input": args.input,
                "parallelism": "shard%d" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "bytes_per_sample": w["bytes"],
                "kernel_ms_avg": kern_avg_s * 1e3,
                "kernel_ms_min": float(np.min(kern_ms)),
            },
            "from_profile": from_profile(
                args.workload + ("_noseed" if args.no_seed else "")),
            "bit_exact_vs_oracle": check,
            "digest": "%016x" % digest,
            "digest_check": digest_check,
        }
        roof = out["roofline"]
        if power is not None:
            roof["power"] = power
        pm = None
        if not args.no_pmc and world == 1:
            try:
                pm = measure_pmc(args)
            except Exception as e:            # never lose the main line
                pm = {"error": repr(e)}
            roof["pmc"] = pm
            if "hbm_bytes_per_launch" in pm:
                roof["traffic"] = pm["hbm_bytes_per_launch"]
                roof["traffic_over_algorithmic"] = roof["traffic"] / (
                    w["bytes"] * n)
        add_valu(roof, n / kern_avg_s, pm, power, out["from_profile"])
        if full is not None:
            out["full_recurrence_kernel"] = full
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, leg=leg)
        emit(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()




def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--single-process", action="store_true",
                    help="one host process drives all --gpus devices through "
                    "the C++ cordic_group layer (no torch.distributed)")
    ap.add_argument("--spawn", action="store_true",
                    help="go through the self-launch path (re-exec under "
                    "torch.distributed.run) even for --gpus 1")
    ap.add_argument("--no-other-paths", action="store_true",
                    help="skip the informational rates of the other entry "
                    "points after the default (cfg2) run")
    ap.add_argument("--host-paths-only", action="store_true",
                    help="print only the host-array entry points' rates "
                    "(other_paths.host_arrays of the default line)")
    ap.add_argument("--log2-samples", type=int, default=30,
                    help="samples per GPU = 2^this")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-digest", action="store_true",
                    help="skip the oracle digest over ALL samples (rank 0, "
                    "all host cores, ~2 s per 2^30 samples on 16 cores)")
    ap.add_argument("--no-copy-probe", action="store_true")
    ap.add_argument("--no-placement", action="store_true",
                    help="take the group's arrays as hipMalloc hands them out "
                         "instead of probing candidate allocations")
    ap.add_argument("--no-power", action="store_true",
                    help="skip the hwmon power / clock samples and the two "
                         "seconds of sustained running behind the timed region")
    ap.add_argument("--pmc-counters",
                    default="FETCH_SIZE,WRITE_SIZE,SQ_INSTS_VALU",
                    help="comma-separated rocprofv3 counters, one pass each")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip measuring roofline.traffic (two rocprofv3 --pmc "
                    "passes, FETCH_SIZE and WRITE_SIZE, over a 3-step run of "
                    "this workload; 1-GPU runs only, ~20 s)")
    ap.add_argument("--no-single-process-check", action="store_true",
                    help="multi-process runs: skip the extra one-process "
                    "cordic_group measurement on rank 0")
    ap.add_argument("--gather", action="store_true",
                    help="also time collecting the outputs on one GPU")
    ap.add_argument("--input", default="ramp", choices=["ramp", "random"],
                    help="ramp = BASELINE.json's deterministic inputs; random "
                    "= uniformly random words (worst-case switching activity: "
                    "the chip clocks lower, MI355X_MICROARCH.md DVFS)")
    ap.add_argument("--no-seed", action="store_true",
                    help="constant-vector feeds: full 16-stage recurrence per "
                    "sample instead of the table-seeded kernel")
    ap.add_argument("--static-chunks", action="store_true",
                    help="seeded kernel: one contiguous chunk per persistent "
                    "block instead of the address-ordered tile queue (A/B)")
    ap.add_argument("--no-tails", action="store_true",
                    help="seeded kernel: phase recurrence behind the seeds "
                    "instead of the direction-tail lookups (A/B)")
    ap.add_argument("--generic", action="store_true",
                    help="force the generic (not unrolled) kernel")
    ap.add_argument("--nstages", type=int, default=0,
                    help="experiments: the workload's core with this many stages "
                    "(gencordic -n); the line's config says so")
    ap.add_argument("--ramp-shift", type=int, default=-1,
                    help="experiments: phase ramp n << this (steeper ramps)")
    args = ap.parse_args()
    if args.ramp_shift >= 0:
        w0 = WORKLOADS[args.workload]
        w0["shift"] = args.ramp_shift
        w0["desc"] += " [--ramp-shift %d]" % args.ramp_shift
    if args.nstages:
        w0 = WORKLOADS[args.workload]
        w0["cli"] = tuple(w0["cli"][:5]) + (args.nstages,)
        w0["desc"] += " [--nstages %d]" % args.nstages
    if args.no_placement:
        # read by cordic_group_create / cordic_arrays_alloc (and inherited by
        # the ranks and sub-runs this process starts)
        os.environ["CORDIC_GROUP_PLACEMENT"] = "0"

    if args.host_paths_only:
        print(json.dumps(host_paths()))
        return
    launch = resolve_launch(args)
    if launch == "spawn":
        respawn(args)               # does not return
    claim_stdout()
    w = WORKLOADS[args.workload]
    if w["kind"] in RW and not w.get("io16"):
        return run_group(args, w, launch)
    if launch == "single-process":
        raise SystemExit("bench.py: --single-process covers the p2r / nco / "
                         "r2p workloads on 32-bit containers")
    return run_direct(args, w, launch)


if __name__ == "__main__":
    main()
