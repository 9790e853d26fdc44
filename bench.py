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
no data-path collective (weak scaling); a digest all-reduce after the timed
region checks the shards, `--gather` additionally times collecting the
outputs on rank 0 over RCCL.

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

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TOPS = 78.6          # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz

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


def cpu_baseline(workload, seconds=12.0):
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
    t0 = time.perf_counter()
    total = L.orc_throughput(C.byref(ocfg), kind, cores, seconds, mul, x0, 0)
    wall = time.perf_counter() - t0
    return {
        "value": total / wall / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d samples of the %s workload (%d threads x 2^16-sample "
                  "blocks for %.0f s) through oracle/liboracle.so: gcc -O2 "
                  "scalar restatement of the reference RTL -- the reference "
                  "itself has no CPU compute path" % (total, workload, cores,
                                                      seconds),
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


def other_paths(ca, dev, log2n=29, steps=8):
    """Rates of the other entry points of the engine on this GPU (single-GPU
    default run only; informational, not `value`): 2^log2n samples, HIP events
    over `steps` launches after four warm-up launches, results spot-checked against the
    oracle.  Keys are bench.py workload names."""
    import oracle_lib as O
    n = 1 << log2n
    res = {}

    def timed(fn):
        for _ in range(4):      # the GPU idled during cpu_baseline: re-clock
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    idx = torch.arange(0, n, 65521, device=dev)

    def i32(k=1):
        return [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(k)]

    # cfg3: r2p 20 stages on I/Q ramps
    cfg = ca.Config.from_cli(1, 24, 24, 2, -1, 20)
    xin, yin, a, b = i32(4)
    ca.fill_iq_ramp(xin, yin, 0, 0x9E3779B1, 0x85EBCA77, 24)
    ms = timed(lambda: ca.r2p(cfg, xin, yin, a, b))
    rm, rp = O.topolar(O.config_cli(1, 24, 24, 2, -1, 20),
                       xin[idx].cpu().numpy(), yin[idx].cpu().numpy())
    ok = bool(np.array_equal(a[idx].cpu().numpy(), rm) and np.array_equal(
        b[idx].cpu().numpy().view(np.uint32), rp))
    res["cfg3"] = {"Msamples_per_s": n / ms / 1e3, "bytes_per_sample": 16,
                   "bit_exact_vs_oracle": ok}
    # p2rxy: per-sample x, y and phase
    cfg = ca.Config.from_cli(0, 32, 32, 2, 32, 16)
    ocfg = O.config_cli(0, 32, 32, 2, 32, 16)
    ph = i32()[0]
    ca.fill_phase_ramp(ph, 0, 2)
    ca.fill_iq_ramp(xin, yin, 0, 0x9E3779B1, 0x85EBCA77, 32)
    ms = timed(lambda: ca.p2r(cfg, xin, yin, ph, a, b))
    rx, ry = O.rotate(ocfg, xin[idx].cpu().numpy(), yin[idx].cpu().numpy(),
                      ph[idx].cpu().numpy().view(np.uint32))
    ok = bool(np.array_equal(a[idx].cpu().numpy(), rx)
              and np.array_equal(b[idx].cpu().numpy(), ry))
    res["p2rxy"] = {"Msamples_per_s": n / ms / 1e3, "bytes_per_sample": 20,
                    "bit_exact_vs_oracle": ok}
    # cfg5: fused NCO, store only
    plan = ca.Plan(cfg)
    ms = timed(lambda: plan.nco(n, 0, 0x01234567, 0, 2**31 - 1, 0, a, b))
    pn = ((idx.cpu().numpy().astype(np.uint64) * np.uint64(0x01234567))
          & np.uint64(0xffffffff)).astype(np.uint32)
    rx, ry = O.rotate(ocfg, 2**31 - 1, 0, pn)
    ok = bool(np.array_equal(a[idx].cpu().numpy(), rx)
              and np.array_equal(b[idx].cpu().numpy(), ry))
    res["cfg5"] = {"Msamples_per_s": n / ms / 1e3, "bytes_per_sample": 8,
                   "bit_exact_vs_oracle": ok}
    # quadtbl: the checked-in quadratic-interpolation core
    quad = ca.Quad(-1, 13, 2, 18)
    ms = timed(lambda: quad.lookup(ph, a))
    oq = O.quad_cli(-1, 13, 2, 18)
    ok = bool(np.array_equal(a[idx].cpu().numpy(), O.quad_lookup(
        oq, O.quad_tables(oq), ph[idx].cpu().numpy().view(np.uint32))))
    res["quadtbl"] = {"Msamples_per_s": n / ms / 1e3, "bytes_per_sample": 8,
                      "bit_exact_vs_oracle": ok}
    return res


def _pmc_valu(key, samples_per_launch, samples_per_s):
    """VALU instructions per sample (rocprofv3 SQ_INSTS_VALU of
    the committed PMC pass, x64 lanes, / samples per launch) and the lane-op
    rate that implies at the measured sample rate, against the nominal
    256 CU x 64 lanes x 2.4 GHz = 39.3e12 lane-ops/s.  (gfx950 issues the
    full-rate integer ops faster than one wave per 4 cycles --
    profiles/valu_microbench_r01.txt measures 58-65e12 lane-ops/s for
    add/xor/shift and 37e12 for the half-rate ops -- so the fraction of the
    nominal figure can exceed 1.)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            e = json.load(f).get(key, {})
        per_sample = e["SQ_INSTS_VALU"] * 64.0 / samples_per_launch
    except (OSError, ValueError, KeyError):
        return None
    return {"valu_instr_per_sample": per_sample,
            "lane_ops_per_s": per_sample * samples_per_s,
            "frac_of_nominal_39e12": per_sample * samples_per_s / 39.3e12}


def _pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json), or None if that workload was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            return json.load(f).get(key, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


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
    phase = torch.empty(n, dtype=torch.int32, device=dev)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    ca.fill_phase_ramp(phase, index0, w["shift"])
    if args.input == "random":
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        phase.random_(-2**31, 2**31 - 1, generator=gen)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
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
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
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
        print(json.dumps({
            "metric": "Msamples/sec (%s)" % args.workload,
            "value": float(world) * n * args.steps / elapsed / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, w["desc"]),
                       "samples_per_gpu": n, "pw": tab.pw, "ow": tab.ow,
                       "entries": tab.entries,
                       "kernel": "quad_lookup" if quad else "table_lookup",
                       "input": args.input, "parallelism": "shard%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": _pmc_traffic(args.workload),
                         "bytes_per_sample": w["bytes"],
                         "kernel_ms_avg": avg * 1e3},
            "bit_exact_vs_oracle": ok}))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-other-paths", action="store_true",
                    help="skip the informational rates of the other entry "
                    "points after the default (cfg2) run")
    ap.add_argument("--log2-samples", type=int, default=30,
                    help="samples per GPU = 2^this")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="also time gathering the outputs on rank 0 (RCCL)")
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
    ap.add_argument("--generic", action="store_true",
                    help="force the generic (not unrolled) kernel")
    args = ap.parse_args()

    import cordic_amd as ca

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1 or "RANK" in os.environ:
        # launched by torch.distributed.run: RCCL process group (also for a
        # single rank, so that the collective path can be exercised on 1 GPU)
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
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
    a = torch.empty(n, dtype=sdt, device=dev)
    b = torch.empty(n, dtype=sdt, device=dev)
    if w["kind"] == "p2r":
        phase = torch.empty(n, dtype=torch.int32, device=dev)
        ca.fill_phase_ramp(phase, index0, w["shift"])
        if args.input == "random":
            gen = torch.Generator(device=dev).manual_seed(1234 + rank)
            phase.random_(-2**31, 2**31 - 1, generator=gen)
        if io16:
            phase = phase.to(torch.int16)   # the low 16 bits: n mod 2^16

        plan = ca.Plan(cfg)

        def step():
            plan.p2r_const(x0, y0, phase, a, b)
    elif w["kind"] == "p2rxy":
        phase = torch.empty(n, dtype=torch.int32, device=dev)
        xin = torch.empty(n, dtype=torch.int32, device=dev)
        yin = torch.empty(n, dtype=torch.int32, device=dev)
        ca.fill_phase_ramp(phase, index0, w["shift"])
        ca.fill_iq_ramp(xin, yin, index0, 0x9E3779B1, 0x85EBCA77, iw)
        if args.input == "random":
            gen = torch.Generator(device=dev).manual_seed(1234 + rank)
            phase.random_(-2**31, 2**31 - 1, generator=gen)

        def step():
            ca.p2r(cfg, xin, yin, phase, a, b)
    elif w["kind"] == "r2p":
        xin = torch.empty(n, dtype=torch.int32, device=dev)
        yin = torch.empty(n, dtype=torch.int32, device=dev)
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

    for _ in range(args.warmup):
        step()
    barrier()

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
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    kern_avg_s = float(np.mean(kern_ms)) / 1e3

    # ---- after the timed region: correctness of what was just computed
    d = torch.zeros(1, dtype=torch.int64, device=dev)
    ca.digest_u32(a.view(torch.int32), index0 // (2 if io16 else 1), d)
    ca.digest_u32(b.view(torch.int32),
                  index0 // (2 if io16 else 1) + (1 << 40), d)
    if dist is not None:
        dist.all_reduce(d, op=dist.ReduceOp.SUM)     # digests of shards add
    torch.cuda.synchronize()
    digest = int(d.cpu().numpy().view(np.uint64)[0])

    check = None
    if rank == 0:
        import oracle_lib as O
        ocfg = O.config_cli(MODE[m], iw, ow, xtra, pw, ns)
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

    # ---- constant-vector feeds: also time the full-recurrence kernel (every
    # sample runs all micro-rotations) so both numbers are on record
    full = None
    if (w["kind"] in ("p2r", "nco") and not args.no_seed and not args.generic
            and plan.seed_info["stages"] > 0):
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

    gather_ms = None
    if args.gather and dist is not None:
        # collect the output shards on rank 0 (RCCL gather over xGMI); timed
        # separately, never folded into `value`
        outs = None
        if rank == 0:
            outs = [torch.empty(2 * n, dtype=sdt, device=dev)
                    for _ in range(world)]
        ab = torch.cat([a, b])
        barrier()
        t1 = time.perf_counter()
        dist.gather(ab, outs, dst=0)
        barrier()
        gather_ms = (time.perf_counter() - t1) * 1e3
        del outs, ab
        # end to end, pipelined: one step computed in 8 chunks, every chunk's
        # outputs sent to rank 0 on a side stream while the next chunk is
        # being computed (cordic_amd/shard.py:pipelined_gather)
        gather_pipelined = None
        if w["kind"] == "p2r":
            from cordic_amd.shard import pipelined_gather
            a3, b3 = torch.zeros_like(a), torch.zeros_like(b)
            comm = torch.cuda.Stream(device=dev)

            def compute_chunk(lo, hi):
                plan.p2r_const(x0, y0, phase[lo:hi], a3[lo:hi], b3[lo:hi])
            barrier()
            t1 = time.perf_counter()
            got = pipelined_gather(compute_chunk, [a3, b3], chunks=8, dst=0,
                                   comm_stream=comm)
            barrier()
            ms = (time.perf_counter() - t1) * 1e3
            ok = bool(torch.equal(a3, a) and torch.equal(b3, b))
            if rank == 0:
                ok = ok and bool(torch.equal(got[0][0], a)
                                 and torch.equal(got[1][0], b))
            gather_pipelined = {"ms_compute_and_gather": ms, "chunks": 8,
                                "outputs_identical": ok}
            del got, a3, b3

    if rank == 0:
        total = float(world) * n * args.steps
        value = total / elapsed / 1e6
        achieved = w["bytes"] * n / kern_avg_s / 1e9
        traffic = _pmc_traffic(args.workload
                               + ("_noseed" if args.no_seed else ""))
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
            "config": {
                "workload": "%s: %s" % (args.workload, w["desc"]),
                "samples_per_gpu": n,
                "iw": cfg.iw, "ow": cfg.ow, "ww": cfg.ww, "pw": cfg.pw,
                "nstages": cfg.nstages, "rotations": cfg.nlive,
                "kernel": "generic" if args.generic else (
                    "unrolled" if (args.no_seed or w["kind"] in ("r2p", "p2rxy"))
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
                "traffic": traffic,
                "bytes_per_sample": w["bytes"],
                "kernel_ms_avg": kern_avg_s * 1e3,
                "kernel_ms_min": float(np.min(kern_ms)),
                "note": ("table-seeded kernel: within ~5 % of an arithmetic-"
                         "free kernel with the same 4 B in / 8 B out traffic "
                         "(tools/hbm_pattern_bench.hip reaches 0.62-0.66 of "
                         "peak for this pattern); the full-recurrence kernel "
                         "is integer-VALU bound (DESIGN.md 4.5)"
                         if (w["kind"] in ("p2r", "nco") and not args.no_seed
                             and not args.generic) else
                         "integer-VALU bound, not HBM bound: DESIGN.md 4.5"),
            },
            "valu": _pmc_valu(args.workload
                              + ("_noseed" if args.no_seed else ""),
                              1 << 30, n / kern_avg_s),
            "bit_exact_vs_oracle": check,
            "digest": "%016x" % digest,
        }
        if full is not None:
            out["full_recurrence_kernel"] = full
        if gather_ms is not None:
            out["gather_ms"] = gather_ms
            if gather_pipelined is not None:
                out["gather_pipelined"] = gather_pipelined
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        if world == 1 and args.workload == "cfg2" and not args.no_other_paths:
            del a, b, phase
            torch.cuda.empty_cache()
            out["other_paths"] = other_paths(ca, dev)
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
