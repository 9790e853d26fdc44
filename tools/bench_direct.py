"""bench_direct.py -- workloads outside the cordic_group layer: 16-bit sample
containers (cfg1), per-sample x / y vectors (p2rxy, ddc) and the table cores,
on torch tensors through the stateless / plan entry points; one process per
GPU, same timing discipline as bench.py's run_group."""
import json
import os
import sys
import time

import numpy as np
import torch

import build_stamp

T0 = time.perf_counter()
from bench_common import (HBM_PEAK_GBS, MODE, SHARE_GPU, WORKLOADS, coll_device,
                          dist_init, emit, ranks_on_this_node, usable_cpus)
from bench_line import publish
from bench_oracle import cpu_baseline, oracle_digest_leg, reduce_digest_legs
from bench_pmc import from_profile, measure_pmc
from bench_power import finish_power, start_power
from bench_valu import add_valu

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
    for _ in range(max(0, args.warmup - 1)):
        tab.lookup(phase, out)
    barrier()
    if args.warmup >= 1:        # (the last warm-up behind the barrier: bench.py)
        tab.lookup(phase, out)
        torch.cuda.synchronize()
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
        line["wall_s"] = time.perf_counter() - T0
        publish(line, args.detail, emit)
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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
        cfg = cfg.with_flags(cfg.flags | ca.FLAG_FORCE_GENERIC)
    if args.no_seed:
        cfg = cfg.with_flags(cfg.flags | ca.FLAG_NO_SEED)
    if args.static_chunks:
        cfg = cfg.with_flags(cfg.flags | ca.FLAG_STATIC_CHUNKS)
    n = 1 << args.log2_samples
    index0 = rank * n                   # shard by global sample index
    x0, y0 = (1 << (iw - 1)) - 1, 0

    # ---- resident inputs / outputs
    io16 = bool(w.get("io16"))
    sdt = torch.int16 if io16 else torch.int32
    # the arrays of the job, placed by measurement (cordic_arrays_alloc: up to
    # two read + two written arrays; a third input is taken as it comes)
    nread = {"p2r": 1, "p2rxy": 2, "r2p": 2, "ddc": 2}[w["kind"]]
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
    elif w["kind"] == "ddc":
        # the fused NCO mixer: the per-sample vectors of p2rxy, the phase from
        # the in-kernel accumulator (no phase array: 16 B per sample)
        xin = arrays.tensor(0, torch.int32)
        yin = arrays.tensor(1, torch.int32)
        ca.fill_iq_ramp(xin, yin, index0, 0x9E3779B1, 0x85EBCA77, iw)
        if args.input == "random":
            gen = torch.Generator(device=dev).manual_seed(1234 + rank)
            xin.random_(-2**31, 2**31 - 1, generator=gen)
            yin.random_(-2**31, 2**31 - 1, generator=gen)
        plan = ca.Plan(cfg.with_flags(ca.FLAG_NO_TAILS) if args.no_tails else cfg)

        def step():
            plan.mix(0, 0x01234567, index0, xin, yin, a, b)
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
    for _ in range(max(1, args.warmup - 1)):
        step()
    barrier()
    if args.warmup >= 1:        # (the last warm-up behind the barrier: bench.py)
        step()
        torch.cuda.synchronize()
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
                         args.steps, float(n),
                         seconds=2.0 if args.full else 1.0)
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
                            threads=max(1, usable_cpus()
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
        elif w["kind"] == "ddc":
            ph = ((idx.astype(np.uint64) + np.uint64(index0))
                  * np.uint64(0x01234567) & np.uint64(0xffffffff))
            ra, rb = O.rotate(ocfg, xin[ti].cpu().numpy(),
                              yin[ti].cpu().numpy(), ph.astype(np.uint32))
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
                    if w["kind"] in ("p2rxy", "ddc") else
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
        add_valu(roof, n / kern_avg_s, pm, power, out["from_profile"],
                 args.workload)
        if full is not None:
            out["full_recurrence_kernel"] = full
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, leg=leg)
        out["wall_s"] = time.perf_counter() - T0
        publish(out, args.detail, emit)
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
