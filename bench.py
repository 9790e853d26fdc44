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
layer of the C ABI.  `value` is always compute only; with
more than one GPU the line ALSO carries, behind the timed region and never
inside `value`, the final gather onto one GPU (8e (ii)): `gather.rccl` =
ncclSend / ncclRecv of the C++ layer between the ranks, `gather.peer` = peer
copies of the one-process layout, and `scale` = compute only next to
compute + gather.  A gather that cannot run is a labelled `gather.*.error`;
nothing behind the timed region can lose the main line (LineGuard).

`--gpus N` always means N GPUs: started by torch.distributed.run the world
size must equal N; started plainly with N > 1 the script re-executes itself
under torch.distributed.run with N ranks (one GPU each); fewer than N visible
GPUs is an error, never a silent 1-GPU run.  `--single-process` instead
drives all N devices from one host process (cordic_group, no process group).

Rank 0 prints ONE JSON line of at most 4 KB (tools/bench_line.py: the contract
keys, `roofline`, `cpu_baseline`, `digest_check`, `full_recurrence`, `scale`;
numbers and short tokens only) and writes everything else it collected to
--detail (default ./bench_detail.json).  The default command measures: the K
timed steps, the oracle's digest of EVERY output, the full-recurrence kernel,
one second of sustained running (power / clock), three rocprofv3 counter
passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU) and the CPU baseline -- about
half a minute.  `--full` adds what round 5's default line carried: copy probes
on the run's own arrays, the other BASELINE configurations at their sizes, the
host-array entry points and the small-batch sweep (minutes; all of it lands in
the detail file).  The helpers that do not decide the metric live in
tools/bench_*.py.
"""
import argparse
import json
import os
import sys
import threading
import time

T_START = time.perf_counter()   # `wall_s` of the line counts from here
ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import build_stamp  # noqa: E402
from bench_common import (HBM_PEAK_GBS, MODE, RW, SHARE_GPU, WORKLOADS,  # noqa: E402
                          RawWords, claim_stdout, coll_device, dist_init, emit,
                          ranks_on_this_node, spot_indices, usable_cpus)
from bench_line import publish  # noqa: E402
from bench_oracle import (cpu_baseline, oracle_digest_leg,  # noqa: E402
                          reduce_digest_legs)
from bench_pmc import from_profile, measure_pmc  # noqa: E402
from bench_power import PowerSampler, finish_power, start_power  # noqa: E402,F401
from bench_probes import copy_probe  # noqa: E402
from bench_valu import add_valu  # noqa: E402

XGMI_LINK_GBS = 153.0       # MI355X_MICROARCH.md: one xGMI link, one direction
GATHER_CHUNKS = 8           # pieces per shard behind the compute


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


class LineGuard:
    """Nothing that runs BEHIND the timed region may lose the line.  Phases
    that can block for reasons outside this script -- an RCCL communicator that
    never forms, a peer copy between devices that cannot see each other -- are
    armed with a limit; when it expires rank 0 prints the line it has (the
    phase labelled `error: timed out`) and every rank leaves with status 0,
    so the launcher sees a finished job and the driver a complete record."""

    def __init__(self, rank, detail_path=None):
        self.rank = rank
        self.detail_path = detail_path
        self.line = None            # rank 0: the record so far (a dict)
        self.on_timeout = None      # rank 0: phase -> None, patches self.line
        self._gen = 0
        self._lock = threading.Lock()

    def arm(self, phase, limit_s):
        with self._lock:
            self._gen += 1
            gen = self._gen
        t = threading.Thread(target=self._watch, args=(phase, limit_s, gen),
                             daemon=True)
        t.start()

    def disarm(self):
        with self._lock:
            self._gen += 1

    def _watch(self, phase, limit_s, gen):
        end = time.monotonic() + limit_s
        while time.monotonic() < end:
            time.sleep(0.05)
            with self._lock:
                if self._gen != gen:
                    return
        with self._lock:
            if self._gen != gen:
                return
            self._gen += 1
        sys.stderr.write("bench.py: rank %d: '%s' exceeded %.0f s\n"
                         % (self.rank, phase, limit_s))
        if self.rank == 0 and self.line is not None:
            try:
                if self.on_timeout:
                    self.on_timeout(phase, limit_s)
                if self.detail_path is None:
                    emit(json.dumps(self.line))
                else:
                    publish(self.line, self.detail_path, emit)
            finally:
                os._exit(0)
        time.sleep(3.0)             # let rank 0 write first
        os._exit(0)


def agree(pg, ok):
    """True iff `ok` on every rank (host process group: a failed rank must not
    leave the others inside a collective that never completes)."""
    if pg is None:
        return bool(ok)
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=pg)
    return bool(t.item())


def _gather_numbers(ms_by_chunks, compute_ms, n, total, nlocal):
    """ms per step with the outputs collected on one GPU -> what SURVEY 8(e)
    asks for: rate into the root, how much of the transfer hid behind the
    compute, and the per-link model beside it."""
    serial = ms_by_chunks.get(1)
    piped = ms_by_chunks.get(GATHER_CHUNKS, serial)
    remote = float(total - 1) * n * 8.0      # bytes that cross devices
    out = {"ms": piped, "chunks": GATHER_CHUNKS,
           "ms_chunks1": serial, "ms_compute_only": compute_ms,
           "bytes_into_root": float(total) * n * 8.0,
           "bytes_from_other_devices": remote,
           "GBps_into_root": float(total) * n * 8.0 / (piped * 1e-3) / 1e9}
    # every remote shard arrives over its own xGMI link (7 links, <= 7 peers):
    # the links run in parallel, so the model is ONE shard over ONE link
    out["model_ms"] = (n * 8.0 / (XGMI_LINK_GBS * 1e9) * 1e3) if total > 1 else 0.0
    out["model"] = ("one shard's outputs (8 B x 2^%.1f samples) over one xGMI "
                    "link at %.0f GB/s; %d peer links in parallel"
                    % (np.log2(n), XGMI_LINK_GBS, max(0, total - 1)))
    if serial and serial > compute_ms:
        copy = serial - compute_ms          # the transfer by itself
        out["ms_transfer_alone"] = copy
        hidden = serial - piped             # what piece-wise forwarding saved
        out["overlap_frac"] = max(0.0, min(1.0, hidden / min(copy, compute_ms)))
        if remote:
            out["GBps_from_other_devices_transfer_alone"] = remote / (
                copy * 1e-3) / 1e9
    out["Msamples_per_s"] = float(total) * n / (piped * 1e-3) / 1e6
    return out


def gather_peer(args, grp, step, devices, n, n_total, total, compute_ms,
                oracle_sum):
    """One process, every device: cordic_group_set_gather -- each finished piece
    leaves for the root device with hipMemcpyPeerAsync on the shard's copy
    stream (SDMA over xGMI, no CUs) while the next piece computes."""
    import cordic_amd as ca
    root = ca.Group(grp.cfg, devices=[devices[0]], first_shard=0, total_shards=1)
    try:
        root.reserve(n_total, 0)
        _, rp, _ = root.buffers(0)
        k = max(3, min(args.steps, 10))
        ms = {}
        for chunks in (1, GATHER_CHUNKS):
            grp.set_gather(devices[0], rp[2], rp[3], chunks)
            step(grp)
            grp.sync()
            t1 = time.perf_counter()
            for _ in range(k):
                step(grp)
            grp.sync()
            ms[chunks] = (time.perf_counter() - t1) / k * 1e3
        got = root.digest(n_total)
        want = grp.digest(n_total)
        grp.set_gather(-1)
    finally:
        root.close()
    res = _gather_numbers(ms, compute_ms, n, total, total)
    res.update({
        "mode": "hipMemcpyPeerAsync, %d pieces per shard behind the compute "
                "(cordic_group_set_gather); root = device %d"
                % (GATHER_CHUNKS, devices[0]),
        "steps": k, "outputs_identical": got == want,
        "root_digest": "%016x" % got,
        "root_digest_equals_oracle": (got == oracle_sum
                                      if oracle_sum is not None else None)})
    return res


def gather_rccl(args, grp, step, dist, host_pg, dev, devices, n, n_total, rank,
                world, barrier, compute_ms, oracle_sum):
    """One process per GPU: the C++ layer's own forwarding -- ncclSend of every
    finished piece to shard 0, whose process posts the matching ncclRecv
    (cordic_group_set_gather_rccl).  torch.distributed only carries the
    128-byte RCCL id and the barriers.  Every fallible step is agreed on by
    all ranks before the next collective one starts."""
    import cordic_amd as ca
    uid, err = None, None
    try:
        uid = ca.rccl_unique_id()           # also: can librccl be opened here?
    except Exception as e:
        err = repr(e)
    if not agree(host_pg, uid is not None):
        return {"error": "RCCL not usable on every rank: %s" % err}
    box = [uid if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=host_pg)
    ok = True
    try:
        grp.rccl_init(box[0])
    except Exception as e:
        ok, err = False, repr(e)
    if not agree(host_pg, ok):
        return {"error": "cordic_group_rccl_init failed on a rank: %s" % err}
    root = rp = None
    try:
        if rank == 0:
            root = ca.Group(grp.cfg, devices=[devices[0]], first_shard=0,
                            total_shards=1)
            root.reserve(n_total, 0)
            _, rp, _ = root.buffers(0)
    except Exception as e:
        ok, err = False, repr(e)
    if not agree(host_pg, ok):
        if root is not None:
            root.close()
        return {"error": "no room for the gathered arrays on the root: %s" % err}
    k = max(3, min(args.steps, 10))
    ms = {}
    try:
        for chunks in (1, GATHER_CHUNKS):
            grp.set_gather_rccl(0, rp[2] if rp else None, rp[3] if rp else None,
                                chunks)
            step(grp)
            barrier()
            t1 = time.perf_counter()
            for _ in range(k):
                step(grp)
            barrier()
            t = torch.tensor([(time.perf_counter() - t1) / k * 1e3],
                             dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=host_pg)
            ms[chunks] = float(t.item())
        grp.set_gather_rccl(-1)
        d = grp.digest(n_total)
    except Exception as e:
        ok, err = False, repr(e)
    if not agree(host_pg, ok):
        if root is not None:
            root.close()
        return {"error": "forwarding failed on a rank: %s" % err}
    t = torch.tensor([d - (1 << 64) if d >= 1 << 63 else d], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=host_pg)
    want = int(t.item()) & 0xFFFFFFFFFFFFFFFF
    res = None
    if rank == 0:
        got = root.digest(n_total)
        root.close()
        res = _gather_numbers(ms, compute_ms, n, world, 1)
        res.update({
            "mode": "RCCL ncclSend / ncclRecv to shard 0, %d pieces per shard "
                    "behind the compute (cordic_group_set_gather_rccl)"
                    % GATHER_CHUNKS,
            "steps": k, "outputs_identical": got == want,
            "root_digest": "%016x" % got,
            "root_digest_equals_oracle": (got == oracle_sum
                                          if oracle_sum is not None else None),
            "rccl_library": os.environ.get("CORDIC_RCCL_LIB") or "librccl"})
    return res


def single_process_block(args, ndev, expect):
    """The C++ one-process layer (cordic_group over every GPU, no process
    group) measured on the same node: a SEPARATE `bench.py --single-process`
    process with a time limit, so that nothing it does can take the main
    result down with it.  Returns a digest of its line (its `gather.peer`
    included)."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE",
                        "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "BENCH_SELF_SPAWNED")}
    import tempfile
    fd, dpath = tempfile.mkstemp(prefix="bench_sp_", suffix=".json", dir="/tmp")
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(ndev),
           "--single-process", "--workload", args.workload, "--steps",
           str(args.steps), "--warmup", str(args.warmup), "--log2-samples",
           str(args.log2_samples), "--input", args.input,
           "--no-cpu-baseline", "--no-pmc", "--no-power", "--no-full-digest",
           "--detail", dpath]
    for flag, on in (("--no-seed", args.no_seed), ("--generic", args.generic),
                     ("--static-chunks", args.static_chunks)):
        if on:
            cmd.append(flag)
    try:
        r = subprocess.run(cmd, env=env, text=True, capture_output=True,
                           timeout=args.single_process_limit)
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not rows:
            return {"error": "rc %d: %s" % (r.returncode, r.stderr[-400:])}
        with open(dpath) as f:
            d = json.load(f)
    finally:
        try:
            os.unlink(dpath)
        except OSError:
            pass
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


def run_group(args, w, launch):
    """p2r / nco / r2p workloads through the C++ multi-GPU layer of the C ABI
    (cordic_group_*): this process drives `nlocal` shards of `total`."""
    import cordic_amd as ca
    import oracle_lib as O
    from gpu_util import cpu_digest

    m, iw, ow, xtra, pw, ns = w["cli"]
    cfg = ca.Config.from_cli(MODE[m], iw, ow, xtra, pw, ns)
    if args.generic:
        cfg = cfg.with_flags(cfg.flags | ca.FLAG_FORCE_GENERIC)
    if args.no_seed:
        cfg = cfg.with_flags(cfg.flags | ca.FLAG_NO_SEED)
    if args.static_chunks:
        cfg = cfg.with_flags(cfg.flags | ca.FLAG_STATIC_CHUNKS)
    if args.no_tails:
        cfg = cfg.with_flags(cfg.flags | ca.FLAG_NO_TAILS)
    if args.no_lj:
        cfg = cfg.with_flags(cfg.flags | ca.FLAG_NO_LJ)

    dist = host_pg = None
    rank, world, local = 0, 1, 0
    if launch == "torchrun":
        import torch.distributed as dist
        rank = int(os.environ["RANK"])
        world = int(os.environ["WORLD_SIZE"])
        local = 0 if SHARE_GPU else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist_init(dist, rank, world, local)
        # host-side group: agreement and waiting without a GPU kernel spinning
        host_pg = dist.new_group(backend="gloo")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if launch == "single-process":
        total, nlocal, first = args.gpus, args.gpus, 0
        devices = [0] * args.gpus if SHARE_GPU else list(range(args.gpus))
    else:
        total, nlocal, first, devices = world, 1, rank, [local]
    guard = LineGuard(rank, args.detail)
    phases = {}                 # seconds per phase of this command (rank 0)
    t_phase = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        phases[name] = phases.get(name, 0.0) + now - t_phase[0]
        t_phase[0] = now

    n = 1 << args.log2_samples
    n_total = n * total
    kind = w["kind"]
    x0, y0 = (1 << (iw - 1)) - 1, 0
    grp = ca.Group(cfg, devices=devices, first_shard=first, total_shards=total)
    # since round 6 the library takes its arrays as hipMalloc hands them out
    # unless asked (include/cordic_amd.h, "Placement"); the bench asks -- for
    # --placement-spares arrays of ITS OWN memory (six: 24 GiB, under a tenth of
    # what is free), spent only while no pair of written arrays is fast -- and
    # says so in the line (roofline.placement)
    grp.set_placement(0 if args.no_placement else args.placement_spares)
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
        # kernel by itself (cordic_plan_set_min_samples; --log2-samples under
        # ~21 without CORDIC_SEED_MIN_SAMPLES=0): the line names what ran
        seeded, seed_stages, tails = False, 0, []

    # ---- same-run copy probes on the very arrays of shard 0 (before)
    _, ptrs, _ = grp.buffers(0)
    probes = []
    lap("setup")
    if args.copy_probe:
        with torch.cuda.device(devices[0]):
            probes.append(copy_probe(ptrs, n, RW[kind]))

    lap("copy_probe")
    sampler = start_power(devices[0], rank == 0 and not args.no_power)

    # W untimed warm-up steps, the last of them BEHIND the barrier: the ranks'
    # rendezvous (an RCCL kernel plus host round trips) leaves the device idle
    # long enough for its clock to sag, and the step that follows runs ~0.3 ms
    # long (profiles/r05/torchrun_vs_direct.txt) -- that step is warm-up, not
    # workload.  The timed region still starts behind barrier + synchronize.
    for _ in range(max(0, args.warmup - 1)):
        step(grp)
    barrier()
    if args.warmup >= 1:
        step(grp)
        grp.sync()
        torch.cuda.synchronize()

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
    lap("timed_region")
    power = finish_power(sampler, lambda: step(grp), grp.sync, t0, elapsed,
                         args.steps, float(nlocal) * n,
                         seconds=2.0 if args.full else 1.0)
    lap("sustained_window")
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
                            threads=max(1, usable_cpus()
                                        // ranks_on_this_node(world)))
    legs = None
    oracle_sum = None
    if leg is not None:
        legs = reduce_digest_legs(dist, dev, world,
                                  local_digest == leg["digest"], leg)
        oracle_sum = legs[2]
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
            all_ok, samples_all, _, slowest = legs
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

    lap("oracle_digest")
    # ---- constant-vector feeds: also time the full-recurrence kernel (every
    # sample runs all micro-rotations) so both numbers are on record
    full = None
    if seeded:
        grp2 = ca.Group(cfg.with_flags(cfg.flags | ca.FLAG_NO_SEED), devices=devices,
                        first_shard=first, total_shards=total)
        for sh in range(nlocal):        # same inputs, bit for bit
            grp2.reserve(n_total, 1 if kind == "p2r" else 0)
            if kind == "p2r":
                _, p1, _ = grp.buffers(sh)
                grp2.write(sh, 0, 0, RawWords(p1[0], n))
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
                "what": "every sample runs every micro-rotation (what "
                        "north_star describes literally); the default kernel "
                        "looks the first %d up and runs the rest" % seed_stages,
                "from_profile": from_profile(args.workload + "_noseed")}
        grp2.close()

    # ---- same-run copy probes again (after): the memory system may have
    # changed state under sustained load
    lap("full_recurrence")
    if args.copy_probe:
        with torch.cuda.device(devices[0]):
            probes.append(copy_probe(ptrs, n, RW[kind]))
        lap("copy_probe")
    grp_placement = grp.placement(0)

    # ---- the record so far: everything the metric needs.  What follows (the
    # gathers, the one-process block, the CPU baseline, other_paths) is added
    # to it; none of it can take it away (LineGuard).
    out = None
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
            spares_allowed=args.placement_spares,
            what="cordic_group_set_placement(grp, %d): arrays allocated with "
                 "two spares -- further ones, up to that number and a tenth of "
                 "the free memory, only while no pair of written arrays is fast "
                 "--, arithmetic-free probes of the job's traffic over the role "
                 "assignments, best kept (off in the library by default; "
                 "--no-placement: as hipMalloc hands them out)"
                 % args.placement_spares)
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
                         ".run%s); RCCL: digest all-reduce, and behind the "
                         "timed region the final gather" % (
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
        guard.line = out

    # ---- the final gather onto one GPU, ALWAYS with more
    # than one GPU (and with --gather on one), never inside `value`
    compute_ms = elapsed / args.steps * 1e3
    gather = None

    def timed_out(phase, limit_s):
        if phase in ("rccl", "peer"):
            g = out.setdefault("gather", gather if gather is not None else {})
            g.setdefault(phase, {})["error"] = (
                "timed out after %.0f s" % limit_s)
        else:
            out["launch"]["%s_error" % phase] = (
                "timed out after %.0f s" % limit_s)
    guard.on_timeout = timed_out
    if (total > 1 or args.gather) and not args.no_gather:
        gather = {}
        if launch == "single-process":
            guard.arm("peer", args.gather_limit)
            try:
                gather["peer"] = gather_peer(args, grp, step, devices, n, n_total,
                                             total, compute_ms, oracle_sum)
            except Exception as e:            # never lose the main line
                gather["peer"] = {"error": repr(e)}
            guard.disarm()
        elif dist is not None:
            if host_pg is not None:
                dist.barrier(group=host_pg)   # the limits start together
            guard.arm("rccl", args.gather_limit)
            try:
                gather["rccl"] = gather_rccl(args, grp, step, dist, host_pg, dev,
                                             devices, n, n_total, rank, world,
                                             barrier, compute_ms, oracle_sum)
            except Exception as e:
                gather["rccl"] = {"error": repr(e)}
            guard.disarm()
    grp.close()

    single = None
    lap("gather")
    if (launch == "torchrun" and not args.no_single_process_check
            and (world > 1 or os.environ.get("BENCH_FORCE_SINGLE_CHECK"))):
        # the C++ one-process layer on the same GPUs, for the record (and the
        # peer-copy gather): rank 0 drives every device while the other ranks
        # wait on the HOST (a gloo barrier: no GPU kernel spins meanwhile)
        torch.cuda.empty_cache()
        if rank == 0:
            try:
                single = single_process_block(args, world, digest)
            except Exception as e:            # never lose the main line
                single = {"error": repr(e)}
        guard.arm("ranks_rejoin", args.single_process_limit + 120.0)
        # (a rank may have left: LineGuard)
        try:
            dist.barrier(group=host_pg)
        except Exception as e:
            if out is not None:
                out["launch"]["rejoin_error"] = repr(e)
        guard.disarm()

    if rank == 0:
        if gather is not None:
            sg = (single or {}).get("gather") or {}
            if "peer" not in gather and "peer" in sg:
                gather["peer"] = dict(sg["peer"], measured_by=(
                    "the embedded one-process run (single_process_cordic_group)"))
            out["gather"] = gather
            out["scale"] = {
                "what": "SURVEY 8(e): (i) compute only = `value`; (ii) the same "
                        "steps with every output forwarded to one GPU",
                "compute_only": {"ms_per_step": compute_ms,
                                 "Msamples_per_s": out["value"]},
                "compute_plus_gather": {
                    k: ({"ms_per_step": v["ms"],
                         "Msamples_per_s": v["Msamples_per_s"],
                         "overlap_frac": v.get("overlap_frac"),
                         "model_ms": v.get("model_ms")}
                        if "error" not in v else {"error": v["error"]})
                    for k, v in gather.items() if v}}
        if single is not None:
            out["single_process_cordic_group"] = single
        lap("single_process_check")
        roof = out["roofline"]
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
            lap("pmc_passes")
        add_valu(roof, n / kern_avg_s, pm, power, out["from_profile"],
                 args.workload)
        if not args.no_cpu_baseline and total == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, leg=leg)
            lap("cpu_baseline")
        if total == 1 and args.workload == "cfg2" and args.full:
            import bench_paths
            torch.cuda.empty_cache()
            out["other_paths"] = bench_paths.other_paths(args)
            try:
                out["other_paths"]["host_arrays"] = bench_paths.host_paths()
            except Exception as e:            # never lose the main line
                out["other_paths"]["host_arrays"] = {"error": repr(e)}
            try:
                out["other_paths"]["small_batches"] = bench_paths.small_batches()
            except Exception as e:            # never lose the main line
                out["other_paths"]["small_batches"] = {"error": repr(e)}
            try:
                out["other_paths"]["small_batches_xy"] = (
                    bench_paths.small_batches_xy())
            except Exception as e:            # never lose the main line
                out["other_paths"]["small_batches_xy"] = {"error": repr(e)}
            try:
                out["other_paths"]["small_batches_nco"] = (
                    bench_paths.small_batches_nco())
            except Exception as e:            # never lose the main line
                out["other_paths"]["small_batches_nco"] = {"error": repr(e)}
            lap("other_paths")
        out["phases_s"] = phases
        out["wall_s"] = time.perf_counter() - T_START
        publish(out, args.detail, emit)
        sys.stdout.flush()
    if dist is not None:
        guard.line = None                     # printed: nothing left to save
        guard.arm("shutdown", 120.0)
        try:
            dist.barrier(group=host_pg)
            dist.destroy_process_group()
        except Exception:
            pass
        guard.disarm()


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
    ap.add_argument("--full", action="store_true",
                    help="everything round 5's default command ran: copy "
                    "probes before and after the timed region, two seconds of "
                    "sustained running, and (cfg2, one GPU) the other BASELINE "
                    "configurations at their sizes, the host-array entry points "
                    "and the small-batch sweep; minutes instead of half a "
                    "minute, all of it in the detail file")
    ap.add_argument("--detail", default="bench_detail.json",
                    help="where the full record goes (the printed line is a "
                    "<= 4 KB selection of it: tools/bench_line.py); '-' or "
                    "/dev/null: nowhere")
    ap.add_argument("--copy-probe", action="store_true",
                    help="arithmetic-free twins of the kernel's traffic on the "
                    "run's own arrays, before and after the timed region "
                    "(roofline.copy_frac in the detail file; part of --full)")
    ap.add_argument("--no-other-paths", action="store_true",
                    help="(accepted for round 1-5 command lines: the other "
                    "entry points are measured only with --full)")
    ap.add_argument("--host-paths-only", action="store_true",
                    help="print only the host-array entry points' rates "
                    "(other_paths.host_arrays of the default line)")
    ap.add_argument("--small-batches-only", action="store_true",
                    help="print only the small-batch rows (one call per job "
                    "against one job set; other_paths.small_batches* of --full)")
    ap.add_argument("--log2-samples", type=int, default=None,
                    help="samples per GPU = 2^this (default: the size "
                    "BASELINE.json states for the workload -- 2^32 for cfg5 / "
                    "cfg5seq, 2^30 otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-digest", action="store_true",
                    help="skip the oracle digest over ALL samples (rank 0, "
                    "all host cores, ~2 s per 2^30 samples on 16 cores)")
    ap.add_argument("--no-copy-probe", action="store_true",
                    help="(accepted for older command lines: see --copy-probe)")
    ap.add_argument("--no-placement", action="store_true",
                    help="take the group's arrays as hipMalloc hands them out "
                         "instead of probing candidate allocations")
    ap.add_argument("--placement-spares", type=int, default=6,
                    help="spare arrays the bench lets cordic_group take while it "
                    "places the run's arrays (cordic_group_set_placement(grp, N); "
                    "1 = the library's own two)")
    ap.add_argument("--no-power", action="store_true",
                    help="skip the hwmon power / clock samples and the two "
                         "seconds of sustained running behind the timed region")
    ap.add_argument("--pmc-counters",
                    default="FETCH_SIZE,WRITE_SIZE,SQ_INSTS_VALU+SQ_INSTS_VALU_INT64",
                    help="comma-separated rocprofv3 counters, one pass each "
                    "(A+B: both in one pass)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip measuring roofline.traffic (two rocprofv3 --pmc "
                    "passes, FETCH_SIZE and WRITE_SIZE, over a 3-step run of "
                    "this workload; 1-GPU runs only, ~20 s)")
    ap.add_argument("--no-single-process-check", action="store_true",
                    help="multi-process runs: skip the extra one-process "
                    "cordic_group measurement on rank 0")
    ap.add_argument("--gather", action="store_true",
                    help="time collecting the outputs on one GPU also with "
                    "--gpus 1 (with more than one GPU it always runs)")
    ap.add_argument("--no-gather", action="store_true",
                    help="multi-GPU runs: skip the gather block")
    ap.add_argument("--single-process-limit", type=float, default=240.0,
                    help="seconds the embedded one-process run of a multi-rank "
                    "job may take (it is dropped with a labelled error beyond)")
    ap.add_argument("--gather-limit", type=float, default=180.0,
                    help="seconds the gather phase may take before the line "
                    "is printed without it (gather.*.error)")
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
    ap.add_argument("--no-lj", action="store_true",
                    help="A/B: the right-justified kernels (32-bit container "
                    "for WW <= 32) instead of the left-justified ones")
    ap.add_argument("--nstages", type=int, default=0,
                    help="experiments: the workload's core with this many stages "
                    "(gencordic -n); the line's config says so")
    ap.add_argument("--ramp-shift", type=int, default=-1,
                    help="experiments: phase ramp n << this (steeper ramps)")
    args = ap.parse_args()
    if args.log2_samples is None:
        args.log2_samples = WORKLOADS[args.workload].get("log2_samples", 30)
    if args.full:
        args.copy_probe = True
    if args.no_copy_probe:
        args.copy_probe = False
    if args.no_other_paths and args.full:
        raise SystemExit("bench.py: --full and --no-other-paths disagree")
    if args.ramp_shift >= 0:
        w0 = WORKLOADS[args.workload]
        w0["shift"] = args.ramp_shift
        w0["desc"] += " [--ramp-shift %d]" % args.ramp_shift
    if args.nstages:
        w0 = WORKLOADS[args.workload]
        w0["cli"] = tuple(w0["cli"][:5]) + (args.nstages,)
        w0["desc"] += " [--nstages %d]" % args.nstages
    # placement of the arrays: off in the library unless asked (round 6); the
    # bench asks -- run_group through cordic_group_set_placement, the stateless
    # workloads' cordic_arrays_alloc through the environment (also inherited
    # by the ranks and sub-runs this process starts)
    os.environ["CORDIC_GROUP_PLACEMENT"] = ("0" if args.no_placement
                                            else str(max(1, args.placement_spares)))

    if args.host_paths_only:
        import bench_paths
        print(json.dumps(bench_paths.host_paths()))
        return
    if args.small_batches_only:
        import bench_paths
        print(json.dumps({"small_batches": bench_paths.small_batches(),
                          "small_batches_xy": bench_paths.small_batches_xy(),
                          "small_batches_nco": bench_paths.small_batches_nco()}))
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
    import bench_direct
    return bench_direct.run_direct(args, w, launch)


if __name__ == "__main__":
    main()
