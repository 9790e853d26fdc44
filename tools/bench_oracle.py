"""bench_oracle.py -- the CPU legs of a bench line: the oracle's digest of
EVERY sample a rank computed (checker) and the cpu_baseline block (the same
work, timed).  oracle/ is test infrastructure: only this leg uses it."""
import os
import time

import torch

from bench_common import (MODE, WORKLOADS, coll_device, cpu_model,
                          usable_cpus)

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
    fcw = 0x01234567 if kind in ("nco", "ddc") else (1 << w.get("shift", 0))
    cores = threads or usable_cpus()
    d, secs = O.job_digest(ocfg, "mix" if kind == "ddc" else kind, start, n, 0,
                           fcw, x0, y0, threads=cores)
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
    cores = usable_cpus()
    kind = 1 if w["kind"] == "r2p" else 0
    mul = 0x01234567 if w["kind"] in ("nco", "ddc") else (1 << w.get("shift", 0))
    x0 = (1 << (iw - 1)) - 1
    t0 = time.perf_counter()
    n1 = L.orc_throughput(C.byref(ocfg), kind, 1, 1.0, mul, x0, 0)
    one = n1 / (time.perf_counter() - t0)
    if leg is not None and leg["seconds"] >= 1.0:
        # the digest leg already pushed every sample of this run through the
        # oracle on all cores: that IS the bounded sample (not done twice)
        total, wall, cores = leg["samples"], leg["seconds"], leg["cores"]
        sample = "all %d samples of this run (%s), %.1f s" % (total, workload, wall)
        note = ("%d threads drawing 2^16-sample blocks; the outputs' digest is "
                "what digest_check compares with the device's; includes making "
                "the inputs and the digest (~4 %% of the work)" % cores)
    else:
        t0 = time.perf_counter()
        total = L.orc_throughput(C.byref(ocfg), kind, cores, seconds, mul,
                                 x0, 0)
        wall = time.perf_counter() - t0
        sample = "%d samples of %s, %.0f s" % (total, workload, seconds)
        note = "%d threads x 2^16-sample blocks" % cores
    return {
        "value": total / wall / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": sample,
        "sample_note": note + "; through oracle/liboracle.so: gcc -O2 scalar "
                       "restatement of the reference RTL -- the reference "
                       "itself has no CPU compute path",
        "value_1thread": one / 1e6,
        "cpu": cpu_model(),
        "cpus_visible": os.cpu_count(),
    }
