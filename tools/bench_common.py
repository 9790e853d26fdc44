"""bench_common.py -- what bench.py and its helper modules (tools/bench_*.py)
share: the workload table, the process-group helpers, the one-line emitter."""
import os
import sys

TOOLS = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(TOOLS)
for _p in (ROOT, os.path.join(ROOT, "tests"), TOOLS):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

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
    "ddc": dict(kind="ddc", cli=("p2r", 32, 32, 2, 32, 16), bytes=16,
                desc="fused NCO mixer (down-converter): per-sample x, y "
                "vectors rotated by the in-kernel accumulator phase = "
                "n*0x01234567, 16-stage 32-bit core (cordic_plan_mix)"),
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
    # BASELINE.json configs[4]: 4 G samples on one GPU (2 x 16 GiB of outputs)
    "cfg5": dict(kind="nco", cli=("p2r", 32, 32, 2, 32, 16), bytes=8,
                 log2_samples=32,
                 desc="fused NCO (phase = n*0x01234567) + 16-stage p2r, "
                 "store only"),
    "cfg5seq": dict(kind="nco", cli=("sp2r", 32, 32, 2, 32, 16), bytes=8,
                    log2_samples=32,
                    desc="fused NCO + seqcordic arithmetic (NSTAGES-2)"),
}
MODE = {"p2r": 0, "r2p": 1, "sp2r": 2, "sr2p": 3}


def usable_cpus():
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


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


KERNEL_OF = {"cfg2": "rotator_seeded", "cfg4": "rotator_seeded",
             "cfg5": "rotator_seeded", "cfg5seq": "rotator_seeded",
             "cfg1": "rotator_seeded", "cfg3": "topolar_lj",
             "nat32": "rotator_seeded", "nat24": "rotator_seeded",
             "nat16": "rotator_seeded", "natr2p24": "topolar_lj",
             "p2rxy": "rotator_xydir", "ddc": "rotator_xydir",
             "quadtbl": "quad_lookup",
             "quadtbl24": "quad_lookup", "sintbl": "table_lookup",
             "qtrtbl": "table_lookup", "qtrtbl16": "table_lookup",
             "qtrtbl24": "table_lookup"}


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
        print(line, flush=True)
        return
    data = (line + "\n").encode()
    while data:
        data = data[os.write(_STDOUT_FD, data):]


RW = {"p2r": (1, 2), "nco": (0, 2), "r2p": (2, 2)}   # arrays read / written


def spot_indices(n):
    """(offset, count) windows of a shard checked against the oracle: both
    ends and 61 windows spread through the middle."""
    win = min(n, 4096)
    offs = {0, n - win}
    for k in range(1, 62):
        offs.add(min(n - win, (k * (n // 62)) // 4 * 4))
    return sorted((o, win) for o in offs)


class RawWords:
    """A device address + word count, shaped like what Group.write accepts."""

    def __init__(self, ptr, n):
        self._p, self._n = ptr, n

    def data_ptr(self):
        return self._p

    def numel(self):
        return self._n
