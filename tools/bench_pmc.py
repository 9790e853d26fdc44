"""bench_pmc.py -- roofline.traffic / instructions per sample of a bench line:
separate rocprofv3 --pmc passes over a short run of the same command, and the
committed counters of earlier sessions (profiles/pmc_latest.json)."""
import json
import os
import sys

from bench_common import KERNEL_OF, ROOT

BENCH = os.path.join(ROOT, "bench.py")

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


def subject_command(args):
    """The process a counter pass runs: tools/pmc_subject (a C caller of the C
    ABI: the workload's launches and nothing else, ~1 s) for the workloads it
    covers, otherwise a stripped `bench.py` child (16-bit containers, table
    cores, A/B flags: ~6 s of Python start-up per pass)."""
    import cordic_amd as ca
    from bench_common import MODE, WORKLOADS
    w = WORKLOADS[args.workload]
    subject = os.path.join(ROOT, "tools", "pmc_subject")
    native = (w["kind"] in ("p2r", "nco", "r2p", "p2rxy", "ddc")
              and not w.get("io16") and args.input == "ramp"
              and os.access(subject, os.X_OK)
              and not getattr(args, "no_lj", False))
    if native:
        m, iw, ow, xtra, pw, ns = w["cli"]
        flags = 0
        for on, bit in ((args.no_seed, ca.FLAG_NO_SEED),
                        (args.generic, ca.FLAG_FORCE_GENERIC),
                        (args.static_chunks, ca.FLAG_STATIC_CHUNKS),
                        (args.no_tails, ca.FLAG_NO_TAILS)):
            if on:
                flags |= bit
        return [subject, w["kind"], str(MODE[m]), str(iw), str(ow), str(xtra),
                str(pw), str(ns), str(args.log2_samples),
                str(w.get("shift", 0)), "3", "0x%x" % flags]
    base = [sys.executable, BENCH, "--workload",
            args.workload, "--steps", "3", "--warmup", "1", "--log2-samples",
            str(args.log2_samples), "--input", args.input, "--no-cpu-baseline",
            "--no-pmc", "--no-power", "--no-full-digest", "--no-placement",
            "--detail", os.devnull]
    for flag, on in (("--no-seed", args.no_seed), ("--generic", args.generic),
                     ("--static-chunks", args.static_chunks),
                     ("--no-tails", args.no_tails),
                     ("--no-lj", getattr(args, "no_lj", False))):
        if on:
            base.append(flag)
    return base


def measure_pmc(args):
    """HBM bytes per launch and VALU instructions per sample of the workload's
    kernel, MEASURED now: separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE,
    SQ_INSTS_VALU -- the first two do not fit one pass, and PMC is never
    combined with tracing) over a 3-launch run of the workload (subject_command), corrected as MI355X_MICROARCH.md prescribes for gfx950
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
    base = subject_command(args)
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    # one rocprofv3 pass per comma-separated item; "A+B" collects A and B in
    # the same pass (counters of one block that fit together: the SQ ones)
    counters = [c for c in args.pmc_counters.split(",") if c]
    for item in counters:
        names = [c for c in item.split("+") if c]
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            r = subprocess.run(["rocprofv3", "--pmc"] + names + [
                                "--output-format", "csv", "-d", td, "--"] + base,
                               cwd="/tmp", env=env, capture_output=True,
                               text=True, timeout=150)
            rows = {c: [] for c in names}
            for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"),
                               recursive=True):
                for row in csv.DictReader(open(f)):
                    if (row["Counter_Name"] in rows and kern
                            and kern in row["Kernel_Name"]):
                        # (not the one-block launch that builds a plan's seed
                        # image: same kernel, build mode, no samples)
                        try:
                            if int(row["Grid_Size"]) <= int(row["Workgroup_Size"]):
                                continue
                        except (KeyError, ValueError):
                            pass
                        rows[row["Counter_Name"]].append(float(row["Counter_Value"]))
            for ctr in names:
                if not rows[ctr]:
                    return {"error": "no %s rows for %s (rocprofv3 rc %d)"
                            % (ctr, kern, r.returncode)}
                vals[ctr] = (sum(rows[ctr]) / len(rows[ctr]), len(rows[ctr]))
    out = {"kernel": kern, "passes": counters,
           "subject": os.path.basename(base[0] if base[0] != sys.executable
                                       else base[1])}
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
    # the EXECUTED share of 64-bit integer instructions (v_mad_i64_i32 and the
    # 64-bit shifts / adds: all half rate), for the per-opcode VALU model
    for ctr, key in (("SQ_INSTS_VALU_INT64", "valu_int64_per_sample"),
                     ("SQ_INSTS_VALU_INT32", "valu_int32_per_sample")):
        if ctr in vals:
            out[key] = vals[ctr][0] * 64.0 / float(1 << args.log2_samples)
    return out
