"""bench_line.py -- the ONE line bench.py prints, and the file beside it.

The driver keeps a few KB of stdout and parses the last line: round 5's line
had grown to 20.7 KB and was lost (BENCH_r05.json: parsed null).  Since round 6
the record a run collects is split in two:

  bench_detail.json   everything (power windows, placement, probes, PMC detail,
                      the per-opcode model, other workloads, host paths ...),
                      written to --detail (default: ./bench_detail.json);
  the line            a fixed selection of it: the contract keys, `roofline`
                      and `cpu_baseline` as numbers and short tokens -- no
                      prose -- at most LINE_LIMIT bytes, strict JSON (no NaN /
                      Infinity), checked here before it is written.

tests/test_bench_launch.py holds the line to that on a CPU box (compact() on
round 5's committed 20 KB records) and on a GPU box (the real command at 1 and
8 ranks).
"""
import json
import math
import os
import sys

LINE_LIMIT = 4096

TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
       "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
CONFIG = ("workload", "samples_per_gpu", "iw", "ow", "ww", "pw", "nstages",
          "entries", "kernel", "input", "parallelism")
ROOF = ("bound", "achieved", "peak", "unit", "frac", "traffic",
        "bytes_per_sample", "kernel_ms_avg", "traffic_over_algorithmic",
        "valu_fraction", "valu_issue_fraction", "limiter", "sclk_ghz",
        "placement")
CPU = ("value", "unit", "cores", "kind", "sample", "value_1thread", "cpu")


def _num(v, digits=6):
    """floats shortened to `digits` significant figures (the line is for
    reading a result, the detail file keeps every bit); non-finite -> None"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        if not math.isfinite(v):
            return None
        if v == 0.0:
            return 0.0
        return float("%.*g" % (digits, v))
    return v


def _pick(src, keys):
    return {k: _num(src[k]) for k in keys if k in src}


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def compact(detail, detail_path=None):
    """the line (a dict) of a detail record"""
    line = _pick(detail, TOP)
    cfg = detail.get("config") or {}
    line["config"] = _pick(cfg, CONFIG)
    if "workload" in line["config"]:
        line["config"]["workload"] = _short(line["config"]["workload"], 160)
    if "kernel" in line["config"]:
        line["config"]["kernel"] = _short(line["config"]["kernel"], 80)
    roof = detail.get("roofline") or {}
    r = _pick(roof, ROOF)
    if isinstance(r.get("limiter"), str):       # token only: "power", "hbm" ...
        r["limiter"] = r["limiter"].split(":")[0].split()[0][:16]
    if isinstance(r.get("placement"), dict):
        r["placement"] = "on" if r["placement"].get("candidates") else "off"
    if "sclk_ghz" not in r and isinstance(roof.get("valu"), dict):
        r["sclk_ghz"] = _num(roof["valu"].get("sclk_ghz"))
    line["roofline"] = r
    cpu = detail.get("cpu_baseline")
    if cpu:
        c = _pick(cpu, CPU)
        if "sample" in c:
            c["sample"] = _short(c["sample"], 96)
        if "cpu" in c:
            c["cpu"] = _short(c["cpu"], 48)
        line["cpu_baseline"] = c
    dc = detail.get("digest_check")
    if dc:
        line["digest_check"] = {"samples": dc.get("samples"),
                                "equal": dc.get("equal")}
    if detail.get("bit_exact_vs_oracle") is not None:
        line["bit_exact_vs_oracle"] = detail["bit_exact_vs_oracle"]
    full = detail.get("full_recurrence_kernel")
    if full:
        line["full_recurrence"] = _pick(full, ("value_per_gpu", "hbm_frac"))
    sc = detail.get("scale")
    if sc:
        # (i) compute only, (ii) compute + the final gather onto
        # one GPU -- RCCL send / recv between the ranks (north_star), and peer
        # copies of the one-process layout; Msamples/s, None = did not run
        cpg = sc.get("compute_plus_gather") or {}

        def rate(k):
            v = cpg.get(k) or {}
            return _num(v.get("Msamples_per_s")) if "error" not in v else None
        s = {"compute_only": _num((sc.get("compute_only") or {})
                                  .get("Msamples_per_s")),
             "compute_plus_gather": rate("rccl")}
        if "peer" in cpg:
            s["compute_plus_gather_peer"] = rate("peer")
            if "rccl" not in cpg:
                s["compute_plus_gather"] = s["compute_plus_gather_peer"]
        line["scale"] = s
    b = detail.get("build") or {}
    if b.get("kernel_sources_sha256"):
        line["build"] = b["kernel_sources_sha256"][:16]
    if detail.get("wall_s") is not None:
        line["wall_s"] = _num(detail["wall_s"], 4)
    if detail_path:
        line["detail"] = detail_path
    return line


def _reject_constant(name):
    raise ValueError("not strict JSON: %s" % name)


def check(text):
    """the contract of the printed line; raises ValueError"""
    if "\n" in text:
        raise ValueError("the line holds a newline")
    if len(text.encode()) > LINE_LIMIT:
        raise ValueError("the line is %d bytes (limit %d)"
                         % (len(text.encode()), LINE_LIMIT))
    d = json.loads(text, parse_constant=_reject_constant)
    missing = [k for k in TOP + ("config", "roofline") if k not in d]
    if missing:
        raise ValueError("keys missing from the line: %s" % ", ".join(missing))
    return d


def render(detail, detail_path=None):
    """detail -> the text of the line, shrunk if it must be (it never has to
    with the selection above; the loop is the guarantee, not the plan)"""
    line = compact(detail, detail_path)
    for drop in (None, "wall_s", "build", "full_recurrence", "scale",
                 "cpu_baseline"):
        if drop:
            line.pop(drop, None)
        text = json.dumps(line, allow_nan=False, separators=(", ", ": "))
        if len(text.encode()) <= LINE_LIMIT:
            return text
    line["config"] = {"workload": _short(line["config"].get("workload", ""), 60)}
    return json.dumps(line, allow_nan=False)


def _clean(o):
    """non-finite floats -> None, recursively (the detail file is strict JSON
    too)"""
    if isinstance(o, float):
        return o if math.isfinite(o) else None
    if isinstance(o, dict):
        return {str(k): _clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    return o


def write_detail(detail, path):
    """-> the path written, or None (an unwritable cwd must not lose the
    line: /tmp is tried next)"""
    if path in (None, "", "-", os.devnull):         # asked not to write one
        return None
    for p in (path, os.path.join("/tmp", os.path.basename(path))):
        try:
            tmp = "%s.%d.tmp" % (p, os.getpid())
            with open(tmp, "w") as f:
                json.dump(_clean(detail), f, indent=1, allow_nan=False)
                f.write("\n")
            os.replace(tmp, p)
            return p
        except (OSError, ValueError) as e:
            sys.stderr.write("bench.py: detail file %s: %r\n" % (p, e))
    return None


def publish(detail, path, emit):
    """write the detail file, print the line"""
    where = write_detail(detail, path)
    emit(render(detail, where))
