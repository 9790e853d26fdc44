#!/usr/bin/env python3
"""DESIGN.md section 4.4's kernel table, GENERATED from the committed bench
records of one sweep (profiles/bench_r06/*.json: tools/gpu_session.sh sweep on
one box, one code state) -- so every number in it IS a committed line's.

    python tools/design_table.py            # print the block
    python tools/design_table.py --write    # replace the block in DESIGN.md

tests/test_design_numbers.py (CPU) regenerates the block, compares it with
DESIGN.md, and checks that every line was measured on the current kernel
sources (build.kernel_sources_sha256, tools/build_stamp.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWEEP = os.path.join(ROOT, "profiles", "bench_r06")
BEGIN = "<!-- BEGIN generated: tools/design_table.py -->"
END = "<!-- END generated -->"
ORDER = ["cfg2", "cfg4", "cfg5", "cfg5seq", "p2rxy", "ddc", "cfg3", "cfg1", "nat32",
         "nat24", "nat16", "natr2p24", "sintbl", "qtrtbl16", "qtrtbl24",
         "qtrtbl", "quadtbl", "quadtbl24"]


def load(name):
    try:
        with open(os.path.join(SWEEP, name)) as f:
            text = f.read()
    except OSError:
        return None
    try:                    # round 6: the run's detail record, one JSON document
        return json.loads(text)
    except ValueError:
        pass
    try:                    # rounds 1-5: the printed line
        lines = [ln for ln in text.splitlines() if ln.startswith("{")]
        return json.loads(lines[-1])
    except (ValueError, IndexError):
        return None


def lines():
    """{workload: {"ramp": line, "random": line}} of what the sweep holds"""
    out = {}
    for w in ORDER:
        e = {k: load("%s_%s.json" % (w, k)) for k in ("ramp", "random")}
        if e["ramp"]:
            out[w] = e
    return out


def fmt(v, nd=0):
    return "—" if v is None else ("%.*f" % (nd, v))


def block():
    rows = ["| workload | kernel | B/sample | instr/sample | ramp: Gsample/s | "
            "HBM frac | valu_fraction | valu_issue_fraction | bound | limiter | "
            "random: Gsample/s | HBM frac | full recurrence: Gsample/s |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for w, e in lines().items():
        a, b = e["ramp"], e["random"]
        ra = a["roofline"]
        valu = ra.get("valu") or {}
        full = a.get("full_recurrence_kernel")
        rows.append("| %s | `%s` | %d | %s | %s | %.3f | %s | %s | %s | %s | %s | %s | %s |" % (
            a["config"]["workload"].split(":")[0], a["config"]["kernel"],
            ra["bytes_per_sample"], fmt(valu.get("instr_per_sample"), 1),
            fmt(a["value"] / 1e3), ra["frac"], fmt(ra.get("valu_fraction"), 2),
            fmt(ra.get("valu_issue_fraction"), 2), ra.get("bound", "hbm"),
            (ra.get("limiter") or "—").split(":")[0].split(" (")[0],
            fmt(b["value"] / 1e3) if b else "—",
            ("%.3f" % b["roofline"]["frac"]) if b else "—",
            fmt(full["value_per_gpu"] / 1e3) if full else "—"))
    d = load("default.json")
    tail = []
    if d:
        cb = d.get("cpu_baseline") or {}
        tail.append("")
        tail.append("Default command at the END of the same sweep (`python bench.py`): %s Gsample/s, "
                    "%.3f of the HBM peak (same-run copy %s), `digest_check` over %d "
                    "samples equal: %s; CPU beside it: %s Msample/s on %s threads "
                    "(%s on one)." % (
                        fmt(d["value"] / 1e3), d["roofline"]["frac"],
                        fmt(d["roofline"].get("copy_frac"), 3),
                        (d.get("digest_check") or {}).get("samples", 0),
                        (d.get("digest_check") or {}).get("equal"),
                        fmt(cb.get("value")), cb.get("cores"),
                        fmt(cb.get("value_1thread"), 1)))
        o = load("default_driver_order.json")
        if o:
            tail.append("The same command as the session's FIRST bench run (the driver's order, "
                        "`default_driver_order.json`): %s Gsample/s, %.3f (same-run copy %s): each "
                        "run places its arrays anew (§3)." % (fmt(o["value"] / 1e3), o["roofline"]["frac"],
                                           fmt(o["roofline"].get("copy_frac"), 3)))
        b = d.get("build") or {}
        tail.append("Code state: `kernel_sources_sha256` %s, commit %s." % (
            (b.get("kernel_sources_sha256") or "?")[:16], (b.get("git_head") or "?")[:10]))
    return "\n".join([BEGIN] + rows + tail + [END])


def main():
    text = block()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        a, b = s.index(BEGIN), s.index(END) + len(END)
        open(p, "w").write(s[:a] + text + s[b:])
    else:
        print(text)


if __name__ == "__main__":
    main()
