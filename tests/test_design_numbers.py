"""DESIGN.md section 4.4 against the committed sweep:
the kernel table is GENERATED from profiles/bench_r06/*.json
(tools/design_table.py), and every one of those lines was measured on the
kernel sources as they are now (tools/build_stamp.py: SHA-256 over the device /
launch / table-builder sources without comments and white space) -- a kernel
edit after the sweep fails this test until the sweep is re-run."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import build_stamp  # noqa: E402
import design_table  # noqa: E402


def test_the_table_in_design_md_is_the_generated_one():
    lines = design_table.lines()
    if not lines:
        pytest.skip("profiles/bench_r06/ holds no sweep yet")
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert design_table.BEGIN in text and design_table.END in text
    a = text.index(design_table.BEGIN)
    b = text.index(design_table.END) + len(design_table.END)
    assert text[a:b] == design_table.block(), (
        "DESIGN.md section 4.4 differs from profiles/bench_r06: run "
        "python tools/design_table.py --write")


def test_every_sweep_line_was_measured_on_the_current_kernel_sources():
    lines = design_table.lines()
    if not lines:
        pytest.skip("profiles/bench_r06/ holds no sweep yet")
    now = build_stamp.kernel_sources_sha256()
    stale = []
    for w, e in lines.items():
        for kind, d in e.items():
            if d is None:
                continue
            got = (d.get("build") or {}).get("kernel_sources_sha256")
            if got != now:
                stale.append("%s_%s" % (w, kind))
            assert d["bit_exact_vs_oracle"] is True, (w, kind)
    d = design_table.load("default.json")
    assert d is not None, "the sweep's default line is missing"
    if (d.get("build") or {}).get("kernel_sources_sha256") != now:
        stale.append("default")
    assert not stale, ("measured on other kernel sources (re-run tools/"
                       "gpu_session.sh sweep): %s" % ", ".join(stale))
    assert d["digest_check"]["equal"] is True
    assert d["digest_check"]["samples"] == d["config"]["samples_per_gpu"]


def test_the_hash_ignores_comments_and_white_space_only():
    a = "int x = 1; // note\n/* block\n comment */\tint  y;"
    b = "int x = 1;\nint y;"
    assert build_stamp.normalized(a) == build_stamp.normalized(b)
    assert build_stamp.normalized("int x = 2;") != build_stamp.normalized("int x = 1;")


def _kernel_name(sym):
    """`void ns::kernel<template args>` of a demangled kernel symbol: up to the
    parameter list (the first '(' outside the template brackets), without the
    host stub's marker"""
    depth = 0
    for i, ch in enumerate(sym):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            sym = sym[:i]
            break
    return sym.replace("__device_stub__", "").strip()


def test_kernel_name_parsing():
    a = ("void cordic_amd::dev::rotator_seeded<cordic_amd::dev::WideLJ<29>, 16, 11, "
         "(cordic_amd::Feed)0, false, cordic_amd::dev::Io32, false, true, false>"
         "(cordic_amd::dev::CoreParams, cordic_amd::dev::SeedArgs, "
         "cordic_amd::dev::Io32::uvec const*, unsigned long)")
    b = a.replace("dev::rotator_seeded", "dev::__device_stub__rotator_seeded").replace(
        "cordic_amd::dev::Io32::uvec const*", "unsigned int __vector(4) const*")
    assert _kernel_name(a) == _kernel_name(b)
    assert _kernel_name(a).endswith("false, true, false>")


def test_cited_traces_are_of_kernels_the_library_still_has():
    """a committed rocprofv3 kernel-trace summary whose
    kernel signatures no longer exist in libcordic_amd.so is a trace of OTHER
    code.  Every cordic_amd kernel named in profiles/r06/**/kernel_stats.csv
    (and the default command's trace) must be a kernel of the current build
    (nm -C: the host stubs carry the same template arguments)."""
    import csv
    import glob
    import shutil
    import subprocess
    so = os.path.join(ROOT, "cordic_amd", "libcordic_amd.so")
    traces = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06", "**",
                                           "*kernel_stats.csv"), recursive=True))
    if not traces:
        pytest.skip("profiles/r06 holds no kernel trace yet")
    if not os.path.exists(so) or shutil.which("nm") is None:
        pytest.skip("no library / nm to compare with")
    have = set()
    for ln in subprocess.check_output(["nm", "-C", so], text=True).splitlines():
        parts = ln.split(None, 2)
        if len(parts) == 3 and "cordic_amd::" in parts[2]:
            have.add(_kernel_name(parts[2]))
    gone = []
    for path in traces:
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Name") or ""
                if "cordic_amd::" in name and _kernel_name(name) not in have:
                    gone.append("%s: %s" % (os.path.relpath(path, ROOT),
                                            _kernel_name(name)[:140]))
    assert not gone, "traces of kernels this build does not have:\n" + "\n".join(gone)

