"""DESIGN.md section 4.4 against the committed sweep:
the kernel table is GENERATED from profiles/bench_r05/*.json
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
        pytest.skip("profiles/bench_r05/ holds no sweep yet")
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert design_table.BEGIN in text and design_table.END in text
    a = text.index(design_table.BEGIN)
    b = text.index(design_table.END) + len(design_table.END)
    assert text[a:b] == design_table.block(), (
        "DESIGN.md section 4.4 differs from profiles/bench_r05: run "
        "python tools/design_table.py --write")


def test_every_sweep_line_was_measured_on_the_current_kernel_sources():
    lines = design_table.lines()
    if not lines:
        pytest.skip("profiles/bench_r05/ holds no sweep yet")
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
