#!/usr/bin/env python3
"""Identity of the code state a bench line was measured on (round 4, VERDICT
item 7: one code state per profile directory).

kernel_sources_sha256: SHA-256 over the device / launch / table-builder sources
with comments and white space removed -- what decides a kernel's speed; a
comment edit does not change it, a code edit does.  lib_sha256: the built
library itself.  git_head: `git rev-parse HEAD` where a .git exists, else what
__graft_entry__.build() recorded in cordic_amd/BUILD_INFO.json (the GPU boxes
get a snapshot without .git)."""
import hashlib
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cordic_amd", "csrc")
KERNEL_SOURCES = ("cordic_device.h", "cordic_xydir.h", "cordic_internal.h",
                  "cordic_launch.h", "cordic_inst_body.h",
                  "cordic_inst_xydir_body.h", "cordic_kernels.hip",
                  "cordic_plan.cpp", "cordic_quality.hip", "cordic_stream.hip")

_COMMENT = re.compile(r"//[^\n]*|/\*.*?\*/", re.S)


def normalized(text):
    return re.sub(r"\s+", " ", _COMMENT.sub(" ", text)).strip()


def kernel_sources_sha256():
    h = hashlib.sha256()
    names = list(KERNEL_SOURCES) + sorted(
        f for f in os.listdir(CSRC) if f.startswith("cordic_inst_")
        and f.endswith(".hip"))
    for name in names:
        with open(os.path.join(CSRC, name), encoding="utf-8") as f:
            h.update(name.encode() + b"\0" + normalized(f.read()).encode() + b"\0")
    return h.hexdigest()


def lib_sha256(path=None):
    path = path or os.environ.get("CORDIC_AMD_LIB") or os.path.join(
        ROOT, "cordic_amd", "libcordic_amd.so")
    h = hashlib.sha256()
    try:
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
    except OSError:
        return None
    return h.hexdigest()


def git_head():
    if os.path.isdir(os.path.join(ROOT, ".git")):
        try:
            head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"],
                                           text=True).strip()
            dirty = bool(subprocess.check_output(
                ["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"],
                text=True).strip())
            return head, dirty
        except (OSError, subprocess.CalledProcessError):
            pass
    try:
        with open(os.path.join(ROOT, "cordic_amd", "BUILD_INFO.json")) as f:
            d = json.load(f)
        return d.get("git_head"), d.get("git_dirty")
    except (OSError, ValueError):
        return None, None


def stamp():
    head, dirty = git_head()
    return {"git_head": head, "git_dirty": dirty,
            "kernel_sources_sha256": kernel_sources_sha256(),
            "lib_sha256": lib_sha256()}


def write_build_info():
    head, dirty = git_head()
    if head is None:
        return      # no .git and no earlier record: nothing to add, nothing lost
    with open(os.path.join(ROOT, "cordic_amd", "BUILD_INFO.json"), "w") as f:
        json.dump({"git_head": head, "git_dirty": dirty,
                   "kernel_sources_sha256": kernel_sources_sha256()}, f)


if __name__ == "__main__":
    print(json.dumps(stamp(), indent=1))
