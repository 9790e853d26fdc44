#!/usr/bin/env python3
"""Rebuild profiles/pmc_latest.json -- the counters bench.py quotes under
`from_profile` -- from the per-workload summaries of one profiling round:

    python tools/pmc_latest.py profiles/r02

Each profiles/<round>/<workload>/summary.json (tools/profile_workload.sh ->
tools/pmc_summary.py) contributes the kernel that runs the workload; for the
constant-vector workloads the same profile also holds the full-recurrence
kernel of bench.py's comparison leg, stored as <workload>_noseed."""
import json
import os
import sys

rnd = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "pmc_latest.json")
try:
    db = json.load(open(path))
except (OSError, ValueError):
    db = {}
MAIN = {"cfg2": "rotator_seeded", "cfg4": "rotator_seeded",
        "cfg5": "rotator_seeded", "cfg1": "rotator_seeded",
        "cfg3": None, "p2rxy": "rotator_xydir", "ddc": "rotator_xydir",
        "quadtbl": "quad_lookup",
        "nat32": "rotator_seeded", "nat24": "rotator_seeded",
        "nat16": "rotator_seeded", "natr2p24": "topolar_lj"}
for w in sorted(os.listdir(os.path.join(ROOT, rnd))):
    f = os.path.join(ROOT, rnd, w, "summary.json")
    if not os.path.exists(f):
        continue
    ks = json.load(open(f))["kernels"]
    src = "%s/%s/summary.json" % (rnd, w)
    main = MAIN.get(w)
    if w == "cfg3":
        main = "topolar_lj" if "topolar_lj" in ks else "topolar_unrolled"
    if main in ks:
        db[w] = dict(ks[main], kernel=main, source=src)
    if main == "rotator_seeded" and "rotator_unrolled" in ks:
        db[w + "_noseed"] = dict(ks["rotator_unrolled"],
                                 kernel="rotator_unrolled", source=src)
db["_source"] = "profiles/pmc_latest.json (rebuilt from %s by tools/pmc_latest.py)" % rnd
json.dump(db, open(path, "w"), indent=1, sort_keys=True)
print("updated", path, "with", sorted(k for k in db if not k.startswith("_")))
