#!/bin/bash
# Translation-cache counters of a slow and a fast role assignment of the SAME
# three hipMalloc arrays (tools/hbm_vmm_probe T): separate rocprofv3 --pmc
# passes, counters only (never combined with tracing).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tlb
mkdir -p $OUT
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
	tag=$(echo $set | cut -d' ' -f1)
	rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- $R/tools/hbm_vmm_probe T > $OUT/$tag.log 2>&1
	python3 - "$OUT/$tag" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tiles" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)[-6:]
for k, i in enumerate(ids):
    print("slowest" if k < 3 else "fastest", i, rows[i])
PY
	grep "^T " $OUT/$tag.log | tail -3
done
