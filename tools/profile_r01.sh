# rocprofv3 session: kernel-trace stats + PMC passes (separate runs, as the
# MI355X guide prescribes; never combined with sys/hip tracing).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=${1:-cfg2}
BENCH="python $R/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- $BENCH > $OUT/stats_$W.log 2>&1
BENCH3="python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$W -- $BENCH3 > $OUT/pmc_fetch_$W.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$W -- $BENCH3 > $OUT/pmc_write_$W.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_$W -- $BENCH3 > $OUT/pmc_sq_$W.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2_$W -- $BENCH3 > $OUT/pmc_sq2_$W.log 2>&1
find $OUT -name "*.csv" | head -40
for f in $(find $OUT/stats_$W -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
python3 - <<PY
import csv, glob, collections
for pat in ["pmc_fetch_$W","pmc_write_$W","pmc_sq_$W","pmc_sq2_$W"]:
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % pat, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", f)
        for k, d in acc.items():
            print(k, {c: (len(v), sum(v)/len(v)) for c, v in d.items()})
PY
