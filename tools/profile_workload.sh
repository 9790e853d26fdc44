# rocprofv3 session for one bench.py workload: kernel-trace stats, then PMC
# passes in SEPARATE runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC
# is never combined with sys/hip tracing).
#   bash tools/profile_workload.sh <workload> [extra bench.py flags] 
R=$GRAFT_REPO_ROOT
W=${1:-cfg2}; shift
EXTRA="$@"
TAG=$W$(echo "$EXTRA" | tr -d ' -')
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (the stats pass keeps its detail record: the HIP-event time of the same run,
# and the stamp of the code state, go into summary.json beside the trace's)
BENCH="python $R/bench.py --workload $W $EXTRA --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-power --no-full-digest --detail $OUT/bench_detail.json"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
BENCH3="python $R/bench.py --workload $W $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-power --no-full-digest --detail /dev/null"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH3 > $OUT/pmc_write.log 2>&1
# (10 steps here: the shader clock is read off the last dispatch of this pass)
BENCH10="python $R/bench.py --workload $W $EXTRA --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-power --no-full-digest --detail /dev/null"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH10 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -- $BENCH3 > $OUT/pmc_lds.log 2>&1
# (round 5) the executed 32- / 64-bit integer split: the 64-bit ones are the half-rate v_mad_i64_i32
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/pmc_int -- $BENCH3 > $OUT/pmc_int.log 2>&1
python3 $R/tools/pmc_summary.py $OUT $TAG
# the summaries are what travels back (gpurun merges at most 64 MiB): keep
# summary.json, kernel_stats.csv and the logs, drop the raw per-dispatch files
cp $(ls $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats.csv 2>/dev/null
if [ -s $OUT/summary.json ]; then
	find $OUT -name '*counter_collection.csv' -delete
	find $OUT -name '*kernel_trace.csv' -delete
fi
