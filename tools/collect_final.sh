#!/bin/bash
# collect_final.sh <round> -- copy what tools/final_session.sh left under
# gpurun_out/ into profiles/bench_<round>/ and profiles/<round>/ (the copies the
# documents cite), then regenerate DESIGN.md's kernel table from them.
R=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
mkdir -p profiles/bench_$R profiles/$R
rm -f profiles/bench_$R/*_ramp.json profiles/bench_$R/*_random.json
cp gpurun_out/bench_sweep/*.json gpurun_out/bench_sweep/default.line profiles/bench_$R/
cp gpurun_out/final/default_driver_order.json gpurun_out/final/default_driver_order.line profiles/bench_$R/
cp gpurun_out/final/full_detail.json gpurun_out/final/full.line profiles/bench_$R/
cp gpurun_out/final/bench_8_ranks_one_gpu.json gpurun_out/final/bench_8_ranks_one_gpu.line profiles/$R/
grep -E "passed|failed" gpurun_out/final/gputests.log > profiles/$R/gputests_final.txt
tail -n 3 gpurun_out/final/smoke.log >> profiles/$R/gputests_final.txt
cp gpurun_out/env.log profiles/$R/box_env.txt 2>/dev/null
cp gpurun_out/final/fuzz.txt profiles/$R/fuzz.txt 2>/dev/null
f=$(ls gpurun_out/final/default_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" profiles/$R/default_kernel_stats.csv
cp gpurun_out/final/default_stats_detail.json profiles/$R/default_kernel_stats_detail.json
for w in cfg1 cfg2 cfg3 cfg4 cfg5 ddc p2rxy; do
	mkdir -p profiles/$R/$w
	cp gpurun_out/prof/$w/summary.json gpurun_out/prof/$w/kernel_stats.csv profiles/$R/$w/
done
for w in cfg2 cfg4 p2rxy; do
	mkdir -p profiles/$R/${w}_random
	cp gpurun_out/prof/${w}inputrandom/summary.json gpurun_out/prof/${w}inputrandom/kernel_stats.csv profiles/$R/${w}_random/
done
python tools/pmc_latest.py profiles/$R > /dev/null 2>&1
python tools/design_table.py --write
