#!/bin/bash
# Same-box A/B of one bench.py flag:  bash tools/ab_flag.sh "<flag>" "cfg2 cfg4" [ROUNDS]
cd $GRAFT_REPO_ROOT
FLAG=$1; WL=${2:-cfg2}; ROUNDS=${3:-3}
for w in $WL; do for r in $(seq 1 $ROUNDS); do for f in "" "$FLAG"; do
python tools/bench_row.py "$w ${AB_ARGS} ${f:-default}" --workload $w --steps 200 --warmup 20 \
	--no-cpu-baseline --no-pmc --no-full-digest ${AB_ARGS} $f
done; done; done
