#!/bin/bash
# N separate processes of the cfg2 line with CORDIC_PLACEMENT_DEBUG=1: what the
# placement probes saw and what the run then delivered.
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-5}); do
CORDIC_PLACEMENT_DEBUG=1 python bench.py --workload cfg2 --no-cpu-baseline --no-other-paths --no-pmc 2> gpurun_out/pr_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; pl=r['placement']
print('run $i', round(d['value']), round(r['frac'],3), 'copy', round(r.get('copy_frac',0),3), 'candidates', pl['candidates'], 'probes', pl['probes'], 'best_ms', round(pl['best_ms'],3), 'worst_ms', round(pl['worst_ms'],3))"
done
