#!/bin/bash
# scratch: full GPU suite, then bench lines of the seeded workloads
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests.log 2>&1; echo "rc=$?" >> gpurun_out/tests.log
for w in cfg2 cfg4 cfg5 cfg1 cfg2 cfg5; do
  python bench.py --workload $w --steps 50 --no-cpu-baseline --no-other-paths --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$w', round(d['value']), round(r['frac'],3), r.get('copy_frac'), r.get('frac_over_copy'), d.get('bit_exact_vs_oracle'))" >> gpurun_out/ab.log
done
