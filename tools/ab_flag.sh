#!/bin/bash
# Same-box A/B of one bench.py flag:  bash tools/ab_flag.sh "<flag>" "cfg2 cfg4" [ROUNDS]
cd $GRAFT_REPO_ROOT
FLAG=$1; WL=${2:-cfg2}; ROUNDS=${3:-3}
for w in $WL; do for r in $(seq 1 $ROUNDS); do for f in "" "$FLAG"; do
python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-other-paths --no-pmc ${AB_ARGS} $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; p=(r.get('power') or {}).get('sustained') or {}
print('$w', '${AB_ARGS}', '${f:-default}', round(d['value']), round(r['frac'],3), 'copy', round(r.get('copy_frac',0),3), 'sclk', p.get('sclk_mhz_median'), 'W', p.get('socket_w_median'), d['bit_exact_vs_oracle'])"
done; done; done
