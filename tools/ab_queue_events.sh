cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-paths --no-pmc"
for i in 1 2 3; do
  for u in 0 1; do
    CORDIC_QUEUE_UNCHECKED=$u $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('unchecked=$u', round(d['value']), round(r['frac'],3), 'copy', round(r.get('copy_frac',0),3), 'kms', round(r['kernel_ms_avg'],4), round(r['kernel_ms_min'],4), 'sclk', r['power']['sustained']['sclk_mhz_median'], r['power']['sustained']['socket_w_median'])"
  done
done
