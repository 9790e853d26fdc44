cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "seed or nco or full_size" 2>&1 | tail -2
for r in 1 2 3; do
  python bench.py --no-cpu-baseline --no-other-paths 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('cfg2', round(d['value']), round(r['frac'],4), round(r['copy_frac'],4), round(r['frac_over_copy'],4), round(r['copy_frac_same_distribution'],4), d['bit_exact_vs_oracle'], d['full_recurrence_kernel']['outputs_identical_to_seeded_kernel'])"
done
python bench.py --workload cfg5 --no-cpu-baseline --no-other-paths 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('cfg5', round(d['value']), round(r['frac'],4), round(r['copy_frac'],4), round(r['frac_over_copy'],4), d['bit_exact_vs_oracle'])"
python bench.py --workload cfg4 --no-cpu-baseline --no-other-paths 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('cfg4', round(d['value']), round(r['frac'],4), round(r['copy_frac'],4), round(r['frac_over_copy'],4), d['bit_exact_vs_oracle'])"
} > gpurun_out/ab.log 2>&1
