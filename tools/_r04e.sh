mkdir -p gpurun_out/r04e
(time timeout 2400 python -m pytest tests -m gpu -q --durations=12) > gpurun_out/r04e/gputests.log 2>&1; echo rc=$? >> gpurun_out/r04e/gputests.log
for i in 1 2; do
for w in p2rxy nat24 nat32; do for inp in ramp random; do for lib in libcordic_amd.so lib_ab.so; do
CORDIC_AMD_LIB=$PWD/cordic_amd/$lib python bench.py --workload $w --input $inp --steps 100 --warmup 10 --no-cpu-baseline --no-other-paths --no-pmc --no-full-digest 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; p=(r.get('power') or {}).get('sustained') or {}
print('$w $inp $lib', d['config']['kernel'], round(d['value']), round(r['frac'],3), 'sclk', p.get('sclk_mhz_median'), 'W', p.get('socket_w_median'), d['bit_exact_vs_oracle'])" >> gpurun_out/r04e/ab_b96.txt
done; done; done; done
