mkdir -p gpurun_out/r04c
(time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=5 -k "plan_p2r or the_fast_paths") > gpurun_out/r04c/tests.log 2>&1; echo rc=$? >> gpurun_out/r04c/tests.log
for i in 1 2 3; do
for f in "" "--no-tails"; do
python bench.py --workload p2rxy --steps 100 --warmup 10 --no-cpu-baseline --no-other-paths --no-pmc --no-full-digest $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; p=(r.get('power') or {}).get('sustained') or {}
print('p2rxy ramp', '$f', d['config']['kernel'], round(d['value']), round(r['frac'],3), 'sclk', p.get('sclk_mhz_median'), 'W', p.get('socket_w_median'), d['bit_exact_vs_oracle'])" >> gpurun_out/r04c/ab_xy.txt
python bench.py --workload p2rxy --input random --steps 100 --warmup 10 --no-cpu-baseline --no-other-paths --no-pmc --no-full-digest $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; p=(r.get('power') or {}).get('sustained') or {}
print('p2rxy random', '$f', d['config']['kernel'], round(d['value']), round(r['frac'],3), 'sclk', p.get('sclk_mhz_median'), 'W', p.get('socket_w_median'), d['bit_exact_vs_oracle'])" >> gpurun_out/r04c/ab_xy.txt
done; done
python bench.py --workload p2rxy --steps 30 --warmup 5 --no-cpu-baseline --no-other-paths --pmc-counters SQ_INSTS_VALU,SQ_INSTS_LDS,SQ_LDS_BANK_CONFLICT > gpurun_out/r04c/p2rxy_pmc.json 2>gpurun_out/r04c/p2rxy_pmc.err
for lg in 22 23 24; do CORDIC_HOST_CHUNK_LOG2=$lg python bench.py --host-paths-only > gpurun_out/r04c/host_paths_chunk$lg.json 2>> gpurun_out/r04c/host.err; done
