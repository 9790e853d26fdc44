cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/tests_h.log 2>&1; grep -E "passed|failed|error" gpurun_out/tests_h.log | tail -3
for s in 951 952 953; do timeout 400 python tools/fuzz_gpu.py 1500 $s 2>&1 | tail -2 | cut -c1-100; done | tee gpurun_out/fuzz_tails.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_session.sh prof:cfg4 > /dev/null 2>&1
mkdir -p gpurun_out/bench_sweep
for inp in ramp random; do python bench.py --workload cfg4 --input $inp --no-cpu-baseline --no-other-paths > gpurun_out/bench_sweep/cfg4_${inp}.json 2>/dev/null; done
python bench.py > gpurun_out/bench_sweep/default.json 2>/dev/null
python - <<'PY'
import json
for f in ('cfg4_ramp','cfg4_random','default'):
    d=json.loads(open('gpurun_out/bench_sweep/%s.json'%f).readline()); r=d['roofline']
    print(f, round(d['value']), round(r['frac'],3), r.get('bound'), r.get('valu_fraction'), (r.get('valu') or {}).get('instr_per_sample'), r.get('copy_frac'))
    if f=='default':
        for k,o in d.get('other_paths',{}).items(): print('   ', k, round(o.get('Msamples_per_s',0)), o['roofline']['frac'], o['roofline'].get('valu_fraction'))
PY
