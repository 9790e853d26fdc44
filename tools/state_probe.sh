#!/bin/bash
# What puts a box into the "slow" memory state?  cfg2 bench (short form) with
# rocm-smi clocks / temperatures sampled DURING the run, on a fresh box, then
# again after the GPU test suite, then after a 60 s pause.
cd $GRAFT_REPO_ROOT
one() {
	python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-other-paths --no-pmc 2>/dev/null > gpurun_out/sp.json &
	pid=$!
	sleep 7
	rocm-smi --showclocks --showtemp --showpower 2>/dev/null | grep -E "mclk|fclk|sclk|socclk|Temperature|Power" | sed 's/  */ /g' | tr '\n' ';'
	echo
	wait $pid
	python -c "
import json
d=json.loads(open('gpurun_out/sp.json').readline()); r=d['roofline']
print('   $1', round(d['value']), round(r['frac'],3), 'copy', round(r.get('copy_frac',0),3), 'place_best', round(r['placement']['best_ms'],3), 'sclk', r['power']['sustained']['sclk_mhz_median'], r['power']['sustained']['socket_w_median'])"
}
one fresh1; one fresh2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | grep -E "passed|failed"
one after_parity_tests1; one after_parity_tests2
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed"
one after_full_suite1; one after_full_suite2
sleep 60
one after_60s_idle
