#!/bin/bash
# gpu_session.sh -- the one script behind every `gpurun` call of this repo.
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <task> [<task> ...]'
# Each task appends to gpurun_out/<task>.log; summaries worth keeping are
# copied by hand into profiles/.  Tasks:
#   env        toolchain probe (verilator / iverilog / yosys), rocm-smi clocks
#   tests      pytest -m gpu
#   smoke      __graft_entry__.smoke()
#   bench      python bench.py (default line)           [BENCH_ARGS=...]
#   hbm        tools/hbm_probe2 streaming-pattern sweep  [HBM_ARGS=...]
#   valu       tools/valu_microbench
#   prof:<w>   rocprofv3 stats + PMC passes of bench.py --workload <w>
#   sweep      every bench.py workload x {ramp, random} + the default line
#   states     alternate copy probe / bench while logging clocks and power
#   ab:<flags> build a second library with HIPFLAGS_EXTRA=<flags> and A/B it
#   abrun      A/B against a prebuilt cordic_amd/lib_ab.so  [AB_WORKLOADS=...]
#   power      rocm-smi power / sclk sampled WHILE each workload runs 6000 steps
#              [POWER_WORKLOADS="cfg5 cfg2 ..."]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
snap() { rocm-smi --showtemp --showclocks --showpower --showperflevel 2>/dev/null \
	| grep -E "Temperature|clk|Power|Perf" | sed 's/  */ /g' | tr '\n' ';'; echo; }
for task in "$@"; do
	name=${task%%:*}; arg=${task#*:}; [ "$arg" = "$task" ] && arg=
	log=gpurun_out/$name.log
	echo "=== $task $(date +%T)" | tee -a $log
	case $name in
	env)
		{ for t in verilator iverilog vvp yosys ghdl go javac node; do
			printf '%-10s %s\n' $t "$(command -v $t || echo absent)"; done
		  nproc; grep -m1 'model name' /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null
		  rocm-smi --showclocks --showpower --showtemp --showperflevel 2>&1 | grep -v '^$'
		  rocminfo 2>/dev/null | grep -E 'Marketing|Compute Unit|Max Clock|gfx' | head -12; } >> $log 2>&1 ;;
	tests)	timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" >> $log; echo "rc=${PIPESTATUS[0]}" >> $log ;;
	smoke)	timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> $log 2>&1; echo "rc=$?" >> $log ;;
	bench)	timeout 900 python bench.py ${BENCH_ARGS} >> $log 2>&1; echo "rc=$?" >> $log ;;
	hbm)	snap >> $log; timeout 600 ./tools/hbm_probe2 ${HBM_ARGS} >> $log 2>&1; snap >> $log ;;
	valu)	timeout 600 ./tools/valu_microbench >> $log 2>&1 ;;
	prof)	timeout 1200 bash tools/profile_workload.sh $arg >> $log 2>&1 ;;
	states)
		for i in $(seq 1 ${STATE_ROUNDS:-10}); do
			echo "== round $i: $(./tools/hbm_probe2 30 20 marker | grep -m1 persist)" >> $log
			python bench.py --steps 300 --no-cpu-baseline --no-other-paths --no-pmc --no-power 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   bench 300 steps', round(d['value']), round(d['roofline']['frac'],3), d['roofline'].get('copy_frac'))" >> $log
			snap >> $log
		done ;;
	power)	# is a kernel power-limited?  sample while it runs (steady state: ~8 s in)
		for w in ${POWER_WORKLOADS:-cfg5 cfg2 cfg4 cfg1 cfg3 p2rxy quadtbl}; do
			echo "== $w" >> $log
			python bench.py --workload $w --steps 6000 --warmup 5 --no-cpu-baseline --no-other-paths \
				--no-pmc --no-copy-probe --no-power > gpurun_out/power_$w.json 2>/dev/null &
			pid=$!
			sleep 8
			for i in 1 2 3 4; do snap >> $log; sleep 0.5; done
			wait $pid
			python -c "
import json
d=json.loads(open('gpurun_out/power_$w.json').readline()); print('   value', round(d['value']), round(d['roofline']['frac'],3))" >> $log
		done ;;
	ab)
		make -C cordic_amd/csrc -j8 BUILD=build_ab OUT=$PWD/cordic_amd/lib_ab.so HIPFLAGS_EXTRA="$arg" > gpurun_out/ab_build.log 2>&1
		for r in 1 2 3; do for lib in libcordic_amd.so lib_ab.so; do
			CORDIC_AMD_LIB=$PWD/cordic_amd/$lib python bench.py ${BENCH_ARGS} --no-cpu-baseline --no-other-paths --no-pmc --no-power 2>/dev/null \
			| python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', round(d['value']), round(d['roofline']['frac'],3))" >> $log
		done; done ;;
	abrun)	# the same A/B against a cordic_amd/lib_ab.so built beforehand (here:
		# make -C cordic_amd/csrc BUILD=build_ab OUT=.../lib_ab.so HIPFLAGS_EXTRA=...;
		# take lib_ab.so out of .gpurunignore for the call)  [AB_WORKLOADS=...]
		for w in ${AB_WORKLOADS:-cfg3}; do for r in 1 2 3; do for lib in libcordic_amd.so lib_ab.so; do
			CORDIC_AMD_LIB=$PWD/cordic_amd/$lib python bench.py --workload $w ${BENCH_ARGS} --no-cpu-baseline --no-other-paths --no-pmc --no-power 2>/dev/null \
			| python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$w $lib', round(d['value']), round(d['roofline']['frac'],3), d['bit_exact_vs_oracle'])" >> $log
		done; done; done ;;
	sweep)	# one bench.py line per workload x {ramp, random}; the ramp lines
		# carry this run's SQ_INSTS_VALU pass (instr/sample); every line is
		# stamped with the code state (build.kernel_sources_sha256)
		mkdir -p gpurun_out/bench_sweep
		for w in ${SWEEP_WORKLOADS:-cfg2 cfg1 cfg3 cfg4 cfg5 cfg5seq p2rxy ddc nat32 nat24 nat16 natr2p24 sintbl qtrtbl qtrtbl16 qtrtbl24 quadtbl quadtbl24}; do
			python bench.py --workload $w --input ramp --no-cpu-baseline --no-other-paths \
				--pmc-counters SQ_INSTS_VALU+SQ_INSTS_VALU_INT64 > gpurun_out/bench_sweep/${w}_ramp.json 2>> $log
			python bench.py --workload $w --input random --no-cpu-baseline --no-other-paths --no-pmc \
				> gpurun_out/bench_sweep/${w}_random.json 2>> $log
		done
		python bench.py > gpurun_out/bench_sweep/default.json 2>> $log ;;
	*)	echo "unknown task $task" | tee -a $log ;;
	esac
done
