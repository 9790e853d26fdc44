#!/bin/bash
# gpu_session.sh -- the one script behind every `gpurun` call of this repo.
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <task> [<task> ...]'
# Each task appends to gpurun_out/<task>.log; summaries worth keeping are
# copied by hand into profiles/.  Tasks:
#   env        toolchain probe (verilator / iverilog / yosys), rocm-smi clocks
#   tests      pytest -m gpu
#   smoke      __graft_entry__.smoke()
#   bench      python bench.py (default line)           [BENCH_ARGS=...]
#   valu       tools/valu_microbench
#   prof:<w>   rocprofv3 stats + PMC passes of bench.py --workload <w>
#   sweep      every bench.py workload x {ramp, random} + the default line
#   ab:<flags> build a second library with HIPFLAGS_EXTRA=<flags> and A/B it
#   abrun      A/B against a prebuilt cordic_amd/lib_ab.so  [AB_WORKLOADS=...]
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
	valu)	timeout 600 ./tools/valu_microbench >> $log 2>&1 ;;
	prof)	timeout 1200 bash tools/profile_workload.sh $arg >> $log 2>&1 ;;
	ab)	# build a second library with HIPFLAGS_EXTRA=<flags> ON THE BOX and A/B it
		make -C cordic_amd/csrc -j8 BUILD=build_ab OUT=$PWD/cordic_amd/lib_ab.so HIPFLAGS_EXTRA="$arg" > gpurun_out/ab_build.log 2>&1
		bash tools/ab_libs.sh "${AB_WORKLOADS:-cfg3}" 3 >> $log 2>&1 ;;
	abrun)	# the same against a cordic_amd/lib_ab.so built beforehand  [AB_WORKLOADS=...]
		bash tools/ab_libs.sh "${AB_WORKLOADS:-cfg3}" 3 >> $log 2>&1 ;;
	sweep)	# one bench.py run per workload x {ramp, random}: the DETAIL record
		# (what the printed line is a selection of) is what is kept; the ramp
		# runs carry this run's SQ_INSTS_VALU pass (instr/sample); every
		# record is stamped with the code state (build.kernel_sources_sha256)
		mkdir -p gpurun_out/bench_sweep
		for w in ${SWEEP_WORKLOADS:-cfg2 cfg1 cfg3 cfg4 cfg5 cfg5seq p2rxy ddc nat32 nat24 nat16 natr2p24 sintbl qtrtbl qtrtbl16 qtrtbl24 quadtbl quadtbl24}; do
			python bench.py --workload $w --input ramp --no-cpu-baseline --copy-probe \
				--detail gpurun_out/bench_sweep/${w}_ramp.json \
				--pmc-counters SQ_INSTS_VALU+SQ_INSTS_VALU_INT64 > gpurun_out/bench_sweep/${w}_ramp.line 2>> $log
			python bench.py --workload $w --input random --no-cpu-baseline --no-pmc --copy-probe \
				--detail gpurun_out/bench_sweep/${w}_random.json \
				> gpurun_out/bench_sweep/${w}_random.line 2>> $log
		done
		python bench.py --detail gpurun_out/bench_sweep/default.json > gpurun_out/bench_sweep/default.line 2>> $log ;;
	*)	echo "unknown task $task" | tee -a $log ;;
	esac
done
