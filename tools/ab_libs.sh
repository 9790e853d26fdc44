#!/bin/bash
# Same-box A/B of two builds of the library: libcordic_amd.so against
# cordic_amd/lib_ab.so (built beforehand with other HIPFLAGS_EXTRA:
#   make -C cordic_amd/csrc BUILD=build_ab OUT=$PWD/cordic_amd/lib_ab.so HIPFLAGS_EXTRA=-D...
# and taken out of .gpurunignore for the call), alternating, ROUNDS times per
# workload.   bash tools/ab_libs.sh "cfg2 cfg4 cfg5" [ROUNDS]
# AB_LIBS="libcordic_amd.so lib_x.so lib_y.so" compares more than two builds;
# AB_ARGS adds bench.py flags (e.g. --pmc-counters SQ_INSTS_VALU instead of the
# default --no-pmc).
cd $GRAFT_REPO_ROOT
WL=${1:-cfg2}; ROUNDS=${2:-3}
for w in $WL; do
	for r in $(seq 1 $ROUNDS); do for lib in ${AB_LIBS:-libcordic_amd.so lib_ab.so}; do
		CORDIC_AMD_LIB=$PWD/cordic_amd/$lib python tools/bench_row.py "$w $lib" --workload $w \
			--steps 200 --warmup 20 --no-cpu-baseline ${AB_PMC:---no-pmc} ${AB_ARGS}
	done; done
done
