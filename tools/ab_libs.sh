#!/bin/bash
# Same-box A/B of two builds of the library: libcordic_amd.so against
# cordic_amd/lib_ab.so (built beforehand with other HIPFLAGS_EXTRA), alternating,
# ROUNDS times per workload.   bash tools/ab_libs.sh "cfg2 cfg4 cfg5" [ROUNDS]
# AB_LIBS="libcordic_amd.so lib_x.so lib_y.so" compares more than two builds.
cd $GRAFT_REPO_ROOT
WL=${1:-cfg2}; ROUNDS=${2:-3}
for w in $WL; do
	for r in $(seq 1 $ROUNDS); do for lib in ${AB_LIBS:-libcordic_amd.so lib_ab.so}; do
		CORDIC_AMD_LIB=$PWD/cordic_amd/$lib python bench.py --workload $w --steps 200 --warmup 20 \
			--no-cpu-baseline --no-other-paths --no-pmc ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; p=(r.get('power') or {}).get('sustained') or {}
print('$w', '$lib', round(d['value']), round(r['frac'],3), 'copy', round(r.get('copy_frac',0),3), 'sclk', p.get('sclk_mhz_median'), 'W', p.get('socket_w_median'), d['bit_exact_vs_oracle'])"
	done; done
done
