cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -C cordic_amd/csrc -j16 BUILD=build_ab OUT=$PWD/cordic_amd/lib_ab.so HIPFLAGS_EXTRA="$AB_FLAGS" > gpurun_out/ab_build.log 2>&1
{
for r in 1 2 3 4; do for lib in libcordic_amd.so lib_ab.so; do for w in cfg2 cfg5; do
  CORDIC_AMD_LIB=$PWD/cordic_amd/$lib python bench.py --workload $w --no-cpu-baseline --no-other-paths 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lib $w', round(d['value']), round(r['frac'],4), round(r['copy_frac'],4), round(r['frac_over_copy'],4), d['bit_exact_vs_oracle'])"
done; done; done
} > gpurun_out/ab.log 2>&1
