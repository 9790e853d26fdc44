# same-box A/B of the in-tree library against cordic_amd/lib_old.so (a build of
# an earlier revision): WORKLOADS="cfg2 cfg5" bash tools/gpu_ab_old.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/hbm_layout_probe | head -1
for rep in 1 2 3; do for v in new old; do for w in ${WORKLOADS:-cfg2}; do
if [ $v = old ]; then export CORDIC_AMD_LIB=$PWD/cordic_amd/lib_old.so; else unset CORDIC_AMD_LIB; fi
timeout 300 python bench.py --workload $w $BENCHFLAGS --no-cpu-baseline --no-other-paths > gpurun_out/b.json 2> gpurun_out/b.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b.json"))
    print("rep$rep $v $w", round(d["value"]), "Msps", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), d["bit_exact_vs_oracle"])
except Exception as e:
    print("rep$rep $v $w FAILED", e, open("gpurun_out/b.err").read()[-600:])
PY
done; done; done
