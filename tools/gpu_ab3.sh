# same-box interleaved comparison of N library variants: each arg is a flag set
# (WORKLOADS="cfg2 cfg3" BENCHFLAGS="--no-seed" bash tools/gpu_ab3.sh "" "-DX")
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for F in "$@"; do i=$((i+1)); make -C cordic_amd/csrc -j64 BUILD=build_v$i OUT=$PWD/cordic_amd/lib_v$i.so CXXFLAGS_EXTRA="$(echo "$F" | sed 's/-mllvm [^ ]*//g')" HIPFLAGS_EXTRA="$(echo "$F" | grep -o -- '-mllvm [^ ]*' | tr '\n' ' ')" > gpurun_out/build_v$i.log 2>&1 || tail -5 gpurun_out/build_v$i.log; done
N=$i
for rep in 1 2 3; do for v in $(seq 1 $N); do for w in ${WORKLOADS:-cfg2}; do
CORDIC_AMD_LIB=$PWD/cordic_amd/lib_v$v.so timeout 300 python bench.py --workload $w $BENCHFLAGS --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b.json"))
    print("rep$rep v$v $w", round(d["value"]), "Msps", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), d["bit_exact_vs_oracle"])
except Exception as e:
    print("rep$rep v$v $w FAILED", e, open("gpurun_out/b.err").read()[-600:])
PY
done; done; done
