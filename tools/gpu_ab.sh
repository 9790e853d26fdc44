# same-box interleaved A/B of library variants: tools/gpu_ab.sh "<flagsA>" "<flagsB>" workload...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FA="$1"; FB="$2"; shift 2
make -C cordic_amd/csrc -j64 BUILD=build_a OUT=$PWD/cordic_amd/lib_a.so CXXFLAGS_EXTRA="$FA" > gpurun_out/build_a.log 2>&1 || tail -5 gpurun_out/build_a.log
make -C cordic_amd/csrc -j64 BUILD=build_b OUT=$PWD/cordic_amd/lib_b.so CXXFLAGS_EXTRA="$FB" > gpurun_out/build_b.log 2>&1 || tail -5 gpurun_out/build_b.log
for rep in 1 2 3; do for v in a b; do for w in "$@"; do
CORDIC_AMD_LIB=$PWD/cordic_amd/lib_$v.so timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b.json"))
    print("rep$rep $v $w", round(d["value"]), "Msps", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), d["bit_exact_vs_oracle"])
except Exception as e:
    print("rep$rep $v $w FAILED", e, open("gpurun_out/b.err").read()[-600:])
PY
done; done; done
