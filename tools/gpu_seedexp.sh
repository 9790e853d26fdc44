cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1
tail -5 gpurun_out/pytest_gpu.txt
for flags in "" "-DCORDIC_PLAIN_STORES"; do
if [ -n "$flags" ]; then make -C cordic_amd/csrc clean > /dev/null; make -C cordic_amd/csrc -j32 CXXFLAGS_EXTRA="$flags" > gpurun_out/build.log 2>&1 || tail -5 gpurun_out/build.log; fi
for w in cfg2 cfg4 cfg3; do for i in ramp; do
timeout 300 python bench.py --workload $w --input $i --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b.json"))
    print("[$flags] $w $i", round(d["value"]), "Msps", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), d["bit_exact_vs_oracle"])
except Exception as e:
    print("[$flags] $w $i FAILED", e, open("gpurun_out/b.err").read()[-800:])
PY
done; done; done
