cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for w in ${WORKLOADS:-sintbl qtrtbl cfg2}; do for i in ramp random; do
timeout 300 python bench.py --workload $w --input $i --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b.json"))
    print("$w $i", round(d["value"]), "Msps", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), d["bit_exact_vs_oracle"])
except Exception as e:
    print("$w $i FAILED", e, open("gpurun_out/b.err").read()[-600:])
PY
done; done
