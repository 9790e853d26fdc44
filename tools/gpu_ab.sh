cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in cfg2 cfg3; do for i in ramp random; do python bench.py --workload $w --input $i --no-cpu-baseline --steps 20 > gpurun_out/ab_${w}_$i.json 2>gpurun_out/ab.err; python - <<PY
import json
d=json.load(open("gpurun_out/ab_${w}_$i.json"))
print("$w $i", round(d["value"]), "Msps", d["ms_per_step"], d["roofline"]["frac"], d["bit_exact_vs_oracle"])
PY
done; done
