# GPU session script (round 1): parity tests, then one bench line per workload
# (seeded and full-recurrence, ramp and random inputs).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1
tail -4 gpurun_out/pytest_gpu.txt
for w in cfg2 cfg3 cfg4 cfg5 cfg5seq; do for s in "" "--no-seed"; do for i in ramp random; do
if [ "$w" = "cfg3" ] && [ -n "$s" ]; then continue; fi
tag=${w}${s:+_noseed}_$i
timeout 300 python bench.py --workload $w $s --input $i --no-cpu-baseline > gpurun_out/bench/$tag.json 2> gpurun_out/bench/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench/$tag.json"))
    print("$tag", round(d["value"]), "Msps", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), d["bit_exact_vs_oracle"], d["config"]["kernel"])
except Exception as e:
    print("$tag FAILED", e, open("gpurun_out/bench/$tag.err").read()[-500:])
PY
done; done; done
