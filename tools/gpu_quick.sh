cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/hbm_pattern_bench | grep -E "1R2W nt +chunked blocks  2048|0R2W nt +chunked blocks  2048|2R2W nt +chunked blocks  2048"
for rep in 1 2; do for w in ${WORKLOADS:-cfg2 cfg5 cfg4}; do
timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b.json"))
    fr=d.get("full_recurrence_kernel") or {}
    print("rep$rep $w", round(d["value"]), "Msps", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), d["bit_exact_vs_oracle"], "full:", round(fr.get("ms_per_step",0),3))
except Exception as e:
    print("rep$rep $w FAILED", e, open("gpurun_out/b.err").read()[-600:])
PY
done; done
