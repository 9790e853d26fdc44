# regression + evidence run after the io16 / quadtbl / stream / unit-gain work:
# every bench workload (ramp and random inputs) into gpurun_out/bench_r01c/,
# rocprofv3 stats + PMC for the new kernels.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench_r01c
for w in cfg2 cfg1 cfg3 cfg4 cfg5 cfg5seq p2rxy sintbl qtrtbl qtrtbl16 quadtbl quadtbl24; do for i in ramp random; do
timeout 300 python bench.py --workload $w --input $i --no-cpu-baseline --no-other-paths > gpurun_out/bench_r01c/${w}_$i.json 2> gpurun_out/b.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_r01c/${w}_$i.json"))
    f=d.get("full_recurrence_kernel") or {}
    print("$w $i", round(d["value"]), "Msps", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), d["bit_exact_vs_oracle"], "full:", round(f.get("value_per_gpu",0)))
except Exception as e:
    print("$w $i FAILED", e, open("gpurun_out/b.err").read()[-600:])
PY
done; done




