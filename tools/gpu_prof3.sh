cd $GRAFT_REPO_ROOT
for w in cfg2 cfg4 cfg5 cfg1; do bash tools/profile_r01.sh $w > gpurun_out/prof_$w.log 2>&1; done
for w in cfg2 cfg4 cfg5 cfg1; do python3 - <<PY
import json
d=json.load(open("gpurun_out/prof/$w/summary.json"))
for k,v in d["kernels"].items():
    print("$w", k, round(v.get("hbm_bytes_per_launch",0)/2**30,4), "GiB", round(v.get("kernel_trace_avg_ns",0)/1e6,3), "ms", "VALU/sample", round(v.get("SQ_INSTS_VALU",0)*64/2**30,1))
PY
done
