#!/bin/bash
# throttle_probe.sh <workload> -- what holds the shader clock under 2.4 GHz while
# a VALU-bound kernel runs below the socket power cap?  Violation / throttle
# accumulators of the SMU (amd-smi metric --throttle, gpu_metrics) before and
# while the workload runs for ~25 s.
w=${1:-cfg3}
mkdir -p gpurun_out/throttle
out=gpurun_out/throttle/$w.txt
{ which amd-smi rocm-smi; amd-smi version 2>&1 | head -3; } > $out 2>&1
echo "== idle" >> $out
amd-smi metric -g 0 --throttle >> $out 2>&1 || amd-smi metric --throttle >> $out 2>&1
python bench.py --workload $w --steps 5000 --warmup 5 --no-cpu-baseline --no-other-paths --no-pmc --no-copy-probe --no-power > gpurun_out/throttle/${w}_line.json 2>/dev/null &
pid=$!
sleep 12
for i in 1 2 3; do
	echo "== running, sample $i" >> $out
	amd-smi metric -g 0 --throttle >> $out 2>&1
	amd-smi metric -g 0 --clock --power 2>&1 | head -40 >> $out
	sleep 2
done
wait $pid
echo "== after" >> $out
amd-smi metric -g 0 --throttle >> $out 2>&1
python -c "
import json
d=json.loads(open('gpurun_out/throttle/${w}_line.json').readline()); print('value', round(d['value']), d['ms_per_step'])" >> $out
