# final session of round 4: one box, one code state
mkdir -p gpurun_out/r04g
bash tools/gpu_session.sh env sweep > gpurun_out/r04g/session.log 2>&1
rm -rf gpurun_out/prof
for w in cfg2 cfg3 cfg4 cfg5 p2rxy cfg1 nat24 nat32; do timeout 900 bash tools/profile_workload.sh $w > gpurun_out/r04g/prof_$w.log 2>&1; done
for w in cfg2 cfg4 nat24 p2rxy; do timeout 900 bash tools/profile_workload.sh $w --input random > gpurun_out/r04g/prof_${w}_random.log 2>&1; done
