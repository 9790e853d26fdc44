mkdir -p gpurun_out/r04d
(time timeout 1700 python -m pytest tests -m gpu -x -q --durations=12) > gpurun_out/r04d/gputests.log 2>&1; echo rc=$? >> gpurun_out/r04d/gputests.log
for w in cfg2 cfg4 nat24; do
  timeout 900 bash tools/profile_workload.sh $w --input random > gpurun_out/r04d/prof_${w}_random.log 2>&1
  timeout 900 bash tools/profile_workload.sh $w > gpurun_out/r04d/prof_${w}.log 2>&1
done
timeout 900 bash tools/profile_workload.sh p2rxy > gpurun_out/r04d/prof_p2rxy.log 2>&1
timeout 900 bash tools/profile_workload.sh p2rxy --input random > gpurun_out/r04d/prof_p2rxy_random.log 2>&1
