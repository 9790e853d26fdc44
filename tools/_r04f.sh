mkdir -p gpurun_out/r04f
(time timeout 600 python -m pytest tests/test_host_pipeline.py -m gpu -q) > gpurun_out/r04f/hosttests.log 2>&1; echo rc=$? >> gpurun_out/r04f/hosttests.log
for seed in 401 402 403; do (timeout 900 python tools/fuzz_gpu.py 1500 $seed) >> gpurun_out/r04f/fuzz.txt 2>&1; echo "rc=$? seed=$seed" >> gpurun_out/r04f/fuzz.txt; done
python bench.py --host-paths-only > gpurun_out/r04f/host_paths.json 2> gpurun_out/r04f/host_paths.err
