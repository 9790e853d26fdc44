# GPU session script (round 1): tests, microbench, bench lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 ./tools/valu_microbench > gpurun_out/valu_microbench.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1
tail -5 gpurun_out/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
cat gpurun_out/bench_cfg2.json
for w in cfg3 cfg4 cfg5 cfg5seq; do timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; cat gpurun_out/bench_$w.json; done
cat gpurun_out/valu_microbench.txt
