cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_r01.sh cfg2 > gpurun_out/profile_cfg2.txt 2>&1
bash tools/profile_r01.sh cfg2 --no-seed > gpurun_out/profile_cfg2_noseed.txt 2>&1
bash tools/profile_r01.sh cfg3 > gpurun_out/profile_cfg3.txt 2>&1
tail -45 gpurun_out/profile_cfg2.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
