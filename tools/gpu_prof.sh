cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof
./tools/hbm_pattern_bench > gpurun_out/hbm_pattern.txt 2>&1
./tools/valu_microbench > gpurun_out/valu_microbench.txt 2>&1
bash tools/profile_r01.sh cfg2 > gpurun_out/profile_cfg2.txt 2>&1
bash tools/profile_r01.sh cfg2 --no-seed > gpurun_out/profile_cfg2_noseed.txt 2>&1
bash tools/profile_r01.sh cfg3 > gpurun_out/profile_cfg3.txt 2>&1
bash tools/profile_r01.sh cfg5 > gpurun_out/profile_cfg5.txt 2>&1
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
grep -E "chunked blocks  2048|blocks 65536" gpurun_out/hbm_pattern.txt
