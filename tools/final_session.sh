#!/bin/bash
# final_session.sh -- the ONE gpurun call whose outputs fill profiles/bench_rNN/
# and profiles/rNN/ (one box, one code state): the driver's order first
# (pytest -m gpu, smoke(), the default bench command: its line AND its detail
# record), the rocprofv3 kernel-trace summary of the default command, the sweep
# (one stamped detail record per workload x {ramp, random}), the per-workload
# profiles (stats + separate PMC passes, ramp and random phases), `bench.py
# --full` (other BASELINE configurations at their sizes, host arrays, small
# batches), the 8-rank line.  Afterwards, here:  bash tools/collect_final.sh rNN
#   gpurun --timeout 5400 -- 'bash tools/final_session.sh'
OUT=gpurun_out/final
mkdir -p $OUT
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $OUT/gputests.log 2>&1; echo rc=$? >> $OUT/gputests.log
(time timeout 600 python -c "import __graft_entry__ as g; g.smoke()") > $OUT/smoke.log 2>&1; echo rc=$? >> $OUT/smoke.log
(time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/default_driver_order.line 2> $OUT/default_driver_order.err
cp bench_detail.json $OUT/default_driver_order.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/default_stats -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --detail $GRAFT_REPO_ROOT/$OUT/default_stats_detail.json > $GRAFT_REPO_ROOT/$OUT/default_stats.log 2>&1
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/bench_sweep gpurun_out/prof
bash tools/gpu_session.sh env sweep > $OUT/session.log 2>&1
for w in cfg2 cfg3 cfg4 cfg5 p2rxy ddc cfg1; do timeout 900 bash tools/profile_workload.sh $w > $OUT/prof_$w.log 2>&1; done
for w in cfg2 cfg4 p2rxy; do timeout 900 bash tools/profile_workload.sh $w --input random > $OUT/prof_${w}_random.log 2>&1; done
(time timeout 1800 python bench.py --full --detail $OUT/full_detail.json) > $OUT/full.line 2> $OUT/full.err
# the driver's multi-rank command, default flags, 8 ranks SHARING this GPU (test
# switch; RCCL entry points from tests/rccl_shim): the line an 8-GPU node prints
BENCH_TEST_SHARE_GPU=1 CORDIC_RCCL_LIB=$PWD/tests/rccl_shim/librccl_shim.so HSA_ENABLE_IPC_MODE_LEGACY=0 \
	timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
	--master-port 29533 bench.py --gpus 8 --workload cfg4 --log2-samples 26 --steps 20 --warmup 5 \
	--detail $OUT/bench_8_ranks_one_gpu.json > $OUT/bench_8_ranks_one_gpu.line 2> $OUT/bench_8_ranks_one_gpu.err
for seed in 11 12 13; do timeout 900 python tools/fuzz_gpu.py 1500 $seed 2>&1 | grep "fuzz ok" >> $OUT/fuzz.txt; done
wc -c $OUT/*.line
