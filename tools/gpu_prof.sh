cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_r01.sh cfg2 > gpurun_out/profile_cfg2.txt 2>&1
tail -60 gpurun_out/profile_cfg2.txt
