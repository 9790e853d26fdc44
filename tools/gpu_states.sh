# What distinguishes the fast and the slow HBM state?  Log temperatures, clocks
# and power next to the 1R2W copy rate while the memory system is kept busy.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
snap() { rocm-smi --showtemp --showclocks --showpower --showperflevel 2>/dev/null | grep -E "Temperature|sclk|mclk|fclk|socclk|Power|Perf" | sed 's/  */ /g' | tr '\n' ';' ; echo; }
echo "== idle"; snap
for i in $(seq 1 14); do
  echo "== round $i: $(./tools/hbm_layout_probe | head -1 | sed 's/  */ /g')"
  python bench.py --steps 300 --no-cpu-baseline --no-other-paths | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   bench 300 steps', round(d['value']), round(d['roofline']['frac'],3))"
  snap
done
