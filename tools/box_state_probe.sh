#!/bin/bash
# box_state_probe.sh -- what differs between a box whose 1R2W copy reaches 0.86 of
# the HBM peak and one where it reaches 0.77?  Static facts (partition modes,
# VBIOS, memory vendor), then HBM / hotspot temperatures and memory / fabric
# clocks once a second while `python bench.py` runs, and the line's copy_frac.
mkdir -p gpurun_out/boxstate
o=gpurun_out/boxstate/probe.txt
{
echo "== static"
rocm-smi --showmemorypartition --showcomputepartition --showmemvendor --showvbios 2>&1 | grep -v "^$" | grep -v "====" 
amd-smi static -g 0 --vram --partition --limit 2>&1 | grep -v "^$" | head -60
echo "== idle"
amd-smi metric -g 0 --temperature --clock 2>&1 | grep -E "HOTSPOT|MEM|EDGE|CLK:|MEM_|FCLK|SOC|_0:|VCLK|DCLK" | head -40
} > $o 2>&1
python bench.py --no-cpu-baseline --no-other-paths --no-pmc --steps 400 > gpurun_out/boxstate/line.json 2>/dev/null &
pid=$!
i=0
while kill -0 $pid 2>/dev/null; do
	i=$((i+1))
	echo "== t=$i s" >> $o
	amd-smi metric -g 0 --temperature --clock --power 2>&1 | grep -E "SOCKET_POWER|HOTSPOT|TEMPERATURE_MEM|MEM:|EDGE|MEM_0|FCLK_0|SOC_0" -A1 | grep -E "SOCKET_POWER|HOTSPOT|MEM|EDGE|CLK:" | tr '\n' ';' >> $o
	echo >> $o
	python - >> $o 2>/dev/null <<'PY'
import amdsmi
amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]
m = amdsmi.amdsmi_get_gpu_metrics_info(h)
keys = ("temperature_hotspot", "temperature_mem", "temperature_vrsoc", "current_uclk",
        "current_socclk", "average_umc_activity", "average_gfx_activity", "current_gfxclk",
        "indep_throttle_status", "throttle_status", "mem_activity_acc", "pcie_bandwidth_inst")
print("   gpu_metrics:", {k: m.get(k) for k in keys})
PY
	sleep 1
done
python -c "
import json
d=json.loads(open('gpurun_out/boxstate/line.json').readline()); r=d['roofline']
print('== line: value', round(d['value']), 'frac', round(r['frac'],3), 'copy_frac', r.get('copy_frac'), 'written pair best', r['placement']['written_pair_best_ms'], 'candidates', r['placement']['candidates'])
print('   copy before', r['copy']['before']); print('   copy after', r['copy']['after'])" >> $o
cat $o
