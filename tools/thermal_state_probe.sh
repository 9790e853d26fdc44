#!/bin/bash
# thermal_state_probe.sh -- does the copy ceiling of a box follow its HBM
# temperature?  bench.py (short) cold, after ~2.5 min of sustained load, and
# after 2 min of idling, with temperature_mem / hotspot from gpu_metrics.
mkdir -p gpurun_out/thermal
o=gpurun_out/thermal/probe.txt
: > $o
temps() { python - <<'PY'
import amdsmi
amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]
m = amdsmi.amdsmi_get_gpu_metrics_info(h)
print("   mem %s C  hotspot %s C  vrsoc %s C  uclk %s  power %s W" % (
    m.get("temperature_mem"), m.get("temperature_hotspot"), m.get("temperature_vrsoc"),
    m.get("current_uclk"), m.get("average_socket_power") or m.get("current_socket_power")))
PY
}
line() { python bench.py --steps 200 --no-cpu-baseline --no-other-paths --no-pmc --no-power 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('   bench: %d Gsample/s  frac %.3f  copy_frac %.3f  written pair best %.3f ms  candidates %d' % (round(d['value']/1e3), r['frac'], r['copy_frac'], r['placement']['written_pair_best_ms'], r['placement']['candidates']))"; }
echo "== cold $(date +%T)" >> $o; temps >> $o; line >> $o; temps >> $o
echo "== sustained load (cfg2, 70000 steps), temperatures every 10 s" >> $o
python bench.py --workload cfg2 --steps 70000 --warmup 5 --no-cpu-baseline --no-other-paths --no-pmc --no-copy-probe --no-power > gpurun_out/thermal/load.json 2>/dev/null &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 10; temps >> $o; done
python -c "
import json
d=json.loads(open('gpurun_out/thermal/load.json').readline()); print('   load run: %d Gsample/s over %d steps' % (round(d['value']/1e3), d['steps']))" >> $o
echo "== right after the load $(date +%T)" >> $o; temps >> $o; line >> $o; temps >> $o
echo "== second short run" >> $o; line >> $o; temps >> $o
sleep 120
echo "== after 2 min idle $(date +%T)" >> $o; temps >> $o; line >> $o; temps >> $o
cat $o
