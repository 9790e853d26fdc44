#!/bin/bash
# sched_probe_run.sh -- tools/sched_probe twice, with the shader clock the
# device reported meanwhile (hwmon freq1_input, every 20 ms)
mkdir -p gpurun_out/sched
f=$(ls /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input 2>/dev/null | head -1)
( while true; do cat $f; sleep 0.02; done ) > gpurun_out/sched/clock.txt 2>/dev/null &
bg=$!
for i in 1; do ./tools/sched_probe; done > gpurun_out/sched/probe.txt 2>&1
kill $bg
sort -n gpurun_out/sched/clock.txt | awk '{a[NR]=$1} END {print "sclk Hz while running: min", a[1], "median", a[int(NR/2)+1], "max", a[NR], "samples", NR}' >> gpurun_out/sched/probe.txt
cat gpurun_out/sched/probe.txt
