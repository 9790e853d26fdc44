#!/bin/bash
# energy_probe.sh -- joules per VALU instruction under sustained load: each
# single-opcode variant of tools/sched_probe for 6 s from 8 waves a SIMD, socket
# power and shader clock from hwmon (median of the last 4 s), against idle.
mkdir -p gpurun_out/energy
o=gpurun_out/energy/probe.txt
h=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* | head -1)
med() { sort -n | awk '{a[NR]=$1} END {print a[int((NR+1)/2)]}'; }
sleep 3
idle=$(for i in $(seq 1 20); do cat $h/power1_input; sleep 0.1; done | med)
echo "# idle socket power $((idle/1000000)) W" > $o
echo "# variant: rate held; socket W (median); sclk MHz (median); nJ per wave-instruction over idle" >> $o
for v in onlybitop3 onlyxor onlyashr onlyalign only32 onlymadv onlymads onlymad sample_major nop_32; do
	./tools/sched_probe --sustain $v 6 > gpurun_out/energy/$v.txt &
	pid=$!
	sleep 2
	: > gpurun_out/energy/$v.pw; : > gpurun_out/energy/$v.ck
	while kill -0 $pid 2>/dev/null; do cat $h/power1_input >> gpurun_out/energy/$v.pw; cat $h/freq1_input >> gpurun_out/energy/$v.ck; sleep 0.1; done
	w=$(med < gpurun_out/energy/$v.pw); c=$(med < gpurun_out/energy/$v.ck)
	python3 - "$v" "$w" "$c" "$idle" >> $o <<'PY'
import re, sys
v, w, c, idle = sys.argv[1], float(sys.argv[2]) / 1e6, float(sys.argv[3]) / 1e6, float(sys.argv[4]) / 1e6
t = open("gpurun_out/energy/%s.txt" % v).read()
m = re.search(r"([\d.e+]+) wave-instructions/s/SIMD", t)
rate = float(m.group(1))
simds = 1024
print("%-14s %.4g winstr/s/SIMD (%.2f cyc at the clock held)  %6.0f W  %5.0f MHz  %.3f nJ" % (
    v, rate, c * 1e6 / rate, w, c, (w - idle) / (rate * simds) * 1e9))
PY
	sleep 2
done
cat $o
