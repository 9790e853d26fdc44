mkdir -p gpurun_out/r04h
(time timeout 600 python -m pytest tests/test_host_pipeline.py -m gpu -q) > gpurun_out/r04h/hosttests.log 2>&1; echo rc=$? >> gpurun_out/r04h/hosttests.log
for t in 4 6 8 12; do for plain in 0 1; do
CORDIC_HOST_THREADS=$t CORDIC_HOST_PLAIN_MEMCPY=$plain python bench.py --host-paths-only 2>/dev/null | python -c "
import json,sys
h=json.loads(sys.stdin.readline())
print('threads $t plain_memcpy $plain', {k:(round(v['Msamples_per_s']), round(v['frac_of_slower_pcie_direction'],3)) for k,v in h.items() if isinstance(v,dict) and 'Msamples_per_s' in v})" >> gpurun_out/r04h/host_threads.txt
done; done
