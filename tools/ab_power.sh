for lib in libcordic_amd.so lib_ab.so libcordic_amd.so lib_ab.so; do
CORDIC_AMD_LIB=$PWD/cordic_amd/$lib python bench.py --workload cfg3 --no-cpu-baseline --no-other-paths --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; p=r['power']
print('$lib', round(d['value']), round(r['frac'],3), 'timed', p['timed_region']['socket_w_median'], p['timed_region']['sclk_mhz_median'], 'sustained', p['sustained']['socket_w_median'], p['sustained']['sclk_mhz_median'], round(p['sustained']['msamples_per_s_local_shards']), p.get('nj_per_sample'))"
done
