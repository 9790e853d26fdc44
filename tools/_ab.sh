cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bounds.py tests/test_tools.py -x -q -k "r2p or topolar or polar or cfg3 or pol" 2>&1 | tail -3
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import numpy as np, torch, cordic_amd as ca, oracle_lib as O
from gpu_util import gpu_r2p
rng = np.random.RandomState(7)
bad = 0
for (iw, ow, xtra, pw, ns) in [(24,24,2,-1,20),(24,24,2,-1,16),(24,24,2,-1,18),(24,24,2,-1,24),(13,13,2,-1,-1),(16,16,2,-1,-1),(24,24,2,-1,22),(20,24,1,30,19),(26,26,0,-1,24),(8,8,2,-1,-1),(12,16,3,-1,12),(24,24,2,32,24)]:
    for mode in (ca.R2P, ca.SR2P):
        try:
            cfg = ca.Config.from_cli(mode, iw, ow, xtra, pw, ns); ocfg = O.config_cli(mode, iw, ow, xtra, pw, ns)
        except Exception as e:
            print("skip", mode, iw, ow, xtra, pw, ns, e); continue
        n = (1 << 18) + 3
        lim = 1 << (cfg.iw - 1)
        x = rng.randint(-lim, lim, size=n).astype(np.int32); y = rng.randint(-lim, lim, size=n).astype(np.int32)
        # adversarial: axes, diagonals, extremes, tiny vectors
        sp = np.array([0, 1, -1, lim-1, -lim, lim//2, -lim//2, 2, -2, 3], dtype=np.int32)
        gx, gy = np.meshgrid(sp, sp); x[:100] = gx.ravel(); y[:100] = gy.ravel()
        small = rng.randint(-8, 8, size=(2, 4096)).astype(np.int32); x[100:4196] = small[0]; y[100:4196] = small[1]
        m, p = gpu_r2p(cfg, x, y)
        rm, rp = O.topolar(ocfg, x, y)
        nb = int((m != rm).sum() + (p != rp).sum()); bad += nb
        print(mode, iw, ow, xtra, cfg.ww, cfg.pw, cfg.nstages, cfg.nlive, "wrap", cfg.needs_wrap, "mismatches", nb, flush=True)
print("TOTAL", bad)
PY
for r in 1 2; do python bench.py --workload cfg3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('cfg3', round(d['value']), round(r['frac'],4), round(r['copy_frac'],4), d['bit_exact_vs_oracle'])"; done
} > gpurun_out/ab.log 2>&1
