#!/usr/bin/env python3
"""One-off differential fuzz: random cores x random samples, GPU (plain entry
points and plans) against the oracle.  Not part of the test suite (minutes);
run on a GPU box:  python tools/fuzz_gpu.py [configs] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cordic_amd as ca  # noqa: E402
import oracle_lib as O  # noqa: E402
from gpu_util import gpu_nco, gpu_p2r, gpu_plan_nco, gpu_plan_p2r, gpu_r2p  # noqa: E402
from test_gpu_parity import rand_inputs  # noqa: E402

ncfg = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
done = refused = 0
paths = {}
for t in range(ncfg):
    mode = int(rng.randint(4))
    iw, ow = int(rng.randint(1, 33)), int(rng.randint(1, 33))
    xtra = int(rng.randint(0, 8))
    pw = int(rng.randint(3, 33))
    ns = int(rng.randint(1, 65))
    try:
        cfg = ca.Config.from_cli(mode, iw, ow, xtra, pw, ns)
    except ca.CordicError:
        try:
            O.config_cli(mode, iw, ow, xtra, pw, ns)
            raise SystemExit("oracle accepts what the product refuses: %r"
                             % ((mode, iw, ow, xtra, pw, ns),))
        except ValueError:
            refused += 1
            continue
    ocfg = O.config_cli(mode, iw, ow, xtra, pw, ns)
    n = int(rng.choice([1, 3, 4, 5, 257, 1024, 4099]))
    x, y, ph = rand_inputs(rng, iw, pw, n)
    key = (mode, "wrap" if cfg.needs_wrap else ("wide" if cfg.ww > 35 else
           "lj" if cfg.ww > 32 else "narrow"))
    paths[key] = paths.get(key, 0) + 1
    if mode in (ca.P2R, ca.SP2R):
        a = gpu_p2r(cfg, x, y, ph)
        b = O.rotate(ocfg, x, y, ph)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (t, "p2r")
        x0, y0 = int(x[n // 2]), int(y[0])
        plan = ca.Plan(cfg)
        a = gpu_plan_p2r(plan, x0, y0, ph)
        b = O.rotate(ocfg, x0, y0, ph)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (t, "plan")
        fcw, p0, i0 = int(rng.randint(1 << 32)), int(rng.randint(1 << 32)), int(rng.randint(1 << 40))
        a = gpu_plan_nco(plan, n, p0, fcw, i0, x0, y0)
        b = O.nco(ocfg, n, p0 & 0xffffffff, fcw, i0, x0, y0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (t, "nco")
        plan.close()
    else:
        a = gpu_r2p(cfg, x, y)
        b = O.topolar(ocfg, x, y)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (t, "r2p")
    done += 1
print("fuzz ok: %d cores checked, %d refused by both; paths %s" % (done, refused, paths))
