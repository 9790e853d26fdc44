"""place_time.py -- how long a group's placement takes (3 x 4 GiB, twice more), and what a
fresh hipMalloc / hipFree of 4 GiB costs (0.2-1 s when the memory is new)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import cordic_amd as ca
torch.cuda.init()
cfg = ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 16)
for rep in range(3):
    t0 = time.perf_counter()
    g = ca.Group(cfg, devices=[0])
    g.fill_phase_ramp(1 << 30, 0)
    g.sync()
    t1 = time.perf_counter()
    print("group + placement of 3 x 4 GiB: %.2f s" % (t1 - t0), g.placement(0))
    t0 = time.perf_counter(); g.close(); print("close %.2f s" % (time.perf_counter() - t0))
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
p = ctypes.c_void_p()
for rep in range(3):
    t0 = time.perf_counter(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(4 << 30)); t1 = time.perf_counter()
    hip.hipFree(p); t2 = time.perf_counter()
    print("hipMalloc 4 GiB %.3f s, hipFree %.3f s" % (t1 - t0, t2 - t1))
