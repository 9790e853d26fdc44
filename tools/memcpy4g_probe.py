#!/usr/bin/env python3
"""Does a device-to-device hipMemcpy of >= 2^32 bytes copy everything?
(round 4: the 3-rank shim gather of 4 GiB pieces mismatched while 512 MiB
pieces and hipMemcpyPeerAsync of 4 GiB pieces were fine)"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cordic_amd as ca
from gpu_util import gpu_digest

hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpyPeerAsync.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
D2D = 3
n = (1 << 30) + (1 << 20)
src = torch.empty(n, dtype=torch.int32, device="cuda")
dst = torch.empty(n, dtype=torch.int32, device="cuda")
ca.fill_phase_ramp(src, 12345, 0)
torch.cuda.synchronize()
for words in (1 << 29, (1 << 30) - 1024, 1 << 30, (1 << 30) + 4096):
    want = gpu_digest(src, 0, n=words)
    for name, fn in (
            ("hipMemcpy", lambda b: hip.hipMemcpy(dst.data_ptr(), src.data_ptr(), b, D2D)),
            ("hipMemcpyAsync(null)", lambda b: hip.hipMemcpyAsync(dst.data_ptr(), src.data_ptr(), b, D2D, None)),
            ("hipMemcpyPeerAsync", lambda b: hip.hipMemcpyPeerAsync(dst.data_ptr(), 0, src.data_ptr(), 0, b, None))):
        dst.zero_()
        torch.cuda.synchronize()
        rc = fn(words * 4)
        torch.cuda.synchronize()
        got = gpu_digest(dst, 0, n=words)
        tail = int(dst[words - 1].item()) == int(src[words - 1].item())
        print("%-22s %11d bytes rc=%d %s last word %s" % (
            name, words * 4, rc, "OK" if got == want else "MISMATCH", tail))
