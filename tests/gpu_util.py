"""Helpers for the -m gpu tests: numpy <-> device plumbing around the C ABI
and the CPU twin of the device digest."""
import numpy as np
import torch

import cordic_amd as ca

DEV = "cuda:0"


def dev_i32(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(DEV)


def to_np(t, dtype=np.int32):
    return t.cpu().numpy().view(dtype)


def gpu_p2r(cfg, x, y, phase, offset=0):
    """offset > 0 places every buffer `offset` words past a 16-byte boundary
    (exercises the unaligned path)."""
    phase = np.ascontiguousarray(phase, dtype=np.uint32)
    n = phase.size

    def buf(src=None):
        t = torch.zeros(n + offset + 4, dtype=torch.int32, device=DEV)
        v = t[offset:offset + n]
        if src is not None and n:
            v.copy_(dev_i32(src))
        return v
    dph = buf(phase)
    ox, oy = buf(), buf()
    if np.ndim(x) == 0:
        ca.p2r_const(cfg, int(x), int(y), dph, ox, oy, n=n)
    else:
        dx = buf(np.asarray(x, dtype=np.int32))
        dy = buf(np.asarray(y, dtype=np.int32))
        ca.p2r(cfg, dx, dy, dph, ox, oy, n=n)
    torch.cuda.synchronize()
    return to_np(ox), to_np(oy)


def gpu_r2p(cfg, x, y, offset=0):
    x = np.ascontiguousarray(x, dtype=np.int32)
    n = x.size

    def buf(src=None):
        t = torch.zeros(n + offset + 4, dtype=torch.int32, device=DEV)
        v = t[offset:offset + n]
        if src is not None and n:
            v.copy_(dev_i32(src))
        return v
    dx, dy = buf(x), buf(np.asarray(y, dtype=np.int32))
    mag, oph = buf(), buf()
    ca.r2p(cfg, dx, dy, mag, oph, n=n)
    torch.cuda.synchronize()
    return to_np(mag), to_np(oph, np.uint32)


def gpu_nco(cfg, n, phase0, fcw, index0, x0, y0):
    ox = torch.zeros(n, dtype=torch.int32, device=DEV)
    oy = torch.zeros(n, dtype=torch.int32, device=DEV)
    ca.nco(cfg, n, phase0, fcw, index0, x0, y0, ox, oy)
    torch.cuda.synchronize()
    return to_np(ox), to_np(oy)


def cpu_digest(words, index0=0):
    """CPU twin of cordic_digest_u32 (cordic_kernels.hip: digest_mix)."""
    w = np.ascontiguousarray(words).view(np.uint32).astype(np.uint64)
    idx = np.arange(w.size, dtype=np.uint64) + np.uint64(index0)
    with np.errstate(over="ignore"):
        z = (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + w
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return int(z.sum(dtype=np.uint64))


def gpu_digest(t, index0=0, n=None):
    d = torch.zeros(1, dtype=torch.int64, device=DEV)
    ca.digest_u32(t, index0, d, n=n)
    torch.cuda.synchronize()
    return int(d.cpu().numpy().view(np.uint64)[0])


def gpu_plan_p2r(plan, x0, y0, phase):
    phase = np.ascontiguousarray(phase, dtype=np.uint32)
    n = phase.size
    dph = dev_i32(phase)
    ox = torch.zeros(n, dtype=torch.int32, device=DEV)
    oy = torch.zeros(n, dtype=torch.int32, device=DEV)
    plan.p2r_const(int(x0), int(y0), dph, ox, oy, n=n)
    torch.cuda.synchronize()
    return to_np(ox), to_np(oy)


def gpu_plan_nco(plan, n, phase0, fcw, index0, x0, y0):
    ox = torch.zeros(n, dtype=torch.int32, device=DEV)
    oy = torch.zeros(n, dtype=torch.int32, device=DEV)
    plan.nco(n, phase0, fcw, index0, x0, y0, ox, oy)
    torch.cuda.synchronize()
    return to_np(ox), to_np(oy)
