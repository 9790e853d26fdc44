"""bench_probes.py -- arithmetic-free twins of a kernel's memory traffic on the
bench's own arrays (tools/libhbmprobe.so): the same-run copy ceiling."""
import os

from bench_common import ROOT

_probe_lib = None


def hbm_probe(in0, in1, out0, out1, nwords, r, w, mode, reps, stream=0):
    """tools/libhbmprobe.so: average ms per launch of an arithmetic-free
    kernel reading r and writing w arrays of nwords 32-bit words -- the same
    traffic as the CORDIC kernel, on the bench's own buffers (which it
    OVERWRITES).  None if the library is not built."""
    global _probe_lib
    import ctypes as C
    if _probe_lib is None:
        path = os.path.join(ROOT, "tools", "libhbmprobe.so")
        if not os.path.exists(path):
            _probe_lib = False
        else:
            _probe_lib = C.CDLL(path)
            _probe_lib.hbm_probe.restype = C.c_float
            _probe_lib.hbm_probe.argtypes = [C.c_void_p] * 4 + [
                C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    if not _probe_lib:
        return None
    ms = _probe_lib.hbm_probe(in0, in1, out0, out1, nwords, r, w, mode, reps,
                              stream)
    return float(ms) if ms > 0 else None


def copy_probe(ptrs, n, rw, reps=10):
    """The copy patterns over the arrays in `ptrs` = [in0, in1, out0, out1]:
    `tiles` = the best streaming pattern found on this chip (one-shot 4 KiB
    tiles), `queued` = the seeded kernel's own work distribution; `_nt` = the
    same with non-temporal loads and stores."""
    r, w = rw
    res = {}
    for name, mode in (("tiles", 0), ("queued", 1), ("tiles_nt", 2),
                       ("queued_nt", 3)):
        ms = hbm_probe(ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, r, w, mode, reps)
        if ms is not None:
            res[name + "_ms"] = ms
    return res
