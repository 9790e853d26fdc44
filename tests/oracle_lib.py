"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "liboracle.so")

P2R, R2P, SP2R, SR2P = 0, 1, 2, 3
MAX_STAGES = 64


class OrcConfig(C.Structure):
    _fields_ = [
        ("mode", C.c_int),
        ("iw", C.c_int), ("ow", C.c_int), ("nxtra", C.c_int),
        ("ww", C.c_int), ("pw", C.c_int), ("nstages", C.c_int),
        ("angle", C.c_uint32 * MAX_STAGES),
        ("quantization_variance", C.c_double),
        ("phase_variance_rad", C.c_double),
        ("gain", C.c_double),
        ("best_possible_cnr", C.c_double),
        ("clocks_per_output", C.c_int),
    ]


def build():
    src = os.path.join(ORACLE_DIR, "cordic_oracle.c")
    if (not os.path.exists(_SO)
            or os.path.getmtime(_SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"],
                              stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        i32p = C.POINTER(C.c_int32)
        u32p = C.POINTER(C.c_uint32)
        cfgp = C.POINTER(OrcConfig)
        L.orc_config_cli.argtypes = [cfgp] + [C.c_int] * 6
        L.orc_config_core.argtypes = [cfgp] + [C.c_int] * 6
        for name in ("orc_p2r", "orc_seq_p2r", "orc_rotate"):
            getattr(L, name).argtypes = [cfgp, C.c_size_t, i32p, i32p,
                                         C.c_int, u32p, i32p, i32p]
            getattr(L, name).restype = None
        for name in ("orc_r2p", "orc_seq_r2p", "orc_topolar"):
            getattr(L, name).argtypes = [cfgp, C.c_size_t, i32p, i32p,
                                         i32p, u32p]
            getattr(L, name).restype = None
        L.orc_seq_p2r_cycle.argtypes = [cfgp, C.c_int32, C.c_int32,
                                        C.c_uint32, i32p, i32p]
        L.orc_seq_r2p_cycle.argtypes = [cfgp, C.c_int32, C.c_int32,
                                        i32p, u32p]
        L.orc_nco.argtypes = [cfgp, C.c_size_t, C.c_uint32, C.c_uint32,
                              C.c_uint64, C.c_int32, C.c_int32, i32p, i32p]
        L.orc_nco.restype = None
        L.orc_mixer.argtypes = [cfgp, C.c_size_t, C.c_uint32, C.c_uint32,
                              C.c_uint64, i32p, i32p, i32p, i32p]
        L.orc_mixer.restype = None
        L.orc_cordic_gain.restype = C.c_double
        L.orc_cordic_gain.argtypes = [C.c_int]
        L.orc_phase_variance.restype = C.c_double
        L.orc_phase_variance.argtypes = [C.c_int, C.c_int]
        L.orc_transform_quantization_variance.restype = C.c_double
        L.orc_transform_quantization_variance.argtypes = [C.c_int] * 3
        for name in ("orc_calc_stages1", "orc_calc_phase_bits"):
            getattr(L, name).argtypes = [C.c_int]
        L.orc_calc_stages2.argtypes = [C.c_int, C.c_int]
        L.orc_nextlg.argtypes = [C.c_uint]
        L.orc_table_config.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] * 2
        L.orc_table_values.argtypes = [C.c_int, C.c_int, C.c_int, i32p]
        L.orc_table_values.restype = None
        L.orc_table_lookup.argtypes = [C.c_int, C.c_int, C.c_int, i32p,
                                       C.c_size_t, u32p, i32p]
        L.orc_table_lookup.restype = None
        L.orc_throughput.restype = C.c_uint64
        L.orc_throughput.argtypes = [cfgp, C.c_int, C.c_int, C.c_double,
                                     C.c_uint32, C.c_int32, C.c_int32]
        L.orc_digest.restype = C.c_uint64
        L.orc_digest.argtypes = [cfgp, C.c_int, C.c_int, C.c_uint64,
                                 C.c_uint64, C.c_uint32, C.c_uint32,
                                 C.c_int32, C.c_int32, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_double)]
        L.orc_digest_words.restype = C.c_uint64
        L.orc_digest_words.argtypes = [u32p, C.c_size_t, C.c_uint64]
        _lib = L
    return _lib


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def config_cli(mode, iw=-1, ow=-1, xtra=2, pw=-1, nstages=-1):
    """gencordic -t <mode> -i iw -o ow -x xtra -p pw -n nstages"""
    cfg = OrcConfig()
    rc = lib().orc_config_cli(C.byref(cfg), mode, iw, ow, xtra, pw, nstages)
    if rc:
        raise ValueError("orc_config_cli rc=%d" % rc)
    return cfg


def config_core(mode, nstages, iw, ow, nxtra, pw):
    cfg = OrcConfig()
    rc = lib().orc_config_core(C.byref(cfg), mode, nstages, iw, ow, nxtra, pw)
    if rc:
        raise ValueError("orc_config_core rc=%d" % rc)
    return cfg


def rotate(cfg, x, y, phase):
    """p2r (pipelined or sequential per cfg.mode).  x, y: int32 arrays of
    len(phase), or scalars (broadcast)."""
    phase = np.ascontiguousarray(phase, dtype=np.uint32)
    n = phase.size
    if np.isscalar(x) or np.ndim(x) == 0:
        xa = np.array([x], dtype=np.int32)
        ya = np.array([y], dtype=np.int32)
        stride = 0
    else:
        xa = np.ascontiguousarray(x, dtype=np.int32)
        ya = np.ascontiguousarray(y, dtype=np.int32)
        assert xa.size == n and ya.size == n
        stride = 1
    ox = np.empty(n, dtype=np.int32)
    oy = np.empty(n, dtype=np.int32)
    lib().orc_rotate(C.byref(cfg), n, _i32(xa), _i32(ya), stride,
                     _u32(phase), _i32(ox), _i32(oy))
    return ox, oy


def topolar(cfg, x, y):
    xa = np.ascontiguousarray(x, dtype=np.int32)
    ya = np.ascontiguousarray(y, dtype=np.int32)
    n = xa.size
    mag = np.empty(n, dtype=np.int32)
    ph = np.empty(n, dtype=np.uint32)
    lib().orc_topolar(C.byref(cfg), n, _i32(xa), _i32(ya), _i32(mag), _u32(ph))
    return mag, ph


def nco(cfg, n, phase0, fcw, index0, x0, y0):
    ox = np.empty(n, dtype=np.int32)
    oy = np.empty(n, dtype=np.int32)
    lib().orc_nco(C.byref(cfg), n, phase0, fcw, index0, x0, y0,
                  _i32(ox), _i32(oy))
    return ox, oy


def mix(cfg, phase0, fcw, index0, x, y):
    """fused NCO mixer: phase[i] = phase0 + (index0 + i) * fcw on i_phase,
    per-sample x / y on i_xval / i_yval"""
    xa = np.ascontiguousarray(x, dtype=np.int32)
    ya = np.ascontiguousarray(y, dtype=np.int32)
    n = xa.size
    assert ya.size == n
    ox = np.empty(n, dtype=np.int32)
    oy = np.empty(n, dtype=np.int32)
    lib().orc_mixer(C.byref(cfg), n, phase0 & 0xffffffff, fcw & 0xffffffff,
                  index0, _i32(xa), _i32(ya), _i32(ox), _i32(oy))
    return ox, oy


IQ_MULX, IQ_MULY = 0x9E3779B1, 0x85EBCA77      # SURVEY.md 8(d) config 3 ramps


def usable_cpus():
    """Hardware threads this process may use: affinity mask cut down to the
    cgroup CPU quota (the GPU boxes show 256 CPUs and grant 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def job_digest(cfg, kind, start, n, phase0=0, fcw=1, x0=0, y0=0,
               mulx=IQ_MULX, muly=IQ_MULY, threads=None):
    """(digest, seconds): the oracle's outputs for EVERY sample of a synthetic
    job as the device's position-aware digest -- sum of mix(g, out0[g]) +
    mix(g + 2^40, out1[g]) over g in [start, start+n).  kind "p2r" / "nco":
    constant (x0, y0), phase[g] = phase0 + g*fcw; "r2p": the I/Q ramps;
    "p2rxy" / "mix": the I/Q ramps rotated by phase[g] (read from an array,
    resp. generated by the fused accumulator: the same job)."""
    k = {"p2r": 0, "nco": 0, "r2p": 1, "p2rxy": 2, "mix": 2}[kind]
    sec = C.c_double(0.0)
    d = lib().orc_digest(C.byref(cfg), k, threads or usable_cpus(), start, n,
                         phase0 & 0xffffffff, fcw & 0xffffffff, x0, y0,
                         mulx, muly, C.byref(sec))
    return int(d), sec.value


def digest_words(words, index0=0):
    """C twin of gpu_util.cpu_digest (for arrays too large for numpy temps)"""
    w = np.ascontiguousarray(words).view(np.uint32)
    return int(lib().orc_digest_words(_u32(w), w.size, index0))


class SeqRegs(C.Structure):
    _fields_ = [("prex", C.c_int64), ("prey", C.c_int64), ("xv", C.c_int64),
                ("yv", C.c_int64), ("preph", C.c_uint32), ("ph", C.c_uint32),
                ("cangle", C.c_uint32), ("state", C.c_uint32),
                ("idle", C.c_int32), ("pre_valid", C.c_int32),
                ("aux", C.c_int32), ("o_done", C.c_int32),
                ("o_aux", C.c_int32), ("o0", C.c_int32), ("o1", C.c_int32)]


def seq_regs():
    """power-on register file of a sequential core"""
    r = SeqRegs()
    lib().orc_seq_regs_init(C.byref(r))
    return r


def seq_trace(cfg, stb, x, y, phase=None, reset=None, aux=None, regs=None):
    """Register-level model of rtl/seqcordic.v / rtl/seqpolar.v over a whole
    port trace (off-protocol strobes included).  Returns (o0, o1, o_aux,
    o_busy, o_done) per clock; `regs` (SeqRegs) carries the state across
    calls and is updated in place."""
    L = lib()
    L.orc_seq_trace.restype = None
    n = len(stb)
    u8 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint8)
    stb_, rs_, ax_ = u8(stb), u8(reset), u8(aux)
    x_ = np.ascontiguousarray(x, dtype=np.int32)
    y_ = np.ascontiguousarray(y, dtype=np.int32)
    ph_ = (np.zeros(n, dtype=np.uint32) if phase is None else
           np.ascontiguousarray(phase, dtype=np.uint32))
    o0 = np.empty(n, dtype=np.int32)
    o1 = np.empty(n, dtype=np.int32)
    oa = np.empty(n, dtype=np.uint8)
    bs = np.empty(n, dtype=np.uint8)
    dn = np.empty(n, dtype=np.uint8)
    r = regs if regs is not None else seq_regs()
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    L.orc_seq_trace.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 12
    L.orc_seq_trace(C.byref(cfg), n, vp(stb_), vp(rs_), vp(ax_), vp(x_),
                    vp(y_), vp(ph_), vp(o0), vp(o1), vp(oa), vp(bs), vp(dn),
                    C.byref(r))
    return o0, o1, oa, bs, dn


def seq_p2r_cycle(cfg, x, y, phase):
    ox, oy = C.c_int32(), C.c_int32()
    t = lib().orc_seq_p2r_cycle(C.byref(cfg), x, y, phase,
                                C.byref(ox), C.byref(oy))
    return t, ox.value, oy.value


def seq_r2p_cycle(cfg, x, y):
    m, p = C.c_int32(), C.c_uint32()
    t = lib().orc_seq_r2p_cycle(C.byref(cfg), x, y, C.byref(m), C.byref(p))
    return t, m.value, p.value


TBL, QTR = 4, 5


def table_config(kind, iw=-1, ow=-1, pw=-1):
    a, b = C.c_int(), C.c_int()
    rc = lib().orc_table_config(kind, iw, ow, pw, C.byref(a), C.byref(b))
    if rc:
        raise ValueError("orc_table_config rc=%d" % rc)
    return a.value, b.value


def table_values(kind, pw, ow):
    n = (1 << pw) if kind == TBL else (1 << (pw - 2))
    out = np.empty(n, dtype=np.int32)
    lib().orc_table_values(kind, pw, ow, _i32(out))
    return out


def table_lookup(kind, pw, ow, tbl, phase):
    phase = np.ascontiguousarray(phase, dtype=np.uint32)
    out = np.empty(phase.size, dtype=np.int32)
    lib().orc_table_lookup(kind, pw, ow, _i32(np.ascontiguousarray(tbl)),
                           phase.size, _u32(phase), _i32(out))
    return out


# ---- quadratically interpolated sine core (oracle: orc_quad_*)

class OrcQuad(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "pw", "ow", "xtra", "wid", "ww", "lgtbl", "dxbits", "cbits", "lbits",
        "qbits")] + [("scale", C.c_long), ("itbl_err", C.c_double),
                     ("tbl_err", C.c_double), ("spur_db", C.c_double)]


def _quad_protos():
    L = lib()
    if getattr(L, "_quad_ready", False):
        return L
    qp, lp = C.POINTER(OrcQuad), C.POINTER(C.c_long)
    L.orc_quad_cli.argtypes = [qp] + [C.c_int] * 4
    L.orc_quad_core.argtypes = [qp] + [C.c_int] * 3
    L.orc_quad_tables.argtypes = [qp, lp, lp, lp]
    L.orc_quad_lookup.argtypes = [qp, lp, lp, lp, C.c_size_t,
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    L.orc_quad_lookup.restype = None
    L._quad_ready = True
    return L


def quad_cli(iw=-1, ow=-1, xtra=2, pw=-1):
    q = OrcQuad()
    rc = _quad_protos().orc_quad_cli(C.byref(q), iw, ow, xtra, pw)
    if rc:
        raise ValueError("orc_quad_cli rc=%d" % rc)
    return q


def quad_core(pw, ow, nxtra):
    q = OrcQuad()
    rc = _quad_protos().orc_quad_core(C.byref(q), pw, ow, nxtra)
    if rc:
        raise ValueError("orc_quad_core rc=%d" % rc)
    return q


def quad_tables(q):
    n = 1 << q.lgtbl
    t = [np.empty(n, dtype=np.int64) for _ in range(3)]
    lp = C.POINTER(C.c_long)
    rc = _quad_protos().orc_quad_tables(C.byref(q),
                                        *[a.ctypes.data_as(lp) for a in t])
    if rc:
        raise ValueError("orc_quad_tables rc=%d" % rc)
    return t


def quad_lookup(q, tables, phase):
    phase = np.ascontiguousarray(phase, dtype=np.uint32)
    out = np.empty(phase.size, dtype=np.int32)
    lp = C.POINTER(C.c_long)
    t = [np.ascontiguousarray(a, dtype=np.int64) for a in tables]
    _quad_protos().orc_quad_lookup(C.byref(q), *[a.ctypes.data_as(lp) for a in t],
                                   phase.size, _u32(phase), _i32(out))
    return out
