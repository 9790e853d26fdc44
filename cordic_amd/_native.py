"""ctypes bindings of include/cordic_amd.h (no arithmetic lives here)."""
import ctypes as C
import os

P2R, R2P, SP2R, SR2P = 0, 1, 2, 3
MAX_STAGES = 64
FLAG_FORCE_GENERIC = 0x1
FLAG_NO_LJ = 0x4
FLAG_NO_SEED = 0x8
FLAG_STATIC_CHUNKS = 0x20
FLAG_UNIT_GAIN = 0x10
FLAG_NO_TAILS = 0x40
# enum cordic_status (the codes tests assert on)
ERR_MODE = -1
ERR_UNSUPPORTED = -6
ERR_ARGS, ERR_DEVICE, ERR_CONTAINER = -7, -8, -9

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # CORDIC_AMD_LIB: an alternative build of the same library (A/B work)
    return os.environ.get("CORDIC_AMD_LIB",
                          os.path.join(_HERE, "libcordic_amd.so"))


class CordicError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().cordic_strerror(status).decode()
        super().__init__("%s: %s (status %d)" % (what, msg, status))


class _CConfig(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("iw", C.c_int32), ("ow", C.c_int32),
        ("nextra", C.c_int32), ("ww", C.c_int32), ("pw", C.c_int32),
        ("nstages", C.c_int32), ("clocks_per_output", C.c_int32),
        ("quantization_variance", C.c_double),
        ("phase_variance_rad", C.c_double),
        ("gain", C.c_double), ("best_possible_cnr", C.c_double),
        ("has_reset", C.c_int32), ("has_aux", C.c_int32),
        ("async_reset", C.c_int32),
        ("nlive", C.c_int32), ("needs_wrap", C.c_int32),
        ("flags", C.c_uint32),
        ("angle", C.c_uint32 * MAX_STAGES),
    ]


_lib = None

_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_cfgp = C.POINTER(_CConfig)

# name -> (restype, argtypes): every symbol include/cordic_amd.h declares
ABI = {
    "cordic_abi_version": (C.c_int, []),
    "cordic_strerror": (C.c_char_p, [C.c_int]),
    "cordic_config_init": (C.c_int, [_cfgp] + [C.c_int] * 6),
    "cordic_config_init_core": (C.c_int, [_cfgp] + [C.c_int] * 6),
    "cordic_config_from_args": (C.c_int, [_cfgp, C.c_int,
                                          C.POINTER(C.c_char_p), C.c_char_p,
                                          C.c_size_t, C.POINTER(C.c_int)]),
    "cordic_config_write_header": (C.c_int, [_cfgp, C.c_char_p, C.c_char_p,
                                             C.c_size_t]),
    "cordic_nextlg": (C.c_int, [C.c_uint]),
    "cordic_gain": (C.c_double, [C.c_int]),
    "cordic_phase_variance": (C.c_double, [C.c_int, C.c_int]),
    "cordic_transform_quantization_variance": (C.c_double, [C.c_int] * 3),
    "cordic_angles": (C.c_int, [C.c_int, C.c_int, _u32p]),
    "cordic_calc_stages_ww": (C.c_int, [C.c_int, C.c_int]),
    "cordic_calc_stages": (C.c_int, [C.c_int]),
    "cordic_calc_phase_bits": (C.c_int, [C.c_int]),
    "cordic_p2r": (C.c_int, [_cfgp, C.c_size_t, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cordic_p2r_const": (C.c_int, [_cfgp, C.c_size_t, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "cordic_nco": (C.c_int, [_cfgp, C.c_size_t, C.c_uint32, C.c_uint32,
                             C.c_uint64, C.c_int32, C.c_int32, C.c_void_p,
                             C.c_void_p, C.c_void_p]),
    "cordic_r2p": (C.c_int, [_cfgp, C.c_size_t, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "cordic_p2r16": (C.c_int, [_cfgp, C.c_size_t, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p]),
    "cordic_p2r16_const": (C.c_int, [_cfgp, C.c_size_t, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "cordic_nco16": (C.c_int, [_cfgp, C.c_size_t, C.c_uint32, C.c_uint32,
                               C.c_uint64, C.c_int32, C.c_int32, C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    "cordic_r2p16": (C.c_int, [_cfgp, C.c_size_t, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "cordic_plan_p2r16_const": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "cordic_plan_nco16": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32,
                                    C.c_uint32, C.c_uint64, C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "cordic_plan_create": (C.c_int, [_cfgp, C.POINTER(C.c_void_p)]),
    "cordic_plan_destroy": (None, [C.c_void_p]),
    "cordic_plan_config": (_cfgp, [C.c_void_p]),
    "cordic_plan_seed_info": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p]),
    "cordic_plan_p2r_const": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "cordic_plan_nco": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32,
                                  C.c_uint32, C.c_uint64, C.c_int32,
                                  C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "cordic_seed_table": (C.c_size_t, [_cfgp, _u32p, C.c_size_t]),
    "cordic_dir_table": (C.c_size_t, [_cfgp, _u32p, C.c_size_t]),
    "cordic_plan_tail_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32)]),
    "cordic_plan_queue_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cordic_plan_prepare": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_void_p]),
    "cordic_host_set_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "cordic_host_lane_stats": (C.c_int, [C.c_int, C.c_void_p]),
    "cordic_mix": (C.c_int, [_cfgp, C.c_size_t, C.c_uint32, C.c_uint32,
                             C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p]),
    "cordic_plan_mix": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32,
                                  C.c_uint32, C.c_uint64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "cordic_jobset_create": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t,
                                       C.c_void_p, C.POINTER(C.c_void_p)]),
    "cordic_jobset_destroy": (None, [C.c_void_p]),
    "cordic_jobset_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32)]),
    "cordic_plan_run_jobs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p]),
    "cordic_plan_p2r_const_batch": (C.c_int, [C.c_void_p, C.c_size_t,
                                              C.c_void_p, C.c_int32, C.c_int32,
                                              C.c_void_p]),
    "cordic_plan_nco_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_int32, C.c_int32, C.c_void_p]),
    "cordic_plan_r2p_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_void_p]),
    "cordic_plan_p2r_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_void_p]),
    "cordic_plan_mix_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_void_p]),
    "cordic_jobset_reap": (None, []),
    "cordic_plan_image_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64)]),
    "cordic_plan_set_min_samples": (C.c_int, [C.c_void_p, C.c_longlong]),
    "cordic_plan_p2r": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "cordic_plan_dir_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32)]),
    "cordic_table_queue_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cordic_quad_queue_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cordic_device_count": (C.c_int, []),
    "cordic_shard_range": (C.c_int, [C.c_uint64, C.c_int, C.c_int,
                                     C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint64)]),
    "cordic_group_create": (C.c_int, [_cfgp, C.c_int, C.POINTER(C.c_int),
                                      C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]),
    "cordic_group_destroy": (None, [C.c_void_p]),
    "cordic_group_size": (C.c_int, [C.c_void_p]),
    "cordic_group_range": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int,
                                     C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint64)]),
    "cordic_group_reserve": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int]),
    "cordic_group_set_placement": (C.c_int, [C.c_void_p, C.c_int]),
    "cordic_arrays_alloc": (C.c_int, [C.c_size_t, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p), C.c_void_p]),
    "cordic_arrays_free": (None, [C.POINTER(C.c_void_p), C.c_int]),
    "cordic_group_placement": (C.c_int, [C.c_void_p, C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
                               + [C.POINTER(C.c_float)] * 4),
    "cordic_group_fill_phase_ramp": (C.c_int, [C.c_void_p, C.c_uint64,
                                               C.c_int]),
    "cordic_group_fill_iq_ramp": (C.c_int, [C.c_void_p, C.c_uint64,
                                            C.c_uint32, C.c_uint32, C.c_int]),
    "cordic_group_p2r_const": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int32,
                                         C.c_int32]),
    "cordic_group_nco": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32,
                                   C.c_uint32, C.c_int32, C.c_int32]),
    "cordic_group_r2p": (C.c_int, [C.c_void_p, C.c_uint64]),
    "cordic_group_sync": (C.c_int, [C.c_void_p]),
    "cordic_group_digest": (C.c_int, [C.c_void_p, C.c_uint64,
                                      C.POINTER(C.c_uint64)]),
    "cordic_group_set_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_int]),
    "cordic_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "cordic_group_rccl_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cordic_group_set_gather_rccl": (C.c_int, [C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p,
                                               C.c_int]),
    "cordic_group_mark": (C.c_int, [C.c_void_p, C.c_int]),
    "cordic_group_elapsed": (C.c_int, [C.c_void_p, C.c_int, C.c_int,
                                       C.POINTER(C.c_float),
                                       C.POINTER(C.c_float)]),
    "cordic_group_buffers": (C.c_int, [C.c_void_p, C.c_int,
                                       C.POINTER(C.c_int)] +
                             [C.POINTER(C.c_void_p)] * 4 +
                             [C.POINTER(C.c_uint64)]),
    "cordic_group_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint64,
                                    C.c_uint64, C.c_void_p]),
    "cordic_group_write": (C.c_int, [C.c_void_p, C.c_int, C.c_int,
                                     C.c_uint64, C.c_uint64, C.c_void_p]),
    "cordic_gain_annihilator": (C.c_uint32, [C.c_int]),
    "cordic_config_gain_annihilator": (C.c_uint32, [_cfgp]),
    "cordic_stream_create": (C.c_int, [_cfgp, C.POINTER(C.c_void_p)]),
    "cordic_stream_destroy": (None, [C.c_void_p]),
    "cordic_stream_workspace": (C.c_size_t, [C.c_size_t]),
    "cordic_stream_reserve": (C.c_int, [C.c_void_p, C.c_size_t]),
    "cordic_stream_latency": (C.c_int, [C.c_void_p]),
    "cordic_stream_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cordic_stream_ticks": (C.c_int, [C.c_void_p, C.c_size_t] +
                            [C.c_void_p] * 10),
    "cordic_table_lds_mode": (C.c_int, [C.c_void_p]),
    "cordic_seq_create": (C.c_int, [_cfgp, C.POINTER(C.c_void_p)]),
    "cordic_seq_destroy": (None, [C.c_void_p]),
    "cordic_seq_workspace": (C.c_size_t, [C.c_size_t]),
    "cordic_seq_reserve": (C.c_int, [C.c_void_p, C.c_size_t]),
    "cordic_seq_ticks": (C.c_int, [C.c_void_p, C.c_size_t] +
                         [C.c_void_p] * 12),
    "cordic_seq_violations": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cordic_quad_config_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_int]),
    "cordic_quad_config_init_core": (C.c_int, [C.c_void_p, C.c_int, C.c_int,
                                               C.c_int]),
    "cordic_quad_tables": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p,
                                     C.c_size_t]),
    "cordic_quad_write_header": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p,
                                           C.c_size_t]),
    "cordic_quad_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "cordic_quad_destroy": (None, [C.c_void_p]),
    "cordic_quad_lookup": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "cordic_table_config_init": (C.c_int, [C.c_void_p] + [C.c_int] * 4),
    "cordic_table_values": (C.c_int, [C.c_void_p, _i32p, C.c_size_t]),
    "cordic_table_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "cordic_table_destroy": (None, [C.c_void_p]),
    "cordic_table_lookup": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "cordic_p2r_host": (C.c_int, [_cfgp, C.c_size_t, _i32p, _i32p, C.c_int,
                                  _u32p, _i32p, _i32p]),
    "cordic_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "cordic_host_free": (None, [C.c_void_p]),
    "cordic_host_release": (None, []),
    "cordic_host_last_stats": (C.c_int, [C.c_void_p]),
    "cordic_r2p_host": (C.c_int, [_cfgp, C.c_size_t, _i32p, _i32p, _i32p,
                                  _u32p]),
    "cordic_fill_phase_ramp": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64,
                                         C.c_int, C.c_void_p]),
    "cordic_fill_iq_ramp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_uint64, C.c_uint32, C.c_uint32,
                                      C.c_int, C.c_void_p]),
    "cordic_digest_u32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64,
                                    C.c_void_p, C.c_void_p]),
    "cordic_last_kernel": (C.c_int, []),
    "cordic_quality_create": (C.c_int, [_cfgp, C.POINTER(C.c_void_p)]),
    "cordic_quality_destroy": (None, [C.c_void_p]),
    "cordic_quality_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cordic_quality_p2r": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "cordic_quality_nco": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32,
                                     C.c_uint32, C.c_uint64, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "cordic_quality_r2p": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "cordic_quality_p2r_result": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cordic_quality_r2p_result": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cordic_fill_circle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_uint64, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]),
}


def lib():
    """Load libcordic_amd.so; raise (never fall back) if it is absent."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64; whichever copy is loaded
        # first serves the whole process.  Tensors handed to this library are
        # allocated by torch, so torch's runtime must be the one in use: load
        # it before libcordic_amd.so pulls in /opt/rocm's copy (two runtimes
        # in one process make every launch fail).  Pure C++ callers never see
        # this file and simply use the system runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError(
                "%s not built: run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (there is no CPU fallback)" % path)
        L = C.CDLL(path)
        for name, (res, args) in ABI.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise CordicError(rc, what)


class Config:
    """One generated core (cordic_config)."""

    def __init__(self, c):
        self.c = c

    @classmethod
    def from_cli(cls, mode, iw=-1, ow=-1, xtra=2, pw=-1, nstages=-1):
        c = _CConfig()
        _check(lib().cordic_config_init(C.byref(c), mode, iw, ow, xtra, pw,
                                        nstages), "cordic_config_init")
        return cls(c)

    @classmethod
    def from_core(cls, mode, nstages, iw, ow, nxtra, pw):
        c = _CConfig()
        _check(lib().cordic_config_init_core(C.byref(c), mode, nstages, iw,
                                             ow, nxtra, pw),
               "cordic_config_init_core")
        return cls(c)

    @classmethod
    def from_args(cls, args):
        """args: gencordic command line without the program name."""
        if isinstance(args, str):
            args = args.split()
        argv = (C.c_char_p * (len(args) + 1))(
            b"gencordic", *[a.encode() for a in args])
        c = _CConfig()
        fname = C.create_string_buffer(512)
        hdr = C.c_int(0)
        _check(lib().cordic_config_from_args(C.byref(c), len(args) + 1, argv,
                                             fname, 512, C.byref(hdr)),
               "cordic_config_from_args")
        cfg = cls(c)
        cfg.fname = fname.value.decode()
        cfg.c_header = bool(hdr.value)
        return cfg

    def header_text(self, name):
        buf = C.create_string_buffer(4096)
        n = lib().cordic_config_write_header(C.byref(self.c), name.encode(),
                                             buf, 4096)
        if n < 0:
            raise CordicError(n, "cordic_config_write_header")
        return buf.value.decode()

    def __getattr__(self, k):
        return getattr(self.c, k)

    @property
    def angles(self):
        return list(self.c.angle[: self.c.nstages])

    def with_flags(self, flags):
        c = _CConfig()
        C.memmove(C.byref(c), C.byref(self.c), C.sizeof(_CConfig))
        c.flags = flags
        return Config(c)

    @property
    def ref(self):
        return C.byref(self.c)


class Plan:
    """cordic_plan: a generated core bound to the current device."""

    def __init__(self, cfg):
        self.cfg = cfg
        h = C.c_void_p()
        _check(lib().cordic_plan_create(cfg.ref, C.byref(h)),
               "cordic_plan_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().cordic_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def seed_info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().cordic_plan_seed_info(self._h, C.byref(a), C.byref(b),
                                           C.byref(c)),
               "cordic_plan_seed_info")
        return dict(stages=a.value, nleaves=b.value, nbuckets=c.value)

    @property
    def dir_groups(self):
        """stages per looked-up group of the per-sample-vector path
        (cordic_plan_p2r); [] = that feed runs cordic_p2r's kernel"""
        n = C.c_int32()
        st = (C.c_int32 * 5)()
        _check(lib().cordic_plan_dir_info(self._h, C.byref(n), st),
               "cordic_plan_dir_info")
        return [int(st[g]) for g in range(n.value)]

    def p2r(self, x, y, phase, ox, oy, n=None, stream=None):
        """per-sample vectors through the plan (directions looked up)"""
        n = phase.numel() if n is None else n
        _check(lib().cordic_plan_p2r(self._h, n, _ptr(x), _ptr(y), _ptr(phase),
                                     _ptr(ox), _ptr(oy), _stream(stream)),
               "cordic_plan_p2r")

    def mix(self, phase0, fcw, index0, x, y, ox, oy, n=None, stream=None):
        """fused NCO mixer through the plan (directions looked up)"""
        n = x.numel() if n is None else n
        _check(lib().cordic_plan_mix(self._h, n, phase0 & 0xffffffff,
                                     fcw & 0xffffffff, index0, _ptr(x), _ptr(y),
                                     _ptr(ox), _ptr(oy), _stream(stream)),
               "cordic_plan_mix")

    @property
    def queue_info(self):
        """tile-queue ring of the handle (include/cordic_amd.h)"""
        return _queue_info("cordic_plan_queue_info", self._h)

    def prepare(self, x0, y0, stream=None):
        """build the seed image of the constant vector (x0, y0) now"""
        _check(lib().cordic_plan_prepare(self._h, x0, y0, _stream(stream)),
               "cordic_plan_prepare")

    @property
    def image_info(self):
        """seed images the plan holds; launches served from one / not"""
        a, b, c = C.c_int32(), C.c_uint64(), C.c_uint64()
        _check(lib().cordic_plan_image_info(self._h, C.byref(a), C.byref(b),
                                            C.byref(c)),
               "cordic_plan_image_info")
        return dict(held=a.value, hits=b.value, misses=c.value)

    def p2r_const_batch(self, jobs, x0, y0, stream=None):
        """one-shot batch of phase-array jobs (cordic_plan_p2r_const_batch)"""
        _check(lib().cordic_plan_p2r_const_batch(
            self._h, len(jobs), _job_array(jobs), x0, y0, _stream(stream)),
            "cordic_plan_p2r_const_batch")

    def nco_batch(self, jobs, x0, y0, stream=None):
        _check(lib().cordic_plan_nco_batch(
            self._h, len(jobs), _job_array(jobs), x0, y0, _stream(stream)),
            "cordic_plan_nco_batch")

    def xy_batch(self, kind, jobs, stream=None):
        """one-shot batch of data-fed jobs (cordic_plan_r2p_batch / _p2r_batch
        / _mix_batch by kind)"""
        name = {JOBS_R2P: "cordic_plan_r2p_batch",
                JOBS_P2R_XY: "cordic_plan_p2r_batch",
                JOBS_MIX: "cordic_plan_mix_batch"}[kind]
        _check(getattr(lib(), name)(self._h, len(jobs), _job_array(jobs),
                                    _stream(stream)), name)

    def set_min_samples(self, n):
        """batch size from which the table-driven kernels serve (< 0: default)"""
        _check(lib().cordic_plan_set_min_samples(self._h, n),
               "cordic_plan_set_min_samples")

    @property
    def tail_groups(self):
        """stages per looked-up group behind the seeds ([] = no direction tails)"""
        n = C.c_int32()
        st = (C.c_int32 * 4)()
        _check(lib().cordic_plan_tail_info(self._h, C.byref(n), st),
               "cordic_plan_tail_info")
        return [int(st[g]) for g in range(n.value)]

    def p2r_const(self, x0, y0, phase, ox, oy, n=None, stream=None):
        n = phase.numel() if n is None else n
        if _is16(ox):
            _same16("cordic_plan_p2r16_const", phase, ox, oy)
            _check(lib().cordic_plan_p2r16_const(
                self._h, n, x0, y0, _ptr(phase), _ptr(ox), _ptr(oy),
                _stream(stream)), "cordic_plan_p2r16_const")
            return
        _check(lib().cordic_plan_p2r_const(self._h, n, x0, y0, _ptr(phase),
                                           _ptr(ox), _ptr(oy),
                                           _stream(stream)),
               "cordic_plan_p2r_const")

    def nco(self, n, phase0, fcw, index0, x0, y0, ox, oy, stream=None):
        if _is16(ox):
            _same16("cordic_plan_nco16", ox, oy)
            _check(lib().cordic_plan_nco16(
                self._h, n, phase0 & 0xffffffff, fcw & 0xffffffff, index0,
                x0, y0, _ptr(ox), _ptr(oy), _stream(stream)),
                "cordic_plan_nco16")
            return
        _check(lib().cordic_plan_nco(self._h, n, phase0 & 0xffffffff,
                                     fcw & 0xffffffff, index0, x0, y0,
                                     _ptr(ox), _ptr(oy), _stream(stream)),
               "cordic_plan_nco")


class _CJob(C.Structure):
    _fields_ = [("d_phase", C.c_void_p), ("phase0", C.c_uint32),
                ("fcw", C.c_uint32), ("index0", C.c_uint64),
                ("d_oxval", C.c_void_p), ("d_oyval", C.c_void_p),
                ("n", C.c_uint64),
                ("d_xval", C.c_void_p), ("d_yval", C.c_void_p)]


JOBS_PHASE_ARRAYS, JOBS_NCO, JOBS_R2P, JOBS_P2R_XY, JOBS_MIX = 0, 1, 2, 3, 4


def _job_array(jobs):
    """jobs: dicts with phase (tensor or None), ox, oy, n; for NCO / MIX jobs
    phase0, fcw, index0; for the data-fed kinds x, y (R2P: ox = o_mag, oy =
    o_phase) -> a C array of cordic_job"""
    arr = (_CJob * max(1, len(jobs)))()
    for k, jb in enumerate(jobs):
        ph = jb.get("phase")
        arr[k].d_phase = _ptr(ph) if ph is not None else None
        arr[k].phase0 = jb.get("phase0", 0) & 0xffffffff
        arr[k].fcw = jb.get("fcw", 0) & 0xffffffff
        arr[k].index0 = jb.get("index0", 0)
        arr[k].d_oxval = _ptr(jb["ox"])
        arr[k].d_oyval = _ptr(jb["oy"])
        arr[k].n = jb["n"]
        for key, field in (("x", "d_xval"), ("y", "d_yval")):
            v = jb.get(key)
            setattr(arr[k], field, _ptr(v) if v is not None else None)
    return arr


class Jobset:
    """cordic_jobset: many small jobs cut into tiles once, run in one launch."""

    def __init__(self, plan, kind, jobs):
        self.plan = plan
        self._keep = jobs          # the tensors behind the addresses
        arr = _job_array(jobs)
        h = C.c_void_p()
        _check(lib().cordic_jobset_create(plan._h, kind, len(jobs), arr,
                                          C.byref(h)), "cordic_jobset_create")
        self._h = h

    @property
    def info(self):
        a, b, c = C.c_uint64(), C.c_uint32(), C.c_uint32()
        _check(lib().cordic_jobset_info(self._h, C.byref(a), C.byref(b),
                                        C.byref(c)), "cordic_jobset_info")
        return dict(samples=a.value, tiles=b.value, tail_samples=c.value)

    def run(self, x0=0, y0=0, stream=None, plan=None):
        _check(lib().cordic_plan_run_jobs((plan or self.plan)._h, self._h, x0,
                                          y0, _stream(stream)),
               "cordic_plan_run_jobs")

    def close(self):
        if getattr(self, "_h", None):
            lib().cordic_jobset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def jobset_reap():
    lib().cordic_jobset_reap()


class Group:
    """cordic_group: the local shards of a multi-GPU job (include/
    cordic_amd.h, "multi-GPU jobs").  devices=None: ordinals 0..nlocal-1."""

    IN0, IN1, OUT0, OUT1 = 0, 1, 2, 3

    def __init__(self, cfg, nlocal=1, devices=None, first_shard=0,
                 total_shards=None):
        self.cfg = cfg
        dv = None
        if devices is not None:
            nlocal = len(devices)
            dv = (C.c_int * nlocal)(*devices)
        total = nlocal if total_shards is None else total_shards
        h = C.c_void_p()
        _check(lib().cordic_group_create(cfg.ref, nlocal, dv, first_shard,
                                         total, C.byref(h)),
               "cordic_group_create")
        self._h = h
        self.nlocal, self.first, self.total = nlocal, first_shard, total

    def close(self):
        if getattr(self, "_h", None):
            lib().cordic_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def range(self, n_total, shard):
        a, c = C.c_uint64(), C.c_uint64()
        _check(lib().cordic_group_range(self._h, n_total, shard, C.byref(a),
                                        C.byref(c)), "cordic_group_range")
        return a.value, c.value

    def reserve(self, n_total, inputs):
        _check(lib().cordic_group_reserve(self._h, n_total, inputs),
               "cordic_group_reserve")

    def fill_phase_ramp(self, n_total, shift):
        _check(lib().cordic_group_fill_phase_ramp(self._h, n_total, shift),
               "cordic_group_fill_phase_ramp")

    def fill_iq_ramp(self, n_total, mulx, muly, bits):
        _check(lib().cordic_group_fill_iq_ramp(self._h, n_total, mulx, muly,
                                               bits),
               "cordic_group_fill_iq_ramp")

    def p2r_const(self, n_total, x0, y0):
        _check(lib().cordic_group_p2r_const(self._h, n_total, x0, y0),
               "cordic_group_p2r_const")

    def nco(self, n_total, phase0, fcw, x0, y0):
        _check(lib().cordic_group_nco(self._h, n_total, phase0 & 0xffffffff,
                                      fcw & 0xffffffff, x0, y0),
               "cordic_group_nco")

    def r2p(self, n_total):
        _check(lib().cordic_group_r2p(self._h, n_total), "cordic_group_r2p")

    def sync(self):
        _check(lib().cordic_group_sync(self._h), "cordic_group_sync")

    def digest(self, n_total):
        d = C.c_uint64()
        _check(lib().cordic_group_digest(self._h, n_total, C.byref(d)),
               "cordic_group_digest")
        return d.value

    def set_gather(self, root_device, out0=None, out1=None, chunks=8):
        """out0 / out1: device tensors or raw device addresses."""
        def addr(t):
            return t if isinstance(t, int) or t is None else _ptr(t)
        _check(lib().cordic_group_set_gather(self._h, root_device, addr(out0),
                                             addr(out1), chunks),
               "cordic_group_set_gather")

    def set_placement(self, enable):
        """False / 0: off (the default); True / 1: two spare arrays; N >= 2:
        up to N spares while no written pair is fast (include/cordic_amd.h)"""
        _check(lib().cordic_group_set_placement(self._h, int(enable)),
               "cordic_group_set_placement")

    def placement(self, local_shard=0):
        """What the last allocation of the shard's arrays saw."""
        c, p = C.c_int(), C.c_int()
        wb, ww, b, w = (C.c_float() for _ in range(4))
        _check(lib().cordic_group_placement(self._h, local_shard, C.byref(c),
                                            C.byref(p), C.byref(wb),
                                            C.byref(ww), C.byref(b),
                                            C.byref(w)),
               "cordic_group_placement")
        return {"candidates": c.value, "probes": p.value,
                "written_pair_best_ms": wb.value,
                "written_pair_worst_ms": ww.value,
                "best_ms": b.value, "worst_ms": w.value}

    def rccl_init(self, unique_id):
        """Join the job's RCCL communicator (collective over all shards);
        unique_id: the 128 bytes of rccl_unique_id() from one process."""
        if len(unique_id) != RCCL_ID_BYTES:
            raise ValueError("an RCCL unique id is %d bytes" % RCCL_ID_BYTES)
        buf = C.create_string_buffer(bytes(unique_id), RCCL_ID_BYTES)
        _check(lib().cordic_group_rccl_init(self._h, buf),
               "cordic_group_rccl_init")

    def set_gather_rccl(self, root_shard, out0=None, out1=None, chunks=8):
        """out0 / out1 (root's process only): device tensors or addresses."""
        def addr(t):
            return t if isinstance(t, int) or t is None else _ptr(t)
        _check(lib().cordic_group_set_gather_rccl(self._h, root_shard,
                                                  addr(out0), addr(out1),
                                                  chunks),
               "cordic_group_set_gather_rccl")

    def mark(self, slot):
        _check(lib().cordic_group_mark(self._h, slot), "cordic_group_mark")

    def elapsed(self, a, b):
        """(max over the local shards, [per shard]) in ms."""
        mx = C.c_float()
        per = (C.c_float * self.nlocal)()
        _check(lib().cordic_group_elapsed(self._h, a, b, C.byref(mx), per),
               "cordic_group_elapsed")
        return mx.value, list(per)

    def buffers(self, local_shard):
        dev = C.c_int()
        p = [C.c_void_p() for _ in range(4)]
        cap = C.c_uint64()
        _check(lib().cordic_group_buffers(self._h, local_shard, C.byref(dev),
                                          *[C.byref(q) for q in p],
                                          C.byref(cap)),
               "cordic_group_buffers")
        return dev.value, [q.value for q in p], cap.value

    def read(self, local_shard, array, offset, count, dtype=None):
        import numpy as np
        out = np.empty(count, dtype=np.int32 if dtype is None else dtype)
        _check(lib().cordic_group_read(self._h, local_shard, array, offset,
                                       count, out.ctypes.data), "cordic_group_read")
        return out

    def read_into(self, local_shard, array, offset, dst):
        """dst: device tensor (or numpy array) of 32-bit words to fill."""
        if hasattr(dst, "ctypes"):
            addr, count = dst.ctypes.data, dst.size
        else:
            addr, count = _ptr(dst), dst.numel()
        _check(lib().cordic_group_read(self._h, local_shard, array, offset,
                                       count, addr), "cordic_group_read")

    def write(self, local_shard, array, offset, src):
        """src: numpy array (host) or device tensor of 32-bit words."""
        if hasattr(src, "ctypes"):
            addr, count = src.ctypes.data, src.size
        else:
            addr, count = _ptr(src), src.numel()
        _check(lib().cordic_group_write(self._h, local_shard, array, offset,
                                        count, addr), "cordic_group_write")


class Arrays:
    """cordic_arrays_alloc: n_read + n_write device arrays of `nbytes` bytes
    (plain hipMalloc; placed by measurement with CORDIC_GROUP_PLACEMENT=1 in
    the environment: include/cordic_amd.h, "Placement"), as torch tensor views;
    read arrays first."""

    class _View:
        def __init__(self, ptr, count, typestr):
            self.__cuda_array_interface__ = {
                "shape": (count,), "typestr": typestr, "data": (ptr, False),
                "version": 2}

    def __init__(self, nbytes, n_read, n_write, stream=None):
        self.nbytes, self.count = int(nbytes), n_read + n_write
        self._p = (C.c_void_p * self.count)()
        _check(lib().cordic_arrays_alloc(self.nbytes, n_read, n_write, self._p,
                                         stream), "cordic_arrays_alloc")
        self.ptrs = [int(self._p[i]) for i in range(self.count)]

    def tensor(self, k, dtype):
        import torch
        code = {torch.int32: "<i4", torch.int16: "<i2"}[dtype]
        per = 4 if dtype == torch.int32 else 2
        return torch.as_tensor(Arrays._View(self.ptrs[k], self.nbytes // per, code),
                               device="cuda")

    def close(self):
        if getattr(self, "_p", None) is not None:
            lib().cordic_arrays_free(self._p, self.count)
            self._p = None

    def __del__(self):
        self.close()


RCCL_ID_BYTES = 128


def rccl_unique_id():
    """128 bytes for Group.rccl_init, from ONE process of the job."""
    buf = C.create_string_buffer(RCCL_ID_BYTES)
    _check(lib().cordic_rccl_unique_id(buf), "cordic_rccl_unique_id")
    return buf.raw


def device_count():
    return lib().cordic_device_count()


def shard_range(n_total, shard, total_shards):
    a, c = C.c_uint64(), C.c_uint64()
    _check(lib().cordic_shard_range(n_total, shard, total_shards, C.byref(a),
                                    C.byref(c)), "cordic_shard_range")
    return a.value, c.value


TBL, QTR = 4, 5


class _CTableConfig(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pw", C.c_int32), ("ow", C.c_int32),
                ("entries", C.c_int32)]


class _CQueueInfo(C.Structure):
    _fields_ = [("eager_slots", C.c_int32), ("captured_capacity", C.c_int32),
                ("captured_used", C.c_int32), ("fallback_launches", C.c_uint64)]


def _queue_info(fn, handle):
    q = _CQueueInfo()
    _check(getattr(lib(), fn)(handle, C.byref(q)), fn)
    return {k: int(getattr(q, k)) for k, _ in _CQueueInfo._fields_}


class Table:
    """A -t tbl / -t qtr core: cordic_table_config + its device table."""

    def __init__(self, kind, iw=-1, ow=-1, pw=-1, device=True):
        self.c = _CTableConfig()
        _check(lib().cordic_table_config_init(C.byref(self.c), kind, iw, ow,
                                              pw), "cordic_table_config_init")
        self._h = None
        if device:
            h = C.c_void_p()
            _check(lib().cordic_table_create(C.byref(self.c), C.byref(h)),
                   "cordic_table_create")
            self._h = h

    pw = property(lambda self: self.c.pw)
    ow = property(lambda self: self.c.ow)
    entries = property(lambda self: self.c.entries)

    def values(self):
        import numpy as np
        out = np.empty(self.c.entries, dtype=np.int32)
        _check(lib().cordic_table_values(C.byref(self.c),
                                         out.ctypes.data_as(_i32p), out.size),
               "cordic_table_values")
        return out

    @property
    def lds_mode(self):
        return lib().cordic_table_lds_mode(self._h)

    @property
    def queue_info(self):
        return _queue_info("cordic_table_queue_info", self._h)

    def lookup(self, phase, val, n=None, stream=None):
        n = phase.numel() if n is None else n
        _check(lib().cordic_table_lookup(self._h, n, _ptr(phase), _ptr(val),
                                         _stream(stream)),
               "cordic_table_lookup")

    def close(self):
        if self._h:
            lib().cordic_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _CQuadConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "pw", "ow", "xtra", "tbl_width", "ww", "lgtbl", "entries", "dxbits",
        "cbits", "lbits", "qbits", "has_reset", "has_aux")] + [
        ("scale", C.c_int64), ("itbl_err", C.c_double),
        ("tbl_err", C.c_double), ("spur_db", C.c_double)]


class Quad:
    """A -t qtbl core (quadratically interpolated sine): cordic_quad_config +
    its device tables.  from_cli mirrors gencordic's flags; from_core the
    emitter's (phase_bits, ow, nxtra) tuple."""

    def __init__(self, iw=-1, ow=-1, xtra=2, pw=-1, device=True, core=None):
        self.c = _CQuadConfig()
        if core is not None:
            _check(lib().cordic_quad_config_init_core(C.byref(self.c), *core),
                   "cordic_quad_config_init_core")
        else:
            _check(lib().cordic_quad_config_init(C.byref(self.c), iw, ow, xtra,
                                                 pw), "cordic_quad_config_init")
        self._h = None
        if device:
            h = C.c_void_p()
            _check(lib().cordic_quad_create(C.byref(self.c), C.byref(h)),
                   "cordic_quad_create")
            self._h = h

    def __getattr__(self, name):
        if name in ("c", "_h"):
            raise AttributeError(name)
        return getattr(self.c, name)

    def tables(self):
        import numpy as np
        out = [np.empty(self.c.entries, dtype=np.int32) for _ in range(3)]
        _check(lib().cordic_quad_tables(
            C.byref(self.c), *[o.ctypes.data_as(_i32p) for o in out],
            self.c.entries), "cordic_quad_tables")
        return out

    def header(self, name="quadtbl"):
        buf = C.create_string_buffer(4096)
        n = lib().cordic_quad_write_header(C.byref(self.c), name.encode(), buf,
                                           4096)
        if n < 0:
            raise CordicError(n, "cordic_quad_write_header")
        return buf.value.decode()

    @property
    def queue_info(self):
        return _queue_info("cordic_quad_queue_info", self._h)

    def lookup(self, phase, val, n=None, stream=None):
        n = phase.numel() if n is None else n
        _check(lib().cordic_quad_lookup(self._h, n, _ptr(phase), _ptr(val),
                                        _stream(stream)), "cordic_quad_lookup")

    def close(self):
        if self._h:
            lib().cordic_quad_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """cordic_stream: a pipelined core (p2r / r2p) stepped in blocks of clocks
    with per-clock i_ce / i_reset / i_aux, pipeline state carried between
    calls (include/cordic_amd.h, "clocked view")."""

    def __init__(self, cfg):
        self.cfg = cfg
        h = C.c_void_p()
        _check(lib().cordic_stream_create(cfg.ref, C.byref(h)),
               "cordic_stream_create")
        self._h = h

    @property
    def latency(self):
        return lib().cordic_stream_latency(self._h)

    def reserve(self, ticks):
        _check(lib().cordic_stream_reserve(self._h, ticks),
               "cordic_stream_reserve")

    def reset(self, stream=None):
        _check(lib().cordic_stream_reset(self._h, _stream(stream)),
               "cordic_stream_reset")

    def ticks(self, x, y, phase, out0, out1, oaux=None, ce=None, reset=None,
              aux=None, n=None, stream=None):
        n = x.numel() if n is None else n
        _check(lib().cordic_stream_ticks(
            self._h, n, _ptr(ce), _ptr(reset), _ptr(aux), _ptr(x), _ptr(y),
            _ptr(phase), _ptr(out0), _ptr(out1), _ptr(oaux), _stream(stream)),
            "cordic_stream_ticks")

    def close(self):
        if self._h:
            lib().cordic_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Seq:
    """cordic_seq: a sequential core (sp2r / sr2p) behind its i_stb / o_busy /
    o_done handshake, stepped in blocks of clocks."""

    def __init__(self, cfg):
        self.cfg = cfg
        h = C.c_void_p()
        _check(lib().cordic_seq_create(cfg.ref, C.byref(h)), "cordic_seq_create")
        self._h = h

    def ticks(self, stb, x, y, phase, out0, out1, busy=None, done=None,
              oaux=None, reset=None, aux=None, n=None, stream=None):
        n = stb.numel() if n is None else n
        _check(lib().cordic_seq_ticks(
            self._h, n, _ptr(stb), _ptr(reset), _ptr(aux), _ptr(x), _ptr(y),
            _ptr(phase), _ptr(out0), _ptr(out1), _ptr(busy), _ptr(done),
            _ptr(oaux), _stream(stream)), "cordic_seq_ticks")

    @property
    def violations(self):
        v = C.c_uint64()
        _check(lib().cordic_seq_violations(self._h, C.byref(v)),
               "cordic_seq_violations")
        return v.value

    def close(self):
        if self._h:
            lib().cordic_seq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _CP2RQuality(C.Structure):
    _fields_ = ([("n", C.c_uint64)]
                + [(k, C.c_double) for k in (
                    "avg_err", "max_err", "mag", "input_mag", "alpha",
                    "cnr_db", "expected_err", "avg_limit", "max_limit")]
                + [("max_err_index", C.c_uint64)]
                + [(k, C.c_int32) for k in (
                    "pass_avg", "pass_max", "pass_alpha", "pass")]
                + [(k, C.c_double) for k in (
                    "sum_err2", "sum_xy", "sum_sq", "sum_d", "sum_in2")])


class _CR2PQuality(C.Structure):
    _fields_ = ([("n", C.c_uint64)]
                + [(k, C.c_double) for k in (
                    "max_phase_err", "max_mag_err", "avg_phase_err",
                    "avg_mag_err", "mean_phase_err",
                    "expected_avg_phase_err", "phase_limit", "mag_limit")]
                + [("max_phase_err_index", C.c_uint64),
                   ("max_mag_err_index", C.c_uint64)]
                + [(k, C.c_int32) for k in (
                    "pass_phase", "pass_mag", "pass")])


def _as_dict(st):
    return {k: getattr(st, k) for k, _ in st._fields_}


class Quality:
    """cordic_quality: the reference benches' statistics and thresholds
    (bench/cpp/cordic_tb.cpp:223-337, topolar_tb.cpp:222-315), reduced on the
    device; calls accumulate until reset()."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._h = C.c_void_p()
        _check(lib().cordic_quality_create(cfg.ref, C.byref(self._h)),
               "cordic_quality_create")

    def reset(self, stream=None):
        _check(lib().cordic_quality_reset(self._h, _stream(stream)),
               "cordic_quality_reset")

    def p2r(self, x, y, phase, ox, oy, n=None, stream=None):
        """x, y: ints (the constant vector) or int32 tensors"""
        n = phase.numel() if n is None else n
        if isinstance(x, int):
            a = (None, None, x, y)
        else:
            a = (_ptr(x), _ptr(y), 0, 0)
        _check(lib().cordic_quality_p2r(self._h, n, a[0], a[1], a[2], a[3],
                                        _ptr(phase), _ptr(ox), _ptr(oy),
                                        _stream(stream)),
               "cordic_quality_p2r")

    def nco(self, n, phase0, fcw, index0, x0, y0, ox, oy, stream=None):
        _check(lib().cordic_quality_nco(self._h, n, phase0 & 0xffffffff,
                                        fcw & 0xffffffff, index0, x0, y0,
                                        _ptr(ox), _ptr(oy), _stream(stream)),
               "cordic_quality_nco")

    def r2p(self, x, y, imag, omag, ophase, n=None, stream=None):
        n = x.numel() if n is None else n
        _check(lib().cordic_quality_r2p(self._h, n, _ptr(x), _ptr(y), imag,
                                        _ptr(omag), _ptr(ophase),
                                        _stream(stream)),
               "cordic_quality_r2p")

    def p2r_result(self):
        r = _CP2RQuality()
        _check(lib().cordic_quality_p2r_result(self._h, C.byref(r)),
               "cordic_quality_p2r_result")
        return _as_dict(r)

    def r2p_result(self):
        r = _CR2PQuality()
        _check(lib().cordic_quality_r2p_result(self._h, C.byref(r)),
               "cordic_quality_r2p_result")
        return _as_dict(r)

    def close(self):
        if self._h:
            lib().cordic_quality_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fill_circle(x, y, index0, lgnsamples, iw, pw, n=None, stream=None):
    n = x.numel() if n is None else n
    _check(lib().cordic_fill_circle(_ptr(x), _ptr(y), n, index0, lgnsamples,
                                    iw, pw, _stream(stream)),
           "cordic_fill_circle")


KERNEL_GENERIC, KERNEL_UNROLLED, KERNEL_SEEDED, KERNEL_LEFT_JUSTIFIED = 1, 2, 3, 4
KERNEL_DIRECTIONS = 5


def last_kernel():
    """enum cordic_kernel_family of this thread's most recent launch"""
    return lib().cordic_last_kernel()


def seed_table(cfg):
    """Host-side seed table words of a core (numpy uint32) or None."""
    import numpy as np
    cap = 4 + 4096 * 6 + 4 + 4 * (6 + 2 * 4096 + 2 * 256)
    buf = np.zeros(cap, dtype=np.uint32)
    n = lib().cordic_seed_table(cfg.ref, buf.ctypes.data_as(_u32p), cap)
    return buf[:n].copy() if n else None


def dir_table(cfg):
    """Host-side direction tables of the per-sample-vector path (numpy
    uint32 words) or None."""
    import numpy as np
    cap = 4 + 5 * (6 + 2 * 4096 + 2 * 256)
    buf = np.zeros(cap, dtype=np.uint32)
    n = lib().cordic_dir_table(cfg.ref, buf.ctypes.data_as(_u32p), cap)
    return buf[:n].copy() if n else None


def _ptr(t):
    """Device pointer of a torch tensor (or an int / None passed through)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _stream(stream):
    if stream is None:
        import torch
        return torch.cuda.current_stream().cuda_stream
    if isinstance(stream, int):
        return stream
    return stream.cuda_stream


def _is16(t):
    """int16 / uint16 sample tensors select the 16-bit-container entry points
    (include/cordic_amd.h: cordic_*16)."""
    return t.element_size() == 2


def _same16(what, *tensors):
    if any(t.element_size() != 2 for t in tensors):
        raise TypeError("%s: every sample array must be 16-bit" % what)


def p2r(cfg, x, y, phase, ox, oy, n=None, stream=None):
    n = phase.numel() if n is None else n
    if _is16(ox):
        _same16("cordic_p2r16", x, y, phase, ox, oy)
        _check(lib().cordic_p2r16(cfg.ref, n, _ptr(x), _ptr(y), _ptr(phase),
                                  _ptr(ox), _ptr(oy), _stream(stream)),
               "cordic_p2r16")
        return
    _check(lib().cordic_p2r(cfg.ref, n, _ptr(x), _ptr(y), _ptr(phase),
                            _ptr(ox), _ptr(oy), _stream(stream)), "cordic_p2r")


def p2r_const(cfg, x0, y0, phase, ox, oy, n=None, stream=None):
    n = phase.numel() if n is None else n
    if _is16(ox):
        _same16("cordic_p2r16_const", phase, ox, oy)
        _check(lib().cordic_p2r16_const(cfg.ref, n, x0, y0, _ptr(phase),
                                        _ptr(ox), _ptr(oy), _stream(stream)),
               "cordic_p2r16_const")
        return
    _check(lib().cordic_p2r_const(cfg.ref, n, x0, y0, _ptr(phase), _ptr(ox),
                                  _ptr(oy), _stream(stream)),
           "cordic_p2r_const")


def nco(cfg, n, phase0, fcw, index0, x0, y0, ox, oy, stream=None):
    if _is16(ox):
        _same16("cordic_nco16", ox, oy)
        _check(lib().cordic_nco16(cfg.ref, n, phase0 & 0xffffffff,
                                  fcw & 0xffffffff, index0, x0, y0, _ptr(ox),
                                  _ptr(oy), _stream(stream)), "cordic_nco16")
        return
    _check(lib().cordic_nco(cfg.ref, n, phase0 & 0xffffffff, fcw & 0xffffffff,
                            index0, x0, y0, _ptr(ox), _ptr(oy),
                            _stream(stream)), "cordic_nco")


def mix(cfg, phase0, fcw, index0, x, y, ox, oy, n=None, stream=None):
    """cordic_mix: per-sample x / y rotated by phase0 + (index0 + i) * fcw"""
    n = x.numel() if n is None else n
    _check(lib().cordic_mix(cfg.ref, n, phase0 & 0xffffffff, fcw & 0xffffffff,
                            index0, _ptr(x), _ptr(y), _ptr(ox), _ptr(oy),
                            _stream(stream)), "cordic_mix")


def r2p(cfg, x, y, mag, ophase, n=None, stream=None):
    n = x.numel() if n is None else n
    if _is16(mag):
        _same16("cordic_r2p16", x, y, mag, ophase)
        _check(lib().cordic_r2p16(cfg.ref, n, _ptr(x), _ptr(y), _ptr(mag),
                                  _ptr(ophase), _stream(stream)),
               "cordic_r2p16")
        return
    _check(lib().cordic_r2p(cfg.ref, n, _ptr(x), _ptr(y), _ptr(mag),
                            _ptr(ophase), _stream(stream)), "cordic_r2p")


class _CHostStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("chunks", C.c_int32),
                ("chunk_samples", C.c_int32), ("staged_inputs", C.c_int32),
                ("staged_outputs", C.c_int32), ("copy_threads", C.c_int32),
                ("seeded_plan", C.c_int32), ("lanes", C.c_int32),
                ("seconds", C.c_double)]


class HostArray:
    """n 32-bit words of PINNED host memory (cordic_host_alloc) as a numpy
    array: what the host-array entry points DMA in place."""

    def __init__(self, n, dtype="int32"):
        import weakref
        import numpy as np
        p = C.c_void_p()
        _check(lib().cordic_host_alloc(C.byref(p), max(1, n) * 4),
               "cordic_host_alloc")
        buf = (C.c_uint32 * max(1, n)).from_address(p.value)
        # the pinned block lives as long as ANY numpy view of it does (views
        # keep `buf` alive through their .base chain): it is released when the
        # ctypes buffer is collected, not when close() is called
        weakref.finalize(buf, lib().cordic_host_free, C.c_void_p(p.value))
        self.array = np.frombuffer(buf, dtype=np.uint32, count=n).view(dtype)

    def close(self):
        """drop this handle's reference (the memory goes once no view of
        `.array` is left)"""
        self.array = None


def host_last_stats():
    st = _CHostStats()
    _check(lib().cordic_host_last_stats(C.byref(st)), "cordic_host_last_stats")
    return {k: getattr(st, k) for k, _ in _CHostStats._fields_}


def host_release():
    lib().cordic_host_release()


def host_set_devices(devices):
    """device ordinals the host-array calls spread over ([] / None: the
    current device only)"""
    devices = list(devices or [])
    arr = (C.c_int * max(1, len(devices)))(*devices)
    _check(lib().cordic_host_set_devices(arr, len(devices)),
           "cordic_host_set_devices")


def host_lane_stats(lane):
    st = _CHostStats()
    _check(lib().cordic_host_lane_stats(lane, C.byref(st)),
           "cordic_host_lane_stats")
    return {k: getattr(st, k) for k, _ in _CHostStats._fields_}


def _host_out(out, n, dtypes, what):
    """caller-provided output arrays of a host-array call: the C pipeline
    writes n contiguous words through the raw pointer, so anything else than a
    writable C-contiguous array of exactly n 32-bit words is refused here"""
    import numpy as np
    if out is None:
        return tuple(np.empty(n, dtype=d) for d in dtypes)
    if len(out) != len(dtypes):
        raise ValueError("%s: out= wants %d arrays" % (what, len(dtypes)))
    for a in out:
        if (not isinstance(a, np.ndarray) or a.dtype.itemsize != 4
                or a.dtype.kind not in "iu" or a.ndim != 1 or a.size != n
                or not a.flags.c_contiguous or not a.flags.writeable):
            raise ValueError("%s: out= arrays must be writable, C-contiguous, "
                             "one-dimensional, 32-bit, %d elements" % (what, n))
    return tuple(out)


def p2r_host(cfg, x, y, phase, out=None):
    """cordic_p2r_host on numpy arrays (x, y scalars: constant vectors).
    out = (ox, oy): write into these int32 arrays (e.g. HostArray.array)."""
    import numpy as np
    phase = np.ascontiguousarray(phase, dtype=np.uint32)
    n = phase.size
    scalar = np.ndim(x) == 0
    if scalar:
        xa = np.array([x], dtype=np.int32)
        ya = np.array([y], dtype=np.int32)
    else:
        xa = np.ascontiguousarray(x, dtype=np.int32)
        ya = np.ascontiguousarray(y, dtype=np.int32)
        if xa.size != n or ya.size != n:
            raise ValueError("cordic_p2r_host: x, y and phase differ in length")
    ox, oy = _host_out(out, n, (np.int32, np.int32), "cordic_p2r_host")
    _check(lib().cordic_p2r_host(
        cfg.ref, n, xa.ctypes.data_as(_i32p), ya.ctypes.data_as(_i32p),
        1 if scalar else 0, phase.ctypes.data_as(_u32p),
        ox.ctypes.data_as(_i32p), oy.ctypes.data_as(_i32p)),
        "cordic_p2r_host")
    return ox, oy


def r2p_host(cfg, x, y, out=None):
    import numpy as np
    xa = np.ascontiguousarray(x, dtype=np.int32)
    ya = np.ascontiguousarray(y, dtype=np.int32)
    n = xa.size
    if ya.size != n:
        raise ValueError("cordic_r2p_host: x and y differ in length")
    mag, ph = _host_out(out, n, (np.int32, np.uint32), "cordic_r2p_host")
    _check(lib().cordic_r2p_host(
        cfg.ref, n, xa.ctypes.data_as(_i32p), ya.ctypes.data_as(_i32p),
        mag.ctypes.data_as(_i32p), ph.ctypes.data_as(_u32p)),
        "cordic_r2p_host")
    return mag, ph


def fill_phase_ramp(phase, index0, shift, n=None, stream=None):
    n = phase.numel() if n is None else n
    _check(lib().cordic_fill_phase_ramp(_ptr(phase), n, index0, shift,
                                        _stream(stream)),
           "cordic_fill_phase_ramp")


def fill_iq_ramp(x, y, index0, mulx, muly, bits, n=None, stream=None):
    n = x.numel() if n is None else n
    _check(lib().cordic_fill_iq_ramp(_ptr(x), _ptr(y), n, index0, mulx, muly,
                                     bits, _stream(stream)),
           "cordic_fill_iq_ramp")


def digest_u32(words, index0, digest, n=None, stream=None):
    """digest (int64 device tensor, 1 element) += digest of words."""
    n = words.numel() if n is None else n
    _check(lib().cordic_digest_u32(_ptr(words), n, index0, _ptr(digest),
                                   _stream(stream)), "cordic_digest_u32")
