"""cordic_amd -- MI355X-native CORDIC rotation engine (host-side Python view).

The product is ``libcordic_amd.so``: C++ host code + hand-written gfx950 HIP
kernels behind the C ABI of ``include/cordic_amd.h``.  This package is only a
ctypes view of that ABI for tests, ``bench.py`` and Python callers; it holds no
arithmetic of its own and there is NO CPU fallback -- if the library is
missing, importing the bindings raises.

Reference interfaces mirrored (see include/cordic_amd.h for file:line):
  Config.from_cli(...)   <->  gencordic -t .. -i .. -o .. -p .. -n .. -x ..
  Config.from_core(...)  <->  basiccordic()/topolar()/seqcordic()/seqpolar()
  p2r / p2r_const / nco  <->  Vcordic i_xval,i_yval,i_phase -> o_xval,o_yval
  r2p                    <->  Vtopolar i_xval,i_yval -> o_mag,o_phase
"""
from ._native import (  # noqa: F401
    P2R, R2P, SP2R, SR2P,
    FLAG_FORCE_GENERIC, FLAG_NO_LJ, FLAG_NO_SEED, FLAG_NO_TAILS, FLAG_STATIC_CHUNKS,
    FLAG_UNIT_GAIN,
    ERR_ARGS, ERR_DEVICE, ERR_CONTAINER, ERR_UNSUPPORTED, ERR_MODE,
    Config, CordicError, Plan, Jobset, JOBS_PHASE_ARRAYS, JOBS_NCO, JOBS_R2P, JOBS_P2R_XY, JOBS_MIX, jobset_reap, Group, Arrays, device_count, shard_range, rccl_unique_id, RCCL_ID_BYTES, Table, TBL, QTR, Quad, Stream, Seq, seed_table, dir_table, Quality, fill_circle, last_kernel, KERNEL_GENERIC, KERNEL_UNROLLED, KERNEL_SEEDED, KERNEL_LEFT_JUSTIFIED, KERNEL_DIRECTIONS,
    lib, lib_path,
    p2r, p2r_const, nco, mix, r2p,
    p2r_host, r2p_host, HostArray, host_last_stats, host_release,
    host_set_devices, host_lane_stats,
    fill_phase_ramp, fill_iq_ramp, digest_u32,
)

__all__ = [
    "P2R", "R2P", "SP2R", "SR2P", "Config", "CordicError", "Plan",
    "seed_table", "Quality", "fill_circle", "last_kernel", "KERNEL_GENERIC", "KERNEL_UNROLLED", "KERNEL_SEEDED", "KERNEL_LEFT_JUSTIFIED", "Table", "TBL", "QTR", "Quad", "Stream", "Seq", "lib", "lib_path",
    "p2r", "p2r_const", "nco", "r2p", "p2r_host", "r2p_host",
    "fill_phase_ramp", "fill_iq_ramp", "digest_u32",
]
