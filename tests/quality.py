"""The reference's own acceptance criteria, restated on arrays.

p2r: bench/cpp/cordic_tb.cpp:127-139 (phase ramp), :223-337 (statistics and
thresholds).  r2p: bench/cpp/topolar_tb.cpp:127-147 (input circle),
:222-256 and :303-315 (statistics and thresholds).  Used to tie the oracle
(and, on the GPU, the engine's output) to what the reference itself tests.
"""
import numpy as np


def p2r_bench_inputs(iw, pw, lgn=None):
    """cordic_tb.cpp:61-69,127-139 with LGNSAMPLES = PW (shift 0)."""
    lgn = pw if lgn is None else lgn
    n = 1 << lgn
    i = np.arange(n, dtype=np.int64)
    phase = ((i << (pw - lgn)) & ((1 << pw) - 1)).astype(np.uint32)
    return phase, (1 << (iw - 1)) - 1, 0


def p2r_quality(cfg, phase, ixv, iyv, ox, oy):
    """Statistics of cordic_tb.cpp:223-337.  cfg needs iw, ow, pw,
    quantization_variance, phase_variance_rad, gain, best_possible_cnr."""
    iw, ow, pw = cfg.iw, cfg.ow, cfg.pw
    gain = cfg.gain
    n = phase.size
    ixv = np.broadcast_to(np.asarray(ixv, dtype=np.float64), (n,))
    iyv = np.broadcast_to(np.asarray(iyv, dtype=np.float64), (n,))
    ph = phase.astype(np.float64) * np.pi * 2.0 / float(1 << pw)
    dx = np.cos(ph) * ixv - np.sin(ph) * iyv
    dy = np.sin(ph) * ixv + np.cos(ph) * iyv
    dx *= gain
    dy *= gain
    shift = iw + 1 - ow
    # cordic_tb.cpp:241-248 (its OW > IW+1 branch divides by zero; use the
    # scaling it means: outputs carry OW-IW-1 extra bits)
    dx *= 2.0 ** (-shift)
    dy *= 2.0 ** (-shift)
    oxf = ox.astype(np.float64)
    oyf = oy.astype(np.float64)
    err2 = (dx - oxf) ** 2 + (dy - oyf) ** 2
    sumxy = np.sum(dx * oxf) + np.sum(dy * oyf)
    sumsq = np.sum(oxf * oxf + oyf * oyf)
    averr = np.sqrt(np.sum(err2) / n)
    mxerr = np.sqrt(err2.max())
    scale = np.sqrt(float(ixv[0]) ** 2 + float(iyv[0]) ** 2)
    expected = (cfg.quantization_variance
                + cfg.phase_variance_rad * scale * scale * gain * gain)
    alpha = sumxy / sumsq
    cnr = 10.0 * np.log10((scale * gain) ** 2 / (averr * averr))
    ok = (averr <= 1.5 * np.sqrt(expected)
          and mxerr <= 5.2 * np.sqrt(expected)
          and abs(alpha - 1.0) <= 0.01)
    return dict(averr=averr, mxerr=mxerr, alpha=alpha, cnr=cnr,
                sigma=np.sqrt(expected), ok=bool(ok))


def sfdr_dbc(ox, oy):
    """cordic_tb.cpp:342-371 (printed there, not asserted): one full turn of
    the phase ramp puts the tone in FFT bin 1."""
    z = ox.astype(np.float64) + 1j * oy.astype(np.float64)
    f = np.abs(np.fft.fft(z)) ** 2
    master = f[1]
    spur = max(f[0], f[2:].max())
    return 10.0 * np.log10(master / spur)


def r2p_bench_inputs(iw, pw):
    """topolar_tb.cpp:127-141 with LGNSAMPLES = PW: two turns of a circle of
    radius 2^(IW-1)-1, components truncated toward zero by (int)."""
    n = 1 << pw
    i = np.arange(n, dtype=np.int64)
    lv = i << 1                      # i << (PW-(LGNSAMPLES-1))
    ip = lv.astype(np.int32).astype(np.int64)   # (int)lv
    ph = ip.astype(np.float64) * np.pi / float(1 << (pw - 1))
    mg = float((1 << (iw - 1)) - 1)
    x = np.trunc(mg * np.cos(ph)).astype(np.int32)
    y = np.trunc(mg * np.sin(ph)).astype(np.int32)
    return x, y, int(mg)


def r2p_quality(cfg, x, y, imag, omag, ophase):
    """topolar_tb.cpp:222-256,303-315."""
    iw, ow, pw = cfg.iw, cfg.ow, cfg.pw
    maxphase = 2.0 ** pw
    rad_to_phase = maxphase / np.pi / 2.0
    dp = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    ep = dp * rad_to_phase
    ep = np.where(ep < 0, ep + maxphase, ep)
    # the bench sign extends o_phase from PW bits (:177-181)
    oph = ophase.astype(np.int64)
    oph = np.where(oph >= (1 << (pw - 1)), oph - (1 << pw), oph)
    dperr = oph.astype(np.float64) - ep
    dperr = np.where(dperr > maxphase / 2, dperr - maxphase, dperr)
    dperr = np.where(dperr < -maxphase / 2, dperr + maxphase, dperr)
    mxperr = np.abs(dperr).max()
    emag = imag * 2.0 ** (iw - 1 - ow)
    mxverr = np.abs(omag.astype(np.float64) - emag * cfg.gain).max()
    exp_ph = np.sqrt(cfg.phase_variance_rad * rad_to_phase * rad_to_phase)
    exp_ph = max(exp_ph, 1.0)
    ok = (mxperr <= 3.4 * exp_ph
          and mxverr <= 2.0 * np.sqrt(cfg.quantization_variance))
    return dict(mxperr=mxperr, mxverr=mxverr, phase_limit=3.4 * exp_ph,
                mag_limit=2.0 * np.sqrt(cfg.quantization_variance),
                ok=bool(ok))
