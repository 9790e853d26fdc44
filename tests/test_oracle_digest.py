"""oracle/cordic_oracle.c: orc_digest -- the oracle's outputs for every sample
of a synthetic job condensed to the device's position-aware digest.  Checked
here (CPU) against the sample-level oracle entry points and the numpy twin of
the device digest, at ragged sizes, across the 2^32 index wrap, with 1 and
several threads."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_util import cpu_digest


def _inputs(start, n, iw):
    g = (np.arange(n, dtype=np.uint64) + np.uint64(start)).astype(np.uint32)
    sh = 32 - iw

    def ramp(mul):
        with np.errstate(over="ignore"):
            v = (g * np.uint32(mul)) >> np.uint32(8)
            return (v << np.uint32(sh)).astype(np.uint32).view(np.int32) >> sh
    return g, ramp(O.IQ_MULX), ramp(O.IQ_MULY)


@pytest.mark.parametrize("start,n", [(0, 1), (5, 70001), ((1 << 32) - 777, 140000),
                                     (7 << 30, 65536)])
@pytest.mark.parametrize("threads", [1, 3])
def test_rotator_job_digest(start, n, threads):
    for mode, ns, fcw, ph0 in ((O.P2R, 16, 4, 0), (O.P2R, 24, 1, 0),
                               (O.SP2R, 16, 0x01234567, 99)):
        cfg = O.config_cli(mode, 32, 32, 2, 32, ns)
        g, _, _ = _inputs(start, n, 32)
        with np.errstate(over="ignore"):
            ph = np.uint32(ph0) + g * np.uint32(fcw)
        ox, oy = O.rotate(cfg, 2**31 - 1, 0, ph)
        want = (cpu_digest(ox, start) + cpu_digest(oy, start + (1 << 40))) % 2**64
        got, _ = O.job_digest(cfg, "p2r", start, n, ph0, fcw, 2**31 - 1, 0,
                              threads=threads)
        assert got == want


@pytest.mark.parametrize("start,n", [(0, 3), (123456, 66000), ((1 << 32) - 5, 70000)])
def test_converter_and_vector_job_digest(start, n):
    cfg = O.config_cli(O.R2P, 24, 24, 2, -1, 20)
    g, x, y = _inputs(start, n, 24)
    m, p = O.topolar(cfg, x, y)
    want = (cpu_digest(m, start) + cpu_digest(p, start + (1 << 40))) % 2**64
    assert O.job_digest(cfg, "r2p", start, n, threads=2)[0] == want

    cfg = O.config_cli(O.P2R, 32, 32, 2, 32, 16)
    g, x, y = _inputs(start, n, 32)
    with np.errstate(over="ignore"):
        ph = g << np.uint32(2)
    ox, oy = O.rotate(cfg, x, y, ph)
    want = (cpu_digest(ox, start) + cpu_digest(oy, start + (1 << 40))) % 2**64
    assert O.job_digest(cfg, "p2rxy", start, n, 0, 4, threads=2)[0] == want


def test_digest_words_is_the_numpy_twin():
    rng = np.random.RandomState(1)
    w = rng.randint(-2**31, 2**31 - 1, size=100003, dtype=np.int64).astype(np.int32)
    assert O.digest_words(w, 1 << 41) == cpu_digest(w, 1 << 41)
    assert O.digest_words(w[:0], 3) == 0
