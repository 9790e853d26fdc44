"""cordic_p2r_host / cordic_r2p_host (cordic_host.cpp): the chunked copy
pipeline behind the host-array entry points -- ragged sizes around the chunk
boundaries, pageable / pinned / mixed arrays, constant and per-sample vectors,
bit for bit against the oracle; and that a busy stream of the caller is never
waited for."""
import time

import numpy as np
import pytest

import cordic_amd as ca
import oracle_lib as O

pytestmark = pytest.mark.gpu

CHUNK = 4 << 20


def both(*a):
    return ca.Config.from_cli(*a), O.config_cli(*a)


def _inputs(n, iw, seed):
    rng = np.random.RandomState(seed)
    lo, hi = -(1 << (iw - 1)), (1 << (iw - 1))
    x = rng.randint(lo, hi, size=n, dtype=np.int64).astype(np.int32)
    y = rng.randint(lo, hi, size=n, dtype=np.int64).astype(np.int32)
    ph = rng.randint(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    return x, y, ph


@pytest.mark.parametrize("n", [1, 3, 1000, (1 << 18) + 1, CHUNK - 1, CHUNK,
                               CHUNK + 5, 3 * CHUNK + 77])
def test_pageable_arrays_every_size(n):
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    x, y, ph = _inputs(n, 32, n & 0xffff)
    a = ca.p2r_host(cfg, x, y, ph)
    st = ca.host_last_stats()
    assert st["samples"] == n and st["chunks"] == -(-n // CHUNK)
    assert st["seeded_plan"] == 0
    if n * 4 > 1 << 20:
        assert st["staged_inputs"] == 3 and st["staged_outputs"] == 2
        assert st["copy_threads"] >= 1
    b = O.rotate(ocfg, x, y, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a = ca.p2r_host(cfg, 2**31 - 1, 0, ph)
    assert ca.host_last_stats()["seeded_plan"] == (1 if n >= 4 else 0)
    b = O.rotate(ocfg, 2**31 - 1, 0, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    cfg, ocfg = both(ca.R2P, 24, 24, 2, -1, 20)
    x, y, _ = _inputs(n, 24, n + 1)
    a = ca.r2p_host(cfg, x, y)
    b = O.topolar(ocfg, x, y)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("pin_in,pin_out", [(True, True), (True, False),
                                            (False, True)])
def test_pinned_and_mixed_arrays(pin_in, pin_out):
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 24)
    n = 5 * CHUNK + 12345
    _, _, ph = _inputs(n, 32, 7)
    keep = []
    if pin_in:
        h = ca.HostArray(n, "uint32"); keep.append(h)
        h.array[:] = ph
        ph_in = h.array
    else:
        ph_in = ph
    out = None
    if pin_out:
        ox, oy = ca.HostArray(n), ca.HostArray(n); keep += [ox, oy]
        ox.array[:] = -1; oy.array[:] = -1
        out = (ox.array, oy.array)
    a = ca.p2r_host(cfg, 2**31 - 1, 0, ph_in, out=out)
    st = ca.host_last_stats()
    assert st["staged_inputs"] == (0 if pin_in else 1)
    assert st["staged_outputs"] == (0 if pin_out else 2)
    b = O.rotate(ocfg, 2**31 - 1, 0, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # a second, different core through the same cached pipeline: new plan
    cfg2, ocfg2 = both(ca.P2R, 32, 32, 2, 32, 16)
    a = ca.p2r_host(cfg2, 12345, -777, ph_in, out=out)
    b = O.rotate(ocfg2, 12345, -777, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for h in keep:
        h.close()


def test_other_streams_of_the_caller_are_not_waited_for():
    """Round 3 synchronised the whole DEVICE; now only the pipeline's own
    streams: a long-running stream of the caller must still be busy when the
    host-array call returns."""
    import torch
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(device=dev)
    big_cfg = ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 24).with_flags(ca.FLAG_NO_SEED)
    n_big = 1 << 28
    phs = torch.zeros(n_big, dtype=torch.int32, device=dev)
    oa = torch.empty_like(phs); ob = torch.empty_like(phs)
    _, _, ph = _inputs(1000, 32, 3)
    ca.p2r_host(cfg, 2**31 - 1, 0, ph)      # pipeline and plan exist from here on
    torch.cuda.synchronize()
    done = torch.cuda.Event()
    for _ in range(80):                     # ~80 x 1.4 ms of queued kernels
        ca.p2r_const(big_cfg, 1, 0, phs, oa, ob, stream=side)
    done.record(side)
    t0 = time.perf_counter()
    a = ca.p2r_host(cfg, 2**31 - 1, 0, ph)
    dt = time.perf_counter() - t0
    still_busy = not done.query()
    torch.cuda.synchronize()
    b = O.rotate(ocfg, 2**31 - 1, 0, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert still_busy, "the host-array call waited for an unrelated stream (%.1f ms)" % (dt * 1e3)


def test_release_and_reuse():
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    _, _, ph = _inputs(CHUNK + 9, 32, 4)
    a = ca.p2r_host(cfg, 5, 6, ph)
    ca.host_release()
    b = ca.p2r_host(cfg, 5, 6, ph)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    r = O.rotate(ocfg, 5, 6, ph)
    assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1])


def test_pageable_arrays_at_odd_offsets():
    """Caller arrays that start 4, 12 or 20 bytes past a 64-byte line (views
    into larger buffers): the staging copies' aligned non-temporal body has
    unaligned heads and tails on both sides."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    n = 2 * CHUNK + 1237
    _, _, ph = _inputs(n + 8, 32, 9)
    for k_in, k_out in ((1, 3), (5, 1), (3, 5)):
        phv = ph[k_in:k_in + n]
        oxb = np.full(n + 8, -7, dtype=np.int32)
        oyb = np.full(n + 8, -7, dtype=np.int32)
        out = (oxb[k_out:k_out + n], oyb[k_out:k_out + n])
        assert phv.ctypes.data % 16 != 0 and out[0].ctypes.data % 16 != 0
        a = ca.p2r_host(cfg, 2**31 - 1, 0, phv, out=out)
        assert ca.host_last_stats()["staged_outputs"] == 2
        b = O.rotate(ocfg, 2**31 - 1, 0, phv)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # nothing written outside the views
        assert (oxb[:k_out] == -7).all() and (oxb[k_out + n:] == -7).all()
        assert (oyb[:k_out] == -7).all() and (oyb[k_out + n:] == -7).all()


def test_concurrent_callers_queue_up_behind_each_other():
    """The pipeline of a device serialises host-array calls (a mutex): four
    host threads with different cores and sizes, every result its own."""
    import threading
    jobs = [((ca.P2R, 32, 32, 2, 32, 16), CHUNK + 11, 2**31 - 1, 0),
            ((ca.P2R, 32, 32, 2, 32, 24), 2 * CHUNK + 5, 12345, -9),
            ((ca.P2R, 16, 16, 2, -1, -1), 70001, 32767, 1),
            ((ca.P2R, 32, 32, 2, 32, 16), 3, -5, 77)]
    out, err = [None] * len(jobs), []

    def work(k):
        try:
            args, n, x0, y0 = jobs[k]
            cfg = ca.Config.from_cli(*args)
            _, _, ph = _inputs(n, 32, 40 + k)
            for _ in range(3):
                out[k] = (ca.p2r_host(cfg, x0, y0, ph), ph)
        except Exception as e:              # pragma: no cover
            err.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for k, (args, n, x0, y0) in enumerate(jobs):
        ocfg = O.config_cli(*args)
        (a, b), ph = out[k]
        pw = ocfg.pw
        rx, ry = O.rotate(ocfg, x0, y0, ph & np.uint32((1 << pw) - 1 if pw < 32
                                                       else 0xffffffff))
        assert np.array_equal(a, rx) and np.array_equal(b, ry), k


@pytest.mark.parametrize("pinned", [False, True])
def test_two_lanes_on_one_device_split_the_job(pinned):
    """cordic_host_set_devices (round 5): with a device
    list every host-array call is cut into contiguous parts, one pipeline per
    list entry, each driven by its own host thread -- with 8 GPUs that is 8
    PCIe links.  devices = [0, 0]: two pipelines on THIS device; both must do
    their share, the parts must not disturb each other, the bits must be the
    oracle's."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    n = 5 * CHUNK + 77
    x, y, ph = _inputs(n, 32, 21)
    keep = []

    def arr(a, dtype):
        if not pinned:
            return a
        h = ca.HostArray(n, dtype)
        keep.append(h)
        h.array[:] = a
        return h.array
    xi, yi, pi = arr(x, "int32"), arr(y, "int32"), arr(ph, "uint32")
    ca.host_set_devices([0, 0])
    try:
        for scalar in (True, False):
            if scalar:
                a = ca.p2r_host(cfg, 2**31 - 1, 0, pi)
                b = O.rotate(ocfg, 2**31 - 1, 0, ph)
            else:
                a = ca.p2r_host(cfg, xi, yi, pi)
                b = O.rotate(ocfg, x, y, ph)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            st = ca.host_last_stats()
            assert st["lanes"] == 2 and st["samples"] == n and st["chunks"] == 6
            l0, l1 = ca.host_lane_stats(0), ca.host_lane_stats(1)
            # 6 chunks over 2 lanes: 3 whole chunks, and 2 chunks + 77 samples
            assert l0["samples"] == 3 * CHUNK and l1["samples"] == 2 * CHUNK + 77
            assert l0["chunks"] == 3 and l1["chunks"] == 3
            if scalar:
                assert st["seeded_plan"] == 1
        # the converter through the same lanes
        rcfg, rocfg = both(ca.R2P, 24, 24, 2, -1, 20)
        x24, y24, _ = _inputs(n, 24, 22)
        a = ca.r2p_host(rcfg, x24, y24)
        b = O.topolar(rocfg, x24, y24)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert ca.host_last_stats()["lanes"] == 2
        # a job of one chunk is not worth a second lane
        a = ca.p2r_host(cfg, 5, 6, ph[:CHUNK])
        assert ca.host_last_stats()["lanes"] == 1
        b = O.rotate(ocfg, 5, 6, ph[:CHUNK])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    finally:
        ca.host_set_devices([])
    a = ca.p2r_host(cfg, 2**31 - 1, 0, pi)
    assert ca.host_last_stats()["lanes"] == 1
    with pytest.raises(ca.CordicError):
        ca.host_set_devices([0, 99])
    for h in keep:
        h.close()
    ca.host_release()
