"""The plan's cache of seeded-kernel prologues (round 5, include/cordic_amd.h:
"the plan KEEPS that prologue's result"): a launch served from an image must
produce the bits of a launch that computed its own prologue -- and of the
oracle -- whatever the vector, the container, the feed, the stream, and
whether or not a HIP graph is being captured."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cordic_amd as ca
import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if torch.cuda.is_available():
    from gpu_util import (DEV, dev_i32, gpu_plan_nco, gpu_plan_p2r, to_np)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def both(mode, iw=-1, ow=-1, xtra=2, pw=-1, ns=-1, flags=0):
    cfg = ca.Config.from_cli(mode, iw, ow, xtra, pw, ns)
    if flags:
        cfg = cfg.with_flags(flags)
    return cfg, O.config_cli(mode, iw, ow, xtra, pw, ns)


def phases(n, seed=5):
    rng = np.random.RandomState(seed)
    ph = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    ph[:64] = np.arange(64, dtype=np.uint32) * np.uint32(0x04000000)
    return ph


CORES = {
    # container lj29, one group of tails on every row
    "cfg2": ((ca.P2R, 32, 32, 2, 32, 16), 0),
    # lj29, two groups, rows choose tails / recurrence
    "cfg4": ((ca.P2R, 32, 32, 2, 32, 24), 0),
    # lj30 (WW 19 carried left-justified), tails 4+4
    "nat16": ((ca.P2R, 16, 16, 2, -1, -1), 0),
    # the same core in the 32-bit container
    "nat16_narrow": ((ca.P2R, 16, 16, 2, -1, -1), ca.FLAG_NO_LJ),
    # no tails in the image (A/B flag): seeds only
    "cfg4_notails": ((ca.P2R, 32, 32, 2, 32, 24), ca.FLAG_NO_TAILS),
    # dynamic-exit instance with the fused output scaling
    "cfg2_unit_gain": ((ca.P2R, 32, 32, 2, 32, 16), ca.FLAG_UNIT_GAIN),
    # a stage count without a static instance (dynamic exit)
    "ns12": ((ca.P2R, 24, 24, 2, 28, 12), 0),
    # sequential core's arithmetic
    "cfg5seq": ((ca.SP2R, 32, 32, 2, 32, 16), 0),
}


@pytest.mark.parametrize("name", sorted(CORES))
def test_launches_served_from_an_image_equal_the_oracle(name):
    args, flags = CORES[name]
    cfg, ocfg = both(*args, flags=flags)
    if flags & ca.FLAG_UNIT_GAIN:
        pytest.skip("oracle has no unit-gain twin; covered by the A/B below")
    plan = ca.Plan(cfg)
    assert plan.seed_info["stages"] == 11
    n = (1 << 18) + 7
    ph = phases(n)
    hi = (1 << (cfg.iw - 1)) - 1
    vecs = [(hi, 0), (-hi - 1, 12345 % hi), (0, hi), (3, -5)]
    for rep in range(3):            # 1st: builds, 2nd / 3rd: served
        for x0, y0 in vecs:
            gx, gy = gpu_plan_p2r(plan, x0, y0, ph)
            rx, ry = O.rotate(ocfg, x0, y0, ph)
            assert np.array_equal(gx, rx) and np.array_equal(gy, ry), (rep, x0, y0)
            assert ca.last_kernel() == ca.KERNEL_SEEDED
    info = plan.image_info
    assert info["held"] == len(vecs) and info["hits"] >= 2 * len(vecs)
    # the NCO feed reads the images the phase-array launches built
    before = plan.image_info["hits"]
    for x0, y0 in vecs[:2]:
        a = gpu_plan_nco(plan, 50001, 0x1234567, 0x9e3779b9, (3 << 32) + 17, x0, y0)
        b = O.nco(ocfg, 50001, 0x1234567, 0x9e3779b9, (3 << 32) + 17, x0, y0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert plan.image_info["hits"] == before + 2
    assert plan.image_info["held"] == len(vecs)
    plan.close()


@pytest.mark.parametrize("name", sorted(CORES))
def test_image_on_and_off_give_the_same_bits(name):
    """CORDIC_SEED_IMAGES=0 at plan creation: every block computes its own
    prologue (the round-4 behaviour).  Same outputs, bit for bit."""
    args, flags = CORES[name]
    cfg, _ = both(*args, flags=flags)
    n = (1 << 17) + 12
    ph = phases(n, 9)
    hi = (1 << (cfg.iw - 1)) - 1
    on = ca.Plan(cfg)
    os.environ["CORDIC_SEED_IMAGES"] = "0"
    try:
        off = ca.Plan(cfg)
    finally:
        del os.environ["CORDIC_SEED_IMAGES"]
    for x0, y0 in [(hi, 0), (-77, hi // 3)]:
        for _ in range(2):
            a = gpu_plan_p2r(on, x0, y0, ph)
            b = gpu_plan_p2r(off, x0, y0, ph)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert on.image_info["held"] == 2 and on.image_info["hits"] == 2
    assert off.image_info == dict(held=0, hits=0, misses=0)
    on.close(); off.close()


def test_more_vectors_than_slots_fall_back_to_the_in_kernel_prologue():
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    plan = ca.Plan(cfg)
    ph = phases(1 << 16, 2)
    for k in range(12):             # 8 slots
        for _ in range(2):
            x0, y0 = 1000 + 17 * k, -3 * k
            gx, gy = gpu_plan_p2r(plan, x0, y0, ph)
            rx, ry = O.rotate(ocfg, x0, y0, ph)
            assert np.array_equal(gx, rx) and np.array_equal(gy, ry), k
    info = plan.image_info
    assert info["held"] == 8
    assert info["hits"] == 8 and info["misses"] == 8 + 2 * 4
    with pytest.raises(ca.CordicError) as e:
        plan.prepare(5, 5)          # no slot left
    assert e.value.status == ca.ERR_UNSUPPORTED
    plan.close()


def test_inputs_are_keyed_by_the_port_value_not_the_word():
    """i_xval / i_yval are taken modulo their port width: two words with the
    same low IW bits are the same vector and share an image."""
    cfg, ocfg = both(ca.P2R, 13, 13, 2)
    plan = ca.Plan(cfg)
    ph = phases(1 << 15, 4) & np.uint32((1 << cfg.pw) - 1)
    a = gpu_plan_p2r(plan, 4095, -4096, ph)
    b = gpu_plan_p2r(plan, 4095 + (5 << 13), -4096 - (1 << 13), ph)
    r = O.rotate(ocfg, 4095, -4096, ph)
    for g in (a, b):
        assert np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1])
    assert plan.image_info["held"] == 1 and plan.image_info["hits"] == 1
    plan.close()


def test_int16_and_int32_calls_on_one_plan_keep_separate_images():
    """The 16-bit containers run the 32-bit register form, the 32-bit arrays
    of the same core the left-justified one: different seed layouts."""
    cfg, ocfg = both(ca.P2R, 16, 16, 2, 16, 16)
    plan = ca.Plan(cfg)
    n = 1 << 16
    ph = (np.arange(n, dtype=np.uint32) * np.uint32(40503)) & np.uint32(0xffff)
    rx, ry = O.rotate(ocfg, 32767, 0, ph)
    p32 = dev_i32(ph)
    p16 = p32.to(torch.int16)
    for _ in range(2):
        a32 = torch.zeros(n, dtype=torch.int32, device=DEV)
        b32 = torch.zeros_like(a32)
        plan.p2r_const(32767, 0, p32, a32, b32)
        a16 = torch.zeros(n, dtype=torch.int16, device=DEV)
        b16 = torch.zeros_like(a16)
        plan.p2r_const(32767, 0, p16, a16, b16)
        torch.cuda.synchronize()
        assert np.array_equal(to_np(a32), rx) and np.array_equal(to_np(b32), ry)
        assert np.array_equal(a16.cpu().numpy(), rx.astype(np.int16))
        assert np.array_equal(b16.cpu().numpy(), ry.astype(np.int16))
    assert plan.image_info["held"] == 2 and plan.image_info["hits"] == 2
    plan.close()


def test_prepare_builds_the_int16_image_only_for_plans_that_use_int16_arrays():
    """cordic_plan_prepare took TWO of a 16-bit core's eight
    write-once slots per vector, for an int16 image most plans never use.  Now
    the int16 image is built by prepare only once the plan has served an
    int16 call."""
    cfg, _ = both(ca.P2R, 16, 16, 2, 16, 16)
    plan = ca.Plan(cfg)
    for k in range(8):                  # eight vectors, eight slots
        plan.prepare(100 + k, k)
    assert plan.image_info["held"] == 8
    plan.close()
    plan = ca.Plan(cfg)
    n = 1 << 14
    p16 = torch.zeros(n, dtype=torch.int16, device=DEV)
    a16 = torch.zeros_like(p16)
    b16 = torch.zeros_like(p16)
    plan.p2r_const(7, 0, p16, a16, b16)         # an int16 call: its own image
    torch.cuda.synchronize()
    assert plan.image_info["held"] == 1
    plan.prepare(9, 9)                          # now both containers
    assert plan.image_info["held"] == 3
    plan.close()


def test_first_use_on_one_stream_next_use_on_another():
    """The build runs on the first launch's stream; a launch on another stream
    right behind it (nothing synchronised in between) is ordered behind the
    build on the device."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 24)
    n = (1 << 20) + 4
    ph = phases(n, 11)
    dph = dev_i32(ph)
    x0 = (1 << 31) - 1
    rx, ry = O.rotate(ocfg, x0, 0, ph)
    for trial in range(4):
        plan = ca.Plan(cfg)
        s = [torch.cuda.Stream(device=DEV) for _ in range(3)]
        outs = [[torch.zeros(n, dtype=torch.int32, device=DEV) for _ in range(2)]
                for _ in range(6)]
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(outs):
            plan.p2r_const(x0, 0, dph, a, b, n=n, stream=s[k % 3])
        torch.cuda.synchronize()
        for a, b in outs:
            assert np.array_equal(to_np(a), rx) and np.array_equal(to_np(b), ry)
        assert plan.image_info["held"] == 1
        plan.close()


def test_a_captured_launch_only_uses_a_finished_image():
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    n = 1 << 18
    ph = phases(n, 13)
    dph = dev_i32(ph)
    x0 = (1 << 31) - 1
    rx, ry = O.rotate(ocfg, x0, 0, ph)
    ox = torch.zeros(n, dtype=torch.int32, device=DEV)
    oy = torch.zeros(n, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    # (a) nothing prepared: the captured launch computes its own prologue
    plan = ca.Plan(cfg)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.p2r_const(x0, 0, dph, ox, oy)
    assert plan.image_info == dict(held=0, hits=0, misses=1)
    for _ in range(3):
        ox.zero_(); oy.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(to_np(ox), rx) and np.array_equal(to_np(oy), ry)
    plan.close()
    # (b) prepared and synchronised: the graph holds the image's address
    plan = ca.Plan(cfg)
    plan.prepare(x0, 0)             # returns with the image complete
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.p2r_const(x0, 0, dph, ox, oy)
    assert plan.image_info == dict(held=1, hits=1, misses=1)   # (prepare: the miss)
    for _ in range(3):
        ox.zero_(); oy.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(to_np(ox), rx) and np.array_equal(to_np(oy), ry)
    # eager launches with other vectors in between do not disturb it
    gpu_plan_p2r(plan, 5, 6, ph)
    ox.zero_(); oy.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(to_np(ox), rx) and np.array_equal(to_np(oy), ry)
    plan.close()


def test_prepare_is_refused_where_there_is_no_seed_table():
    cfg, _ = both(ca.R2P, 24, 24, 2, -1, 20)
    plan = ca.Plan(cfg)
    with pytest.raises(ca.CordicError) as e:
        plan.prepare(1, 2)
    assert e.value.status == ca.ERR_UNSUPPORTED
    plan.close()
    cfg, _ = both(ca.P2R, 32, 32, 8, 32, 24)       # WW 41: no seeded kernel
    plan = ca.Plan(cfg)
    with pytest.raises(ca.CordicError):
        plan.prepare(1, 2)
    plan.close()


def test_min_samples_is_a_property_of_the_plan():
    """cordic_plan_set_min_samples: the batch size from which THIS plan's
    table kernels serve, independent of the process-wide default."""
    cfg, ocfg = both(ca.P2R, 32, 32, 2, 32, 16)
    n = 1 << 16
    ph = phases(n, 17)
    x0 = (1 << 31) - 1
    rx, ry = O.rotate(ocfg, x0, 0, ph)
    a, b = ca.Plan(cfg), ca.Plan(cfg)
    a.set_min_samples(n + 1)        # this batch is too small for the tables
    b.set_min_samples(n)            # ... and just large enough here
    ga = gpu_plan_p2r(a, x0, 0, ph)
    assert ca.last_kernel() == ca.KERNEL_UNROLLED
    gb = gpu_plan_p2r(b, x0, 0, ph)
    assert ca.last_kernel() == ca.KERNEL_SEEDED
    for g in (ga, gb):
        assert np.array_equal(g[0], rx) and np.array_equal(g[1], ry)
    a.set_min_samples(0)
    gpu_plan_p2r(a, x0, 0, ph)
    assert ca.last_kernel() == ca.KERNEL_SEEDED
    a.close(); b.close()


def test_group_shards_and_host_pipeline_use_images_without_the_env_override():
    """the host-array pipeline's 2^22-sample chunks never reached
    the seeded kernel with the library's defaults.  With the image the tables
    serve from 2^22 samples: run both layers in a process WITHOUT
    CORDIC_SEED_MIN_SAMPLES and look at what ran."""
    script = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np, torch
import cordic_amd as ca, oracle_lib as O
args = (ca.P2R, 32, 32, 2, 32, 16)
cfg, ocfg = ca.Config.from_cli(*args), O.config_cli(*args)
n = (1 << 23) + 4099
ph = (np.arange(n, dtype=np.uint32) << np.uint32(2))
a, b = ca.p2r_host(cfg, 2**31 - 1, 0, ph)
st = ca.host_last_stats()
assert st["chunks"] == 3 and st["seeded_plan"] == 1, st
assert ca.last_kernel() == ca.KERNEL_UNROLLED      # the 4099-sample last chunk
want, _ = O.job_digest(ocfg, "p2r", 0, n, 0, 4, 2**31 - 1, 0)
got = (O.digest_words(a, 0) + O.digest_words(b, 1 << 40)) %% 2**64
assert got == want
n2 = 1 << 22
a, b = ca.p2r_host(cfg, 2**31 - 1, 0, ph[:n2])
assert ca.last_kernel() == ca.KERNEL_SEEDED, ca.last_kernel()
ca.host_release()
g = ca.Group(cfg, devices=[0, 0])
g.fill_phase_ramp(1 << 23, 2)
for _ in range(3):
    g.p2r_const(1 << 23, 2**31 - 1, 0)
g.sync()
assert ca.last_kernel() == ca.KERNEL_SEEDED         # 2^22 per shard
assert g.digest(1 << 23) == O.job_digest(ocfg, "p2r", 0, 1 << 23, 0, 4, 2**31 - 1, 0)[0]
g.close()
print("ok")
""" % (ROOT, ROOT)
    env = {k: v for k, v in os.environ.items() if k != "CORDIC_SEED_MIN_SAMPLES"}
    r = subprocess.run([sys.executable, "-c", script], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_the_lower_batch_size_threshold_needs_an_image_that_serves_the_launch():
    """a plan takes its table kernels from 2^22 samples where the
    prologue comes from an image, from 2^23 where every block computes it.
    A NINTH constant vector (all eight slots taken) gets no image: at 2^22
    samples it has to run the plain kernel, while a vector that holds a slot
    runs the seeded one.  (The library's own thresholds: a child process
    without the suite's CORDIC_SEED_MIN_SAMPLES=0.)"""
    import subprocess
    import sys
    prog = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import torch, cordic_amd as ca\n"
        "cfg = ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 16)\n"
        "plan = ca.Plan(cfg)\n"
        "n = 1 << 22\n"
        "ph = torch.zeros(n, dtype=torch.int32, device='cuda')\n"
        "a = torch.empty_like(ph); b = torch.empty_like(ph)\n"
        "ca.fill_phase_ramp(ph, 0, 8)\n"
        "fam = []\n"
        "for k in range(9):\n"
        "    plan.p2r_const(1000 + k, 0, ph, a, b)\n"
        "    torch.cuda.synchronize(); fam.append(ca.last_kernel())\n"
        "plan.p2r_const(1003, 0, ph, a, b); fam.append(ca.last_kernel())\n"
        "half = torch.zeros(n // 2, dtype=torch.int32, device='cuda')\n"
        "plan.p2r_const(1003, 0, half, a[:n // 2], b[:n // 2]); fam.append(ca.last_kernel())\n"
        "big = torch.zeros(2 * n, dtype=torch.int32, device='cuda')\n"
        "a2 = torch.empty_like(big); b2 = torch.empty_like(big)\n"
        "plan.p2r_const(1008, 0, big, a2, b2); fam.append(ca.last_kernel())\n"
        "torch.cuda.synchronize()\n"
        "print('FAM', fam, plan.image_info['held'])\n"
        % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
           os.path.dirname(os.path.abspath(__file__))))
    env = {k: v for k, v in os.environ.items() if k != "CORDIC_SEED_MIN_SAMPLES"}
    r = subprocess.run([sys.executable, "-c", prog], env=env, text=True,
                       capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("FAM")][-1]
    fam = eval(line[4:line.rindex("]") + 1])
    S, U = ca.KERNEL_SEEDED, ca.KERNEL_UNROLLED
    # eight vectors take a slot each and the seeded kernel; the ninth runs the
    # plain one at this size; a slot holder still the seeded one; below 2^22
    # everybody the plain one; at 2^23 the ninth the seeded one (own prologue)
    assert fam == [S] * 8 + [U, S, U, S], fam
    assert line.rstrip().endswith(" 8")
