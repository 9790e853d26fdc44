"""Host logic of the seed tables (cordic_plan.cpp) without a GPU: the leaves
must partition [-45deg, +45deg) and the bucket lookup must return, for any
folded phase, the leaf whose direction pattern and phase offset are exactly
what the stage recurrence of rtl/cordic.v:262-280 produces."""
import numpy as np
import pytest

import cordic_amd as ca

CASES = [(ca.P2R, 32, 32, 2, 32, 16), (ca.P2R, 32, 32, 2, 32, 24),
         (ca.SP2R, 32, 32, 2, 32, 16), (ca.P2R, 13, 13, 2, -1, -1),
         (ca.P2R, 16, 16, 2, 16, 16), (ca.P2R, 12, 12, 2, 14, 14),
         (ca.P2R, 30, 30, 2, 32, 16), (ca.SP2R, 13, 13, 2, 20, 16)]


def parse(words):
    m, S, nb, L = (int(v) for v in words[:4])
    buckets = words[4:4 + nb * 2].reshape(nb, 2).astype(np.int64)
    leaves = words[4 + nb * 2:4 + nb * 2 + L * 2].reshape(L, 2).astype(np.int64)
    return m, S, nb, L, buckets, leaves


def recurrence(p0, ang, m):
    """directions and residual phase after m stages (exact integers)."""
    pat = np.zeros(p0.shape, dtype=np.int64)
    p = p0.astype(np.int64).copy()
    for i in range(m):
        pos = p >= 0
        pat = (pat << 1) | pos
        p = np.where(pos, p - ang[i], p + ang[i])
    return pat, p


@pytest.mark.parametrize("args", CASES)
def test_lookup_matches_recurrence(args):
    cfg = ca.Config.from_cli(*args)
    words = ca.seed_table(cfg)
    assert words is not None
    m, S, nb, L, buckets, leaves = parse(words)
    assert m == 11 and nb == 1 << (30 - S) and 1 <= L <= 2048
    assert nb * 8 + L * 64 + 64 <= 160 * 1024       # fits the LDS of one CU
    ang = [a << (32 - cfg.pw) for a in cfg.angles]
    rng = np.random.RandomState(1)
    r = rng.randint(0, 1 << 30, 200000).astype(np.int64)
    # plus every value around every breakpoint the recurrence can produce
    extra = []
    for i in range(m + 1):
        for signs in range(1 << i):
            off = sum((ang[j] if (signs >> j) & 1 else -ang[j])
                      for j in range(i))
            for d in (-2, -1, 0, 1, 2):
                v = off + d + (1 << 29)
                if 0 <= v < (1 << 30):
                    extra.append(v)
    r = np.concatenate([r, np.array(extra, dtype=np.int64),
                        np.array([0, 1, (1 << 30) - 1], dtype=np.int64)])
    b = r >> S
    # at most one leaf boundary per bucket: one compare
    j = buckets[b, 1] + (buckets[b, 0] - r < 0)
    assert j.max() < L
    # what the kernel does with it (cordic_device.h: rotator_seeded): a
    # bucket without a boundary gets its last phase as the bound, and the
    # compare is bit 29 of (bound-1) - pb with the quadrant bits still in pb
    bound = np.where(buckets[:, 0] == 0x7fffffff,
                     ((np.arange(nb, dtype=np.int64) + 1) << S) - 1,
                     buckets[:, 0])
    for q in range(4):
        pb = r + (q << 30)
        c = (((bound[b] - pb) & 0xffffffff) >> 29) & 1
        assert np.array_equal(buckets[b, 1] + c, j)
    pat, pm = recurrence(r - (1 << 29), ang, m)
    assert np.array_equal(leaves[j, 0], pat)
    # off + 2^29 is stored modulo 2^32 (a leaf at the lower edge can have an
    # offset a little below -2^29); the kernels use it modulo 2^29 / 2^30
    off = leaves[j, 1] - (1 << 29)
    assert np.array_equal(((r - (1 << 29)) - off) & 0xffffffff, pm & 0xffffffff)
    # residual phase stays far inside 32 bits (the kernels rely on it)
    assert np.abs(pm).max() <= 1 << 29


def test_ineligible_cores_have_no_table():
    assert ca.seed_table(ca.Config.from_cli(ca.R2P, 13, 13, 2)) is None
    assert ca.seed_table(ca.Config.from_cli(ca.P2R, 32, 32, 3, 32, 16)) is None  # WW 36
    assert ca.seed_table(ca.Config.from_cli(ca.P2R, 8, 8, 2, 12, 6)) is None    # < 11 stages


def test_degenerate_angle_tables_are_not_seeded():
    """PW = 3: every angle after the first truncates to zero, so the residual
    after the seed stages can be as large as the folded phase itself -- more
    than the 29-bit field the left-justified kernels keep.  Such cores must
    not get a seed table (they run the full recurrence); found by the fuzz
    (sp2r -i 9 -o 31 -x 3 -p 3 -n 33)."""
    cfg = ca.Config.from_cli(ca.SP2R, 9, 31, 3, 3, 33)
    assert cfg.ww == 35 and cfg.nlive == 31
    assert ca.seed_table(cfg) is None
    # while ordinary cores of the same width are
    assert ca.seed_table(ca.Config.from_cli(ca.SP2R, 9, 31, 3, 24, 33)) is not None


# ------------------------------------------------------- direction tails

def parse_tails(words):
    """the groups appended behind the seed table (cordic_plan.cpp)"""
    m, S, nb, L, buckets, leaves = parse(words)
    at = 4 + nb * 2 + L * 2
    if at >= len(words):
        return None
    n, bias0, bias_last = int(words[at]), int(words[at + 1]), int(words[at + 2])
    at += 4
    groups = []
    for _ in range(n):
        t, S2, nb2, nl2 = (int(v) for v in words[at:at + 4])
        at += 6
        bk = words[at:at + nb2 * 2].reshape(nb2, 2).astype(np.int64)
        at += nb2 * 2
        lf = words[at:at + nl2 * 2].reshape(nl2, 2).astype(np.int64)
        at += nl2 * 2
        groups.append(dict(t=t, S=S2, nb=nb2, nl=nl2, buckets=bk, leaves=lf))
    assert at == len(words)
    return dict(bias0=bias0, bias_last=bias_last, groups=groups)


def schedule(r):
    """cordic_internal.h: dt_levels / dt_size -- stages 11..24 at most, in the
    fewest groups of at most seven stages, equal sizes, the longer ones last;
    none under five stages; what is left runs the recurrence."""
    if r < 5:
        return [], r
    c = min(r, 25 - 11)
    n = (c + 6) // 7
    return [c // n + (1 if g >= n - c % n else 0) for g in range(n)], r - c


@pytest.mark.parametrize("args", [(ca.P2R, 32, 32, 2, 32, 16),
                                  (ca.P2R, 32, 32, 2, 32, 18),
                                  (ca.P2R, 32, 32, 2, 32, 30),
                                  (ca.SP2R, 32, 32, 2, 32, -1),
                                  (ca.P2R, 24, 24, 2, -1, -1),
                                  (ca.P2R, 16, 16, 2, -1, -1),
                                  (ca.P2R, 32, 32, 2, 32, 21),
                                  (ca.P2R, 32, 32, 2, 32, 24),
                                  (ca.SP2R, 32, 32, 2, 32, 22),
                                  (ca.P2R, 32, 32, 2, 32, 26),
                                  (ca.P2R, 32, 32, 2, 32, 20),
                                  (ca.P2R, 31, 31, 2, 30, 21),
                                  (ca.P2R, 30, 30, 3, 32, 23)])
def test_direction_tails_match_the_recurrence(args):
    """Behind the seed stages the directions of each group of stages, looked
    up by the biased residual (bucket + one compare), are the directions the
    exact phase recurrence takes, and the residual handed on is exact -- for
    random phases and for every group boundary +-2."""
    cfg = ca.Config.from_cli(*args)
    words = ca.seed_table(cfg)
    m, S, nb, L, buckets, leaves = parse(words)
    tails = parse_tails(words)
    sizes, rest = schedule(cfg.nlive - m)
    assert tails is not None and [g["t"] for g in tails["groups"]] == sizes
    ang = [a << (32 - cfg.pw) for a in cfg.angles]
    rng = np.random.RandomState(7)
    r = rng.randint(0, 1 << 30, 300000).astype(np.int64)

    def first_level(r):
        j = buckets[r >> S, 1] + (buckets[r >> S, 0] - r < 0)
        off = leaves[j, 1] - (1 << 29)          # stored modulo 2^32
        d = ((r - (1 << 29)) - off) & 0xffffffff
        return np.where(d >= 1 << 31, d - (1 << 32), d)   # residual behind the seeds

    res = first_level(r)
    # plus residuals around every boundary of every group: walk the groups
    # once with the random set to learn the biases, then add +-2 around each
    # bound of each group, mapped back to the FIRST residual is not needed --
    # each group is checked on its own incoming residual below
    u = res + tails["bias0"]
    assert u.min() >= 0 and u.max() < (1 << 28)
    k0 = m
    incoming = res.copy()
    for g in tails["groups"]:
        bk, lf, t = g["buckets"], g["leaves"], g["t"]
        bounds = bk[bk[:, 0] != 0x7fffffff, 0] + 1
        extra = np.concatenate([bounds + d for d in (-2, -1, 0, 1, 2)])
        extra = extra[(extra >= 0) & (extra <= u.max())]
        u = np.concatenate([u, extra])
        # the residual those u stand for (bias of this group's input)
        bias_in = u[0] - incoming[0]
        incoming = u - bias_in
        b = u >> g["S"]
        assert b.max() < g["nb"]
        j = bk[b, 1] + (bk[b, 0] - u < 0)
        assert j.max() < g["nl"]
        pat, after = recurrence(incoming, ang[k0:k0 + t], t)
        assert np.array_equal(lf[j, 0], pat)
        u = u - lf[j, 1]
        # u is now the next group's (or the chain's) biased residual
        assert np.array_equal(u - u[0], after - after[0])
        incoming = after
        k0 += t
    assert np.array_equal(u - tails["bias_last"], incoming)
    assert k0 + rest == cfg.nlive


def test_cores_without_room_or_need_have_no_tails():
    # fewer than three stages behind the seeds: nothing to look up
    w = ca.seed_table(ca.Config.from_cli(ca.P2R, 16, 16, 2, 16, 16))   # 13 live
    assert parse_tails(w) is None
    # fewer than five: not measured, none built
    for ns in (13, 14, 15):
        w = ca.seed_table(ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, ns))
        assert parse_tails(w) is None, ns
    t = parse_tails(ca.seed_table(ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 16)))
    assert [g["t"] for g in t["groups"]] == [5]
    # the last stages of a 29-stage core move the phase by 1..5 units: their
    # leaves are narrower than the smallest bucket; the tails stop at stage 24
    t = parse_tails(ca.seed_table(ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 30)))
    assert [g["t"] for g in t["groups"]] == [7, 7]
    # (WW <= 32 cores have tails like the wide ones: multipliers -/+ 1)
    t = parse_tails(ca.seed_table(ca.Config.from_cli(ca.P2R, 13, 13, 2, -1, -1)))   # 16 live
    assert [g["t"] for g in t["groups"]] == [5]
    t = parse_tails(ca.seed_table(ca.Config.from_cli(ca.P2R, 16, 16, 2, -1, -1)))
    assert [g["t"] for g in t["groups"]] == [4, 4]
