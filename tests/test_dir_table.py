"""Host logic of the direction tables of the per-sample-vector path
(cordic_plan.cpp: build_dir_table; kernel: cordic_xydir.h) without a GPU: the
chain of lookups -- bucket, one compare, leaf, next group's index -- must give,
for any folded phase, exactly the rotation directions of rtl/cordic.v:262-280
for every looked-up stage, and leave the exact residual phase behind."""
import numpy as np
import pytest

import cordic_amd as ca

CASES = [((ca.P2R, 32, 32, 2, 32, 16), [5, 5, 5]),
         ((ca.P2R, 32, 32, 2, 32, 24), [4, 4, 5, 5, 5]),
         ((ca.P2R, 32, 32, 2, 32, -1), [4, 5, 5, 5, 5]),
         ((ca.P2R, 24, 24, 2, -1, -1), [4, 5, 5, 5, 5]),
         ((ca.P2R, 16, 16, 2, -1, -1), [4, 4, 5, 5]),
         ((ca.SP2R, 32, 32, 2, 32, 16), [4, 4, 5]),
         ((ca.P2R, 13, 13, 2, -1, -1), None)]


def parse(words):
    n, bias0, bias_last = int(words[0]), int(words[1]), int(words[2])
    at, groups = 4, []
    for _ in range(n):
        t, sh, nb, nl = (int(v) for v in words[at:at + 4])
        at += 6
        bk = words[at:at + nb * 2].reshape(nb, 2).astype(np.int64)
        at += nb * 2
        lf = words[at:at + nl * 2].reshape(nl, 2).astype(np.int64)
        at += nl * 2
        groups.append((t, sh, nb, nl, bk, lf))
    assert at == words.size
    return bias0, bias_last, groups


@pytest.mark.parametrize("args,sizes", CASES)
def test_lookup_chain_matches_the_recurrence(args, sizes):
    cfg = ca.Config.from_cli(*args)
    words = ca.dir_table(cfg)
    if sizes is None:
        sizes = [g[0] for g in parse(words)[2]] if words is not None else []
    assert words is not None
    bias0, bias_last, groups = parse(words)
    assert [g[0] for g in groups] == sizes
    ang = [a << (32 - cfg.pw) for a in cfg.angles]
    rng = np.random.RandomState(2)
    p0 = rng.randint(-(1 << 29), 1 << 29, 300000).astype(np.int64)
    # every breakpoint of the first 12 stages, +/- 2
    sums = {0}
    for a in ang[:12]:
        sums = sums | {s + a for s in sums} | {s - a for s in sums}
        if len(sums) > 60000:
            break
    brk = np.array(sorted(sums), dtype=np.int64)
    brk = (brk[:, None] + np.arange(-2, 3)[None, :]).ravel()
    brk = brk[(brk >= -(1 << 29)) & (brk < (1 << 29))]
    p0 = np.concatenate([p0, brk, [-(1 << 29), (1 << 29) - 1, 0, -1]])
    # stage 1 is the fold's
    pos = p0 >= 0
    p = np.where(pos, p0 - ang[0], p0 + ang[0])
    u = p + bias0
    stage = 1
    for t, sh, nb, nl, bk, lf in groups:
        assert u.min() >= 0 and u.max() < (1 << 30)
        b = u >> sh
        assert b.max() < nb
        # the kernel's compare: sign of (bound-1) - u as a 32-bit difference
        c = (((bk[b, 0] - u) & 0xffffffff) >> 31) & 1
        j = bk[b, 1] + c
        assert j.max() < nl
        pat = np.zeros(p.shape, dtype=np.int64)
        for i in range(t):                  # rtl/cordic.v:262-280
            pos = p >= 0
            pat = (pat << 1) | pos
            p = np.where(pos, p - ang[stage + i], p + ang[stage + i])
        stage += t
        assert np.array_equal(lf[j, 0], pat)
        u = (u - lf[j, 1]) & 0xffffffff
    # behind the last group: u - bias_last is the exact residual phase
    r = ((u - bias_last + (1 << 31)) & 0xffffffff) - (1 << 31)
    assert np.array_equal(r, p)
    assert stage == min(cfg.nlive, 25)
    if cfg.nlive > stage:                   # the recurrence takes it from here
        assert np.abs(p).max() < 1 << 28
    # LDS the kernel needs: fold rows + per group buckets and leaf entries
    lds = 128 + sum(nb * 8 + nl * 48 for _, _, nb, nl, _, _ in groups)
    assert lds < 60 * 1024


def test_cores_without_a_table():
    for args in ((ca.P2R, 32, 32, 3, 32, 16),       # WW 36
                 (ca.R2P, 24, 24, 2, -1, 20),
                 (ca.P2R, 9, 31, 3, 3, 33)):        # degenerate angle table
        assert ca.dir_table(ca.Config.from_cli(*args)) is None
