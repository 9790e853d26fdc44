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


def chain_errors(cfg, words, sizes=None, nrandom=300000):
    """number of phases on which the chain of lookups disagrees with the
    recurrence (directions of a group, or the residual behind the last one)"""
    bias0, bias_last, groups = parse(words)
    if sizes is not None:
        assert [g[0] for g in groups] == sizes
    sizes = [g[0] for g in groups]
    ang = [a << (32 - cfg.pw) for a in cfg.angles]
    rng = np.random.RandomState(2)
    p0 = rng.randint(-(1 << 29), 1 << 29, nrandom).astype(np.int64)
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
    u = (p + bias0) & 0xffffffff
    stage = 1
    bad = np.zeros(p0.shape, dtype=bool)
    for t, sh, nb, nl, bk, lf in groups:
        b = np.minimum(u >> sh, nb - 1)
        # the kernel's compare: sign of (bound-1) - u as a 32-bit difference
        c = (((bk[b, 0] - u) & 0xffffffff) >> 31) & 1
        j = np.minimum(bk[b, 1] + c, nl - 1)
        pat = np.zeros(p.shape, dtype=np.int64)
        for i in range(t):                  # rtl/cordic.v:262-280
            pos = p >= 0
            pat = (pat << 1) | pos
            p = np.where(pos, p - ang[stage + i], p + ang[stage + i])
        stage += t
        bad |= lf[j, 0] != pat
        u = (u - lf[j, 1]) & 0xffffffff
    # behind the last group: u - bias_last is the exact residual phase
    r = ((u - bias_last + (1 << 31)) & 0xffffffff) - (1 << 31)
    bad |= r != p
    assert stage == min(cfg.nlive, 25) == 1 + sum(sizes)
    return int(bad.sum()), groups, np.abs(p).max()


@pytest.mark.parametrize("args,sizes", CASES)
def test_lookup_chain_matches_the_recurrence(args, sizes):
    cfg = ca.Config.from_cli(*args)
    words = ca.dir_table(cfg)
    assert words is not None
    errors, groups, pmax = chain_errors(cfg, words, sizes)
    assert errors == 0
    if cfg.nlive > 25:                      # the recurrence takes it from here
        assert pmax < 1 << 28
    # LDS the kernel needs: fold rows + per group buckets and leaf entries
    lds = 128 + sum(nb * 8 + nl * 48 for _, _, nb, nl, _, _ in groups)
    assert lds < 60 * 1024


def test_the_chain_check_discriminates():
    """One wrong word anywhere in the tables -- a direction bit, an offset, a
    bucket's bound or first leaf -- must show up: 40 single-word mutants of
    cfg2's tables, every one caught."""
    cfg = ca.Config.from_cli(ca.P2R, 32, 32, 2, 32, 16)
    words = ca.dir_table(cfg)
    assert chain_errors(cfg, words, nrandom=20000)[0] == 0
    rng = np.random.RandomState(7)
    _, _, groups = parse(words)
    # word positions of every group's buckets and leaves
    at, spans = 4, []
    for t, sh, nb, nl, _, _ in groups:
        at += 6
        spans.append(("bucket", at, nb * 2)); at += nb * 2
        spans.append(("leaf", at, nl * 2)); at += nl * 2
    caught = 0
    trials = 0
    shifts = [g[1] for g in groups for _ in range(2)]
    for (kind, base, count), sh in zip(spans, shifts):
        for _ in range(7):
            k = base + int(rng.randint(count))
            m = words.copy()
            if kind == "leaf" and (k - base) % 2 == 0:
                m[k] ^= np.uint32(1 << int(rng.randint(3)))     # a direction
            elif kind == "leaf":
                m[k] = np.uint32((int(m[k]) + 1) & 0xffffffff)  # offset off by one
            else:
                # only buckets that hold a boundary are certainly reachable (the
                # table is padded to a power of two); move the boundary by a
                # quarter of a bucket, or start the bucket one leaf too late
                k -= (k - base) % 2
                if int(m[k]) == 0x7fffffff:
                    continue
                if rng.randint(2):
                    m[k] = np.uint32(int(m[k]) + (1 << (sh - 2)))
                else:
                    m[k + 1] = np.uint32(int(m[k + 1]) + 1)
            trials += 1
            caught += chain_errors(cfg, m)[0] > 0
    assert trials >= 30 and caught == trials, (caught, trials)


def test_cores_without_a_table():
    for args in ((ca.P2R, 32, 32, 3, 32, 16),       # WW 36
                 (ca.R2P, 24, 24, 2, -1, 20),
                 (ca.P2R, 9, 31, 3, 3, 33)):        # degenerate angle table
        assert ca.dir_table(ca.Config.from_cli(*args)) is None
