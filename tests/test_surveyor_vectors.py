"""SURVEY.md 8(c)'s surveyor-derived vectors as a committed test.

They come from a restatement written independently of this repository's
oracle (by the surveyor, before any code here existed), so agreement is a
second reading of rtl/cordic.v:85-86,131-188,231-314 and rtl/topolar.v:83-84,
118-152,212-272 -- not a reference fixture.  Checked against the oracle (CPU)
and the engine (GPU: stateless entry points, plans, the NCO form)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from test_oracle_golden import parse_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cases():
    with open(os.path.join(ROOT, "tests", "golden", "surveyor_vectors.json")) as f:
        d = json.load(f)
    return {k: v for k, v in d.items() if not k.startswith("_")}


def arrays(e):
    s = e["samples"]
    x = np.array([v["x"] for v in s], dtype=np.int32)
    y = np.array([v["y"] for v in s], dtype=np.int32)
    if "phase" in s[0]:
        ph = np.array([int(v["phase"], 16) for v in s], dtype=np.uint32)
        return x, y, ph, (np.array([v["ox"] for v in s], dtype=np.int32),
                          np.array([v["oy"] for v in s], dtype=np.int32))
    return x, y, None, (np.array([v["mag"] for v in s], dtype=np.int32),
                        np.array([int(v["ophase"], 16) for v in s],
                                 dtype=np.uint32))


@pytest.mark.parametrize("name", sorted(cases()))
def test_oracle_reproduces_the_surveyors_vectors(name):
    e = cases()[name]
    d = parse_args(e["args"])
    c = O.config_cli(d["mode"], d["iw"], d["ow"], d["xtra"], d["pw"], d["n"])
    for k, v in e["expect"].items():
        assert getattr(c, k) == v, (name, k)
    x, y, ph, want = arrays(e)
    if ph is not None:
        got = O.rotate(c, x, y, ph)
    else:
        got = O.topolar(c, x, y)
    assert got[0].tolist() == want[0].tolist(), name
    assert got[1].tolist() == want[1].tolist(), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases()))
def test_gpu_reproduces_the_surveyors_vectors(name):
    import cordic_amd as ca
    from gpu_util import gpu_p2r, gpu_r2p, gpu_plan_p2r
    e = cases()[name]
    d = parse_args(e["args"])
    cfg = ca.Config.from_cli(d["mode"], d["iw"], d["ow"], d["xtra"], d["pw"], d["n"])
    x, y, ph, want = arrays(e)
    if ph is None:
        got = gpu_r2p(cfg, x, y)
        assert got[0].tolist() == want[0].tolist()
        assert got[1].tolist() == want[1].tolist()
        return
    # per-sample vectors, padded so the vector kernels (4 samples per lane) run
    rep = 64
    gx, gy = gpu_p2r(cfg, np.tile(x, rep), np.tile(y, rep), np.tile(ph, rep))
    assert gx.tolist() == np.tile(want[0], rep).tolist()
    assert gy.tolist() == np.tile(want[1], rep).tolist()
    # constant vector through the stateless call and through a plan (seeded)
    plan = ca.Plan(cfg)
    for k in range(len(ph)):
        p = np.full(256, ph[k], dtype=np.uint32)
        for fn in (lambda: gpu_p2r(cfg, int(x[k]), int(y[k]), p),
                   lambda: gpu_plan_p2r(plan, int(x[k]), int(y[k]), p)):
            a, b = fn()
            assert set(a.tolist()) == {int(want[0][k])}, (name, k)
            assert set(b.tolist()) == {int(want[1][k])}, (name, k)
    plan.close()
