"""What bench.py says HOLDS a kernel (roofline.limiter, tools/bench_valu.py) and
how it reads the SMU's throttle accumulators (tools/bench_power.py): the rule
order -- same-run copy ceiling, then the power limiter, then VALU issue, then
latency -- on synthetic inputs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_power  # noqa: E402
import bench_valu  # noqa: E402


def _acc(acc, ppt):
    return {"acc": acc, "ppt": ppt, "prochot": 0, "socket_thermal": 0,
            "vr_thermal": 0, "hbm_thermal": 0}


def test_throttle_fraction_is_against_the_busy_time():
    # 2.3 s between the reads, 2.0 s of them busy, 1.5 s under the PPT limiter
    d = bench_power.Throttle.delta(_acc(1000, 10), _acc(3300, 1510), 2000.0)
    assert d["window_ms"] == 2300 and d["ppt_ms"] == 1500
    assert d["ppt_frac"] == 0.75
    assert d["socket_thermal_ms"] == 0
    # more residency than busy time (tick granularity): clamped
    assert bench_power.Throttle.delta(_acc(0, 0), _acc(2100, 2050), 2000.0)["ppt_frac"] == 1.0
    assert bench_power.Throttle.delta(None, _acc(1, 1), 10.0) is None


def test_a_device_without_the_smi_library_reads_none():
    t = bench_power.Throttle(0)
    if t.h is None:                     # (this container: no GPU)
        assert t.read() is None


def _power(watts, ppt_frac=None, cap=1400.0):
    p = {"limit_w": cap,
         "sustained": {"socket_w_median": watts, "sclk_mhz_median": 2100.0}}
    if ppt_frac is not None:
        p["throttle"] = {"ppt_frac": ppt_frac}
    p["at_cap"] = bool(watts >= 0.985 * cap or (ppt_frac or 0) >= 0.5)
    return p


def _roof(frac, copy_frac):
    return {"frac": frac, "copy_frac": copy_frac}


def test_limiter_rule_order():
    pm = {"valu_instr_per_sample": 160.0, "valu_int64_per_sample": 67.0}
    # at the copy ceiling: hbm, whatever the power says
    r = bench_valu.add_valu(_roof(0.85, 0.857), 566e9, {"valu_instr_per_sample": 46.0},
                            _power(1399.0, 0.9), None, "cfg2")
    assert r["limiter"] == "hbm" and r["limiter_note"].startswith("hbm:")
    # far from both ceilings with the PPT limiter holding the clock: power,
    # also when the socket reads under its cap
    r = bench_valu.add_valu(_roof(0.44, 0.83), 221e9, pm, _power(1291.0, 0.72), None, "cfg3")
    assert r["limiter"] == "power"
    assert r["limiter_note"].startswith("power: PPT limiter active 72 %")
    assert "1291 W of 1400" in r["limiter_note"]
    assert r["bound"] == "valu"         # the nearer ceiling, as before
    # no throttling: the issue port if it is busy enough ...
    r = bench_valu.add_valu(_roof(0.44, 0.83), 221e9, pm, _power(900.0, 0.0), None, "cfg3")
    assert r["limiter"] == "valu"
    # ... else latency
    r = bench_valu.add_valu(_roof(0.30, 0.83), 100e9, pm, _power(900.0, 0.0), None, "cfg3")
    assert r["limiter"] == "latency"
