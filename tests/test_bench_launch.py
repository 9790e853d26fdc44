"""bench.py --gpus N can never report a different GPU count than it ran on
("Next round" 1): without enough visible GPUs it fails loudly,
a WORLD_SIZE that disagrees with --gpus is refused, and on a GPU box the
self-launch path (re-exec under torch.distributed.run) and the one-process
cordic_group path both produce a line with n_gpus == --gpus."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


_DETAIL_N = [0]


def _detail_path():
    """a fresh --detail path per run (the default, ./bench_detail.json, is for
    the driver's single run)"""
    import tempfile
    _DETAIL_N[0] += 1
    return os.path.join(tempfile.gettempdir(), "bench_detail_test_%d_%d.json"
                        % (os.getpid(), _DETAIL_N[0]))


def run(args, env=None, timeout=600):
    if "--detail" not in args:
        args = list(args) + ["--detail", _detail_path()]
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, text=True,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout)


def test_more_gpus_than_visible_is_a_loud_error():
    r = run(["--gpus", "64"])
    assert r.returncode != 0
    assert re.search(r"--gpus 64 needs 64 visible GPUs, found \d+", r.stderr)
    assert r.stdout.strip() == ""           # no result line at all


def test_world_size_must_equal_gpus():
    r = run(["--gpus", "8"], env={"RANK": "0", "WORLD_SIZE": "4",
                                  "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 8 but WORLD_SIZE=4" in r.stderr
    assert r.stdout.strip() == ""


def test_launch_decisions_and_respawn_command(monkeypatch):
    """The N > 1 launch logic without GPUs: with N GPUs visible a plain
    `--gpus N` becomes `python -m torch.distributed.run --nproc-per-node N
    ... bench.py --gpus N ...` (one rank per GPU, 127.0.0.1 rendezvous), a
    torchrun-started rank is accepted only if WORLD_SIZE == N, and
    --single-process stays in this process."""
    import argparse
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)

    def ns(**kw):
        d = dict(gpus=1, single_process=False, spawn=False)
        d.update(kw)
        return argparse.Namespace(**d)
    assert bench.resolve_launch(ns(gpus=1)) == "direct"
    assert bench.resolve_launch(ns(gpus=1, spawn=True)) == "spawn"
    assert bench.resolve_launch(ns(gpus=8)) == "spawn"
    assert bench.resolve_launch(ns(gpus=8, single_process=True)) == "single-process"
    with pytest.raises(SystemExit, match="needs 9 visible GPUs, found 8"):
        bench.resolve_launch(ns(gpus=9))
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert bench.resolve_launch(ns(gpus=8)) == "torchrun"
    with pytest.raises(SystemExit, match="--gpus 4 but WORLD_SIZE=8"):
        bench.resolve_launch(ns(gpus=4))
    monkeypatch.delenv("RANK")
    monkeypatch.delenv("WORLD_SIZE")

    seen = {}

    def fake_exec(prog, argv, env):
        seen.update(prog=prog, argv=argv, env=env)
        raise RuntimeError("exec")
    monkeypatch.setattr(bench.os, "execvpe", fake_exec)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20",
                                      "--warmup", "5", "--spawn"])
    with pytest.raises(RuntimeError, match="exec"):
        bench.respawn(ns(gpus=8))
    a = seen["argv"]
    assert a[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and a[a.index("--nproc-per-node") + 1] == "8"
    assert a[a.index("--master-addr") + 1] == "127.0.0.1"
    tail = a[a.index(os.path.abspath(BENCH)) + 1:]
    assert tail == ["--gpus", "8", "--steps", "20", "--warmup", "5"]   # --spawn dropped
    assert seen["env"]["BENCH_SELF_SPAWNED"] == "1"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_line  # noqa: E402


def _line(stdout, want_detail=True):
    """The contract: stdout carries ONE line, the JSON record -- nothing else
    (RCCL's version banner and the like go to stderr) -- of at most 4 KB,
    strict JSON, with the contract keys (bench_line.check).  Returns the
    DETAIL record the line points to (what the line is a selection of), after
    checking that the line agrees with it."""
    rows = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(rows) == 1 and rows[0].startswith("{"), stdout
    line = bench_line.check(rows[0])
    if not want_detail:
        return line
    with open(line["detail"]) as f:
        d = json.load(f)
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step"):
        assert line[k] == pytest.approx(d[k], rel=1e-5), k
    assert line["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-5)
    assert line["config"]["samples_per_gpu"] == d["config"]["samples_per_gpu"]
    d["_line"] = line
    return d


def test_the_line_of_round_fives_records_fits():
    """bench_line.render on the records that were LOST in round 5 (the 20.7 KB
    default line the driver could not parse, and the 8-rank line): at most
    4 KB, strict JSON, contract keys, `roofline` and `cpu_baseline` there as
    numbers and tokens."""
    seen = 0
    for rel in ("profiles/bench_r05/default_driver_order.json",
                "profiles/bench_r05/default.json",
                "profiles/r05/bench_8_ranks_one_gpu.json"):
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            text = [ln for ln in f.read().splitlines() if ln.startswith("{")][-1]
        old = json.loads(text)
        seen += 1
        out = bench_line.render(old, "bench_detail.json")
        assert len(out.encode()) <= 4096 < len(text), (rel, len(out))
        line = bench_line.check(out)
        assert line["value"] == pytest.approx(old["value"], rel=1e-5)
        r = line["roofline"]
        assert r["frac"] == pytest.approx(old["roofline"]["frac"], rel=1e-5)
        assert set(r) <= set(bench_line.ROOF)
        assert r["limiter"] in ("power", "hbm", "valu", "latency")
        assert r["placement"] in ("on", "off")
        for v in r.values():
            assert v is None or isinstance(v, (int, float)) or len(v) <= 16
        if "cpu_baseline" in old:
            c = line["cpu_baseline"]
            assert c["cores"] == old["cpu_baseline"]["cores"]
            assert len(c["sample"]) <= 96
        if old["n_gpus"] > 1:
            sc = line["scale"]
            assert sc["compute_only"] == pytest.approx(old["value"], rel=1e-5)
            for v in sc.values():
                assert v is None or isinstance(v, float)
    assert seen, "no committed round-5 record found"


def test_the_line_is_strict_json_and_shrinks_rather_than_overflows():
    inf = float("inf")
    detail = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1,
              "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "int64",
              "data": "synthetic",
              "config": {"workload": "w" * 5000, "kernel": "k" * 5000,
                         "samples_per_gpu": 1},
              "roofline": {"bound": "hbm", "achieved": float("nan"), "peak": 8000.0,
                           "unit": "GB/s", "frac": inf, "traffic": None,
                           "limiter": "power: a long sentence " * 50,
                           "power": {"x": ["y" * 100] * 100}},
              "cpu_baseline": {"value": 1.0, "unit": "u", "cores": 1, "kind": "port",
                               "sample": "s" * 5000, "cpu": "c" * 500}}
    out = bench_line.render(detail, "/tmp/x.json")
    assert len(out) <= 4096
    line = bench_line.check(out)            # NaN / Infinity would raise here
    assert line["roofline"]["achieved"] is None and line["roofline"]["frac"] is None
    assert line["roofline"]["limiter"] == "power"
    assert "power" not in line["roofline"]
    with pytest.raises(ValueError):
        bench_line.check('{"value": NaN}')
    with pytest.raises(ValueError):
        bench_line.check("{" + '"a": 1, ' * 1000 + '"b": 2}')


def test_the_detail_file_is_written_atomically_and_never_over_devnull(tmp_path):
    p = tmp_path / "d.json"
    assert bench_line.write_detail({"a": float("nan"), "b": [1, 2]}, str(p)) == str(p)
    assert json.loads(p.read_text()) == {"a": None, "b": [1, 2]}
    before = os.stat(os.devnull)
    assert bench_line.write_detail({"a": 1}, os.devnull) is None
    after = os.stat(os.devnull)             # not replaced by a regular file
    assert (before.st_ino, before.st_mode) == (after.st_ino, after.st_mode)
    # an unwritable place falls back to /tmp and says where
    got = bench_line.write_detail({"a": 1}, "/proc/nope/bench_detail_test.json")
    assert got == "/tmp/bench_detail_test.json"
    os.unlink(got)


SMALL = ["--steps", "5", "--warmup", "1", "--log2-samples", "22",
         "--no-cpu-baseline", "--no-other-paths", "--no-pmc", "--no-power"]


@pytest.mark.gpu
def test_the_drivers_default_command_prints_one_small_line_quickly(tmp_path):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` exactly as the driver
    runs it (cwd = the repo, no other flag): ONE line of at most 4 KB with
    `roofline` (traffic measured by this run's counter passes) and
    `cpu_baseline`, the full record in ./bench_detail.json, and all of it
    inside a minute (round 5: 20.7 KB after 116 s, lost)."""
    import shutil
    import time
    e = {k: v for k, v in os.environ.items()
         if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CORDIC_SEED_MIN_SAMPLES")}
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "20",
                        "--warmup", "5"], cwd=ROOT, env=e, text=True, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(rows) == 1
    line = bench_line.check(rows[0])
    assert wall < 60.0, (wall, line.get("wall_s"))
    assert line["detail"] == "bench_detail.json"
    with open(os.path.join(ROOT, "bench_detail.json")) as f:
        d = json.load(f)
    assert d["value"] == pytest.approx(line["value"], rel=1e-5)
    assert line["metric"].startswith("Msamples/sec (sin+cos pairs) at 16-stage/32-bit")
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5
    assert line["config"]["samples_per_gpu"] == 1 << 30
    assert line["digest_check"] == {"samples": 1 << 30, "equal": True}
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "valu") and roof["peak"] == 8000.0
    assert roof["placement"] == "on"
    assert roof["limiter"] in ("power", "hbm", "valu", "latency")
    assert 0.3 < roof["frac"] < 1.0
    if shutil.which("rocprofv3"):
        assert 0.99 < roof["traffic_over_algorithmic"] < 1.02
        assert roof["traffic"] == pytest.approx(
            12.0 * (1 << 30) * roof["traffic_over_algorithmic"], rel=1e-4)
        assert 30 < d["roofline"]["valu"]["instr_per_sample"] < 60
        assert d["roofline"]["pmc"]["subject"] == "pmc_subject"
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert 0 < cpu["value_1thread"] <= cpu["value"] * 1.05
    full = line["full_recurrence"]
    assert 0 < full["value_per_gpu"] < line["value"]
    assert d["full_recurrence_kernel"]["outputs_identical_to_seeded_kernel"] is True
    # where the time of the command went, phase by phase (detail file)
    assert sum(d["phases_s"].values()) <= d["wall_s"] + 0.5
    assert "other_paths" not in d               # that is --full


@pytest.mark.gpu
def test_self_spawn_path_one_gpu():
    r = run(["--gpus", "1", "--spawn"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["launch"]["world_size"] == 1
    assert "self-spawned" in d["launch"]["mode"]
    assert d["bit_exact_vs_oracle"] is True
    assert d["digest_check"]["equal"] is True
    assert d["full_recurrence_kernel"]["outputs_identical_to_seeded_kernel"]


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["--spawn", "--single-process"])
def test_gather_goes_through_the_c_abi(layout):
    """--gather on ONE GPU: RCCL send/recv of the C++ layer in the
    process-per-GPU layout (a world of one here), peer copies in the
    one-process layout; what arrives equals what the shards hold AND what the
    oracle computes for the whole job."""
    r = run(["--gpus", "1", layout, "--gather"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    key = "rccl" if layout == "--spawn" else "peer"
    g = d["gather"][key]
    assert "error" not in g, g
    assert g["outputs_identical"] is True and g["root_digest_equals_oracle"] is True
    assert g["ms"] > 0 and g["ms_chunks1"] > 0 and g["chunks"] == 8
    assert g["GBps_into_root"] > 0 and g["model_ms"] == 0.0     # no remote shard
    assert ("set_gather_rccl" if layout == "--spawn" else "set_gather)") in g["mode"]
    sc = d["scale"]
    assert sc["compute_only"]["Msamples_per_s"] == d["value"]
    assert sc["compute_plus_gather"][key]["ms_per_step"] == g["ms"]


@pytest.mark.gpu
def test_multi_process_run_also_measures_the_one_process_layer():
    """What a >1-GPU torchrun line carries: rank 0 runs `bench.py
    --single-process` as a guarded subprocess while the other ranks wait on
    the host (forced here on one GPU)."""
    r = run(["--gpus", "1", "--spawn"] + SMALL,
            env={"BENCH_FORCE_SINGLE_CHECK": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    sp = d["single_process_cordic_group"]
    assert "error" not in sp, sp
    assert sp["n_gpus"] == 1 and sp["bit_exact_vs_oracle"] is True
    assert sp["digest_equals_multi_process_run"] is True


@pytest.mark.gpu
def test_single_process_path_and_direct_agree():
    a = _line(run(["--gpus", "1", "--single-process"] + SMALL).stdout)
    b = _line(run(["--gpus", "1", "--copy-probe"] + SMALL).stdout)
    assert a["n_gpus"] == b["n_gpus"] == 1
    assert a["digest"] == b["digest"]
    assert a["bit_exact_vs_oracle"] and b["bit_exact_vs_oracle"]
    assert "copy_frac" in b["roofline"]


@pytest.mark.gpu
def test_traffic_is_measured_in_the_same_run():
    """roofline.traffic comes from two rocprofv3 --pmc passes of this very
    command: algorithmic bytes within 1 %."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not installed")
    d = _line(run(["--gpus", "1", "--steps", "5", "--warmup", "1",
                   "--log2-samples", "26", "--no-cpu-baseline",
                   "--no-other-paths"]).stdout)
    r = d["roofline"]
    assert "error" not in r["pmc"], r["pmc"]
    assert 0.99 < r["traffic_over_algorithmic"] < 1.02


@pytest.mark.gpu
def test_power_and_clock_are_sampled_in_the_same_run():
    """roofline.power: socket power and shader clock from the device's hwmon
    files while the kernel runs (timed region + two more seconds)."""
    import glob
    if not glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        pytest.skip("no amdgpu hwmon files in this container")
    args = ["--steps", "20", "--warmup", "2", "--log2-samples", "26",
            "--no-cpu-baseline", "--no-other-paths", "--no-pmc"]
    d = _line(run(["--gpus", "1"] + args).stdout)
    pw = d["roofline"]["power"]
    assert 500 <= pw["limit_w"] <= 3000
    su = pw["sustained"]
    assert su["samples"] >= 50
    assert 50 < su["socket_w_median"] <= pw["limit_w"] * 1.05
    assert 90 <= su["sclk_mhz_median"] <= 2500


@pytest.mark.gpu
def test_two_gpus_on_a_one_gpu_box_is_refused():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has several GPUs")
    r = run(["--gpus", "2"] + SMALL)
    assert r.returncode != 0 and "needs 2 visible GPUs, found 1" in r.stderr


def test_power_sampler_host_logic(tmp_path, monkeypatch):
    """PowerSampler on a fake hwmon directory: units, windows, limit."""
    import importlib.util
    import time as _time
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    hw = tmp_path / "hwmon0"
    hw.mkdir()
    (hw / "power1_input").write_text("1396000000\n")     # microwatts
    (hw / "freq1_input").write_text("2145000000\n")      # Hz
    (hw / "power1_cap").write_text("1400000000\n")
    monkeypatch.setattr(bench.PowerSampler, "_find",
                        classmethod(lambda cls, device: str(hw)))
    s = bench.PowerSampler(0, period=0.001)
    t0 = _time.perf_counter()
    s.start()
    _time.sleep(0.05)
    (hw / "power1_input").write_text("1400000000\n")
    _time.sleep(0.05)
    s.stop()
    t1 = _time.perf_counter()
    w = s.window(t0, t1)
    assert w["samples"] >= 10
    assert w["socket_w_max"] == 1400.0 and 1396.0 <= w["socket_w_median"] <= 1400.0
    assert w["sclk_mhz_median"] == 2145.0 and s.limit_w() == 1400.0
    assert s.window(t1 + 1, t1 + 2) is None
    # no hwmon files at all (this container): the sampler reports nothing
    monkeypatch.undo()
    import glob
    if not glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        assert bench.PowerSampler(0).dir is None


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["cfg4", "cfg3", "p2rxy", "qtrtbl24"])
def test_every_line_says_which_ceiling_binds(workload):
    """roofline.achieved (HBM) AND valu_fraction, with the
    evidence which bound is hit -- instruction count from this run's own
    SQ_INSTS_VALU pass, clock from this run's hwmon samples."""
    r = run(["--workload", workload, "--steps", "6", "--warmup", "2",
             "--log2-samples", "24", "--no-cpu-baseline", "--no-other-paths",
             "--no-copy-probe", "--pmc-counters", "SQ_INSTS_VALU"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    roof = d["roofline"]
    assert d["bit_exact_vs_oracle"] is True
    assert roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    if workload == "qtrtbl24":
        assert roof["bound"] == "hbm" and "lds mode 3" in d["config"]["kernel"]
        return
    v = roof["valu"]
    assert "rocprofv3 --pmc pass of this run" in v["instr_source"]
    # (the seeded kernel builds its table in every block: at 2^24 samples that
    # prologue adds ~20 % to cfg4's 85 instructions per sample with the
    # direction tails)
    want = {"cfg4": (85, 115), "cfg3": (150, 175), "p2rxy": (98, 115)}[workload]
    assert want[0] < v["instr_per_sample"] < want[1], v
    # two VALU figures: the same instruction count at
    # the machine's full issue rate (2 cycles per wave-instruction per
    # SIMD-32), and this instruction MIX at what its opcodes cost
    assert roof["valu_fraction"] == v["frac"]
    assert roof["valu_issue_fraction"] == v["issue_fraction"]
    assert v["issue_fraction"] == pytest.approx(
        v["achieved_Tinstr_per_s"] * 1e12 / 64 * 2.0
        / (1024 * v["sclk_ghz"] * 1e9))
    assert v["issue_fraction"] < v["frac"] <= 2.0 * v["issue_fraction"] * 1.001
    assert roof["bound"] in ("hbm", "valu")
    assert roof["bound"] == ("hbm" if roof["frac"] >= roof["valu_fraction"]
                             else "valu")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("workload,ranks", [("cfg2", 2), ("cfg4", 3), ("cfg3", 2),
                                            ("cfg5", 2), ("p2rxy", 2),
                                            ("cfg4", 8)])      # the node's rank count
def test_the_drivers_multi_rank_command_on_one_gpu(workload, ranks):
    """The command the driver runs for N > 1 --
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N
            --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    -- executed for real with N ranks SHARING device 0 (BENCH_TEST_SHARE_GPU:
    gloo process group; RCCL refuses two ranks on one device).  Everything
    rank-dependent in bench.py runs: RANK / WORLD_SIZE, shard s of N by global
    index, max-over-ranks timing, whole-job `value`, the digest summed over
    the ranks -- which must equal the ORACLE's digest of the whole job."""
    import numpy as np
    import oracle_lib as O
    from gpu_util import cpu_digest
    lg = 20
    shim = os.path.join(ROOT, "tests", "rccl_shim", "librccl_shim.so")
    if not os.path.exists(shim):
        pytest.skip("RCCL shim not built")
    # (the ranks share one device, which real RCCL refuses: the group's RCCL
    # entry points come from tests/rccl_shim -- everything above them, the
    # C++ forwarding included, is the product path)
    e = dict(os.environ, BENCH_TEST_SHARE_GPU="1", CORDIC_RCCL_LIB=shim,
             HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", str(ranks),
           "--workload", workload, "--steps", "4", "--warmup", "1",
           "--log2-samples", str(lg), "--no-cpu-baseline", "--no-other-paths",
           "--no-pmc", "--no-power", "--no-copy-probe", "--detail", _detail_path()]
    r = subprocess.run(cmd, env=e, text=True, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == ranks and d["scaling"] == "weak"
    if "launch" in d:       # the cordic_group workloads (p2rxy runs stateless)
        assert d["launch"]["world_size"] == ranks
        assert "TEST" in d["launch"]["mode"]
        assert len(d["launch"]["per_rank_Msamples_per_s"]) == ranks
    n = 1 << lg
    assert d["value"] == pytest.approx(
        ranks * n * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6)
    assert d["bit_exact_vs_oracle"] is True
    # the whole job on the CPU: n_total samples by GLOBAL index
    n_total = ranks * n
    idx = np.arange(n_total, dtype=np.uint64)
    amp = 2 ** 31 - 1
    if workload in ("cfg2", "cfg4"):
        sh, ns = (2, 16) if workload == "cfg2" else (0, 24)
        c = O.config_cli(O.P2R, 32, 32, 2, 32, ns)
        ph = ((idx << np.uint64(sh)) & np.uint64(0xffffffff)).astype(np.uint32)
        a, b = O.rotate(c, amp, 0, ph)
    elif workload == "cfg5":
        c = O.config_cli(O.P2R, 32, 32, 2, 32, 16)
        ph = ((idx * np.uint64(0x01234567)) & np.uint64(0xffffffff)).astype(np.uint32)
        a, b = O.rotate(c, amp, 0, ph)
    else:
        i32 = idx.astype(np.uint32)

        def ramp(mul, bits):
            v = ((i32 * np.uint32(mul)) >> np.uint32(8)).astype(np.int64)
            v &= (1 << bits) - 1
            return ((v ^ (1 << (bits - 1))) - (1 << (bits - 1))).astype(np.int32)
        if workload == "cfg3":
            c = O.config_cli(O.R2P, 24, 24, 2, -1, 20)
            a, b = O.topolar(c, ramp(0x9E3779B1, 24), ramp(0x85EBCA77, 24))
        else:
            c = O.config_cli(O.P2R, 32, 32, 2, 32, 16)
            ph = ((idx << np.uint64(2)) & np.uint64(0xffffffff)).astype(np.uint32)
            a, b = O.rotate(c, ramp(0x9E3779B1, 32), ramp(0x85EBCA77, 32), ph)
    want = (cpu_digest(a, 0) + cpu_digest(b, 1 << 40)) % 2 ** 64
    assert int(d["digest"], 16) == want, (d["digest"], "%016x" % want)
    # every rank compared ITS shard with the oracle (threaded orc_digest); the
    # line carries the reduced verdict over all ranks' samples
    dc = d["digest_check"]
    assert dc["equal"] is True and dc["samples"] == n_total and dc["ranks"] == ranks
    assert int(dc["oracle"], 16) == want
    if "single_process_cordic_group" in d:
        sp = d["single_process_cordic_group"]
        assert "error" not in sp, sp
    if "launch" not in d:
        return
    # the DEFAULT multi-rank line carries both scalings of
    # SURVEY 8(e) -- compute only (`value`) and compute + the final gather,
    # over the C++ layer's RCCL send / recv between the ranks and over peer
    # copies in the embedded one-process run -- and what arrives at the root
    # has the ORACLE's digest of the whole job
    g = d["gather"]
    for key in ("rccl", "peer"):
        assert "error" not in g[key], g[key]
        assert g[key]["outputs_identical"] is True
        assert g[key]["ms"] > 0 and g[key]["ms_chunks1"] > 0
        assert g[key]["chunks"] == 8 and g[key]["model_ms"] > 0
        assert g[key]["GBps_into_root"] == pytest.approx(
            n_total * 8 / (g[key]["ms"] * 1e-3) / 1e9)
    assert g["rccl"]["root_digest_equals_oracle"] is True
    assert int(g["rccl"]["root_digest"], 16) == want
    assert "set_gather_rccl" in g["rccl"]["mode"]
    assert "one-process" in g["peer"]["measured_by"]
    sc = d["scale"]
    assert sc["compute_only"]["Msamples_per_s"] == d["value"]
    for key in ("rccl", "peer"):
        assert sc["compute_plus_gather"][key]["ms_per_step"] == g[key]["ms"]
        assert sc["compute_plus_gather"][key]["Msamples_per_s"] == pytest.approx(
            n_total / (g[key]["ms"] * 1e-3) / 1e6)
    # ... and the printed line carries them as bare numbers (<= 4 KB at any
    # rank count: _line checked that)
    ls = d["_line"]["scale"]
    assert ls["compute_only"] == pytest.approx(d["value"], rel=1e-5)
    assert ls["compute_plus_gather"] == pytest.approx(
        sc["compute_plus_gather"]["rccl"]["Msamples_per_s"], rel=1e-5)
    assert ls["compute_plus_gather_peer"] == pytest.approx(
        sc["compute_plus_gather"]["peer"]["Msamples_per_s"], rel=1e-5)


def _torchrun(ranks, extra, env, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", str(ranks)] + extra
    if "--detail" not in cmd:
        cmd += ["--detail", _detail_path()]
    e = dict(os.environ, **env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    return subprocess.run(cmd, env=e, text=True, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


QUIET = ["--no-cpu-baseline", "--no-other-paths", "--no-pmc", "--no-power",
         "--no-copy-probe"]


@pytest.mark.gpu
def test_a_gather_that_cannot_run_is_a_labelled_error_not_a_lost_line():
    """Two ranks on ONE device with the REAL librccl: RCCL refuses (or never
    completes) the communicator.  The line still comes out, complete, with
    `gather.rccl.error` -- what the first contact with an 8-GPU node must
    never lose is the compute-only measurement."""
    r = _torchrun(2, ["--workload", "cfg4", "--steps", "4", "--warmup", "1",
                      "--log2-samples", "20", "--gather-limit", "60",
                      "--no-single-process-check"] + QUIET,
                  {"BENCH_TEST_SHARE_GPU": "1", "CORDIC_RCCL_LIB": ""})
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["bit_exact_vs_oracle"] is True and d["digest_check"]["equal"] is True
    assert "error" in d["gather"]["rccl"], d["gather"]


@pytest.mark.gpu
def test_a_stalled_gather_is_cut_off_and_the_line_survives():
    """The transport stalls (the RCCL stand-in delays every exchange by 20 s)
    and the phase has 6 s: rank 0 prints what it has with `gather.rccl.error:
    timed out`, every rank exits 0 (LineGuard)."""
    shim = os.path.join(ROOT, "tests", "rccl_shim", "librccl_shim.so")
    if not os.path.exists(shim):
        pytest.skip("RCCL shim not built")
    r = _torchrun(2, ["--workload", "cfg2", "--steps", "4", "--warmup", "1",
                      "--log2-samples", "20", "--gather-limit", "6",
                      "--no-single-process-check"] + QUIET,
                  {"BENCH_TEST_SHARE_GPU": "1", "CORDIC_RCCL_LIB": shim,
                   "CORDIC_SHIM_DELAY_MS": "20000",
                   "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["digest_check"]["equal"] is True
    assert "timed out" in d["gather"]["rccl"]["error"]
    assert "exceeded 6 s" in r.stderr


@pytest.mark.gpu
def test_one_rank_under_torchrun_measures_what_the_direct_run_measures():
    """N = 1 through the driver's multi-rank command (process group, host
    group, guards) against the plain N = 1 run.  Which arrays a process gets
    decides a few per cent of the rate (include/cordic_amd.h, "Placement"), so
    the two PROCESSES are not compared with each other: in each of them the
    whole-job `value` (host clock around barrier + synchronize) has to agree
    with the kernels' own HIP-event time within 2 % -- the rendezvous must not
    leak into the timed region."""
    # (BASELINE's size and 100 steps: behind the rank-to-rank barrier that
    # opens the timed region the first step runs ~0.3 ms long -- the device
    # idled through the collective -- which is 1.7 % of a 20-step run and
    # 0.15 % of this one; profiles/r05/torchrun_vs_direct.txt)
    args = ["--steps", "100", "--warmup", "5", "--log2-samples", "30",
            "--no-full-digest"] + QUIET
    for r in (_torchrun(1, args, {}), run(["--gpus", "1"] + args)):
        assert r.returncode == 0, r.stderr[-2000:]
        d = _line(r.stdout)
        by_events = (1 << 30) / d["roofline"]["kernel_ms_avg"] / 1e3
        assert abs(d["value"] - by_events) / by_events < 0.02, (d["value"], by_events)


def test_line_guard_prints_the_line_and_exits_zero(tmp_path):
    """LineGuard without a GPU: a phase that outlives its limit -> rank 0
    writes the record it has (patched by on_timeout) and the process ends with
    status 0; a phase that finishes in time changes nothing."""
    prog = tmp_path / "guard.py"
    prog.write_text(
        "import sys, time, json\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "g = bench.LineGuard(0)\n"
        "g.line = {'value': 1.0, 'gather': {}}\n"
        "g.on_timeout = lambda ph, lim: g.line['gather'].update({ph: {'error': 'timed out'}})\n"
        "g.arm('quick', 5.0); time.sleep(0.1); g.disarm()\n"
        "g.arm('rccl', 0.3)\n"
        "time.sleep(30)\n"
        "print('never')\n" % ROOT)
    r = subprocess.run([sys.executable, str(prog)], text=True, timeout=120,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(rows) == 1
    d = json.loads(rows[0])
    assert d == {"value": 1.0, "gather": {"rccl": {"error": "timed out"}}}
    # (a guard without a --detail path prints the record as it is; bench.py's
    # own guard goes through bench_line.publish)
    assert "'rccl' exceeded 0 s" in r.stderr
    # a rank other than 0 leaves quietly (after rank 0 had time to write)
    prog.write_text(prog.read_text().replace("LineGuard(0)", "LineGuard(3)"))
    r = subprocess.run([sys.executable, str(prog)], text=True, timeout=120,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout.strip() == ""
