"""bench_power.py -- socket power and shader clock of the GPU while the timed
kernels run (hwmon files read by a host thread; nothing is launched)."""
import os
import threading
import time

class PowerSampler(threading.Thread):
    """Socket power and shader clock of one GPU while it works, read by a host
    thread from the amdgpu hwmon files of THAT device (matched by PCI bus id):
    power1_input (microwatts), freq1_input (sclk, Hz), power1_cap (the limit).
    Plain file reads: nothing is launched on the GPU and no tool is started,
    so the timed region is not disturbed.  The CORDIC kernels turn out to run
    at the power limit with the clock below its 2.4 GHz maximum (DESIGN.md
    4.7); this puts the evidence into the bench line itself."""

    def __init__(self, device, period=0.002):
        super().__init__(daemon=True)
        self.period = period
        self.dir = self._find(device)
        self.rows = []                  # (t, watts, sclk MHz)
        self._halt = threading.Event()

    @staticmethod
    def _bus_id(device):
        import ctypes
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
                return None
            return buf.value.decode().lower()
        except OSError:
            return None

    @classmethod
    def _find(cls, device):
        import glob
        cands = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
        cands = [c for c in cands
                 if os.path.exists(os.path.join(c, "power1_input"))
                 and os.path.exists(os.path.join(c, "freq1_input"))]
        bus = cls._bus_id(device)
        for c in cands:
            real = os.path.realpath(os.path.dirname(os.path.dirname(c))).lower()
            if bus and real.endswith(bus):
                return c
        return cands[0] if len(cands) == 1 else None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def run(self):
        while not self._halt.is_set():
            try:
                self.rows.append((time.perf_counter(),
                                  self._read("power1_input") / 1e6,
                                  self._read("freq1_input") / 1e6))
            except (OSError, ValueError):
                pass
            time.sleep(self.period)

    def stop(self):
        self._halt.set()
        self.join()

    def window(self, t0, t1):
        """Statistics of the samples taken in [t0, t1]."""
        w = [r for r in self.rows if t0 <= r[0] <= t1]
        if not w:
            return None
        pw = sorted(r[1] for r in w)
        ck = sorted(r[2] for r in w)
        return {"samples": len(w), "seconds": t1 - t0,
                "socket_w_median": pw[len(pw) // 2], "socket_w_max": pw[-1],
                "sclk_mhz_median": ck[len(ck) // 2], "sclk_mhz_min": ck[0]}

    def limit_w(self):
        try:
            return self._read("power1_cap") / 1e6
        except (OSError, ValueError):
            return None


class Throttle:
    """The SMU's residency accumulators of one GPU (amdsmi gpu_metrics: a
    millisecond counter and, beside it, the milliseconds spent with the PPT
    power limiter / PROCHOT / a thermal limiter holding the clocks down).  Two
    reads around a window say what FRACTION of it the device was throttled and
    by what -- the socket reads 1.28-1.33 kW of its 1.4 kW cap while the
    VALU-bound kernels run, and it is the PPT limiter that keeps the shader
    clock at 2.0-2.2 GHz (profiles/r05/throttle_cfg3.txt)."""

    KEYS = (("acc", "accumulation_counter"), ("ppt", "ppt_residency_acc"),
            ("prochot", "prochot_residency_acc"),
            ("socket_thermal", "socket_thm_residency_acc"),
            ("vr_thermal", "vr_thm_residency_acc"),
            ("hbm_thermal", "hbm_thm_residency_acc"))

    def __init__(self, device):
        self.h = None
        try:
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            bus = PowerSampler._bus_id(device)
            for h in hs:
                try:
                    bdf = str(amdsmi.amdsmi_get_gpu_device_bdf(h)).lower()
                except Exception:
                    continue
                if bus and bdf.endswith(bus[-7:]):
                    self.h = h
            if self.h is None and len(hs) == 1:
                self.h = hs[0]
        except Exception:
            self.h = None

    def read(self):
        if self.h is None:
            return None
        try:
            m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
            out = {}
            for k, name in self.KEYS:
                v = m.get(name)
                out[k] = int(v) if isinstance(v, int) else None
            return out if out["acc"] is not None else None
        except Exception:
            return None

    @staticmethod
    def delta(a, b, busy_ms):
        """what happened between two reads, against the busy_ms of GPU work in
        between (the accumulators run on while the device idles around it)"""
        if not a or not b or b["acc"] is None or a["acc"] is None:
            return None
        d = {"window_ms": b["acc"] - a["acc"], "busy_ms": round(busy_ms, 1),
             "source": "amdsmi gpu_metrics residency accumulators (1 ms ticks), "
                       "read before and after the sustained window"}
        for k in ("ppt", "prochot", "socket_thermal", "vr_thermal", "hbm_thermal"):
            if a.get(k) is not None and b.get(k) is not None:
                d[k + "_ms"] = b[k] - a[k]
        if "ppt_ms" in d and busy_ms > 0:
            d["ppt_frac"] = min(1.0, d["ppt_ms"] / busy_ms)
        return d


def start_power(device, enabled):
    if not enabled:
        return None
    sp = PowerSampler(device)
    if sp.dir is None:
        return None
    sp.throttle = Throttle(device)
    sp.start()
    return sp


def finish_power(sampler, step, sync, t0, elapsed, steps, samples_per_step,
                 seconds=1.0):
    """The timed region is short (the governor is still settling): keep the
    same kernel going for `seconds` more (one by default, two with bench.py
    --full) and sample that as well."""
    if sampler is None:
        return None
    t1 = time.perf_counter()
    ms_step = elapsed / steps
    more = max(1, min(4000, int(seconds / max(ms_step, 1e-6))))
    th = getattr(sampler, "throttle", None)
    th0 = th.read() if th else None
    t1 = time.perf_counter()
    for _ in range(more):
        step()
    sync()
    t2 = time.perf_counter()
    th1 = th.read() if th else None
    sampler.stop()
    power = {"source": "amdgpu hwmon of the device (power1_input, "
                       "freq1_input), host thread, every 2 ms",
             "limit_w": sampler.limit_w(),
             "timed_region": sampler.window(t0, t1),
             "sustained": sampler.window(t1 + (t2 - t1) / 2, t2)}
    if power["sustained"]:
        # for information only: `value` is the K timed steps
        power["sustained"]["steps"] = more
        power["sustained"]["msamples_per_s_local_shards"] = (
            samples_per_step * more / (t2 - t1) / 1e6)
        # At the cap the clock is whatever the power budget allows: a kernel
        # that stalls less then runs at a lower clock, and what raises the
        # rate is less ENERGY per sample (fewer / cheaper instructions, fewer
        # LDS and HBM bytes), not fewer stalls.
        w = power["sustained"].get("socket_w_median")
        thr = Throttle.delta(th0, th1, (t2 - t1) * 1e3)
        if thr:
            power["throttle"] = thr
        if w and power["limit_w"]:
            # at the cap = the socket reads its limit, or the PPT limiter held
            # the clocks for most of the window (it does so from ~0.92 of the cap)
            power["at_cap"] = bool(w >= 0.985 * power["limit_w"]
                                   or (thr or {}).get("ppt_frac", 0) >= 0.5)
            if samples_per_step:
                power["nj_per_sample"] = (
                    w / (power["sustained"]["msamples_per_s_local_shards"] * 1e6) * 1e9)
    return power
