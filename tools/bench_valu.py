"""bench_valu.py -- the VALU side of a bench line's roofline block.

SURVEY.md 8(d): "report roofline.achieved (HBM) AND valu_fraction".  Two
figures, because "VALU-bound" can mean two things:

  valu_issue_fraction   every VALU instruction priced at the MACHINE's issue
                        rate: a wave-instruction occupies its SIMD-32 for 2
                        cycles (MI355X_MICROARCH.md).  1.0 = no formulation of
                        the same instruction COUNT could run faster.
  valu_fraction         every instruction priced at what ITS opcode costs
                        (tools/valu_microbench.hip: VOP2 32-bit and
                        v_bitop3_b32 issue in 2 cycles, every other VOP3, the
                        64-bit ops and v_mad_i64_i32 in 4), weighted by the hot
                        loop's own histogram (tools/dump_isa.py ->
                        profiles/isa/hot_loops.json) and scaled to the
                        instruction count this run measured (SQ_INSTS_VALU).
                        1.0 = THIS instruction mix cannot issue faster; the gap
                        to valu_issue_fraction is the price of the half-rate
                        opcodes, most of it v_mad_i64_i32
                        (int64_share_of_issue_time).

Both at the shader clock the kernel actually held (hwmon, same run).
"""
import json
import os
import re

from bench_common import ROOT

N_SIMD, WAVE_LANES = 1024, 64
FULL_RATE_CYCLES = 2.0          # MI355X_MICROARCH.md: wave64 on a SIMD-32
HALF_RATE_CYCLES = 4.0
# round 1-4 figure, kept as the fallback for kernels without a histogram: in
# the mixed stream of a micro-rotation a SIMD with 8 resident waves issues one
# VALU wave-instruction every 3.65 cycles (tools/stage_microbench.hip)
MIXED_STREAM_CYCLES = 3.65
SCLK_MAX_GHZ = 2.4

# which listing of profiles/isa/ is the workload's hot kernel
LISTING_OF = {"cfg2": "rotator_seeded_lj29_16", "cfg4": "rotator_seeded_lj29_24",
              "cfg5": "rotator_seeded_lj29_16_nco", "p2rxy": "rotator_xydir_lj29_16",
              "ddc": "rotator_xydir_lj29_16", "cfg3": "topolar_lj_20",
              "cfg2_noseed": "rotator_unrolled_lj29_16",
              # no listing of their own: the 32-bit opcode mix of the same kernel
              # family (the executed counts are their own)
              "cfg5seq": "rotator_seeded_lj29_16_nco", "cfg1": "rotator_seeded_lj29_16",
              "nat16": "rotator_seeded_lj29_24", "nat24": "rotator_seeded_lj29_24",
              "nat32": "rotator_seeded_lj29_24", "natr2p24": "topolar_lj_20"}

_MICRO = re.compile(r"^(\w+)\s+[\d.]+ ms\s+[\d.]+ T lane-ops/s\s+([\d.]+) cyc")


def opcode_cycles():
    """{microbenchmark name: cycles per wave-instruction per SIMD}: measured
    back to back (tools/valu_microbench.hip), snapped to the two classes the
    hardware has -- full rate (2 cycles) below 3.2 measured cycles at the
    nominal 2.4 GHz, half rate (4) above; v_cndmask_b32 (VOP2 form, ~22) keeps
    its own figure (the kernels never use it).  The newest committed table
    wins."""
    table = {}
    for rel in ("profiles/valu_microbench_r01.txt",
                "profiles/r05/valu_microbench.txt"):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                for ln in f:
                    m = _MICRO.match(ln)
                    if m:
                        c = float(m.group(2))
                        table[m.group(1)] = (FULL_RATE_CYCLES if c < 3.2 else
                                             HALF_RATE_CYCLES if c < 8 else c)
        except OSError:
            pass
    return table


def cycles_of(mnemonic, table):
    """cycles of one v_* mnemonic of a listing"""
    op = mnemonic[2:] if mnemonic.startswith("v_") else mnemonic
    for name in (op, re.sub(r"_e(32|64)$", "", op)):
        if name in table:
            return table[name]
    base = re.sub(r"_e(32|64)$", "", op)
    # 32-bit two-operand moves / logic the microbenchmark has no line for
    if base in ("mov_b32", "and_b32", "subrev_u32", "max_u32", "min_u32"):
        return table.get("or_b32", FULL_RATE_CYCLES) if base != "min_u32" \
            else table.get("min_i32", HALF_RATE_CYCLES)
    return HALF_RATE_CYCLES          # VOP3, 64-bit, compares, lane ops


def hot_loop(workload):
    try:
        with open(os.path.join(ROOT, "profiles", "isa", "hot_loops.json")) as f:
            db = json.load(f)
    except (OSError, ValueError):
        return None, None
    stem = LISTING_OF.get(workload)
    return (db.get(stem), stem) if stem else (None, None)


_IS64 = re.compile(r"^v_(mad_[iu]64_[iu]32|lshl_add_u64|ashrrev_i64|lshrrev_b64|"
                   r"lshlrev_b64|mov_b64|add_co_u32|addc_co_u32|cmp_\w+_[iu]64)")


def opcode_model(workload, instr_per_sample, int64_per_sample=None):
    """Issue cycles per sample, per-opcode priced.  The listing's hot loop
    holds code a given core does not execute (both bodies of a kernel that
    chooses per row, three rounding variants), so its histogram is not taken
    at face value: the EXECUTED count comes from this run's SQ_INSTS_VALU, the
    executed share of 64-bit integer instructions (v_mad_i64_i32 above all:
    half rate) from SQ_INSTS_VALU_INT64 where that pass ran, and the listing
    only says what fraction of the remaining 32-bit instructions is half rate
    (VOP3 forms: v_lshl_add_u32, v_bfe, v_mad_u32_u24, v_and_or_b32 ...)."""
    loop, stem = hot_loop(workload)
    if not loop or not loop.get("hist"):
        return None
    table = opcode_cycles()
    valu = {k: v for k, v in loop["hist"].items() if k.startswith("v_")}
    n_static = float(sum(valu.values()))
    if not n_static:
        return None
    w64 = {k: v for k, v in valu.items() if _IS64.match(k)}
    w32 = {k: v for k, v in valu.items() if not _IS64.match(k)}
    n64_static, n32_static = float(sum(w64.values())), float(sum(w32.values()))
    half32 = sum(v for k, v in w32.items()
                 if cycles_of(k, table) > FULL_RATE_CYCLES) / max(n32_static, 1.0)
    cyc32 = (sum(v * cycles_of(k, table) for k, v in w32.items())
             / max(n32_static, 1.0))
    cyc64 = (sum(v * cycles_of(k, table) for k, v in w64.items())
             / max(n64_static, 1.0)) if n64_static else HALF_RATE_CYCLES
    if int64_per_sample is not None and int64_per_sample <= instr_per_sample:
        n64, src64 = int64_per_sample, "SQ_INSTS_VALU_INT64 of this run"
    else:
        n64 = instr_per_sample * n64_static / n_static
        src64 = "share in the listing (no SQ_INSTS_VALU_INT64 pass)"
    n32 = instr_per_sample - n64
    cyc = n64 * cyc64 + n32 * cyc32
    return {"listing": "profiles/isa/%s.s" % stem,
            "static_instr_per_sample": n_static / loop["samples_per_pass"],
            "int64_instr_per_sample": n64, "int64_source": src64,
            "int32_half_rate_share": half32,
            "cycles_per_instruction": cyc / instr_per_sample,
            "cycles_per_sample": cyc,
            "int64_share_of_issue_time": n64 * cyc64 / cyc,
            "source": "executed counts (SQ_INSTS_VALU, SQ_INSTS_VALU_INT64) x "
                      "per-opcode cycles (tools/valu_microbench.hip: 2 full "
                      "rate, 4 half rate); the 32-bit mix from the hot-loop "
                      "histogram (tools/dump_isa.py)"}


def valu_block(samples_per_s, instr_per_sample, sclk_ghz, instr_source,
               sclk_source, workload=None, int64_per_sample=None):
    if not instr_per_sample:
        return None
    clk = sclk_ghz or SCLK_MAX_GHZ
    simd_cycles_per_s = N_SIMD * clk * 1e9
    wave_instr_per_s = samples_per_s * instr_per_sample / WAVE_LANES
    issue = wave_instr_per_s * FULL_RATE_CYCLES / simd_cycles_per_s
    model = (opcode_model(workload, instr_per_sample, int64_per_sample)
             if workload else None)
    if model:
        frac = (samples_per_s * model["cycles_per_sample"] / WAVE_LANES
                / simd_cycles_per_s)
        how = "per-opcode model"
    else:
        frac = wave_instr_per_s * MIXED_STREAM_CYCLES / simd_cycles_per_s
        how = ("no histogram for this kernel: %.2f cycles per instruction, the "
               "mixed-stream microbenchmark (tools/stage_microbench.hip)"
               % MIXED_STREAM_CYCLES)
    out = {"instr_per_sample": instr_per_sample,
           "instr_source": instr_source,
           "sclk_ghz": clk, "sclk_source": sclk_source if sclk_ghz else
           "nominal maximum (no hwmon samples)",
           "achieved_Tinstr_per_s": samples_per_s * instr_per_sample / 1e12,
           "issue_fraction": issue,
           "issue_fraction_what": "every instruction at 2 cycles per "
                                  "wave-instruction per SIMD-32 "
                                  "(MI355X_MICROARCH.md)",
           "frac": frac, "frac_what": how,
           "frac_at_2.4GHz": frac * clk / SCLK_MAX_GHZ}
    if model:
        out["model"] = model
    return out


def add_valu(roof, samples_per_s, pm, power, prof, workload=None):
    """roofline.valu / valu_fraction / valu_issue_fraction / bound from this
    run's SQ_INSTS_VALU pass (or, without one, the committed profile) and this
    run's clock."""
    instr = src = None
    if pm and pm.get("valu_instr_per_sample"):
        instr, src = pm["valu_instr_per_sample"], (
            "SQ_INSTS_VALU x 64 / samples, rocprofv3 --pmc pass of this run")
    elif prof and prof.get("valu_instr_per_sample"):
        instr, src = prof["valu_instr_per_sample"], (
            "committed profile (%s), not re-measured" % prof.get("source"))
    sclk = ssrc = None
    for key in ("sustained", "timed_region"):
        if power and power.get(key) and power[key].get("sclk_mhz_median"):
            sclk = power[key]["sclk_mhz_median"] / 1e3
            ssrc = "hwmon freq1_input median, %s window of this run" % key
            break
    vb = valu_block(samples_per_s, instr, sclk, src, ssrc, workload,
                    (pm or {}).get("valu_int64_per_sample"))
    if vb:
        roof["valu"] = vb
        roof["valu_fraction"] = vb["frac"]
        roof["valu_issue_fraction"] = vb["issue_fraction"]
        # whichever ceiling the kernel sits nearer to
        roof["bound"] = "hbm" if roof["frac"] >= vb["frac"] else "valu"
        # ... and what actually holds it there: the copy
        # ceiling of this run's arrays, VALU issue, or -- for kernels that
        # saturate neither -- the socket's power limit, which sets the clock
        cf = roof.get("copy_frac")
        cap = (power or {}).get("at_cap")
        if cf and roof["frac"] >= 0.95 * cf:
            roof["limiter"] = "hbm: at the same-run copy ceiling (%.3f of %.3f)" % (
                roof["frac"], cf)
        elif cap:
            thr = power.get("throttle") or {}
            held = ("PPT limiter active %.0f %% of the sustained window"
                    % (100 * thr["ppt_frac"])) if "ppt_frac" in thr else "socket at its limit"
            roof["limiter"] = ("power: %s (%.0f W of %.0f), clock %.2f GHz; "
                               "neither HBM (%.2f) nor VALU issue (%.2f) saturated"
                               % (held, power["sustained"]["socket_w_median"],
                                  power.get("limit_w") or 0,
                                  vb["sclk_ghz"], roof["frac"], vb["frac"]))
        elif vb["frac"] >= 0.70:
            roof["limiter"] = ("valu: this instruction mix occupies the issue port "
                               "%.0f %% of the time"
                               % (100 * vb["frac"]))
        else:
            roof["limiter"] = ("latency / LDS: HBM %.2f, VALU issue %.2f, socket "
                               "below its limit" % (roof["frac"], vb["frac"]))
        # the line carries the token in front of the colon; the sentence stays
        # in the detail file
        roof["limiter_note"] = roof["limiter"]
        roof["limiter"] = roof["limiter"].split(":")[0].split(" ")[0]
        roof["bound_note"] = (
            "hbm frac %.3f vs valu_fraction %.3f (this instruction mix at the "
            "clock the power limit allowed) / valu_issue_fraction %.3f (the "
            "same count at the machine's full issue rate)"
            % (roof["frac"], vb["frac"], vb["issue_fraction"]))
    return roof
