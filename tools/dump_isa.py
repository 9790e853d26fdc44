#!/usr/bin/env python3
"""dump_isa.py -- ISA evidence for the hot kernels (profiles/isa/).

For each kernel named below: the gfx950 disassembly of the shipped code object
(llvm-objdump -d of the bundle inside cordic_amd/csrc/build/*.o), its register
/ LDS / scratch figures from the code-object metadata, and an instruction
histogram of its hot loop (the stretch between the first and the last
full-width global store, i.e. one pass over 4 samples per lane).

    python tools/dump_isa.py            # writes profiles/isa/*
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "cordic_amd", "csrc", "build")
OUT = os.path.join(ROOT, "profiles", "isa")
LLVM = "/opt/rocm/lib/llvm/bin"

# (file stem, object, demangled-name regex, what it is)
KERNELS = [
    ("rotator_seeded_lj29_16", "cordic_inst_seed_lj29.o",
     r"rotator_seeded<cordic_amd::dev::WideLJ<29>, 16, 11, \(cordic_amd::Feed\)0, false, cordic_amd::dev::Io32, false, true, false>",
     "cfg2 headline: seeded p2r, WW 35, 16 stages (5 after the seed: one group of "
     "direction tails on every row)"),
    ("rotator_seeded_lj29_24", "cordic_inst_seed_lj29.o",
     r"rotator_seeded<cordic_amd::dev::WideLJ<29>, 24, 11, \(cordic_amd::Feed\)0, false, cordic_amd::dev::Io32, false, true, false>",
     "cfg4: seeded p2r, 24 stages (13 after the seed: direction tails in groups of "
     "6 and 7 on coherent rows, phase recurrence on the others)"),
    ("rotator_seeded_lj29_16_nco", "cordic_inst_seed_lj29.o",
     r"rotator_seeded<cordic_amd::dev::WideLJ<29>, 16, 11, \(cordic_amd::Feed\)2, false, cordic_amd::dev::Io32, false, true, false>",
     "cfg5: fused NCO + seeded p2r with the group of direction tails, store only"),
    ("rotator_unrolled_lj29_16", "cordic_inst_rot_lj29.o",
     r"rotator_unrolled<cordic_amd::dev::WideLJ<29>, 16, 2, \(cordic_amd::Feed\)0, false, cordic_amd::dev::Io32, false>",
     "cfg2 full recurrence, constant vector"),
    ("rotator_unrolled_lj29_16_xy", "cordic_inst_rot_lj29.o",
     r"rotator_unrolled<cordic_amd::dev::WideLJ<29>, 16, 2, \(cordic_amd::Feed\)1, false, cordic_amd::dev::Io32, false>",
     "p2rxy: per-sample x, y and phase"),
    ("rotator_xydir_lj29_16", "cordic_inst_xydir_lj29.o",
     r"rotator_xydir<29, 16, false>",
     "p2rxy through a plan (round 4): per-sample x, y and phase, directions of "
     "stages 2-16 looked up in three groups of five"),
    ("topolar_lj_20", "cordic_inst_pol_lj.o",
     r"topolar_lj<20, false, cordic_amd::dev::Io32, false, true>",
     "cfg3 r2p, left-justified form (7 instructions per micro-rotation)"),
]


# hot loop of a listing = the innermost loop with this many full-width stores;
# one pass of it covers this many samples per lane.  The seeded kernels' tile
# loop does two rows per rendezvous (4 stores, 8 samples); everything else one
# vector per array (2 stores, 4 samples).
def loop_shape(stem):
    return (4, 8) if stem.startswith("rotator_seeded") else (2, 4)


def sh(*cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw).stdout


def code_object(obj, td):
    path = os.path.join(td, os.path.basename(obj))
    subprocess.run(["cp", obj, path], check=True)
    sh(os.path.join(LLVM, "llvm-objdump"), "--offloading", path, cwd=td)
    for f in os.listdir(td):
        if f.startswith(os.path.basename(obj)) and "gfx950" in f:
            return os.path.join(td, f)
    raise RuntimeError("no gfx950 bundle in " + obj)


def metadata(co, mangled):
    notes = sh(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
    i = notes.find(".name:           " + mangled)
    j = notes.rfind("  - .agpr_count", 0, i) if i >= 0 else -1
    blk = notes[j:notes.find("  - .agpr_count", i)] if j >= 0 else ""
    out = {}
    for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count",
                "sgpr_spill_count", "group_segment_fixed_size",
                "private_segment_fixed_size", "max_flat_workgroup_size"):
        m = re.search(r"\.%s:\s*(\d+)" % key, blk)
        out[key] = int(m.group(1)) if m else None
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    rows = []
    loops = {}
    with tempfile.TemporaryDirectory() as td:
        for stem, obj, pat, what in KERNELS:
            co = code_object(os.path.join(BUILD, obj), td)
            dis = sh(os.path.join(LLVM, "llvm-objdump"), "-d", co)
            dem = sh(os.path.join(LLVM, "llvm-objdump"), "-d", "-C", "--no-show-raw-insn", co)
            # locate the kernel by its demangled name, take the same block of
            # the mangled listing (labels are in the same order)
            heads_d = [(m.start(), m.group(1)) for m in re.finditer(r"^[0-9a-f]+ <(.*)>:$", dem, re.M)]
            heads_m = [(m.start(), m.group(1)) for m in re.finditer(r"^[0-9a-f]+ <(.*)>:$", dis, re.M)]
            idx = [k for k, (_, nm) in enumerate(heads_d) if re.search(pat, nm)]
            if len(idx) != 1:
                sys.exit("%s: %d kernels match %s" % (obj, len(idx), pat))
            k = idx[0]
            mangled = heads_m[k][1]
            end = heads_m[k + 1][0] if k + 1 < len(heads_m) else len(dis)
            body = dis[heads_m[k][0]:end]
            raw = [ln for ln in body.splitlines()[1:] if ln.strip()]
            addr = []
            for ln in raw:
                m = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
                addr.append(int(m.group(1), 16) if m else None)
            lines = [body.splitlines()[0]] + [
                re.sub(r"\s*//.*$", "", ln).rstrip() for ln in raw]
            ins = [ln.split()[0] for ln in lines[1:]]
            # hot loop = the backward branch (simm16 >= 0x8000, target =
            # address + 4 + 4 * (simm16 - 65536)) whose body holds the most
            # VALU instructions
            best = (0, len(ins) - 1, -1)
            want_st, per_pass = loop_shape(stem)
            for i, ln in enumerate(lines[1:]):
                m = re.match(r"\s*s_cbranch_\w+\s+(\d+)", ln)
                if not m or int(m.group(1)) < 0x8000 or addr[i] is None:
                    continue
                target = addr[i] + 4 + 4 * (int(m.group(1)) - 65536)
                b = next((k for k, a in enumerate(addr) if a == target), None)
                if b is None:
                    continue
                n_valu = sum(1 for t in ins[b:i + 1] if t.startswith("v_"))
                n_st = sum(1 for t in ins[b:i + 1]
                           if t.startswith("global_store_dwordx"))
                # one pass stores each output array once per row: a range
                # with more stores spans two loops (a kernel with two work
                # distributions)
                if n_st == want_st and n_valu > best[2]:
                    best = (b, i, n_valu)
            lo, hi = best[0], best[1]
            hot = ins[lo:hi + 1]
            hist = collections.Counter(hot)
            valu = sum(v for t, v in hist.items() if t.startswith("v_"))
            md = metadata(co, mangled)
            with open(os.path.join(OUT, stem + ".s"), "w") as f:
                f.write("; %s\n; %s\n; object %s, llvm-objdump -d --no-show-raw-insn\n"
                        % (what, heads_d[k][1], obj))
                f.write("; registers / memory: %s\n" % md)
                f.write("; hot loop = listing lines %d..%d (one pass: %d samples per lane), "
                        "%d VALU instructions = %.1f per sample\n"
                        % (lo + 2, hi + 2, per_pass, valu, valu / float(per_pass)))
                f.write("; histogram of the hot loop: %s\n\n" % ", ".join(
                    "%s x%d" % kv for kv in hist.most_common()))
                f.write("\n".join(lines) + "\n")
            alloc = ((md["vgpr_count"] or 1) + 7) // 8 * 8
            waves = min(8, 512 // alloc)
            if "seeded" in stem:
                # 136 KiB of dynamic LDS per 1024-thread block: one block,
                # 16 waves, per CU
                waves = min(waves, 4)
            rows.append((stem, what, md, valu / float(per_pass), hist, waves))
            loops[stem] = {"samples_per_pass": per_pass, "valu": valu,
                           "hist": dict(hist), "kernel": heads_d[k][1]}
    # machine-readable twin: tools/bench_valu.py prices these histograms per
    # opcode (roofline.valu.model of every bench line)
    import json
    with open(os.path.join(OUT, "hot_loops.json"), "w") as f:
        json.dump(loops, f, indent=1, sort_keys=True)
    with open(os.path.join(OUT, "README.md"), "w") as f:
        f.write("# ISA of the hot kernels (gfx950, shipped code objects)\n\n"
                "Written by `tools/dump_isa.py` from `cordic_amd/csrc/build/*.o`. "
                "\"VALU / sample\" counts the `v_*` instructions of the hot loop "
                "divided by the samples a lane handles per pass (4; 8 in the "
                "seeded kernels' two-row tile loop); waves / SIMD is "
                "what the VGPR allocation allows (512 VGPRs per SIMD lane, "
                "granule 8, at most 8).\n\n"
                "| listing | kernel | VGPR | SGPR | static LDS B | scratch B | spills | waves/SIMD | VALU / sample | v_mad_i64_i32 | v_bitop3_b32 | v_ashrrev_i32 |\n"
                "|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for stem, what, md, vps, hist, waves in rows:
            f.write("| `%s.s` | %s | %s | %s | %s | %s | %s | %d | %.1f | %d | %d | %d |\n" % (
                stem, what, md["vgpr_count"], md["sgpr_count"],
                md["group_segment_fixed_size"], md["private_segment_fixed_size"],
                (md["vgpr_spill_count"] or 0) + (md["sgpr_spill_count"] or 0),
                waves, vps, hist.get("v_mad_i64_i32", 0),
                hist.get("v_bitop3_b32", 0), hist.get("v_ashrrev_i32_e32", 0)))
    print(open(os.path.join(OUT, "README.md")).read())


if __name__ == "__main__":
    main()
