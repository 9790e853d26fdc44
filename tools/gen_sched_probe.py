#!/usr/bin/env python3
"""gen_sched_probe.py -- writes tools/sched_probe.hip: does the ORDER of the
r2p micro-rotation's seven instructions matter on gfx950?

The converter's hot loop (profiles/isa/topolar_lj_20.s) executes 160 VALU
instructions per sample at 3.8 cycles each where the per-opcode prices (2 / 4
cycles, tools/valu_microbench.hip) add up to 2.9; the compiler's schedule
carries 84 `s_nop 0` per 637 VALU instructions (between a v_bitop3_b32 or a
v_mad_i64_i32 and the instruction that reads its result).  This probe runs the
SAME seven opcodes with the SAME dependencies (4 samples per lane, stages
1..20) as one inline-asm block per variant -- inside an asm block the compiler
inserts nothing -- and times them from 8 / 4 / 2 / 1 waves per SIMD:

  sample_major      the 7 instructions of sample 0, then of sample 1, ...
                    (every consumer directly behind its producer), no s_nop
  sample_major_nop  the same with the compiler's s_nop 0 placement
  instr_major       instruction i of samples 0..3, then instruction i+1
                    (no consumer within 3 slots of its producer)
  mads_last         the 32-bit work of all four samples first, then the 12 mads

All variants must agree bit for bit (the printed checksum): a wrong one would
mean the hardware NEEDS the wait state the compiler pads.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NS = 4                      # samples per lane
STAGES = list(range(1, 21))
ITERS = 256
SKEW = os.environ.get('SCHED_PROBE_SKEW', '0') == '1'

# registers: sample s owns v[10+12*s ...]: X lo,hi  Y lo,hi  P lo,hi  m mn sx sy
def R(s, name):
    base = 10 + 12 * s
    off = {"xl": 0, "xh": 1, "yl": 2, "yh": 3, "pl": 4, "ph": 5,
           "m": 6, "mn": 7, "sx": 8, "sy": 9}[name]
    return base + off


def pair(s, name):
    lo = R(s, name + "l")
    return "v[%d:%d]" % (lo, lo + 1)


C1, C2 = 4, 5               # v4, v5: the bitop3 constants


def instrs(s, k):
    """the seven instructions of one micro-rotation of sample s, in program order,
    as (text, kind) with kind in {"b", "x", "a", "m"}"""
    return [
        ("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0xec" % (R(s, "m"), R(s, "yh"), C1, C2), "b"),
        ("v_xor_b32_e32 v%d, v%d, v%d" % (R(s, "mn"), R(s, "m"), C2), "x"),
        ("v_ashrrev_i32_e32 v%d, %d, v%d" % (R(s, "sx"), k, R(s, "xh")), "a"),
        ("v_ashrrev_i32_e32 v%d, %d, v%d" % (R(s, "sy"), k, R(s, "yh")), "a"),
        ("v_mad_i64_i32 %s, vcc, v%d, v%d, %s" % (pair(s, "x"), R(s, "sy"), R(s, "m"), pair(s, "x")), "m"),
        ("v_mad_i64_i32 %s, vcc, v%d, v%d, %s" % (pair(s, "y"), R(s, "sx"), R(s, "mn"), pair(s, "y")), "m"),
        ("v_mad_i64_i32 %s, vcc, s%d, v%d, %s" % (pair(s, "p"), 20 + (k % 8), R(s, "m"), pair(s, "p")), "m"),
    ]


def body(variant):
    out = []
    for k in STAGES:
        per = [instrs(s, k) for s in range(NS)]
        if variant == "sample_major":
            for s in range(NS):
                out += [t for t, _ in per[s]]
        elif variant == "sample_major_nop":
            for s in range(NS):
                for i, (t, kind) in enumerate(per[s]):
                    out.append(t)
                    if kind == "b":
                        out.append("s_nop 0")     # bitop3 -> xor
                    if i == 3:
                        out.append("s_nop 0")     # ashr -> mad
        elif variant == "instr_major":
            for i in range(7):
                for s in range(NS):
                    out.append(per[s][i][0])
        elif variant == "mads_last":
            for i in range(4):
                for s in range(NS):
                    out.append(per[s][i][0])
            for s in range(NS):
                for i in range(4, 7):
                    out.append(per[s][i][0])
        elif variant == "only32":           # the four 32-bit instructions alone
            for s in range(NS):
                out += [t for t, kind in per[s] if kind != "m"]
        elif variant == "onlymad":          # the three multiply-adds alone
            for s in range(NS):
                out += [t for t, kind in per[s] if kind == "m"]
        elif variant in ONLY:               # one opcode alone (energy probe)
            i = ONLY[variant]
            for s in range(NS):
                t, kind = per[s][i]
                # result into the scratch register so that the chains stay short
                out += [t] * 4
        elif variant == "onlyalign":        # v_alignbit_b32 (the direction-bit collect)
            for s in range(NS):
                out += ["v_alignbit_b32 v%d, v%d, v%d, 31" % (R(s, "m"), R(s, "m"), R(s, "yh"))] * 4
        elif variant in ("vop2_stage", "vop2_stage_im", "xad_stage", "xad_stage_im"):
            # the same micro-rotation in a 32-bit container (x = xh, y = yh, p = pl)
            # without multiply-adds: conditional negate as (v ^ m) - m with m the
            # sign mask of y -- 12 full-rate VOP2, or 10 with v_xad_u32
            # ((a ^ b) + c, a half-rate VOP3); _im = instruction-major over the samples
            def st(s):
                x, y, pp = R(s, "xh"), R(s, "yh"), R(s, "pl")
                m, nm, sy, sx = R(s, "m"), R(s, "mn"), R(s, "sy"), R(s, "sx")
                sa = 20 + (k % 8)
                if variant.startswith("vop2"):
                    return ["v_ashrrev_i32_e32 v%d, 31, v%d" % (m, y),
                            "v_ashrrev_i32_e32 v%d, %d, v%d" % (sy, k, y),
                            "v_ashrrev_i32_e32 v%d, %d, v%d" % (sx, k, x),
                            "v_xor_b32_e32 v%d, v%d, v%d" % (sy, sy, m),
                            "v_sub_u32_e32 v%d, v%d, v%d" % (sy, sy, m),
                            "v_add_u32_e32 v%d, v%d, v%d" % (x, x, sy),
                            "v_xor_b32_e32 v%d, v%d, v%d" % (sx, sx, m),
                            "v_sub_u32_e32 v%d, v%d, v%d" % (sx, sx, m),
                            "v_sub_u32_e32 v%d, v%d, v%d" % (y, y, sx),
                            "v_xor_b32_e32 v%d, s%d, v%d" % (nm, sa, m),
                            "v_sub_u32_e32 v%d, v%d, v%d" % (nm, nm, m),
                            "v_add_u32_e32 v%d, v%d, v%d" % (pp, pp, nm)]
                return ["v_ashrrev_i32_e32 v%d, 31, v%d" % (m, y),
                        "v_ashrrev_i32_e32 v%d, %d, v%d" % (sy, k, y),
                        "v_ashrrev_i32_e32 v%d, %d, v%d" % (sx, k, x),
                        "v_not_b32_e32 v%d, v%d" % (nm, m),
                        "v_xad_u32 v%d, v%d, v%d, v%d" % (x, sy, m, x),
                        "v_sub_u32_e32 v%d, v%d, v%d" % (x, x, m),
                        "v_xad_u32 v%d, v%d, v%d, v%d" % (y, sx, nm, y),
                        "v_sub_u32_e32 v%d, v%d, v%d" % (y, y, nm),
                        "v_xad_u32 v%d, s%d, v%d, v%d" % (pp, sa, m, pp),
                        "v_sub_u32_e32 v%d, v%d, v%d" % (pp, pp, m)]
            blocks = [st(s) for s in range(NS)]
            if variant.endswith("_im"):
                for i in range(len(blocks[0])):
                    for b in blocks:
                        out.append(b[i])
            else:
                for b in blocks:
                    out += b
        elif variant == "nop_all":          # a wait state behind EVERY instruction
            for s in range(NS):
                for t, _ in per[s]:
                    out += [t, "s_nop 0"]
        elif variant == "instr_major_nop":  # ... behind every group of four
            for i in range(7):
                for s in range(NS):
                    out.append(per[s][i][0])
                out.append("s_nop 0")
        elif variant == "nop_mads":         # ... behind every multiply-add only
            for s in range(NS):
                for t, kind in per[s]:
                    out.append(t)
                    if kind == "m":
                        out.append("s_nop 0")
        elif variant == "nop_32":           # ... behind every 32-bit instruction only
            for s in range(NS):
                for t, kind in per[s]:
                    out.append(t)
                    if kind != "m":
                        out.append("s_nop 0")
        elif variant == "alternate":        # 32-bit and 64-bit instructions alternate
            for s in range(0, NS, 2):
                a, b = per[s], per[s + 1]
                order = [a[0], a[2], a[1], a[3], b[0], a[4], b[2], a[5], b[1], a[6],
                         b[3], b[4], b[5], b[6]]
                out += [t for t, _ in order]
        elif variant in RULES:              # sample-major + a yield rule
            y, where = RULES[variant]
            for s in range(NS):
                n32 = 0
                for i, (t, kind) in enumerate(per[s]):
                    out.append(t)
                    if kind != "m":
                        n32 += 1
                    if where(i, kind, n32):
                        out.append(y)
        elif variant == "instr_major_nop32":    # instr_major, a yield behind every 32-bit one
            for i in range(7):
                for s in range(NS):
                    out.append(per[s][i][0])
                    if i < 4:
                        out.append("s_nop 0")
        else:
            raise ValueError(variant)
    return out


def kernel(variant):
    lines = []
    # inputs -> fixed registers
    lines.append("v_mov_b32 v%d, %%1" % C1)
    lines.append("v_mov_b32 v%d, %%2" % C2)
    for s in range(NS):
        for j, name in enumerate(("xl", "xh", "yl", "yh", "pl", "ph")):
            lines.append("v_mad_u32_u24 v%d, %%3, %d, %%4" % (R(s, name), 3 + 7 * s + j))
    for j in range(8):
        lines.append("s_mov_b32 s%d, 0x%x" % (20 + j, 0x13579b + 0x111 * j))
    lines.append("s_movk_i32 s30, %d" % ITERS)
    if SKEW:
        # waves of a real kernel drift apart (loads, stores, waitcnts); here
        # they would march in lockstep.  SKEW: wave w of a SIMD first issues
        # 3 w dummy instructions, so the eight sit at different places of the
        # seven-instruction pattern.
        lines.append("s_getreg_b32 s31, hwreg(HW_REG_HW_ID, 0, 4)")
        lines.append("s_mul_i32 s31, s31, 3")
        lines.append("2:")
        lines.append("s_cmp_eq_u32 s31, 0")
        lines.append("s_cbranch_scc1 3f")
        lines.append("v_mov_b32 v6, v6")
        lines.append("s_sub_u32 s31, s31, 1")
        lines.append("s_branch 2b")
        lines.append("3:")
    lines.append("1:")
    lines += body(variant)
    lines.append("s_sub_u32 s30, s30, 1")
    lines.append("s_cmp_lg_u32 s30, 0")
    lines.append("s_cbranch_scc1 1b")
    # fold everything into %0
    lines.append("v_mov_b32 %0, 0")
    for s in range(NS):
        for name in ("xl", "xh", "yl", "yh", "pl", "ph"):
            lines.append("v_xor_b32 %%0, %%0, v%d" % R(s, name))
    text = "\\n\\t".join(lines)
    clob = ['"vcc"', '"scc"'] + ['"s%d"' % i for i in range(20, 32)] \
        + ['"v%d"' % C1, '"v%d"' % C2, '"v6"'] + ['"v%d"' % i for i in range(10, 10 + 12 * NS)]
    return """__global__ __launch_bounds__(256) void k_%s(uint32_t *out, uint32_t c1, uint32_t c2)
{
	uint32_t r, t = threadIdx.x + blockIdx.x * 256u, u = c1 * 2654435761u;
	asm volatile("%s"
		: "=&v"(r) : "v"(c1), "v"(c2), "v"(t), "v"(u) : %s);
	out[t] = r;
}
""" % (variant, text, ", ".join(clob))


# one opcode alone, four times per sample and stage (index into instrs())
ONLY = {"onlybitop3": 0, "onlyxor": 1, "onlyashr": 2, "onlymadv": 4, "onlymads": 6}
# sample-major order + (yield instruction, where(i, kind, count of 32-bit so far))
RULES = {
    "nop_32_s1": ("s_nop 1", lambda i, k, n: k != "m"),
    "nop_32_s3": ("s_nop 3", lambda i, k, n: k != "m"),
    "nop_bx": ("s_nop 0", lambda i, k, n: k in "bx"),
    "nop_32_alt": ("s_nop 0", lambda i, k, n: k != "m" and n % 2 == 0),
    "nop_32_last": ("s_nop 0", lambda i, k, n: i == 3),
    "nop_32_and_end": ("s_nop 0", lambda i, k, n: k != "m" or i == 6),
    "salu_32": ("s_mov_b32 s31, s30", lambda i, k, n: k != "m"),
    "setprio_32": ("s_setprio 0", lambda i, k, n: k != "m"),
    "sleep_32": ("s_sleep 0", lambda i, k, n: k != "m"),
}
VARIANTS = ["sample_major", "sample_major_nop", "instr_major", "mads_last", "only32",
            "onlymad", "nop_all", "instr_major_nop", "nop_mads", "nop_32", "alternate",
            "instr_major_nop32"] + list(RULES) + list(ONLY) + ["onlyalign", "vop2_stage", "vop2_stage_im", "xad_stage", "xad_stage_im"]
PER_STAGE = {"only32": 4, "onlymad": 3, "onlyalign": 4}
PER_STAGE.update({k: 4 for k in ONLY})
PER_STAGE.update({"vop2_stage": 12, "vop2_stage_im": 12, "xad_stage": 10, "xad_stage_im": 10})

MAIN = r"""
int main(int argc, char **argv)
{
	hipDeviceProp_t p;
	CHECK(hipGetDeviceProperties(&p, 0));
	const int cus = p.multiProcessorCount;
	printf("device %s, %d CUs; %d stages x %d samples x 7 instructions, %d iterations\n",
		p.gcnArchName, cus, NSTAGES, NSAMP, NITERS);
	uint32_t *out;
	const size_t cap = (size_t)cus * 8 * 256;
	CHECK(hipMalloc(&out, cap * 4));
	std::vector<uint32_t> h(cap);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	struct V { const char *name; void (*k)(uint32_t *, uint32_t, uint32_t); int per_stage; };
	const V vs[] = { VLIST };
	if (argc >= 4 && !strcmp(argv[1], "--sustain")) {
		// ./sched_probe --sustain <variant> <seconds>: keep ONE variant running from
		// 8 waves a SIMD (the caller samples the socket power meanwhile) and print
		// the rate it held: wave-instructions per second per SIMD
		const double secs = atof(argv[3]);
		for (const V &v : vs) {
			if (strcmp(v.name, argv[2]))
				continue;
			const int blocks = cus * 8;
			const double per_launch = (double)NSTAGES * NSAMP * v.per_stage * NITERS * 8;
			auto t0 = std::chrono::steady_clock::now();
			long launches = 0;
			double el = 0;
			do {
				for (int r = 0; r < 50; r++)
					hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, out, 0x40000000u, 0x80000000u);
				CHECK(hipDeviceSynchronize());
				launches += 50;
				el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			} while (el < secs);
			printf("%s: %.2f s, %.4g wave-instructions/s/SIMD, %.4g lane-instructions/s (chip)\n",
				v.name, el, launches * per_launch / el,
				launches * per_launch / el * 64 * 4 * cus);
			return 0;
		}
		printf("no variant %s\n", argv[2]);
		return 1;
	}
		printf("%-18s %6s %10s %12s %16s\n", "variant", "waves", "ms", "checksum",
		"cyc/instr @2.4GHz");
	for (int wps = 8; wps >= 1; wps /= 2) {
		const int blocks = cus * wps;		// 4 waves per block: wps waves per SIMD
		for (const V &v : vs) {
			for (int rep = 0; rep < 3; rep++)
				hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, out, 0x40000000u, 0x80000000u);
			CHECK(hipEventRecord(e0));
			const int reps = 10;
			for (int rep = 0; rep < reps; rep++)
				hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, out, 0x40000000u, 0x80000000u);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			ms /= reps;
			CHECK(hipMemcpy(h.data(), out, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost));
			uint64_t sum = 0;
			for (size_t i = 0; i < (size_t)blocks * 256; i++)
				sum = sum * 1099511628211ull + h[i];
			// per SIMD: wps waves, each instr_per_wave instructions
			const double instr_per_wave = (double)NSTAGES * NSAMP * v.per_stage * NITERS;
			const double cyc = ms * 1e-3 * 2.4e9 / (wps * instr_per_wave);
			printf("%-18s %6d %10.3f %012llx %16.2f\n", v.name, wps, ms,
				(unsigned long long)(sum & 0xffffffffffffull), cyc);
		}
	}
	return 0;
}
"""


def main():
    src = ["// sched_probe.hip -- GENERATED by tools/gen_sched_probe.py (see there)",
           "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstdint>",
           "#include <vector>", "#include <chrono>", "#include <cstring>", "#include <cstdlib>",
           "#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \\",
           '\tprintf("HIP error %s at %d\\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)',
           "#define NSTAGES %d" % len(STAGES), "#define NSAMP %d" % NS,
           "#define NITERS %d" % ITERS, ""]
    for v in VARIANTS:
        src.append(kernel(v))
    vlist = ", ".join('{"%s", k_%s, %d}' % (v, v, PER_STAGE.get(v, 7)) for v in VARIANTS)
    src.append(MAIN.replace("VLIST", vlist))
    with open(os.path.join(ROOT, "tools", "sched_probe_skew.hip" if SKEW else "sched_probe.hip"), "w") as f:
        f.write("\n".join(src))


if __name__ == "__main__":
    main()
