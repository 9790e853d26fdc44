#ifndef CORDIC_DEVICE_H
#define CORDIC_DEVICE_H
// cordic_device.h -- gfx950 (CDNA4) device code of the CORDIC rotation engine:
// kernel-argument block, micro-rotation stages, pre/post steps and the
// unrolled kernels.  Included by the instantiation units (cordic_inst_*.hip)
// and by cordic_kernels.hip (generic kernels + launch logic).
//
// One lane owns one sample at a time (kVec consecutive samples per tile pass,
// for 16-byte loads/stores and 4-way ILP over the serially dependent stage
// chain).  No LDS tiling and no MFMA: the path is a pure streaming map --
// 4..16 algorithmic bytes per sample against ~10 (WW<=32) or ~16 (WW<=64)
// integer VALU operations per rotation -- so what matters is (a) the
// instruction count of one micro-rotation, (b) 1 KiB-per-wave coalesced
// global accesses, (c) enough waves in flight to cover HBM latency.
//
// Arithmetic contract (what "bit-exact" refers to): reference rtl/cordic.v
// :85-86,131-188,231-283,288-314 and rtl/topolar.v:83-84,122-152,195-246,
// 251-271, i.e. the Verilog emitted by sw/basiccordic.cpp / sw/topolar.cpp;
// sequential flavours rtl/seqcordic.v:270-324, rtl/seqpolar.v:208,254-307.
//
// Representation:
//  * phase: left-justified in a 32-bit register (P = phase << (32-PW)), so
//    the PW-bit wrap is the natural 32-bit wrap, the sign test is bit 31, the
//    octant is the top 3 bits.  The arctan table is pre-shifted the same way
//    on the host and arrives in SGPRs through the kernel-argument block.
//  * x / y: sign extended in a 32-bit (WW<=32) or 64-bit (WW<=64) container.
//    When WW equals the container width the wrap is natural; when it is
//    narrower the host has proven (cordic_config.cpp: overflow_reachable)
//    that no value can leave the WW-bit range, otherwise the job goes to the
//    generic kernel, which wraps explicitly after every operation.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "cordic_internal.h"

namespace cordic_amd {
namespace dev {

constexpr int kBlock = 256;		// 4 waves: one per SIMD
constexpr int kVec = 4;			// samples per lane per pass (16 B)
constexpr int kTile = kBlock * kVec;	// samples per block per pass
constexpr int kSeedBlock = CORDIC_SEED_BLOCK;	// waves of a block share one table
#ifndef CORDIC_SEED_SUBTILES
#define CORDIC_SEED_SUBTILES 2
#endif
constexpr int kSeedSub = CORDIC_SEED_SUBTILES;	// rows of a tile (rotator_seeded)
constexpr int kSeedStages = CORDIC_SEED_STAGES;	// M: stages replaced by the table

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
typedef int16_t i16x4 __attribute__((ext_vector_type(4)));

// Kernel-argument block: wave-uniform, so hipcc keeps it in SGPRs (s_load).
struct CoreParams {
	uint32_t angle[CORDIC_AMD_MAX_STAGES];	// left-justified arctan table
	int32_t	nlive;		// rotations to perform (generic kernel)
	int32_t	iw;		// port width of i_xval / i_yval
	int32_t	in_shl;		// zeros appended below the input
	int32_t	pw_shl;		// 32 - PW
	int32_t	ww, ow;
	int32_t	r;		// WW - OW: bits dropped at the output
	uint32_t round_bit;	// 1 if WW > OW+1 (convergent rounding) else 0
	int64_t	round_base;	// 2^(r-1) - 1 if rounding else 0
	int32_t	wrap;		// generic kernel: wrap to WW bits explicitly
	int32_t	r_lj;		// left-justified form: r + LJ
	int64_t	round_base_lj;	// round_base << LJ
	int32_t	x0, y0;		// constant-vector feeds (sign extended)
	uint32_t phase0, fcw;	// NCO, left-justified
	uint64_t index0;	// NCO: global index of sample 0
	uint32_t post_mul;	// 0, or the gain-annihilation multiplier
				// (CORDIC_FLAG_UNIT_GAIN): o = (o * post_mul) >> 32
	uint32_t xy_nco;	// per-sample vector feeds: 1 = the phase is not read
				// from an array but generated, phase0 + (index0 + i)
				// * fcw (the fused NCO MIXER, cordic_mix: round 5)
};

// ---------------------------------------------------------------- utilities

__device__ __forceinline__ int32_t sext32(int32_t v, int w)
{
	const int s = 32 - w;		// w in 1..32
	return (int32_t)((uint32_t)v << s) >> s;
}
__device__ __forceinline__ int64_t sext64(int64_t v, int w)
{
	const int s = 64 - w;		// w in 1..64
	return (int64_t)((uint64_t)v << s) >> s;
}

// Measured on MI355X (tools/valu_microbench.hip, profiles/valu_microbench_r01.txt):
// two-operand 32-bit integer VALU ops (add/sub/xor/or/not/shift) issue at the
// full rate, every VOP3 form (three-operand integer ops, carry in/out,
// v_alignbit) and every 64-bit op (v_ashrrev_i64, v_lshl_add_u64,
// v_mad_i64_i32) at ~0.6x of it -- and a 64-bit op costs no more than a
// three-operand 32-bit one.  So the micro-rotation is built around
// v_mad_i64_i32 (D.i64 = S0.i32 * S1.i32 + S2.i64): with s = +/-1 per lane it
// is the conditional negate, the carry chain and the 64-bit add of
//      x' = x -/+ (y >>> k)
// in ONE instruction, where xor/carry formulations need 4-6.  The same
// instruction advances the phase (low word of the result; the high word of
// the phase register pair is don't-care).
//
// v_mad_i64_i32 also writes a carry-out SGPR pair (VCC here, declared as a
// clobber); nothing reads it, so no wait states are owed.  (Letting the
// compiler spread the carry-outs over other SGPR pairs measured no difference:
// the shared VCC is not a bottleneck.)
// -s for s = +/-1 as one full-rate VOP2 (hipcc would fuse (d|1)^-2 into a
// three-operand v_bitop3_b32)
//
// CORDIC_STAGE_YIELD (A/B knob, off).  A wave issues at most one VALU
// instruction per four cycles and a SIMD keeps issuing from the wave it is on:
// a full-rate 32-bit instruction (two cycles of the SIMD) leaves the other two
// empty unless that wave steps aside and another one fills them.
// round 5's wave-issue probe (the r2p micro-rotation's seven instructions, 4 samples a
// lane, 8 waves a SIMD, 2 ms bursts): 4.19 cycles per instruction as written,
// 3.76 with the wait states hipcc happens to pad behind asm-defined registers,
// 3.36 with `s_nop 0` behind every 32-bit instruction and none behind the
// multiply-adds (3.02 = the opcodes' own 2 / 4 cycles); s_nop 1, a wait state
// behind a multiply-add, or an SALU instruction as the filler are all worse
// (profiles/r05/sched_probe.txt).  Built into topolar_lj (=1) the kernel needs
// 1 % fewer cycles and is no faster: it runs for seconds, not bursts, the
// SMU's PPT limiter holds the clock (active 71 % of the time at 1.29 kW), and
// the better schedule is answered with a lower clock: 224 against 229
// Gsample/s at 2.09 / 2.16 GHz, 5.73 / 5.80 nJ per sample
// (profiles/r05/ab_stage_yield.txt).  What is left to gain there is energy
// per sample, not issue slots.
#ifndef CORDIC_STAGE_YIELD
#define CORDIC_STAGE_YIELD 0
#endif
#define CORDIC_YIELD "\n\ts_nop 0"
__device__ __forceinline__ int32_t op_flip(int32_t s)
{
	int32_t r;
	asm("v_xor_b32 %0, -2, %1" : "=v"(r) : "v"(s));
	return r;
}
__device__ __forceinline__ void op_mad(int64_t &acc, int32_t a, int32_t b)
{
	asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
// same with the multiplicand in an SGPR (the arctan table entry)
// acc = a * b (v_mad_i64_i32 with the inline constant 0 as addend: no zeroed
// register pair)
__device__ __forceinline__ int64_t op_mul(int32_t a, int32_t b)
{
	int64_t d;
	asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
	return d;
}

__device__ __forceinline__ void op_mad_s(int64_t &acc, uint32_t a_sgpr, int32_t b)
{
	asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "s"(a_sgpr), "v"(b) : "vcc");
}

// x >>> K of a value held in a 32-bit container (low word of the pair)
template <int K> __device__ __forceinline__ int32_t asr_lo(int64_t v)
{
	return (int32_t)(uint32_t)v >> ((K > 31) ? 31 : K);
}
// low 32 bits of a 64-bit value >>> K; equals the full result whenever it
// fits in 32 bits (K >= WW-32)
template <int K> __device__ __forceinline__ int32_t asr_narrow(int64_t v)
{
	const uint32_t lo = (uint32_t)v;
	const int32_t hi = (int32_t)((uint64_t)v >> 32);
	if constexpr (K <= 31)
		return (int32_t)__builtin_amdgcn_alignbit((uint32_t)hi, lo, K);
	else
		return hi >> ((K - 32 > 31) ? 31 : K - 32);
}

// Container tags.  Both keep x, y and the phase in 64-bit register pairs (the
// mad needs a 64-bit addend); Narrow32 only maintains the low words.
struct Narrow32 { static constexpr bool wide = false; static constexpr int lj = 0; };
struct Wide64   { static constexpr bool wide = true;  static constexpr int lj = 0; };
// Wide64 with x/y carried LEFT-justified by LJ bits after the first 31-LJ
// stages (see rot_stage_lj); LJ = 64-WW makes the 64-bit wrap the WW-bit wrap.
template <int LJ> struct WideLJ {
	static constexpr bool wide = true;
	static constexpr int lj = LJ;
	static constexpr int ngen = 31 - LJ;	// stages with k < 32-LJ
};
// Scale of the residual phase in those kernels: p~ = sext(P) << ps.  The stage
// multipliers are read off the bits of hi(p~) above bit LJ, which therefore
// have to be sign bits: |P| <= 2^29 gives |hi(p~)| <= 2^(ps-3), so ps <= LJ+2
// (31 for LJ = 29, 30; the angles are applied as a << (ps - LJ)).
template <int LJ> struct LjPhase {
	static constexpr int ps = (LJ + 2 > 31) ? 31 : LJ + 2;
};

// ------------------------------------------------------- rotator: p2r stage

// rtl/cordic.v:262-280 with s = +1 where the residual phase is >= 0, -1 where
// it is negative:   x' = x - s*(y>>>k),  y' = y + s*(x>>>k),  p' = p - s*a.
// GENERAL: (y>>>k) may need more than 32 bits (k < WW-32): explicit 64-bit
// shift, conditional negate and add.
template <typename C, int K, bool GENERAL>
__device__ __forceinline__ void rot_stage(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a)
{
	const int32_t d = (int32_t)(uint32_t)p >> 31;	// -1: negative phase
	const int32_t s = d | 1;
	const int32_t ns = op_flip(s);			// -s
	if constexpr (!C::wide) {
		const int32_t sy = asr_lo<K>(y), sx = asr_lo<K>(x);
		op_mad(x, sy, ns);
		op_mad(y, sx, s);
	} else if constexpr (!GENERAL) {
		const int32_t sy = asr_narrow<K>(y), sx = asr_narrow<K>(x);
		op_mad(x, sy, ns);
		op_mad(y, sx, s);
	} else {
		constexpr int k = (K > 63) ? 63 : K;
		const int64_t d64 = (int64_t)d, nd64 = ~d64;
		const int64_t sy = y >> k, sx = x >> k;
		x = x + ((sy + nd64) ^ nd64);	// phase >= 0: x - sy
		y = y + ((sx + d64) ^ d64);	// phase <  0: y - sx
	}
	op_mad_s(p, a, ns);
}

// ---------------------------------------------------- converter: r2p stage

// rtl/topolar.v:226-243 with t = +1 where y >= 0, -1 where y < 0:
//   x' = x + t*(y>>>k),  y' = y - t*(x>>>k),  p' = p + t*a.
template <typename C, int K, bool GENERAL>
__device__ __forceinline__ void pol_stage(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a)
{
	const int32_t d = C::wide ? (int32_t)((uint64_t)y >> 32) >> 31
				  : (int32_t)(uint32_t)y >> 31;
#ifdef CORDIC_POL_VOP2
	// Round-5 experiment (profiles/r05/ab_pol_vop2.txt):
	// the 32-bit container WITHOUT the half-rate multiply-adds -- sign-mask
	// arithmetic on full-rate VOP2 only: 12 instructions per stage (1 mask,
	// 2 shifts, 3 x {xor, sub, add}) against 8 with v_mad_i64_i32 (7 in
	// topolar_lj).  -DCORDIC_POL_VOP2 builds it for the A/B; not the product.
	if constexpr (!C::wide) {
		const int32_t lx = (int32_t)(uint32_t)x, ly = (int32_t)(uint32_t)y;
		const int32_t lp = (int32_t)(uint32_t)p;
		const int32_t sy = ly >> ((K > 31) ? 31 : K), sx = lx >> ((K > 31) ? 31 : K);
		const int32_t nd = ~d;
		// y < 0 (d = -1): x - sy, y + sx, p - a;  y >= 0: x + sy, y - sx, p + a
		const int32_t nx = lx + ((sy ^ d) - d);
		const int32_t ny = ly + ((sx ^ nd) - nd);
		const int32_t np = lp + (((int32_t)a ^ d) - d);
		x = (int64_t)(uint32_t)nx;
		y = (int64_t)(uint32_t)ny;
		p = (int64_t)(uint32_t)np;
		return;
	}
#endif
	const int32_t t = d | 1;
	const int32_t nt = op_flip(t);
	if constexpr (!C::wide) {
		const int32_t sy = asr_lo<K>(y), sx = asr_lo<K>(x);
		op_mad(x, sy, t);
		op_mad(y, sx, nt);
	} else if constexpr (!GENERAL) {
		const int32_t sy = asr_narrow<K>(y), sx = asr_narrow<K>(x);
		op_mad(x, sy, t);
		op_mad(y, sx, nt);
	} else {
		constexpr int k = (K > 63) ? 63 : K;
		const int64_t d64 = (int64_t)d, nd64 = ~d64;
		const int64_t sy = y >> k, sx = x >> k;
		x = x + ((sy + d64) ^ d64);	// y < 0: x - sy
		y = y + ((sx + nd64) ^ nd64);	// y >= 0: y - sx
	}
	op_mad_s(p, a, t);
}

// DYN = true: the instance is unrolled to NLIVE stages as a MAXIMUM and leaves
// through a wave-uniform (scalar) branch after kp.nlive of them, so one
// instance serves every smaller stage count (measured 3-5 % slower than a
// static instance; far faster than the generic kernel).
// Compile-time unrolled stage chain over the kVec samples of a lane: stage
// i of all samples before stage i+1, so the four dependency chains interleave.
// The first NGEN stages of a Wide64 core use the GENERAL form.
template <typename C, int NLIVE, int NGEN, int I = 0, bool DYN = false> struct RotChain {
	static __device__ __forceinline__ void run(int64_t (&x)[kVec],
			int64_t (&y)[kVec], int64_t (&p)[kVec], const CoreParams &kp)
	{
		if constexpr (I < NLIVE) {
			if (!DYN || I < kp.nlive) {
#pragma unroll
				for (int v = 0; v < kVec; v++)
					rot_stage<C, I + 1, (I < NGEN)>(x[v], y[v], p[v],
							kp.angle[I]);
				RotChain<C, NLIVE, NGEN, I + 1, DYN>::run(x, y, p, kp);
			}
		}
	}
};
template <typename C, int NLIVE, int NGEN, int I = 0, bool DYN = false> struct PolChain {
	static __device__ __forceinline__ void run(int64_t (&x)[kVec],
			int64_t (&y)[kVec], int64_t (&p)[kVec], const CoreParams &kp)
	{
		if constexpr (I < NLIVE) {
			if (!DYN || I < kp.nlive) {
#pragma unroll
				for (int v = 0; v < kVec; v++)
					pol_stage<C, I + 1, (I < NGEN)>(x[v], y[v], p[v],
							kp.angle[I]);
				PolChain<C, NLIVE, NGEN, I + 1, DYN>::run(x, y, p, kp);
			}
		}
	}
};

// ------------------------------------- rotator: left-justified wide stage
//
// For WW in 33..35 most of the cost of a Wide64 stage is not the rotation but
// getting (y>>>k) into 32 bits (v_alignbit, VOP3) and the +/-1 multipliers.
// Carry x and y shifted LEFT by LJ (x~ = x << LJ) and the phase as
// p~ = sext(P) << 31.  Then
//   * (y >>> k) for k >= 32-LJ is an arithmetic shift of the HIGH word of y~
//     by k-(32-LJ): one full-rate v_ashrrev_i32, no v_alignbit;
//   * the multipliers are s~ = +/-2^LJ, read straight off the top bits of the
//     high word of p~ (= P>>1, whose two top bits both hold the sign):
//         s~ = (hi(p~) & -2^(LJ+1)) | 2^LJ ,   -s~ = s~ ^ -2^(LJ+1);
//   * x~' = x~ + (-s~)*(y>>>k),  y~' = y~ + s~*(x>>>k),
//     p~' = p~ + (-s~)*(a << (31-LJ))      -- three v_mad_i64_i32.
// The residual phase never leaves [-2^29, 2^29] (after the octant fold
// |P| <= 2^29 and every stage moves it towards zero by a <= 2^28.3), so
// sext(P) << 31 cannot wrap and bit 63 == bit 62 always.
template <int LJ> struct LjConst {
	static constexpr uint32_t bit = 1u << LJ;
	static constexpr uint32_t mask = ~((bit << 1) - 1u);
	static constexpr int first = 32 - LJ;	// first k served by this form
};

// Left-justify the PW-bit phase words of a lane.  v_lshlrev_b32 is a half-rate
// instruction and PW = 32 (shift by 0) is the common case, so the shift sits
// behind a wave-uniform branch; it is written in asm because the compiler
// knows that a shift by zero is the identity and would fold the branch away.
__device__ __forceinline__ void left_justify(const u32x4 in, uint32_t (&P)[4],
		int sh)
{
#pragma unroll
	for (int v = 0; v < 4; v++)
		P[v] = in[v];
	if (sh != 0) {
#pragma unroll
		for (int v = 0; v < 4; v++)
			asm("v_lshlrev_b32 %0, %1, %0" : "+v"(P[v]) : "s"(sh));
	}
}

// (a & c) | b  and  (a & c) ^ b  as one v_bitop3_b32 each (truth tables with
// src0 = 0xF0, src1 = 0xCC, src2 = 0xAA) instead of and + or + xor.  Constants
// live in VGPRs (VOP3 takes no literal and at most one SGPR on gfx950).
// Inline asm although __builtin_amdgcn_bitop3_b32 exists: hipcc pads every
// instruction that depends on an asm-defined VGPR with an s_nop (74 per pass in
// topolar_lj), yet the asm form keeps the stage-major order written here and
// measured 3-7 % FASTER than the builtin, which the scheduler rearranges
// (same-box A/B, profiles/r02/ab_bitop3_builtin.txt; a scheduling fence after
// every stage on top of the asm form: 0.5-1 % slower again).
__device__ __forceinline__ uint32_t op_and_or(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t d;
	asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xec" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}
__device__ __forceinline__ uint32_t op_and_xor(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t d;
	asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}
// a loop-invariant constant pinned in a VGPR
__device__ __forceinline__ uint32_t vgpr_const(uint32_t v)
{
	uint32_t d;
	asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "s"(v));
	return d;
}

struct LjRegs { uint32_t mask, bit, maskbit; };	// VGPR-resident constants

template <int LJ, int K>
__device__ __forceinline__ void rot_stage_lj(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a_scaled, const LjRegs &c)
{
	const uint32_t ph = (uint32_t)((uint64_t)p >> 32);
	const int32_t s = (int32_t)op_and_or(ph, c.bit, c.mask);	// +/- 2^LJ
	const int32_t ns = (int32_t)op_and_xor(ph, c.maskbit, c.mask);	// -s
	constexpr int sh = (K - LjConst<LJ>::first > 31) ? 31
						: K - LjConst<LJ>::first;
	const int32_t sy = (int32_t)((uint64_t)y >> 32) >> sh;
	const int32_t sx = (int32_t)((uint64_t)x >> 32) >> sh;
	op_mad(x, sy, ns);
	op_mad(y, sx, s);
	op_mad_s(p, a_scaled, ns);
}

// The same stage with its multipliers handed in (direction tails of the seeded
// kernel: -s 2^LJ and s 2^LJ come out of a table indexed by the residual
// phase, there is no phase to update): two shifts and two multiply-adds.
template <int LJ, int K>
__device__ __forceinline__ void rot_stage_lj_dir(int64_t &x, int64_t &y, int32_t ns,
		int32_t s)
{
	static_assert(K >= LjConst<LJ>::first, "early stages carry low-word bits");
	constexpr int sh = (K - LjConst<LJ>::first > 31) ? 31
						: K - LjConst<LJ>::first;
	const int32_t sy = (int32_t)((uint64_t)y >> 32) >> sh;
	const int32_t sx = (int32_t)((uint64_t)x >> 32) >> sh;
	op_mad(x, sy, ns);
	op_mad(y, sx, s);
}

// ... and for the 32-bit container (WW <= 32: x and y in the low words,
// multipliers -/+ 1)
template <int K>
__device__ __forceinline__ void rot_stage_dir32(int64_t &x, int64_t &y, int32_t ns,
		int32_t s)
{
	const int32_t sy = asr_lo<K>(y), sx = asr_lo<K>(x);
	op_mad(x, sy, ns);
	op_mad(y, sx, s);
}

// The first 31-LJ stages (k < 32-LJ) on the same left-justified pairs:
// (y >>> k) no longer fits the high word alone,
//     y >>> k = hi(y~) * 2^D + r,   D = 32-LJ-k,   r = the top D bits of lo(y~),
// so (y >>> k) << LJ = hi * 2^(32-k) + r * 2^LJ: two more multiply-adds per
// coordinate (three for k = 1, whose 2^31 is applied as 2 * 2^30) and one
// shift for r, against ~18 instructions for the explicit 64-bit shift /
// conditional negate / add of the GENERAL form.
template <int LJ, int K>
__device__ __forceinline__ void rot_stage_lj_early(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a_scaled, const LjRegs &c)
{
	static_assert(K >= 1 && K < LjConst<LJ>::first, "late stages use rot_stage_lj");
	constexpr int D = LjConst<LJ>::first - K;
	const uint32_t ph = (uint32_t)((uint64_t)p >> 32);
	const int32_t s = (int32_t)op_and_or(ph, c.bit, c.mask);	// +/- 2^LJ
	const int32_t ns = (int32_t)op_and_xor(ph, c.maskbit, c.mask);	// -s
	// +/- 2^(32-k), halved for k = 1 and applied twice
	constexpr int up = (K == 1) ? 30 - LJ : 32 - K - LJ;
	const int32_t sh = (int32_t)((uint32_t)s << up);
	const int32_t nsh = (int32_t)((uint32_t)ns << up);
	const int32_t yh = (int32_t)((uint64_t)y >> 32), xh = (int32_t)((uint64_t)x >> 32);
	const int32_t yr = (int32_t)((uint32_t)y >> (32 - D));
	const int32_t xr = (int32_t)((uint32_t)x >> (32 - D));
	op_mad(x, yh, nsh);
	op_mad(y, xh, sh);
	if constexpr (K == 1) {
		op_mad(x, yh, nsh);
		op_mad(y, xh, sh);
	}
	op_mad(x, yr, ns);
	op_mad(y, xr, s);
	op_mad_s(p, a_scaled, ns);
}

template <int LJ, int NLIVE, int I, bool DYN = false> struct RotChainLJ {
	static __device__ __forceinline__ void run(int64_t (&x)[kVec],
			int64_t (&y)[kVec], int64_t (&p)[kVec], const CoreParams &kp,
			const LjRegs &c)
	{
		if constexpr (I < NLIVE) {
			if (!DYN || I < kp.nlive) {
				const uint32_t a = kp.angle[I] << (LjPhase<LJ>::ps - LJ);
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					if constexpr (I + 1 < LjConst<LJ>::first)
						rot_stage_lj_early<LJ, I + 1>(x[v], y[v], p[v], a, c);
					else
						rot_stage_lj<LJ, I + 1>(x[v], y[v], p[v], a, c);
				}
				RotChainLJ<LJ, NLIVE, I + 1, DYN>::run(x, y, p, kp, c);
			}
		}
	}
};

// ------------------------------------------------------------- pre / post

// rtl/cordic.v:131-188 on a left-justified phase: q = quadrant of
// (phase + 45 deg); rotate the vector by q * 90 deg, remove q * 2^(PW-2).
// Written with sign masks, not selects: v_cndmask_b32 measured ~23 cycles per
// wave-instruction on MI355X (profiles/valu_microbench_r01.txt), ten times a
// plain VOP2.
//   q = 1, 3: swap e_x / e_y;   q = 1, 2: negate x;   q = 2, 3: negate y
template <typename T>
__device__ __forceinline__ void fold_octant(T ex, T ey, uint32_t P, T &x, T &y,
		uint32_t &p)
{
	using U = typename std::make_unsigned<T>::type;
	const uint32_t q = (P + 0x20000000u) >> 30;
	p = P - (q << 30);
	const U ms = (U)(T)(-(int32_t)(q & 1u));		// all ones: swap
	const U mx = (U)(T)(-(int32_t)(((q + 1u) >> 1) & 1u));	// negate x
	const U my = (U)(T)(-(int32_t)(q >> 1));		// negate y
	const U d = ((U)ex ^ (U)ey) & ms;
	const U a = (U)ex ^ d, b = (U)ey ^ d;
	x = (T)((a ^ mx) - mx);
	y = (T)((b ^ my) - my);
}

// rtl/topolar.v:122-152.  With ax = |e_x|, ay = |e_y| (two's complement
// negation, i.e. exactly the -e_xval / -e_yval terms of the case arms):
//   x0 = ax + ay in every quadrant; y0 = ay - ax when the signs agree,
//   ax - ay otherwise; p0 = {1,7,3,5} * 2^(PW-3) for {++,+-,-+,--}.
template <typename T>
__device__ __forceinline__ void fold_quadrant(T ex, T ey, bool xneg, bool yneg,
		T &x, T &y, uint32_t &p)
{
	using U = typename std::make_unsigned<T>::type;
	const U ax = xneg ? (U)0 - (U)ex : (U)ex;
	const U ay = yneg ? (U)0 - (U)ey : (U)ey;
	x = (T)(ax + ay);
	y = (xneg != yneg) ? (T)(ax - ay) : (T)(ay - ax);
	const uint32_t oct = xneg ? (yneg ? 5u : 3u) : (yneg ? 7u : 1u);
	p = oct << 29;
}

// rtl/cordic.v:288-295,311-312 (and the truncating form of
// sw/basiccordic.cpp:433-438 when WW == OW+1, selected by round_bit == 0).
template <typename T>
__device__ __forceinline__ int32_t round_to_ow(T v, const CoreParams &kp)
{
	using U = typename std::make_unsigned<T>::type;
	const U b = ((U)v >> kp.r) & (U)kp.round_bit;
	const T w = (T)((U)v + (U)(T)kp.round_base + b);
	return (int32_t)(w >> kp.r);
}

// the same on a value carried left-justified by LJ bits
// r + LJ == 32 (e.g. WW 35 -> OW 32): the rounded result is the high word;
// the rounding increment (base + b) * 2^LJ is one more mad.
template <int LJ>
__device__ __forceinline__ int32_t round_to_ow_lj32(int64_t v, const CoreParams &kp)
{
	const uint32_t b = (uint32_t)((uint64_t)v >> 32) & kp.round_bit;
	const int32_t t = (int32_t)(b + (uint32_t)kp.round_base);
	op_mad_s(v, 1u << LJ, t);
	return (int32_t)((uint64_t)v >> 32);
}

// r + LJ > 32 (e.g. WW 33 / 34 -> OW 30 / 31, and every WW <= 32 core carried
// at LJ = 30): the same on the high word, then the remaining r + LJ - 32 bits
// are dropped with one 32-bit shift.
template <int LJ>
__device__ __forceinline__ int32_t round_to_ow_lj_hi(int64_t v, const CoreParams &kp,
		uint32_t sh)
{
	const uint32_t b = ((uint32_t)((uint64_t)v >> 32) >> sh) & kp.round_bit;
	const int32_t t = (int32_t)(b + (uint32_t)kp.round_base);
	op_mad_s(v, 1u << LJ, t);
	return (int32_t)((uint64_t)v >> 32) >> sh;
}

template <int LJ>
__device__ __forceinline__ int32_t round_to_ow_lj(int64_t v, const CoreParams &kp)
{
	const uint64_t b = ((uint64_t)v >> kp.r_lj) & (uint64_t)kp.round_bit;
	const int64_t w = (int64_t)((uint64_t)v + (uint64_t)kp.round_base_lj
				+ (b << LJ));
	return (int32_t)(w >> kp.r_lj);
}

// Optional fused output scaling: "You can annihilate this gain by multiplying
// by 32'h%08x and right shifting by 32 bits" (sw/cordiclib.cpp:205-209).  The
// OW-bit output is the signed factor, the constant an unsigned 32-bit one;
// the shift is arithmetic.  Compiled into separate (dynamic-exit) instances,
// UG = true: even a wave-uniform branch on it measured ~1 % on the hot kernels.
__device__ __forceinline__ int32_t unit_gain(int32_t o, uint32_t k)
{
	return (int32_t)(((int64_t)o * (int64_t)(uint64_t)k) >> 32);
}
template <bool UG, typename V>
__device__ __forceinline__ void apply_unit_gain(V &v, const CoreParams &kp)
{
	if constexpr (UG) {
#pragma unroll
		for (int i = 0; i < kVec; i++)
			v[i] = unit_gain(v[i], kp.post_mul);
	}
}

// ------------------------------------------------------- memory accessors


// Memory containers of the sample arrays.  Io32: int32 / uint32 per value
// (any port width).  Io16: int16 / uint16 per value for cores whose ports are
// at most 16 bits wide (the 16-bit benches of the reference keep their
// samples in shorts) -- half the HBM bytes per sample; a lane then moves
// 8 bytes per array per pass.  In registers both are 32-bit.
//
// The vector types that global pointers point to only promise ELEMENT
// alignment: hipcc still emits global_load/store_dwordx4 (dwordx2) for them,
// and gfx950 executes those at any element-aligned address (measured,
// tools/unaligned_probe.hip: a 1R2W stream displaced by 4/8/12 bytes runs at
// 3.9 TB/s against 4.7 TB/s aligned).  So a caller's odd offset costs ~20 %
// instead of dropping the job onto the generic kernel.
typedef u32x4 u32x4g __attribute__((aligned(4)));
typedef i32x4 i32x4g __attribute__((aligned(4)));
typedef u16x4 u16x4g __attribute__((aligned(2)));
typedef i16x4 i16x4g __attribute__((aligned(2)));

struct Io32 {
	typedef u32x4g uvec;
	typedef i32x4g ivec;
	typedef uint32_t uelem;
	typedef int32_t ielem;
	static __device__ __forceinline__ u32x4 widen(u32x4 v) { return v; }
	static __device__ __forceinline__ i32x4 widen(i32x4 v) { return v; }
	static __device__ __forceinline__ u32x4 narrow(u32x4 v) { return v; }
	static __device__ __forceinline__ i32x4 narrow(i32x4 v) { return v; }
};
struct Io16 {
	typedef u16x4g uvec;
	typedef i16x4g ivec;
	typedef uint16_t uelem;
	typedef int16_t ielem;
	static __device__ __forceinline__ u32x4 widen(u16x4 v)
	{
		return __builtin_convertvector(v, u32x4);
	}
	static __device__ __forceinline__ i32x4 widen(i16x4 v)
	{
		return __builtin_convertvector(v, i32x4);
	}
	static __device__ __forceinline__ u16x4 narrow(u32x4 v)
	{
		return __builtin_convertvector(v, u16x4);
	}
	static __device__ __forceinline__ i16x4 narrow(i32x4 v)
	{
		return __builtin_convertvector(v, i16x4);
	}
};

// Input loads: streamed once, never re-read, hence non-temporal (same-box A/B
// at the end of round 2: +2 % on the 20-byte-per-sample p2r, +0.4 % on r2p,
// nothing on the constant-vector full recurrence; -DCORDIC_PLAIN_LOADS turns
// it off).  Macros, not function templates: template deduction would strip
// the element alignment off the pointer's vector typedef (Io32 / Io16).
#ifdef CORDIC_PLAIN_LOADS
#define CORDIC_LOAD_IN(ptr) (*(ptr))
#else
#define CORDIC_LOAD_IN(ptr) __builtin_nontemporal_load(ptr)
#endif

// Output stores.  Outputs are written once and never re-read by the engine.
// Measured on MI355X: the non-temporal form is as good or better for the
// VALU-bound kernels (full recurrence).  The table-seeded kernel, which runs
// close to HBM speed, was faster with plain stores while its blocks walked
// contiguous chunks (round 1: 439 vs 406 Gsample/s on cfg2) and is faster with
// non-temporal loads and stores now that they pull tiles in address order
// (profiles/r02/ab_nontemporal.txt) -- so the choice is per kernel and path.
#if defined(CORDIC_FORCE_NT_STORES)
#define CORDIC_STORE_OUT(NT, ptr, val) __builtin_nontemporal_store((val), (ptr))
#elif defined(CORDIC_FORCE_PLAIN_STORES)
#define CORDIC_STORE_OUT(NT, ptr, val) (*(ptr) = (val))
#else
#define CORDIC_STORE_OUT(NT, ptr, val) \
	do { \
		if constexpr (NT) \
			__builtin_nontemporal_store((val), (ptr)); \
		else \
			*(ptr) = (val); \
	} while (0)
#endif

// --------------------------------------------------------- unrolled rotator

// Processes whole 4-sample groups only (nvec of them); the launcher sends the
// 0..3 trailing samples to the generic kernel.  Keeping the tail out of this
// kernel is what lets hipcc emit global_load_dwordx4 / global_store_dwordx4.
template <typename C, int NLIVE, int NGEN, Feed FEED, bool DYN = false,
		typename IO = Io32, bool UG = false>
__global__ __launch_bounds__(kBlock) void rotator_unrolled(CoreParams kp,
		const typename IO::ivec *__restrict__ xin,
		const typename IO::ivec *__restrict__ yin,
		const typename IO::uvec *__restrict__ phin,
		typename IO::ivec *__restrict__ ox,
		typename IO::ivec *__restrict__ oy, size_t nvec)
{
	using T = typename std::conditional<C::wide, int64_t, int32_t>::type;
	using U = typename std::make_unsigned<T>::type;
	using Z = typename std::conditional<C::wide, int64_t, uint32_t>::type;
	constexpr bool kConstXY = (FEED != Feed::PhaseArray_XYArray);

	// Constant-vector feeds: the octant fold (rtl/cordic.v:131-188) can only
	// produce four vectors, rot90^q(e_x, e_y).  Stage them in LDS once per
	// block; per sample the fold is then one ds_read_b128 indexed by the
	// quadrant instead of ~20 VALU selects/negations.
	__shared__ int64_t fold_tab[4][2];
	// Per-sample vectors in a 64-bit container: the fold is a rotation by
	// q * 90 degrees of the 32-bit port values scaled by 2^in_shl,
	//     x0 = c*i_x - s*i_y,  y0 = s*i_x + c*i_y,  (c, s) in {0, +/-2^in_shl},
	// i.e. FOUR v_mad_i64_i32 on the 32-bit inputs (sign extension, the left
	// shift and the negations all inside the multiply) instead of ~25 64-bit
	// mask / xor / subtract instructions; {c, s, -s} come from a 4-entry LDS
	// table indexed by the quadrant.  Exact: every product fits 64 bits and
	// the additions are the same two's-complement additions (in_shl <= 30).
	constexpr bool kMadFold = !kConstXY && C::wide;
	// The left-justified kernels fold the FIRST micro-rotation into the same
	// four multiply-adds.  With in_shl >= 1 the folded vector is even, so
	// (y0 >>> 1) and (x0 >>> 1) are exact and stage 1 (rtl/cordic.v:262-280,
	// shift 1, direction s = +1 where the folded phase is >= 0) is linear:
	//     x1 = x0 - s*(y0/2) = A*i_x - B*i_y,   y1 = y0 + s*(x0/2) = B*i_x + A*i_y
	//     A = c - s*sn/2,  B = sn + s*c/2   (|A|, |B| <= 1.5 * 2^in_shl),
	// and p1 = p0 - s*a_0: eight table rows (quadrant x direction) instead
	// of four, one 32-bit add for the phase, and one GENERAL 64-bit stage
	// (~15 instructions) less per sample.
	const int live = DYN ? kp.nlive : NLIVE;
	const bool fold1 = kMadFold && C::lj != 0 && kp.in_shl >= 1 && live >= 1;
	__shared__ int32_t rot_tab[8][4];
	if constexpr (kConstXY) {
		if (threadIdx.x < 4) {
			const T ex = (T)((U)(T)kp.x0 << kp.in_shl);
			const T ey = (T)((U)(T)kp.y0 << kp.in_shl);
			T fx, fy;
			uint32_t fp;
			fold_octant<T>(ex, ey, (uint32_t)threadIdx.x << 30, fx, fy, fp);
			// (left-justified kernels: stored as the stages carry them)
			fold_tab[threadIdx.x][0] = (int64_t)((uint64_t)(int64_t)(Z)fx << C::lj);
			fold_tab[threadIdx.x][1] = (int64_t)((uint64_t)(int64_t)(Z)fy << C::lj);
		}
		__syncthreads();
	} else if constexpr (kMadFold) {
		if (threadIdx.x < 8) {
			// q = 0: (x, y); 1: (-y, x); 2: (-x, -y); 3: (y, -x)
			const int q = threadIdx.x >> 1;
			const int32_t dir = (threadIdx.x & 1) ? 1 : -1;	// phase >= 0 : < 0
			const int32_t k = (int32_t)(1u << (kp.in_shl & 31));
			const int32_t c = (q == 0) ? k : (q == 2) ? -k : 0;
			const int32_t sn = (q == 1) ? k : (q == 3) ? -k : 0;
			int32_t a = c, b = sn, dp = 0;
			if (fold1) {
				a = c - dir * (sn / 2);
				b = sn + dir * (c / 2);
				dp = -dir * (int32_t)kp.angle[0];
			}
			rot_tab[threadIdx.x][0] = a;
			rot_tab[threadIdx.x][1] = b;
			rot_tab[threadIdx.x][2] = -b;
			rot_tab[threadIdx.x][3] = dp;
		}
		__syncthreads();
	}

	LjRegs ljc{};
	if constexpr (C::lj != 0) {
		ljc.mask = vgpr_const(LjConst<C::lj>::mask);
		ljc.bit = vgpr_const(LjConst<C::lj>::bit);
		ljc.maskbit = vgpr_const(LjConst<C::lj>::mask | LjConst<C::lj>::bit);
	}

	const size_t stride = (size_t)gridDim.x * kBlock;
	size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x;
	// software prefetch: the loads of pass i+1 are issued before the ~750
	// VALU instructions of pass i, so no wave ever parks on HBM latency
	// the mixer (cordic_mix): per-sample vectors, generated phases -- a
	// wave-uniform choice in the instances of the per-sample feed
	const bool gen_phase = FEED == Feed::Nco_ConstXY
		|| (FEED == Feed::PhaseArray_XYArray && kp.xy_nco != 0);
	typename IO::uvec nph{};
	typename IO::ivec nx{}, ny{};
	if (g < nvec) {
		if constexpr (FEED != Feed::Nco_ConstXY)
			if (!gen_phase)
				nph = CORDIC_LOAD_IN(&phin[g]);
		if constexpr (!kConstXY) {
			nx = CORDIC_LOAD_IN(&xin[g]);
			ny = CORDIC_LOAD_IN(&yin[g]);
		}
	}
	for (; g < nvec; g += stride) {
		const u32x4 tph = IO::widen(nph);
		const i32x4 tx = IO::widen(nx), ty = IO::widen(ny);
		const size_t gn = g + stride;
		if (gn < nvec) {
			if constexpr (FEED != Feed::Nco_ConstXY)
				if (!gen_phase)
					nph = CORDIC_LOAD_IN(&phin[gn]);
			if constexpr (!kConstXY) {
				nx = CORDIC_LOAD_IN(&xin[gn]);
				ny = CORDIC_LOAD_IN(&yin[gn]);
			}
		}

		uint32_t P[kVec];
		if (gen_phase) {
			const uint32_t s0 = (uint32_t)(kp.index0 + g * kVec);
			P[0] = kp.phase0 + s0 * kp.fcw;
#pragma unroll
			for (int v = 1; v < kVec; v++)
				P[v] = P[v - 1] + kp.fcw;
		} else {
			left_justify(tph, P, kp.pw_shl);
		}

		int64_t x[kVec], y[kVec], p[kVec];
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			if constexpr (kConstXY) {
				// q = quadrant of (P + 45 deg); p = P - q * 90 deg
				const uint32_t pb = P[v] + 0x20000000u;
				const uint32_t q16 = ((uint32_t)((int32_t)pb >> 26)) & 0x30u;
				const int64_t *e = reinterpret_cast<const int64_t *>(
					reinterpret_cast<const char *>(&fold_tab[0][0]) + q16);
				x[v] = e[0];
				y[v] = e[1];
				p[v] = (int64_t)((pb & 0x3fffffffu) - 0x20000000u);
			} else if constexpr (kMadFold) {	// launcher: in_shl <= 30
				const int32_t ix = sext32(tx[v], kp.iw);
				const int32_t iy = sext32(ty[v], kp.iw);
				const uint32_t pb = P[v] + 0x20000000u;
				// row = quadrant (bits 31..30) x direction (bit 29 of pb
				// set <=> folded phase >= 0), 16 bytes each
				const uint32_t row = (pb >> 25) & 0x70u;
				const i32x4 m = *reinterpret_cast<const i32x4 *>(
					reinterpret_cast<const char *>(&rot_tab[0][0]) + row);
				int64_t fx = op_mul(iy, m[2]);	// -B * i_y
				op_mad(fx, ix, m[0]);		// + A * i_x
				int64_t fy = op_mul(iy, m[0]);	//  A * i_y
				op_mad(fy, ix, m[1]);		// + B * i_x
				x[v] = fx;
				y[v] = fy;
				p[v] = (int64_t)((pb & 0x3fffffffu) - 0x20000000u
						+ (uint32_t)m[3]);
			} else {
				const int32_t ix = sext32(tx[v], kp.iw);
				const int32_t iy = sext32(ty[v], kp.iw);
				const T ex = (T)((U)(T)ix << kp.in_shl);
				const T ey = (T)((U)(T)iy << kp.in_shl);
				T fx, fy;
				uint32_t fp;
				fold_octant<T>(ex, ey, P[v], fx, fy, fp);
				x[v] = (int64_t)(Z)fx;
				y[v] = (int64_t)(Z)fy;
				p[v] = (int64_t)fp;
			}
		}

		i32x4 rx, ry;
		if constexpr (C::lj == 0) {
			RotChain<C, NLIVE, NGEN, 0, DYN>::run(x, y, p, kp);
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				rx[v] = round_to_ow<T>((T)x[v], kp);
				ry[v] = round_to_ow<T>((T)y[v], kp);
			}
		} else {
			constexpr int LJ = C::lj;
			// left-justified from the first stage on (rot_stage_lj_early
			// serves the stages whose shift is below 32-LJ); the constant
			// vectors come out of their table already shifted
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				if constexpr (!kConstXY) {
					x[v] = (int64_t)((uint64_t)x[v] << LJ);
					y[v] = (int64_t)((uint64_t)y[v] << LJ);
				}
				p[v] = (int64_t)((uint64_t)(int64_t)(int32_t)(uint32_t)p[v]
						<< LjPhase<LJ>::ps);
			}
			if (!fold1)	// else stage 1 came out of the fold's multiply-adds
				RotChainLJ<LJ, (NLIVE < 1 ? NLIVE : 1), 0, DYN>::run(x, y, p, kp, ljc);
			RotChainLJ<LJ, NLIVE, 1, DYN>::run(x, y, p, kp, ljc);
			if (kp.r_lj == 32) {
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					rx[v] = round_to_ow_lj32<LJ>(x[v], kp);
					ry[v] = round_to_ow_lj32<LJ>(y[v], kp);
				}
			} else if (kp.r_lj > 32 && kp.r < 31) {
				const uint32_t sh = (uint32_t)kp.r_lj - 32u;
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					rx[v] = round_to_ow_lj_hi<LJ>(x[v], kp, sh);
					ry[v] = round_to_ow_lj_hi<LJ>(y[v], kp, sh);
				}
			} else {
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					rx[v] = round_to_ow_lj<LJ>(x[v], kp);
					ry[v] = round_to_ow_lj<LJ>(y[v], kp);
				}
			}
		}
		apply_unit_gain<UG>(rx, kp);
		apply_unit_gain<UG>(ry, kp);
		CORDIC_STORE_OUT(true, &ox[g], IO::narrow(rx));
		CORDIC_STORE_OUT(true, &oy[g], IO::narrow(ry));
	}
}

// ------------------------------------------ seeded rotator (const vector)
//
// Constant i_xval / i_yval: the first M micro-rotations are replaced by a
// table lookup (see cordic_plan.cpp for the argument and the table layout).
// Per block: the bucket table is copied to LDS and the (x_M, y_M) seeds of
// every (octant, leaf) are computed with the exact recurrence -- nothing is
// cached between launches.  Per sample: one bucket read, one compare, one
// seed read, then stages M .. NLIVE-1 as in rotator_unrolled.
struct SeedArgs {
	const uint32_t *table;	// device copy of build_seed_table()'s words
	int32_t	S;		// bucket shift
	int32_t	nbuckets;
	int32_t	nleaves;
	// Tile queue: kQueueCounters zeroed counters, kQueueStride words apart
	// (one cache line each), or NULL for the static chunk-per-block sweep.
	uint32_t *queue;
	// direction tails behind the seeds (cordic_internal.h: DtInfo; the words
	// they refer to follow the seed table in `table`)
	DtInfo	dt;
	// Round 5: the block's LDS as the prologue leaves it -- buckets, seeds of
	// every (octant, leaf), tail tables -- kept by the plan per constant vector
	// (cordic_kernels.hip: SeedImages).  `image` != NULL: the prologue is a
	// copy of image_words words (136 KiB from L2 instead of ~10 us of exact
	// recurrences per block).  `image_out` != NULL: BUILD mode -- one block runs
	// the ordinary prologue, writes its LDS there and returns; the image is
	// made by the very code whose result it replaces.
	const uint32_t *image = nullptr;
	uint32_t *image_out = nullptr;
	uint32_t image_words = 0;
	// Round 5, many small jobs in one launch (cordic_jobset): `tiles` != NULL
	// = ntiles descriptors of 32 bytes, one per tile of the whole batch in
	// queue order (cordic_internal.h: TileDesc); dynamic-exit instances only.
	const uint32_t *tiles = nullptr;
	uint32_t ntiles = 0;
};

// LDS of the direction tails of a seeded kernel: per group its buckets
// (8 bytes each, aligned to their total size so that the bucket address is
// (index & mask) | base) and its leaf entries (dt_entry_dwords each).  `at` = first free byte
// behind the seeds and the tile-id slots; returns the new end.
__host__ __device__ inline uint32_t dt_lds_layout(const DtInfo &dt, uint32_t at,
		uint32_t *bucket_base, uint32_t *leaf_base)
{
	for (int g = 0; g < dt.n; g++) {
		const uint32_t bb = (uint32_t)dt.lv[g].nb * 8u;
		at = (at + bb - 1u) & ~(bb - 1u);
		if (bucket_base) bucket_base[g] = at;
		// the leaf entries are read with ds_read_b128 / b96: 16-byte
		// aligned also behind a single 8-byte bucket
		at = (at + bb + 15u) & ~15u;
		if (leaf_base) leaf_base[g] = at;
		at += (uint32_t)dt.lv[g].nl * (uint32_t)dt_entry_dwords(dt.lv[g].t) * 4u;
	}
	return at;
}

constexpr int kQueueCounters = 8;	// one per XCD
constexpr int kQueueStride = 64;	// words between counters
constexpr int kQueueDoneWord = 32;	// blocks that have left the queue

// The queue resets ITSELF: every block, once it has drawn its last ticket
// (all of lane 0's atomics have returned by then), counts itself out, and the
// last one to do so zeroes the counters for the next launch.  No memset node
// in front of the kernel -- a HIP graph that holds a launch replays correctly
// (a captured hipMemsetAsync of the counters did not re-run on replay) -- and
// no extra kernel boundary.  The host zeroes a fresh ring once.
__device__ __forceinline__ void queue_leave(uint32_t *queue)
{
	const uint32_t left = atomicAdd(&queue[kQueueDoneWord], 1u);
	if (left == gridDim.x - 1) {
		for (int j = 0; j < kQueueCounters; j++)
			atomicExch(&queue[j * kQueueStride], 0u);
		atomicExch(&queue[kQueueDoneWord], 0u);
	}
}

// XCC_ID of the XCD this wave runs on (affinity only, never correctness)
__device__ __forceinline__ uint32_t xcc_id()
{
	uint32_t v;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
	return v & 0xfu;
}

// Generic form of the address-ordered tile queue for kernels without a
// cross-tile prefetch (the table cores): persistent BS-thread blocks pull tile
// numbers from the per-XCD counters in `queue` and call body(tile) for each;
// `slot` is three words of LDS.  See rotator_seeded for the why.
template <int BS, typename F>
__device__ __forceinline__ void for_each_queued_tile(uint32_t *queue,
		volatile uint32_t *slot, uint32_t ntiles, F body)
{
	constexpr uint32_t kEnd = 0xffffffffu;
	const uint32_t per = (ntiles + kQueueCounters - 1) / kQueueCounters;
	uint32_t home = 0, tried = 0;		// lane 0 of the block only
	auto grab = [&]() -> uint32_t {
		while (tried < (uint32_t)kQueueCounters) {
			const uint32_t j = (home + tried) % kQueueCounters;
			const uint32_t lo = j * per;
			const uint32_t cnt = lo >= ntiles ? 0u
				: (ntiles - lo < per ? ntiles - lo : per);
			if (cnt != 0) {
				const uint32_t t = atomicAdd(&queue[j * kQueueStride], 1u);
				if (t < cnt)
					return lo + t;
			}
			tried++;
		}
		return kEnd;
	};
	auto lds_barrier = [] {		// LDS only: no wait for global stores
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	};
	if (threadIdx.x == 0) {
		home = xcc_id() % kQueueCounters;
		slot[0] = grab();
		slot[1] = grab();
	}
	lds_barrier();
	uint32_t cur = slot[0];
	int ring = 0;
	while (cur != kEnd) {
		const uint32_t nxt = slot[(ring + 1) % 3];
		body(cur);
		if (threadIdx.x == 0)
			slot[(ring + 2) % 3] = grab();
		lds_barrier();
		cur = nxt;
		ring = (ring + 1) % 3;
	}
	if (threadIdx.x == 0)
		queue_leave(queue);
}

// The same with the next tile's input fetched before this tile is worked on
// (round 3: the table kernels run one or two 1024-thread blocks per CU, so
// without it every tile waits out its own load): `load(tile)` returns the
// lane's input registers, `body(tile, regs)` consumes them.
template <int BS, typename R, typename LD, typename F>
__device__ __forceinline__ void for_each_queued_tile_prefetched(uint32_t *queue,
		volatile uint32_t *slot, uint32_t ntiles, LD load, F body)
{
	constexpr uint32_t kEnd = 0xffffffffu;
	const uint32_t per = (ntiles + kQueueCounters - 1) / kQueueCounters;
	uint32_t home = 0, tried = 0;		// lane 0 of the block only
	auto grab = [&]() -> uint32_t {
		while (tried < (uint32_t)kQueueCounters) {
			const uint32_t j = (home + tried) % kQueueCounters;
			const uint32_t lo = j * per;
			const uint32_t cnt = lo >= ntiles ? 0u
				: (ntiles - lo < per ? ntiles - lo : per);
			if (cnt != 0) {
				const uint32_t t = atomicAdd(&queue[j * kQueueStride], 1u);
				if (t < cnt)
					return lo + t;
			}
			tried++;
		}
		return kEnd;
	};
	auto lds_barrier = [] {		// LDS only: no wait for global stores
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	};
	if (threadIdx.x == 0) {
		home = xcc_id() % kQueueCounters;
		slot[0] = grab();
		slot[1] = grab();
	}
	lds_barrier();
	uint32_t cur = slot[0];
	int ring = 0;
	R regs{};
	if (cur != kEnd)
		regs = load(cur);
	while (cur != kEnd) {
		const uint32_t nxt = slot[(ring + 1) % 3];
		R pre{};
		if (nxt != kEnd)
			pre = load(nxt);
		body(cur, regs);
		if (threadIdx.x == 0)
			slot[(ring + 2) % 3] = grab();
		lds_barrier();
		cur = nxt;
		regs = pre;
		ring = (ring + 1) % 3;
	}
	if (threadIdx.x == 0)
		queue_leave(queue);
}


template <typename C, int NLIVE, int M, Feed FEED, bool DYN = false,
		typename IO = Io32, bool UG = false, bool DT = false, bool DESC = false>
__global__ __launch_bounds__(kSeedBlock) void rotator_seeded(CoreParams kp,
		SeedArgs sa, const typename IO::uvec *__restrict__ phin,
		typename IO::ivec *__restrict__ ox,
		typename IO::ivec *__restrict__ oy, size_t nvec)
{
	using T = typename std::conditional<C::wide, int64_t, int32_t>::type;
	using U = typename std::make_unsigned<T>::type;
	static_assert(FEED != Feed::PhaseArray_XYArray, "constant vector only");
	static_assert(M <= NLIVE, "seed deeper than the core");

	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	// ---- tile geometry of the queued sweep (see "Work distribution" below),
	// ahead of the prologue: round 5 hands every block its first two tiles
	// WITHOUT an atomic -- ticket numbers 0 .. 2 R_j - 1 of range j belong to
	// the R_j blocks whose home is j (home = blockIdx mod 8, rank = blockIdx
	// div 8: ranks r and r + R_j), the counters hand out 2 R_j + k -- so the
	// first tile is known from blockIdx alone and its phases are requested
	// BEFORE the prologue stages the table: a small batch no longer pays two
	// serial atomic round trips and a load latency behind the prologue.
	constexpr uint32_t kTileVecs = (uint32_t)kSeedBlock * kSeedSub;
	static_assert(kTileVecs == kJobTileVecs, "cordic_jobset cuts its tiles on the host");
	const uint32_t ntiles = (uint32_t)((nvec + kTileVecs - 1) / kTileVecs);
	const uint32_t per = (ntiles + kQueueCounters - 1) / kQueueCounters;
	const uint32_t lane = threadIdx.x;
	// vectors of the tile that exist (only the batch's last tile is partial)
	const uint32_t last_live = ntiles
		? (uint32_t)(nvec - (size_t)(ntiles - 1) * kTileVecs) : 0u;
	auto live = [&](uint32_t tile) -> uint32_t {
		return tile == ntiles - 1 ? last_live : kTileVecs;
	};
	// a lane's vector in row s of a tile, clamped into the tile's live
	// part (only the batch's last tile is partial): the loads need no
	// predicate, so nothing has to be preserved around them
	auto row_vec = [&](uint32_t tile, int s) -> size_t {
		uint32_t l = lane + (uint32_t)s * kSeedBlock;
		const uint32_t top = live(tile) - 1u;	// block-uniform
		// (spelled out: left to itself the compiler selects with
		// v_cndmask_b32, ~23 cycles per wave-instruction here, §4.1)
		asm("v_min_u32 %0, %1, %2" : "=v"(l) : "v"(l), "s"(top));
		return (size_t)tile * kTileVecs + l;
	};
	const uint32_t q_home = blockIdx.x % kQueueCounters;
	const uint32_t q_rank = blockIdx.x / kQueueCounters;
	// blocks whose home is j
	auto q_blocks = [&](uint32_t j) -> uint32_t {
		return gridDim.x > j ? (gridDim.x - j + kQueueCounters - 1) / kQueueCounters : 0u;
	};
	typename IO::uvec pa[kSeedSub] = {};
	bool early = false;
	if (!DYN && sa.queue != nullptr && sa.image_out == nullptr
			&& sa.tiles == nullptr) {
		const uint32_t lo = q_home * per;
		const uint32_t cnt = lo >= ntiles ? 0u : (ntiles - lo < per ? ntiles - lo : per);
		early = q_rank < cnt;		// else: fewer tiles than blocks here
		if constexpr (FEED != Feed::Nco_ConstXY) {
			if (early) {
#pragma unroll
				for (int s = 0; s < kSeedSub; s++)
					pa[s] = __builtin_nontemporal_load(&phin[row_vec(lo + q_rank, s)]);
			}
		}
	}
	uint32_t *lds_buckets = lds;				// nbuckets x 2
	uint32_t *lds_seeds = lds + (size_t)sa.nbuckets * 2;	// 4L x 4 words
	const int L = sa.nleaves;
	const uint32_t seed_base = (uint32_t)sa.nbuckets * 8u;

	constexpr int kDtR = NLIVE - M;
	constexpr int kDtN = DT ? dt_levels(kDtR) : 0;
	uint32_t dt_bk[kDtMaxLevels] = {}, dt_lf[kDtMaxLevels] = {};
	if constexpr (DT) {
		static_assert((C::lj != 0 || !C::wide) && !DYN && !UG,
			"direction tails: static left-justified or 32-bit instances");
		static_assert(kDtN >= 1 && kDtN <= kDtMaxLevels, "no group to look up");
		dt_lds_layout(sa.dt, seed_base + 4u * (uint32_t)L * 16u + 16u, dt_bk, dt_lf);
	}
	if (sa.image != nullptr) {
		// the plan holds this prologue's result for (x0, y0): copy it.  All of
		// a thread's loads are in flight together (a plain loop waits out the
		// L2 latency once per 16 bytes: ~9 round trips, 5 us; profiles/r05/
		// small_batch.txt): indices clamped so that the loads need no
		// predicate, stores predicated.  160 KiB / 16 B / 1024 threads = 10.
		const u32x4 *src = reinterpret_cast<const u32x4 *>(sa.image);
		u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
		const uint32_t nv = sa.image_words / 4u;
		constexpr int kCopies = (int)(CORDIC_SEED_LDS_BYTES / 16u / kSeedBlock);
		u32x4 r[kCopies];
#pragma unroll
		for (int k = 0; k < kCopies; k++) {
			uint32_t i = threadIdx.x + (uint32_t)k * kSeedBlock;
			i = i < nv ? i : nv - 1u;
			r[k] = src[i];
		}
#pragma unroll
		for (int k = 0; k < kCopies; k++) {
			const uint32_t i = threadIdx.x + (uint32_t)k * kSeedBlock;
			if (i < nv)
				dst[i] = r[k];
		}
	} else {
	// bucket entries as the lookup wants them: {bound - 1, byte address of
	// the bucket's first leaf in quadrant 0}.  A bucket without a boundary
	// gets its own last phase as the bound (the compare below looks at bit
	// 29 of a difference, so the bound has to stay within 2^29 of every
	// phase of the bucket -- the table's 0x7fffffff would not).
	for (int i = threadIdx.x; i < sa.nbuckets * 2; i += kSeedBlock) {
		const uint32_t w = sa.table[4 + i];
		const uint32_t b = (uint32_t)i >> 1;
		lds_buckets[i] = (i & 1) ? seed_base + w * 16u
			: (w == 0x7fffffffu ? ((b + 1u) << sa.S) - 1u : w);
	}
	const uint32_t *leafmeta = sa.table + 4 + (size_t)sa.nbuckets * 2;
	for (int e = threadIdx.x; e < 4 * L; e += kSeedBlock) {
		const int q = e / L, j = e - q * L;
		const uint32_t pattern = leafmeta[2 * j];
		const T ex = (T)((U)(T)kp.x0 << kp.in_shl);
		const T ey = (T)((U)(T)kp.y0 << kp.in_shl);
		T x, y;
		uint32_t fp;
		fold_octant<T>(ex, ey, (uint32_t)q << 30, x, y, fp);
		for (int i = 0; i < M; i++) {		// rtl/cordic.v:262-280
			const int k = (i + 1 > (int)sizeof(T) * 8 - 1)
					? (int)sizeof(T) * 8 - 1 : i + 1;
			const T sy = y >> k, sx = x >> k;
			if ((pattern >> (M - 1 - i)) & 1u) {	// phase >= 0
				x = (T)((U)x - (U)sy);
				y = (T)((U)y + (U)sx);
			} else {
				x = (T)((U)x + (U)sy);
				y = (T)((U)y - (U)sx);
			}
		}
		// 16-byte seed entry.
		//  Narrow32: {x, y, off + 2^29, 0}.
		//  WideLJ  : {x~ lo, x~ hi, y~ lo, y~ hi} with x~ = x << LJ, so that
		//            one ds_read_b128 lands in the two register pairs of the
		//            stage chain.  The low LJ (>= 29) bits of x~ and y~ are
		//            zero and NOTHING downstream looks at them -- every stage
		//            adds a multiple of 2^LJ and shifts the high word, the
		//            rounding adds a multiple of 2^LJ and drops at least
		//            LJ+1 bits -- so the low 29 bits of x~ lo carry
		//            (off + 2^29) mod 2^29.  That is enough: the host builds
		//            a table only if every residual fits 29 bits signed
		//            (cordic_plan.cpp).
		uint32_t *d = lds_seeds + (size_t)e * 4;
		if constexpr (C::lj != 0) {
			const uint64_t xs = (uint64_t)(int64_t)x << C::lj;
			const uint64_t ys = (uint64_t)(int64_t)y << C::lj;
			d[0] = (uint32_t)xs | (leafmeta[2 * j + 1] & 0x1fffffffu);
			d[1] = (uint32_t)(xs >> 32);
			d[2] = (uint32_t)ys;
			d[3] = (uint32_t)(ys >> 32);
		} else {
			d[0] = (uint32_t)x;
			d[1] = (uint32_t)y;
			d[2] = leafmeta[2 * j + 1];		// off + 2^29
			d[3] = 0;
		}
	}
	// Direction tails (cordic_internal.h: dt_levels; cordic_plan.cpp): per
	// group of stages behind the seeds a bucket table over the biased
	// residual u and leaf entries {ns_0, s_0, ns_1, s_1, ... (the multipliers
	// -/+ s_j 2^LJ of the group's stages), off'} with u_next = u - off'
	// (cordic_internal.h: dt_pairs, dt_entry_dwords).
	if constexpr (DT) {
#pragma unroll
		for (int g = 0; g < kDtN; g++) {
			const DtLevel lv = sa.dt.lv[g];
			const uint32_t *src = sa.table + lv.word;
			uint32_t *bk = lds + dt_bk[g] / 4u;
			for (int i = threadIdx.x; i < lv.nb * 2; i += kSeedBlock) {
				const uint32_t w = src[i];
				bk[i] = (i & 1) ? dt_lf[g] + w * (uint32_t)dt_entry_dwords(lv.t) * 4u : w;
			}
			const uint32_t *lsrc = src + (size_t)lv.nb * 2;
			uint32_t *lf = lds + dt_lf[g] / 4u;
			for (int e = threadIdx.x; e < lv.nl; e += kSeedBlock) {
				const uint32_t pat = lsrc[2 * e];
				uint32_t *d = lf + (size_t)e * dt_entry_dwords(lv.t);
				const int np = dt_pairs(lv.t);
				for (int jj = 0; jj < lv.t; jj++) {
					// bit set: residual >= 0 at that stage, s = +1
					const bool pos = (pat >> (lv.t - 1 - jj)) & 1u;
					// +/- 2^LJ (left-justified pairs) or +/- 1 (Narrow32)
					const uint32_t plus = C::lj != 0 ? LjConst<C::lj>::bit : 1u;
					const uint32_t minus = C::lj != 0
						? (LjConst<C::lj>::mask | LjConst<C::lj>::bit) : 0xffffffffu;
					if (jj < np) {
						d[2 * jj + 0] = pos ? minus : plus;	// -s 2^LJ (x)
						d[2 * jj + 1] = pos ? plus : minus;	//  s 2^LJ (y)
					} else {
						d[np + jj] = pos ? plus : minus;
					}
				}
				d[lv.t + np] = lsrc[2 * e + 1];
			}
		}
	}
	}	// (sa.image == nullptr)
	__syncthreads();
	if (sa.image_out != nullptr) {
		// build mode (one block, no samples): the LDS as it stands IS the image
		const u32x4 *src = reinterpret_cast<const u32x4 *>(lds);
		u32x4 *dst = reinterpret_cast<u32x4 *>(sa.image_out);
		for (uint32_t i = threadIdx.x; i < sa.image_words / 4u; i += kSeedBlock)
			dst[i] = src[i];
		return;
	}

	LjRegs ljc{};
	if constexpr (C::lj != 0) {
		ljc.mask = vgpr_const(LjConst<C::lj>::mask);
		ljc.bit = vgpr_const(LjConst<C::lj>::bit);
		ljc.maskbit = vgpr_const(LjConst<C::lj>::mask | LjConst<C::lj>::bit);
	}
	uint32_t dt_maskv[kDtMaxLevels] = {};
	if constexpr (DT) {
#pragma unroll
		for (int g = 0; g < kDtN; g++)
			dt_maskv[g] = vgpr_const(((uint32_t)sa.dt.lv[g].nb - 1u) << 3);
	}
	const uint32_t k45 = vgpr_const(0x20000000u);	// 45 degrees (VOP3 has no literal)
	const uint32_t bshift = (uint32_t)sa.S - 3;	// bucket -> byte offset
	const uint32_t bmask = ((uint32_t)sa.nbuckets - 1u) << 3;
	const uint32_t qstride = (uint32_t)L * 16u;
	// LDS is addressed by byte offset: this kernel has no static LDS, so the
	// dynamic array starts at LDS address 0 and the per-sample address needs
	// no base added to it (the add of the array's link-time address was one
	// VALU instruction per read).  Checked once per wave.
	typedef const __attribute__((address_space(3))) u32x4 lds_entry;
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
	typedef const __attribute__((address_space(3))) u32x2 lds_bucket;
	if ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)lds != 0u)
		__builtin_trap();

	// One tile pass: 4 samples per lane of vector g, phases in tph; results in
	// rx / ry (the caller stores them).
	// pb = folded phase + 45 deg: bits 31..30 the quadrant q, bits 29..0
	// r = p0 + 2^29.  Nothing below needs r on its own -- the bucket index is
	// a bit field of pb, the compare and the residual work modulo 2^30 resp.
	// 2^29 -- which saves the masking.  Kept apart from the rotation so that
	// the tile loop can consume a row's phases BEFORE it prefetches the next
	// tile's into the same registers (no copies between passes).
	// NCO feed: the phase of vector g = g0 + lane is  phase0 + fcw * (index0 +
	// 4 g0)  [block-uniform when g0 is: scalar arithmetic]  + (4 fcw) * lane
	// [a per-lane constant]: one vector add per row instead of a 64-bit index,
	// a v_mul_lo_u32 and two more adds.
	const uint32_t lane4f = (FEED == Feed::Nco_ConstXY)
			? threadIdx.x * (uint32_t)kVec * kp.fcw : 0u;
	auto fold_pb = [&](size_t g0, uint32_t lane_part, const u32x4 tph,
			uint32_t (&pb)[kVec]) {
		if constexpr (FEED == Feed::Nco_ConstXY) {
			const uint32_t s0 = (uint32_t)(kp.index0 + g0 * kVec);
			pb[0] = ((kp.phase0 + 0x20000000u) + s0 * kp.fcw) + lane_part;
#pragma unroll
			for (int v = 1; v < kVec; v++)
				pb[v] = pb[v - 1] + kp.fcw;
		} else {
#pragma unroll
			for (int v = 0; v < kVec; v++)
				asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(pb[v])
					: "v"(tph[v]), "s"(kp.pw_shl), "v"(k45));
		}
	};
	// Direction tails trade VALU instructions for LDS reads (a bucket and up
	// to 60 bytes of multipliers per group and sample).  Groups of up to five
	// stages have at most 32 leaves: their lookups win on ANY phases (unrelated
	// ones included: 16-stage cores +1...+2.6 %, 19-21 stages +3...+9 %), and
	// cores made of such groups look up on every row (dt_always).  With 64-128
	// leaves per group unrelated phases collide on the banks and the lookup
	// loses a few per cent to the recurrence, while a ramp of any slope or a
	// slow NCO gains: there each wave decides per row -- first and last phase
	// less than 2^24 apart -> tails, else the phase recurrence (two lane reads
	// and a scalar compare; a row of unrelated phases passes once in 2^7).
	auto row_is_coherent = [](const uint32_t (&pb)[kVec]) -> bool {
		const uint32_t pf = __builtin_amdgcn_readfirstlane(pb[0]);
		const uint32_t pl = __builtin_amdgcn_readlane(pb[kVec - 1], 63);
		if constexpr (kDtCoherentLog2 >= 31)
			return true;		// (A/B builds: every row)
		return (pl - pf + (1u << kDtCoherentLog2)) < (2u << kDtCoherentLog2);
	};
	// One row: 4 samples per lane with their folded phases in pb; results in
	// rx / ry (the caller stores them).
	auto pass_pb = [&](auto TAILS_, const uint32_t (&pb)[kVec], i32x4 &rx, i32x4 &ry) {
		// (two instantiations: with the direction tails or with the phase
		// recurrence behind the seeds -- decided per row by the caller)
		constexpr bool kTails = DT && decltype(TAILS_)::value;
		// three passes so that the four bucket reads, then the four seed
		// reads, are in flight together (one s_waitcnt each, not eight).
		// Per sample: lshl_add, lshr, lshr, and | sub, bfe, lshl_add,
		// mad_u24 | sub, bfe, lshl = 11 VALU instructions.
		int64_t x[kVec], y[kVec], p[kVec];
		uint32_t q[kVec];
		u32x2 bk[kVec];
		u32x4 se[kVec];
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			q[v] = pb[v] >> 30;
			bk[v] = *(lds_bucket *)(uintptr_t)((pb[v] >> bshift) & bmask);
		}
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			// r >= bound  <=>  (bound-1) - r < 0; the difference is
			// smaller than a bucket, so its sign is also its bit 29, where
			// the quadrant bits of pb do not reach
			// (written out: the compiler's own selection is one
			// instruction longer)
			uint32_t c, a;
			asm("v_bfe_u32 %0, %1, 29, 1" : "=v"(c) : "v"(bk[v][0] - pb[v]));
			asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(a) : "v"(c), "v"(bk[v][1]));
			asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a)
				: "v"(q[v]), "s"(qstride), "v"(a));
			se[v] = *(lds_entry *)(uintptr_t)a;
		}
		if constexpr (kTails) {
			// ---- the stages behind the seeds take their multipliers from
			// tables indexed by the biased residual u = p_M + bias0 >= 0
			constexpr int LJ = C::lj;
			uint32_t u[kVec];
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				if constexpr (C::lj != 0) {
					x[v] = (int64_t)(((uint64_t)se[v][1] << 32) | se[v][0]);
					y[v] = (int64_t)(((uint64_t)se[v][3] << 32) | se[v][2]);
					asm("v_bfe_u32 %0, %1, 0, 29" : "=v"(u[v])
						: "v"(pb[v] - se[v][0] + sa.dt.bias0));
				} else {
					x[v] = (int64_t)se[v][0];
					y[v] = (int64_t)se[v][1];
					asm("v_bfe_u32 %0, %1, 0, 30" : "=v"(u[v])
						: "v"(pb[v] - se[v][2] + sa.dt.bias0));
				}
			}
			// one group: bucket read, compare, entry read, T stages of 4
			// instructions; per sample lshr, and_or | sub, lshr, lshl_add
			// (| sub for the next group's u) = 5-6 VALU instructions.
			// (The reads of a group depend on each other and on the group
			// before; pipelining them by hand across the groups, with
			// scheduling fences, measured 1.6 % SLOWER -- the four waves per
			// SIMD cover the latency; profiles/r03/ab_tails.txt, form 7.
			// Multipliers in SGPRs for rows that sit on ONE entry of the
			// first group: +4 % without a test (form 11), but with the exact
			// test, the entry read once and 13-15 v_readfirstlane the row
			// stalls on that chain: -3.5...-6 % (form 18).  Not kept.)
			auto group = [&](auto G_) {
				constexpr int G = decltype(G_)::value;
				constexpr int T = dt_size(kDtR, G);
				constexpr int K0 = M + dt_first(kDtR, G);	// stages done
				constexpr bool more = (G + 1 < kDtN) || dt_rest(kDtR) > 0;
				const uint32_t sh3 = (uint32_t)sa.dt.lv[G].shift - 3u;
				u32x2 b2[kVec];
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					uint32_t a;
					asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a)
						: "v"(u[v] >> sh3), "v"(dt_maskv[G]), "s"(dt_bk[G]));
					b2[v] = *(lds_bucket *)(uintptr_t)a;
				}
				constexpr int P = dt_pairs(T);
				constexpr int W = T + P + (more ? 1 : 0);	// dwords used
				constexpr int kStride = dt_entry_dwords(T) * 4;
				uint32_t ea[kVec];
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					const uint32_t c = (b2[v][0] - u[v]) >> 31;	// u >= bound
					if constexpr ((kStride & (kStride - 1)) == 0)
						asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(ea[v])
							: "v"(c), "n"(__builtin_ctz(kStride)), "v"(b2[v][1]));
					else if constexpr (kStride <= 64)
						asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ea[v])
							: "v"(c), "n"(kStride), "v"(b2[v][1]));
					else		// (no inline constant above 64)
						asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ea[v])
							: "v"(c), "s"(kStride), "v"(b2[v][1]));
				}
				uint32_t en[kVec][16];
#pragma unroll
				for (int v = 0; v < kVec; v++) {
#pragma unroll
					for (int at = 0; at < W; at += 4) {
						// only the dwords that are used: the LDS returns
						// 128 bytes a cycle to the CU whatever the lanes
						// ask for
						if (W - at >= 4) {
							const u32x4 t4 = *(lds_entry *)(uintptr_t)(ea[v] + 4u * at);
							en[v][at] = t4[0]; en[v][at + 1] = t4[1];
							en[v][at + 2] = t4[2]; en[v][at + 3] = t4[3];
						} else if (W - at == 3) {
							// (round 4) three dwords left: still a b128 -- a
							// ds_read_b96 is served in 8 lane groups of 8
							// (8 LDS cycles), a b128 in 4 of 16
							// (MI355X_MICROARCH.md, LDS); the entry's stride
							// is a multiple of four dwords, the fourth is
							// padding inside it
#ifdef CORDIC_DT_B96	/* A/B: the round-3 form */
							typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
							const u32x3 t4 = *(const __attribute__((address_space(3)))
								u32x3 *)(uintptr_t)(ea[v] + 4u * at);
#else
							const u32x4 t4 = *(lds_entry *)(uintptr_t)(ea[v] + 4u * at);
#endif
							en[v][at] = t4[0]; en[v][at + 1] = t4[1];
							en[v][at + 2] = t4[2];
						} else if (W - at == 2) {
							const u32x2 t2 = *(lds_bucket *)(uintptr_t)(ea[v] + 4u * at);
							en[v][at] = t2[0]; en[v][at + 1] = t2[1];
						} else {
							en[v][at] = *(const __attribute__((address_space(3)))
								uint32_t *)(uintptr_t)(ea[v] + 4u * at);
						}
					}
				}
				auto stage = [&](auto J_) {
					constexpr int J = decltype(J_)::value;
					if constexpr (J < T) {
#pragma unroll
						for (int v = 0; v < kVec; v++) {
							const int32_t ns_ = (J < P) ? (int32_t)en[v][2 * J]
									: -(int32_t)en[v][P + J];
							const int32_t s_ = (J < P) ? (int32_t)en[v][2 * J + 1]
									: (int32_t)en[v][P + J];
							if constexpr (C::lj != 0)
								rot_stage_lj_dir<LJ, K0 + J + 1>(x[v], y[v], ns_, s_);
							else
								rot_stage_dir32<K0 + J + 1>(x[v], y[v], ns_, s_);
						}
					}
				};
				stage(std::integral_constant<int, 0>{});
				stage(std::integral_constant<int, 1>{});
				stage(std::integral_constant<int, 2>{});
				stage(std::integral_constant<int, 3>{});
				stage(std::integral_constant<int, 4>{});
				stage(std::integral_constant<int, 5>{});
				stage(std::integral_constant<int, 6>{});
				if constexpr (more) {
#pragma unroll
					for (int v = 0; v < kVec; v++)
						u[v] -= en[v][T + P];
				}
			};
			if constexpr (kDtN > 0) group(std::integral_constant<int, 0>{});
			if constexpr (kDtN > 1) group(std::integral_constant<int, 1>{});
			if constexpr (kDtN > 2) group(std::integral_constant<int, 2>{});
			if constexpr (kDtN > 3) group(std::integral_constant<int, 3>{});
			constexpr int kRest = dt_rest(kDtR);
			if constexpr (kRest > 0) {
				// the last stages on the residual phase itself
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					const int32_t r = (int32_t)(u[v] - sa.dt.bias_last);
					if constexpr (C::lj != 0)
						p[v] = (int64_t)(((uint64_t)(uint32_t)(r >> 1) << 32)
								| ((uint32_t)r << 31));
					else
						p[v] = (int64_t)r;
				}
				if constexpr (C::lj != 0)
					RotChainLJ<LJ, NLIVE, NLIVE - kRest, false>::run(x, y, p, kp, ljc);
				else
					RotChain<C, NLIVE, 0, NLIVE - kRest, false>::run(x, y, p, kp);
			}
		} else {
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			if constexpr (C::lj != 0) {
				x[v] = (int64_t)(((uint64_t)se[v][1] << 32) | se[v][0]);
				y[v] = (int64_t)(((uint64_t)se[v][3] << 32) | se[v][2]);
				// residual = (r - off) mod 2^29, sign extended; carried
				// as sext(residual) << 31: high word = bits 28..1 of the
				// difference sign-extended (one v_bfe_i32), low word =
				// bit 0 in bit 31
				const uint32_t d = pb[v] - se[v][0];
				const int32_t hi = __builtin_amdgcn_sbfe((int32_t)d, 1, 28);
				p[v] = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (d << 31));
			} else {
				x[v] = (int64_t)se[v][0];
				y[v] = (int64_t)se[v][1];
				// residual = (r - off) mod 2^30, sign extended
				p[v] = (int64_t)__builtin_amdgcn_sbfe(
					(int32_t)(pb[v] - se[v][2]), 0, 30);
			}
		}
		}

		if constexpr (C::lj == 0) {
			if constexpr (!kTails)
				RotChain<C, NLIVE, 0, M, DYN>::run(x, y, p, kp);
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				rx[v] = round_to_ow<T>((T)x[v], kp);
				ry[v] = round_to_ow<T>((T)y[v], kp);
			}
		} else {
			constexpr int LJ = C::lj;
			static_assert(M >= C::ngen, "seed must cover the general stages");
			if constexpr (!kTails)
				RotChainLJ<LJ, NLIVE, M, DYN>::run(x, y, p, kp, ljc);
			if (kp.r_lj == 32) {
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					rx[v] = round_to_ow_lj32<LJ>(x[v], kp);
					ry[v] = round_to_ow_lj32<LJ>(y[v], kp);
				}
			} else if (kp.r_lj > 32 && kp.r < 31) {
				const uint32_t sh = (uint32_t)kp.r_lj - 32u;
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					rx[v] = round_to_ow_lj_hi<LJ>(x[v], kp, sh);
					ry[v] = round_to_ow_lj_hi<LJ>(y[v], kp, sh);
				}
			} else {
#pragma unroll
				for (int v = 0; v < kVec; v++) {
					rx[v] = round_to_ow_lj<LJ>(x[v], kp);
					ry[v] = round_to_ow_lj<LJ>(y[v], kp);
				}
			}
		}
		apply_unit_gain<UG>(rx, kp);
		apply_unit_gain<UG>(ry, kp);
	};

	// (A/B, profiles/r05/ab_desc_loop.txt: this loop in EVERY static instance
	// would let job sets run them -- +8 % / +22 % on batches of 16- / 24-stage
	// cores -- and costs single jobs 6-7 % where the VALU binds (cfg4 385 ->
	// 362, cfg5 582 -> 541 Gsample/s).  So single jobs keep their instances,
	// and the stage counts BASELINE names (16, 24) get a second static
	// instance, DESC, that job sets run: cordic_internal.h desc_static.)
#ifdef CORDIC_DESC_LOOP_ALL
	constexpr bool kDescLoop = true;	// A/B: every instance walks descriptors
#else
	constexpr bool kDescLoop = DYN || DESC;
#endif
	if constexpr (kDescLoop) {
	if (sa.queue != nullptr) {
		// Dynamic-exit instances: the queue of the static instances (below:
		// read that comment first) one slot deeper and over TILE DESCRIPTORS
		// -- where the tile's phases come from, where its results go, how
		// many of its vectors exist.  For one job they are arithmetic on the
		// kernel's arguments; for a BATCH (cordic_jobset: round 5, many small
		// jobs in one launch, the fixed cost of a launch paid once) they are
		// read from the job set's table, one 32-byte entry per tile in queue
		// order.  The entry of the tile AFTER next is requested at the top of a
		// pass and first looked at a pass later, so its latency is never
		// waited for: hence the fourth slot and the third static ticket.
		struct Desc { uint64_t in, ox, oy; uint32_t live; };
		constexpr int kRing = 4;
		typedef volatile __attribute__((address_space(3))) uint32_t lds_word;
		lds_word *slot = (lds_word *)(uintptr_t)(seed_base + 4u * qstride);
		constexpr uint32_t kEnd = 0xffffffffu;
		const bool batch = sa.tiles != nullptr;
		const uint32_t nt = batch ? sa.ntiles : ntiles;
		const uint32_t per_d = (nt + kQueueCounters - 1) / kQueueCounters;
		uint32_t home = 0, tried = 0;		// lane 0 of the block only
		auto range_of = [&](uint32_t &lo, uint32_t &cnt) {
			const uint32_t j = (home + tried) % kQueueCounters;
			lo = j * per_d;
			cnt = lo >= nt ? 0u : (nt - lo < per_d ? nt - lo : per_d);
			return &sa.queue[j * kQueueStride];
		};
		auto q_base = [&]() -> uint32_t {	// three static tickets per block
			return (uint32_t)(kRing - 1) * q_blocks((home + tried) % kQueueCounters);
		};
		auto draw = [&]() -> uint32_t {
			uint32_t lo, cnt;
			uint32_t *c = range_of(lo, cnt);
			return (tried < (uint32_t)kQueueCounters && cnt != 0)
				? atomicAdd(c, 1u) + q_base() : kEnd;
		};
		auto resolve = [&](uint32_t ticket) -> uint32_t {
			for (;;) {
				if (tried >= (uint32_t)kQueueCounters)
					return kEnd;
				uint32_t lo, cnt;
				uint32_t *c = range_of(lo, cnt);
				if (ticket < cnt)
					return lo + ticket;
				tried++;
				if (tried >= (uint32_t)kQueueCounters)
					return kEnd;
				c = range_of(lo, cnt);
				ticket = cnt != 0 ? atomicAdd(c, 1u) + q_base() : kEnd;
			}
		};
		auto lds_barrier = [] {
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		};
		if (threadIdx.x == 0) {
			home = q_home;
			auto head = [&](uint32_t t) -> uint32_t {
				uint32_t lo, cnt;
				range_of(lo, cnt);
				return (tried == 0 && t < cnt) ? lo + t : resolve(draw());
			};
			const uint32_t R = q_blocks(q_home);
			slot[0] = head(q_rank);
			slot[1] = head(q_rank + R);
			slot[2] = head(q_rank + 2u * R);
		}
		lds_barrier();
		// the descriptor of a tile: block-uniform, kept in SGPRs
		auto uni64 = [](uint64_t v) -> uint64_t {
			// (the builtin returns a signed int: no sign extension into the
			// high word)
			const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
			return ((uint64_t)hi << 32) | lo;
		};
		auto fetch = [&](uint32_t tile) -> Desc {
			Desc d{0, 0, 0, 0};
			if (tile == kEnd)
				return d;
			if (batch) {
				const u32x4 *t = reinterpret_cast<const u32x4 *>(sa.tiles) + 2u * (size_t)tile;
				const u32x4 a = t[0], b = t[1];
				d.in = ((uint64_t)a[1] << 32) | a[0];
				d.ox = ((uint64_t)a[3] << 32) | a[2];
				d.oy = ((uint64_t)b[1] << 32) | b[0];
				d.live = b[2];
			} else {
				const size_t v0 = (size_t)tile * kTileVecs;
				if constexpr (FEED == Feed::Nco_ConstXY)
					d.in = ((uint64_t)kp.fcw << 32)
						| (uint32_t)(kp.phase0 + (uint32_t)(kp.index0 + v0 * kVec) * kp.fcw);
				else
					d.in = (uint64_t)(uintptr_t)(phin + v0);
				d.ox = (uint64_t)(uintptr_t)(ox + v0);
				d.oy = (uint64_t)(uintptr_t)(oy + v0);
				d.live = live(tile);
			}
			return d;
		};
		auto settle = [&](const Desc &d) -> Desc {	// vector registers -> scalars
			return Desc{uni64(d.in), uni64(d.ox), uni64(d.oy),
				(uint32_t)__builtin_amdgcn_readfirstlane(d.live)};
		};
		auto load_rows = [&](const Desc &d, typename IO::uvec (&ph)[kSeedSub]) {
			if constexpr (FEED != Feed::Nco_ConstXY) {
				const typename IO::uvec *p =
					reinterpret_cast<const typename IO::uvec *>((uintptr_t)d.in);
#pragma unroll
				for (int s = 0; s < kSeedSub; s++) {
					uint32_t l = lane + (uint32_t)s * kSeedBlock;
					asm("v_min_u32 %0, %1, %2" : "=v"(l) : "v"(l), "s"(d.live - 1u));
					ph[s] = __builtin_nontemporal_load(&p[l]);
				}
			}
		};
		uint32_t cur = __builtin_amdgcn_readfirstlane(slot[0]);
		uint32_t nxt = __builtin_amdgcn_readfirstlane(slot[1]);
		int ring = 0;
		Desc dcur = settle(fetch(cur)), dnxt = settle(fetch(nxt));
		typename IO::uvec pd[kSeedSub] = {};
		if (early) {
			// (requested before the prologue: the head of this function)
#pragma unroll
			for (int s = 0; s < kSeedSub; s++)
				pd[s] = pa[s];
		} else if (cur != kEnd) {
			load_rows(dcur, pd);
		}
		while (cur != kEnd) {
			// the tile after next: its descriptor is asked for now ...
			const uint32_t nn = __builtin_amdgcn_readfirstlane(slot[(ring + 2) % kRing]);
			const Desc dnn = fetch(nn);
			// ... this tile's phases are consumed ...
			uint32_t pb[kSeedSub][kVec];
#pragma unroll
			for (int s = 0; s < kSeedSub; s++) {
				if constexpr (FEED == Feed::Nco_ConstXY) {
					const uint32_t f = (uint32_t)(dcur.in >> 32);
					const uint32_t p0 = (uint32_t)dcur.in + 0x20000000u
						+ (uint32_t)s * (uint32_t)(kSeedBlock * kVec) * f;
					pb[s][0] = p0 + lane * (uint32_t)kVec * f;
#pragma unroll
					for (int v = 1; v < kVec; v++)
						pb[s][v] = pb[s][v - 1] + f;
				} else {
					const u32x4 tph = IO::widen(pd[s]);
#pragma unroll
					for (int v = 0; v < kVec; v++)
						asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(pb[s][v])
							: "v"(tph[v]), "s"(kp.pw_shl), "v"(k45));
				}
			}
			// ... and the next tile's are prefetched into the same registers
			if (nxt != kEnd)
				load_rows(dnxt, pd);
			uint32_t ahead = 0;
			if (threadIdx.x == 0)
				ahead = draw();
#pragma unroll
			for (int s = 0; s < kSeedSub; s++) {
				if (lane + (uint32_t)s * kSeedBlock < dcur.live) {
					i32x4 rx, ry;
					if constexpr (DT && dt_always(kDtR))
						pass_pb(std::true_type{}, pb[s], rx, ry);
					else if (DT && row_is_coherent(pb[s]))
						pass_pb(std::true_type{}, pb[s], rx, ry);
					else
						pass_pb(std::false_type{}, pb[s], rx, ry);
					typename IO::ivec *o0 = reinterpret_cast<typename IO::ivec *>(
						(uintptr_t)dcur.ox) + (size_t)s * kSeedBlock;
					typename IO::ivec *o1 = reinterpret_cast<typename IO::ivec *>(
						(uintptr_t)dcur.oy) + (size_t)s * kSeedBlock;
					CORDIC_STORE_OUT(true, &o0[lane], IO::narrow(rx));
					CORDIC_STORE_OUT(true, &o1[lane], IO::narrow(ry));
				}
			}
			if (threadIdx.x == 0)
				slot[(ring + 3) % kRing] = resolve(ahead);
			lds_barrier();
			cur = nxt;
			nxt = nn;
			dcur = dnxt;
			dnxt = settle(dnn);
			ring = (ring + 1) % kRing;
		}
		if (threadIdx.x == 0)
			queue_leave(sa.queue);
		return;
	}
	}
	if (!kDescLoop && sa.queue != nullptr) {
		// Work distribution, dynamic: the persistent blocks (they keep the
		// table in LDS) pull 4096-sample tiles from a counter IN ADDRESS
		// ORDER, the way the hardware dispatcher hands out the blocks of a
		// one-tile-per-block launch.  Measured with arithmetic-free kernels
		// of this kernel's traffic (round 2's streaming-pattern probe, profiles/r02/
		// hbm_probe2.txt): a contiguous chunk per block or a grid-stride
		// streams at 0.58-0.63 of the HBM peak -- hundreds of far-apart
		// streams advancing at once -- where one-shot tiles reach 0.72-0.76
		// and this queue 0.72.  One counter per XCD: XCD j sweeps the j-th
		// eighth of the batch, and helps the others once its own is done.
		// A tile is kSeedSub rows of kSeedBlock vectors (row s = vectors
		// tile * kTileVecs + s * kSeedBlock + lane: every row is one
		// contiguous 16 KiB stretch per array): with two rows a lane rotates
		// 8 samples per rendezvous, which halves the barriers, the tickets and
		// the per-pass scalar work per sample.
		// three tile-id slots behind the seeds, addressed like them by byte
		// offset (ds_read / ds_write: a generic `volatile` pointer would
		// become flat loads with a vmcnt(0) wait each)
		typedef volatile __attribute__((address_space(3))) uint32_t lds_word;
		lds_word *slot = (lds_word *)(uintptr_t)(seed_base + 4u * qstride);
		constexpr uint32_t kEnd = 0xffffffffu;
		uint32_t home = 0, tried = 0;		// lane 0 of the block only
		auto range_of = [&](uint32_t &lo, uint32_t &cnt) {
			const uint32_t j = (home + tried) % kQueueCounters;
			lo = j * per;
			cnt = lo >= ntiles ? 0u : (ntiles - lo < per ? ntiles - lo : per);
			return &sa.queue[j * kQueueStride];
		};
		// the ticket is drawn first and looked at later (resolve), so that
		// the atomic's round trip is not waited for where it is issued
		// (tickets 0 .. 2 R_j - 1 of range j are the static head: above)
		auto q_base = [&]() -> uint32_t {
			return 2u * q_blocks((home + tried) % kQueueCounters);
		};
		auto draw = [&]() -> uint32_t {
			uint32_t lo, cnt;
			uint32_t *c = range_of(lo, cnt);
			return (tried < (uint32_t)kQueueCounters && cnt != 0)
				? atomicAdd(c, 1u) + q_base() : kEnd;
		};
		auto resolve = [&](uint32_t ticket) -> uint32_t {
			for (;;) {
				if (tried >= (uint32_t)kQueueCounters)
					return kEnd;
				uint32_t lo, cnt;
				uint32_t *c = range_of(lo, cnt);
				if (ticket < cnt)
					return lo + ticket;
				tried++;		// this XCD's share is done: help the next
				if (tried >= (uint32_t)kQueueCounters)
					return kEnd;
				c = range_of(lo, cnt);
				ticket = cnt != 0 ? atomicAdd(c, 1u) + q_base() : kEnd;
			}
		};
		// Block-wide rendezvous on LDS contents only.  __syncthreads() also
		// fences global memory: every wave would wait until HBM has
		// acknowledged its stores (s_waitcnt vmcnt(0)) once per pass.
		auto lds_barrier = [] {
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		};
		// tile ids run two passes ahead of the compute (a three-slot ring in
		// LDS), so that the phases of the next tile can be prefetched while
		// this one is being rotated
		if (threadIdx.x == 0) {
			// home = blockIdx mod 8: the XCD the dispatcher's round robin
			// puts the block on (affinity only; nothing depends on it)
			home = q_home;
			// a static ticket is a number of the HOME range: once that range
			// has nothing left for this block the next one is drawn
			auto head = [&](uint32_t t) -> uint32_t {
				uint32_t lo, cnt;
				range_of(lo, cnt);
				return (tried == 0 && t < cnt) ? lo + t : resolve(draw());
			};
			slot[0] = head(q_rank);
			slot[1] = head(q_rank + q_blocks(q_home));
		}
		lds_barrier();
		// Tile ids are block-uniform: kept in SGPRs (readfirstlane), so the
		// tile's base addresses are scalar arithmetic and the per-lane part
		// of every address is the constant 32-bit offset threadIdx.x * 16
		// (global_load / global_store with an SGPR base) -- no 64-bit VALU
		// address arithmetic in the pass.
		uint32_t cur = __builtin_amdgcn_readfirstlane(slot[0]);
		int ring = 0;
		// One pass over tile `cur` with its phases in `in`; the phases of the
		// next tile are prefetched into `pre` (which may be `in` itself: the
		// phases are widened into registers first).
		// (Storing the results one pass late, so that the compiler's
		// vmcnt(0) wait for the prefetch never meets a young store, measured
		// no gain: same-box A/B in profiles/r02/ab_delayed_stores.txt.)
		auto tile_pass = [&](typename IO::uvec (&ph)[kSeedSub]) {
			const uint32_t nxt = __builtin_amdgcn_readfirstlane(slot[(ring + 1) % 3]);
			// the phases of this tile are consumed first ...
			uint32_t pb[kSeedSub][kVec];
#pragma unroll
			for (int s = 0; s < kSeedSub; s++)
				fold_pb((size_t)cur * kTileVecs + (size_t)s * kSeedBlock, lane4f,
					IO::widen(ph[s]), pb[s]);
			// ... then the next tile's are prefetched into the same registers
			if constexpr (FEED != Feed::Nco_ConstXY) {
				if (nxt != kEnd) {
#pragma unroll
					for (int s = 0; s < kSeedSub; s++)
						ph[s] = __builtin_nontemporal_load(&phin[row_vec(nxt, s)]);
				}
			}
			// the ticket for the tile after next: drawn now (behind the
			// prefetch, so that nothing waits for it here), looked at after
			// this pass's stores -- its round trip hides behind the rotation
			uint32_t ahead = 0;
			if (threadIdx.x == 0)
				ahead = draw();
#pragma unroll
			for (int s = 0; s < kSeedSub; s++) {
				if (lane + (uint32_t)s * kSeedBlock < live(cur)) {
					const size_t base = (size_t)cur * kTileVecs
							+ (size_t)s * kSeedBlock;
					i32x4 rx, ry;
					if constexpr (DT && dt_always(kDtR))
						pass_pb(std::true_type{}, pb[s], rx, ry);
					else if (DT && row_is_coherent(pb[s]))
						pass_pb(std::true_type{}, pb[s], rx, ry);
					else
						pass_pb(std::false_type{}, pb[s], rx, ry);
					// non-temporal loads AND stores: with the address-
					// ordered queue they are worth +3 % on cfg2 together
					// (0.80 -> 0.82 of the HBM peak, either one alone +1 %;
					// same-box A/B, profiles/r02/ab_nontemporal.txt) -- under
					// the round-1 chunk walk plain stores had been the faster
					CORDIC_STORE_OUT(true, &(ox + base)[lane], IO::narrow(rx));
					CORDIC_STORE_OUT(true, &(oy + base)[lane], IO::narrow(ry));
				}
			}
			if (threadIdx.x == 0)
				slot[(ring + 2) % 3] = resolve(ahead);
			lds_barrier();
			cur = nxt;
			ring = (ring + 1) % 3;
		};
		if constexpr (FEED != Feed::Nco_ConstXY) {
			// (the first tile's phases are on their way since before the
			// prologue, unless this block's home range had no tile for it)
			if (!early && cur != kEnd) {
#pragma unroll
				for (int s = 0; s < kSeedSub; s++)
					pa[s] = __builtin_nontemporal_load(&phin[row_vec(cur, s)]);
			}
		}
		// (Alternating two register sets, so that the compiler needs no
		// copies between passes, doubles the loop body: -8 % on the 24-stage
		// core, nothing elsewhere -- profiles/r02/ab_pingpong.txt.  Consuming
		// the phases before the prefetch needs no copies and no second set.)
		while (cur != kEnd)
			tile_pass(pa);
		if (threadIdx.x == 0)
			queue_leave(sa.queue);
		return;
	}

	// Work distribution, static (no queue): every persistent block sweeps its
	// own contiguous chunk.  Round 5: in the DYNAMIC-EXIT instances only -- the
	// launcher sends a launch without a queue (CORDIC_FLAG_STATIC_CHUNKS, a
	// handle whose ring is exhausted by captured launches) there; a static
	// instance then carries one copy of its stage chains instead of two, which
	// halves the seeded units' code.
	if constexpr (!DYN) {
		__builtin_trap();	// never launched without a queue
	} else {
#ifdef CORDIC_SEED_GRIDSTRIDE
	const size_t stride = (size_t)gridDim.x * kSeedBlock;
	const size_t hi = nvec;
	size_t g = (size_t)blockIdx.x * kSeedBlock + threadIdx.x;
#else
	const size_t stride = kSeedBlock;
	size_t chunk = (nvec + gridDim.x - 1) / gridDim.x;
	chunk = (chunk + kSeedBlock - 1) / kSeedBlock * kSeedBlock;
	const size_t lo = (size_t)blockIdx.x * chunk;
	const size_t hi = (lo + chunk < nvec) ? lo + chunk : nvec;
	size_t g = lo + threadIdx.x;
#endif
	// software prefetch (see rotator_unrolled)
	typename IO::uvec nph{};
	if constexpr (FEED != Feed::Nco_ConstXY)
		if (g < hi)
			nph = CORDIC_LOAD_IN(&phin[g]);
	for (; g < hi; g += stride) {
		const u32x4 tph = IO::widen(nph);
		if constexpr (FEED != Feed::Nco_ConstXY) {
			const size_t gn = g + stride;
			if (gn < hi)
				nph = CORDIC_LOAD_IN(&phin[gn]);
		}
		i32x4 rx, ry;
		uint32_t pb[kVec];
		fold_pb(g, 0u, tph, pb);
		if constexpr (DT && dt_always(kDtR))
			pass_pb(std::true_type{}, pb, rx, ry);
		else if (DT && row_is_coherent(pb))
			pass_pb(std::true_type{}, pb, rx, ry);
		else
			pass_pb(std::false_type{}, pb, rx, ry);
		CORDIC_STORE_OUT(false, &ox[g], IO::narrow(rx));
		CORDIC_STORE_OUT(false, &oy[g], IO::narrow(ry));
	}
	}	// (DYN)
}

// ------------------------------------------------------- unrolled converter

// rtl/topolar.v:122-152 on sign masks (no compares / selects).  With
// mx = -1 where x < 0 (else 0), my likewise, ax = |e_x|, ay = |e_y| as two's
// complement negations:  x0 = ax + ay;  y0 = ay - ax where the signs agree,
// ax - ay otherwise;  p0 = {1,7,3,5} * 2^29 for {++,+-,-+,--}, i.e. bit 31 =
// (y < 0), bit 30 = (x < 0) ^ (y < 0), bit 29 = 1.
template <typename T>
__device__ __forceinline__ void fold_quadrant_masks(T ex, T ey, int32_t ix,
		int32_t iy, T &x, T &y, uint32_t &p)
{
	using U = typename std::make_unsigned<T>::type;
	const int32_t mx = ix >> 31, my = iy >> 31;
	const U ax = ((U)ex ^ (U)(T)mx) - (U)(T)mx;
	const U ay = ((U)ey ^ (U)(T)my) - (U)(T)my;
	const int32_t mxy = mx ^ my;
	x = (T)(ax + ay);
	y = (T)((((ay - ax) ^ (U)(T)mxy)) - (U)(T)mxy);
	p = ((uint32_t)my & 0x80000000u) | ((uint32_t)mxy & 0x40000000u)
		| 0x20000000u;
}

template <typename C, int NLIVE, int NGEN, bool DYN = false,
		typename IO = Io32, bool UG = false>
__global__ __launch_bounds__(kBlock) void topolar_unrolled(CoreParams kp,
		const typename IO::ivec *__restrict__ xin,
		const typename IO::ivec *__restrict__ yin,
		typename IO::ivec *__restrict__ omag,
		typename IO::uvec *__restrict__ oph, size_t nvec)
{
	using T = typename std::conditional<C::wide, int64_t, int32_t>::type;
	using U = typename std::make_unsigned<T>::type;
	using Z = typename std::conditional<C::wide, int64_t, uint32_t>::type;
	const size_t stride = (size_t)gridDim.x * kBlock;
	size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x;
	// software prefetch (see rotator_unrolled)
	typename IO::ivec nx{}, ny{};
	if (g < nvec) {
		nx = CORDIC_LOAD_IN(&xin[g]);
		ny = CORDIC_LOAD_IN(&yin[g]);
	}
	for (; g < nvec; g += stride) {
		const i32x4 tx = IO::widen(nx), ty = IO::widen(ny);
		const size_t gn = g + stride;
		if (gn < nvec) {
			nx = CORDIC_LOAD_IN(&xin[gn]);
			ny = CORDIC_LOAD_IN(&yin[gn]);
		}
		int64_t x[kVec], y[kVec], p[kVec];
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			const int32_t ix = sext32(tx[v], kp.iw);
			const int32_t iy = sext32(ty[v], kp.iw);
			const T ex = (T)((U)(T)ix << kp.in_shl);
			const T ey = (T)((U)(T)iy << kp.in_shl);
			T fx, fy;
			uint32_t fp;
			fold_quadrant_masks<T>(ex, ey, ix, iy, fx, fy, fp);
			x[v] = (int64_t)(Z)fx;
			y[v] = (int64_t)(Z)fy;
			p[v] = (int64_t)fp;
		}

		PolChain<C, NLIVE, NGEN, 0, DYN>::run(x, y, p, kp);

		i32x4 rm;
		u32x4 rp;
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			rm[v] = round_to_ow<T>((T)x[v], kp);
			rp[v] = (uint32_t)p[v] >> kp.pw_shl;	// rtl/topolar.v:269
		}
		apply_unit_gain<UG>(rm, kp);
		CORDIC_STORE_OUT(true, &omag[g], IO::narrow(rm));
		CORDIC_STORE_OUT(true, &oph[g], IO::narrow(rp));
	}
}

// ------------------------------------- converter, left-justified fast form
//
// topolar_unrolled spends 8 VALU instructions per micro-rotation (sign mask,
// |1, ^-2, two shifts, three multiply-adds).  For cores whose registers cannot
// overflow and fit 34 bits (WW <= 34, cfg.needs_wrap == 0: the folded ports
// e = i << in_shl then still fit 32 bits) this form needs 7:
// x, y are carried LEFT-justified by 30 bits in their register pairs
// (x~ = x << 30) from stage 2 on, so that (y >>> k) is an arithmetic shift of
// the HIGH word by k-2 and the multipliers +/-2^30 come straight off the sign
// bit of that word -- t~ = (yh & 2^31) | 2^30 as one v_bitop3_b32 with its
// constants in VGPRs, -t~ = t~ ^ 2^31 as one VOP2 xor -- and the phase is
// accumulated at the same scale, p~ += t~ * a_k, to be shifted back once at
// the end.
//
// What decides the speed here is the instruction COUNT: in this mix the chip
// issues every VALU instruction, full-rate opcode or not, in ~3.65 cycles per
// SIMD (tools/stage_microbench.hip, profiles/r02/stage_microbench.txt); a
// v_bitop3_b32 with an SGPR or inline-constant operand costs ~0.4 more.  The
// first version of this kernel replaced the phase multiply-add by a
// direction-bit collect (one more v_bitop3_b32 with an SGPR mask) plus LDS
// tables for the phase: same count per stage, slower per instruction, and a
// ~12-instruction lookup at the end -- measured 11 % slower per stage.
//
// rtl/topolar.v:122-152 (fold), :217-243 (stages), :251-271 (outputs): same
// values as topolar_unrolled, bit for bit.
struct PolLjRegs { uint32_t sign, p30; };	// 2^31, 2^30 in VGPRs

// scratch registers of one sample's micro-rotations (CORDIC_STAGE_YIELD: kept
// live from stage to stage so that neighbouring statements share no register
// -- hipcc pads a wait state between two asm statements that do, and one
// behind a multiply-add costs more than the ones behind the 32-bit
// instructions gain)
struct PolTmp { uint32_t t, nt, sy, sx; uint64_t cc; };

template <int K>
__device__ __forceinline__ void pol_stage_lj(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a, const PolLjRegs &c, PolTmp &m)
{
	static_assert(K >= 2, "stage 1 runs on the 32-bit values");
	constexpr int sh = (K - 2 > 31) ? 31 : K - 2;
	const uint32_t yh = (uint32_t)((uint64_t)y >> 32);
#if CORDIC_STAGE_YIELD
	// the whole micro-rotation as ONE statement, so that the wait state
	// behind each 32-bit instruction stays where it is put (see
	// CORDIC_STAGE_YIELD); the high words are read before the multiply-adds
	// write their pairs
	const uint32_t xh = (uint32_t)((uint64_t)x >> 32);
	// (the carry-out nobody reads goes to the sample's own SGPR pair: two
	// statements that both clobber VCC count as sharing a register)
	if constexpr (sh == 0) {
		// y's high word is the multiplicand as it stands; x's is copied,
		// the first multiply-add overwrites it
		asm("v_bitop3_b32 %3, %7, %9, %10 bitop3:0xec" CORDIC_YIELD "\n\t"
		    "v_mov_b32 %5, %8" CORDIC_YIELD "\n\t"
		    "v_xor_b32 %4, %3, %10" CORDIC_YIELD "\n\t"
		    "v_mad_i64_i32 %0, %6, %7, %3, %0\n\t"
		    "v_mad_i64_i32 %1, %6, %5, %4, %1\n\t"
		    "v_mad_i64_i32 %2, %6, %11, %3, %2"
		    : "+v"(x), "+v"(y), "+v"(p), "+v"(m.t), "+v"(m.nt), "+v"(m.sx), "+s"(m.cc)
		    : "v"(yh), "v"(xh), "v"(c.p30), "v"(c.sign), "s"(a));
		return;
	}
	asm("v_bitop3_b32 %3, %8, %10, %11 bitop3:0xec" CORDIC_YIELD "\n\t"
	    "v_ashrrev_i32 %5, %12, %8" CORDIC_YIELD "\n\t"
	    "v_ashrrev_i32 %6, %12, %9" CORDIC_YIELD "\n\t"
	    "v_xor_b32 %4, %3, %11" CORDIC_YIELD "\n\t"
	    "v_mad_i64_i32 %0, %7, %5, %3, %0\n\t"
	    "v_mad_i64_i32 %1, %7, %6, %4, %1\n\t"
	    "v_mad_i64_i32 %2, %7, %13, %3, %2"
	    : "+v"(x), "+v"(y), "+v"(p), "+v"(m.t), "+v"(m.nt), "+v"(m.sy), "+v"(m.sx),
	      "+s"(m.cc)
	    : "v"(yh), "v"(xh), "v"(c.p30), "v"(c.sign), "n"(sh), "s"(a));
#else
	const int32_t t = (int32_t)op_and_or(yh, c.p30, c.sign);	// +/- 2^30
	const int32_t nt = (int32_t)((uint32_t)t ^ c.sign);		// -t
	const int32_t sy = (int32_t)yh >> sh;
	const int32_t sx = (int32_t)((uint64_t)x >> 32) >> sh;
	op_mad(x, sy, t);		// x' = x + t * (y >>> k)
	op_mad(y, sx, nt);		// y' = y - t * (x >>> k)
	op_mad_s(p, a, t);		// p' = p + t * a_k
#endif
}

template <int NLIVE, int I, bool DYN> struct PolChainLJ {
	static __device__ __forceinline__ void run(int64_t (&x)[kVec],
			int64_t (&y)[kVec], int64_t (&p)[kVec], const PolLjRegs &c,
			const CoreParams &kp, PolTmp (&m)[kVec])
	{
		if constexpr (I < NLIVE) {
			if (!DYN || I < kp.nlive) {
#pragma unroll
				for (int v = 0; v < kVec; v++)
					pol_stage_lj<I + 1>(x[v], y[v], p[v], kp.angle[I], c, m[v]);
				PolChainLJ<NLIVE, I + 1, DYN>::run(x, y, p, c, kp, m);
			}
		}
	}
};

// Stage 1 on the left-justified pairs: (y >>> 1) is bits 62..31 of y~, one
// v_alignbit_b32 (the value fits 32 bits because y does).
__device__ __forceinline__ void pol_stage1_lj(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a, const PolLjRegs &c)
{
	const uint32_t yh = (uint32_t)((uint64_t)y >> 32);
	const uint32_t xh = (uint32_t)((uint64_t)x >> 32);
	const int32_t t = (int32_t)op_and_or(yh, c.p30, c.sign);
	const int32_t nt = (int32_t)((uint32_t)t ^ c.sign);
	const int32_t sy = (int32_t)__builtin_amdgcn_alignbit(yh, (uint32_t)y, 31);
	const int32_t sx = (int32_t)__builtin_amdgcn_alignbit(xh, (uint32_t)x, 31);
	op_mad(x, sy, t);
	op_mad(y, sx, nt);
	op_mad_s(p, a, t);
}

// Stage 1 of a core with WW = 33 or 34: (y >>> 1) has 32 or 33 bits and does
// not fit the multiplicand; with hi = the high word (= y >>> 2) and r = the top
// bit of the low word (= bit 1 of y),  (y >>> 1) << 30 = hi * 2^31 + r * 2^30:
// the high word twice and r once at +/-2^30 (cf. rot_stage_lj_early).
__device__ __forceinline__ void pol_stage1_lj_early(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a, const PolLjRegs &c)
{
	const int32_t yh = (int32_t)((uint64_t)y >> 32), xh = (int32_t)((uint64_t)x >> 32);
	const int32_t t = (int32_t)op_and_or((uint32_t)yh, c.p30, c.sign);
	const int32_t nt = (int32_t)((uint32_t)t ^ c.sign);
	const int32_t yr = (int32_t)((uint32_t)y >> 31), xr = (int32_t)((uint32_t)x >> 31);
	op_mad(x, yh, t);
	op_mad(y, xh, nt);
	op_mad(x, yh, t);
	op_mad(y, xh, nt);
	op_mad(x, yr, t);
	op_mad(y, xr, nt);
	op_mad_s(p, a, t);
}

// PLAIN (the static instances): the launcher has checked WW <= 32 (stage 1 as
// one v_alignbit_b32 per coordinate) and 2 <= r <= 31 (rounding at the 2^30
// scale), so the kernel carries neither alternative -- as run-time branches
// they cost a block of register copies where the paths join (16 v_mov_b64 per
// pass in profiles/isa/topolar_lj_20.s of round 2).
//
// topolar_lj_sweep: vectors g, g + stride, ... < nvec of ONE contiguous
// stretch (the whole call for topolar_lj, one tile of one job for
// topolar_lj_jobs).
template <int NLIVE, bool DYN, typename IO, bool UG, bool PLAIN>
__device__ __forceinline__ void topolar_lj_sweep(const CoreParams &kp,
		const typename IO::ivec *__restrict__ xin,
		const typename IO::ivec *__restrict__ yin,
		typename IO::ivec *__restrict__ omag,
		typename IO::uvec *__restrict__ oph, size_t nvec, size_t g,
		const size_t stride)
{
	PolLjRegs c;
	c.sign = vgpr_const(0x80000000u);
	c.p30 = vgpr_const(0x40000000u);
	// rounding at the 2^30 scale: the retained bits start at bit r-2 of the
	// high word (cores with r < 2 or r > 31 take the plain 64-bit form below)
	const uint32_t rbw = vgpr_const(kp.round_bit);	// width of the tie bit: 0 or 1
	const int up = 32 - kp.iw;			// port -> sign bit of the word
	const int down = up - kp.in_shl;		// ... and back to e = i << in_shl

	typename IO::ivec nx{}, ny{};		// software prefetch
	if (g < nvec) {
		nx = CORDIC_LOAD_IN(&xin[g]);
		ny = CORDIC_LOAD_IN(&yin[g]);
	}
	for (; g < nvec; g += stride) {
		// (Consuming the ports before the prefetch, as the seeded kernel's
		// tile loop does, changes nothing here: hipcc sinks the predicated
		// prefetch to the loop latch and keeps 8 v_mov_b64 per pass around
		// it.  Making it unpredicated -- index clamped with one v_min_u32,
		// loads in the loop's main block, copies gone -- measured SLOWER in
		// every grid-stride kernel: cfg3 -4 %, p2rxy -1.5 %, full recurrence
		// -2.7 %, profiles/r03/ab_unpredicated_prefetch.txt: with the loads
		// at the top, the vmcnt(0) wait of the next pass falls on this pass's
		// just-issued stores.)
		const i32x4 tx = IO::widen(nx), ty = IO::widen(ny);
		const size_t gn = g + stride;
		if (gn < nvec) {
			nx = CORDIC_LOAD_IN(&xin[gn]);
			ny = CORDIC_LOAD_IN(&yin[gn]);
		}
		int64_t x[kVec], y[kVec], p[kVec];
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			// e = sext(i, IW) << in_shl  (rtl/topolar.v:83-84, 122-123)
			const int32_t ex = (int32_t)((uint32_t)tx[v] << up) >> down;
			const int32_t ey = (int32_t)((uint32_t)ty[v] << up) >> down;
			// rtl/topolar.v:122-152 as four multiply-adds.  With sx, sy =
			// +1 / -1 the signs of e_x, e_y (zero counts as +, as in the
			// case arms):  x0 = |e_x| + |e_y| = e_x sx + e_y sy;  y0 =
			// |e_y| - |e_x| where the signs agree, |e_x| - |e_y| where not
			// = e_y sx - e_x sy.  The multipliers +/-2^30 come off the sign
			// bits (one v_bitop3_b32 each) and leave x0, y0 left-justified
			// where the stages want them.
			const int32_t mx = (int32_t)op_and_or((uint32_t)ex, c.p30, c.sign);
			const int32_t my = (int32_t)op_and_or((uint32_t)ey, c.p30, c.sign);
			const int32_t nmy = (int32_t)((uint32_t)my ^ c.sign);
			x[v] = op_mul(ex, mx);
			op_mad(x[v], ey, my);
			y[v] = op_mul(ey, mx);
			op_mad(y[v], ex, nmy);
			// p0 = {1,7,3,5} * 2^29 for {++,+-,-+,--}
			//    = 2^31 - 2^29 sy (2 + sx)   (mod 2^32),
			// carried like the stage angles at the 2^30 scale: the product
			// of -sy 2^30 (at hand) and 2^29 (2 + sx) = (mx ^ 2^31) >>> 1;
			// the 2^31 joins in the final add.
			const uint32_t l = ((uint32_t)mx ^ c.sign) >> 1;
			p[v] = op_mul(nmy, (int32_t)l);
		}
		if (PLAIN || down >= 2) {	// WW <= 32
#pragma unroll
			for (int v = 0; v < kVec; v++)	// rtl/topolar.v:226-243, k = 1
				pol_stage1_lj(x[v], y[v], p[v], kp.angle[0], c);
		} else {		// WW = 33, 34: (y >>> 1) needs more than 32 bits
#pragma unroll
			for (int v = 0; v < kVec; v++)
				pol_stage1_lj_early(x[v], y[v], p[v], kp.angle[0], c);
		}

		PolTmp m[kVec];
#if CORDIC_STAGE_YIELD
#pragma unroll
		for (int v = 0; v < kVec; v++)	// (registers, no values: nothing to emit)
			asm volatile("" : "=v"(m[v].t), "=v"(m[v].nt), "=v"(m[v].sy), "=v"(m[v].sx),
					"=s"(m[v].cc));
#endif
		PolChainLJ<NLIVE, 1, DYN>::run(x, y, p, c, kp, m);

		i32x4 rm;
		u32x4 rp;
		if (PLAIN || (kp.r >= 2 && kp.r <= 31)) {
			// rtl/topolar.v:251-263 at the 2^30 scale: tie bit r of x is
			// bit r-2 of the high word (r <= 31: base + tie bit <= 2^30
			// is a valid signed multiplicand)
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				const uint32_t xh = (uint32_t)((uint64_t)x[v] >> 32);
				uint32_t b;
				asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(b)
					: "v"(xh), "s"(kp.r - 2), "v"(rbw));
				op_mad_s(x[v], 0x40000000u, (int32_t)(b + (uint32_t)kp.round_base));
				rm[v] = (int32_t)((uint64_t)x[v] >> 32) >> (kp.r - 2);
			}
		} else {
#pragma unroll
			for (int v = 0; v < kVec; v++)	// x may have 34 bits
				rm[v] = round_to_ow<int64_t>(x[v] >> 30, kp);
		}
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			// p0 - 2^31 + sum of t*a_k over the stages, accumulated at 2^30
			const uint32_t acc = (uint32_t)((uint64_t)p[v] >> 30);
			rp[v] = (acc + 0x80000000u) >> kp.pw_shl;	// rtl/topolar.v:269
		}
		apply_unit_gain<UG>(rm, kp);
		CORDIC_STORE_OUT(true, &omag[g], IO::narrow(rm));
		CORDIC_STORE_OUT(true, &oph[g], IO::narrow(rp));
	}
}

template <int NLIVE, bool DYN = false, typename IO = Io32, bool UG = false,
	  bool PLAIN = false>
__global__ __launch_bounds__(kBlock) void topolar_lj(CoreParams kp,
		const typename IO::ivec *__restrict__ xin,
		const typename IO::ivec *__restrict__ yin,
		typename IO::ivec *__restrict__ omag,
		typename IO::uvec *__restrict__ oph, size_t nvec)
{
	topolar_lj_sweep<NLIVE, DYN, IO, UG, PLAIN>(kp, xin, yin, omag, oph, nvec,
		(size_t)blockIdx.x * kBlock + threadIdx.x, (size_t)gridDim.x * kBlock);
}

// Many small jobs in one launch (cordic_jobset, CORDIC_JOBS_R2P; round 6): the
// host has cut every job into tiles of at most `live` whole vectors -- never
// across a job's end -- and block b sweeps tiles b, b + gridDim.x, ... in the
// table's order, which is the jobs' own: neighbouring blocks work on
// neighbouring tiles of the same arrays.  Per tile the descriptor's five
// words arrive as scalar loads and the sweep above runs over that tile alone
// (its software prefetch starts afresh: one exposed load per 8 passes of a
// full tile, hidden by the other waves of the SIMD).
template <int NLIVE, bool DYN = false, bool PLAIN = false>
__global__ __launch_bounds__(kBlock) void topolar_lj_jobs(CoreParams kp,
		const TileDescXY *__restrict__ tiles, uint32_t ntiles)
{
	for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
		const TileDescXY d = tiles[t];
		topolar_lj_sweep<NLIVE, DYN, Io32, false, PLAIN>(kp,
			reinterpret_cast<const i32x4g *>((uintptr_t)d.in0),
			reinterpret_cast<const i32x4g *>((uintptr_t)d.in1),
			reinterpret_cast<i32x4g *>((uintptr_t)d.o0),
			reinterpret_cast<u32x4g *>((uintptr_t)d.o1),
			(size_t)d.live, (size_t)threadIdx.x, (size_t)kBlock);
	}
}

// ------------------------- converter, left-justified form for WW = 35 .. 40
//
// Same idea as topolar_lj with the justification LJ = 64 - WW (29 .. 24; one
// dynamic-exit instance per width: 32-bit I/Q at the default two extra bits is
// WW = 40):
//   * the ports are left-justified in their 32-bit words (e' = i << (32-IW));
//     with multipliers +/-2^30 the fold's four multiply-adds then deliver
//     x0 << LJ exactly (32 - IW + 30 = in_shl + LJ);
//   * the sign of y no longer reaches bit LJ+1 of the high word, so the stage
//     multiplier takes two instructions (sign mask, v_bitop3_b32) instead of
//     one: 8 per regular stage;
//   * stages with k < 32 - LJ take the early form (cf. rot_stage_lj_early):
//     (y >>> k) << LJ = hi * 2^(32-k) + r * 2^LJ, r = the top 32-LJ-k bits of
//     the low word -- 12 instructions (14 for k = 1) against ~18 for the explicit
//     64-bit form of topolar_unrolled<Wide64>.
// rtl/topolar.v:122-152, 217-243, 251-271: same values, bit for bit.
struct PolWideRegs { uint32_t mask, bit; };	// ~(2^(LJ+1) - 1), 2^LJ in VGPRs

template <int LJ, int K>
__device__ __forceinline__ void pol_stage_w(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a, const PolWideRegs &c)
{
	constexpr int first = 32 - LJ;
	const int32_t yh = (int32_t)((uint64_t)y >> 32), xh = (int32_t)((uint64_t)x >> 32);
	const uint32_t d = (uint32_t)(yh >> 31);
	const int32_t t = (int32_t)op_and_or(d, c.bit, c.mask);		// +/- 2^LJ
	const int32_t nt = (int32_t)((uint32_t)t ^ c.mask);		// -t
	if constexpr (K >= first) {
		constexpr int sh = (K - first > 31) ? 31 : K - first;
		const int32_t sy = yh >> sh, sx = xh >> sh;
		op_mad(x, sy, t);
		op_mad(y, sx, nt);
	} else {
		constexpr int dbits = first - K;
		constexpr int up = (K == 1) ? 30 - LJ : 32 - K - LJ;
		const int32_t th = (int32_t)((uint32_t)t << up);
		const int32_t nth = (int32_t)((uint32_t)nt << up);
		const int32_t yr = (int32_t)((uint32_t)y >> (32 - dbits));
		const int32_t xr = (int32_t)((uint32_t)x >> (32 - dbits));
		op_mad(x, yh, th);
		op_mad(y, xh, nth);
		if constexpr (K == 1) {
			op_mad(x, yh, th);
			op_mad(y, xh, nth);
		}
		op_mad(x, yr, t);
		op_mad(y, xr, nt);
	}
	op_mad_s(p, a, t);		// p' = p + t * a_k
}

template <int LJ, int NLIVE, int I> struct PolChainW {
	static __device__ __forceinline__ void run(int64_t (&x)[kVec],
			int64_t (&y)[kVec], int64_t (&p)[kVec], const PolWideRegs &c,
			const CoreParams &kp)
	{
		if constexpr (I < NLIVE) {
			if (I < kp.nlive) {
#pragma unroll
				for (int v = 0; v < kVec; v++)
					pol_stage_w<LJ, I + 1>(x[v], y[v], p[v], kp.angle[I], c);
				PolChainW<LJ, NLIVE, I + 1>::run(x, y, p, c, kp);
			}
		}
	}
};

template <int LJ, int NLIVE, typename IO = Io32, bool UG = false>
__global__ __launch_bounds__(kBlock) void topolar_ljw(CoreParams kp,
		const typename IO::ivec *__restrict__ xin,
		const typename IO::ivec *__restrict__ yin,
		typename IO::ivec *__restrict__ omag,
		typename IO::uvec *__restrict__ oph, size_t nvec)
{
	PolWideRegs c;
	c.bit = vgpr_const(1u << LJ);
	c.mask = vgpr_const(~((2u << LJ) - 1u));
	const uint32_t sign = vgpr_const(0x80000000u), p30 = vgpr_const(0x40000000u);
	const uint32_t rbw = vgpr_const(kp.round_bit);
	const int up = 32 - kp.iw;
	// the rounded magnitude is bits r+LJ .. of x~: in the high word if
	// r + LJ >= 32, with the increment (base + tie) a signed multiplicand
	const bool round_hi = kp.r + LJ >= 32 && kp.r <= 31;

	const size_t stride = (size_t)gridDim.x * kBlock;
	size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x;
	typename IO::ivec nx{}, ny{};		// software prefetch
	if (g < nvec) {
		nx = CORDIC_LOAD_IN(&xin[g]);
		ny = CORDIC_LOAD_IN(&yin[g]);
	}
	for (; g < nvec; g += stride) {
		const i32x4 tx = IO::widen(nx), ty = IO::widen(ny);
		const size_t gn = g + stride;
		if (gn < nvec) {
			nx = CORDIC_LOAD_IN(&xin[gn]);
			ny = CORDIC_LOAD_IN(&yin[gn]);
		}
		int64_t x[kVec], y[kVec], p[kVec];
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			const int32_t ex = (int32_t)((uint32_t)tx[v] << up);
			const int32_t ey = (int32_t)((uint32_t)ty[v] << up);
			// fold and quadrant phase exactly as in topolar_lj, the phase
			// at the 2^LJ scale of the stages
			const int32_t mx = (int32_t)op_and_or((uint32_t)ex, p30, sign);
			const int32_t my = (int32_t)op_and_or((uint32_t)ey, p30, sign);
			const int32_t nmy = (int32_t)((uint32_t)my ^ sign);
			x[v] = op_mul(ex, mx);
			op_mad(x[v], ey, my);
			y[v] = op_mul(ey, mx);
			op_mad(y[v], ex, nmy);
			const uint32_t l = ((uint32_t)mx ^ sign) >> 1;	// 2^29 (2 + sx)
			p[v] = op_mul(nmy >> (30 - LJ), (int32_t)l);	// -sy 2^LJ
		}

		PolChainW<LJ, NLIVE, 0>::run(x, y, p, c, kp);

		i32x4 rm;
		u32x4 rp;
		if (round_hi) {
			const int sh = kp.r + LJ - 32;
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				const uint32_t xh = (uint32_t)((uint64_t)x[v] >> 32);
				uint32_t b;
				asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(b)
					: "v"(xh), "s"(sh), "v"(rbw));
				op_mad_s(x[v], 1u << LJ, (int32_t)(b + (uint32_t)kp.round_base));
				rm[v] = (int32_t)((uint64_t)x[v] >> 32) >> sh;
			}
		} else {
#pragma unroll
			for (int v = 0; v < kVec; v++)
				rm[v] = round_to_ow<int64_t>(x[v] >> LJ, kp);
		}
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			const uint32_t acc = (uint32_t)((uint64_t)p[v] >> LJ);
			rp[v] = (acc + 0x80000000u) >> kp.pw_shl;	// rtl/topolar.v:269
		}
		apply_unit_gain<UG>(rm, kp);
		CORDIC_STORE_OUT(true, &omag[g], IO::narrow(rm));
		CORDIC_STORE_OUT(true, &oph[g], IO::narrow(rp));
	}
}

} // namespace dev
} // namespace cordic_amd
#endif
