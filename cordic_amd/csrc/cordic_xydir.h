// cordic_xydir.h -- rotator for PER-SAMPLE vectors with looked-up directions
// (round 4; cordic_plan_p2r).
//
// rtl/cordic.v:262-280: the direction of every micro-rotation is the sign of
// the residual PHASE; i_xval / i_yval never enter it.  So also when the vector
// changes from sample to sample the directions of all stages are a step
// function of the folded phase with exact integer break points, and the
// multipliers {-s, s} 2^LJ of a stage can be READ (4 VALU instructions per
// stage: two shifts, two v_mad_i64_i32) instead of derived from a phase
// recurrence (7: + the phase multiply-add and two v_bitop3).  What cannot be
// tabulated is the state itself -- there are no seeds here, every stage runs.
//
//   * stage 1 comes out of the octant fold's four multiply-adds, as in
//     rotator_unrolled (cordic_device.h: fold1);
//   * stages 2 .. kDtLastStage: groups of at most 5 stages (<= 32 leaves per
//     group: wins on unrelated phases too, so there is no per-row choice), one
//     bucket read + one entry read per group and sample, tables built by the
//     block's prologue from the plan's {pattern, offset} words
//     (cordic_plan.cpp: build_dir_table);
//   * stages behind that (29-stage cores): phase recurrence on the residual.
//
// Same arithmetic, same results, bit for bit (tests/test_gpu_parity.py:
// test_plan_p2r_*).
#ifndef CORDIC_XYDIR_H
#define CORDIC_XYDIR_H

#include "cordic_device.h"

namespace cordic_amd {
namespace dev {

struct DirArgs {
	const uint32_t *table;	// device copy of build_dir_table()'s words
	DxInfo	dx;
};

// LDS: the fold's 8 rows (16 B each) first, then per group its buckets
// (aligned to their total size) and 16-byte aligned leaf entries
constexpr uint32_t kDxLdsFold = 8u * 16u;
__host__ __device__ inline uint32_t dx_lds_layout(const DxInfo &dx,
		uint32_t *bucket_base, uint32_t *leaf_base)
{
	uint32_t at = kDxLdsFold;
	for (int g = 0; g < dx.n; g++) {
		const uint32_t bb = (uint32_t)dx.lv[g].nb * 8u;
		at = (at + bb - 1u) & ~(bb - 1u);
		if (bucket_base) bucket_base[g] = at;
		at = (at + bb + 15u) & ~15u;
		if (leaf_base) leaf_base[g] = at;
		at += (uint32_t)dx.lv[g].nl * (uint32_t)dt_entry_dwords(dx.lv[g].t) * 4u;
	}
	return at;
}

// An early stage (k < 32-LJ: the shifted operand reaches into the low word,
// cordic_device.h: rot_stage_lj_early) with its multipliers handed in.
template <int LJ, int K>
__device__ __forceinline__ void rot_stage_lj_early_dir(int64_t &x, int64_t &y,
		int32_t ns, int32_t s)
{
	static_assert(K >= 2 && K < LjConst<LJ>::first, "stage 1 is the fold's");
	constexpr int D = LjConst<LJ>::first - K;
	constexpr int up = 32 - K - LJ;		// +/- 2^(32-k) = (+/- 2^LJ) << up
	const int32_t sh = (int32_t)((uint32_t)s << up);
	const int32_t nsh = (int32_t)((uint32_t)ns << up);
	const int32_t yh = (int32_t)((uint64_t)y >> 32), xh = (int32_t)((uint64_t)x >> 32);
	const int32_t yr = (int32_t)((uint32_t)y >> (32 - D));
	const int32_t xr = (int32_t)((uint32_t)x >> (32 - D));
	op_mad(x, yh, nsh);
	op_mad(y, xh, sh);
	op_mad(x, yr, ns);
	op_mad(y, xr, s);
}

// JOBS (round 6; cordic_jobset, CORDIC_JOBS_P2R_XY / CORDIC_JOBS_MIX): the
// arrays are not one stretch but `ntiles` tiles of many jobs, described by the
// host (cordic_internal.h: TileDescXY); block b stages its tables ONCE and
// then sweeps tiles b, b + gridDim.x, ... -- the per-call arguments xin .. nvec
// are unused, and a mixer tile's accumulator starts from the descriptor's
// {fcw, phase} pair instead of kp.phase0 / fcw / index0.
template <int LJ, int NLIVE, bool JOBS = false>
__global__ __launch_bounds__(kBlock) void rotator_xydir(CoreParams kp, DirArgs da,
		const i32x4g *__restrict__ xin_, const i32x4g *__restrict__ yin_,
		const u32x4g *__restrict__ phin_, i32x4g *__restrict__ ox_,
		i32x4g *__restrict__ oy_, size_t nvec_,
		const TileDescXY *__restrict__ tiles, uint32_t ntiles)
{
	static_assert(LJ == 29 || LJ == 30, "WW <= 35 cores");
	constexpr int kN = dx_levels(NLIVE);
	static_assert(kN >= 1 && kN <= kDxMaxLevels, "no group to look up");
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

	uint32_t bk_base[kDxMaxLevels] = {}, lf_base[kDxMaxLevels] = {};
	dx_lds_layout(da.dx, bk_base, lf_base);

	// ---- prologue: the fold's rows and the groups' tables
	if (threadIdx.x < 8) {
		// q = 0: (x, y); 1: (-y, x); 2: (-x, -y); 3: (y, -x); with the first
		// micro-rotation folded in (cordic_device.h: rotator_unrolled, fold1)
		const int q = threadIdx.x >> 1;
		const int32_t dir = (threadIdx.x & 1) ? 1 : -1;	// phase >= 0 : < 0
		const int32_t k = (int32_t)(1u << (kp.in_shl & 31));
		const int32_t c = (q == 0) ? k : (q == 2) ? -k : 0;
		const int32_t sn = (q == 1) ? k : (q == 3) ? -k : 0;
		int32_t *row = reinterpret_cast<int32_t *>(lds) + threadIdx.x * 4;
		const int32_t a = c - dir * (sn / 2), b = sn + dir * (c / 2);
		row[0] = a;
		row[1] = b;
		row[2] = -b;
		// u_0 = p_1 + bias0 = (pb & 0x3fffffff) - 2^29 -/+ a_0 + bias0
		row[3] = (int32_t)(da.dx.bias0 - 0x20000000u
				- (uint32_t)(dir * (int32_t)kp.angle[0]));
	}
#pragma unroll
	for (int g = 0; g < kN; g++) {
		const DtLevel lv = da.dx.lv[g];
		const uint32_t *src = da.table + lv.word;
		uint32_t *bk = lds + bk_base[g] / 4u;
		const uint32_t stride = (uint32_t)dt_entry_dwords(lv.t) * 4u;
		for (int i = threadIdx.x; i < lv.nb * 2; i += kBlock) {
			const uint32_t w = src[i];
			bk[i] = (i & 1) ? lf_base[g] + w * stride : w;
		}
		const uint32_t *lsrc = src + (size_t)lv.nb * 2;
		uint32_t *lf = lds + lf_base[g] / 4u;
		for (int e = threadIdx.x; e < lv.nl; e += kBlock) {
			const uint32_t pat = lsrc[2 * e];
			uint32_t *d = lf + (size_t)e * dt_entry_dwords(lv.t);
			for (int jj = 0; jj < lv.t; jj++) {
				// bit set: residual >= 0 at that stage, s = +1
				const bool pos = (pat >> (lv.t - 1 - jj)) & 1u;
				const uint32_t plus = LjConst<LJ>::bit;
				const uint32_t minus = LjConst<LJ>::mask | LjConst<LJ>::bit;
				d[2 * jj + 0] = pos ? minus : plus;	// -s 2^LJ (x)
				d[2 * jj + 1] = pos ? plus : minus;	//  s 2^LJ (y)
			}
			d[2 * lv.t] = lsrc[2 * e + 1];		// u_next = u - this
		}
	}
	__syncthreads();

	typedef const __attribute__((address_space(3))) u32x4 lds_entry;
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
	typedef const __attribute__((address_space(3))) u32x2 lds_bucket;
	// LDS is addressed by byte offset from 0: no static LDS in this kernel
	if ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)lds != 0u)
		__builtin_trap();

	LjRegs ljc{};
	ljc.mask = vgpr_const(LjConst<LJ>::mask);
	ljc.bit = vgpr_const(LjConst<LJ>::bit);
	ljc.maskbit = vgpr_const(LjConst<LJ>::mask | LjConst<LJ>::bit);
	uint32_t maskv[kDxMaxLevels] = {};
#pragma unroll
	for (int g = 0; g < kN; g++)
		maskv[g] = vgpr_const(((uint32_t)da.dx.lv[g].nb - 1u) << 3);
	const uint32_t k45 = vgpr_const(0x20000000u);
	const bool full_ports = kp.iw == 32;	// wave-uniform: no sign extension

	// the mixer (cordic_plan_mix, round 5): the phases are not read but
	// generated, phase0 + (index0 + i) * fcw -- wave-uniform choice
	const bool gen_phase = kp.xy_nco != 0;
	// one contiguous stretch: vectors g, g + stride, ... < nvec; a mixer's
	// phase of vector g is acc0 + 4 g fcw (+ v fcw for its samples)
	auto sweep = [&](const i32x4g *__restrict__ xin, const i32x4g *__restrict__ yin,
			const u32x4g *__restrict__ phin, i32x4g *__restrict__ ox,
			i32x4g *__restrict__ oy, const size_t nvec, size_t g,
			const size_t stride, const uint32_t acc0, const uint32_t fcw) {
	// software prefetch, as in rotator_unrolled
	u32x4 nph{};
	i32x4 nx{}, ny{};
	if (g < nvec) {
		if (!gen_phase)
			nph = CORDIC_LOAD_IN(&phin[g]);
		nx = CORDIC_LOAD_IN(&xin[g]);
		ny = CORDIC_LOAD_IN(&yin[g]);
	}
	for (; g < nvec; g += stride) {
		const u32x4 tph = nph;
		i32x4 tx = nx, ty = ny;
		const size_t gn = g + stride;
		if (gn < nvec) {
			if (!gen_phase)
				nph = CORDIC_LOAD_IN(&phin[gn]);
			nx = CORDIC_LOAD_IN(&xin[gn]);
			ny = CORDIC_LOAD_IN(&yin[gn]);
		}
		if (!full_ports) {
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				tx[v] = sext32(tx[v], kp.iw);
				ty[v] = sext32(ty[v], kp.iw);
			}
		}
		uint32_t pb[kVec];
		if (gen_phase) {
			pb[0] = (acc0 + 0x20000000u) + (uint32_t)(g * kVec) * fcw;
#pragma unroll
			for (int v = 1; v < kVec; v++)
				pb[v] = pb[v - 1] + fcw;
		} else {
#pragma unroll
			for (int v = 0; v < kVec; v++)
				asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(pb[v])
					: "v"(tph[v]), "s"(kp.pw_shl), "v"(k45));
		}

		int64_t x[kVec], y[kVec];
		uint32_t u[kVec];
		u32x4 m[kVec];
#pragma unroll
		for (int v = 0; v < kVec; v++)
			// row = quadrant (bits 31..30) x direction (bit 29 of pb set
			// <=> folded phase >= 0), 16 bytes each
			m[v] = *(lds_entry *)(uintptr_t)((pb[v] >> 25) & 0x70u);
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			int64_t fx = op_mul(ty[v], (int32_t)m[v][2]);	// -B * i_y
			op_mad(fx, tx[v], (int32_t)m[v][0]);		// + A * i_x
			int64_t fy = op_mul(ty[v], (int32_t)m[v][0]);	//  A * i_y
			op_mad(fy, tx[v], (int32_t)m[v][1]);		// + B * i_x
			x[v] = (int64_t)((uint64_t)fx << LJ);
			y[v] = (int64_t)((uint64_t)fy << LJ);
			u[v] = (pb[v] & 0x3fffffffu) + m[v][3];
		}

		auto group = [&](auto G_) {
			constexpr int G = decltype(G_)::value;
			constexpr int T = dx_size(NLIVE, G);
			constexpr int K0 = dx_first(NLIVE, G);		// stages done
			constexpr bool more = (G + 1 < kN) || dx_rest(NLIVE) > 0;
			const uint32_t sh3 = (uint32_t)da.dx.lv[G].shift - 3u;
			u32x2 b2[kVec];
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				uint32_t a;
				asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a)
					: "v"(u[v] >> sh3), "v"(maskv[G]), "s"(bk_base[G]));
				b2[v] = *(lds_bucket *)(uintptr_t)a;
			}
			constexpr int W = 2 * T + (more ? 1 : 0);	// dwords used
			constexpr int kStride = dt_entry_dwords(T) * 4;
			// the prologue writes 2 T + 1 dwords per entry whatever
			// CORDIC_DT_SINGLES says about the seeded kernel's entries
			static_assert(dt_entry_dwords(T) >= 2 * T + 1, "entry stride");
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
				else
					asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ea[v])
						: "v"(c), "s"(kStride), "v"(b2[v][1]));
			}
			uint32_t en[kVec][12];
#pragma unroll
			for (int v = 0; v < kVec; v++) {
#pragma unroll
				for (int at = 0; at < W; at += 4) {
					if (W - at >= 4) {
						const u32x4 t4 = *(lds_entry *)(uintptr_t)(ea[v] + 4u * at);
						en[v][at] = t4[0]; en[v][at + 1] = t4[1];
						en[v][at + 2] = t4[2]; en[v][at + 3] = t4[3];
					} else if (W - at == 3) {
						// three dwords left: ds_read_b96 (8 lane groups of 8)
						// or a b128 whose fourth dword is padding
#ifndef CORDIC_DX_B128	/* same-box A/B (profiles/r04/ab_b96.txt): a b128 here is
			 * -0.8 % on ramps, +0.5 % on unrelated phases: b96 kept */
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
					constexpr int K = K0 + J + 1;	// this stage's shift
#pragma unroll
					for (int v = 0; v < kVec; v++) {
						const int32_t ns_ = (int32_t)en[v][2 * J];
						const int32_t s_ = (int32_t)en[v][2 * J + 1];
						if constexpr (K < LjConst<LJ>::first)
							rot_stage_lj_early_dir<LJ, K>(x[v], y[v], ns_, s_);
						else
							rot_stage_lj_dir<LJ, K>(x[v], y[v], ns_, s_);
					}
				}
			};
			stage(std::integral_constant<int, 0>{});
			stage(std::integral_constant<int, 1>{});
			stage(std::integral_constant<int, 2>{});
			stage(std::integral_constant<int, 3>{});
			stage(std::integral_constant<int, 4>{});
			if constexpr (more) {
#pragma unroll
				for (int v = 0; v < kVec; v++)
					u[v] -= en[v][2 * T];
			}
		};
		if constexpr (kN > 0) group(std::integral_constant<int, 0>{});
		if constexpr (kN > 1) group(std::integral_constant<int, 1>{});
		if constexpr (kN > 2) group(std::integral_constant<int, 2>{});
		if constexpr (kN > 3) group(std::integral_constant<int, 3>{});
		if constexpr (kN > 4) group(std::integral_constant<int, 4>{});
		constexpr int kRest = dx_rest(NLIVE);
		if constexpr (kRest > 0) {
			// the last stages on the residual phase itself
			int64_t p[kVec];
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				const int32_t r = (int32_t)(u[v] - da.dx.bias_last);
				p[v] = (int64_t)(((uint64_t)(uint32_t)(r >> 1) << 32)
						| ((uint32_t)r << 31));
			}
			RotChainLJ<LJ, NLIVE, NLIVE - kRest, false>::run(x, y, p, kp, ljc);
		}

		i32x4 rx, ry;
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
		CORDIC_STORE_OUT(true, &ox[g], rx);
		CORDIC_STORE_OUT(true, &oy[g], ry);
	}
	};	// sweep
	if constexpr (!JOBS) {
		sweep(xin_, yin_, phin_, ox_, oy_, nvec_,
			(size_t)blockIdx.x * kBlock + threadIdx.x, (size_t)gridDim.x * kBlock,
			kp.phase0 + (uint32_t)kp.index0 * kp.fcw, kp.fcw);
	} else {
		for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
			const TileDescXY d = tiles[t];
			sweep(reinterpret_cast<const i32x4g *>((uintptr_t)d.in0),
				reinterpret_cast<const i32x4g *>((uintptr_t)d.in1),
				reinterpret_cast<const u32x4g *>((uintptr_t)d.in2),
				reinterpret_cast<i32x4g *>((uintptr_t)d.o0),
				reinterpret_cast<i32x4g *>((uintptr_t)d.o1),
				(size_t)d.live, (size_t)threadIdx.x, (size_t)kBlock,
				(uint32_t)d.in2, (uint32_t)(d.in2 >> 32));
		}
	}
}

} // namespace dev
} // namespace cordic_amd
#endif
