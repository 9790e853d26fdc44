// cordic_kernels.hip -- gfx950 (CDNA4) kernels of the CORDIC rotation engine.
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
namespace {

constexpr int kBlock = 256;		// 4 waves: one per SIMD
constexpr int kVec = 4;			// samples per lane per pass (16 B)
constexpr int kTile = kBlock * kVec;	// samples per block per pass

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
	int32_t	x0, y0;		// constant-vector feeds (sign extended)
	uint32_t phase0, fcw;	// NCO, left-justified
	uint64_t index0;	// NCO: global index of sample 0
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

// Single-instruction wrappers.  hipcc's instcombine rewrites the
// conditional-negate identities below into longer add3/xor sequences; making
// the direction masks opaque and naming v_xad_u32 keeps a WW<=32 rotation at
// ten VALU operations.  Each statement is one VALU instruction with VGPR/SGPR
// operands only, so no wait states are owed inside or around it.
__device__ __forceinline__ uint32_t op_sign_mask(uint32_t v)
{
	uint32_t d;
	asm("v_ashrrev_i32 %0, 31, %1" : "=v"(d) : "v"(v));
	return d;
}
__device__ __forceinline__ uint32_t op_not(uint32_t v)
{
	uint32_t d;
	asm("v_not_b32 %0, %1" : "=v"(d) : "v"(v));
	return d;
}
// (a ^ b) + c
__device__ __forceinline__ uint32_t op_xad(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t d;
	asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}
__device__ __forceinline__ uint32_t op_xad_s(uint32_t a_sgpr, uint32_t b,
		uint32_t c)
{
	uint32_t d;
	asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "s"(a_sgpr), "v"(b), "v"(c));
	return d;
}

template <int K> struct ShiftOf {
	static constexpr int s32 = (K > 31) ? 31 : K;
	static constexpr int s64 = (K > 63) ? 63 : K;
};

// ------------------------------------------------------- rotator: p2r stage

// rtl/cordic.v:262-280.  d = -1 where the residual phase is negative.
//   phase <  0: x' = x + (y>>>k), y' = y - (x>>>k), p' = p + a
//   phase >= 0: x' = x - (y>>>k), y' = y + (x>>>k), p' = p - a
// With nd = ~d:  x' = ((y>>>k) ^ nd) + (x - nd)
//                y' = ((x>>>k) ^ d ) + (y - d )
//                p' = ( a      ^ nd) + (p - nd)
template <int K>
__device__ __forceinline__ void rot_stage(int32_t &x, int32_t &y, uint32_t &p,
		uint32_t a)
{
	const uint32_t d = op_sign_mask(p);
	const uint32_t nd = op_not(d);
	const uint32_t sy = (uint32_t)(y >> ShiftOf<K>::s32);
	const uint32_t sx = (uint32_t)(x >> ShiftOf<K>::s32);
	const uint32_t xt = (uint32_t)x - nd;
	const uint32_t yt = (uint32_t)y - d;
	const uint32_t pt = p - nd;
	x = (int32_t)op_xad(sy, nd, xt);
	y = (int32_t)op_xad(sx, d, yt);
	p = op_xad_s(a, nd, pt);
}

template <int K>
__device__ __forceinline__ void rot_stage(int64_t &x, int64_t &y, uint32_t &p,
		uint32_t a)
{
	const int32_t d = (int32_t)p >> 31;
	const int64_t d64 = (int64_t)d;
	const uint32_t neg = p >> 31;
	const int64_t sy = (y >> ShiftOf<K>::s64) ^ d64;
	const int64_t sx = (x >> ShiftOf<K>::s64) ^ d64;
	x = x - sy - (int64_t)neg;
	y = y + sx + (int64_t)neg;
	p = p - (a ^ (uint32_t)d) - neg;
}

// ---------------------------------------------------- converter: r2p stage

// rtl/topolar.v:226-243.  d = -1 where y is negative (below the axis).
//   y <  0: x' = x - (y>>>k), y' = y + (x>>>k), p' = p - a
//   y >= 0: x' = x + (y>>>k), y' = y - (x>>>k), p' = p + a
// With nd = ~d:  x' = ((y>>>k) ^ d ) + (x - d )
//                y' = ((x>>>k) ^ nd) + (y - nd)
//                p' = ( a      ^ d ) + (p - d )
template <int K>
__device__ __forceinline__ void pol_stage(int32_t &x, int32_t &y, uint32_t &p,
		uint32_t a)
{
	const uint32_t d = op_sign_mask((uint32_t)y);
	const uint32_t nd = op_not(d);
	const uint32_t sy = (uint32_t)(y >> ShiftOf<K>::s32);
	const uint32_t sx = (uint32_t)(x >> ShiftOf<K>::s32);
	const uint32_t xt = (uint32_t)x - d;
	const uint32_t yt = (uint32_t)y - nd;
	const uint32_t pt = p - d;
	x = (int32_t)op_xad(sy, d, xt);
	y = (int32_t)op_xad(sx, nd, yt);
	p = op_xad_s(a, d, pt);
}

template <int K>
__device__ __forceinline__ void pol_stage(int64_t &x, int64_t &y, uint32_t &p,
		uint32_t a)
{
	const int64_t d64 = y >> 63;
	const uint32_t d = (uint32_t)d64;
	const uint32_t neg = d & 1u;
	const int64_t sy = (y >> ShiftOf<K>::s64) ^ d64;
	const int64_t sx = (x >> ShiftOf<K>::s64) ^ d64;
	x = x + sy + (int64_t)neg;
	y = y - sx - (int64_t)neg;
	p = p + (a ^ d) + neg;
}

// Compile-time unrolled stage chain over the kVec samples of a lane: stage
// i of all samples before stage i+1, so the four dependency chains interleave.
template <typename T, int NLIVE, int I = 0> struct RotChain {
	static __device__ __forceinline__ void run(T (&x)[kVec], T (&y)[kVec],
			uint32_t (&p)[kVec], const CoreParams &kp)
	{
		if constexpr (I < NLIVE) {
#pragma unroll
			for (int v = 0; v < kVec; v++)
				rot_stage<I + 1>(x[v], y[v], p[v], kp.angle[I]);
			RotChain<T, NLIVE, I + 1>::run(x, y, p, kp);
		}
	}
};
template <typename T, int NLIVE, int I = 0> struct PolChain {
	static __device__ __forceinline__ void run(T (&x)[kVec], T (&y)[kVec],
			uint32_t (&p)[kVec], const CoreParams &kp)
	{
		if constexpr (I < NLIVE) {
#pragma unroll
			for (int v = 0; v < kVec; v++)
				pol_stage<I + 1>(x[v], y[v], p[v], kp.angle[I]);
			PolChain<T, NLIVE, I + 1>::run(x, y, p, kp);
		}
	}
};

// ------------------------------------------------------------- pre / post

// rtl/cordic.v:131-188 on a left-justified phase: q = quadrant of
// (phase + 45 deg); rotate the vector by q * 90 deg, remove q * 2^(PW-2).
template <typename T>
__device__ __forceinline__ void fold_octant(T ex, T ey, uint32_t P, T &x, T &y,
		uint32_t &p)
{
	using U = typename std::make_unsigned<T>::type;
	const uint32_t q = (P + 0x20000000u) >> 30;
	p = P - (q << 30);
	const bool swap = (q & 1u) != 0;
	const T a = swap ? ey : ex;
	const T b = swap ? ex : ey;
	const bool negx = (q == 1u) || (q == 2u);
	const bool negy = (q >= 2u);
	x = negx ? (T)((U)0 - (U)a) : a;
	y = negy ? (T)((U)0 - (U)b) : b;
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

// ------------------------------------------------------- memory accessors

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

// --------------------------------------------------------- unrolled rotator

// Processes whole 4-sample groups only (nvec of them); the launcher sends the
// 0..3 trailing samples to the generic kernel.  Keeping the tail out of this
// kernel is what lets hipcc emit global_load_dwordx4 / global_store_dwordx4.
template <typename T, int NLIVE, Feed FEED>
__global__ __launch_bounds__(kBlock) void rotator_unrolled(CoreParams kp,
		const i32x4 *__restrict__ xin, const i32x4 *__restrict__ yin,
		const u32x4 *__restrict__ phin, i32x4 *__restrict__ ox,
		i32x4 *__restrict__ oy, size_t nvec)
{
	using U = typename std::make_unsigned<T>::type;
	for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < nvec;
			g += (size_t)gridDim.x * kBlock) {
		uint32_t P[kVec];
		int32_t ix[kVec], iy[kVec];
		if constexpr (FEED == Feed::Nco_ConstXY) {
			const uint32_t s0 = (uint32_t)(kp.index0 + g * kVec);
			P[0] = kp.phase0 + s0 * kp.fcw;
#pragma unroll
			for (int v = 1; v < kVec; v++)
				P[v] = P[v - 1] + kp.fcw;
		} else {
			const u32x4 t = phin[g];
#pragma unroll
			for (int v = 0; v < kVec; v++)
				P[v] = t[v] << kp.pw_shl;
		}
		if constexpr (FEED == Feed::PhaseArray_XYArray) {
			const i32x4 tx = xin[g];
			const i32x4 ty = yin[g];
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				ix[v] = sext32(tx[v], kp.iw);
				iy[v] = sext32(ty[v], kp.iw);
			}
		} else {
#pragma unroll
			for (int v = 0; v < kVec; v++) {
				ix[v] = kp.x0;
				iy[v] = kp.y0;
			}
		}

		T x[kVec], y[kVec];
		uint32_t p[kVec];
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			const T ex = (T)((U)(T)ix[v] << kp.in_shl);
			const T ey = (T)((U)(T)iy[v] << kp.in_shl);
			fold_octant<T>(ex, ey, P[v], x[v], y[v], p[v]);
		}

		RotChain<T, NLIVE>::run(x, y, p, kp);

		i32x4 rx, ry;
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			rx[v] = round_to_ow<T>(x[v], kp);
			ry[v] = round_to_ow<T>(y[v], kp);
		}
		// outputs are written once and never re-read here: stream them
		__builtin_nontemporal_store(rx, &ox[g]);
		__builtin_nontemporal_store(ry, &oy[g]);
	}
}

// ------------------------------------------------------- unrolled converter

template <typename T, int NLIVE>
__global__ __launch_bounds__(kBlock) void topolar_unrolled(CoreParams kp,
		const i32x4 *__restrict__ xin, const i32x4 *__restrict__ yin,
		i32x4 *__restrict__ omag, u32x4 *__restrict__ oph, size_t nvec)
{
	using U = typename std::make_unsigned<T>::type;
	for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < nvec;
			g += (size_t)gridDim.x * kBlock) {
		const i32x4 tx = xin[g];
		const i32x4 ty = yin[g];
		T x[kVec], y[kVec];
		uint32_t p[kVec];
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			const int32_t ix = sext32(tx[v], kp.iw);
			const int32_t iy = sext32(ty[v], kp.iw);
			const T ex = (T)((U)(T)ix << kp.in_shl);
			const T ey = (T)((U)(T)iy << kp.in_shl);
			fold_quadrant<T>(ex, ey, ix < 0, iy < 0, x[v], y[v], p[v]);
		}

		PolChain<T, NLIVE>::run(x, y, p, kp);

		i32x4 rm;
		u32x4 rp;
#pragma unroll
		for (int v = 0; v < kVec; v++) {
			rm[v] = round_to_ow<T>(x[v], kp);
			rp[v] = p[v] >> kp.pw_shl;	// rtl/topolar.v:269
		}
		__builtin_nontemporal_store(rm, &omag[g]);
		__builtin_nontemporal_store(rp, &oph[g]);
	}
}

// ------------------------------------------------------------ generic path
//
// Any parameter set, any alignment: one sample per lane per pass, 64-bit
// container, run-time stage count and shifts, optional explicit WW-bit wrap
// after every operation (the literal register semantics).  Slower; used for
// stage counts without an unrolled instance, for cores whose WW-bit registers
// can overflow, and for buffers that are not 16-byte aligned.

__device__ __forceinline__ int64_t wrap_ww(int64_t v, const CoreParams &kp)
{
	return kp.wrap ? sext64(v, kp.ww) : v;
}
__device__ __forceinline__ int32_t round_generic(int64_t v, const CoreParams &kp)
{
	const uint64_t b = ((uint64_t)v >> kp.r) & (uint64_t)kp.round_bit;
	int64_t w = (int64_t)((uint64_t)v + (uint64_t)kp.round_base + b);
	w = wrap_ww(w, kp);
	const int32_t o = (int32_t)(w >> kp.r);
	return kp.wrap ? sext32(o, kp.ow) : o;
}

template <Feed FEED>
__global__ __launch_bounds__(kBlock) void rotator_generic(CoreParams kp,
		const int32_t *__restrict__ xin, const int32_t *__restrict__ yin,
		const uint32_t *__restrict__ phin, int32_t *__restrict__ ox,
		int32_t *__restrict__ oy, size_t n)
{
	const size_t stride = (size_t)gridDim.x * kBlock;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
			i += stride) {
		uint32_t P;
		int32_t ix, iy;
		if constexpr (FEED == Feed::Nco_ConstXY)
			P = kp.phase0 + (uint32_t)(kp.index0 + i) * kp.fcw;
		else
			P = phin[i] << kp.pw_shl;
		if constexpr (FEED == Feed::PhaseArray_XYArray) {
			ix = sext32(xin[i], kp.iw);
			iy = sext32(yin[i], kp.iw);
		} else {
			ix = kp.x0;
			iy = kp.y0;
		}
		const int64_t ex = (int64_t)((uint64_t)(int64_t)ix << kp.in_shl);
		const int64_t ey = (int64_t)((uint64_t)(int64_t)iy << kp.in_shl);
		int64_t x, y;
		uint32_t p;
		fold_octant<int64_t>(ex, ey, P, x, y, p);
		x = wrap_ww(x, kp);
		y = wrap_ww(y, kp);
		for (int s = 0; s < kp.nlive; s++) {
			const int k = (s + 1 > 63) ? 63 : s + 1;
			const uint32_t a = kp.angle[s];
			const int64_t sy = y >> k, sx = x >> k;
			if ((int32_t)p < 0) {
				x = x + sy; y = y - sx; p += a;
			} else {
				x = x - sy; y = y + sx; p -= a;
			}
			x = wrap_ww(x, kp);
			y = wrap_ww(y, kp);
		}
		ox[i] = round_generic(x, kp);
		oy[i] = round_generic(y, kp);
	}
}

__global__ __launch_bounds__(kBlock) void topolar_generic(CoreParams kp,
		const int32_t *__restrict__ xin, const int32_t *__restrict__ yin,
		int32_t *__restrict__ omag, uint32_t *__restrict__ oph, size_t n)
{
	const size_t stride = (size_t)gridDim.x * kBlock;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
			i += stride) {
		const int32_t ix = sext32(xin[i], kp.iw);
		const int32_t iy = sext32(yin[i], kp.iw);
		const int64_t ex = (int64_t)((uint64_t)(int64_t)ix << kp.in_shl);
		const int64_t ey = (int64_t)((uint64_t)(int64_t)iy << kp.in_shl);
		int64_t x, y;
		uint32_t p;
		fold_quadrant<int64_t>(ex, ey, ix < 0, iy < 0, x, y, p);
		x = wrap_ww(x, kp);
		y = wrap_ww(y, kp);
		for (int s = 0; s < kp.nlive; s++) {
			const int k = (s + 1 > 63) ? 63 : s + 1;
			const uint32_t a = kp.angle[s];
			const int64_t sy = y >> k, sx = x >> k;
			if (y < 0) {
				x = x - sy; y = y + sx; p -= a;
			} else {
				x = x + sy; y = y - sx; p += a;
			}
			x = wrap_ww(x, kp);
			y = wrap_ww(y, kp);
		}
		omag[i] = round_generic(x, kp);
		oph[i] = p >> kp.pw_shl;
	}
}

// ------------------------------------------------------ test-input kernels

__global__ __launch_bounds__(kBlock) void fill_phase_ramp(uint32_t *p, size_t n,
		uint64_t index0, int shift)
{
	const size_t stride = (size_t)gridDim.x * kBlock;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
			i += stride)
		p[i] = (uint32_t)(index0 + i) << shift;
}

__global__ __launch_bounds__(kBlock) void fill_iq_ramp(int32_t *x, int32_t *y,
		size_t n, uint64_t index0, uint32_t mulx, uint32_t muly, int bits)
{
	const size_t stride = (size_t)gridDim.x * kBlock;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
			i += stride) {
		const uint32_t g = (uint32_t)(index0 + i);
		x[i] = sext32((int32_t)((g * mulx) >> 8), bits);
		y[i] = sext32((int32_t)((g * muly) >> 8), bits);
	}
}

// mix(): a 64-bit finaliser over (global index, word) so that the digest is
// sensitive to both value and position, yet shards simply add.
__device__ __forceinline__ uint64_t digest_mix(uint64_t idx, uint32_t w)
{
	uint64_t z = (idx + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)w;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

__global__ __launch_bounds__(kBlock) void digest_u32(const uint32_t *w, size_t n,
		uint64_t index0, unsigned long long *out)
{
	__shared__ unsigned long long part[kBlock / 64];
	unsigned long long acc = 0;
	const size_t stride = (size_t)gridDim.x * kBlock;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
			i += stride)
		acc += digest_mix(index0 + i, w[i]);
	for (int off = 32; off > 0; off >>= 1)
		acc += __shfl_down(acc, off, 64);
	if ((threadIdx.x & 63) == 0)
		part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long s = 0;
		for (int k = 0; k < kBlock / 64; k++)
			s += part[k];
		atomicAdd(out, s);
	}
}

// ------------------------------------------------------------ host helpers

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int grid_for(size_t work_items_per_block, size_t n)
{
	static int cus = 0;
	if (cus == 0) {
		int dev = 0;
		hipDeviceProp_t prop;
		if (hipGetDevice(&dev) != hipSuccess ||
		    hipGetDeviceProperties(&prop, dev) != hipSuccess)
			return -1;
		cus = prop.multiProcessorCount;
	}
	const size_t blocks = (n + work_items_per_block - 1) / work_items_per_block;
	const size_t cap = (size_t)cus * 8;	// 8 x 256-thread blocks per CU
	return (int)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

int check_launch()
{
	return (hipGetLastError() == hipSuccess) ? CORDIC_OK : CORDIC_ERR_DEVICE;
}

CoreParams make_params(const cordic_config &c)
{
	CoreParams kp{};
	const bool rot = (c.mode == CORDIC_P2R || c.mode == CORDIC_SP2R);
	const int lsh = 32 - c.pw;
	for (int i = 0; i < CORDIC_AMD_MAX_STAGES; i++)
		kp.angle[i] = (i < c.nstages) ? (c.angle[i] << lsh) : 0u;
	kp.nlive = c.nlive;
	kp.iw = c.iw;
	kp.in_shl = rot ? (c.ww - c.iw - 1) : (c.ww - c.iw - 2);
	kp.pw_shl = lsh;
	kp.ww = c.ww;
	kp.ow = c.ow;
	kp.r = c.ww - c.ow;
	const bool rounding = c.ww > c.ow + 1;
	kp.round_bit = rounding ? 1u : 0u;
	kp.round_base = rounding ? (((int64_t)1 << (kp.r - 1)) - 1) : 0;
	kp.wrap = c.needs_wrap && c.ww < 64;
	return kp;
}

int32_t host_sext(int32_t v, int w)
{
	const int s = 32 - w;
	return (int32_t)((uint32_t)v << s) >> s;
}

// The unrolled instances that exist.  Anything else runs on the generic
// kernels (same results, lower throughput).
#define CORDIC_ROT_STAGES(X) X(13) X(14) X(16) X(18) X(20) X(22) X(24) X(30)
#define CORDIC_POL_STAGES(X) X(16) X(18) X(20) X(24) X(30)

template <typename T, Feed FEED>
bool launch_rot_unrolled(int nlive, int grid, hipStream_t st, const CoreParams &kp,
		const RotatorJob &j)
{
	switch (nlive) {
#define X(N) case N: \
	hipLaunchKernelGGL((rotator_unrolled<T, N, FEED>), dim3(grid), \
		dim3(kBlock), 0, st, kp, (const i32x4 *)j.x, (const i32x4 *)j.y, \
		(const u32x4 *)j.phase, (i32x4 *)j.ox, (i32x4 *)j.oy, j.n / kVec); \
	return true;
	CORDIC_ROT_STAGES(X)
#undef X
	default:
		return false;
	}
}

template <Feed FEED>
int launch_rot_feed(const cordic_config &cfg, const RotatorJob &j, void *stream)
{
	hipStream_t st = static_cast<hipStream_t>(stream);
	CoreParams kp = make_params(cfg);
	kp.x0 = host_sext(j.x0, cfg.iw);
	kp.y0 = host_sext(j.y0, cfg.iw);
	kp.phase0 = j.phase0 << kp.pw_shl;
	kp.fcw = j.fcw << kp.pw_shl;
	kp.index0 = j.index0;

	bool vec_ok = aligned16(j.ox) && aligned16(j.oy);
	if (FEED != Feed::Nco_ConstXY)
		vec_ok = vec_ok && aligned16(j.phase);
	if (FEED == Feed::PhaseArray_XYArray)
		vec_ok = vec_ok && aligned16(j.x) && aligned16(j.y);
	const bool fast_ok = vec_ok && !(cfg.flags & CORDIC_FLAG_FORCE_GENERIC)
		&& (!cfg.needs_wrap || cfg.ww == 32 || cfg.ww == 64);

	if (fast_ok) {
		const int grid = grid_for(kTile, j.n);
		if (grid < 0)
			return CORDIC_ERR_DEVICE;
		const bool done = (j.n < (size_t)kVec) ? false : (cfg.ww <= 32)
			? launch_rot_unrolled<int32_t, FEED>(cfg.nlive, grid, st, kp, j)
			: launch_rot_unrolled<int64_t, FEED>(cfg.nlive, grid, st, kp, j);
		if (done) {
			// 0..3 trailing samples: generic kernel on the remainder
			const size_t head = j.n - j.n % kVec;
			if (head == j.n)
				return check_launch();
			RotatorJob t = j;
			t.n = j.n - head;
			t.ox += head; t.oy += head;
			if (t.phase) t.phase += head;
			if (t.x) t.x += head;
			if (t.y) t.y += head;
			kp.index0 += head;
			hipLaunchKernelGGL((rotator_generic<FEED>), dim3(1),
				dim3(kBlock), 0, st, kp, t.x, t.y, t.phase, t.ox,
				t.oy, t.n);
			return check_launch();
		}
	}
	const int grid = grid_for(kBlock, j.n);
	if (grid < 0)
		return CORDIC_ERR_DEVICE;
	hipLaunchKernelGGL((rotator_generic<FEED>), dim3(grid), dim3(kBlock), 0,
			st, kp, j.x, j.y, j.phase, j.ox, j.oy, j.n);
	return check_launch();
}

template <typename T>
bool launch_pol_unrolled(int nlive, int grid, hipStream_t st, const CoreParams &kp,
		const int32_t *x, const int32_t *y, int32_t *mag, uint32_t *ph,
		size_t n)
{
	switch (nlive) {
#define X(N) case N: \
	hipLaunchKernelGGL((topolar_unrolled<T, N>), dim3(grid), dim3(kBlock), \
		0, st, kp, (const i32x4 *)x, (const i32x4 *)y, (i32x4 *)mag, \
		(u32x4 *)ph, n / kVec); \
	return true;
	CORDIC_POL_STAGES(X)
#undef X
	default:
		return false;
	}
}

} // namespace

// ----------------------------------------------------------------- launchers

int launch_rotator(const cordic_config &cfg, Feed feed, const RotatorJob &job,
		void *stream)
{
	if (cfg.mode != CORDIC_P2R && cfg.mode != CORDIC_SP2R)
		return CORDIC_ERR_MODE;
	if (job.n == 0)
		return CORDIC_OK;
	if (!job.ox || !job.oy)
		return CORDIC_ERR_ARGS;
	switch (feed) {
	case Feed::PhaseArray_ConstXY:
		if (!job.phase) return CORDIC_ERR_ARGS;
		return launch_rot_feed<Feed::PhaseArray_ConstXY>(cfg, job, stream);
	case Feed::PhaseArray_XYArray:
		if (!job.phase || !job.x || !job.y) return CORDIC_ERR_ARGS;
		return launch_rot_feed<Feed::PhaseArray_XYArray>(cfg, job, stream);
	default:
		return launch_rot_feed<Feed::Nco_ConstXY>(cfg, job, stream);
	}
}

int launch_topolar(const cordic_config &cfg, size_t n, const int32_t *x,
		const int32_t *y, int32_t *mag, uint32_t *phase, void *stream)
{
	if (cfg.mode != CORDIC_R2P && cfg.mode != CORDIC_SR2P)
		return CORDIC_ERR_MODE;
	if (n == 0)
		return CORDIC_OK;
	if (!x || !y || !mag || !phase)
		return CORDIC_ERR_ARGS;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const CoreParams kp = make_params(cfg);
	const bool vec_ok = aligned16(x) && aligned16(y) && aligned16(mag)
			&& aligned16(phase);
	const bool fast_ok = vec_ok && !(cfg.flags & CORDIC_FLAG_FORCE_GENERIC)
		&& (!cfg.needs_wrap || cfg.ww == 32 || cfg.ww == 64);
	if (fast_ok) {
		const int grid = grid_for(kTile, n);
		if (grid < 0)
			return CORDIC_ERR_DEVICE;
		const bool done = (n < (size_t)kVec) ? false : (cfg.ww <= 32)
			? launch_pol_unrolled<int32_t>(cfg.nlive, grid, st, kp, x, y,
					mag, phase, n)
			: launch_pol_unrolled<int64_t>(cfg.nlive, grid, st, kp, x, y,
					mag, phase, n);
		if (done) {
			const size_t head = n - n % kVec;
			if (head == n)
				return check_launch();
			hipLaunchKernelGGL(topolar_generic, dim3(1), dim3(kBlock), 0,
				st, kp, x + head, y + head, mag + head, phase + head,
				n - head);
			return check_launch();
		}
	}
	const int grid = grid_for(kBlock, n);
	if (grid < 0)
		return CORDIC_ERR_DEVICE;
	hipLaunchKernelGGL(topolar_generic, dim3(grid), dim3(kBlock), 0, st, kp,
			x, y, mag, phase, n);
	return check_launch();
}

int launch_fill_phase_ramp(uint32_t *p, size_t n, uint64_t index0, int shift,
		void *stream)
{
	if (n == 0) return CORDIC_OK;
	if (!p || shift < 0 || shift > 31) return CORDIC_ERR_ARGS;
	const int grid = grid_for(kBlock * 4, n);
	if (grid < 0) return CORDIC_ERR_DEVICE;
	hipLaunchKernelGGL(fill_phase_ramp, dim3(grid), dim3(kBlock), 0,
			static_cast<hipStream_t>(stream), p, n, index0, shift);
	return check_launch();
}

int launch_fill_iq_ramp(int32_t *x, int32_t *y, size_t n, uint64_t index0,
		uint32_t mulx, uint32_t muly, int bits, void *stream)
{
	if (n == 0) return CORDIC_OK;
	if (!x || !y || bits < 1 || bits > 32) return CORDIC_ERR_ARGS;
	const int grid = grid_for(kBlock * 4, n);
	if (grid < 0) return CORDIC_ERR_DEVICE;
	hipLaunchKernelGGL(fill_iq_ramp, dim3(grid), dim3(kBlock), 0,
			static_cast<hipStream_t>(stream), x, y, n, index0, mulx,
			muly, bits);
	return check_launch();
}

int launch_digest_u32(const uint32_t *w, size_t n, uint64_t index0,
		uint64_t *digest, void *stream)
{
	if (n == 0) return CORDIC_OK;
	if (!w || !digest) return CORDIC_ERR_ARGS;
	const int grid = grid_for(kBlock * 8, n);
	if (grid < 0) return CORDIC_ERR_DEVICE;
	hipLaunchKernelGGL(digest_u32, dim3(grid), dim3(kBlock), 0,
			static_cast<hipStream_t>(stream), w, n, index0,
			reinterpret_cast<unsigned long long *>(digest));
	return check_launch();
}

} // namespace cordic_amd
