// stage_microbench.hip -- how fast does the chip issue the CORDIC
// micro-rotation itself, away from memory?  The product kernels issue one VALU
// instruction per ~4.4 shader cycles per SIMD where the per-opcode rates of
// valu_microbench.hip predict ~3.1 for the same mix; this probe runs the exact
// stage code of cordic_device.h (rot_stage_lj: 2 v_bitop3_b32, 2 v_ashrrev_i32,
// 3 v_mad_i64_i32; pol_stage_lj: 3 / 2 / 2) in a register-only loop and
// reports, per variant,
//   * the shader clock the chip really ran at (s_memtime ticks per
//     s_memrealtime tick x 100 MHz) -- the cycle figures elsewhere assume
//     2.4 GHz,
//   * VALU instructions per SIMD-cycle at that clock.
// Variants: CH independent dependency chains per lane (the product uses 4),
// W waves per SIMD (the product runs 8).
//
//   hipcc --offload-arch=gfx950 -O3 -I../include -I../cordic_amd/csrc \
//         -o stage_microbench stage_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#include "cordic_device.h"

using namespace cordic_amd::dev;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
	printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kRounds = 512;	// passes of 16 stages

// r2p stage with every v_bitop3_b32 operand in a VGPR (the shipped form passes
// 2^31 in an SGPR and 2^30 / -2^30 as inline constants) and the direction-bit
// mask kept in a VGPR that is shifted once per stage for all chains
__device__ __forceinline__ void pol_stage_vg(int64_t &x, int64_t &y, uint32_t &dirs,
		const LjRegs &c, uint32_t bitmask, int sh)
{
	const uint32_t yh = (uint32_t)((uint64_t)y >> 32);
	const int32_t t = (int32_t)op_and_or(yh, c.bit, c.mask);
	const int32_t nt = (int32_t)op_and_xor(yh, c.maskbit, c.mask);
	const int32_t sy = (int32_t)yh >> sh;
	const int32_t sx = (int32_t)((uint64_t)x >> 32) >> sh;
	dirs = op_and_or(yh, dirs, bitmask);
	op_mad(x, sy, t);
	op_mad(y, sx, nt);
}

// r2p stage that keeps the phase multiply-add (7 instructions: 2 bitop3 with
// VGPR constants, 2 shifts, 3 mads) instead of collecting direction bits
__device__ __forceinline__ void pol_stage_mad(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a, const LjRegs &c, int sh)
{
	const uint32_t yh = (uint32_t)((uint64_t)y >> 32);
	const int32_t t = (int32_t)op_and_or(yh, c.bit, c.mask);
	const int32_t nt = (int32_t)op_and_xor(yh, c.maskbit, c.mask);
	const int32_t sy = (int32_t)yh >> sh;
	const int32_t sx = (int32_t)((uint64_t)x >> 32) >> sh;
	op_mad(x, sy, t);
	op_mad(y, sx, nt);
	op_mad_s(p, a, t);
}
// the same with -t as one VOP2 xor of t with 2^31 (a literal)
__device__ __forceinline__ void pol_stage_mad_x(int64_t &x, int64_t &y, int64_t &p,
		uint32_t a, const LjRegs &c, int sh)
{
	const uint32_t yh = (uint32_t)((uint64_t)y >> 32);
	const int32_t t = (int32_t)op_and_or(yh, c.bit, c.mask);
	int32_t nt;
	asm("v_xor_b32 %0, 0x80000000, %1" : "=v"(nt) : "v"(t));
	const int32_t sy = (int32_t)yh >> sh;
	const int32_t sx = (int32_t)((uint64_t)x >> 32) >> sh;
	op_mad(x, sy, t);
	op_mad(y, sx, nt);
	op_mad_s(p, a, t);
}

template <int CH, int KIND>
__global__ __launch_bounds__(256) void stages(uint64_t *out, uint64_t *clk, uint32_t a0,
		uint32_t seed)
{
	int64_t x[CH], y[CH], p[CH];
	uint32_t dirs[CH];
	for (int c = 0; c < CH; c++) {
		x[c] = (int64_t)(threadIdx.x * 977 + c * 131 + seed) << 30;
		y[c] = (int64_t)(threadIdx.x * 331 + c * 17 + seed) << 29;
		p[c] = (int64_t)(int32_t)(threadIdx.x * 7919u + c + seed) << 31;
		dirs[c] = 0;
	}
	LjRegs ljc{};
	ljc.mask = vgpr_const(LjConst<29>::mask);
	ljc.bit = vgpr_const(LjConst<29>::bit);
	ljc.maskbit = vgpr_const(LjConst<29>::mask | LjConst<29>::bit);
	const uint32_t smask = 0x80000000u;
	const uint64_t t0 = __builtin_readcyclecounter();	// s_memtime
	const uint64_t w0 = wall_clock64();			// s_memrealtime, 100 MHz
#pragma unroll 1
	for (int r = 0; r < kRounds; r++) {
		// 16 stages, stage-major over the chains exactly as RotChainLJ does
#define STAGE(K) \
		_Pragma("unroll") for (int c = 0; c < CH; c++) { \
			if constexpr (KIND == 0) rot_stage_lj<29, K>(x[c], y[c], p[c], a0 >> (K & 7), ljc); \
			else pol_stage_lj<(K % 22) + 2>(x[c], y[c], dirs[c], smask); }
		if constexpr (KIND == 3 || KIND == 4) {
			LjRegs pc{};
			pc.mask = vgpr_const(0x80000000u);
			pc.bit = vgpr_const(0x40000000u);
			pc.maskbit = vgpr_const(0xc0000000u);
#pragma unroll
			for (int k = 0; k < 16; k++) {
#pragma unroll
				for (int c = 0; c < CH; c++) {
					if constexpr (KIND == 3)
						pol_stage_mad(x[c], y[c], p[c], a0 >> (k & 7), pc, k);
					else
						pol_stage_mad_x(x[c], y[c], p[c], a0 >> (k & 7), pc, k);
				}
			}
		} else if constexpr (KIND == 2) {
			LjRegs pc{};
			pc.mask = vgpr_const(0x80000000u);
			pc.bit = vgpr_const(0x40000000u);
			pc.maskbit = vgpr_const(0xc0000000u);
			uint32_t bm = vgpr_const(0x40000000u);
#pragma unroll
			for (int k = 0; k < 16; k++) {
#pragma unroll
				for (int c = 0; c < CH; c++)
					pol_stage_vg(x[c], y[c], dirs[c], pc, bm, k);
				asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(bm));
			}
		} else {
		STAGE(3) STAGE(4) STAGE(5) STAGE(6) STAGE(7) STAGE(8) STAGE(9) STAGE(10)
		STAGE(11) STAGE(12) STAGE(13) STAGE(14) STAGE(15) STAGE(16) STAGE(17) STAGE(18)
		}
#undef STAGE
	}
	const uint64_t t1 = __builtin_readcyclecounter();
	const uint64_t w1 = wall_clock64();
	uint64_t s = 0;
	for (int c = 0; c < CH; c++)
		s ^= (uint64_t)x[c] ^ (uint64_t)y[c] ^ (uint64_t)p[c] ^ dirs[c];
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if (threadIdx.x == 0) {
		clk[blockIdx.x * 2] = t1 - t0;
		clk[blockIdx.x * 2 + 1] = w1 - w0;
	}
}

template <int CH, int KIND>
static int run(const char *name, int waves_per_simd, uint64_t *out, uint64_t *clk)
{
	int cus = 256;
	CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
	const int blocks = cus * waves_per_simd;	// 256 threads = 4 waves = 1 per SIMD
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((stages<CH, KIND>), dim3(blocks), dim3(256), 0, 0, out, clk, 0x12e4051du, 1u);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	const int reps = 5;
	for (int r = 0; r < reps; r++)
		hipLaunchKernelGGL((stages<CH, KIND>), dim3(blocks), dim3(256), 0, 0, out, clk, 0x12e4051du, (uint32_t)r);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	ms /= reps;
	std::vector<uint64_t> h((size_t)blocks * 2);
	CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
	double tick = 0, wall = 0;
	for (int b = 0; b < blocks; b++) { tick += (double)h[2 * b]; wall += (double)h[2 * b + 1]; }
	const double mhz = tick / wall * 100.0;		// s_memrealtime = 100 MHz
	const double per_stage = (KIND == 0) ? 7.0 : 7.0;
	const double instr_per_wave = (double)kRounds * 16 * CH * per_stage;
	// per SIMD: waves_per_simd waves, each instr_per_wave instructions, in
	// (tick / blocks) cycles on average
	const double cyc = tick / blocks;
	printf("%-34s CH %d  waves/SIMD %d  %7.3f ms  shader clock %6.0f MHz  "
		"%.2f cycles per VALU instruction per SIMD (%.2f at 2.4 GHz wall)\n",
		name, CH, waves_per_simd, ms, mhz, cyc / (instr_per_wave * waves_per_simd),
		ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd));
	return 0;
}

int main()
{
	uint64_t *out, *clk;
	CHECK(hipMalloc(&out, (size_t)256 * 16 * 256 * 8));
	CHECK(hipMalloc(&clk, (size_t)256 * 16 * 2 * 8));
	printf("# stage = 7 VALU instructions; mix-weighted prediction from valu_microbench: "
		"p2r (3 mad 4.2, 2 bitop3 2.4, 2 ashr 2.5) = 3.2 cycles/instr, r2p (2 mad) = 2.9\n");
	for (int w : {1, 2, 4, 8}) {
		run<4, 0>("p2r stage (rot_stage_lj<29>)", w, out, clk);
		run<8, 0>("p2r stage (rot_stage_lj<29>)", w, out, clk);
	}
	for (int w : {4, 8}) {
		run<2, 0>("p2r stage (rot_stage_lj<29>)", w, out, clk);
		run<4, 1>("r2p stage (pol_stage_lj)", w, out, clk);
		run<8, 1>("r2p stage (pol_stage_lj)", w, out, clk);
		run<4, 2>("r2p stage, all-VGPR bitop3", w, out, clk);
		run<8, 2>("r2p stage, all-VGPR bitop3", w, out, clk);
		run<4, 3>("r2p stage, phase mad kept", w, out, clk);
		run<8, 3>("r2p stage, phase mad kept", w, out, clk);
		run<4, 4>("r2p stage, phase mad, -t by xor", w, out, clk);
		run<8, 4>("r2p stage, phase mad, -t by xor", w, out, clk);
	}
	return 0;
}
