// valu_microbench.hip -- throughput of the integer VALU instructions the
// CORDIC stage can be built from, measured on the machine it runs on.
// Decides the shape of the 64-bit (WW=35) micro-rotation: native 64-bit
// shift/add (v_ashrrev_i64, v_lshl_add_u64) versus 32-bit pairs
// (v_alignbit_b32 + v_ashrrev_i32, v_add_co_u32 + v_addc_co_u32).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_microbench valu_microbench.hip
//   ./valu_microbench            # prints one line per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
	printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 2048;
constexpr int kChains = 16;

// 32-bit ops: 16 independent chains a[i] = op(a[i], b, c)
#define KERNEL32(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c) { \
	uint32_t a[kChains]; \
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i; \
	uint32_t vb = b + threadIdx.x, vc = c ^ threadIdx.x; \
	for (int it = 0; it < kIters; it++) { \
		_Pragma("unroll") \
		for (int i = 0; i < kChains; i++) \
			asm volatile(ASM : "+v"(a[i]) : "v"(vb), "v"(vc) : "vcc", "s10", "s11"); \
	} \
	uint32_t s = 0; \
	for (int i = 0; i < kChains; i++) s ^= a[i]; \
	out[blockIdx.x * 256 + threadIdx.x] = s; \
}

#define KERNEL64(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c) { \
	uint64_t a[kChains]; \
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i; \
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x; \
	uint32_t vc = c ^ threadIdx.x; \
	for (int it = 0; it < kIters; it++) { \
		_Pragma("unroll") \
		for (int i = 0; i < kChains; i++) \
			asm volatile(ASM : "+v"(a[i]) : "v"(vb), "v"(vc) : "vcc", "s10", "s11"); \
	} \
	uint64_t s = 0; \
	for (int i = 0; i < kChains; i++) s ^= a[i]; \
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32)); \
}

KERNEL32(k_add_u32,     "v_add_u32 %0, %0, %1")
KERNEL32(k_sub_u32,     "v_sub_u32 %0, %0, %1")
KERNEL32(k_xor_b32,     "v_xor_b32 %0, %0, %1")
KERNEL32(k_not_b32,     "v_not_b32 %0, %0")
KERNEL32(k_ashr_i32,    "v_ashrrev_i32 %0, 3, %0")
KERNEL32(k_alignbit,    "v_alignbit_b32 %0, %1, %0, 3")
KERNEL32(k_xad_u32,     "v_xad_u32 %0, %0, %1, %2")
KERNEL32(k_add3_u32,    "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_bfe_i32,     "v_bfe_i32 %0, %0, 1, 30")
KERNEL32(k_and_or,      "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_lshl_add,    "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL32(k_mul_lo,      "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mad_u24,     "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_cndmask,     "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp_gt,      "v_cmp_gt_i32 vcc, %0, %1\n\tv_add_u32 %0, %0, %2")
KERNEL32(k_pk_add_u16,  "v_pk_add_u16 %0, %0, %1")
KERNEL32(k_pk_ashr_i16, "v_pk_ashrrev_i16 %0, 3, %0")
KERNEL32(k_mov_dpp,     "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_fma_f32,     "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_or_b32,      "v_or_b32 %0, %0, %1")
KERNEL32(k_lshlrev_b32, "v_lshlrev_b32 %0, 1, %0")
KERNEL32(k_min_i32,     "v_min_i32 %0, %0, %1")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mul_i32_i24, "v_mul_i32_i24 %0, %0, %1")
KERNEL32(k_bitop3,      "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL32(k_perm_b32,    "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_med3_i32,    "v_med3_i32 %0, %0, %1, %2")
KERNEL32(k_xor_inline,  "v_xor_b32 %0, -2, %0")
KERNEL32(k_add_e64,     "v_add_u32_e64 %0, %0, %1")
KERNEL32(k_fmac_f32,    "v_fmac_f32 %0, %1, %2")
KERNEL32(k_pk_fma_f32x, "v_add_f32 %0, %0, %1")
// carry chain pair: lo += b (carry to vcc), hi(%2 reused) += 0 + carry.
// two wait states are owed between the VCC write and the VCC read on gfx940+,
// filled here with the two ops of the *other* half of the pair pattern.
KERNEL32(k_addco_pair,  "v_add_co_u32 %0, vcc, %0, %1\n\ts_nop 1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc")
KERNEL32(k_addco_only,  "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc_only,   "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_subb_e64,    "v_subb_co_u32 %0, s[10:11], %0, %1, s[10:11]")

KERNEL64(k_ashr_i64,    "v_ashrrev_i64 %0, 3, %0")
KERNEL64(k_lshr_b64,    "v_lshrrev_b64 %0, 3, %0")
KERNEL64(k_lshl_add_u64,"v_lshl_add_u64 %0, %0, 0, %1")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %2, %2, %0")
KERNEL64(k_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %2, %2, %0")
KERNEL64(k_mad_i64_sgpr,"v_mad_i64_i32 %0, s[10:11], %2, %2, %0")
KERNEL64(k_pk_add_f32,  "v_pk_add_f32 %0, %0, %1")
KERNEL64(k_mov_b64,     "v_mov_b64 %0, %1")
KERNEL64(k_add_f64,     "v_add_f64 %0, %0, %1")
KERNEL64(k_fma_f64,     "v_fma_f64 %0, %0, %1, %1")

struct Case { const char *name; void (*fn)(uint32_t *, uint32_t, uint32_t); int instr; };

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const int blocks = cus * 8;		// 8 x 256 threads per CU: 8 waves/SIMD
	uint32_t *out;
	CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
	printf("device %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
	std::vector<Case> cases = {
#define C(n, i) { #n, n, i }
		C(k_add_u32, 1), C(k_sub_u32, 1), C(k_xor_b32, 1), C(k_not_b32, 1),
		C(k_ashr_i32, 1), C(k_alignbit, 1), C(k_xad_u32, 1), C(k_add3_u32, 1),
		C(k_bfe_i32, 1), C(k_and_or, 1), C(k_lshl_add, 1), C(k_mul_lo, 1),
		C(k_mad_u24, 1), C(k_cndmask, 1), C(k_cmp_gt, 2), C(k_pk_add_u16, 1),
		C(k_pk_ashr_i16, 1), C(k_mov_dpp, 1), C(k_fma_f32, 1), C(k_or_b32, 1), C(k_lshlrev_b32, 1), C(k_min_i32, 1), C(k_mul_u32_u24, 1), C(k_mul_i32_i24, 1), C(k_bitop3, 1), C(k_perm_b32, 1), C(k_med3_i32, 1), C(k_xor_inline, 1), C(k_add_e64, 1), C(k_fmac_f32, 1), C(k_pk_fma_f32x, 1),
		C(k_addco_pair, 2), C(k_addco_only, 1), C(k_addc_only, 1), C(k_subb_e64, 1),
		C(k_ashr_i64, 1), C(k_lshr_b64, 1), C(k_lshl_add_u64, 1),
		C(k_mad_u64_u32, 1), C(k_mad_i64_i32, 1), C(k_mad_i64_sgpr, 1), C(k_pk_add_f32, 1), C(k_mov_b64, 1), C(k_add_f64, 1), C(k_fma_f64, 1),
	};
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (auto &c : cases) {
		hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
		CHECK(hipDeviceSynchronize());
		float best = 1e30f;
		for (int rep = 0; rep < 5; rep++) {
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		const double lane_ops = (double)blocks * 256 * kIters * kChains;
		const double tops = lane_ops / (best * 1e-3) / 1e12;
		// cycles per wave-statement per SIMD at 2.4 GHz: waves*stmts / (SIMDs*time*clk)
		const double stmts = (double)blocks * 4 * kIters * kChains;
		const double cyc = (best * 1e-3) * 2.4e9 * (cus * 4.0) / stmts;
		printf("%-16s %8.3f ms  %7.2f Tstmt/s (lane)  %5.2f cyc/wave-stmt/SIMD @2.4GHz (%d instr/stmt)\n",
			c.name, best, tops, cyc, c.instr);
	}
	return 0;
}
