// valu_microbench.hip -- one kernel per instruction, all of the same shape.
//
// Throughput of the integer VALU instructions the CORDIC stage can be built
// from, measured on the machine it runs on.  Every kernel issues 16
// independent chains of ONE instruction inside a single asm statement (so the
// compiler adds no s_nop padding between them), 2048 times, from 8 waves per
// SIMD.  Reported: wave-instructions per second per SIMD expressed as cycles
// per instruction at the 2.4 GHz peak clock (the chip may clock lower).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_microbench valu_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
	printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 2048;
constexpr int kChains = 16;
__global__ __launch_bounds__(256) void k_add_u32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_add_u32 %0, %0, %16\n\tv_add_u32 %1, %1, %16\n\tv_add_u32 %2, %2, %16\n\tv_add_u32 %3, %3, %16\n\tv_add_u32 %4, %4, %16\n\tv_add_u32 %5, %5, %16\n\tv_add_u32 %6, %6, %16\n\tv_add_u32 %7, %7, %16\n\tv_add_u32 %8, %8, %16\n\tv_add_u32 %9, %9, %16\n\tv_add_u32 %10, %10, %16\n\tv_add_u32 %11, %11, %16\n\tv_add_u32 %12, %12, %16\n\tv_add_u32 %13, %13, %16\n\tv_add_u32 %14, %14, %16\n\tv_add_u32 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_sub_u32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_sub_u32 %0, %0, %16\n\tv_sub_u32 %1, %1, %16\n\tv_sub_u32 %2, %2, %16\n\tv_sub_u32 %3, %3, %16\n\tv_sub_u32 %4, %4, %16\n\tv_sub_u32 %5, %5, %16\n\tv_sub_u32 %6, %6, %16\n\tv_sub_u32 %7, %7, %16\n\tv_sub_u32 %8, %8, %16\n\tv_sub_u32 %9, %9, %16\n\tv_sub_u32 %10, %10, %16\n\tv_sub_u32 %11, %11, %16\n\tv_sub_u32 %12, %12, %16\n\tv_sub_u32 %13, %13, %16\n\tv_sub_u32 %14, %14, %16\n\tv_sub_u32 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_xor_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_xor_b32 %0, %0, %16\n\tv_xor_b32 %1, %1, %16\n\tv_xor_b32 %2, %2, %16\n\tv_xor_b32 %3, %3, %16\n\tv_xor_b32 %4, %4, %16\n\tv_xor_b32 %5, %5, %16\n\tv_xor_b32 %6, %6, %16\n\tv_xor_b32 %7, %7, %16\n\tv_xor_b32 %8, %8, %16\n\tv_xor_b32 %9, %9, %16\n\tv_xor_b32 %10, %10, %16\n\tv_xor_b32 %11, %11, %16\n\tv_xor_b32 %12, %12, %16\n\tv_xor_b32 %13, %13, %16\n\tv_xor_b32 %14, %14, %16\n\tv_xor_b32 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_or_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_or_b32 %0, %0, %16\n\tv_or_b32 %1, %1, %16\n\tv_or_b32 %2, %2, %16\n\tv_or_b32 %3, %3, %16\n\tv_or_b32 %4, %4, %16\n\tv_or_b32 %5, %5, %16\n\tv_or_b32 %6, %6, %16\n\tv_or_b32 %7, %7, %16\n\tv_or_b32 %8, %8, %16\n\tv_or_b32 %9, %9, %16\n\tv_or_b32 %10, %10, %16\n\tv_or_b32 %11, %11, %16\n\tv_or_b32 %12, %12, %16\n\tv_or_b32 %13, %13, %16\n\tv_or_b32 %14, %14, %16\n\tv_or_b32 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_not_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_not_b32 %0, %0\n\tv_not_b32 %1, %1\n\tv_not_b32 %2, %2\n\tv_not_b32 %3, %3\n\tv_not_b32 %4, %4\n\tv_not_b32 %5, %5\n\tv_not_b32 %6, %6\n\tv_not_b32 %7, %7\n\tv_not_b32 %8, %8\n\tv_not_b32 %9, %9\n\tv_not_b32 %10, %10\n\tv_not_b32 %11, %11\n\tv_not_b32 %12, %12\n\tv_not_b32 %13, %13\n\tv_not_b32 %14, %14\n\tv_not_b32 %15, %15"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_ashrrev_i32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_ashrrev_i32 %0, 3, %0\n\tv_ashrrev_i32 %1, 3, %1\n\tv_ashrrev_i32 %2, 3, %2\n\tv_ashrrev_i32 %3, 3, %3\n\tv_ashrrev_i32 %4, 3, %4\n\tv_ashrrev_i32 %5, 3, %5\n\tv_ashrrev_i32 %6, 3, %6\n\tv_ashrrev_i32 %7, 3, %7\n\tv_ashrrev_i32 %8, 3, %8\n\tv_ashrrev_i32 %9, 3, %9\n\tv_ashrrev_i32 %10, 3, %10\n\tv_ashrrev_i32 %11, 3, %11\n\tv_ashrrev_i32 %12, 3, %12\n\tv_ashrrev_i32 %13, 3, %13\n\tv_ashrrev_i32 %14, 3, %14\n\tv_ashrrev_i32 %15, 3, %15"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_lshrrev_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_lshrrev_b32 %0, 3, %0\n\tv_lshrrev_b32 %1, 3, %1\n\tv_lshrrev_b32 %2, 3, %2\n\tv_lshrrev_b32 %3, 3, %3\n\tv_lshrrev_b32 %4, 3, %4\n\tv_lshrrev_b32 %5, 3, %5\n\tv_lshrrev_b32 %6, 3, %6\n\tv_lshrrev_b32 %7, 3, %7\n\tv_lshrrev_b32 %8, 3, %8\n\tv_lshrrev_b32 %9, 3, %9\n\tv_lshrrev_b32 %10, 3, %10\n\tv_lshrrev_b32 %11, 3, %11\n\tv_lshrrev_b32 %12, 3, %12\n\tv_lshrrev_b32 %13, 3, %13\n\tv_lshrrev_b32 %14, 3, %14\n\tv_lshrrev_b32 %15, 3, %15"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_lshlrev_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3\n\tv_lshlrev_b32 %4, 1, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_lshlrev_b32 %6, 1, %6\n\tv_lshlrev_b32 %7, 1, %7\n\tv_lshlrev_b32 %8, 1, %8\n\tv_lshlrev_b32 %9, 1, %9\n\tv_lshlrev_b32 %10, 1, %10\n\tv_lshlrev_b32 %11, 1, %11\n\tv_lshlrev_b32 %12, 1, %12\n\tv_lshlrev_b32 %13, 1, %13\n\tv_lshlrev_b32 %14, 1, %14\n\tv_lshlrev_b32 %15, 1, %15"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_add_u32_e64(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_add_u32_e64 %0, %0, %16\n\tv_add_u32_e64 %1, %1, %16\n\tv_add_u32_e64 %2, %2, %16\n\tv_add_u32_e64 %3, %3, %16\n\tv_add_u32_e64 %4, %4, %16\n\tv_add_u32_e64 %5, %5, %16\n\tv_add_u32_e64 %6, %6, %16\n\tv_add_u32_e64 %7, %7, %16\n\tv_add_u32_e64 %8, %8, %16\n\tv_add_u32_e64 %9, %9, %16\n\tv_add_u32_e64 %10, %10, %16\n\tv_add_u32_e64 %11, %11, %16\n\tv_add_u32_e64 %12, %12, %16\n\tv_add_u32_e64 %13, %13, %16\n\tv_add_u32_e64 %14, %14, %16\n\tv_add_u32_e64 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_min_i32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_min_i32 %0, %0, %16\n\tv_min_i32 %1, %1, %16\n\tv_min_i32 %2, %2, %16\n\tv_min_i32 %3, %3, %16\n\tv_min_i32 %4, %4, %16\n\tv_min_i32 %5, %5, %16\n\tv_min_i32 %6, %6, %16\n\tv_min_i32 %7, %7, %16\n\tv_min_i32 %8, %8, %16\n\tv_min_i32 %9, %9, %16\n\tv_min_i32 %10, %10, %16\n\tv_min_i32 %11, %11, %16\n\tv_min_i32 %12, %12, %16\n\tv_min_i32 %13, %13, %16\n\tv_min_i32 %14, %14, %16\n\tv_min_i32 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mul_u32_u24(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_mul_u32_u24 %0, %0, %16\n\tv_mul_u32_u24 %1, %1, %16\n\tv_mul_u32_u24 %2, %2, %16\n\tv_mul_u32_u24 %3, %3, %16\n\tv_mul_u32_u24 %4, %4, %16\n\tv_mul_u32_u24 %5, %5, %16\n\tv_mul_u32_u24 %6, %6, %16\n\tv_mul_u32_u24 %7, %7, %16\n\tv_mul_u32_u24 %8, %8, %16\n\tv_mul_u32_u24 %9, %9, %16\n\tv_mul_u32_u24 %10, %10, %16\n\tv_mul_u32_u24 %11, %11, %16\n\tv_mul_u32_u24 %12, %12, %16\n\tv_mul_u32_u24 %13, %13, %16\n\tv_mul_u32_u24 %14, %14, %16\n\tv_mul_u32_u24 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma_f32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_fma_f32 %0, %0, %16, %17\n\tv_fma_f32 %1, %1, %16, %17\n\tv_fma_f32 %2, %2, %16, %17\n\tv_fma_f32 %3, %3, %16, %17\n\tv_fma_f32 %4, %4, %16, %17\n\tv_fma_f32 %5, %5, %16, %17\n\tv_fma_f32 %6, %6, %16, %17\n\tv_fma_f32 %7, %7, %16, %17\n\tv_fma_f32 %8, %8, %16, %17\n\tv_fma_f32 %9, %9, %16, %17\n\tv_fma_f32 %10, %10, %16, %17\n\tv_fma_f32 %11, %11, %16, %17\n\tv_fma_f32 %12, %12, %16, %17\n\tv_fma_f32 %13, %13, %16, %17\n\tv_fma_f32 %14, %14, %16, %17\n\tv_fma_f32 %15, %15, %16, %17"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fmac_f32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_fmac_f32 %0, %16, %17\n\tv_fmac_f32 %1, %16, %17\n\tv_fmac_f32 %2, %16, %17\n\tv_fmac_f32 %3, %16, %17\n\tv_fmac_f32 %4, %16, %17\n\tv_fmac_f32 %5, %16, %17\n\tv_fmac_f32 %6, %16, %17\n\tv_fmac_f32 %7, %16, %17\n\tv_fmac_f32 %8, %16, %17\n\tv_fmac_f32 %9, %16, %17\n\tv_fmac_f32 %10, %16, %17\n\tv_fmac_f32 %11, %16, %17\n\tv_fmac_f32 %12, %16, %17\n\tv_fmac_f32 %13, %16, %17\n\tv_fmac_f32 %14, %16, %17\n\tv_fmac_f32 %15, %16, %17"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_bitop3_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_bitop3_b32 %0, %0, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %1, %1, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %2, %2, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %3, %3, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %4, %4, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %5, %5, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %6, %6, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %7, %7, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %8, %8, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %9, %9, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %10, %10, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %11, %11, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %12, %12, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %13, %13, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %14, %14, %16, %17 bitop3:0x96\n\tv_bitop3_b32 %15, %15, %16, %17 bitop3:0x96"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_alignbit_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_alignbit_b32 %0, %16, %0, 3\n\tv_alignbit_b32 %1, %16, %1, 3\n\tv_alignbit_b32 %2, %16, %2, 3\n\tv_alignbit_b32 %3, %16, %3, 3\n\tv_alignbit_b32 %4, %16, %4, 3\n\tv_alignbit_b32 %5, %16, %5, 3\n\tv_alignbit_b32 %6, %16, %6, 3\n\tv_alignbit_b32 %7, %16, %7, 3\n\tv_alignbit_b32 %8, %16, %8, 3\n\tv_alignbit_b32 %9, %16, %9, 3\n\tv_alignbit_b32 %10, %16, %10, 3\n\tv_alignbit_b32 %11, %16, %11, 3\n\tv_alignbit_b32 %12, %16, %12, 3\n\tv_alignbit_b32 %13, %16, %13, 3\n\tv_alignbit_b32 %14, %16, %14, 3\n\tv_alignbit_b32 %15, %16, %15, 3"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_xad_u32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_xad_u32 %0, %0, %16, %17\n\tv_xad_u32 %1, %1, %16, %17\n\tv_xad_u32 %2, %2, %16, %17\n\tv_xad_u32 %3, %3, %16, %17\n\tv_xad_u32 %4, %4, %16, %17\n\tv_xad_u32 %5, %5, %16, %17\n\tv_xad_u32 %6, %6, %16, %17\n\tv_xad_u32 %7, %7, %16, %17\n\tv_xad_u32 %8, %8, %16, %17\n\tv_xad_u32 %9, %9, %16, %17\n\tv_xad_u32 %10, %10, %16, %17\n\tv_xad_u32 %11, %11, %16, %17\n\tv_xad_u32 %12, %12, %16, %17\n\tv_xad_u32 %13, %13, %16, %17\n\tv_xad_u32 %14, %14, %16, %17\n\tv_xad_u32 %15, %15, %16, %17"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_add3_u32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_add3_u32 %0, %0, %16, %17\n\tv_add3_u32 %1, %1, %16, %17\n\tv_add3_u32 %2, %2, %16, %17\n\tv_add3_u32 %3, %3, %16, %17\n\tv_add3_u32 %4, %4, %16, %17\n\tv_add3_u32 %5, %5, %16, %17\n\tv_add3_u32 %6, %6, %16, %17\n\tv_add3_u32 %7, %7, %16, %17\n\tv_add3_u32 %8, %8, %16, %17\n\tv_add3_u32 %9, %9, %16, %17\n\tv_add3_u32 %10, %10, %16, %17\n\tv_add3_u32 %11, %11, %16, %17\n\tv_add3_u32 %12, %12, %16, %17\n\tv_add3_u32 %13, %13, %16, %17\n\tv_add3_u32 %14, %14, %16, %17\n\tv_add3_u32 %15, %15, %16, %17"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_bfe_i32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_bfe_i32 %0, %0, 1, 30\n\tv_bfe_i32 %1, %1, 1, 30\n\tv_bfe_i32 %2, %2, 1, 30\n\tv_bfe_i32 %3, %3, 1, 30\n\tv_bfe_i32 %4, %4, 1, 30\n\tv_bfe_i32 %5, %5, 1, 30\n\tv_bfe_i32 %6, %6, 1, 30\n\tv_bfe_i32 %7, %7, 1, 30\n\tv_bfe_i32 %8, %8, 1, 30\n\tv_bfe_i32 %9, %9, 1, 30\n\tv_bfe_i32 %10, %10, 1, 30\n\tv_bfe_i32 %11, %11, 1, 30\n\tv_bfe_i32 %12, %12, 1, 30\n\tv_bfe_i32 %13, %13, 1, 30\n\tv_bfe_i32 %14, %14, 1, 30\n\tv_bfe_i32 %15, %15, 1, 30"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_and_or_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_and_or_b32 %0, %0, %16, %17\n\tv_and_or_b32 %1, %1, %16, %17\n\tv_and_or_b32 %2, %2, %16, %17\n\tv_and_or_b32 %3, %3, %16, %17\n\tv_and_or_b32 %4, %4, %16, %17\n\tv_and_or_b32 %5, %5, %16, %17\n\tv_and_or_b32 %6, %6, %16, %17\n\tv_and_or_b32 %7, %7, %16, %17\n\tv_and_or_b32 %8, %8, %16, %17\n\tv_and_or_b32 %9, %9, %16, %17\n\tv_and_or_b32 %10, %10, %16, %17\n\tv_and_or_b32 %11, %11, %16, %17\n\tv_and_or_b32 %12, %12, %16, %17\n\tv_and_or_b32 %13, %13, %16, %17\n\tv_and_or_b32 %14, %14, %16, %17\n\tv_and_or_b32 %15, %15, %16, %17"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_lshl_add_u32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_lshl_add_u32 %0, %0, 1, %16\n\tv_lshl_add_u32 %1, %1, 1, %16\n\tv_lshl_add_u32 %2, %2, 1, %16\n\tv_lshl_add_u32 %3, %3, 1, %16\n\tv_lshl_add_u32 %4, %4, 1, %16\n\tv_lshl_add_u32 %5, %5, 1, %16\n\tv_lshl_add_u32 %6, %6, 1, %16\n\tv_lshl_add_u32 %7, %7, 1, %16\n\tv_lshl_add_u32 %8, %8, 1, %16\n\tv_lshl_add_u32 %9, %9, 1, %16\n\tv_lshl_add_u32 %10, %10, 1, %16\n\tv_lshl_add_u32 %11, %11, 1, %16\n\tv_lshl_add_u32 %12, %12, 1, %16\n\tv_lshl_add_u32 %13, %13, 1, %16\n\tv_lshl_add_u32 %14, %14, 1, %16\n\tv_lshl_add_u32 %15, %15, 1, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_perm_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_perm_b32 %0, %0, %16, %17\n\tv_perm_b32 %1, %1, %16, %17\n\tv_perm_b32 %2, %2, %16, %17\n\tv_perm_b32 %3, %3, %16, %17\n\tv_perm_b32 %4, %4, %16, %17\n\tv_perm_b32 %5, %5, %16, %17\n\tv_perm_b32 %6, %6, %16, %17\n\tv_perm_b32 %7, %7, %16, %17\n\tv_perm_b32 %8, %8, %16, %17\n\tv_perm_b32 %9, %9, %16, %17\n\tv_perm_b32 %10, %10, %16, %17\n\tv_perm_b32 %11, %11, %16, %17\n\tv_perm_b32 %12, %12, %16, %17\n\tv_perm_b32 %13, %13, %16, %17\n\tv_perm_b32 %14, %14, %16, %17\n\tv_perm_b32 %15, %15, %16, %17"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mul_lo_u32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_mul_lo_u32 %0, %0, %16\n\tv_mul_lo_u32 %1, %1, %16\n\tv_mul_lo_u32 %2, %2, %16\n\tv_mul_lo_u32 %3, %3, %16\n\tv_mul_lo_u32 %4, %4, %16\n\tv_mul_lo_u32 %5, %5, %16\n\tv_mul_lo_u32 %6, %6, %16\n\tv_mul_lo_u32 %7, %7, %16\n\tv_mul_lo_u32 %8, %8, %16\n\tv_mul_lo_u32 %9, %9, %16\n\tv_mul_lo_u32 %10, %10, %16\n\tv_mul_lo_u32 %11, %11, %16\n\tv_mul_lo_u32 %12, %12, %16\n\tv_mul_lo_u32 %13, %13, %16\n\tv_mul_lo_u32 %14, %14, %16\n\tv_mul_lo_u32 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mad_u32_u24(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_mad_u32_u24 %0, %0, %16, %17\n\tv_mad_u32_u24 %1, %1, %16, %17\n\tv_mad_u32_u24 %2, %2, %16, %17\n\tv_mad_u32_u24 %3, %3, %16, %17\n\tv_mad_u32_u24 %4, %4, %16, %17\n\tv_mad_u32_u24 %5, %5, %16, %17\n\tv_mad_u32_u24 %6, %6, %16, %17\n\tv_mad_u32_u24 %7, %7, %16, %17\n\tv_mad_u32_u24 %8, %8, %16, %17\n\tv_mad_u32_u24 %9, %9, %16, %17\n\tv_mad_u32_u24 %10, %10, %16, %17\n\tv_mad_u32_u24 %11, %11, %16, %17\n\tv_mad_u32_u24 %12, %12, %16, %17\n\tv_mad_u32_u24 %13, %13, %16, %17\n\tv_mad_u32_u24 %14, %14, %16, %17\n\tv_mad_u32_u24 %15, %15, %16, %17"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cndmask_b32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_cndmask_b32 %0, %0, %16, vcc\n\tv_cndmask_b32 %1, %1, %16, vcc\n\tv_cndmask_b32 %2, %2, %16, vcc\n\tv_cndmask_b32 %3, %3, %16, vcc\n\tv_cndmask_b32 %4, %4, %16, vcc\n\tv_cndmask_b32 %5, %5, %16, vcc\n\tv_cndmask_b32 %6, %6, %16, vcc\n\tv_cndmask_b32 %7, %7, %16, vcc\n\tv_cndmask_b32 %8, %8, %16, vcc\n\tv_cndmask_b32 %9, %9, %16, vcc\n\tv_cndmask_b32 %10, %10, %16, vcc\n\tv_cndmask_b32 %11, %11, %16, vcc\n\tv_cndmask_b32 %12, %12, %16, vcc\n\tv_cndmask_b32 %13, %13, %16, vcc\n\tv_cndmask_b32 %14, %14, %16, vcc\n\tv_cndmask_b32 %15, %15, %16, vcc"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cndmask_e64(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_cndmask_b32 %0, %0, %16, s[10:11]\n\tv_cndmask_b32 %1, %1, %16, s[10:11]\n\tv_cndmask_b32 %2, %2, %16, s[10:11]\n\tv_cndmask_b32 %3, %3, %16, s[10:11]\n\tv_cndmask_b32 %4, %4, %16, s[10:11]\n\tv_cndmask_b32 %5, %5, %16, s[10:11]\n\tv_cndmask_b32 %6, %6, %16, s[10:11]\n\tv_cndmask_b32 %7, %7, %16, s[10:11]\n\tv_cndmask_b32 %8, %8, %16, s[10:11]\n\tv_cndmask_b32 %9, %9, %16, s[10:11]\n\tv_cndmask_b32 %10, %10, %16, s[10:11]\n\tv_cndmask_b32 %11, %11, %16, s[10:11]\n\tv_cndmask_b32 %12, %12, %16, s[10:11]\n\tv_cndmask_b32 %13, %13, %16, s[10:11]\n\tv_cndmask_b32 %14, %14, %16, s[10:11]\n\tv_cndmask_b32 %15, %15, %16, s[10:11]"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc", "s10", "s11");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_add_co_u32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_add_co_u32 %0, vcc, %0, %16\n\tv_add_co_u32 %1, vcc, %1, %16\n\tv_add_co_u32 %2, vcc, %2, %16\n\tv_add_co_u32 %3, vcc, %3, %16\n\tv_add_co_u32 %4, vcc, %4, %16\n\tv_add_co_u32 %5, vcc, %5, %16\n\tv_add_co_u32 %6, vcc, %6, %16\n\tv_add_co_u32 %7, vcc, %7, %16\n\tv_add_co_u32 %8, vcc, %8, %16\n\tv_add_co_u32 %9, vcc, %9, %16\n\tv_add_co_u32 %10, vcc, %10, %16\n\tv_add_co_u32 %11, vcc, %11, %16\n\tv_add_co_u32 %12, vcc, %12, %16\n\tv_add_co_u32 %13, vcc, %13, %16\n\tv_add_co_u32 %14, vcc, %14, %16\n\tv_add_co_u32 %15, vcc, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_addc_co_u32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_addc_co_u32 %0, vcc, %0, %16, vcc\n\tv_addc_co_u32 %1, vcc, %1, %16, vcc\n\tv_addc_co_u32 %2, vcc, %2, %16, vcc\n\tv_addc_co_u32 %3, vcc, %3, %16, vcc\n\tv_addc_co_u32 %4, vcc, %4, %16, vcc\n\tv_addc_co_u32 %5, vcc, %5, %16, vcc\n\tv_addc_co_u32 %6, vcc, %6, %16, vcc\n\tv_addc_co_u32 %7, vcc, %7, %16, vcc\n\tv_addc_co_u32 %8, vcc, %8, %16, vcc\n\tv_addc_co_u32 %9, vcc, %9, %16, vcc\n\tv_addc_co_u32 %10, vcc, %10, %16, vcc\n\tv_addc_co_u32 %11, vcc, %11, %16, vcc\n\tv_addc_co_u32 %12, vcc, %12, %16, vcc\n\tv_addc_co_u32 %13, vcc, %13, %16, vcc\n\tv_addc_co_u32 %14, vcc, %14, %16, vcc\n\tv_addc_co_u32 %15, vcc, %15, %16, vcc"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cmp_gt_i32(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_cmp_gt_i32 vcc, %0, %16\n\tv_cmp_gt_i32 vcc, %1, %16\n\tv_cmp_gt_i32 vcc, %2, %16\n\tv_cmp_gt_i32 vcc, %3, %16\n\tv_cmp_gt_i32 vcc, %4, %16\n\tv_cmp_gt_i32 vcc, %5, %16\n\tv_cmp_gt_i32 vcc, %6, %16\n\tv_cmp_gt_i32 vcc, %7, %16\n\tv_cmp_gt_i32 vcc, %8, %16\n\tv_cmp_gt_i32 vcc, %9, %16\n\tv_cmp_gt_i32 vcc, %10, %16\n\tv_cmp_gt_i32 vcc, %11, %16\n\tv_cmp_gt_i32 vcc, %12, %16\n\tv_cmp_gt_i32 vcc, %13, %16\n\tv_cmp_gt_i32 vcc, %14, %16\n\tv_cmp_gt_i32 vcc, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pk_add_u16(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_pk_add_u16 %0, %0, %16\n\tv_pk_add_u16 %1, %1, %16\n\tv_pk_add_u16 %2, %2, %16\n\tv_pk_add_u16 %3, %3, %16\n\tv_pk_add_u16 %4, %4, %16\n\tv_pk_add_u16 %5, %5, %16\n\tv_pk_add_u16 %6, %6, %16\n\tv_pk_add_u16 %7, %7, %16\n\tv_pk_add_u16 %8, %8, %16\n\tv_pk_add_u16 %9, %9, %16\n\tv_pk_add_u16 %10, %10, %16\n\tv_pk_add_u16 %11, %11, %16\n\tv_pk_add_u16 %12, %12, %16\n\tv_pk_add_u16 %13, %13, %16\n\tv_pk_add_u16 %14, %14, %16\n\tv_pk_add_u16 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pk_mad_i16(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_pk_mad_i16 %0, %0, %16, %17\n\tv_pk_mad_i16 %1, %1, %16, %17\n\tv_pk_mad_i16 %2, %2, %16, %17\n\tv_pk_mad_i16 %3, %3, %16, %17\n\tv_pk_mad_i16 %4, %4, %16, %17\n\tv_pk_mad_i16 %5, %5, %16, %17\n\tv_pk_mad_i16 %6, %6, %16, %17\n\tv_pk_mad_i16 %7, %7, %16, %17\n\tv_pk_mad_i16 %8, %8, %16, %17\n\tv_pk_mad_i16 %9, %9, %16, %17\n\tv_pk_mad_i16 %10, %10, %16, %17\n\tv_pk_mad_i16 %11, %11, %16, %17\n\tv_pk_mad_i16 %12, %12, %16, %17\n\tv_pk_mad_i16 %13, %13, %16, %17\n\tv_pk_mad_i16 %14, %14, %16, %17\n\tv_pk_mad_i16 %15, %15, %16, %17"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mov_dpp(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = threadIdx.x * 31 + i;
	uint32_t vb = b + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %12, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %13, %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %14, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint32_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_ashrrev_i64(uint32_t *out, uint32_t b, uint32_t c) {
	uint64_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i;
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_ashrrev_i64 %0, 3, %0\n\tv_ashrrev_i64 %1, 3, %1\n\tv_ashrrev_i64 %2, 3, %2\n\tv_ashrrev_i64 %3, 3, %3\n\tv_ashrrev_i64 %4, 3, %4\n\tv_ashrrev_i64 %5, 3, %5\n\tv_ashrrev_i64 %6, 3, %6\n\tv_ashrrev_i64 %7, 3, %7\n\tv_ashrrev_i64 %8, 3, %8\n\tv_ashrrev_i64 %9, 3, %9\n\tv_ashrrev_i64 %10, 3, %10\n\tv_ashrrev_i64 %11, 3, %11\n\tv_ashrrev_i64 %12, 3, %12\n\tv_ashrrev_i64 %13, 3, %13\n\tv_ashrrev_i64 %14, 3, %14\n\tv_ashrrev_i64 %15, 3, %15"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint64_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}
__global__ __launch_bounds__(256) void k_lshlrev_b64(uint32_t *out, uint32_t b, uint32_t c) {
	uint64_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i;
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_lshlrev_b64 %0, 3, %0\n\tv_lshlrev_b64 %1, 3, %1\n\tv_lshlrev_b64 %2, 3, %2\n\tv_lshlrev_b64 %3, 3, %3\n\tv_lshlrev_b64 %4, 3, %4\n\tv_lshlrev_b64 %5, 3, %5\n\tv_lshlrev_b64 %6, 3, %6\n\tv_lshlrev_b64 %7, 3, %7\n\tv_lshlrev_b64 %8, 3, %8\n\tv_lshlrev_b64 %9, 3, %9\n\tv_lshlrev_b64 %10, 3, %10\n\tv_lshlrev_b64 %11, 3, %11\n\tv_lshlrev_b64 %12, 3, %12\n\tv_lshlrev_b64 %13, 3, %13\n\tv_lshlrev_b64 %14, 3, %14\n\tv_lshlrev_b64 %15, 3, %15"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint64_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}
__global__ __launch_bounds__(256) void k_lshl_add_u64(uint32_t *out, uint32_t b, uint32_t c) {
	uint64_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i;
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_lshl_add_u64 %0, %0, 0, %16\n\tv_lshl_add_u64 %1, %1, 0, %16\n\tv_lshl_add_u64 %2, %2, 0, %16\n\tv_lshl_add_u64 %3, %3, 0, %16\n\tv_lshl_add_u64 %4, %4, 0, %16\n\tv_lshl_add_u64 %5, %5, 0, %16\n\tv_lshl_add_u64 %6, %6, 0, %16\n\tv_lshl_add_u64 %7, %7, 0, %16\n\tv_lshl_add_u64 %8, %8, 0, %16\n\tv_lshl_add_u64 %9, %9, 0, %16\n\tv_lshl_add_u64 %10, %10, 0, %16\n\tv_lshl_add_u64 %11, %11, 0, %16\n\tv_lshl_add_u64 %12, %12, 0, %16\n\tv_lshl_add_u64 %13, %13, 0, %16\n\tv_lshl_add_u64 %14, %14, 0, %16\n\tv_lshl_add_u64 %15, %15, 0, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint64_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}
__global__ __launch_bounds__(256) void k_mad_u64_u32(uint32_t *out, uint32_t b, uint32_t c) {
	uint64_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i;
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_mad_u64_u32 %0, vcc, %17, %17, %0\n\tv_mad_u64_u32 %1, vcc, %17, %17, %1\n\tv_mad_u64_u32 %2, vcc, %17, %17, %2\n\tv_mad_u64_u32 %3, vcc, %17, %17, %3\n\tv_mad_u64_u32 %4, vcc, %17, %17, %4\n\tv_mad_u64_u32 %5, vcc, %17, %17, %5\n\tv_mad_u64_u32 %6, vcc, %17, %17, %6\n\tv_mad_u64_u32 %7, vcc, %17, %17, %7\n\tv_mad_u64_u32 %8, vcc, %17, %17, %8\n\tv_mad_u64_u32 %9, vcc, %17, %17, %9\n\tv_mad_u64_u32 %10, vcc, %17, %17, %10\n\tv_mad_u64_u32 %11, vcc, %17, %17, %11\n\tv_mad_u64_u32 %12, vcc, %17, %17, %12\n\tv_mad_u64_u32 %13, vcc, %17, %17, %13\n\tv_mad_u64_u32 %14, vcc, %17, %17, %14\n\tv_mad_u64_u32 %15, vcc, %17, %17, %15"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint64_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}
__global__ __launch_bounds__(256) void k_mad_i64_i32(uint32_t *out, uint32_t b, uint32_t c) {
	uint64_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i;
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_mad_i64_i32 %0, vcc, %17, %17, %0\n\tv_mad_i64_i32 %1, vcc, %17, %17, %1\n\tv_mad_i64_i32 %2, vcc, %17, %17, %2\n\tv_mad_i64_i32 %3, vcc, %17, %17, %3\n\tv_mad_i64_i32 %4, vcc, %17, %17, %4\n\tv_mad_i64_i32 %5, vcc, %17, %17, %5\n\tv_mad_i64_i32 %6, vcc, %17, %17, %6\n\tv_mad_i64_i32 %7, vcc, %17, %17, %7\n\tv_mad_i64_i32 %8, vcc, %17, %17, %8\n\tv_mad_i64_i32 %9, vcc, %17, %17, %9\n\tv_mad_i64_i32 %10, vcc, %17, %17, %10\n\tv_mad_i64_i32 %11, vcc, %17, %17, %11\n\tv_mad_i64_i32 %12, vcc, %17, %17, %12\n\tv_mad_i64_i32 %13, vcc, %17, %17, %13\n\tv_mad_i64_i32 %14, vcc, %17, %17, %14\n\tv_mad_i64_i32 %15, vcc, %17, %17, %15"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint64_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}
__global__ __launch_bounds__(256) void k_add_f64(uint32_t *out, uint32_t b, uint32_t c) {
	uint64_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i;
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_add_f64 %0, %0, %16\n\tv_add_f64 %1, %1, %16\n\tv_add_f64 %2, %2, %16\n\tv_add_f64 %3, %3, %16\n\tv_add_f64 %4, %4, %16\n\tv_add_f64 %5, %5, %16\n\tv_add_f64 %6, %6, %16\n\tv_add_f64 %7, %7, %16\n\tv_add_f64 %8, %8, %16\n\tv_add_f64 %9, %9, %16\n\tv_add_f64 %10, %10, %16\n\tv_add_f64 %11, %11, %16\n\tv_add_f64 %12, %12, %16\n\tv_add_f64 %13, %13, %16\n\tv_add_f64 %14, %14, %16\n\tv_add_f64 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint64_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}
__global__ __launch_bounds__(256) void k_fma_f64(uint32_t *out, uint32_t b, uint32_t c) {
	uint64_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i;
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_fma_f64 %0, %0, %16, %16\n\tv_fma_f64 %1, %1, %16, %16\n\tv_fma_f64 %2, %2, %16, %16\n\tv_fma_f64 %3, %3, %16, %16\n\tv_fma_f64 %4, %4, %16, %16\n\tv_fma_f64 %5, %5, %16, %16\n\tv_fma_f64 %6, %6, %16, %16\n\tv_fma_f64 %7, %7, %16, %16\n\tv_fma_f64 %8, %8, %16, %16\n\tv_fma_f64 %9, %9, %16, %16\n\tv_fma_f64 %10, %10, %16, %16\n\tv_fma_f64 %11, %11, %16, %16\n\tv_fma_f64 %12, %12, %16, %16\n\tv_fma_f64 %13, %13, %16, %16\n\tv_fma_f64 %14, %14, %16, %16\n\tv_fma_f64 %15, %15, %16, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint64_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}
__global__ __launch_bounds__(256) void k_pk_add_f32(uint32_t *out, uint32_t b, uint32_t c) {
	uint64_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i;
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_pk_add_f32 %0, %0, %16\n\tv_pk_add_f32 %1, %1, %16\n\tv_pk_add_f32 %2, %2, %16\n\tv_pk_add_f32 %3, %3, %16\n\tv_pk_add_f32 %4, %4, %16\n\tv_pk_add_f32 %5, %5, %16\n\tv_pk_add_f32 %6, %6, %16\n\tv_pk_add_f32 %7, %7, %16\n\tv_pk_add_f32 %8, %8, %16\n\tv_pk_add_f32 %9, %9, %16\n\tv_pk_add_f32 %10, %10, %16\n\tv_pk_add_f32 %11, %11, %16\n\tv_pk_add_f32 %12, %12, %16\n\tv_pk_add_f32 %13, %13, %16\n\tv_pk_add_f32 %14, %14, %16\n\tv_pk_add_f32 %15, %15, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint64_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}
__global__ __launch_bounds__(256) void k_pk_fma_f32(uint32_t *out, uint32_t b, uint32_t c) {
	uint64_t a[kChains];
	for (int i = 0; i < kChains; i++) a[i] = ((uint64_t)threadIdx.x << 33) + i;
	uint64_t vb = ((uint64_t)b << 20) + threadIdx.x;
	uint32_t vc = c ^ threadIdx.x;
	for (int it = 0; it < kIters; it++)
		asm volatile("v_pk_fma_f32 %0, %0, %16, %16\n\tv_pk_fma_f32 %1, %1, %16, %16\n\tv_pk_fma_f32 %2, %2, %16, %16\n\tv_pk_fma_f32 %3, %3, %16, %16\n\tv_pk_fma_f32 %4, %4, %16, %16\n\tv_pk_fma_f32 %5, %5, %16, %16\n\tv_pk_fma_f32 %6, %6, %16, %16\n\tv_pk_fma_f32 %7, %7, %16, %16\n\tv_pk_fma_f32 %8, %8, %16, %16\n\tv_pk_fma_f32 %9, %9, %16, %16\n\tv_pk_fma_f32 %10, %10, %16, %16\n\tv_pk_fma_f32 %11, %11, %16, %16\n\tv_pk_fma_f32 %12, %12, %16, %16\n\tv_pk_fma_f32 %13, %13, %16, %16\n\tv_pk_fma_f32 %14, %14, %16, %16\n\tv_pk_fma_f32 %15, %15, %16, %16"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(vb), "v"(vc) : "vcc");
	uint64_t s = 0;
	for (int i = 0; i < kChains; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}
__global__ __launch_bounds__(256) void k_mad_bank_free(uint32_t *out, uint32_t b, uint32_t c) {
	// explicit registers: probes VGPR bank conflicts (bank = register % 4)
	asm volatile("v_mov_b32 v2, %0\n\tv_mov_b32 v3, %1\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %1\n\t"
		"v_mov_b32 v6, %0\n\tv_mov_b32 v7, %1\n\tv_mov_b32 v8, %0\n\tv_mov_b32 v9, %1"
		:: "v"(b + threadIdx.x), "v"(c ^ threadIdx.x) : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
	for (int it = 0; it < kIters; it++)
		asm volatile("v_mad_i64_i32 v[16:17], vcc, v2, v3, v[16:17]\n\tv_mad_i64_i32 v[18:19], vcc, v4, v5, v[18:19]\n\tv_mad_i64_i32 v[20:21], vcc, v2, v3, v[20:21]\n\tv_mad_i64_i32 v[22:23], vcc, v4, v5, v[22:23]\n\tv_mad_i64_i32 v[24:25], vcc, v2, v3, v[24:25]\n\tv_mad_i64_i32 v[26:27], vcc, v4, v5, v[26:27]\n\tv_mad_i64_i32 v[28:29], vcc, v2, v3, v[28:29]\n\tv_mad_i64_i32 v[30:31], vcc, v4, v5, v[30:31]\n\tv_mad_i64_i32 v[32:33], vcc, v2, v3, v[32:33]\n\tv_mad_i64_i32 v[34:35], vcc, v4, v5, v[34:35]\n\tv_mad_i64_i32 v[36:37], vcc, v2, v3, v[36:37]\n\tv_mad_i64_i32 v[38:39], vcc, v4, v5, v[38:39]\n\tv_mad_i64_i32 v[40:41], vcc, v2, v3, v[40:41]\n\tv_mad_i64_i32 v[42:43], vcc, v4, v5, v[42:43]\n\tv_mad_i64_i32 v[44:45], vcc, v2, v3, v[44:45]\n\tv_mad_i64_i32 v[46:47], vcc, v4, v5, v[46:47]" ::: "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
	uint32_t r;
	asm volatile("v_xor_b32 %0, v16, v47" : "=v"(r) :: "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
	out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_mad_bank_2x2(uint32_t *out, uint32_t b, uint32_t c) {
	// explicit registers: probes VGPR bank conflicts (bank = register % 4)
	asm volatile("v_mov_b32 v2, %0\n\tv_mov_b32 v3, %1\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %1\n\t"
		"v_mov_b32 v6, %0\n\tv_mov_b32 v7, %1\n\tv_mov_b32 v8, %0\n\tv_mov_b32 v9, %1"
		:: "v"(b + threadIdx.x), "v"(c ^ threadIdx.x) : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
	for (int it = 0; it < kIters; it++)
		asm volatile("v_mad_i64_i32 v[16:17], vcc, v4, v5, v[16:17]\n\tv_mad_i64_i32 v[18:19], vcc, v2, v3, v[18:19]\n\tv_mad_i64_i32 v[20:21], vcc, v4, v5, v[20:21]\n\tv_mad_i64_i32 v[22:23], vcc, v2, v3, v[22:23]\n\tv_mad_i64_i32 v[24:25], vcc, v4, v5, v[24:25]\n\tv_mad_i64_i32 v[26:27], vcc, v2, v3, v[26:27]\n\tv_mad_i64_i32 v[28:29], vcc, v4, v5, v[28:29]\n\tv_mad_i64_i32 v[30:31], vcc, v2, v3, v[30:31]\n\tv_mad_i64_i32 v[32:33], vcc, v4, v5, v[32:33]\n\tv_mad_i64_i32 v[34:35], vcc, v2, v3, v[34:35]\n\tv_mad_i64_i32 v[36:37], vcc, v4, v5, v[36:37]\n\tv_mad_i64_i32 v[38:39], vcc, v2, v3, v[38:39]\n\tv_mad_i64_i32 v[40:41], vcc, v4, v5, v[40:41]\n\tv_mad_i64_i32 v[42:43], vcc, v2, v3, v[42:43]\n\tv_mad_i64_i32 v[44:45], vcc, v4, v5, v[44:45]\n\tv_mad_i64_i32 v[46:47], vcc, v2, v3, v[46:47]" ::: "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
	uint32_t r;
	asm volatile("v_xor_b32 %0, v16, v47" : "=v"(r) :: "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
	out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_mad_bank_3same(uint32_t *out, uint32_t b, uint32_t c) {
	// explicit registers: probes VGPR bank conflicts (bank = register % 4)
	asm volatile("v_mov_b32 v2, %0\n\tv_mov_b32 v3, %1\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %1\n\t"
		"v_mov_b32 v6, %0\n\tv_mov_b32 v7, %1\n\tv_mov_b32 v8, %0\n\tv_mov_b32 v9, %1"
		:: "v"(b + threadIdx.x), "v"(c ^ threadIdx.x) : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
	for (int it = 0; it < kIters; it++)
		asm volatile("v_mad_i64_i32 v[16:17], vcc, v4, v8, v[16:17]\n\tv_mad_i64_i32 v[18:19], vcc, v2, v6, v[18:19]\n\tv_mad_i64_i32 v[20:21], vcc, v4, v8, v[20:21]\n\tv_mad_i64_i32 v[22:23], vcc, v2, v6, v[22:23]\n\tv_mad_i64_i32 v[24:25], vcc, v4, v8, v[24:25]\n\tv_mad_i64_i32 v[26:27], vcc, v2, v6, v[26:27]\n\tv_mad_i64_i32 v[28:29], vcc, v4, v8, v[28:29]\n\tv_mad_i64_i32 v[30:31], vcc, v2, v6, v[30:31]\n\tv_mad_i64_i32 v[32:33], vcc, v4, v8, v[32:33]\n\tv_mad_i64_i32 v[34:35], vcc, v2, v6, v[34:35]\n\tv_mad_i64_i32 v[36:37], vcc, v4, v8, v[36:37]\n\tv_mad_i64_i32 v[38:39], vcc, v2, v6, v[38:39]\n\tv_mad_i64_i32 v[40:41], vcc, v4, v8, v[40:41]\n\tv_mad_i64_i32 v[42:43], vcc, v2, v6, v[42:43]\n\tv_mad_i64_i32 v[44:45], vcc, v4, v8, v[44:45]\n\tv_mad_i64_i32 v[46:47], vcc, v2, v6, v[46:47]" ::: "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
	uint32_t r;
	asm volatile("v_xor_b32 %0, v16, v47" : "=v"(r) :: "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
	out[blockIdx.x * 256 + threadIdx.x] = r;
}
struct Case { const char *name; void (*fn)(uint32_t *, uint32_t, uint32_t); };

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const int blocks = cus * 8;		// 8 x 256 threads per CU: 8 waves/SIMD
	uint32_t *out;
	CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
	printf("device %s, %d CUs, clockRate %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
	std::vector<Case> cases = {
		{ "add_u32", k_add_u32 },
		{ "sub_u32", k_sub_u32 },
		{ "xor_b32", k_xor_b32 },
		{ "or_b32", k_or_b32 },
		{ "not_b32", k_not_b32 },
		{ "ashrrev_i32", k_ashrrev_i32 },
		{ "lshrrev_b32", k_lshrrev_b32 },
		{ "lshlrev_b32", k_lshlrev_b32 },
		{ "add_u32_e64", k_add_u32_e64 },
		{ "min_i32", k_min_i32 },
		{ "mul_u32_u24", k_mul_u32_u24 },
		{ "fma_f32", k_fma_f32 },
		{ "fmac_f32", k_fmac_f32 },
		{ "bitop3_b32", k_bitop3_b32 },
		{ "alignbit_b32", k_alignbit_b32 },
		{ "xad_u32", k_xad_u32 },
		{ "add3_u32", k_add3_u32 },
		{ "bfe_i32", k_bfe_i32 },
		{ "and_or_b32", k_and_or_b32 },
		{ "lshl_add_u32", k_lshl_add_u32 },
		{ "perm_b32", k_perm_b32 },
		{ "mul_lo_u32", k_mul_lo_u32 },
		{ "mad_u32_u24", k_mad_u32_u24 },
		{ "cndmask_b32", k_cndmask_b32 },
		{ "cndmask_e64", k_cndmask_e64 },
		{ "add_co_u32", k_add_co_u32 },
		{ "addc_co_u32", k_addc_co_u32 },
		{ "cmp_gt_i32", k_cmp_gt_i32 },
		{ "pk_add_u16", k_pk_add_u16 },
		{ "pk_mad_i16", k_pk_mad_i16 },
		{ "mov_dpp", k_mov_dpp },
		{ "ashrrev_i64", k_ashrrev_i64 },
		{ "lshlrev_b64", k_lshlrev_b64 },
		{ "lshl_add_u64", k_lshl_add_u64 },
		{ "mad_u64_u32", k_mad_u64_u32 },
		{ "mad_i64_i32", k_mad_i64_i32 },
		{ "mad_bank_free", k_mad_bank_free },
		{ "mad_bank_2x2", k_mad_bank_2x2 },
		{ "mad_bank_3same", k_mad_bank_3same },
		{ "add_f64", k_add_f64 },
		{ "fma_f64", k_fma_f64 },
		{ "pk_add_f32", k_pk_add_f32 },
		{ "pk_fma_f32", k_pk_fma_f32 },
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
		const double instr = (double)blocks * 4 * kIters * kChains;
		const double cyc = (best * 1e-3) * 2.4e9 * (cus * 4.0) / instr;
		printf("%-16s %8.3f ms  %7.2f T lane-ops/s  %5.2f cyc/wave-instr/SIMD @2.4GHz\n",
			c.name, best, tops, cyc);
	}
	return 0;
}
