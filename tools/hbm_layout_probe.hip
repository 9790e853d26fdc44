// hbm_layout_probe.hip -- does the relative placement of the three streams of
// cordic_p2r_const (1 array read, 2 written, 4 GiB each) matter?  Same 1R2W
// kernel, arrays carved out of one allocation at base + k * (4 GiB + pad).
// Also: several passes of loads issued before their stores (BATCH).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int BATCH, int TPB = 1024>
__global__ __launch_bounds__(TPB) void kc(const u32x4 *__restrict__ a,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec)
{
	size_t chunk = (nvec + gridDim.x - 1) / gridDim.x;
	chunk = (chunk + TPB - 1) / TPB * TPB;
	const size_t lo = (size_t)blockIdx.x * chunk;
	const size_t hi = lo + chunk < nvec ? lo + chunk : nvec;
	for (size_t g = lo + threadIdx.x; g < hi; g += (size_t)TPB * BATCH) {
		u32x4 v[BATCH];
#pragma unroll
		for (int j = 0; j < BATCH; j++)
			if (g + (size_t)j * TPB < hi) v[j] = a[g + (size_t)j * TPB];
#pragma unroll
		for (int j = 0; j < BATCH; j++)
			if (g + (size_t)j * TPB < hi) {
				c[g + (size_t)j * TPB] = v[j] + 1u;
				d[g + (size_t)j * TPB] = v[j] ^ 5u;
			}
	}
}

template <int BATCH, int TPB = 1024>
int run(const char *tag, size_t pad, char *base, size_t nvec, int blocks)
{
	const size_t span = nvec * 16 + pad;
	const u32x4 *a = (const u32x4 *)base;
	u32x4 *c = (u32x4 *)(base + span), *d = (u32x4 *)(base + 2 * span);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((kc<BATCH, TPB>), dim3(blocks), dim3(TPB), 0, 0, a, c, d, nvec);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f, sum = 0;
	for (int rep = 0; rep < 8; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((kc<BATCH, TPB>), dim3(blocks), dim3(TPB), 0, 0, a, c, d, nvec);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
		sum += ms;
	}
	const double bytes = (double)nvec * 16.0 * 3;
	printf("%-8s pad %9zu B  blocks %4d  best %6.3f ms  avg %6.3f ms  %7.1f GB/s (%.3f of 8 TB/s)\n",
		tag, pad, blocks, best, sum / 8, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12);
	return 0;
}

int main()
{
	const size_t nvec = (size_t)1 << 28;
	char *base;
	CHECK(hipMalloc((void **)&base, 3 * (nvec * 16 + (64u << 20))));
	CHECK(hipMemset(base, 1, nvec * 16));
	const size_t pads[] = {0, 256, 4096, 65536, (1u << 20) + 4096, 2u << 20,
			37 * 4096, (16u << 20) + 12288, 33u << 20};
	for (size_t pad : pads)
		run<1>("batch1", pad, base, nvec, 512);
	// block shape at equal occupancy (32 waves per CU)
	for (int rep = 0; rep < 2; rep++) {
		run<1, 1024>("1024thr", 0, base, nvec, 512);
		run<1, 512>("512thr", 0, base, nvec, 1024);
		run<1, 256>("256thr", 0, base, nvec, 2048);
		run<1, 256>("256thr", 0, base, nvec, 4096);
		run<1, 1024>("1024thr", 0, base, nvec, 1024);
		run<1, 1024>("1024thr", 0, base, nvec, 2048);
	}
	for (size_t pad : {(size_t)0, (size_t)37 * 4096}) {
		run<2>("batch2", pad, base, nvec, 512);
		run<4>("batch4", pad, base, nvec, 512);
		run<1>("batch1", pad, base, nvec, 256);
		run<4>("batch4", pad, base, nvec, 256);
	}
	return 0;
}
