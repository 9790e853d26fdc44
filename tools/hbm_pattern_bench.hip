// hbm_pattern_bench.hip -- what HBM delivers for the CORDIC access patterns
// with (almost) no arithmetic: per sample R bytes read and W bytes written,
// 16 bytes per lane per array, grid-stride over 2^30 samples.
//   pattern 1R2W = cordic_p2r_const (4 B in, 8 B out)
//   pattern 2R2W = cordic_r2p       (8 B in, 8 B out)
//   pattern 0R2W = cordic_nco       (8 B out)
//   pattern 1R1W = plain copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec)
{
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < nvec; g += stride) {
		u32x4 v = {(uint32_t)g, 1, 2, 3};
		if (R >= 1) v = a[g];
		if (R >= 2) v += b[g];
		if (W >= 1) { if (NT) __builtin_nontemporal_store(v, &c[g]); else c[g] = v; }
		if (W >= 2) { u32x4 w = v + 1; if (NT) __builtin_nontemporal_store(w, &d[g]); else d[g] = w; }
	}
}

// persistent blocks, each sweeping its own contiguous chunk
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void kc(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec)
{
	const size_t chunk = (nvec + gridDim.x - 1) / gridDim.x;
	const size_t lo = (size_t)blockIdx.x * chunk;
	const size_t hi = lo + chunk < nvec ? lo + chunk : nvec;
	for (size_t g = lo + threadIdx.x; g < hi; g += 256) {
		u32x4 v = {(uint32_t)g, 1, 2, 3};
		if (R >= 1) v = a[g];
		if (R >= 2) v += b[g];
		if (W >= 1) { if (NT) __builtin_nontemporal_store(v, &c[g]); else c[g] = v; }
		if (W >= 2) { u32x4 w = v + 1; if (NT) __builtin_nontemporal_store(w, &d[g]); else d[g] = w; }
	}
}

template <int R, int W, bool NT>
int runc(const char *name, u32x4 *a, u32x4 *b, u32x4 *c, u32x4 *d, size_t nvec, int blocks)
{
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((kc<R, W, NT>), dim3(blocks), dim3(256), 0, 0, a, b, c, d, nvec);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 8; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((kc<R, W, NT>), dim3(blocks), dim3(256), 0, 0, a, b, c, d, nvec);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double bytes = (double)nvec * 16.0 * (R + W);
	printf("%-10s chunked blocks %5d  %7.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, blocks, best,
		bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12);
	return 0;
}

template <int R, int W, bool NT>
int run(const char *name, u32x4 *a, u32x4 *b, u32x4 *c, u32x4 *d, size_t nvec, int blocks)
{
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k<R, W, NT>), dim3(blocks), dim3(256), 0, 0, a, b, c, d, nvec);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 8; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((k<R, W, NT>), dim3(blocks), dim3(256), 0, 0, a, b, c, d, nvec);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double bytes = (double)nvec * 16.0 * (R + W);
	printf("%-10s blocks %5d  %7.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, blocks, best,
		bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12);
	return 0;
}

int main()
{
	const size_t nvec = (size_t)1 << 28;	// 2^30 samples of 4 B
	u32x4 *a, *b, *c, *d;
	CHECK(hipMalloc(&a, nvec * 16)); CHECK(hipMalloc(&b, nvec * 16));
	CHECK(hipMalloc(&c, nvec * 16)); CHECK(hipMalloc(&d, nvec * 16));
	CHECK(hipMemset(a, 1, nvec * 16)); CHECK(hipMemset(b, 2, nvec * 16));
	for (int blocks : {2048, 8192, 65536}) {
		run<1, 1, false>("1R1W", a, b, c, d, nvec, blocks);
		run<1, 2, false>("1R2W", a, b, c, d, nvec, blocks);
		run<1, 2, true>("1R2W nt", a, b, c, d, nvec, blocks);
		run<2, 2, false>("2R2W", a, b, c, d, nvec, blocks);
		run<2, 2, true>("2R2W nt", a, b, c, d, nvec, blocks);
		run<0, 2, false>("0R2W", a, b, c, d, nvec, blocks);
		run<0, 2, true>("0R2W nt", a, b, c, d, nvec, blocks);
	}
	for (int blocks : {512, 1024, 2048, 4096}) {
		runc<1, 2, false>("1R2W", a, b, c, d, nvec, blocks);
		runc<1, 2, true>("1R2W nt", a, b, c, d, nvec, blocks);
		runc<2, 2, true>("2R2W nt", a, b, c, d, nvec, blocks);
		runc<0, 2, true>("0R2W nt", a, b, c, d, nvec, blocks);
	}
	return 0;
}
