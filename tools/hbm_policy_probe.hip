// hbm_policy_probe.hip -- cache-policy bits of the accesses in the 1R2W one-shot
// tile copy (the best streaming pattern, tools/hbm_probe2.hip): loads and
// stores with every combination of {plain, nt, sc1, sc0 sc1, nt sc1, nt sc0 sc1}.
// One load per lane, waited for inside the same asm statement (a load whose
// destination the compiler does not know to be in flight must not be left
// pending), then two stores.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_policy_probe hbm_policy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
	printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define LD(NAME, BITS) \
__device__ __forceinline__ u32x4 NAME(const u32x4 *p) { u32x4 v; \
	asm volatile("global_load_dwordx4 %0, %1, off " BITS "\n\ts_waitcnt vmcnt(0)" \
		: "=v"(v) : "v"(p) : "memory"); return v; }
#define ST(NAME, BITS) \
__device__ __forceinline__ void NAME(u32x4 *p, u32x4 v) { \
	asm volatile("global_store_dwordx4 %0, %1, off " BITS :: "v"(p), "v"(v) : "memory"); }
LD(ld0, "") LD(ld1, "nt") LD(ld2, "sc1") LD(ld3, "sc0 sc1") LD(ld4, "sc1 nt") LD(ld5, "sc0 sc1 nt")
ST(st0, "") ST(st1, "nt") ST(st2, "sc1") ST(st3, "sc0 sc1") ST(st4, "sc1 nt") ST(st5, "sc0 sc1 nt")

template <int L, int S>
__global__ __launch_bounds__(256) void tiles(const u32x4 *__restrict__ a, u32x4 *__restrict__ c,
		u32x4 *__restrict__ d, size_t nvec)
{
	const size_t t = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
	const size_t g = t * 256 + threadIdx.x;
	if (g >= nvec) return;
	u32x4 v;
	if (L == 0) v = ld0(&a[g]); else if (L == 1) v = ld1(&a[g]); else if (L == 2) v = ld2(&a[g]);
	else if (L == 3) v = ld3(&a[g]); else if (L == 4) v = ld4(&a[g]); else v = ld5(&a[g]);
	const u32x4 w = v + 1;
	if (S == 0) { st0(&c[g], v); st0(&d[g], w); } else if (S == 1) { st1(&c[g], v); st1(&d[g], w); }
	else if (S == 2) { st2(&c[g], v); st2(&d[g], w); } else if (S == 3) { st3(&c[g], v); st3(&d[g], w); }
	else if (S == 4) { st4(&c[g], v); st4(&d[g], w); } else { st5(&c[g], v); st5(&d[g], w); }
}

static const char *kName[6] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
static u32x4 *A, *C, *D;
static size_t NVEC;
static hipEvent_t E0, E1;

template <int L, int S> static int run()
{
	const unsigned blocks = (unsigned)(NVEC / 256);
	hipLaunchKernelGGL((tiles<L, S>), dim3(blocks), dim3(256), 0, 0, A, C, D, NVEC);
	CHECK(hipEventRecord(E0));
	for (int r = 0; r < 10; r++)
		hipLaunchKernelGGL((tiles<L, S>), dim3(blocks), dim3(256), 0, 0, A, C, D, NVEC);
	CHECK(hipEventRecord(E1));
	CHECK(hipEventSynchronize(E1));
	float ms; CHECK(hipEventElapsedTime(&ms, E0, E1)); ms /= 10;
	printf("load %-11s store %-11s %7.3f ms  %.3f of 8 TB/s\n", kName[L], kName[S], ms,
		48.0 * NVEC / (ms * 1e-3) / 8e12);
	return 0;
}
template <int L> static int row()
{
	return run<L, 0>() || run<L, 1>() || run<L, 2>() || run<L, 3>() || run<L, 4>() || run<L, 5>();
}

int main()
{
	NVEC = (size_t)1 << 28;		// 2^30 words per array
	CHECK(hipMalloc(&A, NVEC * 16)); CHECK(hipMalloc(&C, NVEC * 16)); CHECK(hipMalloc(&D, NVEC * 16));
	CHECK(hipMemset(A, 1, NVEC * 16));
	CHECK(hipEventCreate(&E0)); CHECK(hipEventCreate(&E1));
	printf("# 1R2W one-shot 4 KiB tiles over 3 x 4 GiB, 10 launches each\n");
	return row<0>() || row<1>() || row<2>() || row<3>() || row<4>() || row<5>();
}
