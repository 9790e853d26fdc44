// hbm_probe2.hip -- round-2 search for the streaming pattern that gets the
// most out of HBM3E for the CORDIC access patterns (no arithmetic):
//   1R1W (plain copy, the guide's 6.3 TB/s figure), 1R2W (cordic_p2r_const),
//   2R2W (cordic_r2p), 0R2W (cordic_nco).
// Axes: work distribution (grid-stride / one-shot tiles / contiguous chunk per
// persistent block), block -> address mapping (linear or XCD-contiguous: block b
// runs on XCD b%8, so tile (b%8)*T/8 + b/8 gives every XCD one contiguous
// eighth of each array), vectors in flight per lane (1, 2, 4), block size,
// and load / store cache policy (plain, nt, sc1 via inline asm).
//
//   ./hbm_probe2 [log2_samples=30] [reps=10]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum Pol { PLAIN = 0, NT = 1, SC1 = 2 };

template <int P> __device__ __forceinline__ u32x4 ld(const u32x4 *p)
{
	if constexpr (P == NT) return __builtin_nontemporal_load(p);
	else if constexpr (P == SC1) {
		u32x4 v;
		asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
		return v;
	} else return *p;
}
template <int P> __device__ __forceinline__ void st(u32x4 *p, u32x4 v)
{
	if constexpr (P == NT) __builtin_nontemporal_store(v, p);
	else if constexpr (P == SC1)
		asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
	else *p = v;
}

// One-shot tiles: block b owns tile t(b) of BS*U vectors; all U loads of a
// lane are issued before its stores.  XCD = 1: XCD-contiguous tile mapping.
template <int R, int W, int U, int BS, int LP, int SP, int XCD>
__global__ __launch_bounds__(BS) void tiles(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec)
{
	size_t t = blockIdx.x;
	if (XCD) {
		const size_t per = gridDim.x >> 3;
		t = (size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
	}
	const size_t base = t * (size_t)(BS * U) + threadIdx.x;
	u32x4 v[U], w[U];
#pragma unroll
	for (int u = 0; u < U; u++) {
		const size_t g = base + (size_t)u * BS;
		v[u] = u32x4{(uint32_t)g, 1, 2, 3};
		if (R >= 1) v[u] = ld<LP>(&a[g]);
		if (R >= 2) w[u] = ld<LP>(&b[g]);
	}
	if (SP == SC1 && R) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
	for (int u = 0; u < U; u++) {
		const size_t g = base + (size_t)u * BS;
		if (R >= 2) v[u] += w[u];
		if (W >= 1) st<SP>(&c[g], v[u]);
		if (W >= 2) st<SP>(&d[g], v[u] + 1);
	}
}

// Persistent blocks: MODE 0 grid-stride, 1 one contiguous chunk per block,
// 2 chunk per block with XCD-contiguous placement of the chunks.
template <int R, int W, int BS, int LP, int SP, int MODE>
__global__ __launch_bounds__(BS) void persist(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec)
{
	size_t lo, hi, stride;
	if (MODE == 0) {
		lo = (size_t)blockIdx.x * BS; hi = nvec; stride = (size_t)gridDim.x * BS;
	} else {
		size_t t = blockIdx.x;
		if (MODE == 2) t = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
		size_t chunk = (nvec + gridDim.x - 1) / gridDim.x;
		chunk = (chunk + BS - 1) / BS * BS;
		lo = t * chunk; hi = lo + chunk < nvec ? lo + chunk : nvec; stride = BS;
	}
	size_t g = lo + threadIdx.x;
	u32x4 nv{}, nw{};
	if (g < hi) { if (R >= 1) nv = ld<LP>(&a[g]); if (R >= 2) nw = ld<LP>(&b[g]); }
	for (; g < hi; g += stride) {
		u32x4 v = nv, w = nw;
		if (R == 0) v = u32x4{(uint32_t)g, 1, 2, 3};
		const size_t gn = g + stride;
		if (gn < hi) { if (R >= 1) nv = ld<LP>(&a[gn]); if (R >= 2) nw = ld<LP>(&b[gn]); }
		if (R >= 2) v += w;
		if (W >= 1) st<SP>(&c[g], v);
		if (W >= 2) st<SP>(&d[g], v + 1);
	}
}


// Persistent blocks, prefetch loop PEELED: hipcc puts `s_waitcnt vmcnt(0)` at
// the head of the prefetch loop of `persist` (the loop header merges the
// preheader state -- one load outstanding -- with the back-edge state -- load
// + stores outstanding -- and a single wait has to be safe for both), so every
// iteration also waits until HBM has acknowledged the previous iteration's
// stores.  With the first iteration peeled both predecessors of the loop header
// have the queue [load(s) of the next pass, stores of this pass] and the
// compiler's own wait becomes vmcnt(W): only the loads are waited for.
// DEPTH = passes the loads run ahead.
template <int R, int W, int BS, int DEPTH, int MODE>
__global__ __launch_bounds__(BS) void persist_peel(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec)
{
	size_t t = blockIdx.x;
	if (MODE == 2) t = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
	size_t chunk = (nvec + gridDim.x - 1) / gridDim.x;
	chunk = (chunk + BS - 1) / BS * BS;
	size_t lo = t * chunk, hi = lo + chunk < nvec ? lo + chunk : nvec, stride = BS;
	if (MODE == 0) { lo = (size_t)blockIdx.x * BS; hi = nvec; stride = (size_t)gridDim.x * BS; }
	size_t g = lo + threadIdx.x;
	if (g >= hi) return;
	const size_t last = g + (hi - 1 - g) / stride * stride;	// this lane's last vector
	u32x4 q[DEPTH][2];
#pragma unroll
	for (int k = 0; k < DEPTH; k++) {
		const size_t gk = g + k * stride <= last ? g + k * stride : last;
		if (R >= 1) q[k][0] = a[gk];
		if (R >= 2) q[k][1] = b[gk];
	}
	auto pass = [&](size_t gg) {
		u32x4 v = q[0][0], w = q[0][1];
		if (R == 0) v = u32x4{(uint32_t)gg, 1, 2, 3};
#pragma unroll
		for (int k = 0; k + 1 < DEPTH; k++) { q[k][0] = q[k + 1][0]; q[k][1] = q[k + 1][1]; }
		const size_t gn = gg + DEPTH * stride <= last ? gg + DEPTH * stride : last;
		if (R >= 1) q[DEPTH - 1][0] = a[gn];
		if (R >= 2) q[DEPTH - 1][1] = b[gn];
		if (R >= 2) v += w;
		if (W >= 1) c[gg] = v;
		if (W >= 2) d[gg] = v + 1;
	};
	pass(g);				// peeled
	for (g += stride; g < hi; g += stride)
		pass(g);
}

// Ping-pong: the loop is unrolled by two passes with two register sets, the
// loads of pass i+2 are issued as soon as set i has been consumed (before the
// stores of pass i), and the first double pass is peeled so that both
// predecessors of the loop header hold the same queue of outstanding
// operations -- the compiler's wait for set A then is vmcnt(R+2W+...) and
// never covers a store.
template <int R, int W, int BS, int MODE>
__global__ __launch_bounds__(BS) void persist_pp(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec)
{
	size_t t = blockIdx.x;
	if (MODE == 2) t = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
	size_t chunk = (nvec + gridDim.x - 1) / gridDim.x;
	chunk = (chunk + 2 * BS - 1) / (2 * BS) * (2 * BS);
	size_t lo = t * chunk, hi = lo + chunk < nvec ? lo + chunk : nvec;
	const size_t stride = BS;
	size_t g = lo + threadIdx.x;
	if (g >= hi) return;
	const size_t last = g + (hi - 1 - g) / stride * stride;
	auto clampi = [&](size_t i) { return i <= last ? i : last; };
	u32x4 A0{}, A1{}, B0{}, B1{};
	if (R >= 1) A0 = a[g];
	if (R >= 2) A1 = b[g];
	if (R >= 1) B0 = a[clampi(g + stride)];
	if (R >= 2) B1 = b[clampi(g + stride)];
	auto dpass = [&](size_t gg) {
		// pass A
		u32x4 v = A0, w = A1;
		if (R == 0) v = u32x4{(uint32_t)gg, 1, 2, 3};
		if (R >= 1) A0 = a[clampi(gg + 2 * stride)];
		if (R >= 2) A1 = b[clampi(gg + 2 * stride)];
		if (R >= 2) v += w;
		if (W >= 1) c[gg] = v;
		if (W >= 2) d[gg] = v + 1;
		// pass B
		const size_t gb = gg + stride;
		if (gb < hi) {
			u32x4 v2 = B0, w2 = B1;
			if (R == 0) v2 = u32x4{(uint32_t)gb, 1, 2, 3};
			if (R >= 1) B0 = a[clampi(gb + 2 * stride)];
			if (R >= 2) B1 = b[clampi(gb + 2 * stride)];
			if (R >= 2) v2 += w2;
			if (W >= 1) c[gb] = v2;
			if (W >= 2) d[gb] = v2 + 1;
		}
	};
	dpass(g);			// peeled
	for (g += 2 * stride; g < hi; g += 2 * stride)
		dpass(g);
}

// E1: K tiles per block in a (rolled) loop, no prefetch.  ORDER 0: the block's
// tiles are contiguous; ORDER 1: grid-stride.
template <int R, int W, int BS, int ORDER>
__global__ __launch_bounds__(BS) void tilesK(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec, int K)
{
	for (int k = 0; k < K; k++) {
		const size_t t = ORDER ? (size_t)k * gridDim.x + blockIdx.x : (size_t)blockIdx.x * K + k;
		const size_t g = t * BS + threadIdx.x;
		u32x4 v = u32x4{(uint32_t)g, 1, 2, 3};
		if (R >= 1) v = a[g];
		if (R >= 2) v += b[g];
		if (W >= 1) c[g] = v;
		if (W >= 2) d[g] = v + 1;
	}
}

// E2: persistent chunk per block, but every block starts its sweep at a
// different offset inside its chunk (and wraps), so that the blocks are not all
// at the same address modulo the chunk size at the same time.
template <int R, int W, int BS>
__global__ __launch_bounds__(BS) void persist_stagger(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec)
{
	const size_t passes = nvec / gridDim.x / BS;	// power of two sizes only
	const size_t lo = (size_t)blockIdx.x * passes * BS;
	const size_t rot = ((size_t)blockIdx.x * 2654435761u) % passes;
	for (size_t i = 0; i < passes; i++) {
		size_t p = i + rot; if (p >= passes) p -= passes;
		const size_t g = lo + p * BS + threadIdx.x;
		u32x4 v = u32x4{(uint32_t)g, 1, 2, 3};
		if (R >= 1) v = a[g];
		if (R >= 2) v += b[g];
		if (W >= 1) c[g] = v;
		if (W >= 2) d[g] = v + 1;
	}
}

// E3: persistent blocks that pull tiles from a global counter in address order
// (what the hardware dispatcher does for one-shot tiles), next tile fetched one
// pass ahead.  NCTR counters: counter j serves the j-th contiguous 1/NCTR of
// the tiles and is used by blocks with blockIdx % NCTR == j.
template <int R, int W, int BS, int NCTR>
__global__ __launch_bounds__(BS) void dyn(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
		u32x4 *__restrict__ c, u32x4 *__restrict__ d, size_t nvec, unsigned *ctr, int tiles_per_grab)
{
	__shared__ unsigned next[2];
	const unsigned j = blockIdx.x % NCTR;
	const unsigned per = (unsigned)(nvec / BS / NCTR);	// tiles per counter
	unsigned *my = ctr + j * 32;				// own cache line
	if (threadIdx.x == 0) next[0] = atomicAdd(my, (unsigned)tiles_per_grab);
	__syncthreads();
	int ph = 0;
	for (;;) {
		const unsigned t0 = next[ph];
		if (t0 >= per) break;
		if (threadIdx.x == 0) next[ph ^ 1] = atomicAdd(my, (unsigned)tiles_per_grab);
		for (int k = 0; k < tiles_per_grab; k++) {
			const size_t g = ((size_t)j * per + t0 + k) * BS + threadIdx.x;
			u32x4 v = u32x4{(uint32_t)g, 1, 2, 3};
			if (R >= 1) v = a[g];
			if (R >= 2) v += b[g];
			if (W >= 1) c[g] = v;
			if (W >= 2) d[g] = v + 1;
		}
		__syncthreads();
		ph ^= 1;
	}
}

static u32x4 *A, *B, *C, *D;
static size_t NVEC;
static int REPS;
static hipEvent_t E0, E1;

template <typename F> static void timeit(const char *name, int R, int W, F launch)
{
	launch(); launch();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(E0));
	for (int r = 0; r < REPS; r++) launch();
	CHECK(hipEventRecord(E1));
	CHECK(hipEventSynchronize(E1));
	float ms; CHECK(hipEventElapsedTime(&ms, E0, E1));
	ms /= REPS;
	const double bytes = (double)NVEC * 16.0 * (R + W);
	printf("%-58s %7.3f ms %7.1f GB/s %.3f\n", name, ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12);
	fflush(stdout);
}

template <int R, int W, int U, int BS, int LP, int SP, int XCD> static void run_tiles()
{
	char nm[128];
	static const char *pn[] = {"plain", "nt", "sc1"};
	snprintf(nm, sizeof nm, "%dR%dW tiles U%d bs%d ld=%s st=%s %s", R, W, U, BS, pn[LP], pn[SP], XCD ? "xcd" : "lin");
	const size_t blocks = NVEC / ((size_t)BS * U);
	timeit(nm, R, W, [&] { hipLaunchKernelGGL((tiles<R, W, U, BS, LP, SP, XCD>), dim3((unsigned)blocks), dim3(BS), 0, 0, A, B, C, D, NVEC); });
}
template <int R, int W, int BS, int LP, int SP, int MODE> static void run_persist(int blocks)
{
	char nm[128];
	static const char *pn[] = {"plain", "nt", "sc1"};
	static const char *mn[] = {"gridstride", "chunk", "chunk-xcd"};
	snprintf(nm, sizeof nm, "%dR%dW persist %s blocks%d bs%d ld=%s st=%s", R, W, mn[MODE], blocks, BS, pn[LP], pn[SP]);
	timeit(nm, R, W, [&] { hipLaunchKernelGGL((persist<R, W, BS, LP, SP, MODE>), dim3(blocks), dim3(BS), 0, 0, A, B, C, D, NVEC); });
}


template <int R, int W, int BS, int DEPTH, int MODE> static void run_pasm(int blocks)
{
	char nm[128];
	static const char *mn[] = {"gridstride", "chunk", "chunk-xcd"};
	snprintf(nm, sizeof nm, "%dR%dW persist-peel depth%d %s blocks%d bs%d", R, W, DEPTH, mn[MODE], blocks, BS);
	timeit(nm, R, W, [&] { hipLaunchKernelGGL((persist_peel<R, W, BS, DEPTH, MODE>), dim3(blocks), dim3(BS), 0, 0, A, B, C, D, NVEC); });
}

template <int R, int W, int BS, int MODE> static void run_pp(int blocks)
{
	char nm[128];
	static const char *mn[] = {"gridstride", "chunk", "chunk-xcd"};
	snprintf(nm, sizeof nm, "%dR%dW persist-pingpong %s blocks%d bs%d", R, W, mn[MODE], blocks, BS);
	timeit(nm, R, W, [&] { hipLaunchKernelGGL((persist_pp<R, W, BS, MODE>), dim3(blocks), dim3(BS), 0, 0, A, B, C, D, NVEC); });
}
template <int R, int W> static void sweep_asm()
{
	run_pp<R, W, 256, 1>(512);
	run_pp<R, W, 256, 1>(2048);
	run_pp<R, W, 256, 2>(2048);
	run_pp<R, W, 1024, 1>(512);
	run_pp<R, W, 1024, 2>(512);
	run_pasm<R, W, 256, 1, 1>(512);
	run_pasm<R, W, 256, 1, 1>(2048);
	run_pasm<R, W, 256, 1, 2>(2048);
	run_pasm<R, W, 256, 2, 1>(2048);
	run_pasm<R, W, 256, 2, 2>(2048);
	run_pasm<R, W, 256, 4, 2>(2048);
	run_pasm<R, W, 256, 1, 0>(2048);
	run_pasm<R, W, 256, 2, 0>(2048);
	run_pasm<R, W, 1024, 1, 1>(512);
	run_pasm<R, W, 1024, 1, 2>(512);
	run_pasm<R, W, 1024, 2, 1>(512);
	run_pasm<R, W, 1024, 2, 2>(512);
	run_pasm<R, W, 1024, 4, 2>(512);
	run_pasm<R, W, 1024, 1, 2>(256);
	run_pasm<R, W, 1024, 2, 2>(256);
	run_tiles<R, W, 1, 256, PLAIN, PLAIN, 0>();
	run_tiles<R, W, 1, 256, PLAIN, PLAIN, 1>();
	run_tiles<R, W, 1, 1024, PLAIN, PLAIN, 1>();
}


template <int R, int W, int BS, int ORDER> static void run_tilesK(int K)
{
	char nm[128];
	snprintf(nm, sizeof nm, "%dR%dW tilesK K%d bs%d %s", R, W, K, BS, ORDER ? "gridstride" : "contig");
	const size_t blocks = NVEC / ((size_t)BS * K);
	timeit(nm, R, W, [&] { hipLaunchKernelGGL((tilesK<R, W, BS, ORDER>), dim3((unsigned)blocks), dim3(BS), 0, 0, A, B, C, D, NVEC, K); });
}
template <int R, int W, int BS> static void run_stagger(int blocks)
{
	char nm[128];
	snprintf(nm, sizeof nm, "%dR%dW persist-stagger blocks%d bs%d", R, W, blocks, BS);
	timeit(nm, R, W, [&] { hipLaunchKernelGGL((persist_stagger<R, W, BS>), dim3(blocks), dim3(BS), 0, 0, A, B, C, D, NVEC); });
}
static unsigned *CTR;
template <int R, int W, int BS, int NCTR> static void run_dyn(int blocks, int grab)
{
	char nm[128];
	snprintf(nm, sizeof nm, "%dR%dW dyn ctr%d blocks%d bs%d grab%d", R, W, NCTR, blocks, BS, grab);
	timeit(nm, R, W, [&] {
		CHECK(hipMemsetAsync(CTR, 0, 4096, 0));
		hipLaunchKernelGGL((dyn<R, W, BS, NCTR>), dim3(blocks), dim3(BS), 0, 0, A, B, C, D, NVEC, CTR, grab); });
}
template <int R, int W> static void sweep_e()
{
	for (int K : {1, 2, 4, 8, 16, 64}) run_tilesK<R, W, 256, 0>(K);
	for (int K : {2, 8, 64}) run_tilesK<R, W, 256, 1>(K);
	for (int K : {1, 2, 4, 8, 16, 64}) run_tilesK<R, W, 1024, 0>(K);
	for (int K : {8, 64}) run_tilesK<R, W, 1024, 1>(K);
	run_stagger<R, W, 256>(2048);
	run_stagger<R, W, 1024>(512);
	run_dyn<R, W, 1024, 1>(512, 1);
	run_dyn<R, W, 1024, 8>(512, 1);
	run_dyn<R, W, 1024, 8>(512, 2);
	run_dyn<R, W, 1024, 8>(512, 4);
	run_dyn<R, W, 1024, 1>(512, 4);
	run_dyn<R, W, 256, 8>(2048, 1);
	run_dyn<R, W, 256, 8>(2048, 4);
	run_dyn<R, W, 256, 8>(2048, 16);
	run_dyn<R, W, 256, 1>(2048, 16);
}

template <int R, int W> static void sweep()
{
	// one-shot tiles
	run_tiles<R, W, 1, 256, PLAIN, PLAIN, 0>();
	run_tiles<R, W, 1, 256, PLAIN, PLAIN, 1>();
	run_tiles<R, W, 2, 256, PLAIN, PLAIN, 0>();
	run_tiles<R, W, 2, 256, PLAIN, PLAIN, 1>();
	run_tiles<R, W, 4, 256, PLAIN, PLAIN, 0>();
	run_tiles<R, W, 4, 256, PLAIN, PLAIN, 1>();
	run_tiles<R, W, 4, 256, NT, NT, 0>();
	run_tiles<R, W, 4, 256, NT, NT, 1>();
	run_tiles<R, W, 4, 256, PLAIN, NT, 1>();
	run_tiles<R, W, 4, 256, NT, PLAIN, 1>();
	run_tiles<R, W, 4, 256, PLAIN, SC1, 1>();
	run_tiles<R, W, 2, 512, PLAIN, PLAIN, 1>();
	run_tiles<R, W, 2, 1024, PLAIN, PLAIN, 1>();
	run_tiles<R, W, 4, 1024, PLAIN, PLAIN, 1>();
	run_tiles<R, W, 8, 256, PLAIN, PLAIN, 1>();
	run_tiles<R, W, 8, 256, NT, NT, 1>();
	// persistent
	run_persist<R, W, 256, PLAIN, PLAIN, 0>(2048);
	run_persist<R, W, 256, PLAIN, PLAIN, 0>(8192);
	run_persist<R, W, 1024, PLAIN, PLAIN, 1>(512);
	run_persist<R, W, 1024, PLAIN, PLAIN, 2>(512);
	run_persist<R, W, 1024, NT, NT, 2>(512);
	run_persist<R, W, 1024, PLAIN, PLAIN, 2>(2048);
	run_persist<R, W, 256, PLAIN, PLAIN, 2>(2048);
	run_persist<R, W, 256, PLAIN, PLAIN, 2>(8192);
	run_persist<R, W, 256, NT, NT, 2>(8192);
}

int main(int argc, char **argv)
{
	const int lg = argc > 1 ? atoi(argv[1]) : 30;
	REPS = argc > 2 ? atoi(argv[2]) : 10;
	NVEC = (size_t)1 << (lg - 2);
	CHECK(hipMalloc(&A, NVEC * 16)); CHECK(hipMalloc(&B, NVEC * 16));
	CHECK(hipMalloc(&C, NVEC * 16)); CHECK(hipMalloc(&D, NVEC * 16));
	CHECK(hipMemset(A, 1, NVEC * 16)); CHECK(hipMemset(B, 2, NVEC * 16));
	CHECK(hipEventCreate(&E0)); CHECK(hipEventCreate(&E1));
	printf("# 2^%d samples (4 B each) per array; GB/s counts R+W algorithmic bytes; last column = fraction of 8 TB/s\n", lg);
	printf("# state marker (round-1 reference pattern):\n");
	run_persist<1, 2, 256, PLAIN, PLAIN, 1>(512);
	if (argc > 3 && argv[3][0] == 'm') return 0;	// "marker": state probe only
	if (argc > 3 && argv[3][0] == 'e') {		// experiments E1-E3
		CHECK(hipMalloc(&CTR, 4096));
		sweep_e<1, 2>(); sweep_e<0, 2>(); sweep_e<2, 2>();
		run_persist<1, 2, 256, PLAIN, PLAIN, 1>(512);
		return 0;
	}
	if (argc > 3 && argv[3][0] == 'a') {		// "asm": explicit-waitcnt variants only
		sweep_asm<1, 2>(); sweep_asm<2, 2>(); sweep_asm<0, 2>(); sweep_asm<1, 1>();
		run_persist<1, 2, 256, PLAIN, PLAIN, 1>(512);
		return 0;
	}
	sweep<1, 1>();
	sweep<1, 2>();
	printf("# state marker:\n");
	run_persist<1, 2, 256, PLAIN, PLAIN, 1>(512);
	sweep<2, 2>();
	sweep<0, 2>();
	printf("# state marker:\n");
	run_persist<1, 2, 256, PLAIN, PLAIN, 1>(512);
	// hipMemcpy DtoD as the runtime's own idea of a copy
	timeit("1R1W hipMemcpyAsync D2D", 1, 1, [&] { CHECK(hipMemcpyAsync(C, A, NVEC * 16, hipMemcpyDeviceToDevice, 0)); });
	return 0;
}
