// hbm_vmm_probe.cpp -- does the 0.75 <-> 0.82 spread of a 1R2W stream over
// different triples of 4 GiB arrays (profiles/r02/hbm_placement.txt) follow
// the PHYSICAL memory, the VIRTUAL addresses, or neither?
//
// hipMalloc hides both.  The virtual-memory API separates them: physical
// handles (hipMemCreate) and address ranges (hipMemAddressReserve) are created
// independently and mapped onto each other at will (hipMemMap), so the same
// physical triple can be timed at different virtual addresses and different
// physical triples at the same ones.
//
//   A  five 4 GiB physical handles, three fixed address ranges: every ordered
//      (in, out0, out1) triple out of the five -> time per physical choice
//   B  the best and the worst physical triple of A, mapped at other address
//      ranges (six reserved; three disjoint placements + a role rotation)
//   C  ONE 12 GiB physical handle mapped contiguously, arrays carved at
//      0 / 4 / 8 GiB (and role-rotated); repeated with fresh slabs
//   D  arrays striped over 1 GiB physical pieces taken round-robin from one
//      pool (array k = pieces k, k+3, k+6, k+9)
//
// Timing: tools/libhbmprobe.so hbm_probe(), 1R2W, one-shot 4 KiB tiles with
// non-temporal accesses (mode 2), 10 launches after a warm-up launch.
//
//   g++ ... tools/hbm_vmm_probe.cpp -L tools -lhbmprobe   (tools/Makefile)
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" float hbm_probe(const void *in0, const void *in1, void *out0, void *out1,
		size_t nwords, int R, int W, int mode, int reps, void *stream);

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
	exit(1); } } while (0)

static const size_t GiB = (size_t)1 << 30;
static const size_t ARR = 4 * GiB;
static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_acc;

static hipMemGenericAllocationHandle_t phys(size_t bytes)
{
	hipMemGenericAllocationHandle_t h;
	OK(hipMemCreate(&h, bytes, &g_prop, 0));
	return h;
}
static void *reserve(size_t bytes)
{
	void *p = nullptr;
	OK(hipMemAddressReserve(&p, bytes, 0, nullptr, 0));
	return p;
}
static void map(void *va, size_t bytes, hipMemGenericAllocationHandle_t h, size_t off = 0)
{
	OK(hipMemMap(va, bytes, off, h, 0));
	OK(hipMemSetAccess(va, bytes, &g_acc, 1));
}
static void unmap(void *va, size_t bytes) { OK(hipMemUnmap(va, bytes)); }

static double frac(float ms) { return 12.0 * (double)(ARR / 4) / (ms * 1e-3) / 8e12; }
static float t1r2w(void *in, void *o0, void *o1)
{
	const float ms = hbm_probe(in, nullptr, o0, o1, ARR / 4, 1, 2, 2, 10, nullptr);
	if (ms <= 0) { fprintf(stderr, "probe failed\n"); exit(1); }
	return ms;
}

int main()
{
	OK(hipSetDevice(0));
	g_prop = hipMemAllocationProp{};
	g_prop.type = hipMemAllocationTypePinned;
	g_prop.location.type = hipMemLocationTypeDevice;
	g_prop.location.id = 0;
	g_acc.location = g_prop.location;
	g_acc.flags = hipMemAccessFlagsProtReadWrite;
	size_t gmin = 0, grec = 0;
	OK(hipMemGetAllocationGranularity(&gmin, &g_prop, hipMemAllocationGranularityMinimum));
	OK(hipMemGetAllocationGranularity(&grec, &g_prop, hipMemAllocationGranularityRecommended));
	printf("# allocation granularity: minimum %zu B, recommended %zu B\n", gmin, grec);

	// ---------------------------------------------------------------- A
	const int NP = 5;
	hipMemGenericAllocationHandle_t P[NP];
	for (int i = 0; i < NP; i++) P[i] = phys(ARR);
	void *V[6];
	for (int i = 0; i < 6; i++) V[i] = reserve(ARR);
	printf("# A: physical handles P0..P4 (4 GiB each) at FIXED addresses V0 V1 V2 = %p %p %p\n",
		V[0], V[1], V[2]);
	struct Row { int a, b, c; float ms; };
	std::vector<Row> rows;
	for (int a = 0; a < NP; a++)
		for (int b = 0; b < NP; b++)
			for (int c = b + 1; c < NP; c++) {	// out0 / out1 unordered
				if (a == b || a == c) continue;
				map(V[0], ARR, P[a]); map(V[1], ARR, P[b]); map(V[2], ARR, P[c]);
				const float ms = t1r2w(V[0], V[1], V[2]);
				unmap(V[0], ARR); unmap(V[1], ARR); unmap(V[2], ARR);
				rows.push_back(Row{a, b, c, ms});
				printf("A in=P%d out=P%d,P%d  %.3f ms  %.3f\n", a, b, c, ms, frac(ms));
			}
	std::sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.ms < y.ms; });
	const Row best = rows.front(), worst = rows.back();
	printf("# A: best in=P%d out=P%d,P%d %.3f | worst in=P%d out=P%d,P%d %.3f\n",
		best.a, best.b, best.c, frac(best.ms), worst.a, worst.b, worst.c, frac(worst.ms));
	// per written pair: mean over the inputs
	for (int b = 0; b < NP; b++)
		for (int c = b + 1; c < NP; c++) {
			double s = 0; int n = 0;
			for (const Row &r : rows) if (r.b == b && r.c == c) { s += frac(r.ms); n++; }
			printf("A pair out=P%d,P%d mean %.3f over %d inputs\n", b, c, s / n, n);
		}
	// repeat of best / worst: is a physical triple's time stable?
	for (int rep = 0; rep < 2; rep++)
		for (const Row &r : {best, worst}) {
			map(V[0], ARR, P[r.a]); map(V[1], ARR, P[r.b]); map(V[2], ARR, P[r.c]);
			printf("A again in=P%d out=P%d,P%d  %.3f\n", r.a, r.b, r.c, frac(t1r2w(V[0], V[1], V[2])));
			unmap(V[0], ARR); unmap(V[1], ARR); unmap(V[2], ARR);
		}

	// ---------------------------------------------------------------- B
	printf("# B: the same physical triples at OTHER virtual addresses (V3 V4 V5 = %p %p %p, and mixed)\n",
		V[3], V[4], V[5]);
	const int vsets[4][3] = {{0, 1, 2}, {3, 4, 5}, {5, 3, 1}, {2, 4, 0}};
	for (const Row &r : {best, worst})
		for (const auto &vs : vsets) {
			map(V[vs[0]], ARR, P[r.a]); map(V[vs[1]], ARR, P[r.b]); map(V[vs[2]], ARR, P[r.c]);
			printf("B in=P%d out=P%d,P%d at V%d V%d V%d  %.3f\n", r.a, r.b, r.c,
				vs[0], vs[1], vs[2], frac(t1r2w(V[vs[0]], V[vs[1]], V[vs[2]])));
			unmap(V[vs[0]], ARR); unmap(V[vs[1]], ARR); unmap(V[vs[2]], ARR);
		}
	for (int i = 0; i < NP; i++) OK(hipMemRelease(P[i]));

	// ---------------------------------------------------------------- C
	printf("# C: ONE 12 GiB physical handle, arrays at 0 / 4 / 8 GiB of it\n");
	void *S = reserve(3 * ARR);
	for (int slab = 0; slab < 4; slab++) {
		hipMemGenericAllocationHandle_t h = phys(3 * ARR);
		map(S, 3 * ARR, h);
		char *b = static_cast<char *>(S);
		printf("C slab %d roles (0,1,2) %.3f  (1,0,2) %.3f  (2,0,1) %.3f\n", slab,
			frac(t1r2w(b, b + ARR, b + 2 * ARR)),
			frac(t1r2w(b + ARR, b, b + 2 * ARR)),
			frac(t1r2w(b + 2 * ARR, b, b + ARR)));
		unmap(S, 3 * ARR);
		OK(hipMemRelease(h));
	}

	// ---------------------------------------------------------------- D
	printf("# D: arrays striped over 1 GiB physical pieces, round-robin from one pool of 12\n");
	for (int rep = 0; rep < 2; rep++) {
		hipMemGenericAllocationHandle_t piece[12];
		for (int k = 0; k < 12; k++) piece[k] = phys(GiB);
		char *b = static_cast<char *>(S);
		for (int arr = 0; arr < 3; arr++)
			for (int j = 0; j < 4; j++)
				map(b + arr * ARR + j * GiB, GiB, piece[arr + 3 * j]);
		printf("D rep %d striped roles (0,1,2) %.3f (1,0,2) %.3f (2,0,1) %.3f\n", rep,
			frac(t1r2w(b, b + ARR, b + 2 * ARR)),
			frac(t1r2w(b + ARR, b, b + 2 * ARR)),
			frac(t1r2w(b + 2 * ARR, b, b + ARR)));
		unmap(S, 3 * ARR);
		// the same pieces, contiguous per array
		for (int arr = 0; arr < 3; arr++)
			for (int j = 0; j < 4; j++)
				map(b + arr * ARR + j * GiB, GiB, piece[4 * arr + j]);
		printf("D rep %d contiguous pieces (0,1,2) %.3f (1,0,2) %.3f (2,0,1) %.3f\n", rep,
			frac(t1r2w(b, b + ARR, b + 2 * ARR)),
			frac(t1r2w(b + ARR, b, b + 2 * ARR)),
			frac(t1r2w(b + 2 * ARR, b, b + ARR)));
		unmap(S, 3 * ARR);
		for (int k = 0; k < 12; k++) OK(hipMemRelease(piece[k]));
	}

	// ------------------------------------------------- hipMalloc for scale
	printf("# for scale: three hipMalloc arrays, all six role assignments\n");
	void *m[3];
	for (int i = 0; i < 3; i++) OK(hipMalloc(&m[i], ARR));
	const int perm[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
	for (const auto &p : perm)
		printf("M roles (%d,%d,%d) %.3f\n", p[0], p[1], p[2], frac(t1r2w(m[p[0]], m[p[1]], m[p[2]])));
	return 0;
}
