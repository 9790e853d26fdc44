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

static void sweep_virtual();
static void tlb_case();
static void alignment_case();

int main(int argc, char **argv)
{
	setvbuf(stdout, nullptr, _IOLBF, 0);
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
	if (argc > 1 && argv[1][0] == 'E') {
		sweep_virtual();
		return 0;
	}
	if (argc > 1 && argv[1][0] == 'F') {
		alignment_case();
		return 0;
	}
	if (argc > 1 && argv[1][0] == 'T') {
		tlb_case();
		return 0;
	}

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

// E: A..D say the spread follows the VIRTUAL addresses.  Which bits?  One
// physical triple, one 96 GiB address window W, the arrays mapped at
//   in = W + a,  out0 = W + b,  out1 = W + c
//   E1  b - a = 4 GiB + d, c - b = 4 GiB + d for d = 0, 2 MiB ... 2 GiB
//   E2  the whole triple (spacing 4 GiB) moved by 2 MiB ... 16 GiB
//   E3  all six orders of the three arrays at spacing 4 GiB + 2 MiB
//   E4  40 random 2 MiB-aligned placements (fixed seed): the distribution
static void sweep_virtual()
{
	const size_t MiB = (size_t)1 << 20;
	hipMemGenericAllocationHandle_t P[3];
	for (int i = 0; i < 3; i++) P[i] = phys(ARR);
	const size_t WIN = 96 * GiB;
	char *W = static_cast<char *>(reserve(WIN));
	printf("# E: window at %p\n", (void *)W);
	auto at = [&](size_t a, size_t b, size_t c) {
		map(W + a, ARR, P[0]); map(W + b, ARR, P[1]); map(W + c, ARR, P[2]);
		const float ms = t1r2w(W + a, W + b, W + c);
		unmap(W + a, ARR); unmap(W + b, ARR); unmap(W + c, ARR);
		return frac(ms);
	};
	printf("# E1: spacing 4 GiB + d (in lowest)   and the same with in highest\n");
	for (size_t d = 0; d <= 2 * GiB; d = d ? d * 2 : 2 * MiB)
		printf("E1 d=%6zu MiB  in<out0<out1 %.3f   out1<out0<in %.3f\n", d / MiB,
			at(0, ARR + d, 2 * (ARR + d)), at(2 * (ARR + d), ARR + d, 0));
	printf("# E2: triple at spacing 4 GiB, moved as a whole\n");
	for (size_t s = 0; s <= 16 * GiB; s = s ? s * 2 : 2 * MiB)
		printf("E2 shift=%6zu MiB  %.3f\n", s / MiB, at(s, s + ARR, s + 2 * ARR));
	printf("# E3: orders at spacing 4 GiB + 2 MiB (positions 0,1,2 = ascending addresses)\n");
	const int perm[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
	for (const auto &p : perm) {
		const size_t sp = ARR + 2 * MiB;
		printf("E3 in@%d out0@%d out1@%d  %.3f\n", p[0], p[1], p[2],
			at(p[0] * sp, p[1] * sp, p[2] * sp));
	}
	printf("# E4: random 2 MiB-aligned placements\n");
	unsigned long long x = 88172645463325252ull;
	auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
	std::vector<double> fr;
	for (int t = 0; t < 40; t++) {
		size_t o[3];
		for (;;) {
			for (int k = 0; k < 3; k++)
				o[k] = (size_t)(rnd() % ((WIN - ARR) / (2 * MiB))) * 2 * MiB;
			auto apart = [&](int i, int j) { return o[i] + ARR <= o[j] || o[j] + ARR <= o[i]; };
			if (apart(0, 1) && apart(0, 2) && apart(1, 2)) break;
		}
		const double f = at(o[0], o[1], o[2]);
		fr.push_back(f);
		printf("E4 in=+%6zu MiB out0=+%6zu MiB out1=+%6zu MiB  %.3f\n",
			o[0] / MiB, o[1] / MiB, o[2] / MiB, f);
	}
	std::sort(fr.begin(), fr.end());
	printf("# E4: min %.3f  median %.3f  max %.3f\n", fr.front(), fr[fr.size() / 2], fr.back());
}

// T: for a counter pass (rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum ...):
// three hipMalloc arrays, the six role assignments timed, then the SLOWEST
// assignment three times and the FASTEST three times -- the last six
// dispatches of the run, in that order.
static void tlb_case()
{
	void *m[3];
	for (int i = 0; i < 3; i++) OK(hipMalloc(&m[i], ARR));
	printf("# T: hipMalloc arrays at %p %p %p\n", m[0], m[1], m[2]);
	const int perm[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
	int lo = 0, hi = 0;
	float tl = 1e9f, th = 0.f;
	for (int k = 0; k < 6; k++) {
		const float ms = hbm_probe(m[perm[k][0]], nullptr, m[perm[k][1]], m[perm[k][2]],
				ARR / 4, 1, 2, 2, 4, nullptr);
		printf("T roles (%d,%d,%d) %.3f\n", perm[k][0], perm[k][1], perm[k][2], frac(ms));
		if (ms < tl) { tl = ms; lo = k; }
		if (ms > th) { th = ms; hi = k; }
	}
	printf("T slowest (%d,%d,%d), fastest (%d,%d,%d): last 6 dispatches = 3 x slowest, 3 x fastest\n",
		perm[hi][0], perm[hi][1], perm[hi][2], perm[lo][0], perm[lo][1], perm[lo][2]);
	for (int k : {hi, lo})
		printf("T %s %.3f\n", k == hi ? "slowest" : "fastest",
			frac(hbm_probe(m[perm[k][0]], nullptr, m[perm[k][1]], m[perm[k][2]],
				ARR / 4, 1, 2, 2, 2, nullptr)));
}

// F: is it the ALIGNMENT of the virtual address (page-table fragment size:
// a PTE can describe a 2^k-aligned block that is contiguous in both spaces)?
// Three physical handles; for each alignment 2^k a fresh reservation per
// array whose address is a multiple of 2^k but not of 2^(k+1); all three
// arrays equally aligned; 1R2W, three role rotations; twice.
static void alignment_case()
{
	hipMemGenericAllocationHandle_t P[3];
	for (int i = 0; i < 3; i++) P[i] = phys(ARR);
	for (int rep = 0; rep < 2; rep++)
		for (int k : {21, 22, 23, 24, 26, 28, 30, 32, 33}) {
			const size_t al = (size_t)1 << k;
			char *v[3];
			void *base[3];
			for (int i = 0; i < 3; i++) {
				// 2^(k+1)-aligned window of ARR + 2^k bytes, array at + 2^k
				OK(hipMemAddressReserve(&base[i], ARR + 2 * al, 2 * al, nullptr, 0));
				v[i] = static_cast<char *>(base[i]) + al;
				map(v[i], ARR, P[i]);
			}
			printf("F rep %d align 2^%d  in@%p  roles (0,1,2) %.3f (1,2,0) %.3f (2,0,1) %.3f\n",
				rep, k, (void *)v[0], frac(t1r2w(v[0], v[1], v[2])),
				frac(t1r2w(v[1], v[2], v[0])), frac(t1r2w(v[2], v[0], v[1])));
			for (int i = 0; i < 3; i++) {
				unmap(v[i], ARR);
				OK(hipMemAddressFree(base[i], ARR + 2 * al));
			}
		}
}
