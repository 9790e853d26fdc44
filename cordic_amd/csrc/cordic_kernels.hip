// cordic_kernels.hip -- generic (run-time parameterised) kernels, test-input
// kernels and the launch logic that picks between them and the unrolled
// instances (cordic_inst_*.hip).  Device code shared with the instances lives
// in cordic_device.h.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <type_traits>

#include "cordic_device.h"
#include "cordic_internal.h"
#include "cordic_launch.h"
#include "cordic_xydir.h"

namespace cordic_amd {
// cordic_last_kernel(): which kernel family served this thread's most recent
// rotator / converter launch (diagnostic; lets a test assert that the fast
// path really ran instead of a quietly slower one)
thread_local int g_last_kernel = CORDIC_KERNEL_NONE;

// ---- prologue images of the seeded kernels (cordic_internal.h: SeedImages)
struct SeedImages {
	struct Slot {
		uint32_t *d = nullptr;
		size_t	bytes = 0;
		int	container = 0;
		int32_t	x0 = 0, y0 = 0;
		hipEvent_t ready = nullptr;	// behind the build kernel
		bool	settled = false;	// `ready` has been seen complete
	};
	std::mutex mu;
	Slot	slot[kSeedImageSlots];
	int	used = 0;
	unsigned long long hits = 0, misses = 0;
};

SeedImages *seed_images_create() { return new (std::nothrow) SeedImages; }

void seed_images_destroy(SeedImages *c)
{
	if (!c)
		return;
	for (SeedImages::Slot &s : c->slot) {
		if (s.ready) (void)hipEventDestroy(s.ready);
		if (s.d) (void)hipFree(s.d);
	}
	delete c;
}

// wait (on the host) until every image built so far is complete and mark it
// so: what cordic_plan_prepare promises, so that a stream capture started
// right behind it may use the image
bool seed_images_settle(SeedImages *c)
{
	if (!c)
		return true;
	std::lock_guard<std::mutex> lock(c->mu);
	bool fine = true;
	for (int k = 0; k < c->used; k++) {
		SeedImages::Slot &s = c->slot[k];
		if (s.settled)
			continue;
		if (hipEventSynchronize(s.ready) == hipSuccess)
			s.settled = true;
		else {
			(void)hipGetLastError();
			fine = false;
		}
	}
	return fine;
}

void seed_images_info(const SeedImages *c, int32_t *held, uint64_t *hits,
		uint64_t *misses)
{
	SeedImages *m = const_cast<SeedImages *>(c);
	if (held) *held = 0;
	if (hits) *hits = 0;
	if (misses) *misses = 0;
	if (!m)
		return;
	std::lock_guard<std::mutex> lock(m->mu);
	if (held) *held = m->used;
	if (hits) *hits = m->hits;
	if (misses) *misses = m->misses;
}

namespace {
using namespace dev;

// The image of (container, x0, y0) for a launch on `st`, or NULL (the kernel
// then computes its prologue itself, as before).  `build(dst)` enqueues the
// build-mode kernel on `st`.
template <typename F>
const uint32_t *seed_image_for(SeedImages &c, int container, int32_t x0, int32_t y0,
		size_t bytes, hipStream_t st, F build)
{
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	if (st && hipStreamIsCapturing(st, &cs) != hipSuccess) {
		(void)hipGetLastError();
		cs = hipStreamCaptureStatusNone;
	}
	const bool capturing = cs != hipStreamCaptureStatusNone;
	std::lock_guard<std::mutex> lock(c.mu);
	for (int k = 0; k < c.used; k++) {
		SeedImages::Slot &s = c.slot[k];
		if (s.container != container || s.x0 != x0 || s.y0 != y0 || s.bytes != bytes)
			continue;
		if (!s.settled) {
			// (while a stream is being captured nothing may be asked of the
			// runtime -- hipEventQuery invalidates a global-mode capture --
			// and a graph must not depend on an event outside it: a captured
			// launch takes an image only once the host KNOWS it is complete,
			// which cordic_plan_prepare and any later eager launch establish)
			if (capturing) {
				c.misses++;
				return nullptr;
			}
			if (hipEventQuery(s.ready) == hipSuccess) {
				s.settled = true;
			} else {
				(void)hipGetLastError();	// hipErrorNotReady
				if (hipStreamWaitEvent(st, s.ready, 0) != hipSuccess) {
					(void)hipGetLastError();
					c.misses++;
					return nullptr;
				}
			}
		}
		c.hits++;
		return s.d;
	}
	c.misses++;
	if (capturing || c.used >= kSeedImageSlots)
		return nullptr;
	SeedImages::Slot &s = c.slot[c.used];
	if (!s.ready && hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess) {
		(void)hipGetLastError();
		s.ready = nullptr;
		return nullptr;
	}
	if (s.d && s.bytes != bytes) {
		(void)hipFree(s.d);
		s.d = nullptr;
	}
	if (!s.d && hipMalloc((void **)&s.d, bytes) != hipSuccess) {
		(void)hipGetLastError();
		s.d = nullptr;
		return nullptr;
	}
	s.bytes = bytes;
	if (!build(s.d) || hipGetLastError() != hipSuccess
			|| hipEventRecord(s.ready, st) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr;		// (the slot's memory is kept for the next try)
	}
	s.container = container;
	s.x0 = x0;
	s.y0 = y0;
	s.settled = false;
	c.used++;
	return s.d;			// same stream: ordered behind the build
}

// Would seed_image_for() hand this launch an image -- one it holds and may use,
// or one it could build now?  Nothing is changed (the batch-size rule asks
// before the launch is shaped: ADVICE r05 -- a ninth constant vector, or a
// captured launch nobody prepared, computes its prologue in every block and
// wants the higher threshold).
bool seed_image_would_serve(SeedImages &c, int container, int32_t x0, int32_t y0,
		size_t bytes, hipStream_t st)
{
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	if (st && hipStreamIsCapturing(st, &cs) != hipSuccess) {
		(void)hipGetLastError();
		cs = hipStreamCaptureStatusNone;
	}
	const bool capturing = cs != hipStreamCaptureStatusNone;
	std::lock_guard<std::mutex> lock(c.mu);
	for (int k = 0; k < c.used; k++) {
		const SeedImages::Slot &s = c.slot[k];
		if (s.container == container && s.x0 == x0 && s.y0 == y0 && s.bytes == bytes)
			return !capturing || s.settled;
	}
	return !capturing && c.used < kSeedImageSlots;
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
	int32_t o = (int32_t)(w >> kp.r);
	o = kp.wrap ? sext32(o, kp.ow) : o;
	return kp.post_mul ? unit_gain(o, kp.post_mul) : o;
}

// one sample of rtl/cordic.v:131-188, 231-283, 288-314, literally: fold,
// kp.nlive micro-rotations with the explicit WW-bit wrap, rounding
__device__ __forceinline__ void generic_rotate(const CoreParams &kp, uint32_t P,
		int32_t ix, int32_t iy, int32_t &rx, int32_t &ry)
{
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
	rx = round_generic(x, kp);
	ry = round_generic(y, kp);
}

// The samples behind a job's last whole vector, for every job of a batch in
// one launch (cordic_jobset): one lane per sample, addresses from the table.
template <Feed FEED>
__global__ __launch_bounds__(kBlock) void rotator_job_tails(CoreParams kp,
		const TailDesc *__restrict__ t, uint32_t n)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const TailDesc d = t[i];
	uint32_t P;
	if constexpr (FEED == Feed::Nco_ConstXY)
		P = (uint32_t)d.in;
	else
		P = *reinterpret_cast<const uint32_t *>((uintptr_t)d.in) << kp.pw_shl;
	int32_t rx, ry;
	generic_rotate(kp, P, kp.x0, kp.y0, rx, ry);
	*reinterpret_cast<int32_t *>((uintptr_t)d.ox) = rx;
	*reinterpret_cast<int32_t *>((uintptr_t)d.oy) = ry;
}

template <Feed FEED, typename IO = Io32>
__global__ __launch_bounds__(kBlock) void rotator_generic(CoreParams kp,
		const typename IO::ielem *__restrict__ xin,
		const typename IO::ielem *__restrict__ yin,
		const typename IO::uelem *__restrict__ phin,
		typename IO::ielem *__restrict__ ox,
		typename IO::ielem *__restrict__ oy, size_t n)
{
	const size_t stride = (size_t)gridDim.x * kBlock;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
			i += stride) {
		uint32_t P;
		int32_t ix, iy;
		if (FEED == Feed::Nco_ConstXY
				|| (FEED == Feed::PhaseArray_XYArray && kp.xy_nco))
			P = kp.phase0 + (uint32_t)(kp.index0 + i) * kp.fcw;
		else
			P = (uint32_t)phin[i] << kp.pw_shl;
		if constexpr (FEED == Feed::PhaseArray_XYArray) {
			ix = sext32(xin[i], kp.iw);
			iy = sext32(yin[i], kp.iw);
		} else {
			ix = kp.x0;
			iy = kp.y0;
		}
		int32_t rx, ry;
		generic_rotate(kp, P, ix, iy, rx, ry);
		ox[i] = (typename IO::ielem)rx;
		oy[i] = (typename IO::ielem)ry;
	}
}

// one sample of rtl/topolar.v:122-152, 217-243, 251-271, literally
__device__ __forceinline__ void generic_topolar(const CoreParams &kp, int32_t ix,
		int32_t iy, int32_t &mag, uint32_t &ph)
{
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
	mag = round_generic(x, kp);
	ph = p >> kp.pw_shl;
}

template <typename IO = Io32>
__global__ __launch_bounds__(kBlock) void topolar_generic(CoreParams kp,
		const typename IO::ielem *__restrict__ xin,
		const typename IO::ielem *__restrict__ yin,
		typename IO::ielem *__restrict__ omag,
		typename IO::uelem *__restrict__ oph, size_t n)
{
	const size_t stride = (size_t)gridDim.x * kBlock;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
			i += stride) {
		int32_t mag;
		uint32_t ph;
		generic_topolar(kp, sext32(xin[i], kp.iw), sext32(yin[i], kp.iw), mag, ph);
		omag[i] = (typename IO::ielem)mag;
		oph[i] = (typename IO::uelem)ph;
	}
}

// The samples behind a job's last whole vector for the data-fed kinds of a
// job set (round 6): one lane per sample, one TileDescXY per sample.
template <int KIND>
__global__ __launch_bounds__(kBlock) void xy_job_tails(CoreParams kp,
		const TileDescXY *__restrict__ t, uint32_t n)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const TileDescXY d = t[i];
	const int32_t ix = sext32(*reinterpret_cast<const int32_t *>((uintptr_t)d.in0), kp.iw);
	const int32_t iy = sext32(*reinterpret_cast<const int32_t *>((uintptr_t)d.in1), kp.iw);
	int32_t a;
	uint32_t b;
	if constexpr (KIND == CORDIC_JOBS_R2P) {
		generic_topolar(kp, ix, iy, a, b);
	} else {
		const uint32_t P = KIND == CORDIC_JOBS_MIX ? (uint32_t)d.in2
			: *reinterpret_cast<const uint32_t *>((uintptr_t)d.in2) << kp.pw_shl;
		int32_t ry;
		generic_rotate(kp, P, ix, iy, a, ry);
		b = (uint32_t)ry;
	}
	*reinterpret_cast<int32_t *>((uintptr_t)d.o0) = a;
	*reinterpret_cast<uint32_t *>((uintptr_t)d.o1) = b;
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

// rtl/sintable.v:72-77 / rtl/quarterwav.v:86-108: one gather per sample from
// a table that lives in L2 / Infinity Cache (at most 2^25 entries).
template <bool QUARTER>
__device__ __forceinline__ int32_t table_sample(const int32_t *__restrict__ tbl,
		uint32_t ph, int pw, int ow)
{
	const int sh = 32 - ow;
	if constexpr (!QUARTER) {
		return tbl[ph & ((1u << pw) - 1u)];
	} else {
		const uint32_t qm = (1u << (pw - 2)) - 1u;
		const uint32_t idx = ((ph >> (pw - 2)) & 1u) ? (~ph & qm) : (ph & qm);
		int32_t v = tbl[idx];
		if ((ph >> (pw - 1)) & 1u)
			v = -v;
		return (int32_t)((uint32_t)v << sh) >> sh;	// OW-bit wrap
	}
}

// Work distribution of the three table kernels: persistent 1024-thread blocks
// (they keep their table in LDS) that pull 1024-vector tiles from the per-XCD
// counters in `queue` in address order (cordic_device.h: for_each_queued_tile;
// an arithmetic-free 1R1W stream runs at 0.58-0.63 of the HBM peak with one
// contiguous chunk per block and at 0.72-0.79 this way, profiles/r02/
// hbm_probe2.txt).  queue == NULL: the chunk-per-block sweep.
// PF: fetch the next tile's phases before this tile is worked on.  Worth it
// where ONE block fits a CU (the 128 KiB tables: +12 %) or the lookups leave
// the CU (+1.3 %); with two blocks per CU the blocks already cover each
// other's loads and the extra registers cost 2-4 % (profiles/r03/
// tables_prefetch.txt).
template <bool PF, typename F>
__device__ __forceinline__ void sweep_tiles(uint32_t *queue, volatile uint32_t *slot,
		size_t nvec, const u32x4g *pv, F one_vector)
{
	if (queue && !PF) {
		for_each_queued_tile<1024>(queue, slot,
			(uint32_t)((nvec + 1023) / 1024), [&](uint32_t tile) {
				const size_t g = (size_t)tile * 1024 + threadIdx.x;
				if (g < nvec)
					one_vector(g, __builtin_nontemporal_load(&pv[g]));
			});
		return;
	}
	if (queue) {
		// the lane's vector of a tile, clamped into the batch (only the last
		// tile is partial): the prefetch needs no predicate
		const size_t last = nvec - 1;
		for_each_queued_tile_prefetched<1024, u32x4>(queue, slot,
			(uint32_t)((nvec + 1023) / 1024),
			[&](uint32_t tile) -> u32x4 {
				size_t g = (size_t)tile * 1024 + threadIdx.x;
				g = g < last ? g : last;
				return __builtin_nontemporal_load(&pv[g]);
			},
			[&](uint32_t tile, const u32x4 p) {
				const size_t g = (size_t)tile * 1024 + threadIdx.x;
				if (g < nvec)
					one_vector(g, p);
			});
		return;
	}
	size_t chunk = (nvec + gridDim.x - 1) / gridDim.x;
	chunk = (chunk + 1023) / 1024 * 1024;
	const size_t lo = (size_t)blockIdx.x * chunk;
	const size_t hi = (lo + chunk < nvec) ? lo + chunk : nvec;
	for (size_t g = lo + threadIdx.x; g < hi; g += 1024)
		one_vector(g, __builtin_nontemporal_load(&pv[g]));
}

// (without the prefetch: the placement probe, whose loads are its whole work)
template <typename F>
__device__ __forceinline__ void sweep_tiles(uint32_t *queue, volatile uint32_t *slot,
		size_t nvec, F one_vector)
{
	for_each_queued_tile<1024>(queue, slot,
		(uint32_t)((nvec + 1023) / 1024), [&](uint32_t tile) {
			const size_t g = (size_t)tile * 1024 + threadIdx.x;
			if (g < nvec)
				one_vector(g);
		});
}

template <bool QUARTER>
__global__ __launch_bounds__(1024) void table_lookup(
		const int32_t *__restrict__ tbl, const uint32_t *__restrict__ phase,
		int32_t *__restrict__ val, size_t n, int pw, int ow, uint32_t *queue)
{
	__shared__ uint32_t slot[3];
	const size_t nvec = n / kVec;
	const u32x4g *pv = reinterpret_cast<const u32x4g *>(phase);
	i32x4g *ov = reinterpret_cast<i32x4g *>(val);
	sweep_tiles<true>(queue, slot, nvec, pv, [&](size_t g, const u32x4 p) {
		i32x4 o;
#pragma unroll
		for (int v = 0; v < kVec; v++)
			o[v] = table_sample<QUARTER>(tbl, p[v], pw, ow);
		__builtin_nontemporal_store(o, &ov[g]);
	});
	if (blockIdx.x == 0)
		for (size_t i = nvec * kVec + threadIdx.x; i < n; i += 1024)
			val[i] = table_sample<QUARTER>(tbl, phase[i], pw, ow);
}

// Small tables: a packed int16 copy in LDS, so that random phases cost an LDS
// gather instead of an L2 gather (195 -> ~500 Gsample/s measured).  The
// full-wave table of -t tbl is kept as its first quadrant plus the peak entry
// when the HOST has checked that the generated table really has the
// symmetry (sin() of the mirrored argument can differ in the last bit, and
// the entries are truncated, so it is a property to test, not to assume).
//   mode 1: -t qtr table as is           (entries  = 2^(PW-2))
//   mode 2: -t tbl folded to a quadrant  (entries  = 2^(PW-2) + 1)
//   E = int32_t: the same two layouts for outputs wider than 16 bits, the
//   entries taken from the 32-bit table itself (lds modes 3 / 4; 2^15 entries
//   = 128 KiB, one block per CU) -- round 2 gathered those from L2.
template <int MODE, typename E>
__global__ __launch_bounds__(1024) void table_lookup_lds(
		const E *__restrict__ packed, int entries,
		const uint32_t *__restrict__ phase, int32_t *__restrict__ val,
		size_t n, int pw, int ow, uint32_t *queue)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	E *lds16 = reinterpret_cast<E *>(lds_raw);
	__shared__ uint32_t slot[3];
	for (int i = threadIdx.x; i < entries; i += 1024)
		lds16[i] = packed[i];
	__syncthreads();
	const uint32_t qm = (1u << (pw - 2)) - 1u;
	const int sh = 32 - ow;
	auto sample = [&](uint32_t ph) -> int32_t {
		const uint32_t mirror = (ph >> (pw - 2)) & 1u;
		const uint32_t neg = (ph >> (pw - 1)) & 1u;
		int32_t v;
		if constexpr (MODE == 1) {	// rtl/quarterwav.v:86-108
			v = lds16[mirror ? (~ph & qm) : (ph & qm)];
			if (neg) v = -v;
			return (int32_t)((uint32_t)v << sh) >> sh;
		} else {			// quadrant fold of rtl/sintable.v:72-77
			const uint32_t j = ph & qm;
			v = lds16[mirror ? (qm + 1u - j) : j];
			return neg ? -v : v;
		}
	};
	const size_t nvec = n / kVec;
	const u32x4g *pv = reinterpret_cast<const u32x4g *>(phase);
	i32x4g *ov = reinterpret_cast<i32x4g *>(val);
	// (32-bit entries: up to 128 KiB of LDS, one block per CU)
	sweep_tiles<(sizeof(E) > 2)>(queue, slot, nvec, pv, [&](size_t g, const u32x4 p) {
		i32x4 o;
#pragma unroll
		for (int v = 0; v < kVec; v++)
			o[v] = sample(p[v]);
		__builtin_nontemporal_store(o, &ov[g]);
	});
	if (blockIdx.x == 0)
		for (size_t i = nvec * kVec + threadIdx.x; i < n; i += 1024)
			val[i] = sample(phase[i]);
}

// ------------------------------------------------- quadratic sine core
//
// rtl/quadtbl.v on one sample.  Table entries are {C, L, Q, 0} (one 16-byte
// gather).  Every intermediate keeps the width of its RTL register:
//   qprod  QBITS+DXBITS   lsum  LBITS   lprod  LBITS+DXBITS   r_value  CBITS.
struct QuadParams {
	int32_t	pw, ow, xtra, ww, lgtbl, dxbits, cbits, lbits;
};

__device__ __forceinline__ int64_t sext64n(int64_t v, int bits)
{
	const int s = 64 - bits;
	return (int64_t)((uint64_t)v << s) >> s;
}

__device__ __forceinline__ int32_t quad_sample(const i32x4 e, uint32_t ph,
		const QuadParams &qp)
{
	const int sh = qp.dxbits - 1;
	const int32_t dx = (int32_t)(ph & ((1u << sh) - 1u));	// :153 {1'b0, ...}
	const int64_t qprod = (int64_t)e[2] * dx;		// :170
	// :214-221  w_qprod = sign-extended qprod[top : DXBITS-1]; lsum wraps
	const int32_t lsum = (int32_t)sext64n((qprod >> sh) + e[1], qp.lbits);
	const int64_t lprod = (int64_t)lsum * dx;		// :246
	// :270-277  r_value = w_lprod + cv_3 in CBITS bits
	const int64_t r = sext64n((lprod >> sh) + e[0], qp.cbits);
	// :292-300  round to OW bits unless that would overflow
	const uint32_t rw = (uint32_t)r & (uint32_t)((1ull << qp.ww) - 1ull);
	const uint32_t body = (rw >> qp.xtra) & ((1u << (qp.ow - 1)) - 1u);
	const uint32_t top = rw >> (qp.ww - 1);
	uint32_t w = rw;
	const bool pos_max = (top == 0) && body == ((1u << (qp.ow - 1)) - 1u);
	const bool neg_half = (top == 1) && body == (1u << (qp.ow - 2));
	if (!pos_max && !neg_half) {
		const uint32_t b = (rw >> qp.xtra) & 1u;
		w = rw + (1u << (qp.xtra - 1)) - 1u + b;
	}
	const int s = 32 - qp.ow;
	return (int32_t)((w >> qp.xtra) << s) >> s;		// :308
}

// The tables are small (the generator stops growing them at one LSB of fit
// error: at most 2^10 entries for the widest core it can write, 16 KiB
// packed), so every block keeps its own copy in LDS.  1024-thread blocks,
// each sweeping one contiguous chunk: measured ~5 % faster than a grid-stride
// of 256-thread blocks for this 4 B in / 4 B out stream.
__global__ __launch_bounds__(1024) void quad_lookup(
		const i32x4 *__restrict__ tab, QuadParams qp,
		const uint32_t *__restrict__ phase, int32_t *__restrict__ val,
		size_t n, uint32_t *queue)
{
	extern __shared__ __attribute__((aligned(16))) i32x4 lds_tab[];
	__shared__ uint32_t slot[3];
	for (int i = threadIdx.x; i < (1 << qp.lgtbl); i += 1024)
		lds_tab[i] = tab[i];
	__syncthreads();
	const i32x4 *t = lds_tab;
	const uint32_t imask = (1u << qp.lgtbl) - 1u;
	const int ish = qp.dxbits - 1;
	const size_t nvec = n / kVec;
	const u32x4g *pv = reinterpret_cast<const u32x4g *>(phase);
	i32x4g *ov = reinterpret_cast<i32x4g *>(val);
	sweep_tiles<false>(queue, slot, nvec, pv, [&](size_t g, const u32x4 p) {
		i32x4 o;
#pragma unroll
		for (int v = 0; v < kVec; v++)
			o[v] = quad_sample(t[(p[v] >> ish) & imask], p[v], qp);
		__builtin_nontemporal_store(o, &ov[g]);
	});
	if (blockIdx.x == 0)
		for (size_t i = nvec * kVec + threadIdx.x; i < n; i += 1024)
			val[i] = quad_sample(t[(phase[i] >> ish) & imask], phase[i], qp);
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

bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }
bool aligned2(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 1u) == 0; }

// sample arrays are int32 or (io16) int16 underneath the int32_t* of the job
template <typename P> P *advance(P *p, size_t samples, bool io16)
{
	if (!p) return p;
	typedef typename std::conditional<std::is_const<P>::value, const char,
			char>::type B;
	return reinterpret_cast<P *>(reinterpret_cast<B *>(p)
			+ samples * (io16 ? 2 : 4));
}

template <Feed FEED>
void launch_rot_generic(int grid, hipStream_t st, const CoreParams &kp,
		const RotatorJob &t)
{
	if (t.io16)
		hipLaunchKernelGGL((rotator_generic<FEED, Io16>), dim3(grid),
			dim3(kBlock), 0, st, kp, (const int16_t *)t.x,
			(const int16_t *)t.y, (const uint16_t *)t.phase,
			(int16_t *)t.ox, (int16_t *)t.oy, t.n);
	else
		hipLaunchKernelGGL((rotator_generic<FEED, Io32>), dim3(grid),
			dim3(kBlock), 0, st, kp, t.x, t.y, t.phase, t.ox, t.oy, t.n);
}

void launch_pol_generic(int grid, hipStream_t st, const CoreParams &kp,
		const int32_t *x, const int32_t *y, int32_t *mag, uint32_t *ph,
		size_t n, bool io16)
{
	if (io16)
		hipLaunchKernelGGL((topolar_generic<Io16>), dim3(grid), dim3(kBlock),
			0, st, kp, (const int16_t *)x, (const int16_t *)y,
			(int16_t *)mag, (uint16_t *)ph, n);
	else
		hipLaunchKernelGGL((topolar_generic<Io32>), dim3(grid), dim3(kBlock),
			0, st, kp, x, y, mag, ph, n);
}

// CUs of the CURRENT device (plans and groups launch on whatever device the
// caller made current): one relaxed atomic per device ordinal, filled on first
// use -- no shared mutable state beyond that, as the ABI promises.
int cus_of_current_device()
{
	constexpr int kMaxDevices = 64;
	static std::atomic<int> cache[kMaxDevices];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return -1;
	if (dev >= 0 && dev < kMaxDevices) {
		const int c = cache[dev].load(std::memory_order_relaxed);
		if (c > 0)
			return c;
	}
	int cus = 0;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)
			!= hipSuccess || cus <= 0)
		return -1;
	if (dev >= 0 && dev < kMaxDevices)
		cache[dev].store(cus, std::memory_order_relaxed);
	return cus;
}

int grid_for(size_t work_items_per_block, size_t n, int blocks_per_cu = 8)
{
	const int cus = cus_of_current_device();
	if (cus < 0)
		return -1;
	const size_t blocks = (n + work_items_per_block - 1) / work_items_per_block;
	const size_t cap = (size_t)cus * blocks_per_cu;	// resident blocks; the rest is grid-stride

	return (int)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

// hipGetLastError() is sticky per thread: an unrelated earlier failure (e.g. a
// probing call inside another library) must not be blamed on our launch, so
// every launcher clears it first and checks it right after its own launch.
void clear_stale_error() { (void)hipGetLastError(); }

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
	// left-justified wide form: LJ = 64 - WW for WW 35..40, 30 for WW 33, 34
	const int lj = (c.ww >= 35 && c.ww <= 40) ? 64 - c.ww : 30;
	kp.r_lj = kp.r + lj;
	kp.round_base_lj = (int64_t)((uint64_t)kp.round_base << lj);
	kp.post_mul = (c.flags & CORDIC_FLAG_UNIT_GAIN) ? core_gain_annihilator(c) : 0u;
	return kp;
}

int32_t host_sext(int32_t v, int w)
{
	const int s = 32 - w;
	return (int32_t)((uint32_t)v << s) >> s;
}

// Number of leading stages whose shifted operand may not fit in 32 bits
// (k < WW-32), rounded up to an instantiated value.
int general_stages_for(int ww)
{
	const int need = ww - 33;	// stages k = 1 .. WW-33
	if (need <= 2) return 2;
	if (need <= 8) return 8;
	return kAllGeneral;
}

// Every block of the table-seeded kernel rebuilds the seed table in its
// prologue (~10 us: nothing is cached between launches), so a SMALL batch is
// served faster by the full recurrence, which has none: per launch 11.5 us
// against 23 us at 2^20 samples, equal at 2^23 (16 stages) / 2^22.6 (24
// stages), then the seeds win (profiles/r04/small_batch.txt).  A plan therefore
// takes the seeded kernel from this many samples on.  CORDIC_SEED_MIN_SAMPLES
// overrides it (0: always seeded -- what the test suite sets, so that the
// seeded kernels stay covered at test sizes).
#ifndef CORDIC_SEED_MIN_LOG2_WITH_IMAGE
#define CORDIC_SEED_MIN_LOG2_WITH_IMAGE 22
#endif
constexpr int kSeedMinLog2WithImage = CORDIC_SEED_MIN_LOG2_WITH_IMAGE;
long long forced_min_samples()
{
	static const long long forced = [] {
		const char *e = std::getenv("CORDIC_SEED_MIN_SAMPLES");
		return (e && *e) ? std::atoll(e) : -1ll;
	}();
	return forced;
}
size_t seed_min_samples(const cordic_config &cfg, long long plan_value, bool image)
{
	if (plan_value >= 0)
		return (size_t)plan_value;	// cordic_plan_set_min_samples
	if (forced_min_samples() >= 0)
		return (size_t)forced_min_samples();
	// with the prologue served from the plan's image (round 5) the table
	// kernels win from 2^22 samples on (16 stages: 18.7 against 20.7 us per
	// launch there, 16.1 against 13.3 at 2^21; 24 stages: 22.6 / 26.1 and
	// 19.3 / 16.3); where every block has to compute it (no plan image: more
	// than kSeedImageSlots vectors, stream capture before cordic_plan_prepare)
	// from 2^23 / 6 Mi (profiles/r05/small_batch.txt)
	if (image)
		return (size_t)1 << kSeedMinLog2WithImage;
	return cfg.nlive <= 18 ? (size_t)1 << 23 : (size_t)3 << 21;
}
// ... and the same for per-sample vectors with looked-up directions: their
// tables are small, but the plain kernel is still 10 % ahead up to 2^22
// samples (9.2 against 7.9 us per launch at 2^20; 68.8 against 71.9 at 2^24)
size_t dir_min_samples(long long plan_value)
{
	if (plan_value >= 0)
		return (size_t)plan_value;
	if (forced_min_samples() >= 0)
		return (size_t)forced_min_samples();
	return (size_t)1 << 23;
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
	kp.xy_nco = (FEED == Feed::PhaseArray_XYArray && j.xy_nco) ? 1u : 0u;

	// the vector kernels need element alignment only (cordic_device.h: Io32)
	bool (*const vec_aligned)(const void *) = j.io16 ? aligned2 : aligned4;
	bool vec_ok = vec_aligned(j.ox) && vec_aligned(j.oy);
	if (FEED != Feed::Nco_ConstXY && !kp.xy_nco)
		vec_ok = vec_ok && vec_aligned(j.phase);
	if (FEED == Feed::PhaseArray_XYArray)
		vec_ok = vec_ok && vec_aligned(j.x) && vec_aligned(j.y);
	// (per-sample vectors in a 64-bit container are folded with 32-bit
	// multipliers 2^in_shl: cordic_device.h kMadFold)
	const bool fast_ok = vec_ok && !(cfg.flags & CORDIC_FLAG_FORCE_GENERIC)
		&& (!cfg.needs_wrap || cfg.ww == 32 || cfg.ww == 64)
		&& (!j.io16 || cfg.ww <= 32)
		&& !(FEED == Feed::PhaseArray_XYArray && cfg.ww > 32 && kp.in_shl > 30);

	if (fast_ok) {
		const int grid = grid_for(kTile, j.n);
		if (grid < 0)
			return CORDIC_ERR_DEVICE;
		bool done = false;
		// constant-vector feeds with a plan: table-seeded kernel
		uint32_t *const seed_queue = (cfg.flags & CORDIC_FLAG_STATIC_CHUNKS)
				? nullptr : j.queue;
		// which container's instance serves the core: 3 int16 arrays,
		// 0 32-bit registers, 1 / 2 left-justified by 29 / 30
		const int container = j.io16 ? 3
			: (cfg.ww <= 32 && (cfg.needs_wrap
				|| (cfg.flags & CORDIC_FLAG_NO_LJ))) ? 0
			: cfg.ww == 35 ? 1 : 2;
		DtInfo seed_dt = j.dt;
		if (cfg.flags & CORDIC_FLAG_NO_TAILS)
			seed_dt.n = 0;	// A/B: phase recurrence behind the seeds
		// buckets, seeds, the three tile-id slots of the queue and the
		// direction tails
		const size_t seed_lds = (dt_lds_layout(seed_dt,
				(uint32_t)((size_t)j.seed_nbuckets * 8
				+ (size_t)j.seed_nleaves * 4 * 16 + 16), nullptr, nullptr)
				+ 15u) & ~(size_t)15;
		// (the lower batch-size threshold only where an image really serves
		// this launch; asked only for batches between the two thresholds)
		auto big_enough = [&]() -> bool {
			if (j.n < (size_t)kVec)
				return false;
			if (j.n >= seed_min_samples(cfg, j.min_samples, false))
				return true;
			return j.images && cfg.ww <= 35
				&& j.n >= seed_min_samples(cfg, j.min_samples, true)
				&& seed_image_would_serve(*j.images,
					container + (seed_queue ? 0 : 8), kp.x0, kp.y0, seed_lds, st);
		};
		if (FEED != Feed::PhaseArray_XYArray && j.seed_table
				&& j.seed_m == kSeedStages
				&& !(cfg.flags & CORDIC_FLAG_NO_SEED)
				&& (j.prepare_only || big_enough())) {
			static_assert(CORDIC_QUEUE_BYTES
				== kQueueCounters * kQueueStride * 4, "queue layout");
			uint32_t *queue = seed_queue;
			SeedArgs sa{j.seed_table, j.seed_S, j.seed_nbuckets,
					j.seed_nleaves, queue, seed_dt};
			const size_t lds = seed_lds;
			// blocks per CU: 32 waves and 160 KiB of LDS to share
			int per_cu = 32 / (kSeedBlock / 64);
			const int by_lds = (int)((160 * 1024) / (lds ? lds : 1));
			if (by_lds < per_cu) per_cu = by_lds;
			// (a block per tile at most: a block without one would stage the
			// table for nothing)
			const int g2 = grid_for((size_t)kSeedBlock * kVec * kSeedSub, j.n,
					per_cu < 1 ? 1 : per_cu);
			if (g2 < 0)
				return CORDIC_ERR_DEVICE;
			if (per_cu >= 1 && lds <= 160 * 1024) {
				auto run = [&](Feed feed, const SeedArgs &a, const RotatorJob &jj,
						int grid_) -> bool {
					switch (container) {
					case 3: return launch_seed_narrow16(feed, cfg.nlive, grid_,
							st, kp, a, jj, lds);
					case 0: return launch_seed_narrow(feed, cfg.nlive, grid_,
							st, kp, a, jj, lds);
					case 1: return launch_seed_lj29(feed, cfg.nlive, grid_, st,
							kp, a, jj, lds);
					// (every other WW <= 34 core: left-justified by 30)
					default: return cfg.ww < 35 && launch_seed_lj30(feed,
							cfg.nlive, grid_, st, kp, a, jj, lds);
					}
				};
				if (j.images && cfg.ww <= 35) {
					// the prologue's result for this vector, from the plan
					// (a launch without a queue runs the dynamic-exit
					// instance, whose image has no tail tables: own key)
					sa.image = seed_image_for(*j.images,
						container + (queue ? 0 : 8), kp.x0,
						kp.y0, lds, st, [&](uint32_t *dst) {
							SeedArgs b = sa;
							// (queue kept as it is: it selects the
							// instance, build mode never touches it)
							b.image = nullptr;
							b.image_out = dst;
							b.image_words = (uint32_t)(lds / 4);
							RotatorJob bj = j;
							bj.n = 0;
							// (the phase-array instance: the one that
							// carries the direction tails where there are any)
							return run(Feed::PhaseArray_ConstXY, b, bj, 1);
						});
					sa.image_words = (uint32_t)(lds / 4);
				}
				if (j.prepare_only)
					return sa.image ? check_launch() : CORDIC_ERR_UNSUPPORTED;
				done = run(FEED, sa, j, g2);
			}
		}
		if (j.prepare_only)
			return CORDIC_ERR_UNSUPPORTED;	// no table kernel for this core
		if (done)
			g_last_kernel = CORDIC_KERNEL_SEEDED;
		// per-sample vectors with a plan: directions looked up
		// (cordic_xydir.h); cores / counts without an instance fall through
		if (FEED == Feed::PhaseArray_XYArray && j.dir_table && j.dx.n > 0
				&& !j.io16 && j.n >= (size_t)kVec && kp.post_mul == 0
				&& j.n >= dir_min_samples(j.min_samples)
				&& kp.in_shl >= 1 && kp.in_shl <= 30 && cfg.ww <= 35
				&& !cfg.needs_wrap
				&& !(cfg.flags & (CORDIC_FLAG_NO_TAILS | CORDIC_FLAG_NO_LJ))) {
			dev::DirArgs da{j.dir_table, j.dx};
			const size_t lds = dev::dx_lds_layout(j.dx, nullptr, nullptr);
			// (no opt-in to more dynamic LDS than a launch gets by default:
			// a table that large runs the plain kernel instead of failing)
			done = lds > 64 * 1024 ? false : cfg.ww == 35
				? launch_xydir_lj29(cfg.nlive, grid, st, kp, da, j, lds)
				: launch_xydir_lj30(cfg.nlive, grid, st, kp, da, j, lds);
			if (done)
				g_last_kernel = CORDIC_KERNEL_DIRECTIONS;
		}
		if (!done && j.n >= (size_t)kVec) {
			g_last_kernel = CORDIC_KERNEL_UNROLLED;
			const int ngen = general_stages_for(cfg.ww);
			if (j.io16) {
				done = launch_rot_narrow16(FEED, cfg.nlive, grid, st, kp, j);
			} else if (cfg.ww <= 32 && (cfg.needs_wrap
					|| (cfg.flags & CORDIC_FLAG_NO_LJ))) {
				// 32-bit container: its wrap IS the WW = 32 wrap
				done = launch_rot_narrow(FEED, cfg.nlive, grid, st, kp, j);
			} else if (cfg.ww <= 32) {
				// Round 3: WW <= 32 cores run in the left-justified 64-bit
				// container too (LJ = 30: 7 instead of 8 instructions per
				// stage; full recurrence 19 / 27 stages +11 / +17 %, seeded
				// on unrelated phases +4 / +8 %, on ramps -2.5 / 0 %;
				// profiles/r03/ab_tails.txt, part 16)
				done = launch_rot_lj30(FEED, cfg.nlive, grid, st, kp, j);
			} else {
				// left-justified forms first (WW 33..40), the plain
				// 64-bit kernels for everything else
				if (cfg.flags & CORDIC_FLAG_NO_LJ)
					;
				else if (cfg.ww == 35)
					done = launch_rot_lj29(FEED, cfg.nlive, grid, st, kp, j);
				else if (cfg.ww < 35)
					done = launch_rot_lj30(FEED, cfg.nlive, grid, st, kp, j);
				else if (cfg.ww <= 40 && !cfg.needs_wrap) {
					switch (cfg.ww) {
					case 36: done = launch_rot_lj28(FEED, cfg.nlive, grid, st, kp, j); break;
					case 37: done = launch_rot_lj27(FEED, cfg.nlive, grid, st, kp, j); break;
					case 38: done = launch_rot_lj26(FEED, cfg.nlive, grid, st, kp, j); break;
					case 39: done = launch_rot_lj25(FEED, cfg.nlive, grid, st, kp, j); break;
					default: done = launch_rot_lj24(FEED, cfg.nlive, grid, st, kp, j); break;
					}
				}
				if (done)
					;
				else if (ngen == 2)
					done = launch_rot_wide2(FEED, cfg.nlive, grid, st, kp, j);
				else if (ngen == 8)
					done = launch_rot_wide8(FEED, cfg.nlive, grid, st, kp, j);
				else
					done = launch_rot_wideall(FEED, cfg.nlive, grid, st, kp, j);
			}
		}
		if (done) {
			// 0..3 trailing samples: generic kernel on the remainder
			const size_t head = j.n - j.n % kVec;
			if (head == j.n)
				return check_launch();
			RotatorJob t = j;
			t.n = j.n - head;
			t.ox = advance(t.ox, head, j.io16);
			t.oy = advance(t.oy, head, j.io16);
			t.phase = advance(t.phase, head, j.io16);	// (NULL stays NULL: mixer)
			t.x = advance(t.x, head, j.io16);
			t.y = advance(t.y, head, j.io16);
			kp.index0 += head;
			launch_rot_generic<FEED>(1, st, kp, t);
			return check_launch();
		}
	}
	if (j.prepare_only)
		return CORDIC_ERR_UNSUPPORTED;
	const int grid = grid_for(kBlock, j.n);
	if (grid < 0)
		return CORDIC_ERR_DEVICE;
	g_last_kernel = CORDIC_KERNEL_GENERIC;
	launch_rot_generic<FEED>(grid, st, kp, j);
	return check_launch();
}

} // namespace

// ----------------------------------------------------------------- launchers

int launch_rotator(const cordic_config &cfg, Feed feed, const RotatorJob &job,
		void *stream)
{
	clear_stale_error();
	if (cfg.mode != CORDIC_P2R && cfg.mode != CORDIC_SP2R)
		return CORDIC_ERR_MODE;
	if (!config_sane(cfg))
		return CORDIC_ERR_ARGS;
	if (job.prepare_only)
		return launch_rot_feed<Feed::PhaseArray_ConstXY>(cfg, job, stream);
	if (job.n == 0)
		return CORDIC_OK;
	if (!job.ox || !job.oy)
		return CORDIC_ERR_ARGS;
	switch (feed) {
	case Feed::PhaseArray_ConstXY:
		if (!job.phase) return CORDIC_ERR_ARGS;
		return launch_rot_feed<Feed::PhaseArray_ConstXY>(cfg, job, stream);
	case Feed::PhaseArray_XYArray:
		// (the mixer generates its phases: no array)
		if ((!job.phase && !job.xy_nco) || !job.x || !job.y) return CORDIC_ERR_ARGS;
		return launch_rot_feed<Feed::PhaseArray_XYArray>(cfg, job, stream);
	default:
		return launch_rot_feed<Feed::Nco_ConstXY>(cfg, job, stream);
	}
}

int launch_rotator_jobs(const cordic_config &cfg, Feed feed, const RotatorJob &j,
		const JobTables &tabs, void *stream)
{
	clear_stale_error();
	if (cfg.mode != CORDIC_P2R && cfg.mode != CORDIC_SP2R)
		return CORDIC_ERR_MODE;
	if (!config_sane(cfg) || feed == Feed::PhaseArray_XYArray)
		return CORDIC_ERR_ARGS;
	// the seeded kernel's dynamic-exit instance is the one that reads tile
	// descriptors: cores it serves (cordic_plan.cpp: seed_eligible), 32-bit
	// sample arrays, a tile queue
	if (!j.seed_table || j.seed_m != kSeedStages || !j.queue || j.io16
			|| (cfg.flags & (CORDIC_FLAG_FORCE_GENERIC | CORDIC_FLAG_NO_SEED
				| CORDIC_FLAG_STATIC_CHUNKS))
			|| (cfg.needs_wrap && cfg.ww != 32 && cfg.ww != 64)
			|| cfg.ww > 35 || cfg.nlive > kDynStages)
		return CORDIC_ERR_UNSUPPORTED;
	if (tabs.samples == 0)
		return CORDIC_OK;
	hipStream_t st = static_cast<hipStream_t>(stream);
	CoreParams kp = make_params(cfg);
	kp.x0 = host_sext(j.x0, cfg.iw);
	kp.y0 = host_sext(j.y0, cfg.iw);
	if (tabs.ntiles) {
		SeedArgs sa{j.seed_table, j.seed_S, j.seed_nbuckets, j.seed_nleaves,
				j.queue, j.dt};
#ifdef CORDIC_DESC_LOOP_ALL
		constexpr int kJobImageKey = 0;	// A/B: the static instances, their image
		if (cfg.flags & CORDIC_FLAG_NO_TAILS)
			sa.dt.n = 0;
#else
		// WW 35 cores of 16 / 24 stages: a static instance reads the
		// descriptors (cordic_internal.h: desc_static), tails and image as in
		// a single job; all others the dynamic-exit instance, which runs the
		// recurrence behind the seeds: no tail tables, an image slot of its own
		const bool stat = cfg.ww == 35		// (the WideLJ<29> unit carries them)
			&& !(cfg.flags & (CORDIC_FLAG_NO_TAILS | CORDIC_FLAG_NO_LJ))
			&& desc_static(cfg.nlive, j.dt.n);
		const int kJobImageKey = stat ? 0 : 4;
		if (!stat)
			sa.dt.n = 0;
#endif
		sa.tiles = tabs.tiles;
		sa.ntiles = tabs.ntiles;
		const size_t lds = (dt_lds_layout(sa.dt,
				(uint32_t)((size_t)j.seed_nbuckets * 8
				+ (size_t)j.seed_nleaves * 4 * 16 + 16), nullptr, nullptr)
				+ 15u) & ~(size_t)15;
		if (lds > 160 * 1024)
			return CORDIC_ERR_UNSUPPORTED;
		const int cus = cus_of_current_device();
		if (cus < 0)
			return CORDIC_ERR_DEVICE;
		const int grid = (int)(tabs.ntiles < (uint32_t)cus ? tabs.ntiles : (uint32_t)cus);
		const int container = (cfg.ww <= 32 && (cfg.needs_wrap
				|| (cfg.flags & CORDIC_FLAG_NO_LJ))) ? 0 : cfg.ww == 35 ? 1 : 2;
		auto run = [&](const SeedArgs &a, int grid_) -> bool {
			RotatorJob bj = j;
			bj.n = 0;
			switch (container) {
			case 0: return launch_seed_narrow(feed, cfg.nlive, grid_, st, kp, a, bj, lds);
			case 1: return launch_seed_lj29(feed, cfg.nlive, grid_, st, kp, a, bj, lds);
			default: return launch_seed_lj30(feed, cfg.nlive, grid_, st, kp, a, bj, lds);
			}
		};
		if (j.images) {
			sa.image = seed_image_for(*j.images, kJobImageKey + container, kp.x0, kp.y0,
				lds, st, [&](uint32_t *dst) {
					// one block of the same (dynamic-exit) instance in build
					// mode: it returns behind the prologue, the tile table is
					// never looked at
					SeedArgs b = sa;
					b.queue = nullptr;
					b.image_out = dst;
					b.image_words = (uint32_t)(lds / 4);
					return run(b, 1);
				});
			sa.image_words = (uint32_t)(lds / 4);
		}
		if (!run(sa, grid))
			return CORDIC_ERR_UNSUPPORTED;
		g_last_kernel = CORDIC_KERNEL_SEEDED;
		if (int rc = check_launch())
			return rc;
	}
	if (tabs.ntails) {
		const unsigned blocks = (tabs.ntails + kBlock - 1) / kBlock;
		const TailDesc *t = reinterpret_cast<const TailDesc *>(tabs.tails);
		if (feed == Feed::Nco_ConstXY)
			hipLaunchKernelGGL(rotator_job_tails<Feed::Nco_ConstXY>, dim3(blocks),
				dim3(kBlock), 0, st, kp, t, tabs.ntails);
		else
			hipLaunchKernelGGL(rotator_job_tails<Feed::PhaseArray_ConstXY>,
				dim3(blocks), dim3(kBlock), 0, st, kp, t, tabs.ntails);
		return check_launch();
	}
	return CORDIC_OK;
}

int launch_xy_jobs(const cordic_config &cfg, int kind, const RotatorJob &j,
		const JobTables &tabs, void *stream)
{
	clear_stale_error();
	const bool pol = kind == CORDIC_JOBS_R2P;
	if (!config_sane(cfg))
		return CORDIC_ERR_ARGS;
	if (pol != (cfg.mode == CORDIC_R2P || cfg.mode == CORDIC_SR2P))
		return CORDIC_ERR_MODE;
	if (tabs.samples == 0)
		return CORDIC_OK;
	hipStream_t st = static_cast<hipStream_t>(stream);
	CoreParams kp = make_params(cfg);
	kp.xy_nco = kind == CORDIC_JOBS_MIX ? 1u : 0u;
	const TileDescXY *tiles = reinterpret_cast<const TileDescXY *>(tabs.tiles);
	if (tabs.ntiles) {
		// the conditions of the single-call fast paths (launch_topolar:
		// topolar_lj; launch_rot_feed: rotator_xydir), minus the batch-size
		// thresholds -- a job set is one big batch
		if ((cfg.flags & (CORDIC_FLAG_FORCE_GENERIC | CORDIC_FLAG_NO_LJ))
				|| cfg.needs_wrap || cfg.nlive < 1)
			return CORDIC_ERR_UNSUPPORTED;
		const int cus = cus_of_current_device();
		if (cus < 0)
			return CORDIC_ERR_DEVICE;
		const uint32_t cap = (uint32_t)cus * 8u;	// resident blocks
		const int grid = (int)(tabs.ntiles < cap ? tabs.ntiles : cap);
		bool done = false;
		if (pol) {
			if (cfg.ww <= 34)
				done = launch_pol_lj_jobs(cfg.nlive, grid, st, kp, tiles, tabs.ntiles);
			if (done)
				g_last_kernel = CORDIC_KERNEL_LEFT_JUSTIFIED;
		} else if (j.dir_table && j.dx.n > 0 && kp.post_mul == 0 && kp.in_shl >= 1
				&& kp.in_shl <= 30 && cfg.ww <= 35
				&& !(cfg.flags & CORDIC_FLAG_NO_TAILS)) {
			dev::DirArgs da{j.dir_table, j.dx};
			const size_t lds = dev::dx_lds_layout(j.dx, nullptr, nullptr);
			if (lds <= 64 * 1024)
				done = cfg.ww == 35
					? launch_xydir_jobs_lj29(cfg.nlive, grid, st, kp, da, tiles,
						tabs.ntiles, lds)
					: launch_xydir_jobs_lj30(cfg.nlive, grid, st, kp, da, tiles,
						tabs.ntiles, lds);
			if (done)
				g_last_kernel = CORDIC_KERNEL_DIRECTIONS;
		}
		if (!done)
			return CORDIC_ERR_UNSUPPORTED;
		if (int rc = check_launch())
			return rc;
	}
	if (tabs.ntails) {
		const unsigned blocks = (tabs.ntails + kBlock - 1) / kBlock;
		const TileDescXY *t = reinterpret_cast<const TileDescXY *>(tabs.tails);
		if (kind == CORDIC_JOBS_R2P)
			hipLaunchKernelGGL(xy_job_tails<CORDIC_JOBS_R2P>, dim3(blocks),
				dim3(kBlock), 0, st, kp, t, tabs.ntails);
		else if (kind == CORDIC_JOBS_MIX)
			hipLaunchKernelGGL(xy_job_tails<CORDIC_JOBS_MIX>, dim3(blocks),
				dim3(kBlock), 0, st, kp, t, tabs.ntails);
		else
			hipLaunchKernelGGL(xy_job_tails<CORDIC_JOBS_P2R_XY>, dim3(blocks),
				dim3(kBlock), 0, st, kp, t, tabs.ntails);
		return check_launch();
	}
	return CORDIC_OK;
}

int launch_topolar(const cordic_config &cfg, size_t n, const int32_t *x,
		const int32_t *y, int32_t *mag, uint32_t *phase, void *stream,
		bool io16)
{
	clear_stale_error();
	if (cfg.mode != CORDIC_R2P && cfg.mode != CORDIC_SR2P)
		return CORDIC_ERR_MODE;
	if (!config_sane(cfg))
		return CORDIC_ERR_ARGS;
	if (n == 0)
		return CORDIC_OK;
	if (!x || !y || !mag || !phase)
		return CORDIC_ERR_ARGS;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const CoreParams kp = make_params(cfg);
	bool (*const vec_aligned)(const void *) = io16 ? aligned2 : aligned4;
	const bool vec_ok = vec_aligned(x) && vec_aligned(y) && vec_aligned(mag)
			&& vec_aligned(phase);
	const bool fast_ok = vec_ok && !(cfg.flags & CORDIC_FLAG_FORCE_GENERIC)
		&& (!cfg.needs_wrap || cfg.ww == 32 || cfg.ww == 64)
		&& (!io16 || cfg.ww <= 32);
	if (fast_ok) {
		const int grid = grid_for(kTile, n);
		if (grid < 0)
			return CORDIC_ERR_DEVICE;
		bool done = false;
		if (n >= (size_t)kVec) {
			const int ngen = general_stages_for(cfg.ww);
			const bool lj_ok = !io16 && !cfg.needs_wrap && cfg.nlive >= 1
					&& !(cfg.flags & CORDIC_FLAG_NO_LJ);
			if (io16)
				done = launch_pol_narrow16(cfg.nlive, grid, st, kp, x, y,
						mag, phase, n);
			else if (lj_ok && cfg.ww <= 34)
				done = launch_pol_lj(cfg.nlive, grid, st, kp, x, y, mag,
						phase, n);
			else if (lj_ok && cfg.ww <= 40)
				done = launch_pol_ljw(cfg.nlive, grid, st, kp, x, y, mag,
						phase, n);
			// io16 is served by the unrolled narrow kernel, not topolar_lj
			g_last_kernel = (done && !io16) ? CORDIC_KERNEL_LEFT_JUSTIFIED
							: CORDIC_KERNEL_UNROLLED;
			if (done)
				;
			else if (io16)
				;
			else if (cfg.ww <= 32)
				done = launch_pol_narrow(cfg.nlive, grid, st, kp, x, y,
						mag, phase, n);
			else if (ngen <= 8)
				done = launch_pol_wide8(cfg.nlive, grid, st, kp, x, y,
						mag, phase, n);
			else
				done = launch_pol_wideall(cfg.nlive, grid, st, kp, x, y,
						mag, phase, n);
		}
		if (done) {
			const size_t head = n - n % kVec;
			if (head == n)
				return check_launch();
			launch_pol_generic(1, st, kp, advance(x, head, io16),
				advance(y, head, io16), advance(mag, head, io16),
				advance(phase, head, io16), n - head, io16);
			return check_launch();
		}
	}
	const int grid = grid_for(kBlock, n);
	if (grid < 0)
		return CORDIC_ERR_DEVICE;
	g_last_kernel = CORDIC_KERNEL_GENERIC;
	launch_pol_generic(grid, st, kp, x, y, mag, phase, n, io16);
	return check_launch();
}

int launch_fill_phase_ramp(uint32_t *p, size_t n, uint64_t index0, int shift,
		void *stream)
{
	clear_stale_error();
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
	clear_stale_error();
	if (n == 0) return CORDIC_OK;
	if (!x || !y || bits < 1 || bits > 32) return CORDIC_ERR_ARGS;
	const int grid = grid_for(kBlock * 4, n);
	if (grid < 0) return CORDIC_ERR_DEVICE;
	hipLaunchKernelGGL(fill_iq_ramp, dim3(grid), dim3(kBlock), 0,
			static_cast<hipStream_t>(stream), x, y, n, index0, mulx,
			muly, bits);
	return check_launch();
}

int launch_table_lookup(const cordic_table_config &t, const int32_t *d_tbl,
		size_t n, const uint32_t *phase, int32_t *val, void *stream,
		const int16_t *d_lds16, int lds_mode, int lds_entries,
		uint32_t *queue)
{
	clear_stale_error();
	if (n == 0) return CORDIC_OK;
	if (!d_tbl || !phase || !val || !table_sane(t)) return CORDIC_ERR_ARGS;
	if (!aligned4(phase) || !aligned4(val)) return CORDIC_ERR_ARGS;
	if (lds_mode >= 3 || (d_lds16 && lds_mode)) {
		const bool wide = lds_mode >= 3;
		const size_t bytes = ((size_t)lds_entries * (wide ? 4 : 2) + 15) & ~(size_t)15;
		int per_cu = (int)((160 * 1024) / (bytes + 64));
		if (per_cu > 2) per_cu = 2;
		const int grid = grid_for((size_t)1024 * kVec, n, per_cu < 1 ? 1 : per_cu);
		if (grid < 0) return CORDIC_ERR_DEVICE;
		hipStream_t st = static_cast<hipStream_t>(stream);
		const void *kern;
		switch (lds_mode) {
		case 1: kern = (const void *)table_lookup_lds<1, int16_t>; break;
		case 2: kern = (const void *)table_lookup_lds<2, int16_t>; break;
		case 3: kern = (const void *)table_lookup_lds<1, int32_t>; break;
		default: kern = (const void *)table_lookup_lds<2, int32_t>; break;
		}
		bool lds_ok = per_cu >= 1;
		if (lds_ok && bytes + 64 > 64 * 1024)	// + the kernel's static tile-id slots
			lds_ok = hipFuncSetAttribute(kern,
				hipFuncAttributeMaxDynamicSharedMemorySize,
				(int)bytes + 64) == hipSuccess;
		if (lds_ok) {
			if (lds_mode == 1)
				hipLaunchKernelGGL((table_lookup_lds<1, int16_t>), dim3(grid),
					dim3(1024), bytes, st, d_lds16, lds_entries, phase,
					val, n, t.pw, t.ow, queue);
			else if (lds_mode == 2)
				hipLaunchKernelGGL((table_lookup_lds<2, int16_t>), dim3(grid),
					dim3(1024), bytes, st, d_lds16, lds_entries, phase,
					val, n, t.pw, t.ow, queue);
			else if (lds_mode == 3)
				hipLaunchKernelGGL((table_lookup_lds<1, int32_t>), dim3(grid),
					dim3(1024), bytes, st, d_tbl, lds_entries, phase,
					val, n, t.pw, t.ow, queue);
			else
				hipLaunchKernelGGL((table_lookup_lds<2, int32_t>), dim3(grid),
					dim3(1024), bytes, st, d_tbl, lds_entries, phase,
					val, n, t.pw, t.ow, queue);
			return check_launch();
		}
		clear_stale_error();	// the L2 gather kernel below serves the table
	}
	const int grid = grid_for((size_t)1024 * kVec, n, 2);
	if (grid < 0) return CORDIC_ERR_DEVICE;
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (t.kind == CORDIC_QTR)
		hipLaunchKernelGGL(table_lookup<true>, dim3(grid), dim3(1024), 0,
				st, d_tbl, phase, val, n, t.pw, t.ow, queue);
	else
		hipLaunchKernelGGL(table_lookup<false>, dim3(grid), dim3(1024), 0,
				st, d_tbl, phase, val, n, t.pw, t.ow, queue);
	return check_launch();
}

int launch_quad_lookup(const cordic_quad_config &q, const int32_t *d_tables,
		size_t n, const uint32_t *phase, int32_t *val, void *stream,
		uint32_t *queue)
{
	clear_stale_error();
	if (n == 0) return CORDIC_OK;
	if (!d_tables || !phase || !val || !quad_sane(q)) return CORDIC_ERR_ARGS;
	if (!aligned4(phase) || !aligned4(val)) return CORDIC_ERR_ARGS;
	const int grid = grid_for((size_t)1024 * kVec, n, 2);
	if (grid < 0) return CORDIC_ERR_DEVICE;
	hipStream_t st = static_cast<hipStream_t>(stream);
	QuadParams qp{q.pw, q.ow, q.xtra, q.ww, q.lgtbl, q.dxbits, q.cbits, q.lbits};
	const i32x4 *tab = reinterpret_cast<const i32x4 *>(d_tables);
	const size_t bytes = (size_t)q.entries * sizeof(i32x4);
	if (bytes > 64 * 1024)
		return CORDIC_ERR_UNSUPPORTED;
	hipLaunchKernelGGL(quad_lookup, dim3(grid), dim3(1024), bytes, st, tab,
			qp, phase, val, n, queue);
	return check_launch();
}

// Arithmetic-free stand-in for a job's memory traffic: R arrays read, W arrays
// written, 16 bytes per lane, one 256-thread block per 4 KiB tile, tiles dealt
// XCD-contiguously -- the best streaming pattern found on MI355X (DESIGN.md
// 4.3).  cordic_group times it over candidate arrays to decide which
// allocation plays which role (cordic_group.cpp: placement).  The written
// arrays are overwritten.
template <int R, int W>
__global__ __launch_bounds__(256) void stream_probe(const dev::u32x4 *__restrict__ a,
		const dev::u32x4 *__restrict__ b, dev::u32x4 *__restrict__ c,
		dev::u32x4 *__restrict__ d, size_t nvec, int xcd)
{
	size_t t = blockIdx.x;
	if (xcd)
		t = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
	const size_t g = t * 256 + threadIdx.x;
	if (g >= nvec)
		return;
	// non-temporal, like the kernels whose traffic this stands in for
	dev::u32x4 v = dev::u32x4{(uint32_t)g, 1, 2, 3};
	if (R >= 1) v = __builtin_nontemporal_load(&a[g]);
	if (R >= 2) v += __builtin_nontemporal_load(&b[g]);
	if (W >= 1) __builtin_nontemporal_store(v, &c[g]);
	if (W >= 2) __builtin_nontemporal_store(v + 1, &d[g]);
}

// The same traffic in the seeded kernel's own work distribution: persistent
// 1024-thread blocks, one per CU, pulling 1024-vector tiles from the per-XCD
// counters in address order.  Which arrays suit a stream best depends on the
// distribution too (one-shot tiles and the queue disagree on some
// combinations), so placement probes with the one the job will use.
template <int R, int W>
__global__ __launch_bounds__(1024) void stream_probe_queued(
		const dev::u32x4 *__restrict__ a, const dev::u32x4 *__restrict__ b,
		dev::u32x4 *__restrict__ c, dev::u32x4 *__restrict__ d, size_t nvec,
		uint32_t *queue)
{
	__shared__ uint32_t slot[3];
	sweep_tiles(queue, slot, nvec, [&](size_t g) {
		dev::u32x4 v = dev::u32x4{(uint32_t)g, 1, 2, 3};
		if (R >= 1) v = __builtin_nontemporal_load(&a[g]);
		if (R >= 2) v += __builtin_nontemporal_load(&b[g]);
		if (W >= 1) __builtin_nontemporal_store(v, &c[g]);
		if (W >= 2) __builtin_nontemporal_store(v + 1, &d[g]);
	});
}

int launch_stream_probe(int reads, int writes, const void *r0, const void *r1,
		void *w0, void *w1, size_t nwords, void *stream, uint32_t *queue)
{
	clear_stale_error();
	const size_t nvec = nwords / 4;
	if (nvec == 0)
		return CORDIC_OK;
	if (queue) {
		const int grid = grid_for((size_t)1024 * kVec, nwords, 1);
		if (grid < 0)
			return CORDIC_ERR_DEVICE;
		auto goq = [&](auto kern) {
			hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0,
				static_cast<hipStream_t>(stream),
				static_cast<const dev::u32x4 *>(r0), static_cast<const dev::u32x4 *>(r1),
				static_cast<dev::u32x4 *>(w0), static_cast<dev::u32x4 *>(w1), nvec, queue);
		};
		if (reads == 0 && writes == 2) goq(stream_probe_queued<0, 2>);
		else if (reads == 1 && writes == 2) goq(stream_probe_queued<1, 2>);
		else if (reads == 2 && writes == 2) goq(stream_probe_queued<2, 2>);
		else if (reads == 1 && writes == 1) goq(stream_probe_queued<1, 1>);
		else if (reads == 2 && writes == 1) goq(stream_probe_queued<2, 1>);
		else return CORDIC_ERR_ARGS;
		return check_launch();
	}
	const size_t blocks = (nvec + 255) / 256;
	if (blocks > 0x7fffffffu)
		return CORDIC_ERR_ARGS;
	const int xcd = (blocks % 8 == 0) ? 1 : 0;
	hipStream_t st = static_cast<hipStream_t>(stream);
	auto go = [&](auto kern) {
		hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st,
			static_cast<const dev::u32x4 *>(r0), static_cast<const dev::u32x4 *>(r1),
			static_cast<dev::u32x4 *>(w0), static_cast<dev::u32x4 *>(w1), nvec, xcd);
	};
	if (reads == 0 && writes == 2) go(stream_probe<0, 2>);
	else if (reads == 1 && writes == 2) go(stream_probe<1, 2>);
	else if (reads == 2 && writes == 2) go(stream_probe<2, 2>);
	else if (reads == 1 && writes == 1) go(stream_probe<1, 1>);
	else if (reads == 2 && writes == 1) go(stream_probe<2, 1>);
	else return CORDIC_ERR_ARGS;
	return check_launch();
}

int launch_digest_u32(const uint32_t *w, size_t n, uint64_t index0,
		uint64_t *digest, void *stream)
{
	clear_stale_error();
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
