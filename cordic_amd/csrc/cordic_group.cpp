// cordic_group.cpp -- multi-GPU jobs behind the C ABI (include/cordic_amd.h,
// "multi-GPU jobs"; BASELINE.json configs[3]).
//
// One host process drives `nlocal` shards, one HIP device each: hipSetDevice,
// a plan of the core bound to that device, a compute stream and a copy stream.
// Shards are contiguous blocks of the GLOBAL sample index and generate their
// own inputs, so the data path has no collective; what crosses devices is
// (a) 8-byte digests, summed on the host, and (b) optionally the results on
// their way to one consumer device, chunk-pipelined behind the compute with
// hipMemcpyPeerAsync (SDMA over xGMI: no CUs, no RCCL kernels competing with
// the CORDIC kernels for the VALUs).  With one process per GPU the same
// forwarding goes over RCCL instead (ncclSend / ncclRecv pairs, RCCL has no
// gather primitive): cordic_group_rccl_init + cordic_group_set_gather_rccl.
// librccl is opened at run time, on first use, so programs that never gather
// across processes do not depend on it.
//
// Placement.  What HBM delivers to a job's streams depends on WHICH allocations
// they run over: pure read or pure write sweeps reach 0.89 of the 8 TB/s peak
// on every 4 GiB hipMalloc alike, but the same 1R2W stream over different
// choices of three out of eight such arrays ranges from 0.75 to 0.82 -- a
// property of the combination (the two written arrays most of all), stable
// for the life of the allocations, not predictable from the virtual addresses
// (round 2's allocation probes, profiles/r02/hbm_placement.txt).  Since the
// group owns its arrays it can choose: when asked to
// (cordic_group_set_placement; off by default since round 6) it allocates two
// more arrays than a shard needs, times the arithmetic-free twin of the job's
// traffic (launch_stream_probe) over the candidate assignments, keeps the best
// and frees the rest.
//
// The reference has nothing of the kind (bench/cpp/cordic_tb.cpp:127-178 steps
// one model from one thread), so there is no reference text to follow here.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>		// types only: the entry points come from dlopen
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "cordic_amd.h"
#include "cordic_internal.h"

using namespace cordic_amd;

namespace {

constexpr int kMaxMarks = 256;
constexpr int kMaxChunks = 64;

struct Shard {
	int	device = 0;
	int	index = 0;		// global shard index
	hipStream_t compute = nullptr, copy = nullptr;
	cordic_plan *plan = nullptr;
	void	*buf[4] = {nullptr, nullptr, nullptr, nullptr};	// in0 in1 out0 out1
	uint64_t cap = 0;		// words allocated per array
	int	inputs = 0;
	// the stretches [first, second) of input array a the caller supplied
	// (cordic_group_write) since the array was last allocated or filled:
	// disjoint, sorted, merged where they touch -- pieces may arrive in ANY
	// order, a job needs [0, its share) inside the first one
	std::vector<std::pair<uint64_t, uint64_t>> written[2];
	uint64_t *d_digest = nullptr;
	hipEvent_t marks[kMaxMarks] = {};
	hipEvent_t piece[kMaxChunks] = {};
	// recorded on `copy` behind the last forwarded piece of a job: the next
	// job's kernels overwrite out0 / out1 and must not overtake those reads
	hipEvent_t copied = nullptr;
	bool	copy_pending = false;
	ncclComm_t comm = nullptr;	// rank = index of total (cordic_group_rccl_init)
	// what the last placement of this shard's arrays saw
	int	place_candidates = 0, place_probes = 0;
	float	place_best_ms = 0.f, place_worst_ms = 0.f;	// the job's full pattern
	float	place_wbest_ms = 0.f, place_wworst_ms = 0.f;	// its written pair (0R2W)
};

constexpr uint64_t kPlaceMinWords = (uint64_t)1 << 24;	// arrays of 64 MiB and up
constexpr int kPlaceSpare = 2;		// candidates beyond the need: set_placement(grp, 1)
constexpr int kPlaceSpareMost = 16;	// ... and what a caller may ask for by number
// a pair of written arrays is fast from 0.93 of the 8 TB/s peak (two arrays of
// one class: <= 0.82); a caller's larger budget is spent only while none is
constexpr double kPlaceGoodBytesPerMs = 0.93 * 8e9;
constexpr int kPlaceTryAgainst = 3;	// a further candidate is tried against this many
constexpr double kPlaceSeconds = 1.5;	// ... and no further one is taken after this long
// ... and in bytes: a tenth of what is free on the device when the arrays are
// allocated (round 6: the library may not transiently claim tens of GiB of a
// caller's HBM; rounds 4-5 took up to 24 spares while no pair of written
// arrays was fast -- allocations come in classes, profiles/r05/pair_matrix.txt
// -- without asking.  A caller who names a NUMBER of spares,
// cordic_group_set_placement(grp, N) with N >= 2, gets the same search bounded
// by N, by this share of the free memory and by kPlaceSeconds)
constexpr double kPlaceSpareFreeShare = 0.10;

// The eight RCCL entry points the gather needs, resolved once per process.
struct Rccl {
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t,
			hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t,
			hipStream_t) = nullptr;
	bool	usable = false;
};

const Rccl &rccl()
{
	static Rccl api;
	static std::once_flag once;
	std::call_once(once, [] {
		// a process that already carries an RCCL (PyTorch ships its own)
		// gets that one back from the soname lookup
		const char *names[] = {std::getenv("CORDIC_RCCL_LIB"), "librccl.so.1",
			"librccl.so", "/opt/rocm/lib/librccl.so.1"};
		void *h = nullptr;
		for (const char *nm : names)
			if (nm && *nm && (h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)))
				break;
		if (!h)
			return;
		auto sym = [&](const char *nm) { return dlsym(h, nm); };
		api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
		api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
		api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
		api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
		api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
		api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
		api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
		api.usable = api.GetUniqueId && api.CommInitRank && api.CommDestroy
			&& api.GroupStart && api.GroupEnd && api.Send && api.Recv;
	});
	return api;
}

// restores the caller's current device on every exit path
struct DeviceScope {
	int prev = -1;
	DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
	~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

bool ok(hipError_t e) { return e == hipSuccess; }

} // namespace

struct cordic_group {
	cordic_config cfg;
	int	first = 0, total = 1;
	std::vector<Shard> shards;
	// result forwarding
	int	root = -1;
	int32_t	*g0 = nullptr, *g1 = nullptr;
	int	chunks = 1;
	int	rroot = -1;	// root SHARD of the RCCL forwarding (-1: off)
	int	placement = 0;		// spare arrays allowed (0: off): cordic_group_set_placement
	// the job size the shards' input arrays were last filled for by the fill
	// kernels (0: not filled), and how many of them; what the CALLER wrote is
	// tracked per shard and array (Shard::written)
	uint64_t filled_total = 0;
	int	filled_inputs = 0;
};

namespace {

// coverage bookkeeping of the caller-written inputs (Shard::written)
typedef std::vector<std::pair<uint64_t, uint64_t>> Spans;

void add_span(Spans &v, uint64_t lo, uint64_t hi)
{
	if (lo >= hi)
		return;
	Spans out;
	bool placed = false;
	for (const auto &iv : v) {
		if (iv.second < lo) {
			out.push_back(iv);		// wholly before, not touching
		} else if (hi < iv.first) {
			if (!placed) {
				out.emplace_back(lo, hi);
				placed = true;
			}
			out.push_back(iv);
		} else {				// overlaps or touches: absorb
			lo = iv.first < lo ? iv.first : lo;
			hi = iv.second > hi ? iv.second : hi;
		}
	}
	if (!placed)
		out.emplace_back(lo, hi);
	v.swap(out);
}

bool covers(const Spans &v, uint64_t cnt)
{
	return cnt == 0 || (!v.empty() && v[0].first == 0 && v[0].second >= cnt);
}

void shard_span(uint64_t n, int s, int total, uint64_t *start, uint64_t *count)
{
	const uint64_t base = n / (uint64_t)total, rem = n % (uint64_t)total;
	const uint64_t us = (uint64_t)s;
	*start = us * base + (us < rem ? us : rem);
	*count = base + (us < rem ? 1 : 0);
}

void release(Shard &s)
{
	if (!ok(hipSetDevice(s.device)))
		return;
	for (void *&p : s.buf) {
		if (p) (void)hipFree(p);
		p = nullptr;
	}
	if (s.d_digest) (void)hipFree(s.d_digest);
	if (s.comm) (void)rccl().CommDestroy(s.comm);
	if (s.plan) cordic_plan_destroy(s.plan);
	for (hipEvent_t &e : s.marks) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t &e : s.piece) if (e) (void)hipEventDestroy(e);
	if (s.copied) (void)hipEventDestroy(s.copied);
	if (s.compute) (void)hipStreamDestroy(s.compute);
	if (s.copy) (void)hipStreamDestroy(s.copy);
	s = Shard{};
}

struct PlaceStats {
	int	candidates = 0, probes = 0;
	float	wbest = 0.f, wworst = 0.f;	// written arrays alone
	float	best = 0.f, worst = 0.f;	// the job's full pattern
};

// Average milliseconds of the `reads`R `writes`W probe over the given arrays
// on `st` (one warm-up launch, two timed), or a negative value.
float probe_ms(hipStream_t st, int reads, int writes, const void *r0, const void *r1,
		void *w0, void *w1, uint64_t words, uint32_t *queue)
{
	hipEvent_t e0 = nullptr, e1 = nullptr;
	float ms = -1.f;
	auto go = [&]() {
		return launch_stream_probe(reads, writes, r0, r1, w0, w1, (size_t)words, st,
			queue) == CORDIC_OK;
	};
	if (ok(hipEventCreate(&e0)) && ok(hipEventCreate(&e1)) && go()
	    && ok(hipEventRecord(e0, st)) && go() && go()
	    && ok(hipEventRecord(e1, st)) && ok(hipEventSynchronize(e1))
	    && ok(hipEventElapsedTime(&ms, e0, e1)))
		ms *= 0.5f;
	else
		ms = -1.f;
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	return ms;
}

// cordic_group_set_placement / CORDIC_GROUP_PLACEMENT: 0 off, 1 the default
// two spares, N >= 2 that many (kPlaceSpareMost at most)
int placement_spares(int enable)
{
	return enable <= 0 ? 0 : enable == 1 ? kPlaceSpare
		: enable > kPlaceSpareMost ? kPlaceSpareMost : enable;
}

// Allocate nread (0..2) + nwrite (1..2) arrays of `words` 32-bit words on the
// current device.  With `tune`, allocate up to kPlaceSpare more than needed,
// time the arithmetic-free twin of the traffic over the candidate assignments
// -- first the written arrays (two: every pair, 0R2W), then the read ones over
// what is left -- keep the best and free the rest.  A probe that cannot run
// just means "in order".
int alloc_placed(hipStream_t st, uint64_t words, int nread, int nwrite, int spares_allowed,
		void **reads, void **writes, PlaceStats *stats)
{
	const bool tune = spares_allowed > 0;
	const auto t_begin = std::chrono::steady_clock::now();
	const size_t need = (size_t)(nread + nwrite);
	const size_t bytes = (size_t)(words ? words : 1) * 4;
	std::vector<void *> pool;
	// spares: min(kPlaceSpare arrays, a tenth of the device's free memory once
	// the needed arrays are there) -- other shards or processes may share the
	// device
	size_t spares = 0;
	auto spare_budget = [&] {
		size_t fr = 0, tot = 0;
		if (!ok(hipMemGetInfo(&fr, &tot))) {
			(void)hipGetLastError();
			return (size_t)0;
		}
		const size_t by_bytes = (size_t)((double)fr * kPlaceSpareFreeShare) / bytes;
		return by_bytes < (size_t)spares_allowed ? by_bytes : (size_t)spares_allowed;
	};
	// (the first round: two spares at most; a larger allowance is drawn on
	// below, one array at a time, only while no written pair is fast)
	const size_t want = need + (tune ? (size_t)kPlaceSpare : 0);
	for (size_t k = 0; k < want; k++) {
		void *p = nullptr;
		if (k == need)
			spares = spare_budget();
		if (k >= need + spares)
			break;
		if (!ok(hipMalloc(&p, bytes))) {
			(void)hipGetLastError();
			if (k >= need)
				break;		// no room for spares: place what there is
			for (void *q : pool) (void)hipFree(q);
			return CORDIC_ERR_DEVICE;
		}
		pool.push_back(p);
	}
	PlaceStats ps;
	// tile counters for the probes (they run in the seeded kernel's work
	// distribution); without them the probes fall back to one-shot tiles
	uint32_t *queue = nullptr;
	if (tune && (!ok(hipMalloc((void **)&queue, CORDIC_QUEUE_BYTES))
			|| !ok(hipMemsetAsync(queue, 0, CORDIC_QUEUE_BYTES, st)))) {
		(void)hipGetLastError();
		if (queue) (void)hipFree(queue);
		queue = nullptr;
	}
	auto take = [&](void **slot, size_t k) {
		*slot = pool[k];
		pool.erase(pool.begin() + (long)k);
	};
	auto finish = [&](bool in_order) {
		if (queue) {
			(void)hipStreamSynchronize(st);
			(void)hipFree(queue);
		}
		if (in_order) {
			for (int i = 0; i < nwrite; i++) if (!writes[i]) take(&writes[i], 0);
			for (int i = 0; i < nread; i++) if (!reads[i]) take(&reads[i], 0);
		}
		for (void *q : pool) (void)hipFree(q);
		// CORDIC_PLACEMENT_DEBUG=1: what the probes saw, and the chosen
		// assignment probed once more after the rest has been freed (freeing
		// the neighbours changes nothing: profiles/r03/state_probe.txt)
		static const bool debug = [] {
			const char *e = std::getenv("CORDIC_PLACEMENT_DEBUG");
			return e && e[0] == '1';
		}();
		if (debug && tune && nwrite == 2) {
			const float after = probe_ms(st, nread, 2, reads[0], reads[1], writes[0],
					writes[1], words, nullptr);
			std::fprintf(stderr, "[cordic placement] candidates %d probes %d best %.3f ms "
				"worst %.3f ms; chosen assignment after the spares were freed: "
				"%.3f ms\n", ps.candidates, ps.probes, ps.best, ps.worst, after);
		}
		if (stats) *stats = ps;
		return CORDIC_OK;
	};
	for (int i = 0; i < nwrite; i++) writes[i] = nullptr;
	for (int i = 0; i < nread; i++) reads[i] = nullptr;
	if (!tune || pool.size() == need)
		return finish(true);
	ps.candidates = (int)pool.size();
	bool failed = false;
	float best = -1.f, worst = 0.f;
	size_t bi = 0, bj = 1;
	if (nwrite == 2) {
		auto try_pair = [&](size_t i, size_t j) {
			const float ms = probe_ms(st, 0, 2, nullptr, nullptr, pool[i], pool[j], words, queue);
			if (ms < 0.f) { failed = true; return; }
			ps.probes++;
			if (best < 0.f || ms < best) { best = ms; bi = i; bj = j; }
			if (ms > worst) worst = ms;
		};
		for (size_t i = 0; i < pool.size() && !failed; i++)
			for (size_t j = i + 1; j < pool.size() && !failed; j++)
				try_pair(i, j);
		// A caller that allowed more than kPlaceSpare spares: while no pair is
		// fast (all candidates of one class: DESIGN.md section 3), one more
		// array at a time, tried against a few of those at hand -- within the
		// caller's number, the share of the free memory and kPlaceSeconds.
		const float good = (float)((double)words * 8.0 / kPlaceGoodBytesPerMs);
		while (!failed && best > good && pool.size() < need + spares) {
			const std::chrono::duration<double> spent =
				std::chrono::steady_clock::now() - t_begin;
			if (spent.count() > kPlaceSeconds)
				break;
			void *p = nullptr;
			if (!ok(hipMalloc(&p, bytes))) {
				(void)hipGetLastError();
				break;		// no room: make do with what there is
			}
			pool.push_back(p);
			ps.candidates++;
			const size_t last = pool.size() - 1;
			const size_t step = last > (size_t)kPlaceTryAgainst
				? last / (size_t)kPlaceTryAgainst : 1;
			for (size_t i = 0; i < last && !failed && best > good; i += step)
				try_pair(i, last);
		}
		if (failed) {
			(void)hipGetLastError();
			ps = PlaceStats{};
			return finish(true);
		}
		take(&writes[1], bj);	// the larger index first: bi stays valid
		take(&writes[0], bi);
		ps.wbest = ps.best = best;
		ps.wworst = ps.worst = worst;
	} else {
		// one written array: nothing to pair it with yet; it is chosen
		// together with the read arrays below (or taken as it comes)
		if (nread == 0)
			return finish(true);
	}
	if (nread == 0)
		return finish(false);
	best = -1.f; worst = 0.f; bi = 0; bj = 1;
	size_t bw = 0;
	if (nwrite == 1) {
		// (read array[s], written array) together: every ordered choice
		for (size_t w = 0; w < pool.size() && !failed; w++)
			for (size_t i = 0; i < pool.size() && !failed; i++) {
				if (i == w) continue;
				for (size_t j = (nread == 2 ? i + 1 : i); j < (nread == 2 ? pool.size() : i + 1)
						&& !failed; j++) {
					if (j == w) continue;
					const float ms = probe_ms(st, nread, 1, pool[i],
						nread == 2 ? pool[j] : nullptr, pool[w], nullptr, words, queue);
					if (ms < 0.f) { failed = true; break; }
					ps.probes++;
					if (best < 0.f || ms < best) { best = ms; bw = w; bi = i; bj = j; }
					if (ms > worst) worst = ms;
				}
			}
		if (failed) {
			(void)hipGetLastError();
			ps = PlaceStats{};
			return finish(true);
		}
		void *pw = pool[bw], *p0 = pool[bi], *p1 = nread == 2 ? pool[bj] : nullptr;
		writes[0] = pw;
		reads[0] = p0;
		if (nread == 2) reads[1] = p1;
		for (size_t k = pool.size(); k-- > 0;)
			if (pool[k] == pw || pool[k] == p0 || (p1 && pool[k] == p1))
				pool.erase(pool.begin() + (long)k);
	} else if (nread == 1) {
		for (size_t i = 0; i < pool.size() && !failed; i++) {
			const float ms = probe_ms(st, 1, 2, pool[i], nullptr, writes[0], writes[1], words, queue);
			if (ms < 0.f) { failed = true; break; }
			ps.probes++;
			if (best < 0.f || ms < best) { best = ms; bi = i; }
			if (ms > worst) worst = ms;
		}
		if (failed) { (void)hipGetLastError(); return finish(true); }
		take(&reads[0], bi);
	} else {
		for (size_t i = 0; i < pool.size() && !failed; i++)
			for (size_t j = i + 1; j < pool.size() && !failed; j++) {
				const float ms = probe_ms(st, 2, 2, pool[i], pool[j], writes[0], writes[1], words, queue);
				if (ms < 0.f) { failed = true; break; }
				ps.probes++;
				if (best < 0.f || ms < best) { best = ms; bi = i; bj = j; }
				if (ms > worst) worst = ms;
			}
		if (failed) { (void)hipGetLastError(); return finish(true); }
		take(&reads[1], bj);
		take(&reads[0], bi);
	}
	ps.best = best;
	ps.worst = worst;
	return finish(false);
}

int ensure(cordic_group *g, uint64_t n_total, int inputs)
{
	for (Shard &s : g->shards) {
		uint64_t start, cnt;
		shard_span(n_total, s.index, g->total, &start, &cnt);
		if (cnt <= s.cap && inputs <= s.inputs)
			continue;
		if (!ok(hipSetDevice(s.device)))
			return CORDIC_ERR_DEVICE;
		// earlier jobs may still be running on the old arrays
		if (!ok(hipStreamSynchronize(s.compute)) || !ok(hipStreamSynchronize(s.copy)))
			return CORDIC_ERR_DEVICE;
		const uint64_t cap = cnt > s.cap ? cnt : s.cap;
		const int nin = inputs > s.inputs ? inputs : s.inputs;
		if (cap > s.cap) {
			for (void *&p : s.buf) {
				if (p) (void)hipFree(p);
				p = nullptr;
			}
			// nothing is allocated until the calls below succeed, and
			// whatever the inputs held is gone
			s.cap = 0;
			s.inputs = 0;
			s.written[0].clear();
			s.written[1].clear();
			g->filled_total = 0;
			g->filled_inputs = 0;
		}
		s.place_candidates = s.place_probes = 0;
		s.place_best_ms = s.place_worst_ms = 0.f;
		s.place_wbest_ms = s.place_wworst_ms = 0.f;
		if (!s.buf[2] && !s.buf[3]) {
			// everything at once: place the arrays by measurement
			void *rd[2] = {nullptr, nullptr}, *wr[2] = {nullptr, nullptr};
			PlaceStats ps;
			const int tune = cap >= kPlaceMinWords ? g->placement : 0;
			if (int rc = alloc_placed(s.compute, cap, nin, 2, tune, rd, wr, &ps))
				return rc;
			s.buf[0] = rd[0]; s.buf[1] = rd[1];
			s.buf[2] = wr[0]; s.buf[3] = wr[1];
			s.place_candidates = ps.candidates;
			s.place_probes = ps.probes;
			s.place_best_ms = ps.best; s.place_worst_ms = ps.worst;
			s.place_wbest_ms = ps.wbest; s.place_wworst_ms = ps.wworst;
		} else {
			// inputs added behind a job whose results may still be wanted:
			// no probing (it writes), the arrays as they come
			for (int a = 0; a < nin; a++)
				if (!s.buf[a] && !ok(hipMalloc(&s.buf[a], (cap ? cap : 1) * 4)))
					return CORDIC_ERR_DEVICE;
		}
		s.cap = cap;
		s.inputs = nin;
	}
	return CORDIC_OK;
}

// piece c of `chunks` of a shard of cnt samples: pieces start on 4096-sample
// boundaries of the shard, so the vector kernels keep their 16-byte accesses
bool piece_span(uint64_t cnt, int chunks, int c, uint64_t *a, uint64_t *len)
{
	uint64_t psize = (cnt + (uint64_t)chunks - 1) / (uint64_t)chunks;
	psize = (psize + 4095) & ~(uint64_t)4095;
	*a = (uint64_t)c * psize;
	if (*a >= cnt)
		return false;
	*len = (cnt - *a < psize) ? cnt - *a : psize;
	return true;
}

// Piece c of every shard of the job travels to the root shard: each local
// shard sends its own, the root shard posts the matching receives for all
// `total` shards (every rank derives the same piece geometry from n_total),
// all in one RCCL group, on the copy streams.
int rccl_forward(cordic_group *g, uint64_t n_total, int c)
{
	const Rccl &api = rccl();
	bool fine = api.GroupStart() == ncclSuccess;
	for (Shard &s : g->shards) {
		uint64_t start, cnt, a, len;
		shard_span(n_total, s.index, g->total, &start, &cnt);
		fine = fine && ok(hipSetDevice(s.device));	// the communicator's device
		if (fine && piece_span(cnt, g->chunks, c, &a, &len)) {
			fine = api.Send(static_cast<int32_t *>(s.buf[2]) + a, (size_t)len,
					ncclInt32, g->rroot, s.comm, s.copy) == ncclSuccess
				&& api.Send(static_cast<int32_t *>(s.buf[3]) + a, (size_t)len,
					ncclInt32, g->rroot, s.comm, s.copy) == ncclSuccess;
		}
		if (s.index != g->rroot)
			continue;
		for (int r = 0; r < g->total && fine; r++) {
			shard_span(n_total, r, g->total, &start, &cnt);
			if (!piece_span(cnt, g->chunks, c, &a, &len))
				continue;
			fine = api.Recv(g->g0 + start + a, (size_t)len, ncclInt32, r,
					s.comm, s.copy) == ncclSuccess
				&& api.Recv(g->g1 + start + a, (size_t)len, ncclInt32, r,
					s.comm, s.copy) == ncclSuccess;
		}
	}
	// always close the group that was opened
	return (api.GroupEnd() == ncclSuccess && fine) ? CORDIC_OK : CORDIC_ERR_DEVICE;
}

// Run `launch(shard, offset, count)` over every local shard, whole or -- with
// forwarding set -- piece by piece, each piece followed by its copies.
template <typename F>
int for_each_piece(cordic_group *g, uint64_t n_total, int inputs, F launch)
{
	if (!g)
		return CORDIC_ERR_ARGS;
	DeviceScope scope;
	if (int rc = ensure(g, n_total, inputs))
		return rc;
	// jobs that read input arrays need EVERY one of them, on every local
	// shard, either filled for THIS job size or written by the caller over
	// the shard's whole share
	for (const Shard &s : g->shards) {
		uint64_t start, cnt;
		shard_span(n_total, s.index, g->total, &start, &cnt);
		for (int a = 0; a < inputs; a++) {
			const bool filled = g->filled_total == n_total && a < g->filled_inputs;
			if (!filled && !covers(s.written[a], cnt))
				return CORDIC_ERR_ARGS;
		}
	}
	const bool forward = g->root >= 0 || g->rroot >= 0;
	const int chunks = forward ? g->chunks : 1;
	for (Shard &s : g->shards) {
		if (!s.copy_pending)
			continue;
		// the previous job's pieces are still being read by the copy
		// engines / RCCL: this job's kernels write the same arrays
		// (CORDIC_FAULT_SKIP_JOB_ORDER: fault injection, never defined in a
		// product build -- tools/fault_build.sh makes cordic_amd/lib_fault.so
		// with it to prove that the back-to-back tests over the asynchronous
		// RCCL stand-in DO fail without this wait)
#ifndef CORDIC_FAULT_SKIP_JOB_ORDER
		if (!ok(hipSetDevice(s.device)) ||
		    !ok(hipStreamWaitEvent(s.compute, s.copied, 0)))
			return CORDIC_ERR_DEVICE;
#endif
		s.copy_pending = false;
	}
	// piece-major, so that every device has work queued before the first
	// copy is issued
	for (int c = 0; c < chunks; c++) {
		for (Shard &s : g->shards) {
			uint64_t start, cnt, a, len;
			shard_span(n_total, s.index, g->total, &start, &cnt);
			if (!piece_span(cnt, chunks, c, &a, &len))
				continue;
			if (!ok(hipSetDevice(s.device)))
				return CORDIC_ERR_DEVICE;
			if (int rc = launch(s, start, a, len))
				return rc;
			if (!forward)
				continue;
			if (!s.piece[c] && !ok(hipEventCreateWithFlags(&s.piece[c],
					hipEventDisableTiming)))
				return CORDIC_ERR_DEVICE;
			if (!ok(hipEventRecord(s.piece[c], s.compute)) ||
			    !ok(hipStreamWaitEvent(s.copy, s.piece[c], 0)))
				return CORDIC_ERR_DEVICE;
			if (g->root < 0)
				continue;
			const size_t bytes = (size_t)len * 4;
			int32_t *o0 = static_cast<int32_t *>(s.buf[2]) + a;
			int32_t *o1 = static_cast<int32_t *>(s.buf[3]) + a;
			if (!ok(hipMemcpyPeerAsync(g->g0 + start + a, g->root, o0,
					s.device, bytes, s.copy)) ||
			    !ok(hipMemcpyPeerAsync(g->g1 + start + a, g->root, o1,
					s.device, bytes, s.copy)))
				return CORDIC_ERR_DEVICE;
		}
		if (g->rroot >= 0)
			if (int rc = rccl_forward(g, n_total, c))
				return rc;
	}
	if (forward)
		for (Shard &s : g->shards) {
			if (!ok(hipSetDevice(s.device)))
				return CORDIC_ERR_DEVICE;
			if (!s.copied && !ok(hipEventCreateWithFlags(&s.copied,
					hipEventDisableTiming)))
				return CORDIC_ERR_DEVICE;
			if (!ok(hipEventRecord(s.copied, s.copy)))
				return CORDIC_ERR_DEVICE;
			s.copy_pending = true;
		}
	return CORDIC_OK;
}

} // namespace

extern "C" {

int cordic_device_count(void)
{
	int n = 0;
	if (!ok(hipGetDeviceCount(&n))) {
		(void)hipGetLastError();
		return CORDIC_ERR_DEVICE;
	}
	return n;
}

int cordic_shard_range(uint64_t n_total, int shard, int total_shards,
		uint64_t *start, uint64_t *count)
{
	if (total_shards < 1 || shard < 0 || shard >= total_shards || !start
			|| !count)
		return CORDIC_ERR_ARGS;
	shard_span(n_total, shard, total_shards, start, count);
	return CORDIC_OK;
}

void cordic_group_destroy(cordic_group *grp)
{
	if (!grp)
		return;
	DeviceScope scope;
	for (Shard &s : grp->shards)
		release(s);
	delete grp;
}

int cordic_group_create(const cordic_config *cfg, int nlocal, const int *devices,
		int first_shard, int total_shards, cordic_group **out)
{
	if (!cfg || !out || nlocal < 1 || total_shards < 1 || first_shard < 0
			|| first_shard + nlocal > total_shards)
		return CORDIC_ERR_ARGS;
	const int ndev = cordic_device_count();
	if (ndev <= 0)
		return CORDIC_ERR_DEVICE;
	for (int i = 0; i < nlocal; i++) {
		const int d = devices ? devices[i] : i;
		if (d < 0 || d >= ndev)
			return CORDIC_ERR_DEVICE;	// fewer GPUs than shards asked for
	}
	cordic_group *g = new (std::nothrow) cordic_group;
	if (!g)
		return CORDIC_ERR_NOMEM;
	g->cfg = *cfg;
	g->first = first_shard;
	g->total = total_shards;
	// off unless asked: cordic_group_set_placement, or CORDIC_GROUP_PLACEMENT=1
	if (const char *e = std::getenv("CORDIC_GROUP_PLACEMENT"))
		g->placement = placement_spares(std::atoi(e));
	DeviceScope scope;
	g->shards.resize((size_t)nlocal);
	int rc = CORDIC_OK;
	for (int i = 0; i < nlocal && rc == CORDIC_OK; i++) {
		Shard &s = g->shards[(size_t)i];
		s.device = devices ? devices[i] : i;
		s.index = first_shard + i;
		if (!ok(hipSetDevice(s.device)) ||
		    !ok(hipStreamCreateWithFlags(&s.compute, hipStreamNonBlocking)) ||
		    !ok(hipStreamCreateWithFlags(&s.copy, hipStreamNonBlocking)) ||
		    !ok(hipMalloc((void **)&s.d_digest, 8))) {
			rc = CORDIC_ERR_DEVICE;
			break;
		}
		// the "generation" step of the core, once per device
		rc = cordic_plan_create(cfg, &s.plan);
	}
	if (rc != CORDIC_OK) {
		cordic_group_destroy(g);
		return rc;
	}
	*out = g;
	return CORDIC_OK;
}

int cordic_group_size(const cordic_group *grp)
{
	return grp ? (int)grp->shards.size() : CORDIC_ERR_ARGS;
}

int cordic_group_range(const cordic_group *grp, uint64_t n_total, int shard,
		uint64_t *start, uint64_t *count)
{
	if (!grp || shard < 0 || shard >= grp->total || !start || !count)
		return CORDIC_ERR_ARGS;
	shard_span(n_total, shard, grp->total, start, count);
	return CORDIC_OK;
}

int cordic_group_reserve(cordic_group *grp, uint64_t n_total, int inputs)
{
	if (!grp || inputs < 0 || inputs > 2)
		return CORDIC_ERR_ARGS;
	DeviceScope scope;
	return ensure(grp, n_total, inputs);
}

int cordic_arrays_alloc(size_t bytes, int n_read, int n_write, void **ptrs,
		void *stream)
{
	if (!ptrs || n_read < 0 || n_read > 2 || n_write < 1 || n_write > 2)
		return CORDIC_ERR_ARGS;
	const uint64_t words = ((uint64_t)bytes + 3) / 4;
	// plain hipMalloc unless CORDIC_GROUP_PLACEMENT=1 asks for the probes
	int tune = 0;
	if (const char *e = std::getenv("CORDIC_GROUP_PLACEMENT"))
		tune = words >= kPlaceMinWords ? placement_spares(std::atoi(e)) : 0;
	void *rd[2] = {nullptr, nullptr}, *wr[2] = {nullptr, nullptr};
	if (int rc = alloc_placed(static_cast<hipStream_t>(stream), words, n_read, n_write,
			tune, rd, wr, nullptr))
		return rc;
	if (!ok(hipStreamSynchronize(static_cast<hipStream_t>(stream))))
		return CORDIC_ERR_DEVICE;
	for (int i = 0; i < n_read; i++) ptrs[i] = rd[i];
	for (int i = 0; i < n_write; i++) ptrs[n_read + i] = wr[i];
	return CORDIC_OK;
}

void cordic_arrays_free(void **ptrs, int count)
{
	if (!ptrs)
		return;
	for (int i = 0; i < count; i++) {
		if (ptrs[i]) (void)hipFree(ptrs[i]);
		ptrs[i] = nullptr;
	}
}

int cordic_group_set_placement(cordic_group *grp, int enable)
{
	if (!grp)
		return CORDIC_ERR_ARGS;
	grp->placement = placement_spares(enable);
	return CORDIC_OK;
}

int cordic_group_placement(const cordic_group *grp, int local_shard, int *candidates,
		int *probes, float *written_best_ms, float *written_worst_ms,
		float *best_ms, float *worst_ms)
{
	if (!grp || local_shard < 0 || local_shard >= (int)grp->shards.size())
		return CORDIC_ERR_ARGS;
	const Shard &s = grp->shards[(size_t)local_shard];
	if (candidates) *candidates = s.place_candidates;
	if (probes) *probes = s.place_probes;
	if (written_best_ms) *written_best_ms = s.place_wbest_ms;
	if (written_worst_ms) *written_worst_ms = s.place_wworst_ms;
	if (best_ms) *best_ms = s.place_best_ms;
	if (worst_ms) *worst_ms = s.place_worst_ms;
	return CORDIC_OK;
}

int cordic_group_fill_phase_ramp(cordic_group *grp, uint64_t n_total, int shift)
{
	if (!grp)
		return CORDIC_ERR_ARGS;
	DeviceScope scope;
	if (int rc = ensure(grp, n_total, 1))
		return rc;
	for (Shard &s : grp->shards) {
		uint64_t start, cnt;
		shard_span(n_total, s.index, grp->total, &start, &cnt);
		if (!ok(hipSetDevice(s.device)))
			return CORDIC_ERR_DEVICE;
		if (int rc = cordic_fill_phase_ramp(static_cast<uint32_t *>(s.buf[0]),
				(size_t)cnt, start, shift, s.compute))
			return rc;
	}
	for (Shard &s : grp->shards)
		s.written[0].clear();		// overwritten by the ramp
	grp->filled_total = n_total;
	grp->filled_inputs = 1;
	return CORDIC_OK;
}

int cordic_group_fill_iq_ramp(cordic_group *grp, uint64_t n_total, uint32_t mulx,
		uint32_t muly, int bits)
{
	if (!grp)
		return CORDIC_ERR_ARGS;
	DeviceScope scope;
	if (int rc = ensure(grp, n_total, 2))
		return rc;
	for (Shard &s : grp->shards) {
		uint64_t start, cnt;
		shard_span(n_total, s.index, grp->total, &start, &cnt);
		if (!ok(hipSetDevice(s.device)))
			return CORDIC_ERR_DEVICE;
		if (int rc = cordic_fill_iq_ramp(static_cast<int32_t *>(s.buf[0]),
				static_cast<int32_t *>(s.buf[1]), (size_t)cnt, start,
				mulx, muly, bits, s.compute))
			return rc;
	}
	for (Shard &s : grp->shards) {
		s.written[0].clear();
		s.written[1].clear();
	}
	grp->filled_total = n_total;
	grp->filled_inputs = 2;
	return CORDIC_OK;
}

int cordic_group_p2r_const(cordic_group *grp, uint64_t n_total, int32_t xval,
		int32_t yval)
{
	return for_each_piece(grp, n_total, 1,
		[&](Shard &s, uint64_t, uint64_t a, uint64_t len) {
			return cordic_plan_p2r_const(s.plan, (size_t)len, xval, yval,
				static_cast<const uint32_t *>(s.buf[0]) + a,
				static_cast<int32_t *>(s.buf[2]) + a,
				static_cast<int32_t *>(s.buf[3]) + a, s.compute);
		});
}

int cordic_group_nco(cordic_group *grp, uint64_t n_total, uint32_t phase0,
		uint32_t fcw, int32_t xval, int32_t yval)
{
	return for_each_piece(grp, n_total, 0,
		[&](Shard &s, uint64_t start, uint64_t a, uint64_t len) {
			return cordic_plan_nco(s.plan, (size_t)len, phase0, fcw,
				start + a, xval, yval,
				static_cast<int32_t *>(s.buf[2]) + a,
				static_cast<int32_t *>(s.buf[3]) + a, s.compute);
		});
}

int cordic_group_r2p(cordic_group *grp, uint64_t n_total)
{
	return for_each_piece(grp, n_total, 2,
		[&](Shard &s, uint64_t, uint64_t a, uint64_t len) {
			return cordic_r2p(cordic_plan_config(s.plan), (size_t)len,
				static_cast<const int32_t *>(s.buf[0]) + a,
				static_cast<const int32_t *>(s.buf[1]) + a,
				static_cast<int32_t *>(s.buf[2]) + a,
				static_cast<uint32_t *>(s.buf[3]) + a, s.compute);
		});
}

int cordic_group_sync(cordic_group *grp)
{
	if (!grp)
		return CORDIC_ERR_ARGS;
	DeviceScope scope;
	for (Shard &s : grp->shards) {
		if (!ok(hipSetDevice(s.device)) ||
		    !ok(hipStreamSynchronize(s.compute)) ||
		    !ok(hipStreamSynchronize(s.copy)))
			return CORDIC_ERR_DEVICE;
	}
	return CORDIC_OK;
}

int cordic_group_digest(cordic_group *grp, uint64_t n_total, uint64_t *digest)
{
	if (!grp || !digest)
		return CORDIC_ERR_ARGS;
	DeviceScope scope;
	for (Shard &s : grp->shards) {
		uint64_t start, cnt;
		shard_span(n_total, s.index, grp->total, &start, &cnt);
		if (cnt > s.cap)
			return CORDIC_ERR_ARGS;		// no such job has run
		if (!ok(hipSetDevice(s.device)) ||
		    !ok(hipMemsetAsync(s.d_digest, 0, 8, s.compute)))
			return CORDIC_ERR_DEVICE;
		int rc = cordic_digest_u32(static_cast<const uint32_t *>(s.buf[2]),
				(size_t)cnt, start, s.d_digest, s.compute);
		if (rc == CORDIC_OK)
			rc = cordic_digest_u32(static_cast<const uint32_t *>(s.buf[3]),
				(size_t)cnt, start + ((uint64_t)1 << 40), s.d_digest,
				s.compute);
		if (rc != CORDIC_OK)
			return rc;
	}
	uint64_t sum = 0;
	for (Shard &s : grp->shards) {
		uint64_t v = 0;
		if (!ok(hipSetDevice(s.device)) ||
		    !ok(hipStreamSynchronize(s.compute)) ||
		    !ok(hipMemcpy(&v, s.d_digest, 8, hipMemcpyDeviceToHost)))
			return CORDIC_ERR_DEVICE;
		sum += v;			// shards add, mod 2^64
	}
	*digest = sum;
	return CORDIC_OK;
}

int cordic_group_set_gather(cordic_group *grp, int root_device, int32_t *d_out0,
		int32_t *d_out1, int chunks)
{
	if (!grp)
		return CORDIC_ERR_ARGS;
	if (root_device < 0) {
		grp->root = grp->rroot = -1;
		grp->g0 = grp->g1 = nullptr;
		grp->chunks = 1;
		return CORDIC_OK;
	}
	if (!d_out0 || !d_out1 || chunks < 1 || chunks > kMaxChunks
			|| root_device >= cordic_device_count())
		return CORDIC_ERR_ARGS;
	grp->rroot = -1;
	DeviceScope scope;
	// let the copy engines move device to device directly over xGMI; where
	// peer access cannot be enabled hipMemcpyPeerAsync still works (staged)
	for (Shard &s : grp->shards) {
		if (s.device == root_device)
			continue;
		int can = 0;
		if (ok(hipDeviceCanAccessPeer(&can, s.device, root_device)) && can
				&& ok(hipSetDevice(s.device))) {
			const hipError_t e = hipDeviceEnablePeerAccess(root_device, 0);
			if (e != hipSuccess)
				(void)hipGetLastError();	// already enabled is fine
		}
	}
	grp->root = root_device;
	grp->g0 = d_out0;
	grp->g1 = d_out1;
	grp->chunks = chunks;
	return CORDIC_OK;
}

int cordic_rccl_unique_id(void *id)
{
	if (!id)
		return CORDIC_ERR_ARGS;
	const Rccl &api = rccl();
	if (!api.usable)
		return CORDIC_ERR_UNSUPPORTED;	// no librccl in this process / image
	ncclUniqueId u;
	if (api.GetUniqueId(&u) != ncclSuccess)
		return CORDIC_ERR_DEVICE;
	static_assert(sizeof u == CORDIC_RCCL_ID_BYTES, "ncclUniqueId size");
	std::memcpy(id, &u, sizeof u);
	return CORDIC_OK;
}

int cordic_group_rccl_init(cordic_group *grp, const void *id)
{
	if (!grp || !id)
		return CORDIC_ERR_ARGS;
	const Rccl &api = rccl();
	if (!api.usable)
		return CORDIC_ERR_UNSUPPORTED;
	for (const Shard &s : grp->shards)
		if (s.comm)
			return CORDIC_ERR_ARGS;	// once per group
	ncclUniqueId u;
	std::memcpy(&u, id, sizeof u);
	DeviceScope scope;
	// one communicator per local shard, rank = global shard index; several
	// in one process have to be created inside one RCCL group
	bool fine = api.GroupStart() == ncclSuccess;
	for (Shard &s : grp->shards) {
		if (!fine)
			break;
		fine = ok(hipSetDevice(s.device)) &&
			api.CommInitRank(&s.comm, grp->total, u, s.index) == ncclSuccess;
	}
	if (api.GroupEnd() != ncclSuccess || !fine) {
		for (Shard &s : grp->shards) {
			if (s.comm) (void)api.CommDestroy(s.comm);
			s.comm = nullptr;
		}
		return CORDIC_ERR_DEVICE;
	}
	return CORDIC_OK;
}

int cordic_group_set_gather_rccl(cordic_group *grp, int root_shard,
		int32_t *d_out0, int32_t *d_out1, int chunks)
{
	if (!grp)
		return CORDIC_ERR_ARGS;
	if (root_shard < 0)
		return cordic_group_set_gather(grp, -1, nullptr, nullptr, 1);
	if (root_shard >= grp->total || chunks < 1 || chunks > kMaxChunks)
		return CORDIC_ERR_ARGS;
	bool holds_root = false;
	for (const Shard &s : grp->shards) {
		if (!s.comm)
			return CORDIC_ERR_ARGS;	// cordic_group_rccl_init first
		holds_root = holds_root || s.index == root_shard;
	}
	if (holds_root && (!d_out0 || !d_out1))
		return CORDIC_ERR_ARGS;
	grp->root = -1;
	grp->rroot = root_shard;
	grp->g0 = d_out0;
	grp->g1 = d_out1;
	grp->chunks = chunks;
	return CORDIC_OK;
}

int cordic_group_mark(cordic_group *grp, int slot)
{
	if (!grp || slot < 0 || slot >= kMaxMarks)
		return CORDIC_ERR_ARGS;
	DeviceScope scope;
	for (Shard &s : grp->shards) {
		if (!ok(hipSetDevice(s.device)))
			return CORDIC_ERR_DEVICE;
		if (!s.marks[slot] && !ok(hipEventCreate(&s.marks[slot])))
			return CORDIC_ERR_DEVICE;
		if (!ok(hipEventRecord(s.marks[slot], s.compute)))
			return CORDIC_ERR_DEVICE;
	}
	return CORDIC_OK;
}

int cordic_group_elapsed(cordic_group *grp, int slot_a, int slot_b, float *max_ms,
		float *per_shard_ms)
{
	if (!grp || !max_ms || slot_a < 0 || slot_a >= kMaxMarks || slot_b < 0
			|| slot_b >= kMaxMarks)
		return CORDIC_ERR_ARGS;
	DeviceScope scope;
	float worst = 0.f;
	for (size_t i = 0; i < grp->shards.size(); i++) {
		Shard &s = grp->shards[i];
		if (!s.marks[slot_a] || !s.marks[slot_b])
			return CORDIC_ERR_ARGS;
		float ms = 0.f;
		if (!ok(hipSetDevice(s.device)) ||
		    !ok(hipEventSynchronize(s.marks[slot_b])) ||
		    !ok(hipEventElapsedTime(&ms, s.marks[slot_a], s.marks[slot_b])))
			return CORDIC_ERR_DEVICE;
		if (per_shard_ms)
			per_shard_ms[i] = ms;
		if (ms > worst)
			worst = ms;
	}
	*max_ms = worst;
	return CORDIC_OK;
}

int cordic_group_buffers(const cordic_group *grp, int local_shard, int *device,
		void **in0, void **in1, void **out0, void **out1, uint64_t *count)
{
	if (!grp || local_shard < 0 || local_shard >= (int)grp->shards.size())
		return CORDIC_ERR_ARGS;
	const Shard &s = grp->shards[(size_t)local_shard];
	if (device) *device = s.device;
	if (in0) *in0 = s.buf[0];
	if (in1) *in1 = s.buf[1];
	if (out0) *out0 = s.buf[2];
	if (out1) *out1 = s.buf[3];
	if (count) *count = s.cap;
	return CORDIC_OK;
}

int cordic_group_read(cordic_group *grp, int local_shard, int array,
		uint64_t offset, uint64_t count, void *host_dst)
{
	if (!grp || local_shard < 0 || local_shard >= (int)grp->shards.size()
			|| array < 0 || array > 3 || !host_dst)
		return CORDIC_ERR_ARGS;
	Shard &s = grp->shards[(size_t)local_shard];
	if (!s.buf[array] || offset > s.cap || count > s.cap - offset)
		return CORDIC_ERR_ARGS;
	if (count == 0)
		return CORDIC_OK;
	DeviceScope scope;
	if (!ok(hipSetDevice(s.device)) ||
	    !ok(hipStreamSynchronize(s.compute)) ||
	    !ok(hipMemcpy(host_dst, static_cast<const uint32_t *>(s.buf[array])
			+ offset, (size_t)count * 4, hipMemcpyDefault)))
		return CORDIC_ERR_DEVICE;
	return CORDIC_OK;
}

int cordic_group_write(cordic_group *grp, int local_shard, int array,
		uint64_t offset, uint64_t count, const void *src)
{
	if (!grp || local_shard < 0 || local_shard >= (int)grp->shards.size()
			|| array < 0 || array > 3 || !src)
		return CORDIC_ERR_ARGS;
	Shard &s = grp->shards[(size_t)local_shard];
	if (!s.buf[array] || offset > s.cap || count > s.cap - offset)
		return CORDIC_ERR_ARGS;
	if (count == 0)
		return CORDIC_OK;
	DeviceScope scope;
	if (!ok(hipSetDevice(s.device)) ||
	    !ok(hipStreamSynchronize(s.compute)) ||
	    !ok(hipStreamSynchronize(s.copy)) ||
	    !ok(hipMemcpy(static_cast<uint32_t *>(s.buf[array]) + offset, src,
			(size_t)count * 4, hipMemcpyDefault)))
		return CORDIC_ERR_DEVICE;
	// the caller's own inputs: any order, as long as the pieces end up
	// covering the shard's share (a prefix counter rejected a
	// shard written completely but back to front)
	if (array < 2)
		add_span(s.written[array], offset, offset + count);
	return CORDIC_OK;
}

} // extern "C"
