// cordic_abi.cpp -- the extern "C" surface declared in include/cordic_amd.h.
// Thin by design: argument checks, then the host layer (cordic_config.cpp) or
// the device launchers (cordic_kernels.hip).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "cordic_amd.h"
#include "cordic_internal.h"

using namespace cordic_amd;

// ------------------------------------------------------------------- plans
// Tile queues of the seeded kernel (CORDIC_QUEUE_BYTES of device counters,
// zeroed once here and left zeroed by every kernel that used them).  Two
// launches must never share a block of counters while either is running, so a
// slot is handed out again only once the launch that used it has COMPLETED:
//   - eager launches draw from slots [0, kEagerSlots) round robin: each slot
//     carries an event recorded right behind its kernel; a launch that gets a
//     slot whose previous user may still be running is ordered behind it on
//     the device (same stream: nothing to do; other stream: the stream waits
//     for the event) -- the host never waits and may run ahead of the GPU by
//     any number of launches;
//   - a launch issued while its stream is being CAPTURED keeps its slot baked
//     into the graph node and may be replayed at any later time, so it takes a
//     slot from [kEagerSlots, kQueueSlots) that is never handed out again
//     (at most kQueueSlots - kEagerSlots = 208 captured launches per handle
//     for its lifetime; further ones run the static sweep, -5...-8 %, and are
//     counted: cordic_*_queue_info).  A graph exec never runs concurrently
//     with itself, so one slot per captured node is enough; two execs
//     instantiated from the SAME captured graph share the node's slot and
//     must not run concurrently (include/cordic_amd.h says so).
// Stream identity is never taken from the handle's address (a destroyed
// stream's address can be reused): a slot whose event has not completed is
// always waited for on the device, whatever stream asks.
constexpr unsigned kQueueSlots = 256;
constexpr unsigned kEagerSlots = 48;

struct QueueRing {
	enum State : unsigned char { FREE, CLAIMED, RECORDED, RETIRED };
	uint32_t *d = nullptr;		// kQueueSlots x CORDIC_QUEUE_BYTES
	mutable std::mutex mu;
	mutable hipEvent_t ev[kEagerSlots] = {};
	mutable State state[kQueueSlots] = {};
	mutable State prev[kEagerSlots] = {};	// state before the pending claim
	mutable unsigned next = 0, next_captured = kEagerSlots;
	mutable unsigned long long fallbacks = 0;	// launches that got no slot

	bool alloc()
	{
		const size_t bytes = (size_t)kQueueSlots * CORDIC_QUEUE_BYTES;
		if (hipMalloc((void **)&d, bytes) != hipSuccess)
			return false;
		if (hipMemset(d, 0, bytes) != hipSuccess) {
			release();
			return false;
		}
		for (unsigned k = 0; k < kEagerSlots; k++)
			if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) {
				release();
				return false;
			}
		return true;
	}
	void info(cordic_queue_info *out) const
	{
		std::lock_guard<std::mutex> lock(mu);
		out->eager_slots = d ? (int32_t)kEagerSlots : 0;
		out->captured_capacity = d ? (int32_t)(kQueueSlots - kEagerSlots) : 0;
		out->captured_used = (int32_t)(next_captured - kEagerSlots);
		out->fallback_launches = fallbacks;
	}
	void release()
	{
		for (unsigned k = 0; k < kEagerSlots; k++)
			if (ev[k]) {
				(void)hipEventDestroy(ev[k]);
				ev[k] = nullptr;
			}
		if (d) (void)hipFree(d);
		d = nullptr;
	}
	uint32_t *ptr(int slot) const
	{
		return slot < 0 ? nullptr
			: d + (size_t)(slot % (int)kQueueSlots) * (CORDIC_QUEUE_BYTES / 4);
	}
	// a slot no launch in flight uses, or -1 (the caller then launches
	// without a queue)
	int claim(void *stream) const
	{
		if (!d)
			return -1;
		// A/B switch (measurement only): the round-2 behaviour, slots handed
		// out round-robin without looking at what is still in flight
		static const bool unchecked = [] {
			const char *e = std::getenv("CORDIC_QUEUE_UNCHECKED");
			return e && e[0] == '1';
		}();
		if (unchecked) {
			std::lock_guard<std::mutex> lock(mu);
			const unsigned k = next;
			next = (next + 1) % kQueueSlots;
			return (int)k + (int)kQueueSlots;	// launched() ignores it
		}
		hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
		if (stream && hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) != hipSuccess) {
			(void)hipGetLastError();
			cs = hipStreamCaptureStatusNone;
		}
		std::lock_guard<std::mutex> lock(mu);
		if (cs != hipStreamCaptureStatusNone) {
			if (next_captured >= kQueueSlots) {
				fallbacks++;
				return -1;
			}
			state[next_captured] = RETIRED;
			return (int)next_captured++;
		}
		// round robin; a slot whose last launch may still be running is
		// made safe ON THE DEVICE: this stream waits for that launch's
		// event (on the stream that recorded it the wait is free: stream
		// order already serialises the two kernels).  No host wait, and the
		// host may run any number of launches ahead of the GPU without
		// losing the queue.
		for (unsigned i = 0; i < kEagerSlots; i++) {
			const unsigned k = (next + i) % kEagerSlots;
			if (state[k] == CLAIMED || state[k] == RETIRED)
				continue;	// another thread is launching on it
			if (state[k] == RECORDED && hipEventQuery(ev[k]) != hipSuccess) {
				(void)hipGetLastError();	// hipErrorNotReady
				if (hipStreamWaitEvent(static_cast<hipStream_t>(stream),
						ev[k], 0) != hipSuccess) {
					(void)hipGetLastError();
					continue;
				}
			}
			prev[k] = state[k];
			state[k] = CLAIMED;
			next = (k + 1) % kEagerSlots;
			return (int)k;
		}
		fallbacks++;
		return -1;
	}
	// after the launch that uses `slot` has been enqueued (rc = its status)
	void launched(int slot, void *stream, int rc) const
	{
		if (slot < 0 || slot >= (int)kEagerSlots)
			return;
		std::lock_guard<std::mutex> lock(mu);
		if (rc != CORDIC_OK)
			// nothing new ran on it: what was there before the claim
			// stands -- a RECORDED slot keeps its pending event, so the
			// earlier launch is still waited for by the next taker
			state[slot] = prev[slot];
		else if (hipEventRecord(ev[slot], static_cast<hipStream_t>(stream)) == hipSuccess)
			state[slot] = RECORDED;
		else
			state[slot] = RETIRED;	// cannot tell when it is free again
	}
};

struct cordic_plan {
	cordic_config cfg;
	uint32_t *d_table = nullptr;	// device copy of the seed table
	int m = 0, S = 0, nbuckets = 0, nleaves = 0;
	DtInfo dt;			// direction tails behind the seeds (dt.n == 0: none)
	uint32_t *d_dir = nullptr;	// direction tables for per-sample vectors
	DxInfo dx;			// (dx.n == 0: none)
	QueueRing queues;
	// prologue images of the seeded kernels, per constant vector (round 5;
	// cordic_internal.h: SeedImages -- internally locked, write-once slots)
	SeedImages *images = nullptr;
	// batch size from which the table-driven kernels serve (< 0: default)
	std::atomic<long long> min_samples{-1};
	// an int16 entry point has been called on this plan: cordic_plan_prepare
	// then also builds the int16 container's image (its own slot)
	mutable std::atomic<bool> io16_used{false};
};

int cordic_last_kernel(void) { return g_last_kernel; }

int cordic_plan_create(const cordic_config *cfg, cordic_plan **plan)
{
	if (!cfg || !plan)
		return CORDIC_ERR_ARGS;
	if (cfg->mode < CORDIC_P2R || cfg->mode > CORDIC_SR2P)
		return CORDIC_ERR_MODE;
	if (!config_sane(*cfg))
		return CORDIC_ERR_ARGS;
	cordic_plan *p = new (std::nothrow) cordic_plan;
	if (!p)
		return CORDIC_ERR_NOMEM;
	p->cfg = *cfg;
	std::vector<uint32_t> words(4 + 4096 * 4 + 4096 * 2
			+ 4 + kDtMaxLevels * (6 + 2 * 4096 + 2 * 256));
	const size_t nw = build_seed_table(*cfg, CORDIC_SEED_STAGES, words.data(),
			words.size(), &p->dt);
	if (nw) {
		if (hipMalloc((void **)&p->d_table, nw * 4) != hipSuccess ||
		    hipMemcpy(p->d_table, words.data(), nw * 4,
				hipMemcpyHostToDevice) != hipSuccess ||
		    !p->queues.alloc()) {
			if (p->d_table) (void)hipFree(p->d_table);
			p->queues.release();
			delete p;
			return CORDIC_ERR_DEVICE;
		}
		p->m = (int)words[0];
		p->S = (int)words[1];
		p->nbuckets = (int)words[2];
		p->nleaves = (int)words[3];
		// CORDIC_SEED_IMAGES=0: every block computes its prologue (A/B)
		const char *e = std::getenv("CORDIC_SEED_IMAGES");
		if (!(e && e[0] == '0' && e[1] == 0))
			p->images = seed_images_create();	// (NULL: not fatal)
	}
	// direction tables for per-sample vectors (cordic_plan_p2r); independent
	// of the seed table
	{
		std::vector<uint32_t> dw(4 + kDxMaxLevels * (6 + 2 * 4096 + 2 * 256));
		const size_t dn = build_dir_table(*cfg, dw.data(), dw.size(), &p->dx);
		if (dn && (hipMalloc((void **)&p->d_dir, dn * 4) != hipSuccess ||
		    hipMemcpy(p->d_dir, dw.data(), dn * 4,
				hipMemcpyHostToDevice) != hipSuccess)) {
			(void)hipGetLastError();
			if (p->d_dir) (void)hipFree(p->d_dir);
			p->d_dir = nullptr;
			p->dx = DxInfo{};	// not fatal: cordic_p2r's kernel serves
		}
	}
	*plan = p;
	return CORDIC_OK;
}

int cordic_plan_dir_info(const cordic_plan *plan, int32_t *ngroups, int32_t stages[5])
{
	if (!plan || !ngroups)
		return CORDIC_ERR_ARGS;
	*ngroups = plan->d_dir ? plan->dx.n : 0;
	if (stages)
		for (int g = 0; g < kDxMaxLevels; g++)
			stages[g] = g < *ngroups ? plan->dx.lv[g].t : 0;
	return CORDIC_OK;
}

// the plan's direction tables for the per-sample-vector kernels
static void attach_dirs(const cordic_plan *plan, RotatorJob &j)
{
	j.dir_table = plan->d_dir;
	j.dx = plan->dx;
	j.min_samples = plan->min_samples.load(std::memory_order_relaxed);
}

int cordic_plan_p2r(const cordic_plan *plan, size_t n, const int32_t *d_xval,
		const int32_t *d_yval, const uint32_t *d_phase, int32_t *d_oxval,
		int32_t *d_oyval, void *stream)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	RotatorJob j;
	j.x = d_xval; j.y = d_yval; j.phase = d_phase;
	j.ox = d_oxval; j.oy = d_oyval; j.n = n;
	attach_dirs(plan, j);
	return launch_rotator(plan->cfg, Feed::PhaseArray_XYArray, j, stream);
}

int cordic_plan_mix(const cordic_plan *plan, size_t n, uint32_t phase0,
		uint32_t fcw, uint64_t index0, const int32_t *d_xval,
		const int32_t *d_yval, int32_t *d_oxval, int32_t *d_oyval, void *stream)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	RotatorJob j;
	j.x = d_xval; j.y = d_yval;
	j.phase0 = phase0; j.fcw = fcw; j.index0 = index0; j.xy_nco = true;
	j.ox = d_oxval; j.oy = d_oyval; j.n = n;
	attach_dirs(plan, j);
	return launch_rotator(plan->cfg, Feed::PhaseArray_XYArray, j, stream);
}

int cordic_plan_queue_info(const cordic_plan *plan, cordic_queue_info *info)
{
	if (!plan || !info)
		return CORDIC_ERR_ARGS;
	plan->queues.info(info);
	return CORDIC_OK;
}

void cordic_plan_destroy(cordic_plan *plan)
{
	if (!plan)
		return;
	if (plan->d_table)
		(void)hipFree(plan->d_table);
	if (plan->d_dir)
		(void)hipFree(plan->d_dir);
	seed_images_destroy(plan->images);
	plan->queues.release();
	delete plan;
}

const cordic_config *cordic_plan_config(const cordic_plan *plan)
{
	return plan ? &plan->cfg : nullptr;
}

int cordic_plan_seed_info(const cordic_plan *plan, int32_t *stages,
		int32_t *nleaves, int32_t *nbuckets)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	if (stages) *stages = plan->m;
	if (nleaves) *nleaves = plan->nleaves;
	if (nbuckets) *nbuckets = plan->nbuckets;
	return CORDIC_OK;
}

int cordic_plan_tail_info(const cordic_plan *plan, int32_t *ngroups,
		int32_t stages[4])
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	if (ngroups) *ngroups = plan->dt.n;
	if (stages)
		for (int g = 0; g < 4; g++)
			stages[g] = g < plan->dt.n ? plan->dt.lv[g].t : 0;
	return CORDIC_OK;
}

// launch with a tile queue no other launch in flight is using
template <typename F> static int with_queue(const QueueRing &ring, void *stream, F launch)
{
	const int slot = ring.claim(stream);
	const int rc = launch(ring.ptr(slot));
	ring.launched(slot, stream, rc);
	return rc;
}

static void attach_seed(const cordic_plan *plan, RotatorJob &j)
{
	j.seed_table = plan->d_table;
	j.seed_m = plan->m;
	j.seed_S = plan->S;
	j.seed_nbuckets = plan->nbuckets;
	j.seed_nleaves = plan->nleaves;
	j.dt = plan->dt;
	j.images = plan->images;
	j.min_samples = plan->min_samples.load(std::memory_order_relaxed);
}

int cordic_plan_prepare(const cordic_plan *plan, int32_t xval, int32_t yval,
		void *stream)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	if (!plan->d_table || !plan->images)
		return CORDIC_ERR_UNSUPPORTED;
	int rc = CORDIC_OK;
	// the image of the 32-bit arrays' container and -- only once the plan has
	// served an int16 call: an image takes one of its eight write-once slots
	// -- the int16 arrays' 
	for (int io16 = 0; io16 < 2; io16++) {
		if (io16 && (plan->cfg.iw > 16 || plan->cfg.ow > 16
				|| !plan->io16_used.load(std::memory_order_relaxed)))
			break;
		RotatorJob j;
		j.x0 = xval; j.y0 = yval;
		j.io16 = io16 != 0;
		j.prepare_only = true;
		attach_seed(plan, j);
		// the image of the instance a QUEUED launch runs (the ordinary case;
		// a launch without a queue runs the dynamic-exit instance and keeps
		// its own): any counter block stands for "queued" here -- build mode
		// never touches it
		j.queue = plan->queues.d;
		const int r = launch_rotator(plan->cfg, Feed::PhaseArray_ConstXY, j, stream);
		if (!io16)
			rc = r;
	}
	// a set-up call: it returns with the image COMPLETE (one block, ~15 us),
	// so that a capture begun right behind it may use it
	if (rc == CORDIC_OK && !seed_images_settle(plan->images))
		rc = CORDIC_ERR_DEVICE;
	return rc;
}

// ------------------------------------------------------------- job sets
// Many small jobs in one launch (include/cordic_amd.h, "job sets").  The host
// cuts the batch into the seeded kernel's tiles once and keeps the table on
// the device; running the set is then ONE launch of the dynamic-exit instance
// (+ one small launch for trailing samples), whatever the number of jobs.
struct cordic_jobset {
	int	kind = 0;		// enum cordic_jobs_kind
	int	device = 0;
	std::vector<cordic_job> jobs;	// host copy: the one-by-one fallback
	uint32_t *d_tiles = nullptr, *d_tails = nullptr;
	JobTables tabs;
	cordic_config cfg;		// of the plan it was cut for (PW decides NCO words)
};

namespace {
// one-shot batches (cordic_plan_*_batch): the set has to outlive the launch
// that reads it; it is parked here with an event and freed by a later call
// once that event has completed (or by cordic_jobset_reap / process exit)
struct Parked { cordic_jobset *set; hipEvent_t done; };
std::mutex g_parked_mu;
std::vector<Parked> g_parked;

void reap_parked(bool wait)
{
	std::vector<Parked> dead;
	{
		std::lock_guard<std::mutex> lock(g_parked_mu);
		for (size_t k = 0; k < g_parked.size();) {
			Parked &p = g_parked[k];
			const hipError_t e = wait ? hipEventSynchronize(p.done)
						  : hipEventQuery(p.done);
			// (only a launch that is KNOWN to have passed gives its tables
			// back: an event in an error state says nothing about who still
			// reads them, and the set then stays parked for good)
			if (e == hipSuccess) {
				dead.push_back(p);
				g_parked.erase(g_parked.begin() + (long)k);
			} else {
				(void)hipGetLastError();
				k++;
			}
		}
	}
	for (Parked &p : dead) {
		(void)hipEventDestroy(p.done);
		cordic_jobset_destroy(p.set);
	}
}
} // namespace

namespace {
// same core?  field by field: the PODs come from different places and their
// padding bytes are nobody's business
bool same_core(const cordic_config &a, const cordic_config &b)
{
	if (a.mode != b.mode || a.iw != b.iw || a.ow != b.ow || a.nextra != b.nextra
			|| a.ww != b.ww || a.pw != b.pw || a.nstages != b.nstages
			|| a.nlive != b.nlive || a.needs_wrap != b.needs_wrap
			|| a.flags != b.flags)
		return false;
	for (int i = 0; i < a.nstages && i < CORDIC_AMD_MAX_STAGES; i++)
		if (a.angle[i] != b.angle[i])
			return false;
	return true;
}

// Tile length (vectors) of the data-fed kinds: whole passes of a 256-thread
// block, as long as the seeded kernels' tiles where the set is big enough to
// give every resident block (8 per CU) four of them, shorter below.
uint32_t xy_tile_vecs(uint64_t total_vecs)
{
	int cus = 0, dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus,
			hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
		(void)hipGetLastError();
		cus = 256;
	}
	const uint64_t per = total_vecs / ((uint64_t)cus * 8u * 4u);
	uint64_t v = per / kJobPassVecs * kJobPassVecs;
	if (v < kJobPassVecs) v = kJobPassVecs;
	if (v > kJobTileVecs) v = kJobTileVecs;
	return (uint32_t)v;
}
} // namespace

int cordic_jobset_create(const cordic_plan *plan, int kind, size_t njobs,
		const cordic_job *jobs, cordic_jobset **out)
{
	if (!plan || !out || (njobs && !jobs) || kind < CORDIC_JOBS_PHASE_ARRAYS
			|| kind > CORDIC_JOBS_MIX)
		return CORDIC_ERR_ARGS;
	const bool rot = plan->cfg.mode == CORDIC_P2R || plan->cfg.mode == CORDIC_SP2R;
	if (rot == (kind == CORDIC_JOBS_R2P))
		return CORDIC_ERR_MODE;
	const bool xy = kind >= CORDIC_JOBS_R2P;	// per-sample vectors from memory
	const bool phase_array = kind == CORDIC_JOBS_PHASE_ARRAYS
			|| kind == CORDIC_JOBS_P2R_XY;
	const bool gen_phase = kind == CORDIC_JOBS_NCO || kind == CORDIC_JOBS_MIX;
	std::vector<TileDesc> tiles;
	std::vector<TailDesc> tails;
	std::vector<TileDescXY> xtiles, xtails;
	uint64_t samples = 0, vecs = 0;
	const int sh = 32 - plan->cfg.pw;
	for (size_t k = 0; k < njobs; k++) {
		const cordic_job &jb = jobs[k];
		if (jb.n == 0)
			continue;
		if (!jb.d_oxval || !jb.d_oyval || (phase_array && !jb.d_phase)
				|| (xy && (!jb.d_xval || !jb.d_yval))
				|| ((uintptr_t)jb.d_oxval & 3) || ((uintptr_t)jb.d_oyval & 3)
				|| (phase_array && ((uintptr_t)jb.d_phase & 3))
				|| (xy && (((uintptr_t)jb.d_xval & 3) || ((uintptr_t)jb.d_yval & 3))))
			return CORDIC_ERR_ARGS;
		samples += jb.n;
		vecs += jb.n / 4;
	}
	const uint32_t tile_vecs = xy ? xy_tile_vecs(vecs) : kJobTileVecs;
	for (size_t k = 0; k < njobs; k++) {
		const cordic_job &jb = jobs[k];
		if (jb.n == 0)
			continue;
		const uint64_t nvec = jb.n / 4;
		// the phase of sample s of the job, left-justified (NCO / MIX jobs)
		auto nco_word = [&](uint64_t s) -> uint64_t {
			const uint32_t f = jb.fcw << sh;
			const uint32_t p = (jb.phase0 << sh) + (uint32_t)(jb.index0 + s) * f;
			return ((uint64_t)f << 32) | p;
		};
		auto in_word = [&](uint64_t s) -> uint64_t {
			return gen_phase ? nco_word(s)
				: phase_array ? (uint64_t)(uintptr_t)(jb.d_phase + s) : 0;
		};
		auto xy_desc = [&](uint64_t s, uint32_t live) {
			TileDescXY d{};
			d.in0 = (uint64_t)(uintptr_t)(jb.d_xval + s);
			d.in1 = (uint64_t)(uintptr_t)(jb.d_yval + s);
			d.in2 = in_word(s);
			d.o0 = (uint64_t)(uintptr_t)(jb.d_oxval + s);
			d.o1 = (uint64_t)(uintptr_t)(jb.d_oyval + s);
			d.live = live;
			return d;
		};
		for (uint64_t v0 = 0; v0 < nvec; v0 += tile_vecs) {
			const uint32_t live = (uint32_t)(nvec - v0 < tile_vecs ? nvec - v0 : tile_vecs);
			if (xy) {
				xtiles.push_back(xy_desc(v0 * 4, live));
				continue;
			}
			TileDesc d{};
			d.in = in_word(v0 * 4);
			d.ox = (uint64_t)(uintptr_t)(jb.d_oxval + v0 * 4);
			d.oy = (uint64_t)(uintptr_t)(jb.d_oyval + v0 * 4);
			d.live = live;
			tiles.push_back(d);
		}
		for (uint64_t s = nvec * 4; s < jb.n; s++) {
			if (xy) {
				xtails.push_back(xy_desc(s, 0));
				continue;
			}
			TailDesc d{};
			d.in = in_word(s);
			d.ox = (uint64_t)(uintptr_t)(jb.d_oxval + s);
			d.oy = (uint64_t)(uintptr_t)(jb.d_oyval + s);
			tails.push_back(d);
		}
		if (tiles.size() > 0x7fffffffu || tails.size() > 0x7fffffffu
				|| xtiles.size() > 0x7fffffffu || xtails.size() > 0x7fffffffu)
			return CORDIC_ERR_ARGS;
	}
	cordic_jobset *set = new (std::nothrow) cordic_jobset;
	if (!set)
		return CORDIC_ERR_NOMEM;
	set->kind = kind;
	set->cfg = plan->cfg;
	set->jobs.assign(jobs, jobs + njobs);
	if (hipGetDevice(&set->device) != hipSuccess) {
		(void)hipGetLastError();
		set->device = -1;
	}
	auto upload = [](const void *src, size_t bytes, uint32_t **dst) {
		if (!bytes)
			return true;
		return hipMalloc((void **)dst, bytes) == hipSuccess
			&& hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
	};
	const bool up = xy
		? upload(xtiles.data(), xtiles.size() * sizeof(TileDescXY), &set->d_tiles)
			&& upload(xtails.data(), xtails.size() * sizeof(TileDescXY), &set->d_tails)
		: upload(tiles.data(), tiles.size() * sizeof(TileDesc), &set->d_tiles)
			&& upload(tails.data(), tails.size() * sizeof(TailDesc), &set->d_tails);
	if (!up) {
		(void)hipGetLastError();
		cordic_jobset_destroy(set);
		return CORDIC_ERR_DEVICE;
	}
	set->tabs.tiles = set->d_tiles;
	set->tabs.ntiles = (uint32_t)(xy ? xtiles.size() : tiles.size());
	set->tabs.tails = set->d_tails;
	set->tabs.ntails = (uint32_t)(xy ? xtails.size() : tails.size());
	set->tabs.samples = samples;
	*out = set;
	return CORDIC_OK;
}

void cordic_jobset_destroy(cordic_jobset *set)
{
	if (!set)
		return;
	if (set->d_tiles) (void)hipFree(set->d_tiles);
	if (set->d_tails) (void)hipFree(set->d_tails);
	delete set;
}

int cordic_jobset_info(const cordic_jobset *set, uint64_t *samples,
		uint32_t *tiles, uint32_t *tail_samples)
{
	if (!set)
		return CORDIC_ERR_ARGS;
	if (samples) *samples = set->tabs.samples;
	if (tiles) *tiles = set->tabs.ntiles;
	if (tail_samples) *tail_samples = set->tabs.ntails;
	return CORDIC_OK;
}

int cordic_plan_run_jobs(const cordic_plan *plan, const cordic_jobset *set,
		int32_t xval, int32_t yval, void *stream)
{
	if (!plan || !set)
		return CORDIC_ERR_ARGS;
	// cut for this core (the tile table holds PW-scaled phase words) and for
	// this device (it holds device addresses)?
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess) {
		(void)hipGetLastError();
		return CORDIC_ERR_DEVICE;
	}
	if (!same_core(set->cfg, plan->cfg) || dev != set->device)
		return CORDIC_ERR_ARGS;
	if (set->kind >= CORDIC_JOBS_R2P) {
		RotatorJob j;
		if (set->kind != CORDIC_JOBS_R2P)
			attach_dirs(plan, j);
		int rc = launch_xy_jobs(plan->cfg, set->kind, j, set->tabs, stream);
		if (rc != CORDIC_ERR_UNSUPPORTED)
			return rc;
		// no tile-reading instance for this core: the jobs one by one
		for (const cordic_job &jb : set->jobs) {
			if (jb.n == 0)
				continue;
			rc = set->kind == CORDIC_JOBS_R2P
				? launch_topolar(plan->cfg, (size_t)jb.n, jb.d_xval, jb.d_yval,
					jb.d_oxval, reinterpret_cast<uint32_t *>(jb.d_oyval), stream)
				: set->kind == CORDIC_JOBS_MIX
				? cordic_plan_mix(plan, (size_t)jb.n, jb.phase0, jb.fcw, jb.index0,
					jb.d_xval, jb.d_yval, jb.d_oxval, jb.d_oyval, stream)
				: cordic_plan_p2r(plan, (size_t)jb.n, jb.d_xval, jb.d_yval,
					jb.d_phase, jb.d_oxval, jb.d_oyval, stream);
			if (rc != CORDIC_OK)
				return rc;
		}
		return CORDIC_OK;
	}
	const Feed feed = set->kind == CORDIC_JOBS_NCO ? Feed::Nco_ConstXY
						     : Feed::PhaseArray_ConstXY;
	RotatorJob j;
	j.x0 = xval; j.y0 = yval;
	attach_seed(plan, j);
	int rc = with_queue(plan->queues, stream, [&](uint32_t *q) {
		j.queue = q;
		return launch_rotator_jobs(plan->cfg, feed, j, set->tabs, stream);
	});
	if (rc != CORDIC_ERR_UNSUPPORTED)
		return rc;
	// no seeded kernel for this core (or no tile queue to be had right now):
	// the jobs one by one through the ordinary entry points -- same results
	for (const cordic_job &jb : set->jobs) {
		if (jb.n == 0)
			continue;
		rc = feed == Feed::Nco_ConstXY
			? cordic_plan_nco(plan, (size_t)jb.n, jb.phase0, jb.fcw, jb.index0,
				xval, yval, jb.d_oxval, jb.d_oyval, stream)
			: cordic_plan_p2r_const(plan, (size_t)jb.n, xval, yval, jb.d_phase,
				jb.d_oxval, jb.d_oyval, stream);
		if (rc != CORDIC_OK)
			return rc;
	}
	return CORDIC_OK;
}

static int run_batch_once(const cordic_plan *plan, int kind, size_t njobs,
		const cordic_job *jobs, int32_t xval, int32_t yval, void *stream)
{
	// cutting a set allocates and copies (blocking): not inside a capture --
	// and the graph would outlive the tables this call frees behind its launch
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	if (stream && hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs)
			!= hipSuccess) {
		(void)hipGetLastError();
		cs = hipStreamCaptureStatusNone;
	}
	if (cs != hipStreamCaptureStatusNone)
		return CORDIC_ERR_UNSUPPORTED;
	reap_parked(false);
	cordic_jobset *set = nullptr;
	if (int rc = cordic_jobset_create(plan, kind, njobs, jobs, &set))
		return rc;
	const int rc = cordic_plan_run_jobs(plan, set, xval, yval, stream);
	// the tables have to stay until the launch has read them
	hipEvent_t done = nullptr;
	if (hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess
			&& hipEventRecord(done, static_cast<hipStream_t>(stream)) == hipSuccess) {
		std::lock_guard<std::mutex> lock(g_parked_mu);
		g_parked.push_back(Parked{set, done});
	} else {
		(void)hipGetLastError();
		if (done) (void)hipEventDestroy(done);
		(void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
		cordic_jobset_destroy(set);
	}
	return rc;
}

int cordic_plan_p2r_const_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, int32_t xval, int32_t yval, void *stream)
{
	return run_batch_once(plan, CORDIC_JOBS_PHASE_ARRAYS, njobs, jobs, xval, yval,
			stream);
}

int cordic_plan_nco_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, int32_t xval, int32_t yval, void *stream)
{
	return run_batch_once(plan, CORDIC_JOBS_NCO, njobs, jobs, xval, yval, stream);
}

int cordic_plan_r2p_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, void *stream)
{
	return run_batch_once(plan, CORDIC_JOBS_R2P, njobs, jobs, 0, 0, stream);
}

int cordic_plan_p2r_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, void *stream)
{
	return run_batch_once(plan, CORDIC_JOBS_P2R_XY, njobs, jobs, 0, 0, stream);
}

int cordic_plan_mix_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, void *stream)
{
	return run_batch_once(plan, CORDIC_JOBS_MIX, njobs, jobs, 0, 0, stream);
}

void cordic_jobset_reap(void) { reap_parked(true); }

int cordic_plan_set_min_samples(cordic_plan *plan, long long min_samples)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	plan->min_samples.store(min_samples < 0 ? -1 : min_samples,
			std::memory_order_relaxed);
	return CORDIC_OK;
}

int cordic_plan_image_info(const cordic_plan *plan, int32_t *held, uint64_t *hits,
		uint64_t *misses)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	seed_images_info(plan->images, held, hits, misses);
	return CORDIC_OK;
}

int cordic_plan_p2r_const(const cordic_plan *plan, size_t n, int32_t xval,
		int32_t yval, const uint32_t *d_phase, int32_t *d_oxval,
		int32_t *d_oyval, void *stream)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	RotatorJob j;
	j.x0 = xval; j.y0 = yval; j.phase = d_phase;
	j.ox = d_oxval; j.oy = d_oyval; j.n = n;
	attach_seed(plan, j);
	return with_queue(plan->queues, stream, [&](uint32_t *q) {
		j.queue = q;
		return launch_rotator(plan->cfg, Feed::PhaseArray_ConstXY, j, stream);
	});
}

int cordic_plan_nco(const cordic_plan *plan, size_t n, uint32_t phase0,
		uint32_t fcw, uint64_t index0, int32_t xval, int32_t yval,
		int32_t *d_oxval, int32_t *d_oyval, void *stream)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	RotatorJob j;
	j.x0 = xval; j.y0 = yval; j.phase0 = phase0; j.fcw = fcw;
	j.index0 = index0; j.ox = d_oxval; j.oy = d_oyval; j.n = n;
	attach_seed(plan, j);
	return with_queue(plan->queues, stream, [&](uint32_t *q) {
		j.queue = q;
		return launch_rotator(plan->cfg, Feed::Nco_ConstXY, j, stream);
	});
}

// 16-bit containers: the job carries the int16 / uint16 arrays behind its
// int32 pointers (cordic_internal.h: RotatorJob::io16)
namespace {
int fits16(const cordic_config &c, bool phase_array)
{
	if (c.iw > 16 || c.ow > 16 || (phase_array && c.pw > 16))
		return CORDIC_ERR_CONTAINER;
	return CORDIC_OK;
}
template <typename T> const int32_t *as_i32(const T *p)
{
	return reinterpret_cast<const int32_t *>(p);
}
template <typename T> int32_t *as_i32(T *p)
{
	return reinterpret_cast<int32_t *>(p);
}
RotatorJob job16(const int16_t *x, const int16_t *y, const uint16_t *phase,
		int16_t *ox, int16_t *oy, size_t n)
{
	RotatorJob j;
	j.x = as_i32(x); j.y = as_i32(y);
	j.phase = reinterpret_cast<const uint32_t *>(phase);
	j.ox = as_i32(ox); j.oy = as_i32(oy);
	j.n = n;
	j.io16 = true;
	return j;
}
} // namespace

int cordic_p2r16(const cordic_config *cfg, size_t n, const int16_t *d_xval,
		const int16_t *d_yval, const uint16_t *d_phase, int16_t *d_oxval,
		int16_t *d_oyval, void *stream)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	if (int rc = fits16(*cfg, true))
		return rc;
	RotatorJob j = job16(d_xval, d_yval, d_phase, d_oxval, d_oyval, n);
	return launch_rotator(*cfg, Feed::PhaseArray_XYArray, j, stream);
}

int cordic_p2r16_const(const cordic_config *cfg, size_t n, int32_t xval,
		int32_t yval, const uint16_t *d_phase, int16_t *d_oxval,
		int16_t *d_oyval, void *stream)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	if (int rc = fits16(*cfg, true))
		return rc;
	RotatorJob j = job16(nullptr, nullptr, d_phase, d_oxval, d_oyval, n);
	j.x0 = xval; j.y0 = yval;
	return launch_rotator(*cfg, Feed::PhaseArray_ConstXY, j, stream);
}

int cordic_nco16(const cordic_config *cfg, size_t n, uint32_t phase0,
		uint32_t fcw, uint64_t index0, int32_t xval, int32_t yval,
		int16_t *d_oxval, int16_t *d_oyval, void *stream)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	if (int rc = fits16(*cfg, false))
		return rc;
	RotatorJob j = job16(nullptr, nullptr, nullptr, d_oxval, d_oyval, n);
	j.x0 = xval; j.y0 = yval; j.phase0 = phase0; j.fcw = fcw;
	j.index0 = index0;
	return launch_rotator(*cfg, Feed::Nco_ConstXY, j, stream);
}

int cordic_r2p16(const cordic_config *cfg, size_t n, const int16_t *d_xval,
		const int16_t *d_yval, int16_t *d_omag, uint16_t *d_ophase,
		void *stream)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	if (int rc = fits16(*cfg, true))
		return rc;
	return launch_topolar(*cfg, n, as_i32(d_xval), as_i32(d_yval),
			as_i32(d_omag), reinterpret_cast<uint32_t *>(d_ophase),
			stream, true);
}

int cordic_plan_p2r16_const(const cordic_plan *plan, size_t n, int32_t xval,
		int32_t yval, const uint16_t *d_phase, int16_t *d_oxval,
		int16_t *d_oyval, void *stream)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	if (int rc = fits16(plan->cfg, true))
		return rc;
	RotatorJob j = job16(nullptr, nullptr, d_phase, d_oxval, d_oyval, n);
	j.x0 = xval; j.y0 = yval;
	attach_seed(plan, j);
	plan->io16_used.store(true, std::memory_order_relaxed);
	return with_queue(plan->queues, stream, [&](uint32_t *q) {
		j.queue = q;
		return launch_rotator(plan->cfg, Feed::PhaseArray_ConstXY, j, stream);
	});
}

int cordic_plan_nco16(const cordic_plan *plan, size_t n, uint32_t phase0,
		uint32_t fcw, uint64_t index0, int32_t xval, int32_t yval,
		int16_t *d_oxval, int16_t *d_oyval, void *stream)
{
	if (!plan)
		return CORDIC_ERR_ARGS;
	if (int rc = fits16(plan->cfg, false))
		return rc;
	RotatorJob j = job16(nullptr, nullptr, nullptr, d_oxval, d_oyval, n);
	j.x0 = xval; j.y0 = yval; j.phase0 = phase0; j.fcw = fcw;
	j.index0 = index0;
	attach_seed(plan, j);
	plan->io16_used.store(true, std::memory_order_relaxed);
	return with_queue(plan->queues, stream, [&](uint32_t *q) {
		j.queue = q;
		return launch_rotator(plan->cfg, Feed::Nco_ConstXY, j, stream);
	});
}

// ------------------------------------------------------------- table cores
struct cordic_table {
	cordic_table_config cfg;
	int32_t *d_tbl = nullptr;
	// optional packed copy for the LDS kernel (cordic_kernels.hip:
	// table_lookup_lds): mode 1 = quarter-wave table as is, 2 = full-wave
	// table folded to its first quadrant (+ the peak entry)
	int16_t *d_lds16 = nullptr;
	int	lds_mode = 0, lds_entries = 0;
	QueueRing queues;	// optional: without it the chunk-per-block sweep runs
};

namespace {
// A packed int16 table of at most 64 KiB (+ one entry) if the core allows it.
// For -t tbl the fold is only used when every one of the 2^PW entries is
// reproduced by it.
bool pack_for_lds(const cordic_table_config &c, const std::vector<int32_t> &t,
		std::vector<int16_t> *out, int *mode)
{
	if (c.pw < 4)
		return false;
	const int quarter = 1 << (c.pw - 2);
	if (quarter > 32768)
		return false;
	// OW <= 16: a packed int16 copy (modes 1 / 2, two blocks per CU); wider
	// outputs: the 32-bit entries themselves (modes 3 / 4, 128 KiB for 2^15
	// entries, one block per CU), read from the table in HBM by the kernel
	const bool wide = c.ow > 16;
	if (c.kind == CORDIC_QTR) {
		if (!wide) {
			out->resize((size_t)quarter);
			for (int k = 0; k < quarter; k++)
				(*out)[(size_t)k] = (int16_t)t[(size_t)k];
		}
		*mode = wide ? 3 : 1;
		return true;
	}
	const int n = 1 << c.pw;
	for (int i = 0; i < n; i++) {
		const int q = i >> (c.pw - 2), j = i & (quarter - 1);
		int32_t v = t[(size_t)((q & 1) ? quarter - j : j)];
		if (q & 2)
			v = -v;
		if (v != t[(size_t)i])
			return false;
	}
	if (!wide) {
		out->resize((size_t)quarter + 1);
		for (int k = 0; k <= quarter; k++)
			(*out)[(size_t)k] = (int16_t)t[(size_t)k];
	}
	*mode = wide ? 4 : 2;
	return true;
}
} // namespace

int cordic_table_config_init(cordic_table_config *cfg, int kind, int iw, int ow,
		int phase_bits)
{
	return table_derive(cfg, kind, iw, ow, phase_bits);
}

int cordic_table_values(const cordic_table_config *cfg, int32_t *out, size_t cap)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	return table_fill(*cfg, out, cap);
}

int cordic_table_create(const cordic_table_config *cfg, cordic_table **tbl)
{
	if (!cfg || !tbl || !table_sane(*cfg))
		return CORDIC_ERR_ARGS;
	std::vector<int32_t> host((size_t)cfg->entries);
	int rc = table_fill(*cfg, host.data(), host.size());
	if (rc != CORDIC_OK)
		return rc;
	cordic_table *t = new (std::nothrow) cordic_table;
	if (!t)
		return CORDIC_ERR_NOMEM;
	t->cfg = *cfg;
	if (hipMalloc((void **)&t->d_tbl, host.size() * 4) != hipSuccess ||
	    hipMemcpy(t->d_tbl, host.data(), host.size() * 4,
			hipMemcpyHostToDevice) != hipSuccess) {
		if (t->d_tbl) (void)hipFree(t->d_tbl);
		delete t;
		return CORDIC_ERR_DEVICE;
	}
	std::vector<int16_t> packed;
	int mode = 0;
	if (pack_for_lds(*cfg, host, &packed, &mode)) {
		const int quarter = 1 << (cfg->pw - 2);
		if (mode >= 3) {
			// the kernel fills its LDS copy from d_tbl itself
			t->lds_mode = mode;
			t->lds_entries = quarter + (mode == 4 ? 1 : 0);
		} else if (hipMalloc((void **)&t->d_lds16, packed.size() * 2) == hipSuccess
				&& hipMemcpy(t->d_lds16, packed.data(), packed.size() * 2,
					hipMemcpyHostToDevice) == hipSuccess) {
			// optional: on failure the L2 gather kernel serves the table
			t->lds_mode = mode;
			t->lds_entries = (int)packed.size();
		} else {
			if (t->d_lds16) (void)hipFree(t->d_lds16);
			t->d_lds16 = nullptr;
			(void)hipGetLastError();
		}
	}
	if (!t->queues.alloc())
		(void)hipGetLastError();
	*tbl = t;
	return CORDIC_OK;
}

int cordic_table_queue_info(const cordic_table *tbl, cordic_queue_info *info)
{
	if (!tbl || !info)
		return CORDIC_ERR_ARGS;
	tbl->queues.info(info);
	return CORDIC_OK;
}

void cordic_table_destroy(cordic_table *tbl)
{
	if (!tbl)
		return;
	tbl->queues.release();
	if (tbl->d_tbl)
		(void)hipFree(tbl->d_tbl);
	if (tbl->d_lds16)
		(void)hipFree(tbl->d_lds16);
	delete tbl;
}

int cordic_table_lookup(const cordic_table *tbl, size_t n,
		const uint32_t *d_phase, int32_t *d_val, void *stream)
{
	if (!tbl)
		return CORDIC_ERR_ARGS;
	return with_queue(tbl->queues, stream, [&](uint32_t *q) {
		return launch_table_lookup(tbl->cfg, tbl->d_tbl, n, d_phase, d_val,
				stream, tbl->d_lds16, tbl->lds_mode, tbl->lds_entries, q);
	});
}

// ------------------------------------------------- quadratic sine core
struct cordic_quad {
	cordic_quad_config cfg;
	int32_t *d_tab = nullptr;	// entries x {C, L, Q, 0}
	QueueRing queues;
};

int cordic_quad_config_init(cordic_quad_config *cfg, int iw, int ow, int xtra,
		int phase_bits)
{
	return quad_build_from_cli(cfg, iw, ow, xtra, phase_bits);
}

int cordic_quad_config_init_core(cordic_quad_config *cfg, int phase_bits, int ow,
		int nxtra)
{
	return quad_build_core(cfg, phase_bits, ow, nxtra);
}

int cordic_quad_tables(const cordic_quad_config *cfg, int32_t *ctbl,
		int32_t *ltbl, int32_t *qtbl, size_t cap)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	return quad_fill(*cfg, ctbl, ltbl, qtbl, cap);
}

int cordic_quad_write_header(const cordic_quad_config *cfg, const char *name,
		char *buf, size_t cap)
{
	return quad_write_header(cfg, name, buf, cap);
}

int cordic_quad_create(const cordic_quad_config *cfg, cordic_quad **core)
{
	if (!cfg || !core || !quad_sane(*cfg))
		return CORDIC_ERR_ARGS;
	const size_t n = (size_t)cfg->entries;
	std::vector<int32_t> c(n), l(n), q(n), packed(n * 4);
	int rc = quad_fill(*cfg, c.data(), l.data(), q.data(), n);
	if (rc != CORDIC_OK)
		return rc;
	for (size_t k = 0; k < n; k++) {
		packed[4 * k] = c[k];
		packed[4 * k + 1] = l[k];
		packed[4 * k + 2] = q[k];
		packed[4 * k + 3] = 0;
	}
	cordic_quad *h = new (std::nothrow) cordic_quad;
	if (!h)
		return CORDIC_ERR_NOMEM;
	h->cfg = *cfg;
	if (hipMalloc((void **)&h->d_tab, packed.size() * 4) != hipSuccess ||
	    hipMemcpy(h->d_tab, packed.data(), packed.size() * 4,
			hipMemcpyHostToDevice) != hipSuccess) {
		if (h->d_tab) (void)hipFree(h->d_tab);
		delete h;
		return CORDIC_ERR_DEVICE;
	}
	if (!h->queues.alloc())
		(void)hipGetLastError();
	*core = h;
	return CORDIC_OK;
}

int cordic_quad_queue_info(const cordic_quad *core, cordic_queue_info *info)
{
	if (!core || !info)
		return CORDIC_ERR_ARGS;
	core->queues.info(info);
	return CORDIC_OK;
}

void cordic_quad_destroy(cordic_quad *core)
{
	if (!core)
		return;
	core->queues.release();
	if (core->d_tab)
		(void)hipFree(core->d_tab);
	delete core;
}

int cordic_quad_lookup(const cordic_quad *core, size_t n, const uint32_t *d_phase,
		int32_t *d_sin, void *stream)
{
	if (!core)
		return CORDIC_ERR_ARGS;
	return with_queue(core->queues, stream, [&](uint32_t *q) {
		return launch_quad_lookup(core->cfg, core->d_tab, n, d_phase, d_sin,
				stream, q);
	});
}

// Scratch of the clocked views.  cordic_*_reserve sizes it up front; a *_ticks
// call that needs more grows it IN STREAM ORDER on the caller's stream
// (hipFreeAsync / hipMallocAsync): earlier kernels of that stream still see the
// old block, no other stream is stalled and nothing synchronises the device.
// (Not inside a stream capture: reserve first, then capture.)
static int grow_workspace(void **ws, size_t *ws_bytes, size_t need, void *stream)
{
	if (need <= *ws_bytes)
		return CORDIC_OK;
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (*ws && hipFreeAsync(*ws, st) != hipSuccess)
		return CORDIC_ERR_DEVICE;
	*ws = nullptr;
	*ws_bytes = 0;
	if (hipMallocAsync(ws, need, st) != hipSuccess) {
		*ws = nullptr;
		return CORDIC_ERR_DEVICE;
	}
	*ws_bytes = need;
	return CORDIC_OK;
}

// ------------------------------------------------- clocked view (stream)
struct cordic_stream {
	cordic_config cfg;
	StreamState st;
	int latency = 0;
};

void cordic_stream_destroy(cordic_stream *s)
{
	if (!s)
		return;
	StreamState &t = s->st;
	void *ptrs[] = { t.hx, t.hy, t.hph, t.haux, t.epoch, t.born_phase, t.ws };
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	delete s;
}

int cordic_stream_create(const cordic_config *cfg, cordic_stream **out)
{
	if (!cfg || !out)
		return CORDIC_ERR_ARGS;
	if (cfg->mode != CORDIC_P2R && cfg->mode != CORDIC_R2P)
		return CORDIC_ERR_MODE;
	if (!config_sane(*cfg))
		return CORDIC_ERR_ARGS;
	cordic_stream *s = new (std::nothrow) cordic_stream;
	if (!s)
		return CORDIC_ERR_NOMEM;
	s->cfg = *cfg;
	const int L = cfg->nstages + 2;
	s->latency = L;
	StreamState &t = s->st;
	auto zalloc = [](auto **p, size_t bytes) {
		return hipMalloc((void **)p, bytes) == hipSuccess
			&& hipMemset(*p, 0, bytes) == hipSuccess;
	};
	bool ok = zalloc(&t.hx, (size_t)L * 4) && zalloc(&t.hy, (size_t)L * 4)
		&& zalloc(&t.hph, (size_t)L * 4) && zalloc(&t.haux, (size_t)L)
		&& zalloc(&t.epoch, 4);
	// rtl/topolar.v:235-243 on cleared registers: the phase accumulator of a
	// stage born at reset still collects angle[i] of every live stage it
	// passes; after e enabled clocks the output register shows the one born
	// in register NSTAGES-e+1 (cordic_stream.hip).  Skipped stages
	// (rtl/topolar.v:217-225) add nothing.
	std::vector<uint32_t> born((size_t)L + 1, 0u);
	const uint32_t pmask = (cfg->pw >= 32) ? 0xffffffffu : ((1u << cfg->pw) - 1u);
	for (int e = 1; e <= L - 1; e++) {
		uint32_t acc = 0;
		for (int i = cfg->nstages - e + 1; i < cfg->nstages; i++)
			if (i >= 0 && i < cfg->nlive)
				acc += cfg->angle[i];
		born[(size_t)e] = acc & pmask;
	}
	ok = ok && hipMalloc((void **)&t.born_phase, born.size() * 4) == hipSuccess
		&& hipMemcpy(t.born_phase, born.data(), born.size() * 4,
				hipMemcpyHostToDevice) == hipSuccess;
	if (!ok) {
		cordic_stream_destroy(s);
		return CORDIC_ERR_DEVICE;
	}
	*out = s;
	return CORDIC_OK;
}

size_t cordic_stream_workspace(size_t ticks) { return stream_workspace_bytes(ticks); }

int cordic_stream_reserve(cordic_stream *s, size_t max_ticks)
{
	if (!s)
		return CORDIC_ERR_ARGS;
	const size_t need = stream_workspace_bytes(max_ticks);
	if (need <= s->st.ws_bytes)
		return CORDIC_OK;
	// kernels of earlier calls may still be using the old scratch
	if (hipDeviceSynchronize() != hipSuccess)
		return CORDIC_ERR_DEVICE;
	if (s->st.ws) (void)hipFree(s->st.ws);
	s->st.ws = nullptr;
	s->st.ws_bytes = 0;
	if (hipMalloc(&s->st.ws, need) != hipSuccess)
		return CORDIC_ERR_DEVICE;
	s->st.ws_bytes = need;
	return CORDIC_OK;
}

int cordic_stream_latency(const cordic_stream *s) { return s ? s->latency : CORDIC_ERR_ARGS; }

int cordic_stream_reset(cordic_stream *s, void *stream)
{
	if (!s)
		return CORDIC_ERR_ARGS;
	return (hipMemsetAsync(s->st.epoch, 0, 4,
			static_cast<hipStream_t>(stream)) == hipSuccess)
		? CORDIC_OK : CORDIC_ERR_DEVICE;
}

int cordic_stream_ticks(cordic_stream *s, size_t ticks, const uint8_t *d_ce,
		const uint8_t *d_reset, const uint8_t *d_aux, const int32_t *d_xval,
		const int32_t *d_yval, const uint32_t *d_phase, int32_t *d_out0,
		int32_t *d_out1, uint8_t *d_oaux, void *stream)
{
	if (!s)
		return CORDIC_ERR_ARGS;
	if (int rc = grow_workspace(&s->st.ws, &s->st.ws_bytes,
			stream_workspace_bytes(ticks), stream))
		return rc;
	return launch_stream_ticks(s->cfg, s->st, ticks, d_ce, d_reset, d_aux,
			d_xval, d_yval, d_phase, d_out0, d_out1, d_oaux, stream);
}

// ------------------------------------- handshake view, sequential cores
struct cordic_seq {
	cordic_config cfg;
	SeqState st;
};

void cordic_seq_destroy(cordic_seq *s)
{
	if (!s)
		return;
	SeqState &t = s->st;
	void *ptrs[] = { t.c, t.px, t.py, t.pph, t.paux, t.l0, t.l1, t.la,
			 t.violations, t.ws, t.lit };
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	delete s;
}

int cordic_seq_create(const cordic_config *cfg, cordic_seq **out)
{
	if (!cfg || !out)
		return CORDIC_ERR_ARGS;
	if (cfg->mode != CORDIC_SP2R && cfg->mode != CORDIC_SR2P)
		return CORDIC_ERR_MODE;
	if (!config_sane(*cfg))
		return CORDIC_ERR_ARGS;
	cordic_seq *s = new (std::nothrow) cordic_seq;
	if (!s)
		return CORDIC_ERR_NOMEM;
	s->cfg = *cfg;
	SeqState &t = s->st;
	auto zalloc = [](auto **p, size_t bytes) {
		return hipMalloc((void **)p, bytes) == hipSuccess
			&& hipMemset(*p, 0, bytes) == hipSuccess;
	};
	bool ok = zalloc(&t.violations, 8) && zalloc(&t.c, 4)
		&& zalloc(&t.px, 4) && zalloc(&t.py, 4) && zalloc(&t.pph, 4)
		&& zalloc(&t.paux, 4) && zalloc(&t.l0, 4) && zalloc(&t.l1, 4)
		&& zalloc(&t.la, 4);
	// register-level state for off-protocol stretches: power-on registers
	// and the padded arctan table
	std::vector<unsigned char> image(seq_literal_bytes());
	seq_literal_init(*cfg, image.data());
	ok = ok && hipMalloc(&t.lit, image.size()) == hipSuccess
		&& hipMemcpy(t.lit, image.data(), image.size(),
				hipMemcpyHostToDevice) == hipSuccess;
	if (!ok) {
		cordic_seq_destroy(s);
		return CORDIC_ERR_DEVICE;
	}
	*out = s;
	return CORDIC_OK;
}

size_t cordic_seq_workspace(size_t ticks) { return seq_workspace_bytes(ticks); }

int cordic_seq_reserve(cordic_seq *s, size_t max_ticks)
{
	if (!s)
		return CORDIC_ERR_ARGS;
	const size_t need = seq_workspace_bytes(max_ticks);
	if (need <= s->st.ws_bytes)
		return CORDIC_OK;
	if (hipDeviceSynchronize() != hipSuccess)
		return CORDIC_ERR_DEVICE;
	if (s->st.ws) (void)hipFree(s->st.ws);
	s->st.ws = nullptr;
	s->st.ws_bytes = 0;
	if (hipMalloc(&s->st.ws, need) != hipSuccess)
		return CORDIC_ERR_DEVICE;
	s->st.ws_bytes = need;
	return CORDIC_OK;
}

int cordic_seq_ticks(cordic_seq *s, size_t ticks, const uint8_t *d_stb,
		const uint8_t *d_reset, const uint8_t *d_aux, const int32_t *d_xval,
		const int32_t *d_yval, const uint32_t *d_phase, int32_t *d_out0,
		int32_t *d_out1, uint8_t *d_busy, uint8_t *d_done, uint8_t *d_oaux,
		void *stream)
{
	if (!s)
		return CORDIC_ERR_ARGS;
	if (int rc = grow_workspace(&s->st.ws, &s->st.ws_bytes,
			seq_workspace_bytes(ticks), stream))
		return rc;
	return launch_seq_ticks(s->cfg, s->st, ticks, d_stb, d_reset, d_aux, d_xval,
			d_yval, d_phase, d_out0, d_out1, d_busy, d_done, d_oaux,
			stream);
}

int cordic_seq_violations(cordic_seq *s, uint64_t *count)
{
	if (!s || !count)
		return CORDIC_ERR_ARGS;
	unsigned long long v = 0;
	if (hipDeviceSynchronize() != hipSuccess
			|| hipMemcpy(&v, s->st.violations, 8, hipMemcpyDeviceToHost)
				!= hipSuccess)
		return CORDIC_ERR_DEVICE;
	*count = v;
	return CORDIC_OK;
}

int cordic_table_lds_mode(const cordic_table *tbl)
{
	return tbl ? tbl->lds_mode : CORDIC_ERR_ARGS;
}

size_t cordic_dir_table(const cordic_config *cfg, uint32_t *buf, size_t cap_words)
{
	if (!cfg)
		return 0;
	if (!buf || cap_words == 0) {
		std::vector<uint32_t> tmp(4 + kDxMaxLevels * (6 + 2 * 4096 + 2 * 256));
		return build_dir_table(*cfg, tmp.data(), tmp.size(), nullptr);
	}
	return build_dir_table(*cfg, buf, cap_words, nullptr);
}

size_t cordic_seed_table(const cordic_config *cfg, uint32_t *buf, size_t cap_words)
{
	if (!cfg)
		return 0;
	if (!buf || cap_words == 0) {
		std::vector<uint32_t> tmp(4 + 4096 * 4 + 4096 * 2
				+ 4 + kDtMaxLevels * (6 + 2 * 4096 + 2 * 256));
		return build_seed_table(*cfg, CORDIC_SEED_STAGES, tmp.data(), tmp.size());
	}
	return build_seed_table(*cfg, CORDIC_SEED_STAGES, buf, cap_words);
}

extern "C" {

int cordic_abi_version(void) { return CORDIC_AMD_ABI_VERSION; }

const char *cordic_strerror(int status) { return status_text(status); }

int cordic_config_init(cordic_config *cfg, int mode, int iw, int ow, int xtra,
		int phase_bits, int nstages)
{
	return build_from_cli(cfg, mode, iw, ow, xtra, phase_bits, nstages);
}

int cordic_config_init_core(cordic_config *cfg, int mode, int nstages, int iw,
		int ow, int nxtra, int phase_bits)
{
	return build_core(cfg, mode, nstages, iw, ow, nxtra, phase_bits);
}

int cordic_config_from_args(cordic_config *cfg, int argc,
		const char *const *argv, char *fname, size_t fname_cap,
		int *c_header)
{
	return parse_args(cfg, argc, argv, fname, fname_cap, c_header);
}

int cordic_config_write_header(const cordic_config *cfg, const char *name,
		char *buf, size_t cap)
{
	return write_header(cfg, name, buf, cap);
}

int cordic_nextlg(unsigned vl) { return next_lg(vl); }
double cordic_gain(int nstages) { return rotation_gain(nstages); }
uint32_t cordic_gain_annihilator(int nstages) { return gain_annihilator(nstages); }
uint32_t cordic_config_gain_annihilator(const cordic_config *cfg)
{
	return cfg ? core_gain_annihilator(*cfg) : 0u;
}
double cordic_phase_variance(int nstages, int phase_bits)
{
	return phase_variance(nstages, phase_bits);
}
double cordic_transform_quantization_variance(int nstages, int xtrabits,
		int dropped_bits)
{
	return quantization_variance(nstages, xtrabits, dropped_bits);
}
int cordic_angles(int nstages, int phase_bits, uint32_t *out)
{
	if (!out || nstages < 0 || phase_bits < 3 || phase_bits > 32)
		return CORDIC_ERR_ARGS;
	for (int k = 0; k < nstages; k++)
		out[k] = arctan_entry((unsigned)k, phase_bits);
	return CORDIC_OK;
}
int cordic_calc_stages_ww(int working_width, int phase_bits)
{
	return stages_for(phase_bits, working_width);
}
int cordic_calc_stages(int phase_bits) { return stages_for(phase_bits, -1); }
int cordic_calc_phase_bits(int output_width) { return phase_bits_for(output_width); }

// ------------------------------------------------------------------ device

int cordic_p2r(const cordic_config *cfg, size_t n, const int32_t *d_xval,
		const int32_t *d_yval, const uint32_t *d_phase, int32_t *d_oxval,
		int32_t *d_oyval, void *stream)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	RotatorJob j;
	j.x = d_xval; j.y = d_yval; j.phase = d_phase;
	j.ox = d_oxval; j.oy = d_oyval; j.n = n;
	return launch_rotator(*cfg, Feed::PhaseArray_XYArray, j, stream);
}

int cordic_p2r_const(const cordic_config *cfg, size_t n, int32_t xval,
		int32_t yval, const uint32_t *d_phase, int32_t *d_oxval,
		int32_t *d_oyval, void *stream)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	RotatorJob j;
	j.x0 = xval; j.y0 = yval; j.phase = d_phase;
	j.ox = d_oxval; j.oy = d_oyval; j.n = n;
	return launch_rotator(*cfg, Feed::PhaseArray_ConstXY, j, stream);
}

int cordic_nco(const cordic_config *cfg, size_t n, uint32_t phase0, uint32_t fcw,
		uint64_t index0, int32_t xval, int32_t yval, int32_t *d_oxval,
		int32_t *d_oyval, void *stream)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	RotatorJob j;
	j.x0 = xval; j.y0 = yval; j.phase0 = phase0; j.fcw = fcw;
	j.index0 = index0; j.ox = d_oxval; j.oy = d_oyval; j.n = n;
	return launch_rotator(*cfg, Feed::Nco_ConstXY, j, stream);
}

int cordic_mix(const cordic_config *cfg, size_t n, uint32_t phase0, uint32_t fcw,
		uint64_t index0, const int32_t *d_xval, const int32_t *d_yval,
		int32_t *d_oxval, int32_t *d_oyval, void *stream)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	RotatorJob j;
	j.x = d_xval; j.y = d_yval;
	j.phase0 = phase0; j.fcw = fcw; j.index0 = index0; j.xy_nco = true;
	j.ox = d_oxval; j.oy = d_oyval; j.n = n;
	return launch_rotator(*cfg, Feed::PhaseArray_XYArray, j, stream);
}

int cordic_r2p(const cordic_config *cfg, size_t n, const int32_t *d_xval,
		const int32_t *d_yval, int32_t *d_omag, uint32_t *d_ophase,
		void *stream)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	return launch_topolar(*cfg, n, d_xval, d_yval, d_omag, d_ophase, stream);
}

// cordic_p2r_host / cordic_r2p_host: cordic_host.cpp (chunked copy pipeline)

int cordic_fill_phase_ramp(uint32_t *d_phase, size_t n, uint64_t index0,
		int shift, void *stream)
{
	return launch_fill_phase_ramp(d_phase, n, index0, shift, stream);
}

int cordic_fill_iq_ramp(int32_t *d_x, int32_t *d_y, size_t n, uint64_t index0,
		uint32_t mulx, uint32_t muly, int bits, void *stream)
{
	return launch_fill_iq_ramp(d_x, d_y, n, index0, mulx, muly, bits, stream);
}

int cordic_digest_u32(const uint32_t *d_words, size_t n, uint64_t index0,
		uint64_t *d_digest, void *stream)
{
	return launch_digest_u32(d_words, n, index0, d_digest, stream);
}

} // extern "C"
