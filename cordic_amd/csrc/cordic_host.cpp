// cordic_host.cpp -- the host-array entry points of the C ABI
// (cordic_p2r_host / cordic_r2p_host, include/cordic_amd.h): what a caller
// with the reference bench's plain `int` arrays (bench/cpp/cordic_tb.cpp:
// 94-100, topolar_tb.cpp:93-99) uses before it touches HIP itself.
//
// The kernels run two orders of magnitude faster than PCIe delivers their
// operands, so this path is a COPY PIPELINE with a
// kernel in the middle, and it is built to keep both PCIe directions busy:
//
//   * the job is cut into chunks of 2^22 samples (16 MiB per array); three
//     slots of device arrays rotate through three private non-blocking
//     streams -- upload, run, download -- chained by events on the device, so
//     chunk k+1 uploads and chunk k-1 downloads while chunk k computes;
//   * arrays the caller PINNED (cordic_host_alloc, hipHostMalloc,
//     hipHostRegister) are DMA'd in place: no CPU copy at all, the host only
//     enqueues and waits once at the end;
//   * pageable arrays (malloc / new, the reference's own) go through pinned
//     staging buffers, copied by a small pool of host threads (8) while the
//     DMA of the neighbouring chunks runs; the host paces itself so that at
//     most three chunks are queued (copy engines serve queued copies in
//     submission order: unpaced, downloads waiting for their kernels hold up
//     the uploads behind them);
//   * constant vectors (xy_is_scalar) run the table-seeded plan, cached with
//     the pipeline, like cordic_plan_p2r_const;
//   * only the pipeline's own streams are synchronised -- never the device.
//
// The pipeline (streams, device arrays, staging, threads, plan) is created on
// first use, one per device, kept for the life of the process or until
// cordic_host_release(), and serialised by a mutex: concurrent host-array
// calls on one device queue up behind each other (PCIe is the shared resource
// anyway).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "cordic_amd.h"
#include "cordic_internal.h"

namespace {

constexpr int kSlots = 3;
// samples per chunk and array: 2^22 (16 MiB) unless CORDIC_HOST_CHUNK_LOG2
// says otherwise (18..26; read once, when the first pipeline is made)
size_t chunk_samples()
{
	static const size_t v = [] {
		int lg = 22;
		if (const char *e = std::getenv("CORDIC_HOST_CHUNK_LOG2")) {
			const int x = std::atoi(e);
			if (x >= 18 && x <= 26)
				lg = x;
		}
		return (size_t)1 << lg;
	}();
	return v;
}
#define kChunk (chunk_samples())	/* (a run-time constant of the process) */
constexpr size_t kDirectBytes = (size_t)1 << 20;	// small jobs: no staging

bool ok(hipError_t e) { return e == hipSuccess; }

// ---------------------------------------------------------------- threads
// (The staging copies are plain memcpy: a hand-written copy with non-temporal
// stores measured 7-8 % SLOWER end to end than glibc's -- 0.80 against 0.86 of
// the PCIe rate with 4 or 8 threads, profiles/r04/host_threads.txt.)
// parallel copy for the staging copies: T-1 workers + the calling thread
class CopyPool {
public:
	explicit CopyPool(int threads)
	{
		for (int t = 1; t < threads; t++)
			workers_.emplace_back([this] { run(); });
	}
	~CopyPool()
	{
		{
			std::lock_guard<std::mutex> lk(mu_);
			stop_ = true;
		}
		cv_.notify_all();
		for (std::thread &t : workers_)
			t.join();
	}
	int threads() const { return (int)workers_.size() + 1; }
	void copy(void *dst, const void *src, size_t bytes)
	{
		const size_t parts = (size_t)threads();
		if (parts == 1 || bytes < ((size_t)1 << 20)) {
			std::memcpy(dst, src, bytes);
			return;
		}
		size_t per = (bytes + parts - 1) / parts;
		per = (per + 4095) & ~(size_t)4095;
		{
			std::lock_guard<std::mutex> lk(mu_);
			for (size_t off = per; off < bytes; off += per) {
				const size_t len = bytes - off < per ? bytes - off : per;
				tasks_.push_back({static_cast<char *>(dst) + off,
						static_cast<const char *>(src) + off, len});
				pending_++;
			}
		}
		cv_.notify_all();
		std::memcpy(dst, src, per < bytes ? per : bytes);
		// help with what is left, then wait for the stragglers
		for (;;) {
			Task t;
			{
				std::unique_lock<std::mutex> lk(mu_);
				if (tasks_.empty()) {
					done_.wait(lk, [this] { return pending_ == 0; });
					return;
				}
				t = tasks_.back();
				tasks_.pop_back();
			}
			std::memcpy(t.dst, t.src, t.len);
			finish();
		}
	}
private:
	struct Task { char *dst; const char *src; size_t len; };
	void finish()
	{
		std::lock_guard<std::mutex> lk(mu_);
		if (--pending_ == 0)
			done_.notify_all();
	}
	void run()
	{
		for (;;) {
			Task t;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [this] { return stop_ || !tasks_.empty(); });
				if (tasks_.empty())
					return;
				t = tasks_.back();
				tasks_.pop_back();
			}
			std::memcpy(t.dst, t.src, t.len);
			finish();
		}
	}
	std::mutex mu_;
	std::condition_variable cv_, done_;
	std::vector<Task> tasks_;
	std::vector<std::thread> workers_;
	size_t pending_ = 0;
	bool stop_ = false;
};

int pool_threads()
{
	if (const char *e = std::getenv("CORDIC_HOST_THREADS")) {
		const int v = std::atoi(e);
		if (v >= 1 && v <= 64)
			return v;
	}
	// 4 and 8 threads measured 0.86 of the PCIe rate, 6 and 12 0.75-0.78
	// (pieces of 16 MiB / threads: powers of two keep them page-sized)
	unsigned hw = std::thread::hardware_concurrency();
	if (hw == 0) hw = 4;
	return hw >= 16 ? 8 : (hw >= 8 ? 4 : 2);
}

// --------------------------------------------------------------- pipeline
struct Slot {
	void	*din[3] = {nullptr, nullptr, nullptr};	// phase / x, y  (device)
	void	*dout[2] = {nullptr, nullptr};
	void	*hin[3] = {nullptr, nullptr, nullptr};	// pinned staging
	void	*hout[2] = {nullptr, nullptr};
	hipEvent_t up = nullptr, done = nullptr, down = nullptr;
	bool	used = false;
};

struct HostPipe {
	std::mutex mu;
	int	device = 0;
	hipStream_t s_up = nullptr, s_run = nullptr, s_down = nullptr;
	Slot	slot[kSlots];
	CopyPool *pool = nullptr;
	cordic_plan *plan = nullptr;
	cordic_config plan_cfg;
	bool	have_plan = false;
	cordic_host_stats last = {};

	bool init()
	{
		if (s_up)
			return true;
		// high priority: the runtime keeps a separate pool of hardware
		// queues per priority, so these streams never sit in a hardware
		// queue behind a caller's long-running kernels (streams of equal
		// priority share the few hardware queues a process gets), and the
		// pipeline's short kernels do not wait for a busy device
		int lo = 0, hi = 0;
		if (!ok(hipDeviceGetStreamPriorityRange(&lo, &hi)))
			lo = hi = 0;
		if (!ok(hipStreamCreateWithPriority(&s_up, hipStreamNonBlocking, hi)) ||
		    !ok(hipStreamCreateWithPriority(&s_run, hipStreamNonBlocking, hi)) ||
		    !ok(hipStreamCreateWithPriority(&s_down, hipStreamNonBlocking, hi)))
			return false;
		for (Slot &s : slot)
			if (!ok(hipEventCreateWithFlags(&s.up, hipEventDisableTiming)) ||
			    !ok(hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) ||
			    !ok(hipEventCreateWithFlags(&s.down, hipEventDisableTiming)))
				return false;
		return true;
	}
	bool device_array(void *&p)
	{
		return p || ok(hipMalloc(&p, kChunk * 4));
	}
	bool staging(void *&p)
	{
		return p || ok(hipHostMalloc(&p, kChunk * 4, hipHostMallocDefault));
	}
	void release()
	{
		if (s_down) {
			(void)hipStreamSynchronize(s_up);
			(void)hipStreamSynchronize(s_run);
			(void)hipStreamSynchronize(s_down);
		}
		for (Slot &s : slot) {
			for (void *&p : s.din) { if (p) (void)hipFree(p); p = nullptr; }
			for (void *&p : s.dout) { if (p) (void)hipFree(p); p = nullptr; }
			for (void *&p : s.hin) { if (p) (void)hipHostFree(p); p = nullptr; }
			for (void *&p : s.hout) { if (p) (void)hipHostFree(p); p = nullptr; }
			if (s.up) (void)hipEventDestroy(s.up);
			if (s.done) (void)hipEventDestroy(s.done);
			if (s.down) (void)hipEventDestroy(s.down);
			s.up = s.done = s.down = nullptr;
			s.used = false;
		}
		if (s_up) (void)hipStreamDestroy(s_up);
		if (s_run) (void)hipStreamDestroy(s_run);
		if (s_down) (void)hipStreamDestroy(s_down);
		s_up = s_run = s_down = nullptr;
		if (plan) cordic_plan_destroy(plan);
		plan = nullptr;
		have_plan = false;
		delete pool;
		pool = nullptr;
	}
};

std::mutex g_pipes_mu;
// one pipeline per (device, lane): a lane is an entry of the device list of
// cordic_host_set_devices -- the same ordinal listed twice is two pipelines
// on one device (how the one-GPU tests drive the multi-device logic)
std::map<std::pair<int, int>, HostPipe *> g_pipes;
std::vector<int> g_devices;		// empty: the caller's current device
cordic_host_stats g_last_total = {};	// of the most recent multi-lane call

HostPipe *pipe_for(int dev, int lane)
{
	std::lock_guard<std::mutex> lk(g_pipes_mu);
	HostPipe *&p = g_pipes[std::make_pair(dev, lane)];
	if (!p) {
		p = new HostPipe;
		p->device = dev;
	}
	return p;
}

HostPipe *pipe_for_current_device()
{
	int dev = 0;
	if (!ok(hipGetDevice(&dev)))
		return nullptr;
	return pipe_for(dev, 0);
}

// Is this host array something the DMA engines can take as it is?  Pinned /
// registered host memory, managed or device memory: yes.  An address the
// runtime does not know is ordinary pageable memory.
bool dma_ready(const void *p)
{
	hipPointerAttribute_t a;
	std::memset(&a, 0, sizeof a);
	if (hipPointerGetAttributes(&a, p) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	return a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice
		|| a.type == hipMemoryTypeManaged;
}

struct HostJob {
	const cordic_config *cfg = nullptr;
	size_t	n = 0;
	int	nin = 0;			// input arrays per sample
	const void *in[3] = {nullptr, nullptr, nullptr};
	void	*out[2] = {nullptr, nullptr};
	bool	r2p = false, scalar = false;
	int32_t	x0 = 0, y0 = 0;
};

int run_pipeline(HostPipe &hp, const HostJob &j)
{
	const auto t_begin = std::chrono::steady_clock::now();
	if (!hp.init())
		return CORDIC_ERR_DEVICE;
	// the plan (seed table for constant vectors, direction tables for
	// per-sample ones), kept with the pipeline
	if (!j.r2p) {
		if (!hp.have_plan || std::memcmp(&hp.plan_cfg, j.cfg, sizeof *j.cfg) != 0) {
			if (hp.plan) cordic_plan_destroy(hp.plan);
			hp.plan = nullptr;
			hp.have_plan = false;
			if (int rc = cordic_plan_create(j.cfg, &hp.plan))
				return rc;
			hp.plan_cfg = *j.cfg;
			hp.have_plan = true;
		}
	}
	const bool small = j.n * 4 <= kDirectBytes;
	bool stage_in[3] = {false, false, false}, stage_out[2] = {false, false};
	bool any_stage = false;
	for (int a = 0; a < j.nin; a++)
		any_stage |= stage_in[a] = !small && !dma_ready(j.in[a]);
	for (int a = 0; a < 2; a++)
		any_stage |= stage_out[a] = !small && !dma_ready(j.out[a]);
	if (any_stage && !hp.pool)
		hp.pool = new CopyPool(pool_threads());

	const size_t nc = (j.n + kChunk - 1) / kChunk;
	auto span = [&](size_t c, size_t *off, size_t *cnt) {
		*off = c * kChunk;
		*cnt = j.n - *off < kChunk ? j.n - *off : kChunk;
	};
	auto retire = [&](size_t c) -> bool {	// staged outputs of chunk c -> caller
		Slot &s = hp.slot[c % kSlots];
		size_t off, cnt;
		span(c, &off, &cnt);
		if (!ok(hipEventSynchronize(s.down)))
			return false;
		for (int a = 0; a < 2; a++)
			if (stage_out[a])
				hp.pool->copy(static_cast<char *>(j.out[a]) + off * 4,
						s.hout[a], cnt * 4);
		return true;
	};
	const bool host_retires = stage_out[0] || stage_out[1];
	int rc = CORDIC_OK;
	size_t retired = 0;
	bool any_seeded = false;	// what really ran (cordic_last_kernel per chunk)
	for (size_t c = 0; c < nc && rc == CORDIC_OK; c++) {
		Slot &s = hp.slot[c % kSlots];
		size_t off, cnt;
		span(c, &off, &cnt);
		bool fine = true;
		for (int a = 0; a < j.nin && fine; a++)
			fine = hp.device_array(s.din[a]) && (!stage_in[a] || hp.staging(s.hin[a]));
		for (int a = 0; a < 2 && fine; a++)
			fine = hp.device_array(s.dout[a]) && (!stage_out[a] || hp.staging(s.hout[a]));
		// staging of this slot is free once the upload that last read it is
		if (fine && s.used && (stage_in[0] || stage_in[1] || stage_in[2]))
			fine = ok(hipEventSynchronize(s.up));
		// upload: behind the kernel that last read this slot's inputs
		if (fine && s.used)
			fine = ok(hipStreamWaitEvent(hp.s_up, s.done, 0));
		for (int a = 0; a < j.nin && fine; a++) {
			const char *src = static_cast<const char *>(j.in[a]) + off * 4;
			if (stage_in[a]) {
				hp.pool->copy(s.hin[a], src, cnt * 4);
				src = static_cast<const char *>(s.hin[a]);
			}
			fine = ok(hipMemcpyAsync(s.din[a], src, cnt * 4,
					hipMemcpyHostToDevice, hp.s_up));
		}
		fine = fine && ok(hipEventRecord(s.up, hp.s_up));
		// run: behind the upload and behind the download that last read
		// this slot's outputs
		fine = fine && ok(hipStreamWaitEvent(hp.s_run, s.up, 0));
		if (fine && s.used)
			fine = ok(hipStreamWaitEvent(hp.s_run, s.down, 0));
		if (!fine) {
			rc = CORDIC_ERR_DEVICE;
			break;
		}
		if (j.r2p)
			rc = cordic_r2p(j.cfg, cnt, static_cast<int32_t *>(s.din[0]),
					static_cast<int32_t *>(s.din[1]),
					static_cast<int32_t *>(s.dout[0]),
					static_cast<uint32_t *>(s.dout[1]), hp.s_run);
		else if (j.scalar)
			rc = cordic_plan_p2r_const(hp.plan, cnt, j.x0, j.y0,
					static_cast<uint32_t *>(s.din[0]),
					static_cast<int32_t *>(s.dout[0]),
					static_cast<int32_t *>(s.dout[1]), hp.s_run);
		else
			rc = cordic_plan_p2r(hp.plan, cnt, static_cast<int32_t *>(s.din[1]),
					static_cast<int32_t *>(s.din[2]),
					static_cast<uint32_t *>(s.din[0]),
					static_cast<int32_t *>(s.dout[0]),
					static_cast<int32_t *>(s.dout[1]), hp.s_run);
		if (rc != CORDIC_OK)
			break;
		any_seeded |= !j.r2p && j.scalar
			&& cordic_last_kernel() == CORDIC_KERNEL_SEEDED;
		fine = ok(hipEventRecord(s.done, hp.s_run))
			&& ok(hipStreamWaitEvent(hp.s_down, s.done, 0));
		for (int a = 0; a < 2 && fine; a++) {
			void *dst = stage_out[a] ? s.hout[a]
				: static_cast<void *>(static_cast<char *>(j.out[a]) + off * 4);
			fine = ok(hipMemcpyAsync(dst, s.dout[a], cnt * 4,
					hipMemcpyDeviceToHost, hp.s_down));
		}
		fine = fine && ok(hipEventRecord(s.down, hp.s_down));
		s.used = true;
		if (!fine) {
			rc = CORDIC_ERR_DEVICE;
			break;
		}
		// the host copies chunk c-2 out of its staging while c-1 and c
		// are on the wire
		if (host_retires && c + 1 >= (size_t)kSlots) {
			if (!retire(retired++))
				rc = CORDIC_ERR_DEVICE;
		} else if (!host_retires && c + 1 >= (size_t)kSlots) {
			// Nothing for the host to do (every array DMA'd in place) --
			// but it must not run ahead: with all 3 x nc copies queued at
			// once the copy engines serve them in submission order and a
			// download that waits for its kernel holds up the uploads
			// queued behind it (measured: 0.51 of the PCIe rate unpaced,
			// profiles/r04/host_paths_first.json).  Same pacing as the
			// staged path: chunk c-2 has landed before chunk c+1 is queued.
			if (!ok(hipEventSynchronize(hp.slot[(c + 1) % kSlots].down)))
				rc = CORDIC_ERR_DEVICE;
		}
	}
	while (rc == CORDIC_OK && host_retires && retired < nc)
		if (!retire(retired++))
			rc = CORDIC_ERR_DEVICE;
	// everything this call enqueued, and nothing else on the device
	const bool drained = ok(hipStreamSynchronize(hp.s_up))
		&& ok(hipStreamSynchronize(hp.s_run))
		&& ok(hipStreamSynchronize(hp.s_down));
	if (rc == CORDIC_OK && !drained)
		rc = CORDIC_ERR_DEVICE;
	if (rc != CORDIC_OK)
		(void)hipGetLastError();
	cordic_host_stats &st = hp.last;
	st.samples = j.n;
	st.chunks = (int32_t)nc;
	st.chunk_samples = (int32_t)kChunk;
	st.staged_inputs = 0;
	for (int a = 0; a < j.nin; a++) st.staged_inputs += stage_in[a] ? 1 : 0;
	st.staged_outputs = (stage_out[0] ? 1 : 0) + (stage_out[1] ? 1 : 0);
	st.copy_threads = any_stage && hp.pool ? hp.pool->threads() : 0;
	st.seeded_plan = any_seeded ? 1 : 0;	// a chunk ran the table-seeded kernel
	st.lanes = 1;
	st.seconds = std::chrono::duration<double>(
			std::chrono::steady_clock::now() - t_begin).count();
	return rc;
}

int submit(const HostJob &j)
{
	std::vector<int> devs;
	{
		std::lock_guard<std::mutex> lk(g_pipes_mu);
		devs = g_devices;
	}
	// too small to be worth a second PCIe link: one lane
	const size_t nchunks = (j.n + kChunk - 1) / kChunk;
	if (devs.size() <= 1 || nchunks < 2) {
		HostPipe *hp = devs.empty() ? pipe_for_current_device()
					    : pipe_for(devs[0], 0);
		if (!hp)
			return CORDIC_ERR_DEVICE;
		int prev = -1;
		if (!devs.empty()) {
			if (!ok(hipGetDevice(&prev))) prev = -1;
			if (!ok(hipSetDevice(devs[0])))
				return CORDIC_ERR_DEVICE;
		}
		int rc;
		{
			std::lock_guard<std::mutex> lk(hp->mu);
			rc = run_pipeline(*hp, j);
			std::lock_guard<std::mutex> lk2(g_pipes_mu);
			g_last_total = hp->last;
		}
		if (prev >= 0)
			(void)hipSetDevice(prev);
		return rc;
	}
	// Several devices: several PCIe links.  The job is cut into as many
	// contiguous parts (whole chunks) as there are lanes; every lane runs its
	// part through its own pipeline on its own device, from its own host
	// thread (a pipeline is a host-driven loop).  Samples are independent and
	// the parts' arrays do not overlap: nothing is shared but the host's
	// memory bandwidth.
	const auto t_begin = std::chrono::steady_clock::now();
	const size_t lanes = devs.size() < nchunks ? devs.size() : nchunks;
	std::vector<int> rcs(lanes, CORDIC_OK);
	std::vector<cordic_host_stats> st(lanes);
	std::vector<std::thread> th;
	auto part = [&](size_t l, size_t *off, size_t *cnt) {
		const size_t per = (nchunks + lanes - 1) / lanes * kChunk;
		*off = l * per < j.n ? l * per : j.n;
		*cnt = j.n - *off < per ? j.n - *off : per;
	};
	auto lane_fn = [&](size_t l) {
		size_t off, cnt;
		part(l, &off, &cnt);
		if (cnt == 0)
			return;
		if (!ok(hipSetDevice(devs[l]))) {
			rcs[l] = CORDIC_ERR_DEVICE;
			return;
		}
		HostPipe *hp = pipe_for(devs[l], (int)l);
		HostJob sub = j;
		sub.n = cnt;
		for (int a = 0; a < j.nin; a++)
			sub.in[a] = static_cast<const char *>(j.in[a]) + off * 4;
		for (int a = 0; a < 2; a++)
			sub.out[a] = static_cast<char *>(j.out[a]) + off * 4;
		std::lock_guard<std::mutex> lk(hp->mu);
		rcs[l] = run_pipeline(*hp, sub);
		st[l] = hp->last;
	};
	int prev = -1;
	if (!ok(hipGetDevice(&prev))) prev = -1;
	for (size_t l = 1; l < lanes; l++)
		th.emplace_back(lane_fn, l);
	lane_fn(0);
	for (std::thread &t : th)
		t.join();
	if (prev >= 0)
		(void)hipSetDevice(prev);
	cordic_host_stats tot = {};
	for (size_t l = 0; l < lanes; l++) {
		tot.samples += st[l].samples;
		tot.chunks += st[l].chunks;
		tot.chunk_samples = st[l].chunk_samples ? st[l].chunk_samples : tot.chunk_samples;
		tot.staged_inputs = st[l].staged_inputs > tot.staged_inputs
				? st[l].staged_inputs : tot.staged_inputs;
		tot.staged_outputs = st[l].staged_outputs > tot.staged_outputs
				? st[l].staged_outputs : tot.staged_outputs;
		tot.copy_threads += st[l].copy_threads;
		tot.seeded_plan |= st[l].seeded_plan;
	}
	tot.lanes = (int32_t)lanes;
	tot.seconds = std::chrono::duration<double>(
			std::chrono::steady_clock::now() - t_begin).count();
	{
		std::lock_guard<std::mutex> lk(g_pipes_mu);
		g_last_total = tot;
	}
	for (int rc : rcs)
		if (rc != CORDIC_OK)
			return rc;
	return CORDIC_OK;
}

} // namespace

extern "C" {

int cordic_p2r_host(const cordic_config *cfg, size_t n, const int32_t *xval,
		const int32_t *yval, int xy_is_scalar, const uint32_t *phase,
		int32_t *oxval, int32_t *oyval)
{
	if (!cfg || !xval || !yval || !phase || !oxval || !oyval)
		return CORDIC_ERR_ARGS;
	if (n == 0)
		return CORDIC_OK;
	HostJob j;
	j.cfg = cfg; j.n = n;
	j.in[0] = phase;
	j.out[0] = oxval; j.out[1] = oyval;
	if (xy_is_scalar) {
		j.nin = 1; j.scalar = true;
		j.x0 = xval[0]; j.y0 = yval[0];
	} else {
		j.nin = 3;
		j.in[1] = xval; j.in[2] = yval;
	}
	return submit(j);
}

int cordic_r2p_host(const cordic_config *cfg, size_t n, const int32_t *xval,
		const int32_t *yval, int32_t *omag, uint32_t *ophase)
{
	if (!cfg || !xval || !yval || !omag || !ophase)
		return CORDIC_ERR_ARGS;
	if (n == 0)
		return CORDIC_OK;
	HostJob j;
	j.cfg = cfg; j.n = n; j.nin = 2; j.r2p = true;
	j.in[0] = xval; j.in[1] = yval;
	j.out[0] = omag; j.out[1] = ophase;
	return submit(j);
}

int cordic_host_alloc(void **p, size_t bytes)
{
	if (!p)
		return CORDIC_ERR_ARGS;
	*p = nullptr;
	if (!ok(hipHostMalloc(p, bytes ? bytes : 4, hipHostMallocDefault))) {
		(void)hipGetLastError();
		return CORDIC_ERR_NOMEM;
	}
	return CORDIC_OK;
}

void cordic_host_free(void *p)
{
	if (p)
		(void)hipHostFree(p);
}

int cordic_host_last_stats(cordic_host_stats *out)
{
	if (!out)
		return CORDIC_ERR_ARGS;
	std::lock_guard<std::mutex> lk(g_pipes_mu);
	*out = g_last_total;
	return CORDIC_OK;
}

int cordic_host_lane_stats(int lane, cordic_host_stats *out)
{
	if (!out || lane < 0)
		return CORDIC_ERR_ARGS;
	int dev = 0;
	{
		std::lock_guard<std::mutex> lk(g_pipes_mu);
		if (g_devices.empty()) {
			if (lane != 0 || !ok(hipGetDevice(&dev)))
				return CORDIC_ERR_ARGS;
		} else {
			if ((size_t)lane >= g_devices.size())
				return CORDIC_ERR_ARGS;
			dev = g_devices[(size_t)lane];
		}
	}
	HostPipe *hp = pipe_for(dev, lane);
	std::lock_guard<std::mutex> lk(hp->mu);
	*out = hp->last;
	out->lanes = 1;
	return CORDIC_OK;
}

int cordic_host_set_devices(const int *devices, int count)
{
	if (count < 0 || count > 64 || (count > 0 && !devices))
		return CORDIC_ERR_ARGS;
	int ndev = 0;
	if (!ok(hipGetDeviceCount(&ndev))) {
		(void)hipGetLastError();
		return CORDIC_ERR_DEVICE;
	}
	for (int i = 0; i < count; i++)
		if (devices[i] < 0 || devices[i] >= ndev)
			return CORDIC_ERR_DEVICE;
	std::lock_guard<std::mutex> lk(g_pipes_mu);
	g_devices.assign(devices, devices + count);
	return CORDIC_OK;
}

void cordic_host_release(void)
{
	// every pipeline of the process (each on its own device)
	std::vector<HostPipe *> all;
	{
		std::lock_guard<std::mutex> lk(g_pipes_mu);
		for (auto &kv : g_pipes)
			all.push_back(kv.second);
	}
	int prev = -1;
	if (!ok(hipGetDevice(&prev))) prev = -1;
	for (HostPipe *hp : all) {
		std::lock_guard<std::mutex> lk(hp->mu);
		if (ok(hipSetDevice(hp->device)))
			hp->release();
	}
	if (prev >= 0)
		(void)hipSetDevice(prev);
}

} // extern "C"
