// rccl_shim.cpp -- TEST INFRASTRUCTURE, not product: the seven RCCL entry
// points cordic_group's cross-process gather uses (cordic_group.cpp: Rccl),
// implemented over UNIX sockets + HIP IPC so that SEVERAL ranks can share ONE
// GPU.  Real RCCL refuses two ranks on one device, and the boxes this project
// can reach have one GPU, so without this the rank > 0 branches of
// rccl_forward and the piece geometry across processes would never execute
// before the first multi-GPU run.  Selected with CORDIC_RCCL_LIB=<this .so>.
//
// Semantics kept: point-to-point, sends and receives between a pair match in
// issue order, everything between ncclGroupStart and ncclGroupEnd is one
// exchange, and -- since round 4 -- the exchange is ASYNCHRONOUS AND
// STREAM-ORDERED like the real library's:
//   * ncclGroupEnd (and a bare ncclSend / ncclRecv) only ENQUEUES: it records
//     an event on every stream of the exchange (= the data a send reads is
//     produced by work already on its stream), puts a gate behind it on the
//     same stream (a one-wave kernel that waits for a word in pinned host
//     memory), hands the exchange to a helper thread and returns;
//   * the helper thread waits for the events, sleeps CORDIC_SHIM_DELAY_MS
//     (default 0; the tests inject 5 ms) so that the host is far ahead of the
//     transfers, moves the data on a private high-priority stream, and only
//     then opens the gates: work enqueued on those streams AFTER the exchange
//     runs after it, work on OTHER streams does not wait -- exactly what a
//     caller of RCCL may and may not rely on.
// A caller that reuses a send buffer, or reads a receive buffer, without
// ordering itself behind the exchange's stream now gets torn data here too.
// CORDIC_SHIM_SYNC=1 restores the round-3 behaviour (exchange complete before
// ncclGroupEnd returns) for A/B runs.  A gate gives up after
// CORDIC_SHIM_GATE_TIMEOUT_S (default 60) so that a broken exchange fails the
// test instead of hanging the GPU; the next call then returns an error.
//
//   ncclGetUniqueId : a socket path prefix in the 128 id bytes
//   ncclCommInitRank: rank r listens on <prefix>.<r>; every pair gets a socket
//   ncclSend        : {IPC handle of the allocation, offset, bytes} to the peer,
//                     then wait for its acknowledgement
//   ncclRecv        : open the handle, copy device-to-device, acknowledge
//   same-rank pairs : a plain device-to-device copy
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Comm {
	int rank = 0, nranks = 1, device = 0;
	std::string prefix;
	int listener = -1;
	std::vector<int> sock;		// per peer
};

struct Msg {
	hipIpcMemHandle_t handle;
	unsigned long long offset, bytes;
};

struct Op {
	bool send;
	void *buf;
	size_t bytes;
	int peer;
	Comm *comm;
	hipStream_t stream;
};

struct Job {
	unsigned long long seq = 0;
	std::vector<Op> ops;
	std::vector<hipEvent_t> ready;	// one per distinct stream
	std::vector<int> ready_dev;
};

// pinned host memory the gate kernels poll
struct Ctrl {
	volatile unsigned long long done;	// exchanges completed
	volatile unsigned int timed_out;	// a gate gave up
};

thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;
std::map<std::string, void *> g_opened;		// IPC handle bytes -> mapping

std::mutex g_mu;
std::condition_variable g_cv;
std::deque<Job *> g_queue;
std::thread g_helper;
bool g_helper_started = false, g_helper_stop = false;
unsigned long long g_issued = 0;		// exchanges handed to the helper
std::atomic<int> g_error{0};			// sticky ncclResult_t
Ctrl *g_ctrl = nullptr;
std::map<int, hipStream_t> g_private;		// device -> helper's stream

int env_int(const char *name, int dflt)
{
	const char *v = std::getenv(name);
	return (v && *v) ? std::atoi(v) : dflt;
}

// The gate: everything enqueued on the stream behind it waits until the helper
// has completed exchange `seq`.  wall_clock64() ticks at 100 MHz on gfx950.
__global__ void shim_gate(Ctrl *c, unsigned long long seq, unsigned long long timeout_ticks)
{
	const unsigned long long t0 = wall_clock64();
	while (__atomic_load_n(&c->done, __ATOMIC_ACQUIRE) < seq) {
		__builtin_amdgcn_s_sleep(64);
		if (wall_clock64() - t0 > timeout_ticks) {
			c->timed_out = 1;
			break;
		}
	}
}

bool io_all(int fd, void *p, size_t n, bool wr)
{
	char *c = static_cast<char *>(p);
	while (n) {
		const ssize_t k = wr ? write(fd, c, n) : read(fd, c, n);
		if (k <= 0)
			return false;
		c += k; n -= (size_t)k;
	}
	return true;
}

size_t type_bytes(ncclDataType_t t)
{
	switch (t) {
	case ncclInt8: case ncclUint8: return 1;
	case ncclFloat16: return 2;
	case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
	default: return 8;
	}
}

sockaddr_un addr_of(const std::string &prefix, int rank)
{
	sockaddr_un a;
	std::memset(&a, 0, sizeof a);
	a.sun_family = AF_UNIX;
	std::snprintf(a.sun_path, sizeof a.sun_path, "%s.%d", prefix.c_str(), rank);
	return a;
}

// device-to-device copy that does not touch the caller's streams (their gates
// may be closed) nor the null stream (it would wait for them)
hipError_t copy_dd(int device, void *dst, const void *src, size_t bytes, bool sync_mode)
{
	if (sync_mode)
		return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice);
	if (hipSetDevice(device) != hipSuccess)
		return hipErrorInvalidDevice;
	hipStream_t &st = g_private[device];
	if (!st) {
		int lo = 0, hi = 0;
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
		if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi) != hipSuccess)
			return hipErrorUnknown;
	}
	hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
	return e != hipSuccess ? e : hipStreamSynchronize(st);
}

// The exchange itself.  sync_mode: on the caller's thread, after synchronising
// its streams (round-3 behaviour); else on the helper thread, after `ready`.
ncclResult_t do_exchange(const std::vector<Op> &ops, bool sync_mode)
{
	// 1. announce every send to another rank
	for (const Op &o : ops) {
		if (!o.send || o.peer == o.comm->rank)
			continue;
		void *base = nullptr; size_t size = 0;
		Msg m;
		if (hipSetDevice(o.comm->device) != hipSuccess ||
		    hipMemGetAddressRange(&base, &size, o.buf) != hipSuccess ||
		    hipIpcGetMemHandle(&m.handle, base) != hipSuccess)
			return ncclUnhandledCudaError;
		m.offset = (unsigned long long)(static_cast<char *>(o.buf) - static_cast<char *>(base));
		m.bytes = o.bytes;
		if (!io_all(o.comm->sock[(size_t)o.peer], &m, sizeof m, true))
			return ncclSystemError;
	}
	// 2. same-rank pairs, in issue order
	{
		std::vector<const Op *> s, r;
		for (const Op &o : ops)
			if (o.peer == o.comm->rank)
				(o.send ? s : r).push_back(&o);
		if (s.size() != r.size())
			return ncclInvalidUsage;
		for (size_t k = 0; k < s.size(); k++) {
			if (s[k]->bytes != r[k]->bytes)
				return ncclInvalidUsage;
			if (copy_dd(r[k]->comm->device, r[k]->buf, s[k]->buf, s[k]->bytes,
					sync_mode) != hipSuccess)
				return ncclUnhandledCudaError;
		}
	}
	// 3. receives from other ranks: copy out of the sender's allocation
	for (const Op &o : ops) {
		if (o.send || o.peer == o.comm->rank)
			continue;
		const int fd = o.comm->sock[(size_t)o.peer];
		Msg m;
		if (!io_all(fd, &m, sizeof m, false))
			return ncclSystemError;
		if (m.bytes != o.bytes)
			return ncclInvalidUsage;
		const std::string key(reinterpret_cast<const char *>(&m.handle), sizeof m.handle);
		void *base = nullptr;
		auto it = g_opened.find(key);
		if (it != g_opened.end())
			base = it->second;
		else {
			if (hipSetDevice(o.comm->device) != hipSuccess ||
			    hipIpcOpenMemHandle(&base, m.handle,
					hipIpcMemLazyEnablePeerAccess) != hipSuccess)
				return ncclUnhandledCudaError;
			g_opened[key] = base;
		}
		if (copy_dd(o.comm->device, o.buf, static_cast<char *>(base) + m.offset,
				m.bytes, sync_mode) != hipSuccess)
			return ncclUnhandledCudaError;
		char ack = 1;
		if (!io_all(fd, &ack, 1, true))
			return ncclSystemError;
	}
	// 4. a send is complete once the receiver has the data
	for (const Op &o : ops) {
		if (!o.send || o.peer == o.comm->rank)
			continue;
		char ack = 0;
		if (!io_all(o.comm->sock[(size_t)o.peer], &ack, 1, false) || ack != 1)
			return ncclSystemError;
	}
	return ncclSuccess;
}

void helper_main()
{
	const int delay_ms = env_int("CORDIC_SHIM_DELAY_MS", 0);
	for (;;) {
		Job *j = nullptr;
		{
			std::unique_lock<std::mutex> lk(g_mu);
			g_cv.wait(lk, [] { return g_helper_stop || !g_queue.empty(); });
			if (g_queue.empty())
				return;
			j = g_queue.front();
		}
		ncclResult_t rc = ncclSuccess;
		// the data a send reads is produced by work already on its stream
		for (size_t k = 0; k < j->ready.size(); k++)
			if (hipSetDevice(j->ready_dev[k]) != hipSuccess ||
			    hipEventSynchronize(j->ready[k]) != hipSuccess)
				rc = ncclUnhandledCudaError;
		if (delay_ms > 0)
			usleep((useconds_t)delay_ms * 1000u);
		if (rc == ncclSuccess)
			rc = do_exchange(j->ops, false);
		if (rc != ncclSuccess) {
			std::fprintf(stderr, "[rccl_shim %d] exchange %llu failed: %d\n",
					(int)getpid(), j->seq, (int)rc);
			g_error.store((int)rc);
		}
		// open the gates, whatever happened (never leave a stream blocked)
		__atomic_store_n(&g_ctrl->done, j->seq, __ATOMIC_RELEASE);
		for (hipEvent_t e : j->ready)
			(void)hipEventDestroy(e);
		{
			std::lock_guard<std::mutex> lk(g_mu);
			g_queue.pop_front();
		}
		g_cv.notify_all();
		delete j;
	}
}

void drain()
{
	std::unique_lock<std::mutex> lk(g_mu);
	g_cv.wait(lk, [] { return g_queue.empty(); });
}

ncclResult_t run_group()
{
	std::vector<Op> ops;
	ops.swap(g_ops);
	if (int e = g_error.load())
		return (ncclResult_t)e;
	if (g_ctrl && g_ctrl->timed_out) {
		std::fprintf(stderr, "[rccl_shim %d] a gate timed out\n", (int)getpid());
		return ncclInternalError;
	}
	if (ops.empty())
		return ncclSuccess;
	if (env_int("CORDIC_SHIM_SYNC", 0)) {
		for (const Op &o : ops)
			if (hipStreamSynchronize(o.stream) != hipSuccess)
				return ncclUnhandledCudaError;
		return do_exchange(ops, true);
	}
	int dev0 = 0;
	(void)hipGetDevice(&dev0);
	if (!g_ctrl) {
		if (hipHostMalloc(reinterpret_cast<void **>(&g_ctrl), sizeof(Ctrl),
				hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess)
			return ncclUnhandledCudaError;
		g_ctrl->done = 0;
		g_ctrl->timed_out = 0;
	}
	Job *j = new Job;
	j->ops = ops;
	// every ready event first, then the gates: with streams that share a
	// hardware queue no event may sit behind a closed gate of this exchange
	std::vector<hipStream_t> streams;
	std::vector<int> sdev;
	for (const Op &o : ops) {
		bool seen = false;
		for (size_t k = 0; k < streams.size(); k++)
			seen = seen || (streams[k] == o.stream && sdev[k] == o.comm->device);
		if (!seen) {
			streams.push_back(o.stream);
			sdev.push_back(o.comm->device);
		}
	}
	ncclResult_t rc = ncclSuccess;
	for (size_t k = 0; k < streams.size(); k++) {
		hipEvent_t e = nullptr;
		if (hipSetDevice(sdev[k]) != hipSuccess ||
		    hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess ||
		    hipEventRecord(e, streams[k]) != hipSuccess)
			rc = ncclUnhandledCudaError;
		if (e) {
			j->ready.push_back(e);
			j->ready_dev.push_back(sdev[k]);
		}
	}
	const unsigned long long ticks = 100000000ull *
		(unsigned long long)env_int("CORDIC_SHIM_GATE_TIMEOUT_S", 60);
	{
		std::lock_guard<std::mutex> lk(g_mu);
		j->seq = ++g_issued;
	}
	for (size_t k = 0; k < streams.size() && rc == ncclSuccess; k++) {
		if (hipSetDevice(sdev[k]) != hipSuccess)
			rc = ncclUnhandledCudaError;
		hipLaunchKernelGGL(shim_gate, dim3(1), dim3(1), 0, streams[k], g_ctrl,
				j->seq, ticks);
		if (hipGetLastError() != hipSuccess)
			rc = ncclUnhandledCudaError;
	}
	(void)hipSetDevice(dev0);
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (!g_helper_started) {
			g_helper = std::thread(helper_main);
			g_helper_started = true;
		}
		g_queue.push_back(j);	// even after an error: the gates must open
	}
	g_cv.notify_all();
	return rc;
}

ncclResult_t post(bool send, void *buf, size_t count, ncclDataType_t t, int peer,
		ncclComm_t comm, hipStream_t st)
{
	Comm *c = reinterpret_cast<Comm *>(comm);
	if (!c || peer < 0 || peer >= c->nranks)
		return ncclInvalidArgument;
	g_ops.push_back(Op{send, buf, count * type_bytes(t), peer, c, st});
	return g_depth ? ncclSuccess : run_group();
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
	if (!id)
		return ncclInvalidArgument;
	std::memset(id, 0, sizeof *id);
	timespec ts;
	clock_gettime(CLOCK_REALTIME, &ts);
	std::snprintf(id->internal, sizeof id->internal, "/tmp/cordic_rccl_shim.%d.%ld",
			(int)getpid(), (long)ts.tv_nsec);
	return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
	if (!comm || nranks < 1 || rank < 0 || rank >= nranks)
		return ncclInvalidArgument;
	Comm *c = new Comm;
	c->rank = rank; c->nranks = nranks;
	(void)hipGetDevice(&c->device);		// the communicator's device
	c->prefix.assign(id.internal, strnlen(id.internal, sizeof id.internal));
	c->sock.assign((size_t)nranks, -1);
	sockaddr_un me = addr_of(c->prefix, rank);
	unlink(me.sun_path);
	c->listener = socket(AF_UNIX, SOCK_STREAM, 0);
	if (c->listener < 0 || bind(c->listener, (sockaddr *)&me, sizeof me) ||
	    listen(c->listener, nranks))
		return ncclSystemError;
	for (int p = 0; p < rank; p++) {		// connect to the lower ranks
		sockaddr_un a = addr_of(c->prefix, p);
		int fd = -1;
		for (int tries = 0; tries < 6000; tries++) {	// up to 60 s
			fd = socket(AF_UNIX, SOCK_STREAM, 0);
			if (fd >= 0 && connect(fd, (sockaddr *)&a, sizeof a) == 0)
				break;
			if (fd >= 0) close(fd);
			fd = -1;
			usleep(10000);
		}
		int32_t r = rank;
		if (fd < 0 || !io_all(fd, &r, sizeof r, true))
			return ncclSystemError;
		c->sock[(size_t)p] = fd;
	}
	for (int k = rank + 1; k < nranks; k++) {	// accept the higher ones
		const int fd = accept(c->listener, nullptr, nullptr);
		int32_t r = -1;
		if (fd < 0 || !io_all(fd, &r, sizeof r, false) || r <= rank || r >= nranks)
			return ncclSystemError;
		c->sock[(size_t)r] = fd;
	}
	*comm = reinterpret_cast<ncclComm_t>(c);
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
	Comm *c = reinterpret_cast<Comm *>(comm);
	if (!c)
		return ncclSuccess;
	drain();			// exchanges in flight still use the sockets
	for (int fd : c->sock)
		if (fd >= 0) close(fd);
	if (c->listener >= 0) close(c->listener);
	sockaddr_un me = addr_of(c->prefix, c->rank);
	unlink(me.sun_path);
	delete c;
	return ncclSuccess;
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
	if (g_depth <= 0)
		return ncclInvalidUsage;
	return --g_depth ? ncclSuccess : run_group();
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer,
		ncclComm_t comm, hipStream_t st)
{
	return post(true, const_cast<void *>(buf), count, t, peer, comm, st);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer,
		ncclComm_t comm, hipStream_t st)
{
	return post(false, buf, count, t, peer, comm, st);
}

} // extern "C"

// the helper thread must not outlive the library's globals
namespace {
struct AtExit {
	~AtExit()
	{
		if (!g_helper_started)
			return;
		{
			std::unique_lock<std::mutex> lk(g_mu);
			g_cv.wait(lk, [] { return g_queue.empty(); });
			g_helper_stop = true;
		}
		g_cv.notify_all();
		g_helper.join();
	}
} g_at_exit;
}
