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
// exchange, data is read after the work already on `stream`.  Simplification:
// ncclGroupEnd completes the exchange before it returns (the real library
// only enqueues it), which is stricter than what the caller may rely on.
//
//   ncclGetUniqueId : a socket path prefix in the 128 id bytes
//   ncclCommInitRank: rank r listens on <prefix>.<r>; every pair gets a socket
//   ncclSend        : {IPC handle of the allocation, offset, bytes} to the peer,
//                     then wait for its acknowledgement
//   ncclRecv        : open the handle, hipMemcpy device-to-device, acknowledge
//   same-rank pairs : a plain device-to-device copy
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <vector>

namespace {

struct Comm {
	int rank = 0, nranks = 1;
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

thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;
std::map<std::string, void *> g_opened;		// IPC handle bytes -> mapping

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

ncclResult_t run_group()
{
	std::vector<Op> ops;
	ops.swap(g_ops);
	// the data a send reads is produced by work already on its stream
	for (const Op &o : ops)
		if (hipStreamSynchronize(o.stream) != hipSuccess)
			return ncclUnhandledCudaError;
	// 1. announce every send to another rank
	for (const Op &o : ops) {
		if (!o.send || o.peer == o.comm->rank)
			continue;
		void *base = nullptr; size_t size = 0;
		Msg m;
		if (hipMemGetAddressRange(&base, &size, o.buf) != hipSuccess ||
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
			if (hipMemcpy(r[k]->buf, s[k]->buf, s[k]->bytes,
					hipMemcpyDeviceToDevice) != hipSuccess)
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
			if (hipIpcOpenMemHandle(&base, m.handle,
					hipIpcMemLazyEnablePeerAccess) != hipSuccess)
				return ncclUnhandledCudaError;
			g_opened[key] = base;
		}
		if (hipMemcpy(o.buf, static_cast<char *>(base) + m.offset, m.bytes,
				hipMemcpyDeviceToDevice) != hipSuccess)
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
