// cordic_stream.hip -- clock-accurate view of the PIPELINED cores
// (rtl/cordic.v, rtl/topolar.v) for benches that drive i_ce / i_reset / i_aux
// per clock (bench/cpp/testb.h:87-106 tick(); bench/cpp/cordic_tb.cpp:136-176).
//
// A Verilated model is stepped one tick() at a time.  On a GPU the unit of
// work is a BLOCK OF CLOCKS: the caller hands over the port activity of T
// consecutive clocks as arrays and receives the output ports as they stand
// after each of those clocks; the pipeline contents are carried from call to
// call inside the cordic_stream object.
//
// Why this is exact.  Every register of the pipelined cores is either cleared
// (i_reset, which wins over i_ce: rtl/cordic.v:118-124,244-252,304-309),
// held (i_ce low) or loaded from the previous stage (i_ce high).  So after
// any clock the output register holds f(sample that entered L-1 advancing
// clocks before the most recent advancing clock), L = NSTAGES+2 registers
// (pre-rotation, NSTAGES stages, output; the aux shift register
// rtl/cordic.v:100-105,313 has the same length), where an ADVANCING clock is
// one with i_ce && !i_reset.  With e = number of advancing clocks since the
// last reset:
//    e == 0          outputs hold their reset value (0)
//    1 <= e <= L-1   the output register holds what the cleared registers turn
//                    into on their way out: x = y = 0 stays 0 through every
//                    stage and through the rounding, so o_xval/o_yval/o_mag
//                    = 0 and o_aux = 0, while topolar's phase accumulator
//                    still adds angle[i] at every stage it passes (y >= 0
//                    branch, rtl/topolar.v:235-243):
//                    o_phase = sum of angle[i], i = NSTAGES-e+1 .. NSTAGES-1
//    e >= L          f(input of the advancing clock with ordinal e-L+1)
// The clock -> source-sample map is two prefix scans (count of advancing
// clocks, index of the last reset) plus a compaction; f() itself is the batch
// kernel of the core, run on the gathered samples.
#include <hip/hip_runtime.h>

#include <new>

#include <cstdint>

#include "cordic_device.h"
#include "cordic_internal.h"

namespace cordic_amd {
namespace {
using namespace dev;

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;	// clocks per block

struct TickFlags {
	const uint8_t *ce;	// NULL: i_ce = 1 on every clock
	const uint8_t *reset;	// NULL: never reset
	__device__ __forceinline__ bool is_reset(uint32_t t) const
	{
		return reset && reset[t] != 0;
	}
	__device__ __forceinline__ bool advances(uint32_t t) const
	{
		return (!ce || ce[t] != 0) && !is_reset(t);
	}
};

// block-wide inclusive scan of one value per thread (sum) and of the running
// maximum of another, over kScanThreads threads
__device__ __forceinline__ void block_scan(uint32_t &sum, int32_t &mx,
		uint32_t *lds_sum, int32_t *lds_max)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t s = __shfl_up(sum, off, 64);
		const int32_t m = __shfl_up(mx, off, 64);
		if (lane >= off) {
			sum += s;
			mx = (m > mx) ? m : mx;
		}
	}
	if (lane == 63) {
		lds_sum[wave] = sum;
		lds_max[wave] = mx;
	}
	__syncthreads();
	for (int w = 0; w < wave; w++) {
		sum += lds_sum[w];
		mx = (lds_max[w] > mx) ? lds_max[w] : mx;
	}
	__syncthreads();
}

// pass 1: per tile, the number of advancing clocks and the last reset clock
__global__ __launch_bounds__(kScanThreads) void stream_tile_totals(TickFlags f,
		uint32_t T, uint32_t *tile_adv, int32_t *tile_last)
{
	__shared__ uint32_t lds_sum[kScanThreads / 64];
	__shared__ int32_t lds_max[kScanThreads / 64];
	const uint32_t t0 = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
	uint32_t sum = 0;
	int32_t last = -1;
#pragma unroll
	for (int j = 0; j < kScanItems; j++) {
		const uint32_t t = t0 + j;
		if (t < T) {
			if (f.is_reset(t)) last = (int32_t)t;
			sum += f.advances(t) ? 1u : 0u;
		}
	}
	block_scan(sum, last, lds_sum, lds_max);
	if (threadIdx.x == kScanThreads - 1) {
		tile_adv[blockIdx.x] = sum;
		tile_last[blockIdx.x] = last;
	}
}

// pass 2 (one block): exclusive scan of the tile totals, in place
__global__ __launch_bounds__(kScanThreads) void stream_tile_spine(
		uint32_t *tile_adv, int32_t *tile_last, uint32_t ntiles)
{
	__shared__ uint32_t lds_sum[kScanThreads / 64];
	__shared__ int32_t lds_max[kScanThreads / 64];
	__shared__ uint32_t carry_sum;
	__shared__ int32_t carry_max;
	if (threadIdx.x == 0) {
		carry_sum = 0;
		carry_max = -1;
	}
	__syncthreads();
	for (uint32_t base = 0; base < ntiles; base += kScanThreads) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t own_s = (i < ntiles) ? tile_adv[i] : 0u;
		const int32_t own_m = (i < ntiles) ? tile_last[i] : -1;
		uint32_t s = own_s;
		int32_t m = own_m;
		block_scan(s, m, lds_sum, lds_max);
		const uint32_t cs = carry_sum;
		const int32_t cm = carry_max;
		const int32_t prev = __shfl_up(m, 1, 64);
		__syncthreads();
		if (i < ntiles) {
			// exclusive: everything before tile i
			tile_adv[i] = cs + s - own_s;
			int32_t before = cm;
			if ((threadIdx.x & 63) != 0)
				before = (prev > before) ? prev : before;
			else
				for (int w = 0; w < (int)(threadIdx.x >> 6); w++)
					before = (lds_max[w] > before) ? lds_max[w] : before;
			tile_last[i] = before;
		}
		if (threadIdx.x == kScanThreads - 1) {
			carry_sum = cs + s;
			carry_max = (m > cm) ? m : cm;
		}
		__syncthreads();
	}
}

// pass 3: A[t] = advancing clocks in [0, t], R[t] = last reset clock <= t
// (-1: none), pos[k] = the clock of the advancing clock with ordinal k (0-based)
__global__ __launch_bounds__(kScanThreads) void stream_tile_apply(TickFlags f,
		uint32_t T, const uint32_t *tile_adv, const int32_t *tile_last,
		uint32_t *A, int32_t *R, uint32_t *pos)
{
	__shared__ uint32_t lds_sum[kScanThreads / 64];
	__shared__ int32_t lds_max[kScanThreads / 64];
	const uint32_t t0 = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
	uint32_t cnt[kScanItems];
	int32_t lst[kScanItems];
	uint32_t sum = 0;
	int32_t last = -1;
#pragma unroll
	for (int j = 0; j < kScanItems; j++) {
		const uint32_t t = t0 + j;
		if (t < T) {
			if (f.is_reset(t)) last = (int32_t)t;
			sum += f.advances(t) ? 1u : 0u;
		}
		cnt[j] = sum;
		lst[j] = last;
	}
	uint32_t inc_s = sum;
	int32_t inc_m = last;
	block_scan(inc_s, inc_m, lds_sum, lds_max);
	const uint32_t before_s = tile_adv[blockIdx.x] + inc_s - sum;
	// running maximum over the threads before this one
	int32_t before_m = __shfl_up(inc_m, 1, 64);
	if ((threadIdx.x & 63) == 0) {
		before_m = -1;
		for (int w = 0; w < (int)(threadIdx.x >> 6); w++)
			before_m = (lds_max[w] > before_m) ? lds_max[w] : before_m;
	}
	const int32_t tl = tile_last[blockIdx.x];
	before_m = (tl > before_m) ? tl : before_m;
#pragma unroll
	for (int j = 0; j < kScanItems; j++) {
		const uint32_t t = t0 + j;
		if (t < T) {
			const uint32_t a = before_s + cnt[j];
			A[t] = a;
			R[t] = (lst[j] > before_m) ? lst[j] : before_m;
			if (f.advances(t))
				pos[a - 1] = t;
		}
	}
}

struct StreamView {
	const int32_t *x, *y;		// block inputs (ports, one value per clock)
	const uint32_t *phase;		// NULL for topolar
	const uint8_t *aux;		// NULL: i_aux = 0
	const int32_t *hx, *hy;		// history: the L samples in the pipeline
					// when the block starts (oldest first)
	const uint32_t *hph;
	const uint8_t *haux;
	const uint32_t *epoch;		// advancing clocks since the last reset
					// before this block (saturated)
	const uint32_t *A;		// NULL: every clock advances
	const int32_t *R;
	const uint32_t *pos;
	uint32_t T;
	int32_t L;			// NSTAGES + 2
};

constexpr uint32_t kEpochSat = 1u << 30;

// e and, where e >= L, where the sample now in the output register came from
__device__ __forceinline__ uint32_t resolve(const StreamView &v, uint32_t t,
		bool &from_hist, uint32_t &src)
{
	const uint32_t a = v.A ? v.A[t] : t + 1;
	const int32_t r = v.R ? v.R[t] : -1;
	uint32_t e;
	if (r >= 0) {
		e = a - v.A[r];
	} else {
		const uint32_t e0 = *v.epoch;
		e = (e0 >= kEpochSat || a >= kEpochSat) ? kEpochSat : e0 + a;
	}
	from_hist = false;
	src = 0;
	if (e >= (uint32_t)v.L) {
		const int64_t k = (int64_t)a - (v.L - 1);	// ordinal, 1-based
		if (k >= 1) {
			src = v.pos ? v.pos[k - 1] : (uint32_t)(k - 1);
		} else {			// k in -(L-1) .. 0
			from_hist = true;
			src = (uint32_t)(v.L + k - 1);
		}
	}
	return e;
}

// gather the sample behind every clock's output; born[t] = 0 for a real
// sample, e (1..L-1) for a cleared register on its way out, 255 for e == 0
__global__ __launch_bounds__(kBlock) void stream_gather(StreamView v,
		int32_t *gx, int32_t *gy, uint32_t *gph, uint8_t *born,
		uint8_t *oaux)
{
	const uint32_t stride = gridDim.x * kBlock;
	for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < v.T; t += stride) {
		bool hist;
		uint32_t src;
		const uint32_t e = resolve(v, t, hist, src);
		int32_t x = 0, y = 0;
		uint32_t ph = 0;
		uint8_t ax = 0, b = 0;
		if (e >= (uint32_t)v.L) {
			if (hist) {
				x = v.hx[src]; y = v.hy[src];
				ph = v.hph ? v.hph[src] : 0u;
				ax = v.haux[src];
			} else {
				x = v.x[src]; y = v.y[src];
				ph = v.phase ? v.phase[src] : 0u;
				ax = v.aux ? (uint8_t)(v.aux[src] != 0) : (uint8_t)0;
			}
		} else {
			b = (e == 0) ? (uint8_t)255 : (uint8_t)e;
		}
		gx[t] = x;
		gy[t] = y;
		if (gph) gph[t] = ph;
		born[t] = b;
		if (oaux) oaux[t] = ax;
	}
}

__global__ __launch_bounds__(kBlock) void stream_shift_aux(const uint8_t *aux,
		uint8_t *oaux, uint32_t n)
{
	const uint32_t stride = gridDim.x * kBlock;
	for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < n; t += stride)
		oaux[t] = (uint8_t)(aux[t] != 0);
}

// topolar: o_phase of the clocks whose output register holds a cleared stage
__global__ __launch_bounds__(kBlock) void stream_fix_phase(uint32_t T,
		const uint8_t *born, const uint32_t *born_phase, uint32_t *ophase)
{
	const uint32_t stride = gridDim.x * kBlock;
	for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < T; t += stride) {
		const uint8_t b = born[t];
		if (b)
			ophase[t] = (b == 255) ? 0u : born_phase[b];
	}
}

// new history = the last L advancing samples of (old history ++ this block).
// In place: one block of at least L threads, every thread reads what its slot
// needs, then all write (so the object has a single set of state buffers and a
// captured graph of cordic_stream_ticks can be replayed).
__global__ __launch_bounds__(128) void stream_carry(StreamView v, int32_t *nx,
		int32_t *ny, uint32_t *nph, uint8_t *naux, uint32_t *nepoch)
{
	const int H = v.L;
	const uint32_t adv = v.A ? v.A[v.T - 1] : v.T;	// advancing clocks here
	const int j = threadIdx.x;
	int32_t x = 0, y = 0;
	uint32_t ph = 0, e = 0;
	uint8_t ax = 0;
	if (j < H) {
		// ordinal (1-based, this block) of history slot j
		const int64_t k = (int64_t)adv - H + 1 + j;
		if (k >= 1) {
			const uint32_t s = v.pos ? v.pos[k - 1] : (uint32_t)(k - 1);
			x = v.x[s]; y = v.y[s];
			ph = v.phase ? v.phase[s] : 0u;
			ax = v.aux ? (uint8_t)(v.aux[s] != 0) : (uint8_t)0;
		} else {
			const int64_t o = H + k - 1;	// slot of the old history
			if (o >= 0) {
				x = v.hx[o]; y = v.hy[o];
				ph = v.hph[o]; ax = v.haux[o];
			}
		}
	}
	if (j == 0) {
		const int32_t r = v.R ? v.R[v.T - 1] : -1;
		if (r >= 0) {
			e = adv - v.A[r];
		} else {
			const uint32_t e0 = *v.epoch;
			e = (e0 >= kEpochSat || adv >= kEpochSat) ? kEpochSat : e0 + adv;
		}
		if (e > kEpochSat)
			e = kEpochSat;
	}
	__syncthreads();
	if (j < H) {
		nx[j] = x; ny[j] = y; nph[j] = ph; naux[j] = ax;
	}
	if (j == 0)
		*nepoch = e;
}

int grid_1d(size_t n)
{
	const size_t b = (n + kBlock - 1) / kBlock;
	return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

} // namespace

size_t stream_workspace_bytes(size_t T)
{
	const size_t ntiles = (T + kScanTile - 1) / kScanTile;
	// A, R, pos, gx, gy, gph (4 bytes each), born (1), two tile arrays
	return T * 25 + ntiles * 8 + 256;
}

int launch_stream_ticks(const cordic_config &cfg, StreamState &s, size_t T,
		const uint8_t *ce, const uint8_t *reset, const uint8_t *aux,
		const int32_t *x, const int32_t *y, const uint32_t *phase,
		int32_t *o0, int32_t *o1, uint8_t *oaux, void *stream)
{
	(void)hipGetLastError();
	if (T == 0)
		return CORDIC_OK;
	if (T >= (1ull << 31))
		return CORDIC_ERR_ARGS;
	const bool rot = (cfg.mode == CORDIC_P2R);
	if (!rot && cfg.mode != CORDIC_R2P)
		return CORDIC_ERR_MODE;
	if (!x || !y || !o0 || !o1 || (rot && !phase))
		return CORDIC_ERR_ARGS;
	if (stream_workspace_bytes(T) > s.ws_bytes)
		return CORDIC_ERR_ARGS;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const uint32_t n = (uint32_t)T;

	// carve the workspace (every array 16-byte aligned)
	char *w = static_cast<char *>(s.ws);
	auto take = [&](size_t bytes) {
		char *p = w;
		w += (bytes + 15) & ~(size_t)15;
		return p;
	};
	const uint32_t ntiles = (n + kScanTile - 1) / kScanTile;
	int32_t *gx = reinterpret_cast<int32_t *>(take((size_t)n * 4));
	int32_t *gy = reinterpret_cast<int32_t *>(take((size_t)n * 4));
	uint32_t *gph = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
	uint8_t *born = reinterpret_cast<uint8_t *>(take(n));
	uint32_t *A = nullptr, *pos = nullptr;
	int32_t *R = nullptr;

	if (ce || reset) {
		A = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
		R = reinterpret_cast<int32_t *>(take((size_t)n * 4));
		pos = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
		uint32_t *tile_adv = reinterpret_cast<uint32_t *>(take((size_t)ntiles * 4));
		int32_t *tile_last = reinterpret_cast<int32_t *>(take((size_t)ntiles * 4));
		TickFlags f{ce, reset};
		hipLaunchKernelGGL(stream_tile_totals, dim3(ntiles), dim3(kScanThreads),
				0, st, f, n, tile_adv, tile_last);
		hipLaunchKernelGGL(stream_tile_spine, dim3(1), dim3(kScanThreads), 0,
				st, tile_adv, tile_last, ntiles);
		hipLaunchKernelGGL(stream_tile_apply, dim3(ntiles), dim3(kScanThreads),
				0, st, f, n, tile_adv, tile_last, A, R, pos);
	}

	const int L = cfg.nstages + 2;
	StreamView v{x, y, rot ? phase : nullptr, aux, s.hx, s.hy, s.hph, s.haux,
			s.epoch, A, R, pos, n, L};
	// Every clock enabled: from clock L-1 on the output register holds
	// f(input of clock t-L+1) whatever came before the block, so the bulk is
	// the batch kernel on the caller's arrays displaced by L-1 samples (the
	// vector kernels take element-aligned pointers); only the first L-1
	// clocks need the history.
	const bool direct = (A == nullptr) && n > (uint32_t)(L - 1);
	const uint32_t head = direct ? (uint32_t)(L - 1) : n;
	StreamView hv = v;
	hv.T = head;
	hipLaunchKernelGGL(stream_gather, dim3(grid_1d(head)), dim3(kBlock), 0, st,
			hv, gx, gy, rot ? gph : nullptr, born, oaux);
	if (hipGetLastError() != hipSuccess)
		return CORDIC_ERR_DEVICE;

	int rc;
	if (rot) {
		RotatorJob j;
		j.x = gx; j.y = gy; j.phase = gph;
		j.ox = o0; j.oy = o1; j.n = head;
		rc = launch_rotator(cfg, Feed::PhaseArray_XYArray, j, stream);
		if (rc == CORDIC_OK && direct) {
			RotatorJob b;
			b.x = x; b.y = y; b.phase = phase;
			b.ox = o0 + head; b.oy = o1 + head; b.n = n - head;
			rc = launch_rotator(cfg, Feed::PhaseArray_XYArray, b, stream);
		}
	} else {
		rc = launch_topolar(cfg, head, gx, gy, o0,
				reinterpret_cast<uint32_t *>(o1), stream);
		if (rc == CORDIC_OK)
			hipLaunchKernelGGL(stream_fix_phase, dim3(grid_1d(head)),
					dim3(kBlock), 0, st, head, born, s.born_phase,
					reinterpret_cast<uint32_t *>(o1));
		if (rc == CORDIC_OK && direct)
			rc = launch_topolar(cfg, n - head, x, y, o0 + head,
					reinterpret_cast<uint32_t *>(o1) + head, stream);
	}
	if (rc != CORDIC_OK)
		return rc;
	if (direct && oaux) {
		// o_aux[t] = i_aux[t-L+1]
		if (aux)
			hipLaunchKernelGGL(stream_shift_aux, dim3(grid_1d(n - head)),
					dim3(kBlock), 0, st, aux, oaux + head, n - head);
		else if (hipMemsetAsync(oaux + head, 0, n - head, st) != hipSuccess)
			return CORDIC_ERR_DEVICE;
	}
	hipLaunchKernelGGL(stream_carry, dim3(1), dim3(128), 0, st, v, s.hx, s.hy,
			s.hph, s.haux, s.epoch);		// in place
	return (hipGetLastError() == hipSuccess) ? CORDIC_OK : CORDIC_ERR_DEVICE;
}


// ===================================================== sequential cores
//
// rtl/seqcordic.v:226-327 / rtl/seqpolar.v:211-307: the i_stb / o_busy /
// o_done handshake, as a bench that keeps to the protocol sees it.  With
// C = CLOCKS_PER_OUTPUT and c = clocks left until the result registers load
// (0 = idle), one clock is
//     load   = (c == 1)                 o_xval/o_yval/o_aux load, reset or not
//     reset  : c' = 0
//     idle   : i_stb ? accept, c' = C-1 : c' = 0
//     busy   : c' = c-1, o_done = load; i_stb ignored -- except ON the load
//              clock, where the RTL keeps `idle` low and runs its datapath
//              again over its own result: counted as a violation, not
//              reproduced
//     o_busy = (c' != 0)
// a finite-state machine over the clock stream.  It is run in three passes:
// every tile of clocks is simulated from each of the <= 67 possible entry
// states (one thread each), a single wave chains the tiles' entry states, and
// one thread per tile replays its tile from the true entry state writing the
// per-clock flags.  From there on the data path is the pipelined cores': scan
// the accept / load flags, gather the sample behind every clock's output
// registers, run the core's batch kernel, patch in the values carried from
// earlier calls.  Pinned by tests/golden/seq_traces.json (vsim.py traces of
// the emitted RTL).
namespace {

constexpr int kFsmTile = 1024;		// clocks per tile
constexpr int kFsmStates = 72;		// >= max CLOCKS_PER_OUTPUT (64 + 3)
enum : uint32_t { EV_ACCEPT = 1, EV_LOAD = 2, EV_DONE = 4, EV_BUSY = 8, EV_VIOL = 16 };

__device__ __forceinline__ uint32_t fsm_step(uint32_t &c, bool stb, bool rst,
		uint32_t C)
{
	uint32_t ev = 0;
	const bool completing = (c == 1);
	if (completing)
		ev |= EV_LOAD;
	if (rst) {
		c = 0;
	} else if (c == 0) {
		if (stb) {
			c = C - 1;
			ev |= EV_ACCEPT;
		}
	} else {
		if (completing)
			ev |= EV_DONE | (stb ? EV_VIOL : 0u);
		c -= 1;
	}
	if (c != 0)
		ev |= EV_BUSY;
	return ev;
}

// 16 clocks of a byte-per-clock port array in one load when the array allows
// it (tiles start at multiples of 1024 clocks, so only the base matters)
struct Pins16 {
	uint32_t w[4];
	__device__ __forceinline__ bool bit(int k) const
	{
		return ((w[k >> 2] >> ((k & 3) * 8)) & 0xffu) != 0;
	}
};
__device__ __forceinline__ Pins16 load16(const uint8_t *p, uint32_t t,
		uint32_t hi, bool vec)
{
	Pins16 r{{0, 0, 0, 0}};
	if (!p)
		return r;
	if (vec && t + 16 <= hi) {
		const u32x4 v = *reinterpret_cast<const u32x4 *>(p + t);
		r.w[0] = v[0]; r.w[1] = v[1]; r.w[2] = v[2]; r.w[3] = v[3];
	} else {
		for (int k = 0; k < 16 && t + k < hi; k++)
			r.w[k >> 2] |= (uint32_t)(p[t + k] != 0) << ((k & 3) * 8);
	}
	return r;
}
__device__ __forceinline__ bool aligned16p(const void *p)
{
	return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// pass 1: exit state of every tile for every entry state
__global__ __launch_bounds__(128) void seq_fsm_tables(const uint8_t *stb,
		const uint8_t *rst, uint32_t T, uint32_t C, uint8_t *gtab)
{
	const uint32_t tile = blockIdx.x;
	if (threadIdx.x >= C)
		return;
	uint32_t c = threadIdx.x;
	const uint32_t lo = tile * kFsmTile;
	const uint32_t hi = (lo + kFsmTile < T) ? lo + kFsmTile : T;
	const bool vs = aligned16p(stb), vr = aligned16p(rst);
	for (uint32_t t = lo; t < hi; t += 16) {
		const Pins16 ps = load16(stb, t, hi, vs), pr = load16(rst, t, hi, vr);
#pragma unroll
		for (int k = 0; k < 16; k++)
			if (t + k < hi)
				(void)fsm_step(c, ps.bit(k), pr.bit(k), C);
	}
	gtab[(size_t)tile * kFsmStates + threadIdx.x] = (uint8_t)c;
}

// pass 2 (one wave): entry state of every tile; entry[ntiles] = exit state
__global__ __launch_bounds__(64) void seq_fsm_spine(const uint8_t *gtab,
		uint32_t ntiles, const uint32_t *c_in, uint8_t *entry,
		uint32_t *c_out)
{
	__shared__ uint8_t chunk[64 * kFsmStates];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0)
		carry = *c_in;
	__syncthreads();
	for (uint32_t base = 0; base < ntiles; base += 64) {
		const uint32_t cnt = (ntiles - base < 64) ? ntiles - base : 64;
		for (uint32_t i = threadIdx.x; i < cnt * kFsmStates; i += 64)
			chunk[i] = gtab[(size_t)base * kFsmStates + i];
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t c = carry;
			for (uint32_t k = 0; k < cnt; k++) {
				entry[base + k] = (uint8_t)c;
				c = chunk[k * kFsmStates + c];
			}
			carry = c;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		entry[ntiles] = (uint8_t)carry;
		*c_out = carry;
	}
}

// pass 3: per-clock flags from the true entry states
__device__ __forceinline__ void store16(uint8_t *p, uint32_t t, uint32_t hi,
		bool vec, const uint32_t (&w)[4])
{
	if (!p)
		return;
	if (vec && t + 16 <= hi) {
		u32x4 v = {w[0], w[1], w[2], w[3]};
		*reinterpret_cast<u32x4 *>(p + t) = v;
	} else {
		for (int k = 0; k < 16 && t + k < hi; k++)
			p[t + k] = (uint8_t)((w[k >> 2] >> ((k & 3) * 8)) & 0xffu);
	}
}

__global__ __launch_bounds__(64) void seq_fsm_emit(const uint8_t *stb,
		const uint8_t *rst, uint32_t T, uint32_t C, const uint8_t *entry,
		uint32_t ntiles, uint8_t *accept, uint8_t *load, uint8_t *busy,
		uint8_t *done, uint32_t *violations)
{
	const uint32_t tile = blockIdx.x * 64 + threadIdx.x;
	if (tile >= ntiles)
		return;
	uint32_t c = entry[tile];
	const uint32_t lo = tile * kFsmTile;
	const uint32_t hi = (lo + kFsmTile < T) ? lo + kFsmTile : T;
	const bool vs = aligned16p(stb), vr = aligned16p(rst);
	const bool vb = aligned16p(busy), vd = aligned16p(done);
	uint32_t viol = 0;
	for (uint32_t t = lo; t < hi; t += 16) {
		const Pins16 ps = load16(stb, t, hi, vs), pr = load16(rst, t, hi, vr);
		uint32_t wa[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
		uint32_t wb[4] = {0, 0, 0, 0}, wd[4] = {0, 0, 0, 0};
#pragma unroll
		for (int k = 0; k < 16; k++) {
			if (t + k < hi) {
				const uint32_t ev = fsm_step(c, ps.bit(k), pr.bit(k), C);
				const int sh = (k & 3) * 8;
				wa[k >> 2] |= ((ev & EV_ACCEPT) ? 1u : 0u) << sh;
				wl[k >> 2] |= ((ev & EV_LOAD) ? 1u : 0u) << sh;
				wb[k >> 2] |= ((ev & EV_BUSY) ? 1u : 0u) << sh;
				wd[k >> 2] |= ((ev & EV_DONE) ? 1u : 0u) << sh;
				viol += (ev & EV_VIOL) ? 1u : 0u;
			}
		}
		store16(accept, t, hi, true, wa);	// workspace: always aligned
		store16(load, t, hi, true, wl);
		store16(busy, t, hi, vb, wb);
		store16(done, t, hi, vd, wd);
	}
	if (viol)
		atomicAdd(violations, viol);	// of this block: arms seq_literal
}

struct SeqView {
	const int32_t *x, *y;
	const uint32_t *phase;		// NULL for seqpolar
	const uint8_t *aux;
	const int32_t *px, *py;		// sample in flight when the block starts
	const uint32_t *pph;
	const uint8_t *paux;
	const uint32_t *A;		// accepts in [0, t]
	const int32_t *R;		// last load clock <= t, or -1
	const uint32_t *pos;		// clock of the k-th accept
	uint32_t T;
};

// the sample whose result the output registers hold after clock t
__global__ __launch_bounds__(kBlock) void seq_gather(SeqView v, int32_t *gx,
		int32_t *gy, uint32_t *gph, uint8_t *held, uint8_t *oaux)
{
	const uint32_t stride = gridDim.x * kBlock;
	for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < v.T; t += stride) {
		const int32_t r = v.R[t];
		int32_t x = 0, y = 0;
		uint32_t ph = 0;
		uint8_t ax = 0, h = 1;
		if (r >= 0) {
			h = 0;
			const uint32_t k = v.A[r];	// accepts before the load
			if (k == 0) {
				x = *v.px; y = *v.py; ph = *v.pph; ax = *v.paux;
			} else {
				const uint32_t s = v.pos[k - 1];
				x = v.x[s]; y = v.y[s];
				ph = v.phase ? v.phase[s] : 0u;
				ax = v.aux ? (uint8_t)(v.aux[s] != 0) : (uint8_t)0;
			}
		}
		gx[t] = x; gy[t] = y;
		if (gph) gph[t] = ph;
		held[t] = h;
		oaux[t] = ax;
	}
}

// clocks before the first load of this block: the registers still hold what
// the previous call left in them
__global__ __launch_bounds__(kBlock) void seq_fix_held(uint32_t T,
		const uint8_t *held, const int32_t *l0, const int32_t *l1,
		const uint8_t *la, int32_t *o0, int32_t *o1, uint8_t *oaux)
{
	const uint32_t stride = gridDim.x * kBlock;
	for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < T; t += stride)
		if (held[t]) {
			o0[t] = *l0; o1[t] = *l1; oaux[t] = *la;
		}
}

__global__ void seq_carry(SeqView v, const int32_t *o0, const int32_t *o1,
		const uint8_t *oaux, int32_t *npx, int32_t *npy, uint32_t *npph,
		uint8_t *npaux, int32_t *nl0, int32_t *nl1, uint8_t *nla)
{
	// single thread, in place: read, then write
	const uint32_t acc = v.A[v.T - 1];
	int32_t x = *v.px, y = *v.py;
	uint32_t ph = *v.pph;
	uint8_t ax = *v.paux;
	if (acc != 0) {
		const uint32_t s = v.pos[acc - 1];
		x = v.x[s]; y = v.y[s];
		ph = v.phase ? v.phase[s] : 0u;
		ax = v.aux ? (uint8_t)(v.aux[s] != 0) : (uint8_t)0;
	}
	const int32_t a = o0[v.T - 1], b = o1[v.T - 1];
	const uint8_t c = oaux[v.T - 1];
	*npx = x; *npy = y; *npph = ph; *npaux = ax;
	*nl0 = a; *nl1 = b; *nla = c;
}


// ------------------------------------------- off protocol: register level
//
// The passes above reproduce the handshake for every input stream that keeps to
// the protocol.  One thing a bench can do that they cannot express: strobe on
// the very clock that completes a sample.  rtl/seqcordic.v:229-236 /
// rtl/seqpolar.v:215-221 give i_stb precedence over the return to idle, yet
// pre_valid (:243-246) needs idle, so NO sample is loaded: the datapath, which
// rotates on every clock (:270-291 / :259-281), goes round again over its own
// unrounded result with state counting from 0, and a second o_done appears C-1
// clocks later with a value that is a function of the core's registers, not of
// any input.  A bench that holds i_stb high sees exactly that, forever.
//
// Such stretches are reproduced by stepping the core's register file clock by
// clock -- one thread, the literal next-state functions of the RTL.  The block
// passes run as always; seq_fsm_emit counts the off-protocol strobes of the
// block, and if there is one (or the previous block ended inside a re-run) the
// whole block is re-done at register level from the state the block started
// in, every output array overwritten.  Slow (~10^7 clocks/s) and exact.
struct SeqRegs {
	int64_t	prex, prey, xv, yv;	// rtl/seqcordic.v:88-92
	uint32_t preph, ph, cangle, state;
	int32_t	idle, pre_valid, aux, o_done, o_aux, o0, o1;
};

struct SeqLit {
	SeqRegs	regs;
	uint32_t literal;	// regs are authoritative: a re-run is in flight
	uint32_t viol_block;	// off-protocol strobes the block passes saw
	// fast-path state at the start of the block (the passes update it in place)
	uint32_t snap_c;
	int32_t	snap_px, snap_py;
	uint32_t snap_pph, snap_paux;
	int32_t	snap_l0, snap_l1;
	uint32_t snap_la;
	uint32_t table[128];	// cordic_angle[], padded as the emitter pads it
};

struct SeqModel {
	int32_t	rot, ww, pw, iw, ow, in_shl;
	uint32_t pm, smask, tmask, last, C;
};

__device__ __forceinline__ int64_t seq_wrap(int64_t v, int ww)
{
	return sext64(v, ww);
}
__device__ __forceinline__ int64_t seq_asr(int64_t v, uint32_t sh)
{
	return v >> (sh > 63u ? 63u : sh);
}

// rtl/seqcordic.v:124-182 / rtl/seqpolar.v:100-120: the pre-rotation registers
__device__ void seq_prerotate(const SeqModel &m, int32_t ix, int32_t iy, uint32_t iph,
		int64_t &px, int64_t &py, uint32_t &pp)
{
	const int64_t sx = sext32(ix, m.iw), sy = sext32(iy, m.iw);
	const int64_t ex = seq_wrap((int64_t)((uint64_t)sx << m.in_shl), m.ww);
	const int64_t ey = seq_wrap((int64_t)((uint64_t)sy << m.in_shl), m.ww);
	if (m.rot) {
		const uint32_t ph = iph & m.pm, q = 1u << (m.pw - 2);
		switch ((ph >> (m.pw - 3)) & 7u) {
		case 0: case 7: px = ex; py = ey; pp = ph; break;
		case 1: case 2: px = seq_wrap(-ey, m.ww); py = ex; pp = (ph - q) & m.pm; break;
		case 3: case 4: px = seq_wrap(-ex, m.ww); py = seq_wrap(-ey, m.ww);
				pp = (ph - 2 * q) & m.pm; break;
		default:	px = ey; py = seq_wrap(-ex, m.ww); pp = (ph - 3 * q) & m.pm; break;
		}
	} else {
		const uint32_t e = 1u << (m.pw - 3);
		switch ((sx < 0 ? 2 : 0) | (sy < 0 ? 1 : 0)) {
		case 1:  px = seq_wrap(ex - ey, m.ww); py = seq_wrap(ex + ey, m.ww); pp = 7 * e; break;
		case 2:  px = seq_wrap(-ex + ey, m.ww); py = seq_wrap(-ex - ey, m.ww); pp = 3 * e; break;
		case 3:  px = seq_wrap(-ex - ey, m.ww); py = seq_wrap(ex - ey, m.ww); pp = 5 * e; break;
		default: px = seq_wrap(ex + ey, m.ww); py = seq_wrap(-ex + ey, m.ww); pp = e; break;
		}
		pp &= m.pm;
	}
}

// rtl/seqcordic.v:298-303 (convergent rounding) or plain truncation
__device__ int32_t seq_round(const SeqModel &m, int64_t v)
{
	const int r = m.ww - m.ow;
	if (m.ww > m.ow + 1) {
		const int64_t b = (v >> r) & 1;
		const int64_t add = (b << (r - 1)) | (b ? 0 : (((int64_t)1 << (r - 1)) - 1));
		v = seq_wrap(v + add, m.ww);
	}
	return (int32_t)sext64(v >> r, m.ow);
}

// one rising clock edge: every register from its pre-edge value
__device__ void seq_clock(const SeqModel &m, const uint32_t *table, SeqRegs &r,
		bool stb, bool rst, bool aux, int32_t ix, int32_t iy, uint32_t iph)
{
	const bool at_last = r.state >= m.last;			// :318 / :208
	const bool leave = m.rot ? (r.state == m.last) : at_last;	// :235,257 / :218,236
	SeqRegs n = r;
	seq_prerotate(m, ix, iy, iph, n.prex, n.prey, n.preph);
	if (rst) n.aux = 0;						// :100-105
	else if (stb && r.idle) n.aux = aux;
	if (rst) n.idle = 1;						// :226-236
	else if (stb) n.idle = 0;
	else if (leave) n.idle = 1;
	n.pre_valid = rst ? 0 : (stb && r.idle);			// :240-246
	if (rst || r.idle || leave) n.state = 0;			// :252-262
	else n.state = (r.state + 1) & m.smask;
	n.cangle = table[r.state & m.tmask];				// :248-250
	if (r.pre_valid) {						// :270-291
		n.xv = r.prex; n.yv = r.prey; n.ph = r.preph;
	} else {
		const int64_t dy = seq_asr(r.yv, r.state), dx = seq_asr(r.xv, r.state);
		const bool minus = m.rot ? ((r.ph >> (m.pw - 1)) & 1u) != 0 : r.yv < 0;
		if ((m.rot != 0) == minus) {
			// p2r with negative phase, or r2p above the axis
			n.xv = seq_wrap(r.xv + dy, m.ww);
			n.yv = seq_wrap(r.yv - dx, m.ww);
			n.ph = (r.ph + r.cangle) & m.pm;
		} else {
			n.xv = seq_wrap(r.xv - dy, m.ww);
			n.yv = seq_wrap(r.yv + dx, m.ww);
			n.ph = (r.ph - r.cangle) & m.pm;
		}
	}
	n.o_done = rst ? 0 : (at_last ? 1 : 0);			// :308-314
	if (at_last) {							// :316-324
		n.o0 = seq_round(m, r.xv);
		n.o1 = m.rot ? seq_round(m, r.yv) : (int32_t)r.ph;
		n.o_aux = r.aux;
	}
	r = n;
}

__global__ void seq_snapshot(SeqLit *lit, const uint32_t *c, const int32_t *px,
		const int32_t *py, const uint32_t *pph, const uint8_t *paux,
		const int32_t *l0, const int32_t *l1, const uint8_t *la)
{
	lit->viol_block = 0;
	lit->snap_c = *c;
	lit->snap_px = *px; lit->snap_py = *py; lit->snap_pph = *pph;
	lit->snap_paux = *paux;
	lit->snap_l0 = *l0; lit->snap_l1 = *l1; lit->snap_la = *la;
}

__global__ void seq_literal(SeqModel m, SeqLit *lit, uint32_t T, const uint8_t *stb,
		const uint8_t *rst, const uint8_t *aux, const int32_t *x,
		const int32_t *y, const uint32_t *phase, int32_t *o0, int32_t *o1,
		uint8_t *oaux, uint8_t *busy, uint8_t *done, uint32_t *c_out,
		int32_t *px, int32_t *py, uint32_t *pph, uint8_t *paux, int32_t *l0,
		int32_t *l1, uint8_t *la, unsigned long long *violations)
{
	if (!lit->literal && lit->viol_block == 0)
		return;				// the block kept to the protocol
	SeqRegs r;
	uint32_t c;				// the block passes' view, kept alongside
	int32_t sx, sy;				// sample of the run in flight
	uint32_t sph, sax;
	bool rerun;				// that run was started by a re-run
	if (lit->literal) {
		r = lit->regs;
		c = lit->snap_c;
		sx = lit->snap_px; sy = lit->snap_py; sph = lit->snap_pph;
		sax = lit->snap_paux;
		rerun = true;
	} else {
		// registers of the state the block passes carry: idle, or a sample
		// accepted C-1-c clocks ago
		r = SeqRegs{};
		r.idle = 1;
		r.cangle = lit->table[0];
		r.o0 = lit->snap_l0; r.o1 = lit->snap_l1; r.o_aux = (int32_t)lit->snap_la;
		c = lit->snap_c;
		sx = lit->snap_px; sy = lit->snap_py; sph = lit->snap_pph;
		sax = lit->snap_paux;
		rerun = false;
		if (c != 0) {
			seq_clock(m, lit->table, r, true, false, sax != 0, sx, sy, sph);
			for (uint32_t k = 0; k < m.C - 1 - c; k++)
				seq_clock(m, lit->table, r, false, false, false, sx, sy, sph);
		}
	}
	unsigned long long viol = 0;
	for (uint32_t t = 0; t < T; t++) {
		const bool s_ = stb[t] != 0, rs = rst && rst[t] != 0;
		const bool ax = aux && aux[t] != 0;
		const uint32_t ph = phase ? phase[t] : 0u;
		const bool completing = !r.idle && r.state >= m.last;
		// the handshake as the block passes count it (fsm_step), with the
		// re-run: a strobe on the completing clock restarts the count
		if (rs) {
			c = 0; rerun = false;
		} else if (c == 0) {
			if (s_) {
				c = m.C - 1; rerun = false;
				sx = x[t]; sy = y[t]; sph = ph; sax = ax;
			}
		} else if (completing && s_) {
			c = m.C - 1; rerun = true; viol++;
		} else {
			c -= 1;
			if (c == 0) rerun = false;
		}
		seq_clock(m, lit->table, r, s_, rs, ax, x[t], y[t], ph);
		o0[t] = r.o0;
		o1[t] = r.o1;
		if (oaux) oaux[t] = (uint8_t)r.o_aux;
		if (busy) busy[t] = (uint8_t)(r.idle ? 0 : 1);
		if (done) done[t] = (uint8_t)r.o_done;
	}
	lit->regs = r;
	lit->literal = (rerun && c != 0) ? 1u : 0u;
	*c_out = c;
	*px = sx; *py = sy; *pph = sph; *paux = (uint8_t)sax;
	*l0 = r.o0; *l1 = r.o1; *la = (uint8_t)r.o_aux;
	*violations += viol;
}

} // namespace

size_t seq_literal_bytes() { return sizeof(SeqLit); }

void seq_literal_init(const cordic_config &cfg, void *host_image)
{
	SeqLit *l = new (host_image) SeqLit{};
	l->regs.idle = 1;
	// sw/cordiclib.cpp:145-149: the sequential cores' table has
	// 2^nextlg(NSTAGES) entries, all of them computed by the formula
	const unsigned tlen = 1u << next_lg((unsigned)cfg.nstages);
	for (unsigned k = 0; k < tlen && k < 128; k++)
		l->table[k] = arctan_entry(k, cfg.pw);
	l->regs.cangle = l->table[0];
}

size_t seq_workspace_bytes(size_t T)
{
	const size_t ntiles = (T + kFsmTile - 1) / kFsmTile;
	const size_t stiles = (T + kScanTile - 1) / kScanTile;
	// accept, load, held, oaux (1 byte each), A, R, pos, gx, gy, gph (4 each),
	// FSM tables and entry states, scan tile arrays
	return T * 28 + ntiles * (kFsmStates + 1) + stiles * 8 + 512;
}

int launch_seq_ticks(const cordic_config &cfg, SeqState &s, size_t T,
		const uint8_t *stb, const uint8_t *reset, const uint8_t *aux,
		const int32_t *x, const int32_t *y, const uint32_t *phase,
		int32_t *o0, int32_t *o1, uint8_t *busy, uint8_t *done,
		uint8_t *oaux, void *stream)
{
	(void)hipGetLastError();
	if (T == 0)
		return CORDIC_OK;
	if (T >= (1ull << 31))
		return CORDIC_ERR_ARGS;
	const bool rot = (cfg.mode == CORDIC_SP2R);
	if (!rot && cfg.mode != CORDIC_SR2P)
		return CORDIC_ERR_MODE;
	if (!stb || !x || !y || !o0 || !o1 || (rot && !phase))
		return CORDIC_ERR_ARGS;
	const uint32_t C = (uint32_t)cfg.clocks_per_output;
	if (C < 2 || C > (uint32_t)kFsmStates)
		return CORDIC_ERR_UNSUPPORTED;
	if (seq_workspace_bytes(T) > s.ws_bytes)
		return CORDIC_ERR_ARGS;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const uint32_t n = (uint32_t)T;
	const uint32_t ntiles = (n + kFsmTile - 1) / kFsmTile;
	const uint32_t stiles = (n + kScanTile - 1) / kScanTile;

	char *w = static_cast<char *>(s.ws);
	auto take = [&](size_t bytes) {
		char *p = w;
		w += (bytes + 15) & ~(size_t)15;
		return p;
	};
	int32_t *gx = reinterpret_cast<int32_t *>(take((size_t)n * 4));
	int32_t *gy = reinterpret_cast<int32_t *>(take((size_t)n * 4));
	uint32_t *gph = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
	uint32_t *A = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
	int32_t *R = reinterpret_cast<int32_t *>(take((size_t)n * 4));
	uint32_t *pos = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
	uint8_t *accept = reinterpret_cast<uint8_t *>(take(n));
	uint8_t *load = reinterpret_cast<uint8_t *>(take(n));
	uint8_t *held = reinterpret_cast<uint8_t *>(take(n));
	uint8_t *aux_ws = reinterpret_cast<uint8_t *>(take(n));
	uint8_t *gtab = reinterpret_cast<uint8_t *>(take((size_t)ntiles * kFsmStates));
	uint8_t *entry = reinterpret_cast<uint8_t *>(take((size_t)ntiles + 1));
	uint32_t *tile_adv = reinterpret_cast<uint32_t *>(take((size_t)stiles * 4));
	int32_t *tile_last = reinterpret_cast<int32_t *>(take((size_t)stiles * 4));
	uint8_t *oa = oaux ? oaux : aux_ws;

	SeqLit *lit = static_cast<SeqLit *>(s.lit);
	hipLaunchKernelGGL(seq_snapshot, dim3(1), dim3(1), 0, st, lit, s.c, s.px,
			s.py, s.pph, s.paux, s.l0, s.l1, s.la);
	hipLaunchKernelGGL(seq_fsm_tables, dim3(ntiles), dim3(128), 0, st, stb,
			reset, n, C, gtab);
	hipLaunchKernelGGL(seq_fsm_spine, dim3(1), dim3(64), 0, st, gtab, ntiles,
			s.c, entry, s.c);			// in place
	hipLaunchKernelGGL(seq_fsm_emit, dim3((ntiles + 63) / 64), dim3(64), 0, st,
			stb, reset, n, C, entry, ntiles, accept, load, busy, done,
			&lit->viol_block);
	TickFlags f{accept, load};	// "advancing" = accept, "reset" = load
	hipLaunchKernelGGL(stream_tile_totals, dim3(stiles), dim3(kScanThreads), 0,
			st, f, n, tile_adv, tile_last);
	hipLaunchKernelGGL(stream_tile_spine, dim3(1), dim3(kScanThreads), 0, st,
			tile_adv, tile_last, stiles);
	hipLaunchKernelGGL(stream_tile_apply, dim3(stiles), dim3(kScanThreads), 0,
			st, f, n, tile_adv, tile_last, A, R, pos);
	SeqView v{x, y, rot ? phase : nullptr, aux, s.px, s.py, s.pph, s.paux, A,
			R, pos, n};
	hipLaunchKernelGGL(seq_gather, dim3(grid_1d(n)), dim3(kBlock), 0, st, v, gx,
			gy, rot ? gph : nullptr, held, oa);
	if (hipGetLastError() != hipSuccess)
		return CORDIC_ERR_DEVICE;
	int rc;
	if (rot) {
		RotatorJob j;
		j.x = gx; j.y = gy; j.phase = gph;
		j.ox = o0; j.oy = o1; j.n = n;
		rc = launch_rotator(cfg, Feed::PhaseArray_XYArray, j, stream);
	} else {
		rc = launch_topolar(cfg, n, gx, gy, o0, reinterpret_cast<uint32_t *>(o1),
				stream);
	}
	if (rc != CORDIC_OK)
		return rc;
	hipLaunchKernelGGL(seq_fix_held, dim3(grid_1d(n)), dim3(kBlock), 0, st, n,
			held, s.l0, s.l1, s.la, o0, o1, oa);
	hipLaunchKernelGGL(seq_carry, dim3(1), dim3(1), 0, st, v, o0, o1, oa, s.px,
			s.py, s.pph, s.paux, s.l0, s.l1, s.la);	// in place
	// off-protocol strobes in this block (or a re-run still in flight from the
	// last one): redo the block at register level, else return at once
	SeqModel m{};
	m.rot = rot ? 1 : 0;
	m.ww = cfg.ww; m.pw = cfg.pw; m.iw = cfg.iw; m.ow = cfg.ow;
	m.in_shl = rot ? cfg.ww - cfg.iw - 1 : cfg.ww - cfg.iw - 2;
	m.pm = (cfg.pw >= 32) ? 0xffffffffu : ((1u << cfg.pw) - 1u);
	m.smask = (1u << next_lg((unsigned)(rot ? cfg.nstages : cfg.nstages + 1))) - 1u;
	m.tmask = (1u << next_lg((unsigned)cfg.nstages)) - 1u;
	m.last = (uint32_t)(rot ? cfg.nstages - 1 : cfg.nstages + 1);
	m.C = C;
	hipLaunchKernelGGL(seq_literal, dim3(1), dim3(1), 0, st, m, lit, n, stb, reset,
			aux, x, y, rot ? phase : nullptr, o0, o1, oaux, busy, done, s.c,
			s.px, s.py, s.pph, s.paux, s.l0, s.l1, s.la, s.violations);
	return (hipGetLastError() == hipSuccess) ? CORDIC_OK : CORDIC_ERR_DEVICE;
}

} // namespace cordic_amd
