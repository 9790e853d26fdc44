// cordic_quality.hip -- the reference benches' error statistics, reduced on
// the device.
//
// bench/cpp/cordic_tb.cpp:223-337 and bench/cpp/topolar_tb.cpp:222-315 judge a
// core by comparing every output with a double-precision sin/cos (atan2) of the
// same input and reducing to a handful of sums and maxima.  The Verilated bench
// does that on the host over at most 2^PW ints; at PW = 32 its own sample count
// (`const int NSAMPLES = 1ul << PW`) is 0.  Here the comparison runs where the
// samples are: one fp64 sincospi / atan2 per sample, per-thread sums, a block
// reduction, and ONE slot of partial sums per block that the block itself
// accumulates into call after call (no atomics, the grid is fixed per handle,
// so the result is reproducible bit for bit); the host adds the slots up in
// long double.  2^32 samples cost about a second.
//
// Nothing here is on the product's data path: these kernels read what the
// engine wrote.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "cordic_amd.h"
#include "cordic_internal.h"

namespace cordic_amd {
namespace {

constexpr int kQBlock = 256;
constexpr int kSlotDoubles = 8;		// sums [0..5], maxima [6..7]

struct QSlot {
	double s[kSlotDoubles];
	unsigned long long arg[2];	// sample index of maxima 6 and 7
};

__device__ __forceinline__ int32_t q_sext(int32_t v, int w)
{
	return (w >= 32) ? v : (int32_t)((uint32_t)v << (32 - w)) >> (32 - w);
}

struct QParams {
	int iw, ow, pw;
	double gain;		// GAIN of the generated header
	double out_scale;	// p2r: 2^-(IW+1-OW)   r2p: 2^(IW-1-OW)
	double inv_2pw;		// 2^-PW
	double maxphase;	// 2^PW
	double rad_to_phase;	// 2^PW / 2 pi
	uint32_t pmask;		// 2^PW - 1
};

// Sums and maxima of one block -> its slot.  `v[0..5]` are added, `m0`, `m1`
// are maxima with the sample index they were seen at.
__device__ void block_commit(QSlot *slot, double v[6], double m0,
		unsigned long long a0, double m1, unsigned long long a1)
{
	__shared__ double red[kQBlock / 64][kSlotDoubles];
	__shared__ unsigned long long reda[kQBlock / 64][2];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int off = 32; off; off >>= 1) {
		for (int k = 0; k < 6; k++)
			v[k] += __shfl_down(v[k], off, 64);
		const double o0 = __shfl_down(m0, off, 64);
		const unsigned long long b0 = __shfl_down(a0, off, 64);
		if (o0 > m0 || (o0 == m0 && b0 < a0)) { m0 = o0; a0 = b0; }
		const double o1 = __shfl_down(m1, off, 64);
		const unsigned long long b1 = __shfl_down(a1, off, 64);
		if (o1 > m1 || (o1 == m1 && b1 < a1)) { m1 = o1; a1 = b1; }
	}
	if (lane == 0) {
		for (int k = 0; k < 6; k++)
			red[wave][k] = v[k];
		red[wave][6] = m0; red[wave][7] = m1;
		reda[wave][0] = a0; reda[wave][1] = a1;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		QSlot s = *slot;
		for (int w = 0; w < kQBlock / 64; w++) {
			for (int k = 0; k < 6; k++)
				s.s[k] += red[w][k];
			if (red[w][6] > s.s[6] || (red[w][6] == s.s[6] && reda[w][0] < s.arg[0])) {
				s.s[6] = red[w][6]; s.arg[0] = reda[w][0];
			}
			if (red[w][7] > s.s[7] || (red[w][7] == s.s[7] && reda[w][1] < s.arg[1])) {
				s.s[7] = red[w][7]; s.arg[1] = reda[w][1];
			}
		}
		*slot = s;
	}
}

// cordic_tb.cpp:223-279.  Slot sums: 0 sum err^2, 1 sum d.o (sumxy), 2 sum
// |o|^2 (sumsq, mag), 3 sum |d|^2 (sumd), 4 sum |i|^2 (imag), 5 unused;
// maximum 6 = max err^2.
template <bool NCO>
__global__ __launch_bounds__(kQBlock) void quality_p2r(QParams qp, size_t n,
		const int32_t *__restrict__ x, const int32_t *__restrict__ y,
		int32_t x0, int32_t y0, const uint32_t *__restrict__ phase,
		uint32_t phase0, uint32_t fcw, unsigned long long index0,
		const int32_t *__restrict__ ox, const int32_t *__restrict__ oy,
		unsigned long long base, QSlot *slots)
{
	double v[6] = {0, 0, 0, 0, 0, 0};
	double mx = -1.0;
	unsigned long long amx = ~0ull;
	const size_t stride = (size_t)gridDim.x * kQBlock;
	for (size_t i = (size_t)blockIdx.x * kQBlock + threadIdx.x; i < n; i += stride) {
		uint32_t p;
		if (NCO)
			p = phase0 + (uint32_t)(index0 + i) * fcw;
		else
			p = phase[i];
		p &= qp.pmask;
		const double ix = (double)q_sext(x ? x[i] : x0, qp.iw);
		const double iy = (double)q_sext(y ? y[i] : y0, qp.iw);
		// ph = pdata * 2 pi / 2^PW (:232-233); sincospi reduces exactly
		double sn, cs;
		sincospi(2.0 * ((double)p * qp.inv_2pw), &sn, &cs);
		const double k = qp.gain * qp.out_scale;	// :237-248
		const double dx = (cs * ix - sn * iy) * k;
		const double dy = (sn * ix + cs * iy) * k;
		const double rx = (double)ox[i], ry = (double)oy[i];
		const double e2 = (dx - rx) * (dx - rx) + (dy - ry) * (dy - ry);
		v[0] += e2;
		v[1] += dx * rx + dy * ry;
		v[2] += rx * rx + ry * ry;
		v[3] += dx * dx + dy * dy;
		v[4] += ix * ix + iy * iy;
		if (e2 > mx) { mx = e2; amx = base + i; }
	}
	block_commit(slots + blockIdx.x, v, mx, amx, -1.0, ~0ull);
}

// topolar_tb.cpp:222-256.  Slot sums: 0 sum dperr^2, 1 sum mgerr^2, 2 sum
// dperr (bias); maxima 6 = max |dperr|, 7 = max mgerr.
__global__ __launch_bounds__(kQBlock) void quality_r2p(QParams qp, size_t n,
		const int32_t *__restrict__ x, const int32_t *__restrict__ y,
		int32_t imag, const int32_t *__restrict__ omag,
		const uint32_t *__restrict__ ophase, unsigned long long base,
		QSlot *slots)
{
	double v[6] = {0, 0, 0, 0, 0, 0};
	double mp = -1.0, mv = -1.0;
	unsigned long long ap = ~0ull, av = ~0ull;
	const size_t stride = (size_t)gridDim.x * kQBlock;
	for (size_t i = (size_t)blockIdx.x * kQBlock + threadIdx.x; i < n; i += stride) {
		const double ix = (double)q_sext(x[i], qp.iw);
		const double iy = (double)q_sext(y[i], qp.iw);
		double ep = atan2(iy, ix) * qp.rad_to_phase;	// :226-228
		if (ep < 0.0)
			ep += qp.maxphase;
		// the bench sign extends o_phase from PW bits (:177-181)
		const uint32_t opu = ophase[i] & qp.pmask;
		double op = (double)opu;
		if (op >= 0.5 * qp.maxphase)
			op -= qp.maxphase;
		double dp = op - ep;				// :229-233
		while (dp > 0.5 * qp.maxphase) dp -= qp.maxphase;
		while (dp < -0.5 * qp.maxphase) dp += qp.maxphase;
		// :238-246: the circle's nominal radius, or the vector's own length
		const double im = (imag >= 0) ? (double)imag : sqrt(ix * ix + iy * iy);
		const double mg = fabs((double)q_sext(omag[i], qp.ow)
				- im * qp.out_scale * qp.gain);
		v[0] += dp * dp;
		v[1] += mg * mg;
		v[2] += dp;
		if (fabs(dp) > mp) { mp = fabs(dp); ap = base + i; }
		if (mg > mv) { mv = mg; av = base + i; }
	}
	block_commit(slots + blockIdx.x, v, mp, ap, mv, av);
}

// topolar_tb.cpp:127-141: sample i of NSAMPLES = 2^lgn points on TWO turns of
// a circle of radius 2^(IW-1)-1, components truncated toward zero by (int).
__global__ __launch_bounds__(kQBlock) void fill_circle(int32_t *x, int32_t *y,
		size_t n, unsigned long long index0, int sh, int pw, double mg)
{
	const size_t stride = (size_t)gridDim.x * kQBlock;
	const double inv = 1.0 / (double)(1ull << (pw - 1));
	for (size_t i = (size_t)blockIdx.x * kQBlock + threadIdx.x; i < n; i += stride) {
		const long long lv = (long long)((index0 + i) << sh);
		const int ip = (int)lv;				// ipdata[i] = (int)lv
		double sn, cs;
		sincospi((double)ip * inv, &sn, &cs);	// ph = ip * pi / 2^(PW-1)
		x[i] = (int)(mg * cs);
		y[i] = (int)(mg * sn);
	}
}

} // namespace
} // namespace cordic_amd

using namespace cordic_amd;

struct cordic_quality {
	cordic_config cfg;
	QParams qp;
	QSlot *d_slots = nullptr;
	int grid = 0;
	int device = 0;
	unsigned long long count = 0;	// samples accumulated so far
	int kind = -1;			// 0 = p2r sums, 1 = r2p sums
};

static QParams make_qparams(const cordic_config &c)
{
	QParams q{};
	const bool rot = (c.mode == CORDIC_P2R || c.mode == CORDIC_SP2R);
	q.iw = c.iw; q.ow = c.ow; q.pw = c.pw;
	q.gain = c.gain;
	q.out_scale = rot ? std::ldexp(1.0, -(c.iw + 1 - c.ow))
			  : std::ldexp(1.0, c.iw - 1 - c.ow);
	q.inv_2pw = std::ldexp(1.0, -c.pw);
	q.maxphase = std::ldexp(1.0, c.pw);
	q.rad_to_phase = q.maxphase / M_PI / 2.0;
	q.pmask = (c.pw >= 32) ? 0xffffffffu : ((1u << c.pw) - 1u);
	return q;
}

static int zero_slots(cordic_quality *q, hipStream_t st)
{
	std::vector<QSlot> z((size_t)q->grid);
	for (auto &s : z) {
		for (int k = 0; k < kSlotDoubles; k++)
			s.s[k] = (k >= 6) ? -1.0 : 0.0;
		s.arg[0] = s.arg[1] = ~0ull;
	}
	// pageable source: hipMemcpyAsync returns once it has been staged
	if (hipMemcpyAsync(q->d_slots, z.data(), z.size() * sizeof(QSlot),
			hipMemcpyHostToDevice, st) != hipSuccess)
		return CORDIC_ERR_DEVICE;
	if (hipStreamSynchronize(st) != hipSuccess)
		return CORDIC_ERR_DEVICE;
	q->count = 0;
	q->kind = -1;
	return CORDIC_OK;
}

int cordic_quality_create(const cordic_config *cfg, cordic_quality **out)
{
	if (!cfg || !out)
		return CORDIC_ERR_ARGS;
	if (cfg->mode < CORDIC_P2R || cfg->mode > CORDIC_SR2P)
		return CORDIC_ERR_MODE;
	if (!config_sane(*cfg))
		return CORDIC_ERR_ARGS;
	cordic_quality *q = new (std::nothrow) cordic_quality;
	if (!q)
		return CORDIC_ERR_NOMEM;
	q->cfg = *cfg;
	q->qp = make_qparams(*cfg);
	hipDeviceProp_t prop;
	if (hipGetDevice(&q->device) != hipSuccess ||
	    hipGetDeviceProperties(&prop, q->device) != hipSuccess) {
		delete q;
		return CORDIC_ERR_DEVICE;
	}
	q->grid = prop.multiProcessorCount * 8;
	if (hipMalloc((void **)&q->d_slots, (size_t)q->grid * sizeof(QSlot)) != hipSuccess) {
		delete q;
		return CORDIC_ERR_DEVICE;
	}
	if (int rc = zero_slots(q, nullptr)) {
		(void)hipFree(q->d_slots);
		delete q;
		return rc;
	}
	*out = q;
	return CORDIC_OK;
}

void cordic_quality_destroy(cordic_quality *q)
{
	if (!q)
		return;
	if (q->d_slots)
		(void)hipFree(q->d_slots);
	delete q;
}

int cordic_quality_reset(cordic_quality *q, void *stream)
{
	if (!q)
		return CORDIC_ERR_ARGS;
	return zero_slots(q, static_cast<hipStream_t>(stream));
}

static int p2r_common(cordic_quality *q, size_t n, bool nco, const int32_t *x,
		const int32_t *y, int32_t x0, int32_t y0, const uint32_t *phase,
		uint32_t phase0, uint32_t fcw, uint64_t index0, const int32_t *ox,
		const int32_t *oy, void *stream)
{
	if (!q || !ox || !oy || (!nco && !phase) || ((x == nullptr) != (y == nullptr)))
		return CORDIC_ERR_ARGS;
	if (q->cfg.mode != CORDIC_P2R && q->cfg.mode != CORDIC_SP2R)
		return CORDIC_ERR_MODE;
	if (q->kind == 1)
		return CORDIC_ERR_ARGS;		// r2p sums in the slots: reset first
	if (n == 0)
		return CORDIC_OK;
	(void)hipGetLastError();
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (nco)
		hipLaunchKernelGGL(quality_p2r<true>, dim3(q->grid), dim3(kQBlock), 0, st,
			q->qp, n, x, y, x0, y0, phase, phase0, fcw,
			(unsigned long long)index0, ox, oy, q->count, q->d_slots);
	else
		hipLaunchKernelGGL(quality_p2r<false>, dim3(q->grid), dim3(kQBlock), 0, st,
			q->qp, n, x, y, x0, y0, phase, 0u, 0u, 0ull, ox, oy,
			q->count, q->d_slots);
	if (hipGetLastError() != hipSuccess)
		return CORDIC_ERR_DEVICE;
	q->count += n;
	q->kind = 0;
	return CORDIC_OK;
}

int cordic_quality_p2r(cordic_quality *q, size_t n, const int32_t *d_xval,
		const int32_t *d_yval, int32_t xval, int32_t yval,
		const uint32_t *d_phase, const int32_t *d_oxval,
		const int32_t *d_oyval, void *stream)
{
	return p2r_common(q, n, false, d_xval, d_yval, xval, yval, d_phase, 0, 0, 0,
			d_oxval, d_oyval, stream);
}

int cordic_quality_nco(cordic_quality *q, size_t n, uint32_t phase0, uint32_t fcw,
		uint64_t index0, int32_t xval, int32_t yval,
		const int32_t *d_oxval, const int32_t *d_oyval, void *stream)
{
	return p2r_common(q, n, true, nullptr, nullptr, xval, yval, nullptr, phase0,
			fcw, index0, d_oxval, d_oyval, stream);
}

int cordic_quality_r2p(cordic_quality *q, size_t n, const int32_t *d_xval,
		const int32_t *d_yval, int32_t imag, const int32_t *d_omag,
		const uint32_t *d_ophase, void *stream)
{
	if (!q || !d_xval || !d_yval || !d_omag || !d_ophase)
		return CORDIC_ERR_ARGS;
	if (q->cfg.mode != CORDIC_R2P && q->cfg.mode != CORDIC_SR2P)
		return CORDIC_ERR_MODE;
	if (q->kind == 0)
		return CORDIC_ERR_ARGS;
	if (n == 0)
		return CORDIC_OK;
	(void)hipGetLastError();
	hipLaunchKernelGGL(quality_r2p, dim3(q->grid), dim3(kQBlock), 0,
		static_cast<hipStream_t>(stream), q->qp, n, d_xval, d_yval, imag,
		d_omag, d_ophase, q->count, q->d_slots);
	if (hipGetLastError() != hipSuccess)
		return CORDIC_ERR_DEVICE;
	q->count += n;
	q->kind = 1;
	return CORDIC_OK;
}

// device sync + slot sums in long double
static int collect(cordic_quality *q, long double sums[6], double mx[2],
		unsigned long long arg[2])
{
	std::vector<QSlot> h((size_t)q->grid);
	if (hipDeviceSynchronize() != hipSuccess ||
	    hipMemcpy(h.data(), q->d_slots, h.size() * sizeof(QSlot),
			hipMemcpyDeviceToHost) != hipSuccess)
		return CORDIC_ERR_DEVICE;
	for (int k = 0; k < 6; k++) sums[k] = 0.0L;
	mx[0] = mx[1] = -1.0;
	arg[0] = arg[1] = ~0ull;
	for (const QSlot &s : h) {
		for (int k = 0; k < 6; k++)
			sums[k] += (long double)s.s[k];
		for (int m = 0; m < 2; m++)
			if (s.s[6 + m] > mx[m] || (s.s[6 + m] == mx[m] && s.arg[m] < arg[m])) {
				mx[m] = s.s[6 + m];
				arg[m] = s.arg[m];
			}
	}
	return CORDIC_OK;
}

int cordic_quality_p2r_result(cordic_quality *q, cordic_p2r_quality *r)
{
	if (!q || !r)
		return CORDIC_ERR_ARGS;
	if (q->kind != 0 || q->count == 0)
		return CORDIC_ERR_ARGS;
	long double s[6]; double mx[2]; unsigned long long arg[2];
	if (int rc = collect(q, s, mx, arg))
		return rc;
	std::memset(r, 0, sizeof *r);
	const double N = (double)q->count;
	r->n = q->count;
	r->sum_err2 = (double)s[0]; r->sum_xy = (double)s[1];
	r->sum_sq = (double)s[2]; r->sum_d = (double)s[3]; r->sum_in2 = (double)s[4];
	// cordic_tb.cpp:285-337
	const double scale = std::sqrt((double)s[4] / N);	// :112-116 for constant inputs
	const double G = q->cfg.gain;
	const double expected = q->cfg.quantization_variance
		+ q->cfg.phase_variance_rad * scale * scale * G * G;
	r->expected_err = std::sqrt(expected);
	r->avg_err = std::sqrt((double)s[0] / N);
	r->max_err = std::sqrt(mx[0]);
	r->max_err_index = arg[0];
	r->mag = std::sqrt((double)s[2] / N);
	r->input_mag = scale;
	r->alpha = (double)(s[1] / s[2]);
	r->cnr_db = 10.0 * std::log(scale * G * scale * G / (r->avg_err * r->avg_err))
			/ std::log(10.0);
	r->avg_limit = 1.5 * r->expected_err;
	r->max_limit = 5.2 * r->expected_err;
	r->pass_avg = !(r->avg_err > r->avg_limit);
	r->pass_max = !(r->max_err > r->max_limit);
	r->pass_alpha = !(std::fabs(r->alpha - 1.0) > 0.01);
	r->pass = r->pass_avg && r->pass_max && r->pass_alpha && s[2] > 0 && s[4] > 0;
	return CORDIC_OK;
}

int cordic_quality_r2p_result(cordic_quality *q, cordic_r2p_quality *r)
{
	if (!q || !r)
		return CORDIC_ERR_ARGS;
	if (q->kind != 1 || q->count == 0)
		return CORDIC_ERR_ARGS;
	long double s[6]; double mx[2]; unsigned long long arg[2];
	if (int rc = collect(q, s, mx, arg))
		return rc;
	std::memset(r, 0, sizeof *r);
	const double N = (double)q->count;
	r->n = q->count;
	r->max_phase_err = mx[0];
	r->max_phase_err_index = arg[0];
	r->max_mag_err = mx[1];
	r->max_mag_err_index = arg[1];
	r->avg_phase_err = std::sqrt((double)s[0] / N);
	r->avg_mag_err = std::sqrt((double)s[1] / N);
	r->mean_phase_err = (double)s[2] / N;
	// topolar_tb.cpp:303-315
	double e = std::sqrt(q->cfg.phase_variance_rad * q->qp.rad_to_phase
			* q->qp.rad_to_phase);
	r->expected_avg_phase_err = e;
	if (e < 1.0)
		e = 1.0;
	r->phase_limit = 3.4 * e;
	r->mag_limit = 2.0 * std::sqrt(q->cfg.quantization_variance);
	r->pass_phase = !(r->max_phase_err > r->phase_limit);
	r->pass_mag = !(r->max_mag_err > r->mag_limit);
	r->pass = r->pass_phase && r->pass_mag;
	return CORDIC_OK;
}

int cordic_fill_circle(int32_t *d_x, int32_t *d_y, size_t n, uint64_t index0,
		int lgnsamples, int iw, int pw, void *stream)
{
	if (!d_x || !d_y)
		return CORDIC_ERR_ARGS;
	if (iw < 1 || iw > 32 || pw < 3 || pw > 32 || lgnsamples < 1 ||
	    lgnsamples > pw + 1)
		return CORDIC_ERR_ARGS;
	if (n == 0)
		return CORDIC_OK;
	int dev; hipDeviceProp_t prop;
	if (hipGetDevice(&dev) != hipSuccess ||
	    hipGetDeviceProperties(&prop, dev) != hipSuccess)
		return CORDIC_ERR_DEVICE;
	(void)hipGetLastError();
	const size_t want = (n + kQBlock - 1) / kQBlock;
	const size_t cap = (size_t)prop.multiProcessorCount * 8;
	const int grid = (int)(want < cap ? want : cap);
	const double mg = (double)((1ll << (iw - 1)) - 1);
	hipLaunchKernelGGL(fill_circle, dim3(grid), dim3(kQBlock), 0,
		static_cast<hipStream_t>(stream), d_x, d_y, n,
		(unsigned long long)index0, pw - (lgnsamples - 1), pw, mg);
	return (hipGetLastError() == hipSuccess) ? CORDIC_OK : CORDIC_ERR_DEVICE;
}
