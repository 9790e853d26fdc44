// cordic_tb.cpp -- the reference's acceptance benches, driven through the C ABI.
//
// C++ host program (the reference's language) that reproduces what
// bench/cpp/cordic_tb.cpp and bench/cpp/topolar_tb.cpp do with the Verilated
// cores, with the MI355X engine in their place: same sweeps, same statistics,
// same pass thresholds, same report lines, same exit status.
//
//   cordic_tb <gencordic args>        e.g.  cordic_tb -t p2r -i 13 -o 13 -x 2
//   cordic_tb -t r2p -i 13 -o 13 -x 2
//   cordic_tb -t qtbl -o 13 -p 18     (bench/cpp/quadtbl_tb.cpp)
//
// p2r / sp2r : bench/cpp/cordic_tb.cpp:61-69 (x = 2^(IW-1)-1, y = 0),
//              :127-139 (all 2^PW phases), :223-337 (statistics, thresholds),
//              :342-371 (SFDR; printed, not asserted, PW < 26 only).
// qtbl       : bench/cpp/quadtbl_tb.cpp:82-127 (min(2^PW, 2^26) phases, rounded
//              onto the PW-bit grid when PW > 26), :146-177 (max error against
//              sin * (2^(OW-1)-1), threshold |TBL_ERR| + 2), :182-211 (SFDR).
// r2p / sr2p : bench/cpp/topolar_tb.cpp:127-147 (two turns of a circle of
//              radius 2^(IW-1)-1, (int) truncation), :222-256, :303-315.
//
// Build: see tools/Makefile.  Needs a GPU; there is no CPU fallback.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cordic_amd.h"

#define HIP_OK(x) do { if ((x) != hipSuccess) { \
	fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return EXIT_FAILURE; } } while (0)
#define CORDIC_OK_OR_DIE(x) do { int rc_ = (x); if (rc_ != CORDIC_OK) { \
	fprintf(stderr, "ERR: %s\n", cordic_strerror(rc_)); return EXIT_FAILURE; } } while (0)

// in-place radix-2 FFT (the reference links FFTW for this, bench/cpp/fftw.c)
static void fft(std::vector<std::complex<double>> &a)
{
	const size_t n = a.size();
	for (size_t i = 1, j = 0; i < n; i++) {
		size_t bit = n >> 1;
		for (; j & bit; bit >>= 1)
			j ^= bit;
		j ^= bit;
		if (i < j)
			std::swap(a[i], a[j]);
	}
	for (size_t len = 2; len <= n; len <<= 1) {
		const double ang = -2.0 * M_PI / (double)len;
		const std::complex<double> wl(cos(ang), sin(ang));
		for (size_t i = 0; i < n; i += len) {
			std::complex<double> w(1.0, 0.0);
			for (size_t k = 0; k < len / 2; k++) {
				const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
				a[i + k] = u + v;
				a[i + k + len / 2] = u - v;
				w *= wl;
			}
		}
	}
}

static int run_p2r(const cordic_config &cfg)
{
	const int IW = cfg.iw, OW = cfg.ow, PW = cfg.pw;
	if (PW > 28) {
		fprintf(stderr, "ERR: an exhaustive 2^%d sweep does not fit this bench\n", PW);
		return EXIT_FAILURE;
	}
	const size_t n = (size_t)1 << PW;
	const int32_t ix = (int32_t)((1ul << (IW - 1)) - 1), iy = 0;

	uint32_t *d_phase; int32_t *d_x, *d_y;
	HIP_OK(hipMalloc((void **)&d_phase, n * 4));
	HIP_OK(hipMalloc((void **)&d_x, n * 4));
	HIP_OK(hipMalloc((void **)&d_y, n * 4));
	CORDIC_OK_OR_DIE(cordic_fill_phase_ramp(d_phase, n, 0, 0, nullptr));
	cordic_plan *plan;
	CORDIC_OK_OR_DIE(cordic_plan_create(&cfg, &plan));
	CORDIC_OK_OR_DIE(cordic_plan_p2r_const(plan, n, ix, iy, d_phase, d_x, d_y, nullptr));
	HIP_OK(hipDeviceSynchronize());
	std::vector<int32_t> xval(n), yval(n);
	HIP_OK(hipMemcpy(xval.data(), d_x, n * 4, hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(yval.data(), d_y, n * 4, hipMemcpyDeviceToHost));
	cordic_plan_destroy(plan);
	(void)hipFree(d_phase); (void)hipFree(d_x); (void)hipFree(d_y);

	// statistics: cordic_tb.cpp:223-279
	const double GAIN = cfg.gain;
	double scale = sqrt((double)ix * ix + (double)iy * iy);
	double mxerr = 0, averr = 0, mag = 0, sumxy = 0, sumsq = 0;
	const double outscale = pow(2.0, -(IW + 1 - OW));
	for (size_t i = 0; i < n; i++) {
		const double ph = (double)i * M_PI * 2.0 / (double)(1ul << PW);
		double dx = (cos(ph) * ix - sin(ph) * iy) * GAIN * outscale;
		double dy = (sin(ph) * ix + cos(ph) * iy) * GAIN * outscale;
		mag += xval[i] * (double)xval[i] + yval[i] * (double)yval[i];
		double err = (dx - xval[i]) * (dx - xval[i]) + (dy - yval[i]) * (dy - yval[i]);
		sumxy += dx * xval[i] + dy * yval[i];
		sumsq += xval[i] * (double)xval[i] + yval[i] * (double)yval[i];
		averr += err;
		err = sqrt(err);
		if (err > mxerr) mxerr = err;
	}
	bool failed = false;
	const double expected_err = cfg.quantization_variance
		+ cfg.phase_variance_rad * scale * scale * GAIN * GAIN;
	averr = sqrt(averr / (double)n);
	if (mag <= 0) { printf("ERR: Negative magnitude, %f\n", mag); printf("TEST FAILURE\n"); return EXIT_FAILURE; }
	mag = sqrt(mag / (double)n);

	// report: cordic_tb.cpp:315-337
	printf("AVG Err: %.6f Units (%.6f Relative, %.4f Units expected)\n",
		averr, averr / mag, sqrt(expected_err));
	if (averr > 1.5 * sqrt(expected_err))
		failed = true;
	printf("MAX Err: %.6f Units (%.6f Relative, %.6f threshold)\n", mxerr,
		mxerr / mag, 5.2 * sqrt(expected_err));
	if (mxerr > 5.2 * sqrt(expected_err)) {
		printf("ERR: Maximum error is out of bounds\n");
		failed = true;
	}
	printf("  Mag  : %.6f\n", mag);
	printf("(Gain) : %.6f\n", GAIN);
	printf("(alpha): %.6f\n", sumxy / sumsq);
	scale *= GAIN;
	printf("CNR    : %.2f dB (expected %.2f dB)\n",
		10.0 * log(scale * scale / (averr * averr)) / log(10.0),
		cfg.best_possible_cnr);
	if (fabs(sumxy / sumsq - 1.0) > 0.01) {
		printf("(alpha)is out of bounds!\n");
		failed = true;
	}
	if (failed) { printf("TEST FAILURE\n"); return EXIT_FAILURE; }

	// SFDR: cordic_tb.cpp:342-371
	if (PW < 26) {
		std::vector<std::complex<double>> o(n);
		for (size_t k = 0; k < n; k++)
			o[k] = std::complex<double>(xval[k], yval[k]);
		fft(o);
		const double master = std::norm(o[1]);
		double spur = std::norm(o[0]);
		for (size_t k = 2; k < n; k++)
			if (std::norm(o[k]) > spur) spur = std::norm(o[k]);
		printf("SFDR = %7.2f dBc\n", 10 * log(master / spur) / log(10.));
	} else
		printf("Too many phase bits ... skipping SFDR calculation\n");
	printf("SUCCESS!!\n");
	return EXIT_SUCCESS;
}

static int run_r2p(const cordic_config &cfg)
{
	const int IW = cfg.iw, OW = cfg.ow, PW = cfg.pw;
	if (PW > 28) {
		fprintf(stderr, "ERR: an exhaustive 2^%d sweep does not fit this bench\n", PW);
		return EXIT_FAILURE;
	}
	const size_t n = (size_t)1 << PW;
	const double MAXPHASE = pow(2.0, PW), RAD_TO_PHASE = MAXPHASE / M_PI / 2.0;
	std::vector<int32_t> ixval(n), iyval(n), omag(n);
	std::vector<uint32_t> ophase(n);
	std::vector<double> dpdata(n);
	const double mg = (double)((1l << (IW - 1)) - 1);
	for (size_t i = 0; i < n; i++) {	// topolar_tb.cpp:127-147
		const long lv = ((long)i) << 1;	// LGNSAMPLES == PW
		const int ip = (int)lv;
		const double ph = ip * M_PI / (double)(1ul << (PW - 1));
		ixval[i] = (int)(mg * cos(ph));
		iyval[i] = (int)(mg * sin(ph));
		dpdata[i] = atan2((double)iyval[i], (double)ixval[i]);
	}
	CORDIC_OK_OR_DIE(cordic_r2p_host(&cfg, n, ixval.data(), iyval.data(),
			omag.data(), ophase.data()));

	double mxperr = 0, mxverr = 0, sum_perr = 0;	// topolar_tb.cpp:222-256
	for (size_t i = 0; i < n; i++) {
		double epdata = dpdata[i] * RAD_TO_PHASE;
		if (epdata < 0.0) epdata += MAXPHASE;
		// the bench sign extends o_phase from PW bits (:177-181)
		long op = (long)ophase[i];
		if (op >= (1l << (PW - 1))) op -= (1l << PW);
		double dperr = (double)op - epdata;
		while (dperr > MAXPHASE / 2.) dperr -= MAXPHASE;
		while (dperr < -MAXPHASE / 2.) dperr += MAXPHASE;
		if (fabs(dperr) > mxperr) mxperr = fabs(dperr);
		sum_perr += dperr * dperr;
		const double emag = mg * pow(2., (IW - 1 - OW));
		const double mgerr = fabs(omag[i] - emag * cfg.gain);
		if (mgerr > mxverr) mxverr = mgerr;
	}
	sum_perr /= (double)n;
	bool failed = false;			// topolar_tb.cpp:303-330
	double expected_phase_err = sqrt(cfg.phase_variance_rad * RAD_TO_PHASE * RAD_TO_PHASE);
	if (expected_phase_err < 1.0) expected_phase_err = 1.0;
	if (mxperr > 3.4 * expected_phase_err) failed = true;
	if (mxverr > 2.0 * sqrt(cfg.quantization_variance)) failed = true;
	printf("Max phase     error: %.2f (%.6f Rel)\n", mxperr,
		mxperr / (2.0 * (double)(1ul << (PW - 1))));
	printf("Max magnitude error: %9.6f, expect %.2f\n", mxverr,
		2.0 * sqrt(cfg.quantization_variance));
	printf("Avg phase err:       %9.6f, expect %.2f\n", sqrt(sum_perr),
		sqrt(cfg.phase_variance_rad) * RAD_TO_PHASE);
	if (failed) { printf("TEST FAILED!!\n"); return EXIT_FAILURE; }
	printf("SUCCESS\n");
	return EXIT_SUCCESS;
}

static int run_qtbl(int argc, char **argv)
{
	int iw = -1, ow = -1, pw = -1, xtra = 2;
	for (int k = 1; k + 1 < argc; k++) {
		if (!strcmp(argv[k], "-i")) iw = atoi(argv[k + 1]);
		else if (!strcmp(argv[k], "-o")) ow = atoi(argv[k + 1]);
		else if (!strcmp(argv[k], "-p")) pw = atoi(argv[k + 1]);
		else if (!strcmp(argv[k], "-x")) xtra = atoi(argv[k + 1]);
	}
	cordic_quad_config qc;
	CORDIC_OK_OR_DIE(cordic_quad_config_init(&qc, iw, ow, xtra, pw));
	const int PW = qc.pw, OW = qc.ow;
	const long LGNSAMPLES = (PW > 26) ? 26 : PW;
	const size_t n = (size_t)1 << LGNSAMPLES;

	// quadtbl_tb.cpp:100-114
	std::vector<uint32_t> pdata(n);
	const int shift = (int)(PW - LGNSAMPLES);
	for (size_t i = 0; i < n; i++)
		pdata[i] = (uint32_t)((uint64_t)i << shift);
	std::vector<int32_t> sdata(n);
	uint32_t *d_ph; int32_t *d_o;
	cordic_quad *core = nullptr;
	HIP_OK(hipMalloc((void **)&d_ph, n * 4));
	HIP_OK(hipMalloc((void **)&d_o, n * 4));
	HIP_OK(hipMemcpy(d_ph, pdata.data(), n * 4, hipMemcpyHostToDevice));
	CORDIC_OK_OR_DIE(cordic_quad_create(&qc, &core));
	CORDIC_OK_OR_DIE(cordic_quad_lookup(core, n, d_ph, d_o, nullptr));
	HIP_OK(hipDeviceSynchronize());
	HIP_OK(hipMemcpy(sdata.data(), d_o, n * 4, hipMemcpyDeviceToHost));
	cordic_quad_destroy(core);
	(void)hipFree(d_ph); (void)hipFree(d_o);

	// quadtbl_tb.cpp:146-177
	double mxerr = 0.0;
	int imxv = 0, imnv = 0;
	for (size_t i = 0; i < n; i++) {
		double ph = (double)(int)pdata[i];
		ph = ph * M_PI * 2.0 / (double)(1ul << PW);
		const double scl = ((1 << (OW - 1)) - 1);
		const double dsin = sin(ph) * scl;
		const double err = fabs(dsin - sdata[i]);
		if (err > mxerr) mxerr = err;
		if (sdata[i] > imxv) imxv = sdata[i];
		else if (sdata[i] < imnv) imnv = sdata[i];
	}
	printf("MXERR: %f (Expected %f)\n", mxerr, qc.tbl_err);
	const bool failed = fabs(mxerr) > fabs(qc.tbl_err) + 2.;
	printf("MXVAL: 0x%08x\n", imxv);
	printf("MNVAL: 0x%08x\n", imnv);
	if (failed) { printf("TEST FAILURE\n"); return EXIT_FAILURE; }

	// quadtbl_tb.cpp:182-211
	if (PW < 26 && n == ((size_t)1 << PW)) {
		std::vector<std::complex<double>> o(n);
		for (size_t k = 0; k < n; k++)
			o[k] = std::complex<double>(sdata[(k + n / 4) & (n - 1)], sdata[k]);
		fft(o);
		const double master = std::norm(o[1]);
		double spur = std::norm(o[0]);
		for (size_t k = 2; k < n; k++)
			if (std::norm(o[k]) > spur) spur = std::norm(o[k]);
		printf("SFDR = %7.2f dBc\n", 10 * log(master / spur) / log(10.));
	} else if (PW >= 26)
		printf("Too many phase bits ... skipping SFDR calculation\n");
	printf("SUCCESS!!\n");
	return EXIT_SUCCESS;
}

int main(int argc, char **argv)
{
	for (int k = 1; k + 1 < argc; k++)
		if (!strcmp(argv[k], "-t") && !strcmp(argv[k + 1], "qtbl"))
			return run_qtbl(argc, argv);
	cordic_config cfg;
	char fname[256];
	int hdr = 0;
	const int rc = cordic_config_from_args(&cfg, argc, argv, fname, sizeof fname, &hdr);
	if (rc != CORDIC_OK) {
		fprintf(stderr, "ERR: %s\n", cordic_strerror(rc));
		return EXIT_FAILURE;
	}
	if (cfg.mode == CORDIC_P2R || cfg.mode == CORDIC_SP2R)
		return run_p2r(cfg);
	return run_r2p(cfg);
}
