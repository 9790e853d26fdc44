/* sincos.c -- plain C caller of libcordic_amd.so: the sine/cosine generator
 * use of the core (bench/cpp/cordic_tb.cpp:61-69,127-139), start to finish.
 *
 *   gcc -std=c99 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       examples/sincos.c -L cordic_amd -lcordic_amd -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/cordic_amd -lm -o examples/sincos
 *   examples/sincos [-i IW] [-o OW] [-p PW] [-n NSTAGES] [-x XTRA]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "cordic_amd.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != CORDIC_OK) { \
	fprintf(stderr, "%s: %s\n", #call, cordic_strerror(rc_)); return 1; } } while (0)

int main(int argc, char **argv)
{
	int iw = 13, ow = 13, pw = -1, ns = -1, xtra = 2;
	for (int k = 1; k + 1 < argc; k += 2) {
		if (!strcmp(argv[k], "-i")) iw = atoi(argv[k + 1]);
		else if (!strcmp(argv[k], "-o")) ow = atoi(argv[k + 1]);
		else if (!strcmp(argv[k], "-p")) pw = atoi(argv[k + 1]);
		else if (!strcmp(argv[k], "-n")) ns = atoi(argv[k + 1]);
		else if (!strcmp(argv[k], "-x")) xtra = atoi(argv[k + 1]);
	}
	cordic_config cfg;
	CHECK(cordic_config_init(&cfg, CORDIC_P2R, iw, ow, xtra, pw, ns));
	printf("core: IW %d OW %d WW %d PW %d NSTAGES %d gain %.6f\n", cfg.iw, cfg.ow,
		cfg.ww, cfg.pw, cfg.nstages, cfg.gain);

	const size_t n = (size_t)1 << (cfg.pw > 24 ? 24 : cfg.pw);
	const int shift = cfg.pw - (cfg.pw > 24 ? 24 : cfg.pw);
	uint32_t *d_phase;
	int32_t *d_cos, *d_sin;
	if (hipMalloc((void **)&d_phase, n * 4) != hipSuccess ||
	    hipMalloc((void **)&d_cos, n * 4) != hipSuccess ||
	    hipMalloc((void **)&d_sin, n * 4) != hipSuccess) {
		fprintf(stderr, "no device memory (is there a GPU?)\n");
		return 1;
	}
	cordic_plan *core;
	CHECK(cordic_plan_create(&cfg, &core));			/* "generate" once */
	CHECK(cordic_fill_phase_ramp(d_phase, n, 0, shift, NULL));
	const int32_t amp = (1 << (cfg.iw - 1)) - 1;
	CHECK(cordic_plan_p2r_const(core, n, amp, 0, d_phase, d_cos, d_sin, NULL));
	if (hipDeviceSynchronize() != hipSuccess)
		return 1;

	int32_t *c = malloc(n * 4), *s = malloc(n * 4);
	if (!c || !s)
		return 1;
	hipMemcpy(c, d_cos, n * 4, hipMemcpyDeviceToHost);
	hipMemcpy(s, d_sin, n * 4, hipMemcpyDeviceToHost);

	/* bench/cpp/cordic_tb.cpp:238-260: scale and worst deviation from libm */
	const double scale = cfg.gain * amp * pow(2.0, cfg.ow - cfg.iw - 1);
	double worst = 0;
	for (size_t i = 0; i < n; i++) {
		const double ph = 2.0 * M_PI * (double)i / (double)n;
		const double e = hypot(c[i] - scale * cos(ph), s[i] - scale * sin(ph));
		if (e > worst) worst = e;
	}
	printf("%zu phases: worst |error| %.3f output units = %.2e of full scale "
		"(2^-NSTAGES = %.2e)\n", n, worst, worst / scale,
		pow(2.0, -cfg.nstages));
	printf("phase 0: (%d, %d)   45 deg: (%d, %d)   90 deg: (%d, %d)\n", c[0], s[0],
		c[n / 8], s[n / 8], c[n / 4], s[n / 4]);

	cordic_plan_destroy(core);
	hipFree(d_phase); hipFree(d_cos); hipFree(d_sin);
	free(c); free(s);
	return 0;
}
