/* channeliser.c -- plain C caller of the job-set part of the C ABI
 * (include/cordic_amd.h, "job sets"): what a channeliser does every block
 * period -- NCH short blocks of I/Q samples, each mixed down by its own NCO
 * (the core with all three ports live, rtl/cordic.v:58-63) and then converted
 * to magnitude + phase (rtl/topolar.v:59-64; the per-sample loop of
 * bench/cpp/topolar_tb.cpp:127-147) -- as TWO launches for all channels
 * (CORDIC_JOBS_MIX, CORDIC_JOBS_R2P) instead of 2 NCH.  Checks the job sets'
 * outputs word for word against the per-block calls and prints both rates.
 *
 *   gcc -std=c99 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       examples/channeliser.c -L cordic_amd -lcordic_amd -L /opt/rocm/lib \
 *       -lamdhip64 -Wl,-rpath,$PWD/cordic_amd -o tools/channeliser
 *   tools/channeliser [-c CHANNELS] [-l LOG2_BLOCK] [-k REPS]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "cordic_amd.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != CORDIC_OK) { \
	fprintf(stderr, "%s: %s\n", #call, cordic_strerror(rc_)); return 1; } } while (0)
#define HIP(call) do { if ((call) != hipSuccess) { \
	fprintf(stderr, "%s failed\n", #call); return 1; } } while (0)

static float ms_between(hipEvent_t a, hipEvent_t b)
{
	float ms = 0.f;
	hipEventSynchronize(b);
	hipEventElapsedTime(&ms, a, b);
	return ms;
}

int main(int argc, char **argv)
{
	int nch = 256, lg = 14, reps = 10;
	for (int k = 1; k + 1 < argc; k += 2) {
		if (!strcmp(argv[k], "-c")) nch = atoi(argv[k + 1]);
		else if (!strcmp(argv[k], "-l")) lg = atoi(argv[k + 1]);
		else if (!strcmp(argv[k], "-k")) reps = atoi(argv[k + 1]);
	}
	if (nch < 1 || nch > 65536 || lg < 2 || lg > 24 || reps < 1)
		return 2;
	cordic_config mixer, conv;
	/* 16-stage 32-bit rotator (BASELINE config 2's core), 24-bit converter */
	CHECK(cordic_config_init(&mixer, CORDIC_P2R, 32, 32, 2, 32, 16));
	CHECK(cordic_config_init(&conv, CORDIC_R2P, 24, 24, 2, -1, 20));
	cordic_plan *pm, *pc;
	CHECK(cordic_plan_create(&mixer, &pm));
	CHECK(cordic_plan_create(&conv, &pc));

	/* ragged blocks: channel c holds (1 << lg) - 3 (c mod 5) samples */
	size_t total = 0;
	size_t *off = malloc(((size_t)nch + 1) * sizeof *off);
	if (!off) return 1;
	for (int c = 0; c < nch; c++) {
		off[c] = total;
		total += ((size_t)1 << lg) - 3u * (unsigned)(c % 5);
	}
	off[nch] = total;
	int32_t *d_i, *d_q, *d_bi, *d_bq, *d_mag, *d_ph, *d_ref;
	HIP(hipMalloc((void **)&d_i, total * 4));
	HIP(hipMalloc((void **)&d_q, total * 4));
	HIP(hipMalloc((void **)&d_bi, total * 4));	/* baseband I / Q */
	HIP(hipMalloc((void **)&d_bq, total * 4));
	HIP(hipMalloc((void **)&d_mag, total * 4));
	HIP(hipMalloc((void **)&d_ph, total * 4));
	HIP(hipMalloc((void **)&d_ref, 4 * total * 4));	/* per-block results */
	CHECK(cordic_fill_iq_ramp(d_i, d_q, total, 0, 0x9E3779B1u, 0x85EBCA77u, 24, NULL));

	cordic_job *mj = calloc((size_t)nch, sizeof *mj), *cj = calloc((size_t)nch, sizeof *cj);
	if (!mj || !cj) return 1;
	for (int c = 0; c < nch; c++) {
		const size_t n = off[c + 1] - off[c];
		mj[c].d_xval = d_i + off[c];  mj[c].d_yval = d_q + off[c];
		mj[c].phase0 = 0; mj[c].fcw = 0x01234567u * (uint32_t)(c + 1); mj[c].index0 = 0;
		mj[c].d_oxval = d_bi + off[c]; mj[c].d_oyval = d_bq + off[c]; mj[c].n = n;
		/* the converter takes the low 24 bits of the mixer's 32-bit words */
		cj[c].d_xval = d_bi + off[c]; cj[c].d_yval = d_bq + off[c];
		cj[c].d_oxval = d_mag + off[c]; cj[c].d_oyval = d_ph + off[c]; cj[c].n = n;
	}
	cordic_jobset *ms, *cs;
	CHECK(cordic_jobset_create(pm, CORDIC_JOBS_MIX, (size_t)nch, mj, &ms));
	CHECK(cordic_jobset_create(pc, CORDIC_JOBS_R2P, (size_t)nch, cj, &cs));
	uint32_t tiles = 0, tails = 0;
	CHECK(cordic_jobset_info(cs, NULL, &tiles, &tails));
	printf("%d channels x ~2^%d samples (%zu in all); converter set: %u tiles, %u "
		"trailing samples\n", nch, lg, total, tiles, tails);

	hipEvent_t e0, e1;
	HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
	/* (a) two launches for everything */
	CHECK(cordic_plan_run_jobs(pm, ms, 0, 0, NULL));
	CHECK(cordic_plan_run_jobs(pc, cs, 0, 0, NULL));
	HIP(hipEventRecord(e0, NULL));
	for (int k = 0; k < reps; k++) {
		CHECK(cordic_plan_run_jobs(pm, ms, 0, 0, NULL));
		CHECK(cordic_plan_run_jobs(pc, cs, 0, 0, NULL));
	}
	HIP(hipEventRecord(e1, NULL));
	const float ms_sets = ms_between(e0, e1) / (float)reps;
	/* (b) one call per block and stage, into the reference arrays */
	int32_t *r_bi = d_ref, *r_bq = d_ref + total, *r_mag = d_ref + 2 * total,
		*r_ph = d_ref + 3 * total;
	HIP(hipEventRecord(e0, NULL));
	for (int c = 0; c < nch; c++) {
		const size_t n = off[c + 1] - off[c];
		CHECK(cordic_plan_mix(pm, n, 0, mj[c].fcw, 0, d_i + off[c], d_q + off[c],
			r_bi + off[c], r_bq + off[c], NULL));
		CHECK(cordic_r2p(&conv, n, r_bi + off[c], r_bq + off[c], r_mag + off[c],
			(uint32_t *)(r_ph + off[c]), NULL));
	}
	HIP(hipEventRecord(e1, NULL));
	const float ms_calls = ms_between(e0, e1);

	/* word for word */
	int32_t *a = malloc(total * 4), *b = malloc(total * 4);
	if (!a || !b) return 1;
	const int32_t *got[4] = {d_bi, d_bq, d_mag, d_ph}, *want[4] = {r_bi, r_bq, r_mag, r_ph};
	const char *name[4] = {"baseband I", "baseband Q", "magnitude", "phase"};
	int bad = 0;
	for (int k = 0; k < 4; k++) {
		HIP(hipMemcpy(a, got[k], total * 4, hipMemcpyDeviceToHost));
		HIP(hipMemcpy(b, want[k], total * 4, hipMemcpyDeviceToHost));
		const int same = memcmp(a, b, total * 4) == 0;
		printf("%-11s job sets vs per-block calls: %s\n", name[k], same ? "equal" : "DIFFER");
		bad |= !same;
	}
	printf("two job sets : %8.3f ms per block period = %7.1f Msamples/s through both stages\n",
		ms_sets, (double)total / ms_sets / 1e3);
	printf("%5d calls  : %8.3f ms per block period = %7.1f Msamples/s\n", 2 * nch,
		ms_calls, (double)total / ms_calls / 1e3);
	cordic_jobset_destroy(ms); cordic_jobset_destroy(cs);
	cordic_plan_destroy(pm); cordic_plan_destroy(pc);
	return bad;
}
