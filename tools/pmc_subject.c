/* pmc_subject.c -- the process rocprofv3's counter passes run (bench.py,
 * tools/bench_pmc.py): K launches of one workload's kernel through the C ABI,
 * nothing else.  A counter pass needs the kernel's launches and no more; the
 * Python start-up of a `bench.py` child cost ~5 s of each of the three passes
 * of a default run (round 5: 116 s for a 38 ms timed region).
 *
 *   pmc_subject KIND MODE IW OW XTRA PW NSTAGES LOG2N SHIFT STEPS [FLAGS]
 *     KIND   p2r | nco | r2p      through cordic_group (what bench.py times)
 *            p2rxy | ddc          cordic_plan_p2r / cordic_plan_mix
 *     MODE.. gencordic's -t (0 p2r, 1 r2p, 2 sp2r, 3 sr2p) -i -o -x -p -n
 *     SHIFT  phase ramp n << SHIFT (p2r, p2rxy)
 *     FLAGS  cordic_config.flags to OR in (CORDIC_FLAG_NO_SEED ...)
 * Inputs: the generated ramps of bench.py (cordic_tb.cpp:128,138), x = 2^(IW-1)-1,
 * y = 0, fcw = 0x01234567.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cordic_amd.h"

extern int hipDeviceSynchronize(void);	/* libamdhip64 (no HIP headers in a C99 TU) */

#define CHECK(call) do { int rc_ = (call); if (rc_ != CORDIC_OK) { \
	fprintf(stderr, "%s: %s\n", #call, cordic_strerror(rc_)); return 1; } } while (0)

int main(int argc, char **argv)
{
	if (argc < 11) {
		fprintf(stderr, "usage: %s KIND MODE IW OW XTRA PW NSTAGES LOG2N SHIFT "
			"STEPS [FLAGS]\n", argv[0]);
		return 2;
	}
	const char *kind = argv[1];
	const int mode = atoi(argv[2]), iw = atoi(argv[3]), ow = atoi(argv[4]);
	const int xtra = atoi(argv[5]), pw = atoi(argv[6]), ns = atoi(argv[7]);
	const int lg = atoi(argv[8]), shift = atoi(argv[9]), steps = atoi(argv[10]);
	const unsigned flags = argc > 11 ? (unsigned)strtoul(argv[11], NULL, 0) : 0;
	const uint64_t n = (uint64_t)1 << lg;
	const uint32_t fcw = 0x01234567u, mulx = 0x9E3779B1u, muly = 0x85EBCA77u;

	cordic_config cfg;
	CHECK(cordic_config_init(&cfg, mode, iw, ow, xtra, pw, ns));
	cfg.flags |= flags;
	const int32_t x0 = (int32_t)(((uint32_t)1 << (cfg.iw - 1)) - 1);

	if (!strcmp(kind, "p2r") || !strcmp(kind, "nco") || !strcmp(kind, "r2p")) {
		cordic_group *grp;
		int dev = 0;
		CHECK(cordic_group_create(&cfg, 1, &dev, 0, 1, &grp));
		CHECK(cordic_group_set_placement(grp, 0));
		if (kind[0] == 'p')
			CHECK(cordic_group_fill_phase_ramp(grp, n, shift));
		else if (kind[0] == 'r')
			CHECK(cordic_group_fill_iq_ramp(grp, n, mulx, muly, cfg.iw));
		for (int k = 0; k < steps + 1; k++) {
			if (kind[0] == 'p')
				CHECK(cordic_group_p2r_const(grp, n, x0, 0));
			else if (kind[0] == 'r')
				CHECK(cordic_group_r2p(grp, n));
			else
				CHECK(cordic_group_nco(grp, n, 0, fcw, x0, 0));
		}
		CHECK(cordic_group_sync(grp));
		cordic_group_destroy(grp);
		return 0;
	}
	if (!strcmp(kind, "p2rxy") || !strcmp(kind, "ddc")) {
		const int xy = !strcmp(kind, "p2rxy");
		void *p[5] = {0};
		/* two read + two written arrays; p2rxy reads a third */
		CHECK(cordic_arrays_alloc(4 * n, 2, 2, p, NULL));
		void *third[1] = {0};
		if (xy)
			CHECK(cordic_arrays_alloc(4 * n, 0, 1, third, NULL));
		cordic_plan *plan;
		CHECK(cordic_plan_create(&cfg, &plan));
		CHECK(cordic_fill_iq_ramp((int32_t *)p[0], (int32_t *)p[1], n, 0, mulx,
			muly, cfg.iw, NULL));
		if (xy)
			CHECK(cordic_fill_phase_ramp((uint32_t *)third[0], n, 0, shift, NULL));
		for (int k = 0; k < steps + 1; k++) {
			if (xy)
				CHECK(cordic_plan_p2r(plan, n, (int32_t *)p[0], (int32_t *)p[1],
					(uint32_t *)third[0], (int32_t *)p[2], (int32_t *)p[3],
					NULL));
			else
				CHECK(cordic_plan_mix(plan, n, 0, fcw, 0, (int32_t *)p[0],
					(int32_t *)p[1], (int32_t *)p[2], (int32_t *)p[3], NULL));
		}
		if (hipDeviceSynchronize() != 0) {
			fprintf(stderr, "hipDeviceSynchronize failed\n");
			return 1;
		}
		cordic_plan_destroy(plan);
		cordic_arrays_free(p, 4);
		if (xy)
			cordic_arrays_free(third, 1);
		return 0;
	}
	fprintf(stderr, "unknown KIND %s\n", kind);
	return 2;
}
