/* multi_gpu.c -- plain C caller of the multi-GPU part of the C ABI
 * (include/cordic_amd.h, "multi-GPU jobs"): BASELINE.json configs[3],
 * basiccordic 24-stage / 32-bit, phase[n] = (uint32)n, sharded by global
 * sample index over every visible MI355X from ONE host process -- no process
 * group, no collective on the data path; the shards' digests are summed on the
 * host and, with -g, the results are forwarded to device 0 chunk by chunk
 * behind the compute (hipMemcpyPeerAsync over xGMI).
 *
 *   gcc -std=c99 -I include examples/multi_gpu.c -L cordic_amd -lcordic_amd \
 *       -Wl,-rpath,$PWD/cordic_amd -o tools/multi_gpu
 *   tools/multi_gpu [-l LOG2_SAMPLES_PER_GPU] [-s SHARDS] [-d DEV,DEV,...]
 *                   [-n NSTAGES] [-k STEPS] [-g] [-c]
 *     -s/-d  shards and the device each one runs on (default: one per GPU;
 *            a device may be listed twice)
 *     -g     also time compute + gather to the first listed device
 *     -c     check: the same global range as ONE shard must give the same digest
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cordic_amd.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != CORDIC_OK) { \
	fprintf(stderr, "%s: %s\n", #call, cordic_strerror(rc_)); return 1; } } while (0)

int main(int argc, char **argv)
{
	int lg = 30, shards = 0, ns = 24, steps = 10, gather = 0, check = 0;
	int devices[64], ndev_listed = 0;
	for (int k = 1; k < argc; k++) {
		if (!strcmp(argv[k], "-g")) gather = 1;
		else if (!strcmp(argv[k], "-c")) check = 1;
		else if (k + 1 >= argc) { fprintf(stderr, "missing value for %s\n", argv[k]); return 2; }
		else if (!strcmp(argv[k], "-l")) lg = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-s")) shards = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-n")) ns = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-k")) steps = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-d")) {
			char *tok = strtok(argv[++k], ",");
			while (tok && ndev_listed < 64) {
				devices[ndev_listed++] = atoi(tok);
				tok = strtok(NULL, ",");
			}
		}
	}
	const int visible = cordic_device_count();
	if (visible <= 0) {
		fprintf(stderr, "no HIP device visible\n");
		return 1;
	}
	if (ndev_listed && !shards) shards = ndev_listed;
	if (!shards) shards = visible;
	if (shards > 64 || (ndev_listed && ndev_listed != shards)) {
		fprintf(stderr, "-s and -d disagree\n");
		return 2;
	}
	if (!ndev_listed)
		for (int s = 0; s < shards; s++)
			devices[s] = s % visible;

	cordic_config cfg;
	CHECK(cordic_config_init(&cfg, CORDIC_P2R, 32, 32, 2, 32, ns));
	const uint64_t n_total = ((uint64_t)1 << lg) * (uint64_t)shards;
	const int32_t amp = 0x7fffffff;
	printf("core: IW %d OW %d WW %d PW %d NSTAGES %d; %d shard(s) on %d visible "
		"device(s), 2^%d samples each\n", cfg.iw, cfg.ow, cfg.ww, cfg.pw,
		cfg.nstages, shards, visible, lg);

	cordic_group *grp;
	CHECK(cordic_group_create(&cfg, shards, devices, 0, shards, &grp));
	/* every shard writes its own ramp phase[n] = (uint32)n from the global index */
	CHECK(cordic_group_fill_phase_ramp(grp, n_total, 0));
	CHECK(cordic_group_p2r_const(grp, n_total, amp, 0));	/* warm-up */
	CHECK(cordic_group_sync(grp));

	float ms, per[64];
	CHECK(cordic_group_mark(grp, 0));
	for (int k = 0; k < steps; k++)
		CHECK(cordic_group_p2r_const(grp, n_total, amp, 0));
	CHECK(cordic_group_mark(grp, 1));
	CHECK(cordic_group_elapsed(grp, 0, 1, &ms, per));
	printf("compute only : %8.3f ms per step, %9.1f Gsample/s over all shards\n",
		ms / steps, (double)n_total * steps / (ms * 1e-3) / 1e9);
	for (int s = 0; s < shards; s++)
		printf("  shard %d on device %d: %8.3f ms per step\n", s, devices[s],
			per[s] / steps);

	uint64_t digest;
	CHECK(cordic_group_digest(grp, n_total, &digest));
	printf("digest of all outputs: %016llx\n", (unsigned long long)digest);

	if (gather) {
		/* the consumer's arrays live on the first shard's device */
		cordic_group *root;
		int one = devices[0];
		void *g0, *g1;
		CHECK(cordic_group_create(&cfg, 1, &one, 0, 1, &root));
		CHECK(cordic_group_reserve(root, n_total, 0));
		CHECK(cordic_group_buffers(root, 0, NULL, NULL, NULL, &g0, &g1, NULL));
		CHECK(cordic_group_set_gather(grp, devices[0], (int32_t *)g0,
			(int32_t *)g1, 8));
		CHECK(cordic_group_p2r_const(grp, n_total, amp, 0));
		CHECK(cordic_group_sync(grp));
		CHECK(cordic_group_mark(grp, 2));
		for (int k = 0; k < steps; k++)
			CHECK(cordic_group_p2r_const(grp, n_total, amp, 0));
		CHECK(cordic_group_sync(grp));		/* copies included */
		CHECK(cordic_group_mark(grp, 3));
		CHECK(cordic_group_elapsed(grp, 2, 3, &ms, NULL));
		printf("compute + gather to device %d (8 chunks, peer copies behind the "
			"compute): %8.3f ms per step, %9.1f Gsample/s\n", devices[0],
			ms / steps, (double)n_total * steps / (ms * 1e-3) / 1e9);
		/* what arrived equals what the shards hold */
		uint64_t dg;
		CHECK(cordic_group_digest(root, n_total, &dg));
		printf("digest of the gathered arrays: %016llx  %s\n",
			(unsigned long long)dg, dg == digest ? "(equal)" : "(MISMATCH)");
		CHECK(cordic_group_set_gather(grp, -1, NULL, NULL, 0));
		cordic_group_destroy(root);
		if (dg != digest)
			return 1;
	}
	if (check) {
		cordic_group *single;
		int one = devices[0];
		uint64_t d1;
		CHECK(cordic_group_create(&cfg, 1, &one, 0, 1, &single));
		CHECK(cordic_group_fill_phase_ramp(single, n_total, 0));
		CHECK(cordic_group_p2r_const(single, n_total, amp, 0));
		CHECK(cordic_group_digest(single, n_total, &d1));
		printf("same range as one shard      : %016llx  %s\n",
			(unsigned long long)d1, d1 == digest ? "(equal)" : "(MISMATCH)");
		cordic_group_destroy(single);
		if (d1 != digest)
			return 1;
	}
	cordic_group_destroy(grp);
	return 0;
}
