/* multi_proc.c -- the process-per-GPU layout of the multi-GPU part of the C ABI
 * (include/cordic_amd.h, "multi-GPU jobs") in plain C, without MPI or Python:
 * BASELINE.json configs[3], basiccordic 24-stage / 32-bit, phase[n] =
 * (uint32)n.  The launcher forks one worker per GPU BEFORE anything touches
 * the HIP runtime; worker r creates a one-shard cordic_group (shard r of R,
 * on device r), generates its own ramp from the global index and computes it.
 * The only traffic between the processes is
 *   - the 128-byte RCCL id, from worker 0 to the others through pipes,
 *   - the final gather of the results to worker 0's GPU: ncclSend / ncclRecv
 *     over xGMI, piece by piece behind the compute
 *     (cordic_group_rccl_init + cordic_group_set_gather_rccl),
 *   - each worker's 8-byte digest back to the launcher, which sums them and
 *     compares the sum with the digest of the gathered arrays.
 *
 *   gcc -std=c99 -D_GNU_SOURCE -I include examples/multi_proc.c -L cordic_amd \
 *       -lcordic_amd -Wl,-rpath,$PWD/cordic_amd -o tools/multi_proc
 *   tools/multi_proc [-r WORKERS] [-l LOG2_SAMPLES_PER_GPU] [-n NSTAGES]
 *                    [-k STEPS] [-c CHUNKS] [-d DEV,DEV,...] [-R ROOT] [-t TOTAL]
 *     -r  workers = GPUs used (default 1; there is no HIP call in the launcher
 *         to count them with); -d the device of each worker (default 0,1,2...);
 *     -R  the worker whose GPU collects the results (default 0); -t the job
 *         size in samples when it is not WORKERS * 2^L (ragged shards)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/wait.h>

#include "cordic_amd.h"

#define MAXW 64

#define CHECK(call) do { int rc_ = (call); if (rc_ != CORDIC_OK) { \
	fprintf(stderr, "[worker %d] %s: %s\n", rank, #call, cordic_strerror(rc_)); \
	return 1; } } while (0)

struct report {			/* worker -> launcher */
	uint64_t shard_digest, gathered_digest;
	float	compute_ms, gather_ms;
};

static int read_all(int fd, void *buf, size_t n)
{
	char *p = buf;
	while (n) {
		ssize_t k = read(fd, p, n);
		if (k <= 0) return -1;
		p += k; n -= (size_t)k;
	}
	return 0;
}

static int write_all(int fd, const void *buf, size_t n)
{
	const char *p = buf;
	while (n) {
		ssize_t k = write(fd, p, n);
		if (k <= 0) return -1;
		p += k; n -= (size_t)k;
	}
	return 0;
}

static int worker(int rank, int workers, int device, uint64_t n_total, int ns,
		int steps, int chunks, int root_rank, int id_in, const int *id_out,
		int report_fd)
{
	cordic_config cfg;
	CHECK(cordic_config_init(&cfg, CORDIC_P2R, 32, 32, 2, 32, ns));
	const int32_t amp = 0x7fffffff;

	cordic_group *grp;
	CHECK(cordic_group_create(&cfg, 1, &device, rank, workers, &grp));

	/* RCCL bootstrap: worker 0 makes the id, everybody joins */
	unsigned char id[CORDIC_RCCL_ID_BYTES];
	if (rank == 0) {
		CHECK(cordic_rccl_unique_id(id));
		for (int r = 1; r < workers; r++)
			if (write_all(id_out[r], id, sizeof id)) return 1;
	} else if (read_all(id_in, id, sizeof id)) {
		return 1;
	}
	CHECK(cordic_group_rccl_init(grp, id));

	CHECK(cordic_group_fill_phase_ramp(grp, n_total, 0));
	CHECK(cordic_group_p2r_const(grp, n_total, amp, 0));	/* warm-up */
	CHECK(cordic_group_sync(grp));

	struct report rep;
	memset(&rep, 0, sizeof rep);
	CHECK(cordic_group_mark(grp, 0));
	for (int k = 0; k < steps; k++)
		CHECK(cordic_group_p2r_const(grp, n_total, amp, 0));
	CHECK(cordic_group_mark(grp, 1));
	CHECK(cordic_group_elapsed(grp, 0, 1, &rep.compute_ms, NULL));
	rep.compute_ms /= (float)steps;
	CHECK(cordic_group_digest(grp, n_total, &rep.shard_digest));

	/* the consumer's arrays: on the root worker's device, the whole job */
	cordic_group *root = NULL;
	void *g0 = NULL, *g1 = NULL;
	if (rank == root_rank) {
		CHECK(cordic_group_create(&cfg, 1, &device, 0, 1, &root));
		CHECK(cordic_group_reserve(root, n_total, 0));
		CHECK(cordic_group_buffers(root, 0, NULL, NULL, NULL, &g0, &g1, NULL));
	}
	CHECK(cordic_group_set_gather_rccl(grp, root_rank, (int32_t *)g0, (int32_t *)g1, chunks));
	CHECK(cordic_group_p2r_const(grp, n_total, amp, 0));	/* warm-up */
	CHECK(cordic_group_sync(grp));
	CHECK(cordic_group_mark(grp, 2));
	for (int k = 0; k < steps; k++)
		CHECK(cordic_group_p2r_const(grp, n_total, amp, 0));
	CHECK(cordic_group_sync(grp));			/* transfers included */
	CHECK(cordic_group_mark(grp, 3));
	CHECK(cordic_group_elapsed(grp, 2, 3, &rep.gather_ms, NULL));
	rep.gather_ms /= (float)steps;
	CHECK(cordic_group_set_gather_rccl(grp, -1, NULL, NULL, 1));
	if (rank == root_rank) {
		CHECK(cordic_group_digest(root, n_total, &rep.gathered_digest));
		cordic_group_destroy(root);
	}
	cordic_group_destroy(grp);
	return write_all(report_fd, &rep, sizeof rep) ? 1 : 0;
}

int main(int argc, char **argv)
{
	int workers = 1, lg = 28, ns = 24, steps = 10, chunks = 8, root = 0;
	unsigned long long total = 0;
	int devices[MAXW], listed = 0;
	for (int k = 1; k < argc; k++) {
		if (k + 1 >= argc) { fprintf(stderr, "missing value for %s\n", argv[k]); return 2; }
		else if (!strcmp(argv[k], "-r")) workers = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-l")) lg = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-n")) ns = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-k")) steps = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-c")) chunks = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-R")) root = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-t")) total = strtoull(argv[++k], NULL, 0);
		else if (!strcmp(argv[k], "-d")) {
			char *tok = strtok(argv[++k], ",");
			while (tok && listed < MAXW) {
				devices[listed++] = atoi(tok);
				tok = strtok(NULL, ",");
			}
		} else { fprintf(stderr, "unknown option %s\n", argv[k]); return 2; }
	}
	if (listed && workers == 1) workers = listed;
	if (workers < 1 || workers > MAXW || (listed && listed != workers) || steps < 1
			|| lg < 0 || lg > 32 || root < 0 || root >= workers) {
		fprintf(stderr, "bad -r / -d / -k / -l / -R\n");
		return 2;
	}
	const uint64_t n_job = total ? (uint64_t)total
				     : ((uint64_t)1 << lg) * (uint64_t)workers;
	if (!listed)
		for (int r = 0; r < workers; r++) devices[r] = r;

	int idp[MAXW][2], rp[MAXW][2], id_out[MAXW];
	pid_t pid[MAXW];
	for (int r = 0; r < workers; r++) {
		if (pipe(idp[r]) || pipe(rp[r])) { perror("pipe"); return 1; }
		id_out[r] = idp[r][1];
	}
	fflush(stdout);
	for (int r = 0; r < workers; r++) {
		pid[r] = fork();
		if (pid[r] < 0) { perror("fork"); return 1; }
		if (pid[r] == 0) {
			for (int q = 0; q < workers; q++) {
				close(rp[q][0]);
				if (q != r) { close(rp[q][1]); close(idp[q][0]); }
				if (r != 0) close(idp[q][1]);
			}
			_exit(worker(r, workers, devices[r], n_job, ns, steps, chunks,
				root, idp[r][0], id_out, rp[r][1]));
		}
	}
	for (int r = 0; r < workers; r++) {
		close(rp[r][1]); close(idp[r][0]); close(idp[r][1]);
	}

	struct report rep[MAXW];
	int bad = 0;
	for (int r = 0; r < workers; r++) {
		int st = 0;
		if (read_all(rp[r][0], &rep[r], sizeof rep[r])) bad = 1;
		if (waitpid(pid[r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st)) bad = 1;
	}
	if (bad) {
		fprintf(stderr, "a worker failed\n");
		return 1;
	}
	const double n_total = (double)n_job;
	uint64_t sum = 0;
	float cms = 0.f, gms = 0.f;
	for (int r = 0; r < workers; r++) {
		sum += rep[r].shard_digest;		/* shards add, mod 2^64 */
		if (rep[r].compute_ms > cms) cms = rep[r].compute_ms;
		if (rep[r].gather_ms > gms) gms = rep[r].gather_ms;
		printf("  worker %d on device %d: %8.3f ms per step, with the gather %8.3f\n",
			r, devices[r], rep[r].compute_ms, rep[r].gather_ms);
	}
	printf("%d process(es), %llu samples in all, %d stages, results to worker %d\n",
		workers, (unsigned long long)n_job, ns, root);
	printf("compute only            : %8.3f ms per step, %9.1f Gsample/s\n", cms,
		n_total / (cms * 1e-3) / 1e9);
	printf("compute + RCCL gather   : %8.3f ms per step, %9.1f Gsample/s (%d pieces)\n",
		gms, n_total / (gms * 1e-3) / 1e9, chunks);
	printf("sum of the shard digests: %016llx\n", (unsigned long long)sum);
	printf("digest of the gathered  : %016llx  %s\n",
		(unsigned long long)rep[root].gathered_digest,
		rep[root].gathered_digest == sum ? "(equal)" : "(MISMATCH)");
	return rep[root].gathered_digest == sum ? 0 : 1;
}
