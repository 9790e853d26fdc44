/*
 * cordic_amd.h -- C ABI of the MI355X-native CORDIC rotation engine.
 *
 * Drop-in boundary for the hot path of ZipCPU/cordic.
 * The reference has no FFI layer; the path sits behind three concentric
 * interfaces and each one is mirrored here:
 *
 *  1. the parameter surface of the core generator CLI
 *        gencordic -t p2r|r2p|sp2r|sr2p -i IW -o OW -p PW -n NSTAGES -x XTRA
 *        (reference sw/main.cpp:57-92 usage, :139-232 getopt, :260-357
 *        derivation)                       -> cordic_config_init / _from_args
 *  2. the emitter signature
 *        basiccordic(fp,fhp,cmdline,fname,nstages,iw,ow,nxtra,phase_bits,...)
 *        (reference sw/basiccordic.h:46-50, sw/topolar.h:44-49,
 *        sw/seqcordic.h:46-50, sw/seqpolar.h:44-49)  -> cordic_config_init_core
 *     and the generated constants header (reference rtl/cordic.h:46-59,
 *     rtl/topolar.h:46-59, emitted by sw/basiccordic.cpp:449-505 and
 *     sw/topolar.cpp:412-451)                -> cordic_config / _write_header
 *  3. the per-sample port interface of the generated core
 *        i_xval,i_yval [IW] signed, i_phase [PW] -> o_xval,o_yval [OW] signed
 *        (reference rtl/cordic.v:58-63, driven one sample per tick by
 *        bench/cpp/cordic_tb.cpp:127-178)             -> cordic_p2r[_const]
 *        i_xval,i_yval [IW] signed -> o_mag [OW] signed, o_phase [PW]
 *        (reference rtl/topolar.v:59-64, bench/cpp/topolar_tb.cpp:127-187)
 *                                                     -> cordic_r2p
 *
 *  4. around the path, the same generator's other cores and the benches'
 *     clocking:  -t tbl / -t qtr (sw/sintable.cpp)       -> cordic_table_*
 *                -t qtbl (sw/quadtbl.cpp, rtl/quadtbl.v)  -> cordic_quad_*
 *                i_ce / i_reset / i_aux per clock         -> cordic_stream_*
 *                i_stb / o_busy / o_done per clock        -> cordic_seq_*
 *                16-bit sample arrays                     -> cordic_*16
 *
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.
 * All batch entry points take DEVICE pointers (HBM resident) and a HIP stream
 * handle passed as void* (NULL = the null stream); they enqueue work and
 * return without synchronising.  The *_host variants take host pointers and
 * do the PCIe copies themselves.  The stateless functions are re-entrant and
 * the config is an immutable POD the caller owns; plan / table / quad handles
 * are read-only after creation and may be shared between threads; a stream or
 * seq handle carries state and belongs to one caller at a time.
 *
 * Results are bit-exact to the arithmetic of the Verilog the reference
 * generator emits for the same parameters.
 */
#ifndef CORDIC_AMD_H
#define CORDIC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORDIC_AMD_MAX_STAGES	64
#define CORDIC_AMD_ABI_VERSION	1

/* gencordic -t <type>, reference sw/main.cpp:177-194 */
enum cordic_mode {
	CORDIC_P2R  = 0,	/* -t p2r : polar to rectangular, pipelined   */
	CORDIC_R2P  = 1,	/* -t r2p : rectangular to polar, pipelined   */
	CORDIC_SP2R = 2,	/* -t sp2r: sequential core's arithmetic      */
	CORDIC_SR2P = 3		/* -t sr2p: sequential core's arithmetic      */
};

/* The reference exit(EXIT_FAILURE)s / assert()s on bad parameters
 * (sw/main.cpp:212-231, sw/basiccordic.cpp:69); the ABI returns codes. */
enum cordic_status {
	CORDIC_OK		=  0,
	CORDIC_ERR_MODE		= -1,	/* unknown -t value                    */
	CORDIC_ERR_WIDTH	= -2,	/* IW / OW outside 1..32               */
	CORDIC_ERR_PHASE_BITS	= -3,	/* PW < 3, or > 32 (reference table
					   wraps at 32 bits, cordiclib.cpp:155) */
	CORDIC_ERR_WORKING_WIDTH = -4,	/* WW > 64                             */
	CORDIC_ERR_STAGES	= -5,	/* NSTAGES outside 1..64               */
	CORDIC_ERR_UNSUPPORTED	= -6,	/* parameters for which the reference
					   emits a core that cannot elaborate
					   or never raises o_done              */
	CORDIC_ERR_ARGS		= -7,	/* NULL pointer / bad command line     */
	CORDIC_ERR_DEVICE	= -8,	/* HIP runtime error (no GPU, launch)  */
	CORDIC_ERR_CONTAINER	= -9,	/* a port is wider than the 16-bit
					   sample container of a *16 call      */
	CORDIC_ERR_NOMEM	= -10	/* host allocation failed              */
};

/* flags (cordic_config.flags) -- implementation selectors for A/B work */
#define CORDIC_FLAG_FORCE_GENERIC	0x1u	/* never use an unrolled kernel */
#define CORDIC_FLAG_NO_LJ		0x4u	/* rotators with WW <= 35: the
						   right-justified kernels (32-
						   bit container for WW <= 32,
						   64-bit above) instead of the
						   left-justified ones (for A/B) */
#define CORDIC_FLAG_UNIT_GAIN		0x10u	/* fuse the gain-annihilation
						   multiply into the output:
						   o = (o * K) >> 32, K =
						   cordic_config_gain_annihilator */
#define CORDIC_FLAG_NO_SEED		0x8u	/* plans: full recurrence, no
						   seed table (for A/B)         */
#define CORDIC_FLAG_NO_TAILS		0x40u	/* plans: seed table only; the
						   stages behind it run the phase
						   recurrence instead of looking
						   their directions up (for A/B) */
#define CORDIC_FLAG_STATIC_CHUNKS	0x20u	/* plans: one contiguous chunk
						   per persistent block instead
						   of the address-ordered tile
						   queue (for A/B; since round 5
						   on the dynamic-exit instance,
						   the only one that still
						   carries that sweep)          */

/*
 * One generated core.  The first block mirrors, field for field, the
 * constants header the reference writes with -c (rtl/cordic.h:46-59).
 */
typedef struct cordic_config {
	int32_t	mode;			/* enum cordic_mode                   */
	int32_t	iw;			/* IW                                 */
	int32_t	ow;			/* OW                                 */
	int32_t	nextra;			/* NEXTRA (after the CLI's +1 / +2)   */
	int32_t	ww;			/* WW                                 */
	int32_t	pw;			/* PW                                 */
	int32_t	nstages;		/* NSTAGES                            */
	int32_t	clocks_per_output;	/* CLOCKS_PER_OUTPUT (seq cores) or 0 */
	double	quantization_variance;	/* QUANTIZATION_VARIANCE              */
	double	phase_variance_rad;	/* PHASE_VARIANCE_RAD                 */
	double	gain;			/* GAIN                               */
	double	best_possible_cnr;	/* BEST_POSSIBLE_CNR (p2r, sp2r)      */
	int32_t	has_reset;		/* HAS_RESET  (-r / -R)               */
	int32_t	has_aux;		/* HAS_AUX    (-a)                    */
	int32_t	async_reset;		/* ASYNC_RESET (-A)                   */
	/* derived for the device path */
	int32_t	nlive;			/* rotations the core really performs */
	int32_t	needs_wrap;		/* WW-bit overflow reachable: kernels
					   must wrap explicitly               */
	uint32_t flags;
	uint32_t angle[CORDIC_AMD_MAX_STAGES]; /* cordic_angle[], PW-bit      */
} cordic_config;

/* ------------------------------------------------------------------ host */

int	cordic_abi_version(void);
const char *cordic_strerror(int status);

/* CLI level (sw/main.cpp:260-357): xtra is the -x value (reference default
 * 2); iw/ow <= 0 follow the reference's defaulting (:262-270); phase_bits
 * <= 0 and nstages <= 0 are derived with calc_phase_bits / calc_stages. */
int	cordic_config_init(cordic_config *cfg, int mode, int iw, int ow,
		int xtra, int phase_bits, int nstages);

/* Emitter level (sw/basiccordic.h:46-50): nxtra is the already incremented
 * value the emitters receive. */
int	cordic_config_init_core(cordic_config *cfg, int mode, int nstages,
		int iw, int ow, int nxtra, int phase_bits);

/* gencordic-compatible argv ("aAcf:hi:n:o:p:Rrt:vx:", sw/main.cpp:139).
 * argv[0] is the program name.  fname (may be NULL) receives the -f value or
 * the reference's default file name; *c_header receives the -c flag.
 * Table generators (-t tbl/qtr/qtbl) return CORDIC_ERR_MODE. */
int	cordic_config_from_args(cordic_config *cfg, int argc,
		const char *const *argv, char *fname, size_t fname_cap,
		int *c_header);

/* Text of the constants header between "#ifndef <GUARD>" and "#endif"
 * exactly as the reference writes it for `name` (e.g. "cordic" ->
 * CORDIC_H; sw/basiccordic.cpp:449-505, sw/topolar.cpp:412-451,
 * sw/seqcordic.cpp:446-500, sw/seqpolar.cpp:383-420), without the licence
 * banner.  Returns the length (excluding NUL) or a negative status; writes
 * at most cap bytes. */
int	cordic_config_write_header(const cordic_config *cfg, const char *name,
		char *buf, size_t cap);

/* Library functions of sw/cordiclib.h:45-52, exported for callers that used
 * them directly. */
int	cordic_nextlg(unsigned vl);
double	cordic_gain(int nstages);
/* "You can annihilate this gain by multiplying by 32'h%08x and right shifting
 * by 32 bits" (sw/cordiclib.cpp:205-209): that constant for a stage count, and
 * as the generator prints it into the given core (the sequential cores' tables
 * are padded to a power of two first, sw/cordiclib.cpp:145-149).  With
 * CORDIC_FLAG_UNIT_GAIN set in cfg.flags every kernel applies it to o_xval /
 * o_yval / o_mag before the store: o = (int64(o) * K) >> 32. */
uint32_t cordic_gain_annihilator(int nstages);
uint32_t cordic_config_gain_annihilator(const cordic_config *cfg);
double	cordic_phase_variance(int nstages, int phase_bits);
double	cordic_transform_quantization_variance(int nstages, int xtrabits,
		int dropped_bits);
int	cordic_angles(int nstages, int phase_bits, uint32_t *out);
int	cordic_calc_stages_ww(int working_width, int phase_bits);
int	cordic_calc_stages(int phase_bits);
int	cordic_calc_phase_bits(int output_width);

/* ---------------------------------------------------------------- device */

/* Polar to rectangular (sin/cos, vector rotation); cfg->mode P2R or SP2R.
 * Inputs are taken modulo their port width (IW / PW low bits), outputs are
 * sign extended OW-bit values.  Replaces one Vcordic tick() per sample
 * (bench/cpp/cordic_tb.cpp:136-176). */
int	cordic_p2r(const cordic_config *cfg, size_t n,
		const int32_t *d_xval, const int32_t *d_yval,
		const uint32_t *d_phase,
		int32_t *d_oxval, int32_t *d_oyval, void *stream);

/* Same with i_xval / i_yval held constant, the way the reference bench
 * drives the core (cordic_tb.cpp:68-69): 4 B in + 8 B out per sample. */
int	cordic_p2r_const(const cordic_config *cfg, size_t n,
		int32_t xval, int32_t yval, const uint32_t *d_phase,
		int32_t *d_oxval, int32_t *d_oyval, void *stream);

/* Fused NCO: phase[i] = phase0 + (index0 + i) * fcw  (mod 2^PW) generated in
 * the kernel, then the p2r core: 0 B in + 8 B out per sample. */
int	cordic_nco(const cordic_config *cfg, size_t n,
		uint32_t phase0, uint32_t fcw, uint64_t index0,
		int32_t xval, int32_t yval,
		int32_t *d_oxval, int32_t *d_oyval, void *stream);

/* Fused NCO MIXER (down-converter): the core with all three ports live
 * (rtl/cordic.v:58-63) -- per-sample i_xval / i_yval from memory, i_phase from
 * the accumulator phase[i] = phase0 + (index0 + i) * fcw (mod 2^PW, the ramp of
 * bench/cpp/cordic_tb.cpp:128-138 with an arbitrary increment) generated in the
 * kernel: 8 B in + 8 B out per sample instead of the 12 + 8 of cordic_p2r on
 * a materialised phase array.  cordic_plan_mix: the same through a plan, whose
 * direction tables serve it exactly as they serve cordic_plan_p2r. */
int	cordic_mix(const cordic_config *cfg, size_t n,
		uint32_t phase0, uint32_t fcw, uint64_t index0,
		const int32_t *d_xval, const int32_t *d_yval,
		int32_t *d_oxval, int32_t *d_oyval, void *stream);

/* Rectangular to polar (magnitude + atan2); cfg->mode R2P or SR2P.
 * d_ophase receives the raw PW-bit phase (rtl/topolar.v:269).  Replaces one
 * Vtopolar tick() per sample (bench/cpp/topolar_tb.cpp:143-187). */
int	cordic_r2p(const cordic_config *cfg, size_t n,
		const int32_t *d_xval, const int32_t *d_yval,
		int32_t *d_omag, uint32_t *d_ophase, void *stream);

/* Diagnostic: the kernel family that served the calling thread's most recent
 * p2r / nco / r2p launch.  Every family computes the same bits; they differ in
 * speed (seeded > left-justified / unrolled > generic), and a caller -- or a
 * test -- that expects the fast path can check that it got it. */
enum cordic_kernel_family {
	CORDIC_KERNEL_NONE		= 0,
	CORDIC_KERNEL_GENERIC		= 1,	/* run-time stage loop            */
	CORDIC_KERNEL_UNROLLED		= 2,	/* unrolled, one lane = 4 samples */
	CORDIC_KERNEL_SEEDED		= 3,	/* plan: seed table + unrolled    */
	CORDIC_KERNEL_LEFT_JUSTIFIED	= 4,	/* r2p: topolar_lj / topolar_ljw  */
	CORDIC_KERNEL_DIRECTIONS	= 5	/* plan, per-sample vectors: stage
						   directions looked up            */
};
int	cordic_last_kernel(void);

/*
 * Plans: a generated core bound to the current HIP device.
 *
 * gencordic generates a core once and the bench then streams samples through
 * it; cordic_plan_create is that generation step for the GPU.  For rotators it
 * uploads a small "seed table" (cordic_seed_table) that lets the constant-
 * vector entry points -- the sin/cos generator use of the core, i_xval/i_yval
 * fixed as in bench/cpp/cordic_tb.cpp:68-69 -- replace the first 11
 * micro-rotations by an exact table lookup: the (x, y) state after 11 stages
 * depends only on the octant and on the 11 rotation directions, which are a
 * monotone step function of the phase with integer break points.  The (x, y)
 * table itself is computed ON THE DEVICE with the exact recurrence, by the
 * kernel's own prologue, so results are bit-identical for every phase.
 *
 * Since round 5 the plan KEEPS that prologue's result -- the block's LDS image:
 * seeds of every (octant, leaf), buckets, tail tables; 136 KiB -- per constant
 * vector (xval, yval): the first launch with a new vector runs the kernel once
 * in build mode (one block, ~15 us) on the caller's stream, every later launch
 * with that vector copies the image instead of recomputing it (~10 us of every
 * launch before).  Image slots are write-once (cordic_plan_image_info: up to 8
 * vectors per plan; later ones compute in the kernel as before), launches on
 * other streams are ordered behind the build on the device, and a launch that
 * is being CAPTURED into a HIP graph only uses an image the host already
 * knows to be complete -- call cordic_plan_prepare(plan, xval, yval, stream)
 * before capturing to get it (a captured launch without one computes its own
 * prologue, same bits).  CORDIC_SEED_IMAGES=0 in the
 * environment of cordic_plan_create turns the cache off (A/B).
 *
 * Cores that are not eligible (r2p, WW > 35, fewer than 11 live stages) simply
 * run the ordinary kernels -- and so do SMALL batches: below 2^22 samples (2^23
 * / 6 Mi where no image serves the launch) a plan launches the full recurrence,
 * which has no table to stage and is then the faster kernel
 * (profiles/r05/small_batch.txt).  cordic_plan_set_min_samples moves that size
 * for one plan (0: always the tables; < 0: back to the default);
 * CORDIC_SEED_MIN_SAMPLES in the environment is the process-wide default for
 * plans that have not been told (the test suite sets 0).  Same results either
 * way.  Many small jobs in ONE launch: cordic_jobset below.
 *
 * Tile queues and HIP graphs.  A plan (likewise a table / quad handle) owns a
 * ring of tile-queue counter blocks; every launch takes one and the ring hands
 * a block out again only behind the launch that used it (on the device: no
 * host wait).  A launch issued while its stream is being CAPTURED keeps its
 * block for the life of the handle, because the graph may be replayed at any
 * time: a handle serves at most `captured_capacity` (208) captured launches;
 * later ones still compute the same bits but sweep static chunks (-5...-8 %)
 * and are counted in `fallback_launches`.  Re-capturing per shape or per
 * epoch therefore wants a fresh handle now and then.  Two graph execs
 * instantiated from the SAME captured graph share the captured node's block:
 * do not run them concurrently (one exec replayed any number of times, or
 * execs of separately captured graphs, are fine).
 */
typedef struct cordic_plan cordic_plan;

typedef struct cordic_queue_info {
	int32_t	eager_slots;		/* blocks rotated among eager launches  */
	int32_t	captured_capacity;	/* blocks captured launches may take    */
	int32_t	captured_used;		/* ... and how many they have taken     */
	uint64_t fallback_launches;	/* launches that got no block: static
					 * sweep (ring exhausted by captures, or
					 * every eager block claimed right now) */
} cordic_queue_info;

int	cordic_plan_create(const cordic_config *cfg, cordic_plan **plan);
/* Build the seed image of (xval, yval) now, on `stream` (see above), and
 * return when it is complete (a set-up call: it waits ~15 us for one block);
 * not inside a stream capture.  CORDIC_ERR_UNSUPPORTED for cores without a
 * seed table or when all image slots are taken.  The int16 containers
 * (cordic_plan_p2r16_const / cordic_plan_nco16) keep an image of their own:
 * it is built here too once the plan has served an int16 call (a plan that
 * never does keeps all eight slots for the 32-bit arrays). */
int	cordic_plan_prepare(const cordic_plan *plan, int32_t xval, int32_t yval,
		void *stream);
/* images held, launches served from one, launches that computed their own
 * prologue (any pointer may be NULL) */
int	cordic_plan_image_info(const cordic_plan *plan, int32_t *held,
		uint64_t *hits, uint64_t *misses);
/* batch size (samples) from which this plan's table-driven kernels serve a
 * call; < 0: the library's default */
int	cordic_plan_set_min_samples(cordic_plan *plan, long long min_samples);
int	cordic_plan_queue_info(const cordic_plan *plan, cordic_queue_info *info);
void	cordic_plan_destroy(cordic_plan *plan);
const cordic_config *cordic_plan_config(const cordic_plan *plan);
/* stages covered by the seed table (0 = none), its leaves and buckets */
int	cordic_plan_seed_info(const cordic_plan *plan, int32_t *stages,
		int32_t *nleaves, int32_t *nbuckets);
/* The direction tails behind the seed table: the number of stage groups whose
 * multipliers are looked up (0: none) and, in stages[0..3], their sizes; the
 * stages behind the last group run the phase recurrence.  (Plans created with
 * CORDIC_FLAG_NO_TAILS still report the table; the launch ignores it.)         */
int	cordic_plan_tail_info(const cordic_plan *plan, int32_t *ngroups,
		int32_t stages[4]);

/* cordic_p2r through a plan: per-sample i_xval / i_yval / i_phase.  The
 * rotation directions depend on the phase alone (rtl/cordic.v:262-280), so the
 * plan's direction tables serve this feed too: every stage behind the first
 * reads its multipliers instead of running the phase recurrence (4 instead of
 * 7 instructions per micro-rotation; the vector state itself cannot be
 * tabulated, every stage still runs).  Same results as cordic_p2r, bit for
 * bit; cores without such a table (WW > 35, reachable overflow, stage counts
 * without an instance) and batches below 2^23 samples (see above:
 * CORDIC_SEED_MIN_SAMPLES) run cordic_p2r's kernel.  cordic_plan_dir_info: the
 * number of looked-up stage groups (0: none) and their sizes. */
int	cordic_plan_p2r(const cordic_plan *plan, size_t n,
		const int32_t *d_xval, const int32_t *d_yval,
		const uint32_t *d_phase, int32_t *d_oxval, int32_t *d_oyval,
		void *stream);
int	cordic_plan_dir_info(const cordic_plan *plan, int32_t *ngroups,
		int32_t stages[5]);
int	cordic_plan_mix(const cordic_plan *plan, size_t n,
		uint32_t phase0, uint32_t fcw, uint64_t index0,
		const int32_t *d_xval, const int32_t *d_yval,
		int32_t *d_oxval, int32_t *d_oyval, void *stream);

int	cordic_plan_p2r_const(const cordic_plan *plan, size_t n,
		int32_t xval, int32_t yval, const uint32_t *d_phase,
		int32_t *d_oxval, int32_t *d_oyval, void *stream);
int	cordic_plan_nco(const cordic_plan *plan, size_t n,
		uint32_t phase0, uint32_t fcw, uint64_t index0,
		int32_t xval, int32_t yval,
		int32_t *d_oxval, int32_t *d_oyval, void *stream);

/* ------------------------------------------------------------ job sets
 *
 * Many SMALL jobs in one launch (round 5; data-fed kinds round 6).  A launch costs ~10-20 us whatever
 * it computes -- the runtime's dispatch, staging the seed table, filling and
 * draining 256 CUs -- which is the whole run time of a 2^16-sample job: an NCO
 * bank or a channeliser that hands the engine a thousand short blocks gets a
 * few per cent of the rate of one long one (profiles/r05/small_batch.txt).  A
 * job set is those blocks described once: the host cuts them into the seeded
 * kernel's tiles (8192 samples, never across a job's end), keeps the table on
 * the device, and cordic_plan_run_jobs then runs the WHOLE set as one launch of
 * the same kernel walking the same address-ordered tile queue (+ one small
 * launch for the up to three samples behind each job's last whole vector).
 * Jobs may have any length (ragged, zero included) and any 4-byte aligned
 * addresses; they must not overlap each other's outputs.  The set can be run
 * any number of times (new data in the same arrays: a channeliser's steady
 * state), on any stream, with any constant vector, and inside a HIP graph
 * capture; cordic_plan_*_batch are the one-shot forms (table built, uploaded
 * with a blocking copy, run, freed once the launch has passed: they cost a
 * host-side ~50 us per call more, and -- because they allocate and copy -- are
 * NOT legal while `stream` is being captured: CORDIC_ERR_UNSUPPORTED; capture
 * cordic_plan_run_jobs on a set made beforehand instead).  A set belongs to the
 * plan's core and to the device that was current when it was cut
 * (CORDIC_ERR_ARGS from cordic_plan_run_jobs otherwise).
 *
 * kind says where a job's inputs come from:
 *   CORDIC_JOBS_PHASE_ARRAYS  cordic_plan_p2r_const per job: d_phase
 *   CORDIC_JOBS_NCO           cordic_plan_nco per job: phase0, fcw, index0
 * and, since round 6, the DATA-FED calls (what a channeliser hands over:
 * blocks of I/Q samples; xval / yval of cordic_plan_run_jobs are ignored):
 *   CORDIC_JOBS_R2P           cordic_r2p per job (rtl/topolar.v:59-64, driven
 *                             per sample at bench/cpp/topolar_tb.cpp:127-147):
 *                             d_xval, d_yval -> d_oxval = o_mag, d_oyval =
 *                             o_phase (as uint32_t); the plan is one of an r2p /
 *                             sr2p core
 *   CORDIC_JOBS_P2R_XY        cordic_plan_p2r per job (rtl/cordic.v:58-63 with
 *                             all three ports live): d_xval, d_yval, d_phase
 *   CORDIC_JOBS_MIX           cordic_plan_mix per job: d_xval, d_yval rotated by
 *                             phase0 + (index0 + i) * fcw
 * These are cut into tiles of 256 .. 2048 whole vectors (shorter tiles for
 * small sets, so that every CU gets several blocks) and run as ONE launch of
 * the tile-reading instance of the call's own kernel (topolar_lj, rotator_xydir;
 * + one small launch for trailing samples).  Tile-reading instances exist for
 * the left-justified converter (WW <= 34, no reachable overflow, no unit gain:
 * static for 20 and 29 stages, dynamic-exit otherwise) and for the looked-up-
 * direction rotator at WW 35 with 16 / 24 / 29 stages and WW <= 34 with 16 / 19
 * / 27; every other core runs its jobs one by one behind the same call.
 * Results: bit for bit those of the per-job calls.
 * Cores without a table-seeded kernel (WW > 35, fewer than 11 live stages,
 * CORDIC_FLAG_NO_SEED) run constant-vector jobs one by one behind the same call.
 */
typedef struct cordic_job {
	const uint32_t *d_phase;	/* PHASE_ARRAYS, P2R_XY: n words        */
	uint32_t phase0, fcw;		/* NCO, MIX: phase0 +                   */
	uint64_t index0;		/*   (index0 + i) * fcw  (mod 2^PW)     */
	int32_t	*d_oxval, *d_oyval;	/* n words each (R2P: o_mag, o_phase)   */
	uint64_t n;			/* samples                              */
	const int32_t *d_xval, *d_yval;	/* R2P, P2R_XY, MIX: n words each       */
} cordic_job;
enum cordic_jobs_kind {
	CORDIC_JOBS_PHASE_ARRAYS = 0, CORDIC_JOBS_NCO = 1,
	CORDIC_JOBS_R2P = 2, CORDIC_JOBS_P2R_XY = 3, CORDIC_JOBS_MIX = 4
};
typedef struct cordic_jobset cordic_jobset;
int	cordic_jobset_create(const cordic_plan *plan, int kind, size_t njobs,
		const cordic_job *jobs, cordic_jobset **set);
void	cordic_jobset_destroy(cordic_jobset *set);
/* samples of all jobs, tiles the kernel walks, samples served by the
 * trailing-sample launch (any pointer may be NULL) */
int	cordic_jobset_info(const cordic_jobset *set, uint64_t *samples,
		uint32_t *tiles, uint32_t *tail_samples);
int	cordic_plan_run_jobs(const cordic_plan *plan, const cordic_jobset *set,
		int32_t xval, int32_t yval, void *stream);
int	cordic_plan_p2r_const_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, int32_t xval, int32_t yval, void *stream);
int	cordic_plan_nco_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, int32_t xval, int32_t yval, void *stream);
/* the one-shot forms of the data-fed kinds */
int	cordic_plan_r2p_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, void *stream);
int	cordic_plan_p2r_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, void *stream);
int	cordic_plan_mix_batch(const cordic_plan *plan, size_t njobs,
		const cordic_job *jobs, void *stream);
/* waits for and frees what the one-shot forms still hold (they free it
 * themselves, lazily, on later calls) */
void	cordic_jobset_reap(void);

/* ---------------------------------------------- 16-bit sample containers
 *
 * The reference's 16-bit benches keep their samples in shorts
 * (int16/uint16 host containers for -i 16 -o 16 -p 16).
 * These entry points are the calls above on int16_t / uint16_t arrays: same
 * arithmetic, same results (each value is the low 16 bits of what the 32-bit
 * call returns, which is the whole value because the ports fit), half the
 * memory traffic.  They require IW <= 16 and OW <= 16, and PW <= 16 wherever
 * a phase ARRAY is read or written (the NCO forms take PW-bit scalars, any
 * PW); otherwise CORDIC_ERR_CONTAINER.  Arrays should be 8-byte aligned for
 * the vector path. */
int	cordic_p2r16(const cordic_config *cfg, size_t n,
		const int16_t *d_xval, const int16_t *d_yval,
		const uint16_t *d_phase,
		int16_t *d_oxval, int16_t *d_oyval, void *stream);
int	cordic_p2r16_const(const cordic_config *cfg, size_t n,
		int32_t xval, int32_t yval, const uint16_t *d_phase,
		int16_t *d_oxval, int16_t *d_oyval, void *stream);
int	cordic_nco16(const cordic_config *cfg, size_t n,
		uint32_t phase0, uint32_t fcw, uint64_t index0,
		int32_t xval, int32_t yval,
		int16_t *d_oxval, int16_t *d_oyval, void *stream);
int	cordic_r2p16(const cordic_config *cfg, size_t n,
		const int16_t *d_xval, const int16_t *d_yval,
		int16_t *d_omag, uint16_t *d_ophase, void *stream);
int	cordic_plan_p2r16_const(const cordic_plan *plan, size_t n,
		int32_t xval, int32_t yval, const uint16_t *d_phase,
		int16_t *d_oxval, int16_t *d_oyval, void *stream);
int	cordic_plan_nco16(const cordic_plan *plan, size_t n,
		uint32_t phase0, uint32_t fcw, uint64_t index0,
		int32_t xval, int32_t yval,
		int16_t *d_oxval, int16_t *d_oyval, void *stream);

/* Host only: the phase-side seed table of a core as 32-bit words
 *   [0] stages M  [1] bucket shift S  [2] nbuckets  [3] nleaves
 *   nbuckets x {bound-1, first_leaf}   (r = phase + 2^29 domain; at most one
 *            leaf boundary per bucket, 0x7fffffff where there is none)
 *   nleaves  x {direction pattern, offset + 2^29}
 * on the left-justified phase (phase << (32-PW)) after the octant fold,
 * followed -- when the core has them and cap_words leaves room -- by the
 * direction tails behind the seeds (cordic_plan_tail_info):
 *   [0] groups  [1] bias of the first group  [2] bias behind the last  [3] 0,
 *   then per group {stages, bucket shift, nbuckets, nleaves, 0, 0},
 *   nbuckets x {bound-1, first_leaf}, nleaves x {direction pattern, offset}.
 * Returns the number of words written (call with buf = NULL, cap_words = 0 for
 * the full size), or 0 if the core is not eligible or cap_words does not hold
 * even the seed part; a cap between the two returns the SEED PART ONLY -- size
 * the buffer with the query call to get the tails. */
size_t	cordic_seed_table(const cordic_config *cfg, uint32_t *buf, size_t cap_words);
/* Host only: the direction tables of the per-sample-vector path
 * (cordic_plan_p2r) as 32-bit words
 *   [0] groups  [1] bias0  [2] bias behind the last group  [3] 0, then per group
 *   {stages, bucket shift, nbuckets, nleaves, 0, 0},
 *   nbuckets x {bound-1, first_leaf}, nleaves x {direction pattern, offset}:
 * with p1 the residual phase behind the first micro-rotation (left-justified),
 * u = p1 + bias0 indexes group 0; the leaf of u holds the directions of the
 * group's stages (first stage = MSB, 1 = residual >= 0) and u - offset indexes
 * the next group.  Returns the number of words (buf = NULL, cap_words = 0: the
 * size), 0 if the core has no such table or cap_words is too small. */
size_t	cordic_dir_table(const cordic_config *cfg, uint32_t *buf, size_t cap_words);

/*
 * Table cores (row F4): the reference's plain and quarter-wave sine tables,
 * gencordic -t tbl / -t qtr (sw/sintable.cpp; rtl/sintable.v:72-77,
 * rtl/quarterwav.v:86-108).  Not CORDIC -- a gather -- offered so that the
 * reference's other sine generators can be compared on the same device.
 */
enum cordic_table_kind {
	CORDIC_TBL = 4,		/* -t tbl : 2^PW-entry sine table            */
	CORDIC_QTR = 5		/* -t qtr : 2^(PW-2)-entry quarter-wave table */
};

typedef struct cordic_table_config {
	int32_t	kind;		/* enum cordic_table_kind                    */
	int32_t	pw;		/* PW: phase bits                            */
	int32_t	ow;		/* OW: output bits                           */
	int32_t	entries;	/* table length                              */
} cordic_table_config;

typedef struct cordic_table cordic_table;	/* table resident on the device */

/* gencordic's defaulting for -t tbl / -t qtr (sw/main.cpp:330-405): iw is the
 * -i value (taken as the phase width when -p is absent), <= 0 / < 0 = absent. */
int	cordic_table_config_init(cordic_table_config *cfg, int kind, int iw,
		int ow, int phase_bits);
/* The table exactly as the generator writes it to <name>.hex
 * (sw/sintable.cpp:155-166,322-333), as sign-extended OW-bit values. */
int	cordic_table_values(const cordic_table_config *cfg, int32_t *out,
		size_t cap);
int	cordic_table_create(const cordic_table_config *cfg, cordic_table **tbl);
void	cordic_table_destroy(cordic_table *tbl);
int	cordic_table_queue_info(const cordic_table *tbl, cordic_queue_info *info);
/* d_val[i] = o_val of the core for i_phase = d_phase[i] (low PW bits) */
int	cordic_table_lookup(const cordic_table *tbl, size_t n,
		const uint32_t *d_phase, int32_t *d_val, void *stream);
/* which kernel serves this table: 0 = gather from the table in L2, 1 = packed
 * int16 copy of a quarter-wave table in LDS, 2 = full-wave table folded to its
 * first quadrant in LDS (OW <= 16, PW <= 17, and -- for 2 -- the generated
 * table verified to have the symmetry); 3 / 4 = the same two layouts for
 * OW > 16 with the 32-bit entries themselves in LDS (PW <= 17: 2^15 entries
 * = 128 KiB, one block per CU).  A 2^16-entry quadrant (PW 18) does not fit
 * the 160 KiB of a CU at 24 bits and stays on the L2 gather. */
int	cordic_table_lds_mode(const cordic_table *tbl);

/* ---------------------------------- quadratically interpolated sine core
 *
 * gencordic -t qtbl (sw/quadtbl.cpp, rtl/quadtbl.v): three 2^LGTBL-entry
 * coefficient tables C, L, Q indexed by the top LGTBL phase bits, and
 *   o_sin = round((((Q*dx >> (DXBITS-1)) + L) * dx >> (DXBITS-1)) + C)
 * on the remaining phase bits dx, in the exact bit widths of rtl/quadtbl.v
 * :149-153 (lookup), :170 (qprod), :214-221 (lsum), :246 (lprod), :270-277
 * (r_value), :292-300 (convergent rounding with the two no-overflow cases),
 * :308 (o_sin).
 */
typedef struct cordic_quad_config {
	int32_t	pw;		/* PW                                         */
	int32_t	ow;		/* OW                                         */
	int32_t	xtra;		/* XTRA of the emitted core = max(nxtra, 2)   */
	int32_t	tbl_width;	/* OW + nxtra: width the tables are built for */
	int32_t	ww;		/* WW = OW + XTRA                             */
	int32_t	lgtbl;		/* LGTBL                                      */
	int32_t	entries;	/* TBLENTRIES = 2^LGTBL                       */
	int32_t	dxbits;		/* DXBITS = PW - LGTBL + 1                    */
	int32_t	cbits, lbits, qbits;	/* CBITS, LBITS, QBITS                */
	int32_t	has_reset, has_aux;
	int64_t	scale;		/* SCALE    (generated header)                */
	double	itbl_err;	/* ITBL_ERR, full precision                   */
	double	tbl_err;	/* TBL_ERR                                    */
	double	spur_db;	/* SPURDB                                     */
} cordic_quad_config;

typedef struct cordic_quad cordic_quad;	/* tables resident on the device */

/* gencordic -t qtbl -i iw -o ow -x xtra -p phase_bits (sw/main.cpp:444-463);
 * <= 0 = absent.  The table size grows from 16 entries until the fit error
 * is within one LSB of the working width (sw/quadtbl.cpp:306-310). */
int	cordic_quad_config_init(cordic_quad_config *cfg, int iw, int ow,
		int xtra, int phase_bits);
/* the emitter's own tuple: quadtbl(fp, fhp, cmdline, fname, phase_bits, ow,
 * nxtra, ...) (sw/quadtbl.h:47-49), nxtra already incremented */
int	cordic_quad_config_init_core(cordic_quad_config *cfg, int phase_bits,
		int ow, int nxtra);
/* The three tables as the generator writes them to <name>_ctbl.hex,
 * <name>_ltbl.hex, <name>_qtbl.hex (sw/quadtbl.cpp:243-259), as sign-extended
 * values; each array has room for `cap` >= entries values. */
int	cordic_quad_tables(const cordic_quad_config *cfg, int32_t *ctbl,
		int32_t *ltbl, int32_t *qtbl, size_t cap);
/* constants header of the core (sw/quadtbl.cpp:771-811), as
 * cordic_config_write_header */
int	cordic_quad_write_header(const cordic_quad_config *cfg, const char *name,
		char *buf, size_t cap);
int	cordic_quad_create(const cordic_quad_config *cfg, cordic_quad **core);
void	cordic_quad_destroy(cordic_quad *core);
int	cordic_quad_queue_info(const cordic_quad *core, cordic_queue_info *info);
/* d_sin[i] = o_sin of the core for i_phase = d_phase[i] (low PW bits),
 * sign-extended OW-bit values */
int	cordic_quad_lookup(const cordic_quad *core, size_t n,
		const uint32_t *d_phase, int32_t *d_sin, void *stream);

/* ------------------------------------------- clocked view (streaming shim)
 *
 * For benches that step the Verilated PIPELINED cores clock by clock with
 * i_ce / i_reset / i_aux (bench/cpp/testb.h:87-106, cordic_tb.cpp:136-176):
 * a cordic_stream is the core with its pipeline contents; cordic_stream_ticks
 * applies T consecutive clocks, given as arrays with one entry per clock, and
 * returns what the output ports show after each of them -- the latency of
 * NSTAGES+2 enabled clocks, the hold on i_ce = 0, the clearing on i_reset
 * and the i_aux -> o_aux delay line (rtl/cordic.v:100-105,118-124,244-252,
 * 304-313; rtl/topolar.v likewise) are reproduced exactly.  State carries
 * over from call to call.  Modes: CORDIC_P2R, CORDIC_R2P (the sequential
 * cores have no i_ce pipeline: see cordic_seq_* below).
 *
 *   d_ce, d_reset, d_aux : one byte per clock (non-zero = asserted); NULL
 *                          means i_ce = 1 / i_reset = 0 / i_aux = 0 throughout.
 *                          d_reset is i_reset, or !i_areset_n for -A cores.
 *   p2r: d_xval, d_yval, d_phase -> d_out0 = o_xval, d_out1 = o_yval
 *   r2p: d_xval, d_yval (d_phase NULL) -> d_out0 = o_mag, d_out1 = o_phase
 *   d_oaux : o_aux per clock (may be NULL)
 * A freshly created stream is in the reset state.  The call needs
 * cordic_stream_workspace(T) bytes of device scratch; cordic_stream_reserve
 * allocates it up front (synchronising the device), otherwise a call that
 * needs more grows it in stream order on its own stream (hipMallocAsync: no
 * device-wide stall, but not legal inside a stream capture -- reserve first,
 * then capture).  cordic_stream_reserve / cordic_seq_reserve themselves
 * hipDeviceSynchronize and hipFree when they grow the scratch: call them at
 * set-up time, not between launches that should overlap.  Once reserved a call only enqueues kernels, and
 * the pipeline state sits in one set of device buffers that the kernels update
 * in place, so a HIP graph captured around it can be replayed block after
 * block (the same holds for cordic_seq_ticks).
 */
typedef struct cordic_stream cordic_stream;
int	cordic_stream_create(const cordic_config *cfg, cordic_stream **s);
void	cordic_stream_destroy(cordic_stream *s);
size_t	cordic_stream_workspace(size_t ticks);
int	cordic_stream_reserve(cordic_stream *s, size_t max_ticks);
/* latency in enabled clocks from i_* to o_* (NSTAGES + 2) */
int	cordic_stream_latency(const cordic_stream *s);
int	cordic_stream_reset(cordic_stream *s, void *stream);
int	cordic_stream_ticks(cordic_stream *s, size_t ticks,
		const uint8_t *d_ce, const uint8_t *d_reset, const uint8_t *d_aux,
		const int32_t *d_xval, const int32_t *d_yval,
		const uint32_t *d_phase,
		int32_t *d_out0, int32_t *d_out1, uint8_t *d_oaux, void *stream);

/* The SEQUENTIAL cores' handshake (rtl/seqcordic.v:226-327,
 * rtl/seqpolar.v:211-307; bench/cpp/cordic_tb.cpp:146-159) in the same
 * block-of-clocks form.  With C = cfg.clocks_per_output: a sample is taken on
 * a clock with i_stb while the core is idle; o_busy then reads 1 for C-1
 * clocks; on the clock C-1 after the accept o_done reads 1 (for that clock
 * only) and o_xval/o_yval (o_mag/o_phase) and o_aux load; i_reset drops a
 * sample in flight and clears o_done, the output registers keep their values.
 * i_stb while busy is ignored, as in the RTL -- except on the very clock that
 * completes a sample: there the RTL gives i_stb precedence over the return to
 * idle but loads nothing (pre_valid needs idle), so its free-running datapath
 * goes round again over its own unrounded result and a second o_done appears
 * C-1 clocks later (rtl/seqcordic.v:229-246,270-291).  That is reproduced too,
 * bit for bit, for any strobe pattern (a bench that simply holds i_stb high
 * included): blocks that contain such a strobe are re-done by a register-level
 * pass (one thread stepping the core's register file: exact, ~10^7 clocks/s
 * instead of ~10^10), and the register file carries over when a re-run spans
 * calls.  cordic_seq_violations counts those strobes.
 *   d_stb : one byte per clock (required); d_reset, d_aux, d_busy, d_done,
 *   d_oaux may be NULL.  p2r: d_phase required, outputs o_xval / o_yval;
 *   r2p: d_phase NULL, outputs o_mag / o_phase.  Modes CORDIC_SP2R, CORDIC_SR2P.
 */
typedef struct cordic_seq cordic_seq;
int	cordic_seq_create(const cordic_config *cfg, cordic_seq **s);
void	cordic_seq_destroy(cordic_seq *s);
size_t	cordic_seq_workspace(size_t ticks);
int	cordic_seq_reserve(cordic_seq *s, size_t max_ticks);
int	cordic_seq_ticks(cordic_seq *s, size_t ticks,
		const uint8_t *d_stb, const uint8_t *d_reset, const uint8_t *d_aux,
		const int32_t *d_xval, const int32_t *d_yval,
		const uint32_t *d_phase,
		int32_t *d_out0, int32_t *d_out1,
		uint8_t *d_busy, uint8_t *d_done, uint8_t *d_oaux, void *stream);
/* i_stb on completing clocks (each one a re-run of the datapath) seen so far
 * (synchronises the device) */
int	cordic_seq_violations(cordic_seq *s, uint64_t *count);

/* ---------------------------------------------------------- multi-GPU jobs
 *
 * SURVEY.md 8(e) / BASELINE.json configs[3]: samples are independent (every
 * pipeline register of the reference core is per sample, rtl/cordic.v:231-283;
 * the NCO phase is the closed form phase0 + n*fcw), so a job of n_total
 * samples splits into contiguous blocks by GLOBAL sample index.  The reference
 * has no counterpart (one Verilated model stepped by one thread,
 * bench/cpp/cordic_tb.cpp:127-178); this is the host side of the 8-GPU
 * configuration, in C++ behind the C ABI.
 *
 * A cordic_group is `nlocal` shards driven by THIS host process, one HIP
 * device each (hipSetDevice per shard, one plan, one compute stream and one
 * copy stream per shard), out of `total_shards` shards of the whole job:
 *   - one process driving every GPU of a node:  first_shard 0, nlocal ==
 *     total_shards == number of devices (no process group at all);
 *   - one process per GPU (torch.distributed.run / mpirun): nlocal 1,
 *     first_shard = rank, total_shards = world size; the processes only ever
 *     exchange the 8-byte digests.
 * Shard s owns [s*n/S + min(s, n%S), ...) -- sizes differ by at most one -- and
 * GENERATES ITS OWN INPUTS from the global index: no scatter, no collective on
 * the data path.  The group owns the resident buffers of its shards (in0, in1,
 * out0, out1: one 32-bit word per sample each).  Job calls only enqueue;
 * cordic_group_sync waits.  devices == NULL means ordinals 0..nlocal-1; an
 * ordinal may be listed twice (two shards sharing one GPU -- how the
 * single-GPU tests exercise the multi-shard logic).
 */
typedef struct cordic_group cordic_group;

int	cordic_device_count(void);	/* visible HIP devices, or a negative status */
/* [start, start+count) of shard `shard` of `total_shards` of an n_total job
 * (host arithmetic only: no device needed) */
int	cordic_shard_range(uint64_t n_total, int shard, int total_shards,
		uint64_t *start, uint64_t *count);
int	cordic_group_create(const cordic_config *cfg, int nlocal, const int *devices,
		int first_shard, int total_shards, cordic_group **grp);
void	cordic_group_destroy(cordic_group *grp);
int	cordic_group_size(const cordic_group *grp);	/* nlocal */
/* global range of shard `shard` (0 .. total_shards-1) of an n_total job */
int	cordic_group_range(const cordic_group *grp, uint64_t n_total, int shard,
		uint64_t *start, uint64_t *count);
/* (re)allocate the shards' buffers for jobs of n_total samples with `inputs`
 * (0, 1 or 2) input arrays; job calls do this implicitly on first use */
int	cordic_group_reserve(cordic_group *grp, uint64_t n_total, int inputs);
/* A job that READS input arrays (p2r_const: in0; r2p: in0, in1) requires
 * EACH of them, on EVERY local shard, to hold data for that job: filled by
 * cordic_group_fill_* for the same n_total, or written by the caller
 * (cordic_group_write; pieces in any order, overlapping or not, that together
 * cover the shard's whole share) since the arrays were last (re)allocated.  Growing
 * the capacity discards what the inputs held, a fill discards what the caller
 * wrote; a job whose inputs are not all there returns CORDIC_ERR_ARGS instead
 * of computing on uninitialised memory.
 * Back-to-back jobs need no cordic_group_sync between them, also with
 * forwarding set: a job's kernels wait (in stream order, on the device) until
 * the previous job's pieces have left out0 / out1. */
/* Placement of the shards' arrays (OFF unless asked).  What HBM delivers to a
 * job's streams depends on which allocations they run over: allocations come
 * in CLASSES (profiles/r05/pair_matrix.txt) -- two arrays of one class written
 * together run at 0.73-0.81 of the 8 TB/s peak, two of different classes at
 * 0.93-0.95, single arrays all alike; nothing in the addresses shows it, and it
 * holds for the arrays' lifetime.  A caller that wants the last ~5 % of a
 * write-heavy job can ask the group to choose:
 *   cordic_group_set_placement(grp, 1)  before the first reserve / job call, or
 *   CORDIC_GROUP_PLACEMENT=1            in the environment (also read by
 *                                        cordic_arrays_alloc).
 * The group then allocates, for arrays of 64 MiB or more, up to TWO arrays more
 * than it needs -- never more than a tenth of the memory that is free on the
 * device at that moment -- times an arithmetic-free twin of the job's traffic
 * over the candidate role assignments (10 + 3 launches for a 1R2W job: ~60 ms
 * at 4 GiB per array, plus the two extra hipMallocs), keeps the fastest and
 * frees the rest before the call returns.  Cost, once per (re)allocation: that
 * time, and 2 x the array size of HBM held for its duration.  With two spares
 * a process whose first five allocations share a class finds no fast pair
 * (about every second process on the boxes measured): a caller that can spare
 * more names a NUMBER -- cordic_group_set_placement(grp, N) /
 * CORDIC_GROUP_PLACEMENT=N, 2 <= N <= 16 -- and the group then goes on, one
 * array at a time, while no pair of written arrays reaches 0.93 of the HBM
 * peak: at most N spares, still at most a tenth of the free memory (six 4 GiB
 * arrays on an otherwise empty MI355X), and no further array once 1.5 s have
 * passed.  Without any of it -- the default -- the arrays are what hipMalloc
 * hands out, nothing is probed and nothing extra is allocated.
 * cordic_group_placement reports what the last allocation of a shard saw:
 * candidate arrays, probes run, the times of the best and the worst pair of
 * written arrays (0R2W) and, with those chosen, of the best and the worst
 * choice of the read arrays (the job's full pattern); 0 candidates: not tuned.
 * (Rounds 4-5 had this on by default and kept taking candidates, up to 24 and
 * 96 GiB, while no written pair was fast; bench.py asks for six spares for its
 * own arrays and says so: roofline.placement in its line, the candidates and
 * probes in its detail record.) */
int	cordic_group_set_placement(cordic_group *grp, int enable);
/* The same for callers of the stateless entry points: n_read (0..2) +
 * n_write (1..2) arrays of `bytes` bytes each on the current device: plain
 * hipMalloc, or -- with CORDIC_GROUP_PLACEMENT=1 in the environment and arrays
 * of 64 MiB and more -- placed as above; ptrs receives the read arrays
 * first, then the written ones.  Synchronises `stream`, on which the probes
 * run.  cordic_arrays_free releases them (plain hipFree would do). */
int	cordic_arrays_alloc(size_t bytes, int n_read, int n_write, void **ptrs,
		void *stream);
void	cordic_arrays_free(void **ptrs, int count);
int	cordic_group_placement(const cordic_group *grp, int local_shard,
		int *candidates, int *probes, float *written_best_ms,
		float *written_worst_ms, float *best_ms, float *worst_ms);
/* in0[i] = ((start+i) << shift) mod 2^32            (cordic_fill_phase_ramp) */
int	cordic_group_fill_phase_ramp(cordic_group *grp, uint64_t n_total, int shift);
/* in0 / in1 = the deterministic I/Q ramps of cordic_fill_iq_ramp            */
int	cordic_group_fill_iq_ramp(cordic_group *grp, uint64_t n_total,
		uint32_t mulx, uint32_t muly, int bits);
/* out0, out1 = cordic_plan_p2r_const(in0 as i_phase)                        */
int	cordic_group_p2r_const(cordic_group *grp, uint64_t n_total, int32_t xval,
		int32_t yval);
/* out0, out1 = cordic_plan_nco with index0 = the shard's global start       */
int	cordic_group_nco(cordic_group *grp, uint64_t n_total, uint32_t phase0,
		uint32_t fcw, int32_t xval, int32_t yval);
/* out0 = o_mag, out1 = o_phase of cordic_r2p(in0, in1)                      */
int	cordic_group_r2p(cordic_group *grp, uint64_t n_total);
int	cordic_group_sync(cordic_group *grp);
/* Sum over the local shards of cordic_digest_u32(out0, start) +
 * cordic_digest_u32(out1, start + 2^40): position aware, so the digests of all
 * shards of a job add up (mod 2^64) to the digest of the same job computed in
 * one piece.  Synchronises. */
int	cordic_group_digest(cordic_group *grp, uint64_t n_total, uint64_t *digest);
/* Forwarding of results to one consumer ("the final gather"): while set, every
 * job call computes its shard in `chunks` pieces and, as soon as a piece has
 * been computed, copies it into d_out0 / d_out1 (device memory of HIP device
 * root_device, n_total words each, at the piece's GLOBAL offset) on the
 * shard's copy stream -- hipMemcpyPeerAsync, i.e. the SDMA engines over xGMI,
 * so the transfer of piece k overlaps the computation of piece k+1 and costs
 * no CUs.  Only meaningful when this process holds every shard.  root_device
 * < 0 clears it. */
int	cordic_group_set_gather(cordic_group *grp, int root_device,
		int32_t *d_out0, int32_t *d_out1, int chunks);
/* The same forwarding when the shards of the job live in DIFFERENT processes
 * (one process per GPU): RCCL point-to-point over xGMI.  Bootstrap as RCCL
 * prescribes: ONE process -- normally the one holding shard 0 -- calls
 * cordic_rccl_unique_id and hands the CORDIC_RCCL_ID_BYTES bytes to every
 * process of the job by whatever means it has (a file, a pipe, MPI_Bcast,
 * torch.distributed); every process then calls cordic_group_rccl_init
 * (collective: returns when all total_shards ranks have joined; rank = global
 * shard index, one communicator per local shard).  While
 * cordic_group_set_gather_rccl is set, every job call computes its shards in
 * `chunks` pieces and sends each finished piece to shard root_shard
 * (ncclSend on the copy stream; RCCL has no gather primitive), whose process
 * posts the matching ncclRecv into d_out0 / d_out1 (device memory of the root
 * shard's device, n_total words each; ignored, may be NULL, elsewhere).
 * EVERY process of the job has to issue the same job calls with the same
 * n_total and chunks.  root_shard < 0 clears it.  librccl is opened at run
 * time on first use (CORDIC_RCCL_LIB overrides the name);
 * CORDIC_ERR_UNSUPPORTED when it cannot be. */
#define CORDIC_RCCL_ID_BYTES 128
int	cordic_rccl_unique_id(void *id);
int	cordic_group_rccl_init(cordic_group *grp, const void *id);
int	cordic_group_set_gather_rccl(cordic_group *grp, int root_shard,
		int32_t *d_out0, int32_t *d_out1, int chunks);
/* Timing marks: `slot` (0..255) is recorded on every local shard's compute
 * stream; elapsed = max over the local shards of the time between two marks
 * (synchronises), per_shard_ms (may be NULL) receives nlocal values. */
int	cordic_group_mark(cordic_group *grp, int slot);
int	cordic_group_elapsed(cordic_group *grp, int slot_a, int slot_b,
		float *max_ms, float *per_shard_ms);
/* Shard-local views for callers that consume the results in place, and a
 * read-back for checks (the destination may be host or device memory):
 * array 0..3 = in0, in1, out0, out1. */
int	cordic_group_buffers(const cordic_group *grp, int local_shard, int *device,
		void **in0, void **in1, void **out0, void **out1, uint64_t *count);
int	cordic_group_read(cordic_group *grp, int local_shard, int array,
		uint64_t offset, uint64_t count, void *host_dst);
/* the reverse: fill `count` words of a shard's array from host or device
 * memory (e.g. caller-provided inputs instead of the generated ramps) */
int	cordic_group_write(cordic_group *grp, int local_shard, int array,
		uint64_t offset, uint64_t count, const void *src);

/* --------------------------------- the reference benches' pass criteria
 *
 * bench/cpp/cordic_tb.cpp:223-337 and bench/cpp/topolar_tb.cpp:222-315 are the
 * reference's whole acceptance test: every output is compared with a double
 * precision sin/cos (atan2) of its input, reduced to a few sums and maxima,
 * and held against thresholds built from the generated header's
 * QUANTIZATION_VARIANCE / PHASE_VARIANCE_RAD / GAIN.  A cordic_quality does
 * that reduction ON THE DEVICE, over arrays that are already there, so the
 * criteria can be evaluated over all 2^32 phases of a 32-bit core in about a
 * second (the Verilated bench keeps 2^PW ints on the host; at PW = 32 its
 * `const int NSAMPLES = 1ul << PW` is 0).  Calls ACCUMULATE -- a sweep may be
 * fed in pieces -- until cordic_quality_reset; the *_result calls synchronise
 * the device, add the per-block partial sums up on the host (fixed grid, no
 * atomics: reproducible bit for bit) and apply the reference's thresholds.
 * A handle belongs to the device that was current at creation and to one
 * caller at a time; it accumulates either p2r or r2p statistics.
 */
typedef struct cordic_quality cordic_quality;

typedef struct cordic_p2r_quality {	/* cordic_tb.cpp:285-337            */
	uint64_t n;			/* samples accumulated               */
	double	avg_err;		/* "AVG Err" : sqrt(sum err^2 / n)   */
	double	max_err;		/* "MAX Err"                         */
	double	mag;			/* "Mag"     : RMS output magnitude  */
	double	input_mag;		/* RMS input magnitude (`scale`)     */
	double	alpha;			/* "(alpha)" : sumxy / sumsq         */
	double	cnr_db;			/* "CNR"                             */
	double	expected_err;		/* sqrt(QUANTIZATION_VARIANCE +
					   PHASE_VARIANCE_RAD (scale GAIN)^2) */
	double	avg_limit;		/* 1.5 expected_err                  */
	double	max_limit;		/* 5.2 expected_err                  */
	uint64_t max_err_index;		/* sample (in feeding order) of max  */
	int32_t	pass_avg, pass_max, pass_alpha, pass;
	double	sum_err2, sum_xy, sum_sq, sum_d, sum_in2;	/* raw sums  */
} cordic_p2r_quality;

typedef struct cordic_r2p_quality {	/* topolar_tb.cpp:222-256,303-330   */
	uint64_t n;
	double	max_phase_err;		/* "Max phase error", phase units    */
	double	max_mag_err;		/* "Max magnitude error"             */
	double	avg_phase_err;		/* "Avg phase err"                   */
	double	avg_mag_err;		/* RMS magnitude error (not in the
					   reference report)                 */
	double	mean_phase_err;		/* bias of the phase (likewise)      */
	double	expected_avg_phase_err;	/* sqrt(PHASE_VARIANCE_RAD) RAD_TO_PHASE */
	double	phase_limit;		/* 3.4 max(1, that)                  */
	double	mag_limit;		/* 2 sqrt(QUANTIZATION_VARIANCE)     */
	uint64_t max_phase_err_index, max_mag_err_index;
	int32_t	pass_phase, pass_mag, pass;
} cordic_r2p_quality;

int	cordic_quality_create(const cordic_config *cfg, cordic_quality **q);
void	cordic_quality_destroy(cordic_quality *q);
int	cordic_quality_reset(cordic_quality *q, void *stream);
/* p2r / sp2r cores.  d_xval / d_yval NULL: the constant vector (xval, yval),
 * as the bench holds it (cordic_tb.cpp:68-69).  Inputs are taken modulo their
 * port width exactly as the core takes them. */
int	cordic_quality_p2r(cordic_quality *q, size_t n, const int32_t *d_xval,
		const int32_t *d_yval, int32_t xval, int32_t yval,
		const uint32_t *d_phase, const int32_t *d_oxval,
		const int32_t *d_oyval, void *stream);
/* the same for outputs of cordic_nco: the phases are the closed form */
int	cordic_quality_nco(cordic_quality *q, size_t n, uint32_t phase0,
		uint32_t fcw, uint64_t index0, int32_t xval, int32_t yval,
		const int32_t *d_oxval, const int32_t *d_oyval, void *stream);
/* r2p / sr2p cores.  imag >= 0: the magnitude every sample is expected to
 * have before scaling, as the bench uses its circle's nominal radius
 * (topolar_tb.cpp:238-246: imag[i] = (int)mg); imag < 0: each sample's own
 * sqrt(x^2 + y^2). */
int	cordic_quality_r2p(cordic_quality *q, size_t n, const int32_t *d_xval,
		const int32_t *d_yval, int32_t imag, const int32_t *d_omag,
		const uint32_t *d_ophase, void *stream);
int	cordic_quality_p2r_result(cordic_quality *q, cordic_p2r_quality *out);
int	cordic_quality_r2p_result(cordic_quality *q, cordic_r2p_quality *out);
/* The r2p bench's input (topolar_tb.cpp:127-141): samples index0 .. index0+n-1
 * of 2^lgnsamples points on TWO turns of a circle of radius 2^(IW-1)-1,
 * components truncated toward zero by (int); the reference uses lgnsamples =
 * PW.  cos / sin are the device's fp64 sincospi, which may differ from the
 * host libm in the last place (the statistics refer to the integers actually
 * fed, topolar_tb.cpp:142, so they are self-consistent either way). */
int	cordic_fill_circle(int32_t *d_x, int32_t *d_y, size_t n, uint64_t index0,
		int lgnsamples, int iw, int pw, void *stream);

/* Host-array entry points: the same calls on arrays in HOST memory -- the
 * reference bench's own containers (`int` arrays filled and read by the CPU,
 * bench/cpp/cordic_tb.cpp:94-100,127-178; topolar_tb.cpp:93-99,143-187) -- for
 * callers that do not want to touch HIP.  Each call is a chunked copy pipeline
 * (16 MiB per array and chunk, three slots, private upload / run / download
 * streams chained by events): chunk k+1 uploads and chunk k-1 downloads while
 * chunk k computes, constant vectors (xy_is_scalar) run the table-seeded plan,
 * and only the pipeline's own streams are synchronised before the call
 * returns (the results are then in the caller's arrays; other streams of the
 * process are never stalled).  The kernels are ~100x faster than PCIe, so the
 * rate is the interconnect's:
 *   - arrays from cordic_host_alloc (or hipHostMalloc / hipHostRegister) are
 *     DMA'd in place: the call runs at the PCIe rate of the busier direction
 *     (p2r with constant vectors: 4 B up, 8 B down per sample);
 *   - pageable arrays (malloc / new) are staged through pinned buffers by a
 *     pool of host threads (CORDIC_HOST_THREADS, default 8): bounded by the
 *     host's memcpy rate.
 * Inputs and outputs may mix the two kinds.  The pipeline (streams, 3 x 5
 * device arrays, staging, threads, plan) is created on first use per device,
 * kept until cordic_host_release() or process exit, and serialises concurrent
 * host-array calls on its device. */
int	cordic_p2r_host(const cordic_config *cfg, size_t n,
		const int32_t *xval, const int32_t *yval, int xy_is_scalar,
		const uint32_t *phase, int32_t *oxval, int32_t *oyval);
int	cordic_r2p_host(const cordic_config *cfg, size_t n,
		const int32_t *xval, const int32_t *yval,
		int32_t *omag, uint32_t *ophase);
/* pinned host memory for the arrays of the calls above (hipHostMalloc) */
int	cordic_host_alloc(void **p, size_t bytes);
void	cordic_host_free(void *p);
/* drop every cached pipeline of the process (memory, streams, threads) */
void	cordic_host_release(void);
/* what the most recent host-array call on the current device did */
typedef struct cordic_host_stats {
	uint64_t samples;
	int32_t	chunks, chunk_samples;
	int32_t	staged_inputs;		/* input arrays that were pageable   */
	int32_t	staged_outputs;		/* output arrays that were pageable  */
	int32_t	copy_threads;		/* host threads that staged (0: none) */
	int32_t	seeded_plan;		/* 1: a chunk ran the table-seeded
					   kernel (cordic_last_kernel)      */
	int32_t	lanes;			/* pipelines (devices) that shared it */
	double	seconds;		/* wall time inside the call         */
} cordic_host_stats;
int	cordic_host_last_stats(cordic_host_stats *out);
/* Several GPUs = several PCIe links: with a device list set, every host-array
 * call is cut into as many contiguous parts (whole 16 MiB chunks) as the list
 * has entries, and each part runs through its own pipeline on its own device,
 * driven by its own host thread -- the path shards exactly like the compute
 * does.  An ordinal may be listed more than once (two pipelines sharing one
 * device and its link: how the one-GPU tests exercise it).  count == 0: back
 * to the caller's current device.  Process-wide; takes effect for calls that
 * start afterwards.  cordic_host_last_stats then reports the whole call
 * (chunks and copy threads summed, lanes = parts), cordic_host_lane_stats one
 * lane's share of it. */
int	cordic_host_set_devices(const int *devices, int count);
int	cordic_host_lane_stats(int lane, cordic_host_stats *out);

/* ------------------------------------------------ device-side test inputs */

/* d_phase[i] = ((index0 + i) << shift) mod 2^32 : the bench's phase ramp
 * i << (PW-LGNSAMPLES) (cordic_tb.cpp:128-138) for a shard starting at
 * global sample index0. */
int	cordic_fill_phase_ramp(uint32_t *d_phase, size_t n, uint64_t index0,
		int shift, void *stream);
/* d_x[i] = sext(((index0+i) * mulx) >> 8, bits), d_y likewise with muly:
 * deterministic I/Q ramps. */
int	cordic_fill_iq_ramp(int32_t *d_x, int32_t *d_y, size_t n,
		uint64_t index0, uint32_t mulx, uint32_t muly, int bits,
		void *stream);
/* Order-sensitive 64-bit digest of a device word array:
 * sum over i of mix(index0 + i, d_words[i]) mod 2^64, written to *d_digest
 * (device, 8 bytes; ACCUMULATED with atomic add -- zero it first).  The CPU
 * twin lives in tests/; digests of shards add up to the digest of the whole,
 * which is what the multi-GPU check reduces. */
int	cordic_digest_u32(const uint32_t *d_words, size_t n, uint64_t index0,
		uint64_t *d_digest, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CORDIC_AMD_H */
