// cordic_internal.h -- declarations shared by the host and device halves of
// libcordic_amd.so.  Not installed; the public surface is include/cordic_amd.h.
#ifndef CORDIC_INTERNAL_H
#define CORDIC_INTERNAL_H

#include <cstddef>
#include <cstdint>

#include "cordic_amd.h"

namespace cordic_amd {

// ---- host: cordic_config.cpp
uint32_t arctan_entry(unsigned k, int phase_bits);
double	rotation_gain(int nstages);
uint32_t gain_annihilator(int nstages);
uint32_t core_gain_annihilator(const cordic_config &c);
double	phase_variance(int nstages, int phase_bits);
double	quantization_variance(int nstages, int xtrabits, int dropped_bits);
int	next_lg(unsigned vl);
int	stages_for(int phase_bits, int working_width);
int	phase_bits_for(int width);
int	build_core(cordic_config *cfg, int mode, int nstages, int iw, int ow,
		int nxtra, int phase_bits);
int	build_from_cli(cordic_config *cfg, int mode, int iw, int ow, int xtra,
		int phase_bits, int nstages);
int	parse_args(cordic_config *cfg, int argc, const char *const *argv,
		char *fname, size_t fname_cap, int *c_header);
int	write_header(const cordic_config *c, const char *name, char *buf,
		size_t cap);
const char *status_text(int s);
// Caller-owned PODs are checked before anything indexes with their fields:
// could this struct have come out of the matching *_init call?
bool	config_sane(const cordic_config &c);
bool	table_sane(const cordic_table_config &t);
bool	quad_sane(const cordic_quad_config &q);
int	table_derive(cordic_table_config *t, int kind, int iw, int ow, int pw);
int	table_fill(const cordic_table_config &t, int32_t *out, size_t cap);

// ---- host: cordic_quadtbl.cpp
int	quad_build_core(cordic_quad_config *q, int phase_bits, int ow, int nxtra);
int	quad_build_from_cli(cordic_quad_config *q, int iw, int ow, int xtra,
		int phase_bits);
int	quad_fill(const cordic_quad_config &q, int32_t *c, int32_t *l,
		int32_t *qq, size_t cap);
int	quad_write_header(const cordic_quad_config *q, const char *name,
		char *buf, size_t cap);

// Seed tables: stages replaced by the lookup and threads per block of the
// seeded kernels (one table per block).
// M = 11: 1618 leaves x 4 quadrants x 16 B of seeds + 32 KiB of buckets = 136 KiB
// of LDS, ONE 1024-thread block per CU.  Measured against M = 10 (68 KiB, two
// blocks per CU) on one box: cfg2 +1 %, cfg5 +2 %, cfg1 +3 %, cfg4 +4 %
// (profiles/r02/ab_seed_depth.txt): the kernels are instruction- and
// power-bound, one micro-rotation less per sample outweighs the lost
// occupancy.  M = 12 does not fit the 160 KiB.
#ifndef CORDIC_SEED_STAGES
#define CORDIC_SEED_STAGES 11
#endif
// LDS the seeded kernels may use per block: buckets + seeds + tile-id slots
#define CORDIC_SEED_LDS_BYTES (160u * 1024u)
#ifndef CORDIC_SEED_BLOCK
#define CORDIC_SEED_BLOCK 1024
#endif

// "Direction tails" behind the seed table (round 3).  The rotation directions
// of the stages AFTER the seeded ones depend only on the residual phase, again
// as a monotone step function with exact integer break points -- so they can
// be looked up too: a small second table per group of `t` stages, indexed by
// the residual, hands the kernel the stage multipliers (no phase recurrence,
// no multiplier extraction: 4 instead of 7 VALU instructions per stage, ~6 per
// lookup).  The schedule -- how the R = NLIVE - M remaining stages are cut
// into groups -- is fixed here for host and device alike: as few groups as
// hold at most seven stages each (15 dwords of entry), of equal size with the
// longer ones last.  A lookup is not free -- a bucket and up to 60 bytes of
// LDS per sample and group, and the LDS returns 128 bytes a cycle per CU
// whether or not the lanes agree on the address -- so fewer, longer groups
// win: 13 stages as 6+7 run 4 % faster than as 5+5+3, those 5 % faster than
// as 3+3+3+4 (profiles/r03/ab_tails.txt).  Measured gains on a phase ramp by
// stages behind the seeds: 5 (cfg2: one group) +1.5 %, 7 +10 %, 9 +7 %,
// 11 +9 %, 13 (cfg4) +12 %; the NCO of cfg5 (5 behind, unrelated phases)
// +2.6 %.  Fewer than kDtMinStages: not measured, no tails.
constexpr int kDtMaxLevels = 4;
#ifndef CORDIC_DT_MIN_STAGES
#define CORDIC_DT_MIN_STAGES 5
#endif
constexpr int kDtMinStages = CORDIC_DT_MIN_STAGES;
// A row of 256 phases takes the tails when its first and last phase are less
// than 2^kDtCoherentLog2 left-justified units apart.  With the entries spread
// over the banks (dt_entry_dwords) a ramp of ANY slope gains or breaks even
// (cfg4: +12 % at one unit per sample, +3...8 % at 2^7...2^11, 0 at 2^13 and
// 2^16); only unrelated phases lose on the cores with 64-128 leaves per group
// (cfg4 -5.6 %, the 29-stage core -8 %; smaller tables GAIN there too:
// 19 stages +9 %, 23 stages +5 %), and a row of those passes the test once
// in 2^7.  profiles/r03/ab_tails.txt, parts 13-15.
#ifndef CORDIC_DT_COHERENT_LOG2
#define CORDIC_DT_COHERENT_LOG2 24
#endif
constexpr int kDtCoherentLog2 = CORDIC_DT_COHERENT_LOG2;
constexpr int kDtMaxT = 7;		// stages per group at most (entry: 15 dwords)
// Stage i turns the phase by ~2^32 / (2 pi 2^(i+1)) left-justified units
// whatever PW is: behind stage 24 the leaves of a group get narrower than the
// 16 units of the smallest bucket.  The tails stop there; later stages (the
// 29-30 stage cores gencordic derives for 32-bit outputs) run the recurrence
// on the residual the last group leaves.
constexpr int kDtLastStage = 25;
constexpr int dt_covered(int r)			// stages served by lookups
{
	return r < kDtLastStage - CORDIC_SEED_STAGES ? r : kDtLastStage - CORDIC_SEED_STAGES;
}
constexpr int dt_levels(int r)
{
	return r < kDtMinStages ? 0 : (dt_covered(r) + kDtMaxT - 1) / kDtMaxT;
}
constexpr int dt_size(int r, int level)		// stages of group `level`
{
	const int n = dt_levels(r), c = dt_covered(r);
	return n == 0 ? 0 : c / n + (level >= n - c % n ? 1 : 0);
}
constexpr int dt_first(int r, int level)	// stages before group `level`
{
	int done = 0;
	for (int g = 0; g < level; g++) done += dt_size(r, g);
	return done;
}
constexpr int dt_rest_stages(int r) { return dt_levels(r) == 0 ? r : r - dt_covered(r); }
// every group small enough (<= 5 stages: <= 32 leaves on a 12-dword stride)
// for its lookups to win on unrelated phases too: no per-row choice, no
// recurrence body in the kernel
constexpr bool dt_always(int r)
{
	const int n = dt_levels(r);
	return n > 0 && dt_size(r, n - 1) <= 5 && dt_rest_stages(r) == 0;
}
// Job sets (tile descriptors) run the dynamic-exit instance -- except on the
// stage counts BASELINE names, for which the left-justified WW 35 unit
// carries a static instance WITH the descriptor loop and the direction tails:
// 16 stages (one always-lookup group) and 24 stages, phase arrays and -- since
// round 6 -- NCO banks alike: whether a row takes the tails is the kernel's own
// per-row test (first and last phase of the row less than 2^kDtCoherentLog2
// apart), which an NCO job's rows pass or fail by the job's own increment, so
// jobs with unlike increments share one launch.  dtn = the tail groups the
// plan carries.
constexpr bool desc_static(int nlive, int dtn)
{
	return dtn >= 1 && dtn == dt_levels(nlive - CORDIC_SEED_STAGES)
		&& (nlive == 16 || nlive == 24);
}
constexpr int dt_rest(int r)			// stages left to the phase chain
{
	return dt_levels(r) == 0 ? r : r - dt_covered(r);
}
// LDS entry of one leaf of a group of t stages (built by the kernel's prologue
// from the table's {pattern, off'} pairs): the multipliers {-s_j, s_j} 2^LJ of
// the first dt_pairs(t) stages as pairs, s_j 2^LJ alone for the others (the
// kernel negates it: one more instruction, four bytes less to read), then
// off'.  The lookup is bounded by the LDS's return bandwidth as much as by the
// instruction issue (a 16-byte read costs the CU 8 cycles whether or not the
// lanes agree on the address), so the entry is as many bytes as are used, read
// with b128/b96/b64/b32 as they fit, 16-byte aligned.
#ifndef CORDIC_DT_SINGLES
#define CORDIC_DT_SINGLES 0
#endif
constexpr int dt_pairs(int t) { return t > CORDIC_DT_SINGLES ? t - CORDIC_DT_SINGLES : 0; }
#ifndef CORDIC_DT_STRIDE16
#define CORDIC_DT_STRIDE16 20
#endif
constexpr int dt_entry_dwords(int t)
{
	// a 16-dword stride puts every entry of a 6- or 7-stage group on the
	// same four bank groups: measured -29...-45 % on unrelated phases and
	// -21 % on a ramp of 2^9 units per sample, where 20 dwords give -6...+5 %
	// and +3 %
	const int d = (t + dt_pairs(t) + 1 + 3) & ~3;
	return d == 16 ? CORDIC_DT_STRIDE16 : d;
}
// Host-side description of one group (also what the kernel gets as arguments)
struct DtLevel {
	int32_t	t;		// stages in the group
	int32_t	shift;		// bucket = u >> shift   (u = biased residual)
	int32_t	nb;		// buckets (a power of two)
	int32_t	nl;		// leaves
	int32_t	word;		// offset of the group's buckets in the table words
};
struct DtInfo {
	int32_t	n = 0;			// groups (0: the core has no direction tails)
	uint32_t bias0 = 0;		// bias of the seed stage's residual
	uint32_t bias_last = 0;		// bias of the residual behind the last group
	DtLevel	lv[kDtMaxLevels] = {};
};

// ---- direction tables for PER-SAMPLE vectors (cordic_plan_p2r, round 4).
// The directions of ALL stages depend on the phase alone, whatever i_xval /
// i_yval are (rtl/cordic.v:262-280): the first micro-rotation comes out of the
// octant fold's multiply-adds (cordic_device.h: fold1), every later stage up
// to kDtLastStage takes its multipliers from lookups over the residual phase,
// in groups of at most kDxMaxT = 5 stages (<= 32 leaves: the lookups win on
// unrelated phases too, so every row looks up); stages behind that run the
// phase recurrence on the residual the last group leaves.
constexpr int kDxMaxLevels = 5;
constexpr int kDxMaxT = 5;
constexpr int kDxFirst = 1;		// stages the fold performs
constexpr int dx_covered(int nlive)
{
	const int r = nlive - kDxFirst, lim = kDtLastStage - kDxFirst;
	return r < 0 ? 0 : (r < lim ? r : lim);
}
constexpr int dx_levels(int nlive) { return (dx_covered(nlive) + kDxMaxT - 1) / kDxMaxT; }
constexpr int dx_size(int nlive, int level)
{
	const int n = dx_levels(nlive), c = dx_covered(nlive);
	return n == 0 ? 0 : c / n + (level >= n - c % n ? 1 : 0);
}
constexpr int dx_first(int nlive, int level)	// stages before group `level`
{
	int done = kDxFirst;
	for (int g = 0; g < level; g++) done += dx_size(nlive, g);
	return done;
}
constexpr int dx_rest(int nlive) { return nlive - kDxFirst - dx_covered(nlive); }
struct DxInfo {
	int32_t	n = 0;			// groups (0: no table)
	uint32_t bias0 = 0;		// u_0 = p_1 + bias0 (p_1: residual behind stage 1)
	uint32_t bias_last = 0;
	DtLevel	lv[kDxMaxLevels] = {};
};
// words: [0] n [1] bias0 [2] bias_last [3] 0, then per group {t, shift, nb, nl,
// 0, 0}, nb x {bound-1, first_leaf}, nl x {pattern, off'} (as the tails)
size_t	build_dir_table(const cordic_config &c, uint32_t *buf, size_t cap,
		DxInfo *info);

// ---- host: cordic_plan.cpp
bool	seed_eligible(const cordic_config &c, int m);
// `dt` (may be NULL) receives the direction tails appended behind the seed
// table's words; the returned length includes them.
size_t	build_seed_table(const cordic_config &c, int m, uint32_t *buf, size_t cap,
		DtInfo *dt = nullptr);

// ---- device launchers: cordic_kernels.hip
extern thread_local int g_last_kernel;	// enum cordic_kernel_family
// Where the rotator's phase / vector inputs come from.
enum class Feed : int {
	PhaseArray_ConstXY = 0,	// d_phase[], scalar x/y      (cordic_p2r_const)
	PhaseArray_XYArray = 1,	// d_phase[], d_x[], d_y[]    (cordic_p2r)
	Nco_ConstXY	   = 2	// phase from the sample index (cordic_nco)
};

// Round 5: what the seeded kernels' prologue computes -- the seeds of every
// (octant, leaf) for ONE constant vector, the buckets, the tail tables: the
// block's LDS image -- kept on the device by the plan, per (container, x0, y0).
// Write-once slots: an image is built by the kernel itself in build mode
// (cordic_device.h: SeedArgs::image_out) on the stream of the first launch that
// wants it, other streams wait for that launch's event until the host has seen
// it complete, and a slot is never rewritten -- so nothing a launch in flight
// (or a captured graph) reads can change under it.  More than kSeedImageSlots
// different vectors on one plan: the later ones run the in-kernel prologue, as
// every launch did before round 5.  Implemented in cordic_kernels.hip.
constexpr int kSeedImageSlots = 8;
struct SeedImages;
SeedImages *seed_images_create();
void	seed_images_destroy(SeedImages *c);
bool	seed_images_settle(SeedImages *c);	// host wait; false: an event failed
// images held, launches served from one, launches that ran the prologue
void	seed_images_info(const SeedImages *c, int32_t *held, uint64_t *hits,
		uint64_t *misses);

// ---- many small jobs in one launch (cordic_jobset, round 5).  The batch is
// cut into the seeded kernel's tiles on the host, once: one TileDesc per tile
// in queue order, and one TailDesc per sample behind a job's last whole vector
// (jobs need not be multiples of four samples).
struct TileDesc {		// 32 bytes, read by rotator_seeded's dynamic-exit loop
	uint64_t in;		// phase arrays: address of the tile's first i_phase
				// word; NCO: {high: fcw, low: phase of the tile's
				// first sample}, both left-justified
	uint64_t ox, oy;	// addresses of the tile's first o_xval / o_yval words
	uint32_t live;		// vectors (4 samples) of the tile that exist
	uint32_t pad;
};
struct TailDesc {		// 32 bytes, one SAMPLE (rotator_job_tails)
	uint64_t in;		// as TileDesc::in, for this one sample
	uint64_t ox, oy;
	uint64_t pad;
};
// Round 6: the data-fed kernels' jobs (CORDIC_JOBS_R2P / _P2R_XY / _MIX): up
// to three inputs and two outputs per tile.  r2p: in0 = i_xval, in1 = i_yval,
// o0 = o_mag, o1 = o_phase; p2r with per-sample vectors: in2 = address of the
// tile's first i_phase word; mixer: in2 = {high: fcw, low: phase of the tile's
// first sample}, both left-justified (as TileDesc::in of an NCO job).
struct TileDescXY {		// 48 bytes (topolar_lj_jobs, rotator_xydir<.., JOBS>)
	uint64_t in0, in1, in2;
	uint64_t o0, o1;
	uint32_t live;		// whole vectors (4 samples) of the tile; for a
	uint32_t pad;		//   TAIL entry (one sample): unused
};
constexpr uint32_t kJobTileVecs = CORDIC_SEED_BLOCK * 2;	// = dev::kSeedBlock * kSeedSub
// tiles of the data-fed kinds: whole passes of a 256-thread block (256 vectors),
// at most as long as the seeded kernels' tiles, shorter for small sets so that
// every CU still gets several blocks' worth (cordic_abi.cpp: xy_tile_vecs)
constexpr uint32_t kJobPassVecs = 256;
struct JobTables {
	const uint32_t *tiles = nullptr;	// device: ntiles x TileDesc (TileDescXY)
	uint32_t ntiles = 0;
	const uint32_t *tails = nullptr;	// device: ntails x TailDesc (TileDescXY)
	uint32_t ntails = 0;
	uint64_t samples = 0;			// of all jobs
};

struct RotatorJob {
	const int32_t  *x = nullptr, *y = nullptr;	// Feed::PhaseArray_XYArray
	const uint32_t *phase = nullptr;
	int32_t  x0 = 0, y0 = 0;			// const feeds
	uint32_t phase0 = 0, fcw = 0;			// NCO
	uint64_t index0 = 0;				// NCO
	// Feed::PhaseArray_XYArray with generated phases (the fused NCO mixer:
	// per-sample i_xval / i_yval, i_phase = phase0 + (index0 + i) * fcw)
	bool	 xy_nco = false;
	int32_t  *ox = nullptr, *oy = nullptr;
	size_t   n = 0;
	// sample arrays hold int16 / uint16 values (the pointers above are then
	// really int16_t* / uint16_t*); needs IW, OW <= 16 and, with a phase
	// array, PW <= 16
	bool	 io16 = false;
	// optional seed table (cordic_plan): device words + their host header
	const uint32_t *seed_table = nullptr;
	int seed_m = 0, seed_S = 0, seed_nbuckets = 0, seed_nleaves = 0;
	DtInfo	dt;			// direction tails of the plan (dt.n == 0: none)
	// per-sample vectors: the plan's direction tables (device words + header)
	const uint32_t *dir_table = nullptr;
	DxInfo	dx;
	// tile queue of the seeded kernel: CORDIC_QUEUE_BYTES of zeroed device
	// memory that no other launch in flight uses (the kernel leaves it zeroed
	// again: cordic_device.h queue_leave); NULL = static chunk-per-block sweep
	uint32_t *queue = nullptr;
	// the plan's cache of prologue images (NULL: every block computes its own)
	SeedImages *images = nullptr;
	// build the image for (x0, y0) and return: no samples (cordic_plan_prepare)
	bool	prepare_only = false;
	// batch size from which the plan takes its table-driven kernels
	// (cordic_plan_set_min_samples); < 0: the library's default
	long long min_samples = -1;
};
#define CORDIC_QUEUE_BYTES 2048	/* 8 counters, one 256-byte line each */

int	launch_rotator(const cordic_config &cfg, Feed feed, const RotatorJob &job,
		void *stream);
// a whole job set through the seeded kernel in ONE launch (+ one small launch
// for the jobs' trailing samples); `job` carries the plan's tables, the
// constant vector and the queue.  CORDIC_ERR_UNSUPPORTED: this core has no
// seeded kernel -- the caller then runs the jobs one by one.
int	launch_rotator_jobs(const cordic_config &cfg, Feed feed, const RotatorJob &job,
		const JobTables &tabs, void *stream);
// the data-fed kinds of a job set in ONE launch (+ one for trailing samples):
// kind = CORDIC_JOBS_R2P / _P2R_XY / _MIX, tabs of TileDescXY; `job` carries a
// rotator plan's direction tables.  CORDIC_ERR_UNSUPPORTED: no tile-reading
// instance for this core -- the caller runs the jobs one by one.
int	launch_xy_jobs(const cordic_config &cfg, int kind, const RotatorJob &job,
		const JobTables &tabs, void *stream);
int	launch_topolar(const cordic_config &cfg, size_t n, const int32_t *x,
		const int32_t *y, int32_t *mag, uint32_t *phase, void *stream,
		bool io16 = false);
int	launch_fill_phase_ramp(uint32_t *p, size_t n, uint64_t index0, int shift,
		void *stream);
int	launch_fill_iq_ramp(int32_t *x, int32_t *y, size_t n, uint64_t index0,
		uint32_t mulx, uint32_t muly, int bits, void *stream);
int	launch_table_lookup(const cordic_table_config &t, const int32_t *d_tbl,
		size_t n, const uint32_t *phase, int32_t *val, void *stream,
		const int16_t *d_lds16 = nullptr, int lds_mode = 0,
		int lds_entries = 0, uint32_t *queue = nullptr);
// ---- clocked view of the pipelined cores: cordic_stream.hip
struct StreamState {
	void	*ws = nullptr;		// scan / gather workspace
	size_t	ws_bytes = 0;
	// the L = NSTAGES+2 samples inside the pipeline and the number of
	// advancing clocks since the last reset; on the device, updated in place
	int32_t	 *hx = nullptr, *hy = nullptr;
	uint32_t *hph = nullptr;
	uint8_t	 *haux = nullptr;
	uint32_t *epoch = nullptr;
	uint32_t *born_phase = nullptr;	// topolar: o_phase of a cleared stage
};
size_t	stream_workspace_bytes(size_t ticks);
int	launch_stream_ticks(const cordic_config &cfg, StreamState &s, size_t T,
		const uint8_t *ce, const uint8_t *reset, const uint8_t *aux,
		const int32_t *x, const int32_t *y, const uint32_t *phase,
		int32_t *o0, int32_t *o1, uint8_t *oaux, void *stream);
// handshake view of the sequential cores: cordic_stream.hip
struct SeqState {
	void	*ws = nullptr;
	size_t	ws_bytes = 0;
	// device-resident, updated in place: clocks left until the sample in
	// flight loads (0 = idle), that sample, and the output registers
	uint32_t *c = nullptr;
	int32_t	 *px = nullptr, *py = nullptr;
	uint32_t *pph = nullptr;
	uint8_t	 *paux = nullptr;
	int32_t	 *l0 = nullptr, *l1 = nullptr;
	uint8_t	 *la = nullptr;
	unsigned long long *violations = nullptr;
	// register-level state for off-protocol stretches (cordic_stream.hip:
	// SeqLit): the core's whole register file, the padded arctan table, a
	// snapshot of the fields above taken at the start of every block
	void	*lit = nullptr;
};
size_t	seq_literal_bytes();
// fills the host image of a fresh SeqLit (power-on registers, padded table)
void	seq_literal_init(const cordic_config &cfg, void *host_image);
size_t	seq_workspace_bytes(size_t ticks);
int	launch_seq_ticks(const cordic_config &cfg, SeqState &s, size_t T,
		const uint8_t *stb, const uint8_t *reset, const uint8_t *aux,
		const int32_t *x, const int32_t *y, const uint32_t *phase,
		int32_t *o0, int32_t *o1, uint8_t *busy, uint8_t *done,
		uint8_t *oaux, void *stream);
int	launch_quad_lookup(const cordic_quad_config &q, const int32_t *d_tables,
		size_t n, const uint32_t *phase, int32_t *val, void *stream,
		uint32_t *queue = nullptr);
int	launch_digest_u32(const uint32_t *w, size_t n, uint64_t index0,
		uint64_t *digest, void *stream);
// arithmetic-free `reads`R `writes`W stream over nwords 32-bit words per array
// (0R2W, 1R2W, 2R2W, 1R1W, 2R1W); the written arrays are overwritten.  With
// `queue` (CORDIC_QUEUE_BYTES of zeroed device memory) the stream runs in the
// seeded kernel's work distribution, else as one-shot 4 KiB tiles.
int	launch_stream_probe(int reads, int writes, const void *r0, const void *r1,
		void *w0, void *w1, size_t nwords, void *stream,
		uint32_t *queue = nullptr);

} // namespace cordic_amd
#endif
