/*
 * cordic_oracle.h -- CPU oracle for the CORDIC rotation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it, and only as the checker / the reported CPU baseline.
 *
 * What this is: a plain-C, one-sample-at-a-time restatement of the integer
 * arithmetic that the reference's core generator EMITS (the reference has no
 * C++ function that computes a CORDIC sample; its executable model is the
 * Verilator build of the emitted Verilog, and Verilator is absent here).
 * Every function cites the reference file:line it follows.
 *
 * Pinning status:
 *   - table / parameter math (angles, gain, variances, WW/PW/NSTAGES
 *     derivation): PINNED against outputs of the real reference generator
 *     built from /root/reference/sw by oracle/Makefile (oracle/_ref/gencordic)
 *     and against the checked-in rtl/{cordic,topolar,seqcordic,seqpolar}.{v,h}
 *     values, committed as fixtures under tests/golden/gencordic_golden.json.
 *   - per-sample arithmetic (pre-rotation, stages, rounding): the reference
 *     ships NO per-sample golden vectors and its only executor (a Verilator
 *     build) cannot be made here, so sample-level parity is UNPINNED BY
 *     REFERENCE FIXTURES.  What ties it to the reference:
 *     (a) tests/vsim.py, this project's own small cycle simulator, EXECUTES
 *         the reference's Verilog text -- the rtl/ .v files where they lie and whatever
 *         oracle/_ref/gencordic emits -- clock by clock, and this oracle must
 *         agree with it sample for sample (tests/test_rtl_vectors.py; the
 *         resulting vectors are committed as tests/golden/rtl_vectors.json
 *         with the script that made them);
 *     (b) the reference's own pass criteria (bench/cpp/cordic_tb.cpp:285-337,
 *         bench/cpp/topolar_tb.cpp:303-315) evaluated on this oracle's output
 *         at the checked-in configuration.
 *     vsim.py is not Verilator and not a reference build; it is a second,
 *     independent reading of the same text.
 */
#ifndef CORDIC_ORACLE_H
#define CORDIC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_P2R = 0, ORC_R2P = 1, ORC_SP2R = 2, ORC_SR2P = 3 };
#define ORC_MAX_STAGES 64

typedef struct orc_config {
	int	mode;
	int	iw, ow, nxtra, ww, pw, nstages;
	uint32_t angle[ORC_MAX_STAGES];	/* PW-bit, right justified */
	double	quantization_variance, phase_variance_rad, gain;
	double	best_possible_cnr;	/* p2r / sp2r only */
	int	clocks_per_output;	/* sequential cores only, else 0 */
} orc_config;

/* sw/cordiclib.cpp restatements */
int	orc_nextlg(unsigned vl);
double	orc_cordic_gain(int nstages);
double	orc_phase_variance(int nstages, int phase_bits);
double	orc_transform_quantization_variance(int nstages, int xtrabits,
		int dropped_bits);
void	orc_cordic_angles(int nstages, int phase_bits, uint32_t *out);
int	orc_calc_stages2(int working_width, int phase_bits);
int	orc_calc_stages1(int phase_bits);
int	orc_calc_phase_bits(int output_width);

/* sw/main.cpp:260-357 -- CLI level: xtra is the -x value (default 2),
 * phase_bits / nstages <= 0 mean "derive".  Returns 0 or a negative code. */
int	orc_config_cli(orc_config *cfg, int mode, int iw, int ow, int xtra,
		int phase_bits, int nstages);
/* emitter level (sw/basiccordic.h:46-50 etc.): nxtra already incremented */
int	orc_config_core(orc_config *cfg, int mode, int nstages, int iw, int ow,
		int nxtra, int phase_bits);

/* Per-sample arithmetic.  xy_stride = 1: x[i], y[i]; 0: x[0], y[0] for all. */
void	orc_p2r(const orc_config *cfg, size_t n, const int32_t *x,
		const int32_t *y, int xy_stride, const uint32_t *phase,
		int32_t *ox, int32_t *oy);
void	orc_r2p(const orc_config *cfg, size_t n, const int32_t *x,
		const int32_t *y, int32_t *omag, uint32_t *ophase);
/* closed forms of the sequential cores */
void	orc_seq_p2r(const orc_config *cfg, size_t n, const int32_t *x,
		const int32_t *y, int xy_stride, const uint32_t *phase,
		int32_t *ox, int32_t *oy);
void	orc_seq_r2p(const orc_config *cfg, size_t n, const int32_t *x,
		const int32_t *y, int32_t *omag, uint32_t *ophase);
/* clock-by-clock models of rtl/seqcordic.v / rtl/seqpolar.v; return the
 * number of ticks from i_stb to o_done (must equal clocks_per_output) or -1 */
int	orc_seq_p2r_cycle(const orc_config *cfg, int32_t x, int32_t y,
		uint32_t phase, int32_t *ox, int32_t *oy);
int	orc_seq_r2p_cycle(const orc_config *cfg, int32_t x, int32_t y,
		int32_t *omag, uint32_t *ophase);

/* register-level model of the sequential cores over whole port traces,
 * including their off-protocol behaviour (i_stb on a completing clock) */
typedef struct orc_seq_regs {
	int64_t	prex, prey, xv, yv;
	uint32_t preph, ph, cangle;
	uint32_t state;
	int32_t	idle, pre_valid, aux, o_done, o_aux;
	int32_t	o0, o1;
} orc_seq_regs;
void	orc_seq_regs_init(orc_seq_regs *r);
void	orc_seq_trace(const orc_config *cfg, size_t T, const uint8_t *stb,
		const uint8_t *rst, const uint8_t *aux, const int32_t *x,
		const int32_t *y, const uint32_t *phase, int32_t *o0, int32_t *o1,
		uint8_t *oaux, uint8_t *busy, uint8_t *done, orc_seq_regs *regs);

/* dispatch on cfg->mode */
void	orc_rotate(const orc_config *cfg, size_t n, const int32_t *x,
		const int32_t *y, int xy_stride, const uint32_t *phase,
		int32_t *ox, int32_t *oy);
void	orc_topolar(const orc_config *cfg, size_t n, const int32_t *x,
		const int32_t *y, int32_t *omag, uint32_t *ophase);
/* NCO: phase[i] = phase0 + (index0+i)*fcw mod 2^PW, then orc_rotate */
void	orc_nco(const orc_config *cfg, size_t n, uint32_t phase0, uint32_t fcw,
		uint64_t index0, int32_t x0, int32_t y0,
		int32_t *ox, int32_t *oy);
/* the same accumulator on i_phase with per-sample i_xval / i_yval (mixer) */
void	orc_mixer(const orc_config *cfg, size_t n, uint32_t phase0, uint32_t fcw,
		uint64_t index0, const int32_t *x, const int32_t *y,
		int32_t *ox, int32_t *oy);

/* table cores (sw/sintable.cpp; kind 4 = -t tbl, 5 = -t qtr) */
int	orc_table_config(int kind, int iw, int ow, int phase_bits, int *pw_out,
		int *ow_out);
void	orc_table_values(int kind, int pw, int ow, int32_t *out);
void	orc_table_lookup(int kind, int pw, int ow, const int32_t *tbl, size_t n,
		const uint32_t *phase, int32_t *out);

/* quadratically interpolated sine core (sw/quadtbl.cpp, rtl/quadtbl.v) */
typedef struct orc_quad {
	int	pw, ow, xtra;		/* XTRA of the emitted core */
	int	wid;			/* OW + nxtra the tables were built for */
	int	ww, lgtbl, dxbits, cbits, lbits, qbits;
	long	scale;
	double	itbl_err, tbl_err, spur_db;
} orc_quad;
/* CLI level (sw/main.cpp:444-463); returns 0 or a negative code when the
 * reference would assert / emit a core that cannot elaborate */
int	orc_quad_cli(orc_quad *q, int iw, int ow, int xtra, int phase_bits);
int	orc_quad_core(orc_quad *q, int phase_bits, int ow, int nxtra);
/* tables of 2^lgtbl signed entries each */
int	orc_quad_tables(const orc_quad *q, long *ctbl, long *ltbl, long *qtbl);
void	orc_quad_lookup(const orc_quad *q, const long *ctbl, const long *ltbl,
		const long *qtbl, size_t n, const uint32_t *phase, int32_t *out);

/* bench.py cpu_baseline: samples processed by nthreads threads in `seconds` */
uint64_t orc_throughput(const orc_config *cfg, int kind, int nthreads,
		double seconds, uint32_t phase_mul, int32_t x0, int32_t y0);

/* The oracle's outputs for EVERY sample g in [start, start+n) of a synthetic
 * job, as the device's position-aware digest (cordic_digest_u32):
 *   sum mix(g, out0[g]) + mix(g + 2^40, out1[g])  mod 2^64.
 * kind 0: rotator, constant (x0,y0), phase[g] = phase0 + (uint32)g*fcw;
 * kind 1: converter on x[g] = sext_iw(((uint32)g*mulx)>>8), y likewise (muly);
 * kind 2: rotator on kind 1's vectors and kind 0's phases.
 * nthreads POSIX threads; *seconds (may be NULL) receives the wall time. */
uint64_t orc_digest(const orc_config *cfg, int kind, int nthreads,
		uint64_t start, uint64_t n, uint32_t phase0, uint32_t fcw,
		int32_t x0, int32_t y0, uint32_t mulx, uint32_t muly,
		double *seconds);
/* sum over i of mix(index0 + i, w[i]): CPU twin of cordic_digest_u32 */
uint64_t orc_digest_words(const uint32_t *w, size_t n, uint64_t index0);

#ifdef __cplusplus
}
#endif
#endif
