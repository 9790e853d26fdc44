/*
 * cordic_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see the header).
 *
 * A literal, scalar restatement of the arithmetic emitted by the reference
 * core generator.  "Literal" means: one C statement per Verilog non-blocking
 * assignment, explicit WW-bit two's-complement wrap after every x/y
 * operation, explicit PW-bit wrap after every phase operation, and the
 * floating-point table math kept in the reference's operation order.
 *
 * Sample-level parity is UNPINNED BY REFERENCE FIXTURES (none exist, and the
 * reference's only executor is a Verilator build that cannot be made here);
 * it is cross-checked against tests/vsim.py executing the reference's Verilog
 * text (see the header).  Table/parameter math is pinned against the real
 * generator's output.
 */
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include "cordic_oracle.h"

/* ---------------------------------------------------------------- helpers */

/* v wrapped to a w-bit two's complement number, sign extended to 64 bits */
static inline int64_t sx(int64_t v, int w)
{
	if (w >= 64)
		return v;
	return (int64_t)((uint64_t)v << (64 - w)) >> (64 - w);
}

/* Verilog ">>>" on a signed WW-bit register held sign extended in 64 bits:
 * shifting by >= the width leaves only copies of the sign bit. */
static inline int64_t asr(int64_t v, unsigned s)
{
	return v >> (s > 63 ? 63 : s);
}

static inline uint32_t pmask(int pw)
{
	return (pw >= 32) ? 0xffffffffu : ((1u << pw) - 1u);
}

/* ------------------------------------------- sw/cordiclib.cpp restatement */

/* sw/cordiclib.cpp:57-63 */
int orc_nextlg(unsigned vl)
{
	unsigned r, lg = 0;
	for (r = 1; r < vl; r <<= 1, lg++)
		;
	return (int)lg;
}

/* sw/cordiclib.cpp:66-80 */
double orc_cordic_gain(int nstages)
{
	double gain = 1.0;
	for (int k = 0; k < nstages; k++) {
		double dgain = 1.0 + pow(2.0, -2. * (k + 1));
		dgain = sqrt(dgain);
		gain = gain * dgain;
	}
	return gain;
}

/* sw/cordiclib.cpp:82-109 */
double orc_phase_variance(int nstages, int phase_bits)
{
	double RAD_TO_PHASE = (1ul << (phase_bits - 1)) / M_PI;
	double variance = 1. / 12.;
	for (unsigned k = 0; k < (unsigned)nstages; k++) {
		double x, err;
		unsigned long phase_value;
		x = atan2(1., pow(2, k + 1)) * RAD_TO_PHASE;
		phase_value = (unsigned)x;
		err = phase_value - x;
		err *= err;
		variance += err;
	}
	variance /= pow(RAD_TO_PHASE, 2.);
	return variance;
}

/* sw/cordiclib.cpp:111-130 */
double orc_transform_quantization_variance(int nstages, int xtrabits,
		int dropped_bits)
{
	double current_variance = pow(2, 2 * xtrabits) / 12.;
	for (int k = 0; k < nstages; k++)
		current_variance = (1 + pow(4, -k - 1)) * current_variance
					+ 1. / 3.;
	if (dropped_bits > 0)
		current_variance = pow(2, -2 * dropped_bits) * current_variance
					+ 1 / 12.;
	return current_variance;
}

/* one table entry, sw/cordiclib.cpp:161-169 (operation order preserved) */
static uint32_t angle_entry(unsigned k, int phase_bits)
{
	double x = atan2(1., pow(2, k + 1));
	x *= (4.0 * (1ul << (phase_bits - 2))) / (M_PI * 2.0);
	return (uint32_t)(unsigned)x;
}

void orc_cordic_angles(int nstages, int phase_bits, uint32_t *out)
{
	for (unsigned k = 0; k < (unsigned)nstages; k++)
		out[k] = angle_entry(k, phase_bits);
}

/* sw/cordiclib.cpp:214-229 */
int orc_calc_stages2(int working_width, int phase_bits)
{
	unsigned nstages;
	for (nstages = 0; nstages < 64; nstages++) {
		if (angle_entry(nstages, phase_bits) == 0)
			break;
		if (working_width <= (int)nstages)
			break;
	}
	return (int)nstages;
}

/* sw/cordiclib.cpp:231-244 */
int orc_calc_stages1(int phase_bits)
{
	unsigned nstages;
	for (nstages = 0; nstages < 64; nstages++)
		if (angle_entry(nstages, phase_bits) == 0)
			break;
	return (int)nstages;
}

/* sw/cordiclib.cpp:246-268 */
int orc_calc_phase_bits(int output_width)
{
	unsigned phase_bits;
	for (phase_bits = 3; phase_bits < 64; phase_bits++) {
		double ds, a;
		a = (2.0 * M_PI / (double)(1ul << phase_bits));
		ds = sin(a);
		ds *= ((1ul << output_width) - 1);
		if (ds < 0.5)
			break;
	}
	if (phase_bits < 3)
		phase_bits = 3;
	return (int)phase_bits;
}

/* ------------------------------------------------ configuration derivation */

static int is_p2r(int mode) { return mode == ORC_P2R || mode == ORC_SP2R; }

/* sw/basiccordic.cpp:67-73,471-496; sw/topolar.cpp:67-75,430-440;
 * sw/seqcordic.cpp:73-79,459-487; sw/seqpolar.cpp:73-80,396-410 */
int orc_config_core(orc_config *cfg, int mode, int nstages, int iw, int ow,
		int nxtra, int phase_bits)
{
	int ww;

	memset(cfg, 0, sizeof(*cfg));
	if (mode < ORC_P2R || mode > ORC_SR2P)
		return -1;
	if (iw <= 0 || ow <= 0 || iw > 32 || ow > 32)
		return -2;
	if (phase_bits < 3 || phase_bits > 32)
		return -3;
	if (nstages < 1 || nstages > ORC_MAX_STAGES)
		return -5;

	ww = iw;
	if (is_p2r(mode)) {
		if (nxtra < 1)
			nxtra = 1;
		if (ww < ow)
			ww = ow;
		ww += nxtra;
	} else {
		if (nxtra < 2)
			nxtra = 2;
		if (ww < ow)
			ww = ow;
		ww += nxtra;
		ww += nxtra;
	}
	if (ww > 64)
		return -4;
	if (mode == ORC_SP2R && (nstages < 2 || ww <= ow + 1))
		return -6;	/* emitted Verilog does not elaborate */
	if (mode == ORC_SR2P && ((nstages + 1) & nstages) == 0)
		return -6;	/* state register cannot reach NSTAGES+1 */

	cfg->mode = mode;
	cfg->iw = iw;
	cfg->ow = ow;
	cfg->nxtra = nxtra;
	cfg->ww = ww;
	cfg->pw = phase_bits;
	cfg->nstages = nstages;
	orc_cordic_angles(nstages, phase_bits, cfg->angle);
	cfg->quantization_variance = orc_transform_quantization_variance(
			nstages, ww - iw, ww - ow);
	cfg->phase_variance_rad = orc_phase_variance(nstages, phase_bits);
	if (is_p2r(mode)) {
		double amplitude, signal_energy, noise_energy;
		cfg->gain = orc_cordic_gain(nstages);
		/* sw/basiccordic.cpp:479-496 */
		amplitude = (1ul << (iw - 1)) - 1.;
		amplitude *= (1ul << ((ww - iw)));
		amplitude *= orc_cordic_gain(nstages);
		amplitude *= pow(2.0, -(ww - ow));
		signal_energy = amplitude * amplitude;
		noise_energy = orc_transform_quantization_variance(nstages,
				ww - iw, ww - ow);
		noise_energy += signal_energy
			* orc_phase_variance(nstages, phase_bits)
			* pow(2, orc_cordic_gain(nstages));
		cfg->best_possible_cnr = 10.0 * log(signal_energy / noise_energy)
						/ log(10.0);
	} else {
		/* sw/topolar.cpp:439-440 */
		cfg->gain = orc_cordic_gain(nstages) * sqrt(2.0) / 2.;
	}
	if (mode == ORC_SP2R)
		cfg->clocks_per_output = nstages + 1; /* sw/seqcordic.cpp:459 */
	if (mode == ORC_SR2P)
		cfg->clocks_per_output = nstages + 3; /* sw/seqpolar.cpp:396 */
	return 0;
}

/* sw/main.cpp:260-279 (p2r, sp2r) and :313-329 (r2p, sr2p) */
int orc_config_cli(orc_config *cfg, int mode, int iw, int ow, int xtra,
		int phase_bits, int nstages)
{
	const int DEFAULT_BITWIDTH = 24;
	int nxtra = xtra, ww;

	if (mode < ORC_P2R || mode > ORC_SR2P) {
		memset(cfg, 0, sizeof(*cfg));
		return -1;
	}
	if ((iw <= 0) && (ow > 0))
		iw = ow;
	if (ow <= 0)
		ow = iw;
	if ((iw <= 0) || (ow <= 0)) {
		iw = DEFAULT_BITWIDTH;
		ow = DEFAULT_BITWIDTH;
	}
	ww = (ow > iw) ? ow : iw;
	if (is_p2r(mode)) {
		nxtra += 1;
		ww += nxtra;
		if (ww > 63) {
			memset(cfg, 0, sizeof(*cfg));
			return -4;
		}
		if (phase_bits <= 0)
			phase_bits = orc_calc_phase_bits(ww);
		if (phase_bits > 32) {
			memset(cfg, 0, sizeof(*cfg));
			return -3;
		}
		if (nstages <= 0)
			nstages = orc_calc_stages2(ww, phase_bits);
	} else {
		nxtra += 2;
		ww += nxtra;
		if (ww > 63) {
			memset(cfg, 0, sizeof(*cfg));
			return -4;
		}
		if (phase_bits <= 0)
			phase_bits = orc_calc_phase_bits(ww);
		if (phase_bits > 32) {
			memset(cfg, 0, sizeof(*cfg));
			return -3;
		}
		if (nstages <= 0)
			nstages = orc_calc_stages1(phase_bits);
	}
	return orc_config_core(cfg, mode, nstages, iw, ow, nxtra, phase_bits);
}

/* ------------------------------------------------------ shared sub-steps */

/* Convergent rounding of a WW-bit value to OW bits.
 * rtl/cordic.v:288-295,311-312 (WW > OW+1) and the "No rounding required"
 * branch sw/basiccordic.cpp:407-444 (WW == OW+1: plain truncation). */
static int32_t round_out(int64_t v, int ww, int ow)
{
	int r = ww - ow;
	if (ww > ow + 1) {
		int64_t b = (v >> r) & 1;
		/* { OW zeros, b, (r-1) copies of !b } */
		int64_t add = (b << (r - 1))
			| (b ? 0 : (((int64_t)1 << (r - 1)) - 1));
		v = sx(v + add, ww);
	}
	return (int32_t)sx(v >> r, ow);
}

/* p2r input extension + octant fold: rtl/cordic.v:85-86,131-188
 * (sw/basiccordic.cpp:137-145,196-287); same text in rtl/seqcordic.v:83-84,
 * 124-182 */
static void p2r_prerotate(const orc_config *c, int32_t ix, int32_t iy,
		uint32_t iph, int64_t *x, int64_t *y, uint32_t *p)
{
	const int ww = c->ww, pw = c->pw;
	const uint32_t pm = pmask(pw);
	int64_t ex, ey;
	uint32_t ph = iph & pm;
	uint32_t q = 1u << (pw - 2);

	/* { sign, i_xval, (WW-IW-1) zeros } */
	ex = sx((int64_t)((uint64_t)sx(ix, c->iw) << (ww - c->iw - 1)), ww);
	ey = sx((int64_t)((uint64_t)sx(iy, c->iw) << (ww - c->iw - 1)), ww);

	switch ((ph >> (pw - 3)) & 7) {
	case 0: case 7:
		*x = ex; *y = ey; *p = ph;
		break;
	case 1: case 2:
		*x = sx(-ey, ww); *y = ex; *p = (ph - q) & pm;
		break;
	case 3: case 4:
		*x = sx(-ex, ww); *y = sx(-ey, ww); *p = (ph - 2 * q) & pm;
		break;
	default: /* 5, 6 */
		*x = ey; *y = sx(-ex, ww); *p = (ph - 3 * q) & pm;
		break;
	}
}

/* one p2r rotation, rtl/cordic.v:262-280 (shift and angle passed in so the
 * sequential core, rtl/seqcordic.v:270-291, can reuse it) */
static void p2r_rotate(int ww, uint32_t pm, int pw, unsigned shift,
		uint32_t ang, int64_t *x, int64_t *y, uint32_t *p)
{
	int64_t xo = *x, yo = *y;
	if ((*p >> (pw - 1)) & 1) {	/* negative phase */
		*x = sx(xo + asr(yo, shift), ww);
		*y = sx(yo - asr(xo, shift), ww);
		*p = (*p + ang) & pm;
	} else {
		*x = sx(xo - asr(yo, shift), ww);
		*y = sx(yo + asr(xo, shift), ww);
		*p = (*p - ang) & pm;
	}
}

/* r2p input extension + quadrant fold: rtl/topolar.v:83-84,122-152
 * (sw/topolar.cpp:139-151,208-251); WW-IW >= 4 always so only the first
 * sign-extension form of sw/topolar.cpp:139-151 is reachable */
static void r2p_prerotate(const orc_config *c, int32_t ix, int32_t iy,
		int64_t *x, int64_t *y, uint32_t *p)
{
	const int ww = c->ww, pw = c->pw;
	int64_t sxi = sx(ix, c->iw), syi = sx(iy, c->iw);
	int64_t ex, ey;
	uint32_t e = 1u << (pw - 3);

	ex = sx((int64_t)((uint64_t)sxi << (ww - c->iw - 2)), ww);
	ey = sx((int64_t)((uint64_t)syi << (ww - c->iw - 2)), ww);

	switch (((sxi < 0) ? 2 : 0) | ((syi < 0) ? 1 : 0)) {
	case 1:
		*x = sx(ex - ey, ww); *y = sx(ex + ey, ww); *p = 7 * e;
		break;
	case 2:
		*x = sx(-ex + ey, ww); *y = sx(-ex - ey, ww); *p = 3 * e;
		break;
	case 3:
		*x = sx(-ex - ey, ww); *y = sx(ex - ey, ww); *p = 5 * e;
		break;
	default:
		*x = sx(ex + ey, ww); *y = sx(-ex + ey, ww); *p = 1 * e;
		break;
	}
	*p &= pmask(pw);
}

/* one r2p rotation, rtl/topolar.v:226-243 / rtl/seqpolar.v:259-280 */
static void r2p_rotate(int ww, uint32_t pm, unsigned shift, uint32_t ang,
		int64_t *x, int64_t *y, uint32_t *p)
{
	int64_t xo = *x, yo = *y;
	if (yo < 0) {			/* yv[WW-1]: below the axis */
		*x = sx(xo - asr(yo, shift), ww);
		*y = sx(yo + asr(xo, shift), ww);
		*p = (*p - ang) & pm;
	} else {
		*x = sx(xo + asr(yo, shift), ww);
		*y = sx(yo - asr(xo, shift), ww);
		*p = (*p + ang) & pm;
	}
}

/* ------------------------------------------------------- pipelined cores */

/* rtl/cordic.v (= sw/basiccordic.cpp output) */
void orc_p2r(const orc_config *c, size_t n, const int32_t *xi,
		const int32_t *yi, int xy_stride, const uint32_t *phase,
		int32_t *ox, int32_t *oy)
{
	const uint32_t pm = pmask(c->pw);
	for (size_t s = 0; s < n; s++) {
		int64_t x, y;
		uint32_t p;
		size_t j = xy_stride ? s : 0;
		p2r_prerotate(c, xi[j], yi[j], phase[s], &x, &y, &p);
		for (int i = 0; i < c->nstages; i++) {
			/* rtl/cordic.v:253-261 */
			if ((c->angle[i] == 0) || (i >= c->ww))
				continue;
			p2r_rotate(c->ww, pm, c->pw, (unsigned)i + 1,
					c->angle[i], &x, &y, &p);
		}
		ox[s] = round_out(x, c->ww, c->ow);
		oy[s] = round_out(y, c->ww, c->ow);
	}
}

/* rtl/topolar.v (= sw/topolar.cpp output) */
void orc_r2p(const orc_config *c, size_t n, const int32_t *xi,
		const int32_t *yi, int32_t *omag, uint32_t *ophase)
{
	const uint32_t pm = pmask(c->pw);
	for (size_t s = 0; s < n; s++) {
		int64_t x, y;
		uint32_t p;
		r2p_prerotate(c, xi[s], yi[s], &x, &y, &p);
		for (int i = 0; i < c->nstages; i++) {
			/* rtl/topolar.v:217-225 */
			if ((c->angle[i] == 0) || (i >= c->ww))
				continue;
			r2p_rotate(c->ww, pm, (unsigned)i + 1, c->angle[i],
					&x, &y, &p);
		}
		omag[s] = round_out(x, c->ww, c->ow);	/* :251-255,268 */
		ophase[s] = p;				/* :269 */
	}
}

/* ------------------------------------------ sequential cores, closed form */

/* rtl/seqcordic.v: the capture at state >= NSTAGES-1 (:318-324) reads xv
 * before that edge's own rotation lands, and the first rotation happens
 * with state == 1, so exactly NSTAGES-2 rotations (shift = state = i+1,
 * cangle = cordic_angle[state-1] = angle[i]) are seen; no zero-angle skip. */
void orc_seq_p2r(const orc_config *c, size_t n, const int32_t *xi,
		const int32_t *yi, int xy_stride, const uint32_t *phase,
		int32_t *ox, int32_t *oy)
{
	const uint32_t pm = pmask(c->pw);
	for (size_t s = 0; s < n; s++) {
		int64_t x, y;
		uint32_t p;
		size_t j = xy_stride ? s : 0;
		p2r_prerotate(c, xi[j], yi[j], phase[s], &x, &y, &p);
		for (int i = 0; i < c->nstages - 2; i++)
			p2r_rotate(c->ww, pm, c->pw, (unsigned)i + 1,
					c->angle[i], &x, &y, &p);
		ox[s] = round_out(x, c->ww, c->ow);
		oy[s] = round_out(y, c->ww, c->ow);
	}
}

/* rtl/seqpolar.v: last_state = state >= NSTAGES+1 (:208), so all NSTAGES
 * rotations are seen; no zero-angle / i>=WW skip (:254-281). */
void orc_seq_r2p(const orc_config *c, size_t n, const int32_t *xi,
		const int32_t *yi, int32_t *omag, uint32_t *ophase)
{
	const uint32_t pm = pmask(c->pw);
	for (size_t s = 0; s < n; s++) {
		int64_t x, y;
		uint32_t p;
		r2p_prerotate(c, xi[s], yi[s], &x, &y, &p);
		for (int i = 0; i < c->nstages; i++)
			r2p_rotate(c->ww, pm, (unsigned)i + 1, c->angle[i],
					&x, &y, &p);
		omag[s] = round_out(x, c->ww, c->ow);
		ophase[s] = p;
	}
}

/* ------------------------------------- sequential cores, clock by clock */

/* rtl/seqcordic.v:124-324, every register updated from the pre-edge values
 * (non-blocking semantics).  The "mem" angle table has 2^nextlg(NSTAGES)
 * entries, all computed by the formula (sw/cordiclib.cpp:145-149). */
int orc_seq_p2r_cycle(const orc_config *c, int32_t ix, int32_t iy,
		uint32_t iph, int32_t *ox, int32_t *oy)
{
	const int ns = c->nstages, ww = c->ww, pw = c->pw;
	const uint32_t pm = pmask(pw);
	const int sbits = orc_nextlg((unsigned)ns);
	const unsigned smask = (1u << sbits) - 1u;
	const unsigned tlen = 1u << sbits;
	uint32_t table[128];
	/* registers */
	int64_t prex = 0, prey = 0, xv = 0, yv = 0;
	uint32_t preph = 0, ph = 0, cangle = 0;
	int idle = 1, pre_valid = 0, o_done = 0;
	unsigned state = 0;
	int32_t o_x = 0, o_y = 0;

	for (unsigned k = 0; k < tlen; k++)
		table[k] = angle_entry(k, pw);

	for (int tick = 1; tick <= 4 * ns + 16; tick++) {
		int i_stb = (tick == 1);
		/* next-state values */
		int64_t n_prex, n_prey, n_xv, n_yv;
		uint32_t n_preph, n_ph, n_cangle;
		int n_idle, n_pre_valid, n_o_done;
		unsigned n_state;
		int32_t n_ox = o_x, n_oy = o_y;

		p2r_prerotate(c, ix, iy, iph, &n_prex, &n_prey, &n_preph);

		if (i_stb)			n_idle = 0;
		else if (state == (unsigned)(ns - 1))	n_idle = 1;
		else				n_idle = idle;

		n_pre_valid = i_stb && idle;
		n_cangle = table[state & (tlen - 1)];

		if (idle)			n_state = 0;
		else if (state == (unsigned)(ns - 1))	n_state = 0;
		else				n_state = (state + 1) & smask;

		n_xv = xv; n_yv = yv; n_ph = ph;
		if (pre_valid) {
			n_xv = prex; n_yv = prey; n_ph = preph;
		} else
			p2r_rotate(ww, pm, pw, state, cangle,
					&n_xv, &n_yv, &n_ph);

		n_o_done = (state >= (unsigned)(ns - 1));
		if (state >= (unsigned)(ns - 1)) {
			n_ox = round_out(xv, ww, c->ow);
			n_oy = round_out(yv, ww, c->ow);
		}

		/* clock edge */
		prex = n_prex; prey = n_prey; preph = n_preph;
		idle = n_idle; pre_valid = n_pre_valid; cangle = n_cangle;
		state = n_state; xv = n_xv; yv = n_yv; ph = n_ph;
		o_done = n_o_done; o_x = n_ox; o_y = n_oy;

		if (o_done) {
			*ox = o_x; *oy = o_y;
			return tick;
		}
	}
	return -1;
}

/* rtl/seqpolar.v:121-307 */
int orc_seq_r2p_cycle(const orc_config *c, int32_t ix, int32_t iy,
		int32_t *omag, uint32_t *ophase)
{
	const int ns = c->nstages, ww = c->ww, pw = c->pw;
	const uint32_t pm = pmask(pw);
	const int sbits = orc_nextlg((unsigned)ns + 1);
	const unsigned smask = (1u << sbits) - 1u;
	const unsigned tlen = 1u << orc_nextlg((unsigned)ns);
	uint32_t table[128];
	int64_t prex = 0, prey = 0, xv = 0, yv = 0;
	uint32_t preph = 0, ph = 0, cangle = 0, o_ph = 0;
	int idle = 1, pre_valid = 0, o_done = 0;
	unsigned state = 0;
	int32_t o_m = 0;

	for (unsigned k = 0; k < tlen; k++)
		table[k] = angle_entry(k, pw);

	for (int tick = 1; tick <= 4 * ns + 16; tick++) {
		int i_stb = (tick == 1);
		int last_state = (state >= (unsigned)(ns + 1));
		int64_t n_prex, n_prey, n_xv, n_yv;
		uint32_t n_preph, n_ph, n_cangle, n_oph = o_ph;
		int n_idle, n_pre_valid, n_o_done;
		unsigned n_state;
		int32_t n_om = o_m;

		r2p_prerotate(c, ix, iy, &n_prex, &n_prey, &n_preph);

		if (i_stb)		n_idle = 0;
		else if (last_state)	n_idle = 1;
		else			n_idle = idle;

		n_pre_valid = i_stb && idle;

		if (idle)		n_state = 0;
		else if (last_state)	n_state = 0;
		else			n_state = (state + 1) & smask;

		n_cangle = table[state & (tlen - 1)];

		n_xv = xv; n_yv = yv; n_ph = ph;
		if (pre_valid) {
			n_xv = prex; n_yv = prey; n_ph = preph;
		} else
			r2p_rotate(ww, pm, state, cangle, &n_xv, &n_yv, &n_ph);

		n_o_done = last_state;
		if (last_state) {
			n_om = round_out(xv, ww, c->ow);
			n_oph = ph;
		}

		prex = n_prex; prey = n_prey; preph = n_preph;
		idle = n_idle; pre_valid = n_pre_valid; cangle = n_cangle;
		state = n_state; xv = n_xv; yv = n_yv; ph = n_ph;
		o_done = n_o_done; o_m = n_om; o_ph = n_oph;

		if (o_done) {
			*omag = o_m; *ophase = o_ph;
			return tick;
		}
	}
	return -1;
}

/* --------------------- sequential cores, whole port traces, register level
 *
 * rtl/seqcordic.v:100-324 / rtl/seqpolar.v:92-307 stepped over T clocks of
 * arbitrary i_stb / i_reset / i_aux / sample inputs, every register updated
 * from its pre-edge value.  Unlike the closed forms this also reproduces what
 * the cores do OFF protocol: an i_stb on the very clock that completes a sample
 * keeps `idle` low without loading the new sample (pre_valid needs idle), so
 * the free-running datapath goes round again over its own result and a second
 * o_done appears C-1 clocks later.  regs (in/out) carries the register file
 * from call to call; a zeroed struct with idle = 1 is the power-on state. */
void orc_seq_regs_init(orc_seq_regs *r)
{
	memset(r, 0, sizeof(*r));
	r->idle = 1;
}

void orc_seq_trace(const orc_config *c, size_t T, const uint8_t *stb,
		const uint8_t *rst, const uint8_t *aux, const int32_t *xi,
		const int32_t *yi, const uint32_t *phi, int32_t *o0, int32_t *o1,
		uint8_t *oaux, uint8_t *busy, uint8_t *done, orc_seq_regs *r)
{
	const int rot = (c->mode == ORC_SP2R);
	const int ns = c->nstages, ww = c->ww, pw = c->pw;
	const uint32_t pm = pmask(pw);
	/* state register width and table length as the emitters size them
	 * (sw/seqcordic.cpp / sw/seqpolar.cpp via nextlg) */
	const int sbits = orc_nextlg((unsigned)(rot ? ns : ns + 1));
	const unsigned smask = (1u << sbits) - 1u;
	const unsigned tlen = 1u << orc_nextlg((unsigned)ns);
	const unsigned last = (unsigned)(rot ? ns - 1 : ns + 1);
	uint32_t table[128];
	for (unsigned k = 0; k < tlen && k < 128; k++)
		table[k] = angle_entry(k, pw);

	for (size_t t = 0; t < T; t++) {
		const int i_stb = stb[t] != 0;
		const int i_rst = rst ? (rst[t] != 0) : 0;
		const int i_aux = aux ? (aux[t] != 0) : 0;
		const int at_last = rot ? (r->state >= last) : (r->state >= last);
		const int eq_last = rot ? (r->state == last) : at_last;
		orc_seq_regs n = *r;

		/* pre-rotation registers load on every clock */
		if (rot)
			p2r_prerotate(c, xi[t], yi[t], phi[t], &n.prex, &n.prey, &n.preph);
		else
			r2p_prerotate(c, xi[t], yi[t], &n.prex, &n.prey, &n.preph);

		if (i_rst)		n.aux = 0;
		else if (i_stb && r->idle) n.aux = i_aux;

		if (i_rst)		n.idle = 1;
		else if (i_stb)		n.idle = 0;
		else if (eq_last)	n.idle = 1;

		n.pre_valid = i_rst ? 0 : (i_stb && r->idle);

		if (i_rst)		n.state = 0;
		else if (r->idle)	n.state = 0;
		else if (eq_last)	n.state = 0;
		else			n.state = (r->state + 1) & smask;

		n.cangle = table[r->state & (tlen - 1)];

		if (r->pre_valid) {
			n.xv = r->prex; n.yv = r->prey; n.ph = r->preph;
		} else if (rot) {
			p2r_rotate(ww, pm, pw, r->state, r->cangle, &n.xv, &n.yv, &n.ph);
		} else {
			r2p_rotate(ww, pm, r->state, r->cangle, &n.xv, &n.yv, &n.ph);
		}

		n.o_done = i_rst ? 0 : at_last;
		if (at_last) {
			n.o0 = round_out(r->xv, ww, c->ow);
			n.o1 = rot ? round_out(r->yv, ww, c->ow) : (int32_t)r->ph;
			n.o_aux = r->aux;
		}
		*r = n;
		o0[t] = r->o0;
		o1[t] = r->o1;
		if (oaux) oaux[t] = (uint8_t)r->o_aux;
		if (busy) busy[t] = (uint8_t)!r->idle;
		if (done) done[t] = (uint8_t)r->o_done;
	}
}

/* ---------------------------------------------------------------- dispatch */

void orc_rotate(const orc_config *c, size_t n, const int32_t *x,
		const int32_t *y, int xy_stride, const uint32_t *phase,
		int32_t *ox, int32_t *oy)
{
	if (c->mode == ORC_SP2R)
		orc_seq_p2r(c, n, x, y, xy_stride, phase, ox, oy);
	else
		orc_p2r(c, n, x, y, xy_stride, phase, ox, oy);
}

void orc_topolar(const orc_config *c, size_t n, const int32_t *x,
		const int32_t *y, int32_t *omag, uint32_t *ophase)
{
	if (c->mode == ORC_SR2P)
		orc_seq_r2p(c, n, x, y, omag, ophase);
	else
		orc_r2p(c, n, x, y, omag, ophase);
}

void orc_nco(const orc_config *c, size_t n, uint32_t phase0, uint32_t fcw,
		uint64_t index0, int32_t x0, int32_t y0,
		int32_t *ox, int32_t *oy)
{
	const uint32_t pm = pmask(c->pw);
	for (size_t s = 0; s < n; s++) {
		uint32_t ph = (uint32_t)(phase0
				+ (uint32_t)(index0 + s) * fcw) & pm;
		orc_rotate(c, 1, &x0, &y0, 0, &ph, &ox[s], &oy[s]);
	}
}

/* The fused NCO MIXER (down-converter): the phase accumulator of orc_nco
 * (bench/cpp/cordic_tb.cpp:128-138) on the core's i_phase port while i_xval /
 * i_yval carry a sample stream (rtl/cordic.v:58-63 with all three ports
 * live).  Per sample it is orc_rotate, unchanged. */
void orc_mixer(const orc_config *c, size_t n, uint32_t phase0, uint32_t fcw,
		uint64_t index0, const int32_t *x, const int32_t *y,
		int32_t *ox, int32_t *oy)
{
	const uint32_t pm = pmask(c->pw);
	for (size_t s = 0; s < n; s++) {
		uint32_t ph = (uint32_t)(phase0
				+ (uint32_t)(index0 + s) * fcw) & pm;
		orc_rotate(c, 1, &x[s], &y[s], 1, &ph, &ox[s], &oy[s]);
	}
}

/* ------------------------------------------------------------ quadtbl core
 *
 * sw/quadtbl.cpp builds three tables in double precision; the statements
 * below are kept in the reference's evaluation order on purpose (an entry is
 * (long)(maxv * coefficient), so the doubles have to agree to the last bit;
 * compile with -ffp-contract=off).  Pinned by tests/golden/quad_golden.json:
 * the .hex files and header constants the real generator writes. */

static double q_sinc(double v)			/* sw/quadtbl.cpp:58-61 */
{
	double x = v * M_PI;
	return sin(x) / x;
}

static long q_max_integer(int width)		/* :63-65 */
{
	return (1l << (width - 1)) - 2l;
}

static double q_est_max_err(double c, double l, double q, double idx, int N)
{						/* :74-115 */
	double lft, rht, mid, ph, er;
	ph = 2.0 * M_PI * idx / (double)N;
	lft = c - sin(ph);
	ph = 2.0 * M_PI * (idx + 1) / (double)N;
	rht = c + l + q - sin(ph);
	mid = 0;
	for (int k = 0; k < 64; k++) {
		double mdx = k / 64.0;
		double mph = 2.0 * M_PI * (idx + mdx) / N;
		double mer = c + (l + q * mdx) * mdx - sin(mph);
		if (fabs(mer) > fabs(mid))
			mid = mer;
	}
	er = lft;
	if (fabs(er) < fabs(rht)) er = rht;
	if (fabs(er) < fabs(mid)) er = mid;
	return er;
}

static int q_pick_tbl_size(int ww)		/* :117-130 */
{
	double limit = pow(0.5, ww);
	for (int lgtbl = 4; lgtbl < 10; lgtbl++)
		if (pow(q_sinc(1.0 - (1. / (1 << lgtbl))), 3.) < limit)
			return lgtbl;
	return 11;
}

/* sw/quadtbl.cpp:132-279.  Returns 0, or -1 where the reference assert()s.
 * c/l/q may be NULL (widths and error only). */
static int q_build(int lgsz, int wid, int *cbits, int *lbits, int *qbits,
		double *tblerr, long *ct, long *lt, long *qt)
{
	if (lgsz <= 2 || wid <= 6)	/* assert(lgsz > 2); assert(wid > 6) */
		return -1;
	int ln = 1 << lgsz, rc = 0;
	long maxv = q_max_integer(wid);
	double dl = M_PI / (double)ln, dph = dl * 2.;
	double *table = calloc(ln, sizeof(double));
	double *slope = calloc(ln, sizeof(double));
	double *dslope = calloc(ln, sizeof(double));
	if (!table || !slope || !dslope) {
		free(table); free(slope); free(dslope);
		return -1;
	}
	for (int i = 0; i < ln; i++)
		table[i] = sin(dph * i + dl);
	for (int i = 1; i < ln - 1; i++)
		slope[i] = (table[i + 1] - table[i - 1]) / 2.0;
	slope[0] = (table[1] - table[ln - 1]) / 2.0;
	slope[ln - 1] = (table[0] - table[ln - 2]) / 2.0;
	for (int i = 1; i < ln - 1; i++)
		dslope[i] = -(table[i] - 0.5 * (table[i + 1] + table[i - 1]));
	dslope[0] = -(table[0] - 0.5 * (table[1] + table[ln - 1]));
	dslope[ln - 1] = -(table[ln - 1] - 0.5 * (table[0] + table[ln - 2]));
	for (int i = 0; i < ln; i++)
		table[i] = 0.75 * sin(dph * i + dl)
			+ (sin(dph * (i - 1) + dl) + sin(dph * (i + 1) + dl)) / 8.0;
	{
		const double del = 1.0, hlfdel = del / 2.0;
		for (int i = 0; i < ln; i++)
			table[i] = dslope[i] * hlfdel * hlfdel
				- slope[i] * hlfdel + table[i];
		for (int i = 0; i < ln; i++)
			slope[i] = slope[i] - del * dslope[i];
	}
	{
		double fctr = pow(1. / q_sinc(dl), 3);
		for (int i = 0; i < ln; i++) table[i] *= fctr;
		for (int i = 0; i < ln; i++) slope[i] *= fctr;
		for (int i = 0; i < ln; i++) dslope[i] *= fctr;
	}
	double mxtbl = 0.0, mxslope = 0.0, mxdslope = 0.0;
	for (int i = 0; i < ln; i++)
		mxtbl = (mxtbl > fabs(table[i])) ? mxtbl : fabs(table[i]);
	for (int i = 0; i < ln; i++) table[i] *= 1. / mxtbl;
	for (int i = 0; i < ln; i++) slope[i] *= 1. / mxtbl;
	for (int i = 0; i < ln; i++) dslope[i] *= 1. / mxtbl;

	double mxerr = 0.0;
	for (int i = 0; i < ln; i++) {
		double err = q_est_max_err(table[i], slope[i], dslope[i], i, ln);
		if (fabs(err) > fabs(mxerr))
			mxerr = err;
	}
	mxerr *= maxv;
	*tblerr = mxerr;

	mxtbl = 0.0;
	for (int i = 0; i < ln; i++)
		mxtbl = (mxtbl > fabs(table[i])) ? mxtbl : fabs(table[i]);
	for (int i = 0; i < ln; i++) {
		mxslope = (mxslope > fabs(slope[i])) ? mxslope : fabs(slope[i]);
		mxdslope = (mxdslope > fabs(dslope[i])) ? mxdslope : fabs(dslope[i]);
	}
	*cbits = wid + (int)ceil(log(mxtbl) / log(2.0));
	*lbits = wid + (int)ceil(-log(1. / mxslope) / log(2.0));
	*qbits = wid + (int)ceil(-log(1. / mxdslope) / log(2.0));

	/* the asserts of :237-241 and of hextable (sw/hexfile.cpp:52-59,81-84) */
	if (*cbits < wid || *cbits >= 31 || *lbits < 1 || *lbits >= 31
			|| *qbits < 1 || *qbits >= 31)
		rc = -1;
	for (int i = 0; !rc && i < ln; i++) {
		if (!(fabs(table[i]) <= (1 << (*cbits - wid)))) rc = -1;
		if (!(fabs(slope[i]) <= pow(2., (*lbits - wid)))) rc = -1;
		if (!(fabs(dslope[i]) <= pow(2., (*qbits - wid)))) rc = -1;
	}
	for (int k = 0; !rc && k < ln; k++) {
		long v[3] = { (long)(maxv * table[k]), (long)(maxv * slope[k]),
			      (long)(maxv * dslope[k]) };
		int b[3] = { *cbits, *lbits, *qbits };
		for (int j = 0; j < 3; j++) {
			long msk = (1l << b[j]) - 1l;
			if ((v[j] > 0) ? (v[j] > msk) : (v[j] < -msk - 1))
				rc = -1;
		}
		if (ct) { ct[k] = v[0]; lt[k] = v[1]; qt[k] = v[2]; }
	}
	free(table); free(slope); free(dslope);
	return rc;
}

/* sw/quadtbl.cpp:281-357: the emitter's view */
int orc_quad_core(orc_quad *q, int phase_bits, int ow, int nxtra)
{
	memset(q, 0, sizeof(*q));
	if (nxtra < 0 || ow < 3 || ow > 32)
		return -1;
	int wid = ow + nxtra;
	if (wid <= 6 || wid > 30)			/* assert(wid > 6) */
		return -1;
	if (phase_bits <= 4 || phase_bits > 32)		/* assert(phase_bits>4) */
		return -1;
	if (phase_bits <= q_pick_tbl_size(wid))		/* assert(phase_bits>lgtbl) */
		return -1;
	int lgtbl = 3, cbits, lbits, qbits;
	double tblerr;
	do {
		lgtbl++;
		if (q_build(lgtbl, wid, &cbits, &lbits, &qbits, &tblerr,
				NULL, NULL, NULL))
			return -2;
	} while ((fabs(tblerr) > 1.0) && (lgtbl < 20));
	if (nxtra < 2)
		nxtra = 2;
	q->pw = phase_bits; q->ow = ow; q->xtra = nxtra; q->wid = wid;
	q->ww = ow + nxtra;				/* localparam WW=(OW+XTRA) */
	q->lgtbl = lgtbl;
	q->dxbits = (phase_bits - lgtbl) + 1;
	q->cbits = cbits; q->lbits = lbits; q->qbits = qbits;
	q->scale = q_max_integer(ow);
	q->itbl_err = tblerr;
	q->tbl_err = tblerr * pow(0.5, ow + nxtra);
	{
		double spur = pow(q_sinc(1.0 - (1. / (1 << lgtbl))), 3.);
		q->spur_db = 20. * log(spur) / log(10.0);
	}
	/* part selects of rtl/quadtbl.v that must stay in range */
	if (q->dxbits < 2 || lbits - qbits - 1 < 0 || cbits - lbits - 1 < 0
			|| cbits < q->ww)
		return -2;
	return 0;
}

int orc_quad_cli(orc_quad *q, int iw, int ow, int xtra, int phase_bits)
{						/* sw/main.cpp:444-463 */
	int nxtra = xtra, ww;
	if ((iw <= 0) && (ow > 0)) iw = ow;
	if (ow <= 0) ow = iw;
	if ((iw <= 0) || (ow <= 0)) { iw = 24; ow = 24; }
	ww = (ow > iw) ? ow : iw;
	nxtra += 1;
	ww += nxtra;
	if (phase_bits <= 0) {
		if (ww < 1 || ww > 62)
			return -1;
		phase_bits = orc_calc_phase_bits(ww);
	}
	return orc_quad_core(q, phase_bits, ow, nxtra);
}

int orc_quad_tables(const orc_quad *q, long *ctbl, long *ltbl, long *qtbl)
{
	int c, l, qq;
	double e;
	return q_build(q->lgtbl, q->wid, &c, &l, &qq, &e, ctbl, ltbl, qtbl);
}

/* value of bits [hi:lo] of v */
static uint64_t q_bits(uint64_t v, int hi, int lo)
{
	return (v >> lo) & ((hi - lo + 1 >= 64) ? ~0ull
				: ((1ull << (hi - lo + 1)) - 1ull));
}

/* rtl/quadtbl.v:140-310, one sample, register by register */
void orc_quad_lookup(const orc_quad *q, const long *ctbl, const long *ltbl,
		const long *qtbl, size_t n, const uint32_t *phase, int32_t *out)
{
	const int PW = q->pw, OW = q->ow, XTRA = q->xtra, WW = q->ww;
	const int DX = q->dxbits, QB = q->qbits, LB = q->lbits, CB = q->cbits;
	for (size_t s = 0; s < n; s++) {
		const uint64_t ph = phase[s] & pmask(PW);
		/* clock 1 (:149-153) */
		const uint64_t idx = q_bits(ph, PW - 1, DX - 1);
		const int64_t qv = sx(qtbl[idx], QB), lv = sx(ltbl[idx], LB),
			cv = sx(ctbl[idx], CB);
		const int64_t dx = (int64_t)q_bits(ph, DX - 2, 0); /* {1'b0, ...} */
		/* clock 2 (:170): signed product in QBITS+DXBITS bits */
		const uint64_t qprod = (uint64_t)sx(qv * dx, QB + DX);
		/* clock 3 (:214-221) */
		uint64_t w_qprod = q_bits(qprod, QB + DX - 1, DX - 1);	/* [QBITS:0] */
		if (LB - QB - 1 > 0 && q_bits(qprod, QB + DX - 1, QB + DX - 1))
			w_qprod |= ((1ull << (LB - QB - 1)) - 1ull) << (QB + 1);
		const int64_t lsum = sx((int64_t)(w_qprod + (uint64_t)lv), LB);
		/* clock 4 (:246) */
		const uint64_t lprod = (uint64_t)sx(lsum * dx, LB + DX);
		/* clock 5 (:270-277) */
		uint64_t w_lprod = q_bits(lprod, LB + DX - 1, DX - 1);	/* [LBITS:0] */
		if (CB - LB - 1 > 0 && q_bits(lprod, LB + DX - 1, LB + DX - 1))
			w_lprod |= ((1ull << (CB - LB - 1)) - 1ull) << (LB + 1);
		const uint64_t r = (uint64_t)sx((int64_t)(w_lprod + (uint64_t)cv), CB)
			& ((1ull << CB) - 1ull);
		/* clock 6 (:292-300) */
		uint64_t w;
		if (!q_bits(r, WW - 1, WW - 1)
				&& q_bits(r, WW - 2, XTRA) == (1ull << (WW - 1 - XTRA)) - 1ull)
			w = r;
		else if (q_bits(r, WW - 1, WW - 2) == 3 && !q_bits(r, WW - 3, XTRA))
			w = r;
		else {
			const uint64_t b = q_bits(r, WW - OW, WW - OW);
			const uint64_t rest = b ? 0 : ((1ull << (WW - OW - 1)) - 1ull);
			w = r + ((b << (WW - OW - 1)) | rest);
		}
		w &= (1ull << WW) - 1ull;
		out[s] = (int32_t)sx((int64_t)q_bits(w, WW - 1, XTRA), OW); /* :308 */
	}
}


/* ------------------------------------------------------------ CPU baseline
 *
 * Timing harness for bench.py's cpu_baseline leg: `nthreads` POSIX threads
 * each push 2^16-sample blocks of the workload through the scalar oracle
 * until `seconds` have passed; returns the total number of samples done.
 * kind 0: rotator with constant (x0, y0), phase[i] = (g * phase_mul) mod 2^32
 * kind 1: converter on the I/Q ramps of SURVEY.md 8d config 3. */
#include <pthread.h>
#include <stdlib.h>
#include <time.h>

typedef struct {
	const orc_config *cfg;
	int kind, tid;
	uint32_t phase_mul;
	int32_t x0, y0;
	double deadline;
	uint64_t done;
} orc_job;

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *orc_worker(void *arg)
{
	orc_job *j = (orc_job *)arg;
	const size_t per = 1u << 16;
	uint32_t *ph = malloc(per * 4);
	int32_t *a = malloc(per * 4), *b = malloc(per * 4);
	int32_t *x = malloc(per * 4), *y = malloc(per * 4);
	const int sh = 32 - j->cfg->iw;
	for (size_t i = 0; i < per; i++) {
		uint32_t g = (uint32_t)((uint64_t)j->tid * per + i);
		ph[i] = g * j->phase_mul;
		x[i] = (int32_t)(((g * 0x9E3779B1u) >> 8) << sh) >> sh;
		y[i] = (int32_t)(((g * 0x85EBCA77u) >> 8) << sh) >> sh;
	}
	while (now_s() < j->deadline) {
		if (j->kind == 1)
			orc_topolar(j->cfg, per, x, y, a, (uint32_t *)b);
		else
			orc_rotate(j->cfg, per, &j->x0, &j->y0, 0, ph, a, b);
		j->done += per;
	}
	free(ph); free(a); free(b); free(x); free(y);
	return NULL;
}

uint64_t orc_throughput(const orc_config *cfg, int kind, int nthreads,
		double seconds, uint32_t phase_mul, int32_t x0, int32_t y0)
{
	if (nthreads < 1)
		nthreads = 1;
	pthread_t *th = malloc(sizeof(pthread_t) * (size_t)nthreads);
	orc_job *jobs = calloc((size_t)nthreads, sizeof(orc_job));
	const double deadline = now_s() + seconds;
	uint64_t total = 0;
	for (int t = 0; t < nthreads; t++) {
		jobs[t].cfg = cfg; jobs[t].kind = kind; jobs[t].tid = t;
		jobs[t].phase_mul = phase_mul; jobs[t].x0 = x0; jobs[t].y0 = y0;
		jobs[t].deadline = deadline;
		pthread_create(&th[t], NULL, orc_worker, &jobs[t]);
	}
	for (int t = 0; t < nthreads; t++) {
		pthread_join(th[t], NULL);
		total += jobs[t].done;
	}
	free(th); free(jobs);
	return total;
}

/* ---------------------------------------------------- whole-job digests
 *
 * The oracle's answer for EVERY sample of a synthetic job, condensed to the
 * 64-bit position-aware digest the device computes over its own outputs
 * (cordic_digest_u32, cordic_amd/csrc/cordic_kernels.hip: digest_mix):
 *     sum over g in [start, start+n) of mix(g, out0[g]) + mix(g + 2^40, out1[g])
 * so that a BASELINE-size run (2^30 .. 2^33 samples) is compared with the
 * oracle on 100 % of its outputs instead of on a strided subset.  The
 * per-sample arithmetic is orc_rotate / orc_topolar, unchanged (rtl/cordic.v:
 * 131-188,231-314, rtl/topolar.v:122-152,195-271); only the inputs are made
 * here, by the same rules as the device's fill kernels:
 *   kind 0  rotator, constant (x0, y0), phase[g] = phase0 + g*fcw mod 2^32
 *           (cfg2: fcw 4; cfg4: fcw 1; cfg5 NCO: fcw 0x01234567)
 *   kind 1  converter, x[g] = sext_iw(((uint32)g*mulx) >> 8), y likewise
 *   kind 2  rotator with the per-sample vectors of kind 1 and kind 0's phase
 * Threads draw 2^16-sample blocks from a shared counter. */
typedef struct {
	const orc_config *cfg;
	int kind;
	uint64_t start, n;
	uint32_t phase0, fcw, mulx, muly;
	int32_t x0, y0;
	uint64_t next;		/* shared block counter (atomic) */
	uint64_t sum;		/* per-thread result */
	void *shared;
} orc_djob;

static inline uint64_t orc_mix(uint64_t idx, uint32_t w)
{
	uint64_t z = (idx + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)w;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

uint64_t orc_digest_words(const uint32_t *w, size_t n, uint64_t index0)
{
	uint64_t s = 0;
	for (size_t i = 0; i < n; i++)
		s += orc_mix(index0 + i, w[i]);
	return s;
}

static void *orc_digest_worker(void *arg)
{
	orc_djob *j = (orc_djob *)arg;
	orc_djob *sh = (orc_djob *)j->shared;
	const size_t per = 1u << 16;
	uint32_t *ph = malloc(per * 4);
	int32_t *a = malloc(per * 4), *b = malloc(per * 4);
	int32_t *x = malloc(per * 4), *y = malloc(per * 4);
	const int shf = 32 - j->cfg->iw;
	uint64_t sum = 0;
	for (;;) {
		const uint64_t blk = __atomic_fetch_add(&sh->next, 1,
				__ATOMIC_RELAXED);
		const uint64_t off = blk * per;
		if (off >= j->n)
			break;
		const size_t cnt = (j->n - off < per) ? (size_t)(j->n - off) : per;
		const uint64_t g0 = j->start + off;
		for (size_t i = 0; i < cnt; i++) {
			const uint32_t g = (uint32_t)(g0 + i);
			ph[i] = j->phase0 + g * j->fcw;
			if (j->kind != 0) {
				x[i] = (int32_t)(((g * j->mulx) >> 8) << shf) >> shf;
				y[i] = (int32_t)(((g * j->muly) >> 8) << shf) >> shf;
			}
		}
		if (j->kind == 1)
			orc_topolar(j->cfg, cnt, x, y, a, (uint32_t *)b);
		else if (j->kind == 2)
			orc_rotate(j->cfg, cnt, x, y, 1, ph, a, b);
		else
			orc_rotate(j->cfg, cnt, &j->x0, &j->y0, 0, ph, a, b);
		sum += orc_digest_words((const uint32_t *)a, cnt, g0);
		sum += orc_digest_words((const uint32_t *)b, cnt,
				g0 + (1ull << 40));
	}
	j->sum = sum;
	free(ph); free(a); free(b); free(x); free(y);
	return NULL;
}

uint64_t orc_digest(const orc_config *cfg, int kind, int nthreads,
		uint64_t start, uint64_t n, uint32_t phase0, uint32_t fcw,
		int32_t x0, int32_t y0, uint32_t mulx, uint32_t muly,
		double *seconds)
{
	if (nthreads < 1)
		nthreads = 1;
	pthread_t *th = malloc(sizeof(pthread_t) * (size_t)nthreads);
	orc_djob *jobs = calloc((size_t)nthreads, sizeof(orc_djob));
	const double t0 = now_s();
	uint64_t total = 0;
	for (int t = 0; t < nthreads; t++) {
		jobs[t].cfg = cfg; jobs[t].kind = kind;
		jobs[t].start = start; jobs[t].n = n;
		jobs[t].phase0 = phase0; jobs[t].fcw = fcw;
		jobs[t].mulx = mulx; jobs[t].muly = muly;
		jobs[t].x0 = x0; jobs[t].y0 = y0;
		jobs[t].shared = &jobs[0];
		pthread_create(&th[t], NULL, orc_digest_worker, &jobs[t]);
	}
	for (int t = 0; t < nthreads; t++) {
		pthread_join(th[t], NULL);
		total += jobs[t].sum;
	}
	if (seconds)
		*seconds = now_s() - t0;
	free(th); free(jobs);
	return total;
}

/* ------------------------------------------------------------ table cores
 *
 * sintable / quarterwav (sw/sintable.cpp): plain table lookups, outside the
 * CORDIC hot path.  kind 4 = -t tbl, 5 = -t qtr. */

/* sw/main.cpp:330-368 (tbl) and :369-405 (qtr): PW / OW defaulting */
int orc_table_config(int kind, int iw, int ow, int phase_bits, int *pw_out,
		int *ow_out)
{
	if (kind != 4 && kind != 5)
		return -1;
	if (kind == 4) {
		if ((iw >= 0) && (phase_bits <= 0)) { phase_bits = iw; iw = -1; }
	} else {
		if ((iw >= 0) && (phase_bits < 0)) { phase_bits = iw; iw = -1; }
	}
	if ((phase_bits > 3) && (ow <= 0)) {
		for (int k = phase_bits - 2; k < phase_bits + 3; k++) {
			if (k < 1 || k > 62)
				continue;
			if (orc_calc_phase_bits(k) == phase_bits) { ow = k; break; }
		}
	}
	if (ow <= 0)
		ow = 24;
	if (phase_bits <= 0)
		phase_bits = orc_calc_phase_bits(ow);
	/* sw/hexfile.cpp:52-59, sw/sintable.cpp:58-66,186-194 */
	if (ow >= 31 || ow < 2 || phase_bits <= 2 || phase_bits >= 26)
		return -2;
	*pw_out = phase_bits;
	*ow_out = ow;
	return 0;
}

/* sw/sintable.cpp:155-166 (full wave) and :322-333 (quarter wave, half a
 * step of phase offset); entries as signed values */
void orc_table_values(int kind, int pw, int ow, int32_t *out)
{
	const int tbl_entries = (1 << pw);
	const long maxv = (1l << (ow - 1)) - 1l;
	if (kind == 4) {
		for (int k = 0; k < tbl_entries; k++) {
			double ph = 2.0 * M_PI * (double)k / (double)tbl_entries;
			long v = (long)((double)(long)maxv * sin(ph));
			out[k] = (int32_t)v;
		}
	} else {
		for (int k = 0; k < tbl_entries / 4; k++) {
			double ph = 2.0 * M_PI * (double)k / (double)tbl_entries;
			ph += M_PI / (double)tbl_entries;
			long v = (long)((double)maxv * sin(ph));
			out[k] = (int32_t)v;
		}
	}
}

/* rtl/sintable.v:72-77 and rtl/quarterwav.v:86-108 per sample */
void orc_table_lookup(int kind, int pw, int ow, const int32_t *tbl, size_t n,
		const uint32_t *phase, int32_t *out)
{
	const uint32_t pm = pmask(pw);
	for (size_t s = 0; s < n; s++) {
		const uint32_t ph = phase[s] & pm;
		if (kind == 4) {
			out[s] = (int32_t)sx(tbl[ph], ow);
		} else {
			const uint32_t qm = (1u << (pw - 2)) - 1u;
			uint32_t idx = ph & qm;
			if ((ph >> (pw - 2)) & 1)
				idx = (~ph) & qm;
			int64_t v = sx(tbl[idx], ow);
			if ((ph >> (pw - 1)) & 1)
				v = -v;
			out[s] = (int32_t)sx(v, ow);
		}
	}
}
