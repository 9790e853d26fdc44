// cordic_config.cpp -- host side of the engine: parameter derivation, arctan
// table, quality constants and the constants-header text.  Pure host C++ (no
// HIP): everything here runs once per generated core, never per sample.
//
// The numbers must equal what the reference generator prints, so the
// floating-point expressions keep the reference's operation order (cited per
// function); the structure around them is this project's own.
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "cordic_amd.h"
#include "cordic_internal.h"

namespace cordic_amd {

// ---------------------------------------------------------------------------
// Table math (reference sw/cordiclib.cpp)
// ---------------------------------------------------------------------------

// cordic_angle[k] in PW-bit phase units.  Reference sw/cordiclib.cpp:161-169:
//   x = atan2(1., pow(2,k+1));  x *= (4.0*(1ul<<(PW-2))) / (M_PI*2.0);
//   phase_value = (unsigned)x;
// The scale factor is evaluated first and x multiplied by it, as there.
uint32_t arctan_entry(unsigned k, int phase_bits)
{
	// the library helpers (cordic_angles, cordic_calc_stages) are callable
	// with any integer; the reference shifts by PW-2 unguarded
	if (phase_bits < 2 || phase_bits > 62)
		return 0;
	const double scale = (4.0 * (double)(1ul << (phase_bits - 2)))
				/ (M_PI * 2.0);
	double x = std::atan2(1., std::pow(2, k + 1));
	x *= scale;
	return (uint32_t)(unsigned)x;
}

// Product over the stages of sqrt(1 + 2^-2(k+1)); sw/cordiclib.cpp:66-80.
double rotation_gain(int nstages)
{
	double g = 1.0;
	// the factor is exactly 1.0 from k = 27 on: no need to follow a huge count
	for (int k = 0; k < nstages && k < 1024; k++)
		g = g * std::sqrt(1.0 + std::pow(2.0, -2. * (k + 1)));
	return g;
}

// "You can annihilate this gain by multiplying by 32'h%08x and right shifting
// by 32 bits": sw/cordiclib.cpp:205-209.
uint32_t gain_annihilator(int nstages)
{
	return (unsigned)(1.0 / rotation_gain(nstages) * (4.0 * (1ul << 30)));
}

// The constant as the generator prints it for this core: the sequential
// emitters pad the table to a power of two before the comment block is
// written (sw/cordiclib.cpp:145-149 via sw/seqcordic.cpp:285,
// sw/seqpolar.cpp:237), the pipelined ones do not.
uint32_t core_gain_annihilator(const cordic_config &c)
{
	int n = c.nstages;
	if (c.mode == CORDIC_SP2R || c.mode == CORDIC_SR2P)
		n = 1 << next_lg((unsigned)c.nstages);
	return gain_annihilator(n);
}

// Variance (radians^2) of the truncated angle table plus the initial 1/12;
// sw/cordiclib.cpp:82-109.
double phase_variance(int nstages, int phase_bits)
{
	// exported as a library helper: any integers may arrive
	if (phase_bits < 1 || phase_bits > 63)
		return std::nan("");
	if (nstages < 0)
		nstages = 0;
	const double rad_to_phase = (double)(1ul << (phase_bits - 1)) / M_PI;
	double var = 1. / 12.;
	for (unsigned k = 0; k < (unsigned)nstages; k++) {
		const double x = std::atan2(1., std::pow(2, k + 1)) * rad_to_phase;
		const unsigned long q = (unsigned)x;
		double e = (double)q - x;
		e *= e;
		var += e;
	}
	var /= std::pow(rad_to_phase, 2.);
	return var;
}

// Rounding-noise model of the x/y datapath; sw/cordiclib.cpp:111-130.
double quantization_variance(int nstages, int xtrabits, int dropped_bits)
{
	double v = std::pow(2, 2 * xtrabits) / 12.;
	for (int k = 0; k < nstages && k < 4096; k++)
		v = (1 + std::pow(4, -k - 1)) * v + 1. / 3.;
	if (dropped_bits > 0)
		v = std::pow(2, -2 * dropped_bits) * v + 1 / 12.;
	return v;
}

// ceil(log2(vl)); sw/cordiclib.cpp:57-63.
int next_lg(unsigned vl)
{
	int lg = 0;
	for (unsigned r = 1; r != 0 && r < vl; r <<= 1)
		lg++;
	return lg;
}

// Stage count at which the table runs out (and, when a working width is
// given, at which the shifts run out); sw/cordiclib.cpp:214-244.
int stages_for(int phase_bits, int working_width /* <0: unbounded */)
{
	int n = 0;
	for (; n < 64; n++) {
		if (arctan_entry((unsigned)n, phase_bits) == 0)
			break;
		if (working_width >= 0 && working_width <= n)
			break;
	}
	return n;
}

// Smallest PW >= 3 with sin(2pi/2^PW) * (2^w - 1) < 1/2; sw/cordiclib.cpp:
// 254-263 (the code uses 2^w - 1, not the 2^(w-1) - 1 of its comment).
int phase_bits_for(int width)
{
	if (width < 0) width = 0;
	if (width > 62) width = 62;
	int pb = 3;
	for (; pb < 64; pb++) {
		const double a = (2.0 * M_PI / (double)(1ul << pb));
		double ds = std::sin(a);
		ds *= (double)((1ul << width) - 1);
		if (ds < 0.5)
			break;
	}
	return pb;
}

// ---------------------------------------------------------------------------
// Overflow reachability.  A kernel whose container is wider than WW only
// matches the WW-bit registers of the generated core if no intermediate value
// can leave the WW-bit range.  Bound the vector magnitude stage by stage: an
// exact micro-rotation scales it by sqrt(1+4^-k) and the two floor()s move
// the point by less than sqrt(2).
// ---------------------------------------------------------------------------
static bool overflow_reachable(const cordic_config &c)
{
	const bool p2r = (c.mode == CORDIC_P2R || c.mode == CORDIC_SP2R);
	// |e_x|,|e_y| <= 2^(WW-2) (p2r) or 2^(WW-3) followed by the x+-y fold
	double m = p2r ? std::sqrt(2.0) * std::ldexp(1.0, c.ww - 2)
		       : std::ldexp(1.0, c.ww - 2);
	for (int i = 0; i < c.nlive; i++)
		m = m * std::sqrt(1.0 + std::ldexp(1.0, -2 * (i + 1)))
			+ std::sqrt(2.0);
	const int r = c.ww - c.ow;
	const double round_add = (c.ww > c.ow + 1) ? std::ldexp(1.0, r - 1) : 0.0;
	return (m + round_add + 1.0) >= std::ldexp(1.0, c.ww - 1);
}

// ---------------------------------------------------------------------------
// Core construction
// ---------------------------------------------------------------------------

static inline bool mode_ok(int m) { return m >= CORDIC_P2R && m <= CORDIC_SR2P; }
static inline bool is_rotator(int m) { return m == CORDIC_P2R || m == CORDIC_SP2R; }

int build_core(cordic_config *cfg, int mode, int nstages, int iw, int ow,
		int nxtra, int phase_bits)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	std::memset(cfg, 0, sizeof(*cfg));
	if (!mode_ok(mode))
		return CORDIC_ERR_MODE;
	if (iw < 1 || iw > 32 || ow < 1 || ow > 32)
		return CORDIC_ERR_WIDTH;
	if (phase_bits < 3 || phase_bits > 32)
		return CORDIC_ERR_PHASE_BITS;
	if (nstages < 1 || nstages > CORDIC_AMD_MAX_STAGES)
		return CORDIC_ERR_STAGES;

	// Working width.  Rotators: sw/basiccordic.cpp:67-73 (one guard bit at
	// least).  Converters: sw/topolar.cpp:67-75 adds nxtra twice.
	const int wide = (iw > ow) ? iw : ow;
	int ww;
	if (is_rotator(mode)) {
		if (nxtra < 1) nxtra = 1;
		ww = wide + nxtra;
	} else {
		if (nxtra < 2) nxtra = 2;
		ww = wide + 2 * nxtra;
	}
	if (ww > 64)
		return CORDIC_ERR_WORKING_WIDTH;

	// Cores the reference cannot really build:
	//  - sp2r with WW == OW+1 emits an "if (i_ce)" with no such port
	//    (sw/seqcordic.cpp:400-417); with NSTAGES < 2 a zero-width state.
	//  - sr2p waits for state >= NSTAGES+1 in a nextlg(NSTAGES+1)-bit
	//    register (sw/seqpolar.cpp:159,239): unreachable if NSTAGES+1 is a
	//    power of two, so o_done never rises.
	if (mode == CORDIC_SP2R && (nstages < 2 || ww <= ow + 1))
		return CORDIC_ERR_UNSUPPORTED;
	if (mode == CORDIC_SR2P && ((nstages + 1) & nstages) == 0)
		return CORDIC_ERR_UNSUPPORTED;

	cfg->mode = mode;
	cfg->iw = iw;
	cfg->ow = ow;
	cfg->nextra = nxtra;
	cfg->ww = ww;
	cfg->pw = phase_bits;
	cfg->nstages = nstages;
	cfg->has_reset = 1;	// reference defaults, sw/main.cpp:99
	cfg->has_aux = 0;
	cfg->async_reset = 0;
	for (int k = 0; k < nstages; k++)
		cfg->angle[k] = arctan_entry((unsigned)k, phase_bits);

	cfg->quantization_variance = quantization_variance(nstages, ww - iw,
							ww - ow);
	cfg->phase_variance_rad = phase_variance(nstages, phase_bits);
	if (is_rotator(mode)) {
		cfg->gain = rotation_gain(nstages);
		// sw/basiccordic.cpp:479-496
		double amplitude = (double)(1ul << (iw - 1)) - 1.;
		amplitude *= (double)(1ul << (ww - iw));
		amplitude *= rotation_gain(nstages);
		amplitude *= std::pow(2.0, -(ww - ow));
		const double signal = amplitude * amplitude;
		double noise = quantization_variance(nstages, ww - iw, ww - ow);
		noise += signal * phase_variance(nstages, phase_bits)
				* std::pow(2, rotation_gain(nstages));
		cfg->best_possible_cnr = 10.0 * std::log(signal / noise)
						/ std::log(10.0);
	} else {
		cfg->gain = rotation_gain(nstages) * std::sqrt(2.0) / 2.;
	}

	// How many rotations the core really performs.
	switch (mode) {
	case CORDIC_P2R:
	case CORDIC_R2P: {
		// rtl/cordic.v:253-261, rtl/topolar.v:217-225: a stage whose
		// angle is zero or whose index reaches WW only copies.  Angles
		// never increase with k, so the live stages are a prefix.
		int live = 0;
		while (live < nstages && cfg->angle[live] != 0 && live < ww)
			live++;
		cfg->nlive = live;
		break;
	}
	case CORDIC_SP2R:
		// rtl/seqcordic.v:318-324 captures before the last two
		// rotations land; no copy-only stages (:270-291).
		cfg->nlive = nstages - 2;
		cfg->clocks_per_output = nstages + 1; // sw/seqcordic.cpp:459
		break;
	default:
		// rtl/seqpolar.v:208 runs every stage; no copy-only stages.
		cfg->nlive = nstages;
		cfg->clocks_per_output = nstages + 3; // sw/seqpolar.cpp:396
		break;
	}
	cfg->needs_wrap = overflow_reachable(*cfg) ? 1 : 0;
	return CORDIC_OK;
}

// gencordic's defaulting (sw/main.cpp:260-279 rotators, :313-329 converters)
int build_from_cli(cordic_config *cfg, int mode, int iw, int ow, int xtra,
		int phase_bits, int nstages)
{
	if (!cfg)
		return CORDIC_ERR_ARGS;
	if (!mode_ok(mode)) {
		std::memset(cfg, 0, sizeof(*cfg));
		return CORDIC_ERR_MODE;
	}
	if (iw <= 0 && ow > 0) iw = ow;
	if (ow <= 0) ow = iw;
	if (iw <= 0 || ow <= 0) iw = ow = 24;	// DEFAULT_BITWIDTH

	const int bump = is_rotator(mode) ? 1 : 2;
	const int nxtra = xtra + bump;
	const int ww_cli = ((ow > iw) ? ow : iw) + nxtra;
	if (ww_cli > 63 || ww_cli < 1) {
		std::memset(cfg, 0, sizeof(*cfg));
		return CORDIC_ERR_WORKING_WIDTH;
	}
	if (phase_bits <= 0)
		phase_bits = phase_bits_for(ww_cli);
	if (phase_bits > 32) {
		std::memset(cfg, 0, sizeof(*cfg));
		return CORDIC_ERR_PHASE_BITS;
	}
	if (nstages <= 0)
		nstages = is_rotator(mode) ? stages_for(phase_bits, ww_cli)
					   : stages_for(phase_bits, -1);
	return build_core(cfg, mode, nstages, iw, ow, nxtra, phase_bits);
}

// ---------------------------------------------------------------------------
// gencordic argv front end (sw/main.cpp:139-232)
// ---------------------------------------------------------------------------
int parse_args(cordic_config *cfg, int argc, const char *const *argv,
		char *fname, size_t fname_cap, int *c_header)
{
	if (!cfg || argc < 1 || !argv)
		return CORDIC_ERR_ARGS;
	int nstages = -1, iw = -1, ow = -1, xtra = 2, pw = -1;
	// sw/main.cpp:99-103: the generator's own flag set.  Every -t clears the
	// first four and sets one; `sequential` and gen_quadtbl are only ever
	// set, never cleared ("-t sr2p -t p2r" generates the SEQUENTIAL rotator).
	bool polar_to_rect = false, rect_to_polar = true, gen_sintable = false,
	     gen_quarterwav = false, gen_quadtbl = false, sequential = false;
	bool reset = true, aux = false, areset = false, hdr = false;
	// fname: -f always sets it, a -t only while it is still unset
	// (sw/main.cpp:151-153,183-211), so the FIRST -t names the default file
	bool have_file = false;
	std::string file;

	// getopt(3) as glibc runs it for "aAcf:hi:n:o:p:Rrt:vx:": flags may be
	// bundled ("-vca"), values glued ("-i13") or separate ("-i 13", the next
	// word whatever it looks like); words that are not options are skipped
	// (glibc permutes them to the end) and "--" ends the options.
	for (int k = 1; k < argc; k++) {
		const char *a = argv[k];
		if (!a)
			return CORDIC_ERR_ARGS;
		if (a[0] != '-' || a[1] == '\0')
			continue;		// not an option: skipped, not a stop
		if (a[1] == '-' && a[2] == '\0')
			break;
		for (const char *p = a + 1; *p; p++) {
			const char f = *p;
			if (std::strchr("finoptx", f)) {
				const char *val = p[1] ? p + 1
					: (k + 1 < argc ? argv[++k] : nullptr);
				if (!val)
					return CORDIC_ERR_ARGS;
				switch (f) {
				case 'f': file = val; have_file = true; break;
				case 'i': iw = std::atoi(val); break;
				case 'n': nstages = std::atoi(val); break;
				case 'o': ow = std::atoi(val); break;
				case 'p': pw = std::atoi(val); break;
				case 'x': xtra = std::atoi(val); break;
				case 't': {
					const char *def = nullptr;
					rect_to_polar = polar_to_rect = false;
					gen_sintable = gen_quarterwav = false;
					if (!std::strcmp(val, "r2p")) {
						def = "topolar.v"; rect_to_polar = true;
					} else if (!std::strcmp(val, "sr2p")) {
						def = "seqpolar.v"; rect_to_polar = true;
						sequential = true;
					} else if (!std::strcmp(val, "p2r")) {
						def = "basiccordic.v"; polar_to_rect = true;
					} else if (!std::strcmp(val, "sp2r")) {
						def = "seqcordic.v"; polar_to_rect = true;
						sequential = true;
					} else if (!std::strcmp(val, "tbl")) {
						def = "sintable.v"; gen_sintable = true;
					} else if (!std::strcmp(val, "qtr")) {
						def = "quarterwav.v"; gen_quarterwav = true;
					} else if (!std::strcmp(val, "qtbl")) {
						def = "quadtbl.v"; gen_quadtbl = true;
					} else
						return CORDIC_ERR_MODE;
					if (!have_file) {
						file = def;
						have_file = true;
					}
					break;
				}
				}
				break;		// value consumed the rest of a
			}
			switch (f) {
			case 'a': aux = true; break;
			case 'A': areset = true; reset = true; break;
			case 'c': hdr = true; break;
			case 'R': reset = false; break;
			case 'r': reset = true; break;
			case 'v': break;
			// usage() and exit: the generator writes no core
			case 'h': return CORDIC_ERR_ARGS;
			default: return CORDIC_ERR_ARGS;
			}
		}
	}
	// sw/main.cpp:260,312: which CORDIC core this command line generates
	// (table generators are cordic_table_* / cordic_quad_* territory)
	int mode;
	if (polar_to_rect)
		mode = sequential ? CORDIC_SP2R : CORDIC_P2R;
	else if (rect_to_polar)
		mode = sequential ? CORDIC_SR2P : CORDIC_R2P;
	else
		return CORDIC_ERR_MODE;
	(void)gen_sintable; (void)gen_quarterwav; (void)gen_quadtbl;
	const char *deffile = "topolar.v";	// no -t, no -f: stdout in the
						// generator; the r2p default here
	int rc = build_from_cli(cfg, mode, iw, ow, xtra, pw, nstages);
	if (rc != CORDIC_OK)
		return rc;
	cfg->has_reset = reset;
	cfg->has_aux = aux;
	cfg->async_reset = areset;
	if (c_header) *c_header = hdr;
	if (fname && fname_cap) {
		const std::string &f = have_file ? file : std::string(deffile);
		std::snprintf(fname, fname_cap, "%s", f.c_str());
	}
	return CORDIC_OK;
}

// ---------------------------------------------------------------------------
// Constants header text
// ---------------------------------------------------------------------------
namespace {
struct TextSink {
	char *buf; size_t cap; size_t len = 0;
	void put(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
	{
		char tmp[256];
		va_list ap;
		va_start(ap, fmt);
		int m = std::vsnprintf(tmp, sizeof(tmp), fmt, ap);
		va_end(ap);
		for (int i = 0; i < m; i++, len++)
			if (buf && len + 1 < cap)
				buf[len] = tmp[i];
	}
	void finish() { if (buf && cap) buf[len < cap ? len : cap - 1] = '\0'; }
};
}

int write_header(const cordic_config *c, const char *name, char *buf, size_t cap)
{
	if (!c || !name || !mode_ok(c->mode))
		return CORDIC_ERR_ARGS;
	// guard = "<name>.h" upper-cased with '.' -> '_'
	std::string guard = std::string(name) + ".h";
	for (auto &ch : guard)
		ch = (ch == '.') ? '_' : (char)std::toupper((unsigned char)ch);

	const bool rot = is_rotator(c->mode);
	const bool seq = (c->mode == CORDIC_SP2R || c->mode == CORDIC_SR2P);
	TextSink o{buf, cap};
	o.put("#ifndef\t%s\n#define\t%s\n", guard.c_str(), guard.c_str());
	if (c->async_reset)
		o.put("#define\tASYNC_RESET\n");
	if (seq) {
		o.put("#ifdef\tCLOCKS_PER_OUTPUT\n#undef\tCLOCKS_PER_OUTPUT\n"
		      "#endif\t// CLOCKS_PER_OUTPUT\n");
		// the rotator's line is followed by a blank line
		o.put("#define\tCLOCKS_PER_OUTPUT\t%d\n%s", c->clocks_per_output,
			rot ? "\n" : "");
	}
	o.put("const int\tIW = %d;\n", c->iw);
	o.put("const int\tOW = %d;\n", c->ow);
	o.put("const int\tNEXTRA = %d;\n", c->nextra);
	o.put("const int\tWW = %d;\n", c->ww);
	o.put("const int\tPW = %d;\n", c->pw);
	o.put("const int\tNSTAGES = %d;\n", c->nstages);
	if (rot) {
		o.put("const double\tQUANTIZATION_VARIANCE = %.4e; // (Units^2)\n",
			c->quantization_variance);
		o.put("const double\tPHASE_VARIANCE_RAD = %.4e; // (Radians^2)\n",
			c->phase_variance_rad);
		o.put("const double\tGAIN = %.16f;\n", c->gain);
		o.put("const double\tBEST_POSSIBLE_CNR = %.2f;\n",
			c->best_possible_cnr);
	} else {
		o.put("const double\tQUANTIZATION_VARIANCE = %.16f; // (Units^2)\n",
			c->quantization_variance);
		o.put("const double\tPHASE_VARIANCE_RAD = %.16f; // (Radians^2)\n",
			c->phase_variance_rad);
		o.put("const double\tGAIN = %.16f;\n", c->gain);
	}
	o.put("const bool\tHAS_RESET = %s;\n", c->has_reset ? "true" : "false");
	o.put("const bool\tHAS_AUX   = %s;\n", c->has_aux ? "true" : "false");
	if (c->has_reset) o.put("#define\tHAS_RESET_WIRE\n");
	if (c->has_aux)   o.put("#define\tHAS_AUX_WIRES\n");
	o.put("#endif\t// %s\n", guard.c_str());
	o.finish();
	return (int)o.len;
}

// ---------------------------------------------------------------------------
// Table cores (sw/sintable.cpp), row F4
// ---------------------------------------------------------------------------
bool config_sane(const cordic_config &c)
{
	if (c.mode < CORDIC_P2R || c.mode > CORDIC_SR2P)
		return false;
	const bool rot = (c.mode == CORDIC_P2R || c.mode == CORDIC_SP2R);
	const int in_shl = rot ? (c.ww - c.iw - 1) : (c.ww - c.iw - 2);
	return c.iw >= 1 && c.iw <= 32 && c.ow >= 1 && c.ow <= 32
		&& c.ww >= c.ow && c.ww <= 64 && in_shl >= 0
		&& c.pw >= 3 && c.pw <= 32
		&& c.nstages >= 1 && c.nstages <= CORDIC_AMD_MAX_STAGES
		&& c.nlive >= 0 && c.nlive <= c.nstages;
}

// sw/sintable.cpp:186-194 / sw/hexfile.cpp:52-59 bounds, and the table length
// the kernels index with (table_lookup: ph & (2^pw - 1), 2^(pw-2) per quadrant)
bool table_sane(const cordic_table_config &t)
{
	if (t.kind != CORDIC_TBL && t.kind != CORDIC_QTR)
		return false;
	if (t.pw < 3 || t.pw > 25 || t.ow < 2 || t.ow > 30)
		return false;
	return t.entries == ((t.kind == CORDIC_TBL) ? (1 << t.pw)
						   : (1 << (t.pw - 2)));
}

// rtl/quadtbl.v:149-153,214-216,244-277 widths as quad_build_core derives them;
// the kernel copies 2^lgtbl entries into 16 * entries bytes of LDS and shifts
// by these widths
bool quad_sane(const cordic_quad_config &q)
{
	return q.lgtbl >= 4 && q.lgtbl <= 12 && q.entries == (1 << q.lgtbl)
		&& q.pw >= q.lgtbl + 1 && q.pw <= 32
		&& q.dxbits == q.pw - q.lgtbl + 1 && q.dxbits >= 2
		&& q.ow >= 2 && q.xtra >= 2 && q.ww == q.ow + q.xtra && q.ww <= 32
		&& q.tbl_width > 6 && q.tbl_width <= 30
		&& q.qbits >= 1 && q.lbits > q.qbits && q.cbits > q.lbits
		&& q.cbits >= q.ww && q.cbits <= 31;
}

int table_derive(cordic_table_config *t, int kind, int iw, int ow, int pw)
{
	if (!t)
		return CORDIC_ERR_ARGS;
	std::memset(t, 0, sizeof(*t));
	if (kind != CORDIC_TBL && kind != CORDIC_QTR)
		return CORDIC_ERR_MODE;
	// sw/main.cpp:332-335 (tbl tests pw <= 0) vs :371-374 (qtr tests pw < 0)
	const bool pw_absent = (kind == CORDIC_TBL) ? (pw <= 0) : (pw < 0);
	if (iw >= 0 && pw_absent)
		pw = iw;
	if (pw > 3 && ow <= 0) {
		for (int k = pw - 2; k < pw + 3; k++)
			if (k >= 1 && k <= 62 && phase_bits_for(k) == pw) {
				ow = k;
				break;
			}
	}
	if (ow <= 0)
		ow = 24;
	if (pw <= 0)
		pw = phase_bits_for(ow);
	// sw/hexfile.cpp:52-59 (ow < 31, >= 4 entries); sw/sintable.cpp:186-194
	// refuses tables above 2^25 entries
	if (ow >= 31 || ow < 2)
		return CORDIC_ERR_WIDTH;
	if (pw <= 2 || pw >= 26)
		return CORDIC_ERR_PHASE_BITS;
	t->kind = kind;
	t->pw = pw;
	t->ow = ow;
	t->entries = (kind == CORDIC_TBL) ? (1 << pw) : (1 << (pw - 2));
	return CORDIC_OK;
}

int table_fill(const cordic_table_config &t, int32_t *out, size_t cap)
{
	if (!out || !table_sane(t) || cap < (size_t)t.entries)
		return CORDIC_ERR_ARGS;
	const int n = 1 << t.pw;
	const long maxv = (1l << (t.ow - 1)) - 1l;
	for (int k = 0; k < t.entries; k++) {
		double ph = 2.0 * M_PI * (double)k / (double)n;
		if (t.kind == CORDIC_QTR)
			ph += M_PI / (double)n;	// half a step: sw/sintable.cpp:330
		out[k] = (int32_t)(long)((double)maxv * std::sin(ph));
	}
	return CORDIC_OK;
}

const char *status_text(int s)
{
	switch (s) {
	case CORDIC_OK:			return "ok";
	case CORDIC_ERR_MODE:		return "unsupported cordic mode";
	case CORDIC_ERR_WIDTH:		return "input/output width outside 1..32";
	case CORDIC_ERR_PHASE_BITS:	return "phase bits outside 3..32";
	case CORDIC_ERR_WORKING_WIDTH:	return "working width above 64 bits";
	case CORDIC_ERR_STAGES:		return "stage count outside 1..64";
	case CORDIC_ERR_UNSUPPORTED:	return "the reference core for these parameters cannot be built or never completes";
	case CORDIC_ERR_ARGS:		return "bad argument";
	case CORDIC_ERR_DEVICE:		return "HIP runtime error";
	case CORDIC_ERR_CONTAINER:	return "port wider than the 16-bit sample container";
	case CORDIC_ERR_NOMEM:		return "out of host memory";
	default:			return "unknown status";
	}
}

} // namespace cordic_amd
