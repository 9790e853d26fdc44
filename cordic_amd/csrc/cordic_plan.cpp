// cordic_plan.cpp -- "seed tables" for constant-vector rotators (host side).
//
// When i_xval / i_yval are held constant (the sin/cos generator use of the
// core, reference bench/cpp/cordic_tb.cpp:68-69), the state (x, y) after the
// first M stages depends only on the octant q and on the M rotation
// directions, and the directions depend only on the folded phase p0:
//     s_i = +1 if p_i >= 0 else -1,   p_{i+1} = p_i - s_i * a_i
// (rtl/cordic.v:262-280).  That map is monotone in p0, so [-45deg, +45deg)
// splits into at most 2^M consecutive intervals ("leaves"), one per reachable
// direction pattern, whose end points are exact integers (partial sums of
// +/- a_i).  A kernel can therefore replace the first M micro-rotations by:
// find the leaf of p0 (bucket table + one compare), fetch (x_M, y_M)
// for (q, leaf) from a table that the kernel itself fills by running the exact
// recurrence once per entry, and continue with stage M.  Results are
// bit-identical for every phase; only the per-sample work changes.
//
// This file builds the phase-side tables (they depend on the arctan table
// only) as a flat array of 32-bit words:
//   [0] M  [1] S (bucket shift)  [2] nbuckets  [3] nleaves
//   buckets: nbuckets x {bound-1, first_leaf}   in the r-domain
//            r = p0 + 2^29 in [0, 2^30); a bucket holds at most ONE leaf
//            boundary (the narrowest leaf of a real arctan table is ~0.85 *
//            2^19 wide at M = 11, so S = 18 does); no boundary: 0x7fffffff
//   leaves : nleaves  x {pattern (M bits, stage 0 = MSB), off + 2^29}
//            where p_M = p0 - off
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "cordic_amd.h"
#include "cordic_internal.h"

namespace cordic_amd {

namespace {
struct Leaf { int64_t lo, hi, off; uint32_t pattern; };

void split(const uint32_t *ang, int m, int i, int64_t lo, int64_t hi, int64_t off,
		uint32_t pat, std::vector<Leaf> &out)
{
	if (lo >= hi)
		return;
	if (i == m) {
		out.push_back({lo, hi, off, pat});
		return;
	}
	const int64_t a = ang[i];
	// p_i = p0 - off.  p_i < 0 (s = -1): p' = p + a;  p_i >= 0 (s = +1): p' = p - a
	split(ang, m, i + 1, lo, std::min(hi, off), off - a, pat << 1, out);
	split(ang, m, i + 1, std::max(lo, off), hi, off + a, (pat << 1) | 1u, out);
}
} // namespace

bool seed_eligible(const cordic_config &c, int m)
{
	if (c.mode != CORDIC_P2R && c.mode != CORDIC_SP2R)
		return false;
	if (c.nlive < m || m < 1 || m > 12)
		return false;
	if (c.ww > 35)
		return false;			// no seeded kernel for these
	if (c.needs_wrap && c.ww != 32 && c.ww != 35)
		return false;
	return true;
}

// Returns the number of words written (0 if the table cannot be built within
// `cap` words or the bucket constraint cannot be met).
size_t build_seed_table(const cordic_config &c, int m, uint32_t *buf, size_t cap,
		DtInfo *dt)
{
	if (dt)
		*dt = DtInfo{};
	if (!seed_eligible(c, m))
		return 0;
	const int lsh = 32 - c.pw;
	uint32_t ang[CORDIC_AMD_MAX_STAGES];
	for (int i = 0; i < m; i++)
		ang[i] = c.angle[i] << lsh;	// left-justified, as the kernels use it

	const int64_t LO = -((int64_t)1 << 29), HI = (int64_t)1 << 29;
	std::vector<Leaf> leaves;
	split(ang, m, 0, LO, HI, 0, 0u, leaves);
	std::sort(leaves.begin(), leaves.end(),
		[](const Leaf &a, const Leaf &b) { return a.lo < b.lo; });
	const size_t L = leaves.size();
	if (L == 0 || L > 4096)
		return 0;
	// The left-justified kernels keep only (off + 2^29) mod 2^29 per leaf and
	// recover the residual p_M = p0 - off as a 29-bit signed number
	// (cordic_device.h: rotator_seeded), so every residual has to fit.  It
	// does by a wide margin for any real arctan table (|p_M| <= a_{M-1}); it
	// does not for degenerate ones (e.g. PW = 3, where every angle after the
	// first truncates to zero): those cores run the full recurrence.
	for (const Leaf &lf : leaves) {
		const int64_t lim = ((int64_t)1 << 28) - 1;
		if (lf.lo - lf.off < -lim || lf.hi - 1 - lf.off > lim)
			return 0;
	}

	// largest bucket (2^S phase units) holding at most one leaf boundary
	// (boundaries on a bucket edge do not count: they are the bucket's
	// first_leaf)
	int S = 26;
	std::vector<int> count;
	for (; S >= 17; S--) {
		count.assign((size_t)1 << (30 - S), 0);
		int worst = 0;
		for (size_t j = 1; j < L; j++) {
			const int64_t r = leaves[j].lo - LO;
			if (r & (((int64_t)1 << S) - 1))
				worst = std::max(worst, ++count[(size_t)(r >> S)]);
		}
		if (worst <= 1)
			break;
	}
	if (S < 17)
		return 0;
	const size_t nb = (size_t)1 << (30 - S);
	// what the kernel keeps in LDS: 8 B per bucket, 4 quadrants x 16 B per
	// leaf, the tile-id slots (cordic_device.h: rotator_seeded)
	if (nb * 8 + L * 64 + 64 > CORDIC_SEED_LDS_BYTES)
		return 0;
	const size_t words = 4 + nb * 2 + L * 2;
	if (!buf || words > cap)
		return 0;

	buf[0] = (uint32_t)m;
	buf[1] = (uint32_t)S;
	buf[2] = (uint32_t)nb;
	buf[3] = (uint32_t)L;
	uint32_t *bk = buf + 4;
	for (size_t b = 0; b < nb; b++) {
		bk[2 * b + 0] = 0x7fffffffu;
		bk[2 * b + 1] = 0;
	}
	// first_leaf[b] = leaf containing the first phase of bucket b
	size_t j = 0;
	for (size_t b = 0; b < nb; b++) {
		const int64_t start = LO + ((int64_t)b << S);
		while (j + 1 < L && leaves[j + 1].lo <= start)
			j++;
		bk[2 * b + 1] = (uint32_t)j;
	}
	for (size_t k = 1; k < L; k++) {
		const int64_t r = leaves[k].lo - LO;		// boundary, r-domain
		const size_t b = (size_t)(r >> S);
		if ((r & (((int64_t)1 << S) - 1)) == 0)
			continue;	// on a bucket edge: already in first_leaf
		bk[2 * b] = (uint32_t)(r - 1);
	}
	uint32_t *lf = bk + 2 * nb;
	for (size_t k = 0; k < L; k++) {
		lf[2 * k + 0] = leaves[k].pattern;
		lf[2 * k + 1] = (uint32_t)(leaves[k].off - LO);	// off + 2^29
	}

	// ---- direction tails (cordic_internal.h: dt_levels): groups of stages
	// behind the seeded ones, each with its own bucket / leaf table over the
	// residual phase.  Appended as
	//   [words] n, bias0, bias_last, 0
	//   per group: t, shift, nb, nl, 0, 0, then nb x {bound-1, first_leaf}
	//   (u-domain: u = residual + bias of the incoming residual; 0x7fffffff:
	//   no boundary), then nl x {pattern (t bits, first stage = MSB),
	//   off'} with  u_next = u - off'  (the next group's -- or, behind the
	//   last one, the phase chain's -- biased residual).
	const int R = c.nlive - m;
	const int ngroups = dt_levels(R);
	if (ngroups == 0 || ngroups > kDtMaxLevels)
		return words;
	std::vector<uint32_t> tail;
	DtInfo info;
	// residual range behind the seed stages
	int64_t rmin = 0, rmax = 0;
	bool first = true;
	for (const Leaf &l : leaves) {
		const int64_t a = l.lo - l.off, b = l.hi - 1 - l.off;
		if (first || a < rmin) rmin = a;
		if (first || b > rmax) rmax = b;
		first = false;
	}
	int64_t bias = -rmin;
	info.bias0 = (uint32_t)bias;
	tail.assign(4, 0u);
	size_t lds_at = nb * 8 + L * 64 + 16;	// behind the seeds and the tile slots
	for (int g = 0; g < ngroups; g++) {
		const int t = dt_size(R, g), s0 = m + dt_first(R, g);
		uint32_t a2[8];
		for (int i = 0; i < t; i++)
			a2[i] = c.angle[s0 + i] << lsh;
		std::vector<Leaf> lv;
		split(a2, t, 0, rmin, rmax + 1, 0, 0u, lv);
		std::sort(lv.begin(), lv.end(),
			[](const Leaf &a, const Leaf &b) { return a.lo < b.lo; });
		const size_t nl = lv.size();
		if (nl == 0 || nl > 256)
			return words;
		const int64_t umax = rmax + bias;		// u in [0, umax]
		if (umax >= ((int64_t)1 << 28))
			return words;
		// largest bucket with at most one boundary (not on its edge)
		int S2 = 24;
		std::vector<int> cnt;
		for (; S2 >= 4; S2--) {
			cnt.assign((size_t)(umax >> S2) + 1, 0);
			int worst = 0;
			for (size_t j = 1; j < nl; j++) {
				const int64_t u = lv[j].lo + bias;
				if (u & (((int64_t)1 << S2) - 1))
					worst = std::max(worst, ++cnt[(size_t)(u >> S2)]);
			}
			if (worst <= 1)
				break;
		}
		if (S2 < 4)
			return words;
		size_t nb2 = 1;
		while (nb2 < (size_t)(umax >> S2) + 1)
			nb2 <<= 1;
		if (nb2 > 4096)
			return words;
		DtLevel &d = info.lv[g];
		d.t = t; d.shift = S2; d.nb = (int32_t)nb2; d.nl = (int32_t)nl;
		d.word = (int32_t)(words + tail.size() + 6);
		const uint32_t hdr[6] = {(uint32_t)t, (uint32_t)S2, (uint32_t)nb2,
				(uint32_t)nl, 0u, 0u};
		tail.insert(tail.end(), hdr, hdr + 6);
		const size_t b0 = tail.size();
		tail.resize(b0 + nb2 * 2);
		size_t j = 0;
		for (size_t b = 0; b < nb2; b++) {
			const int64_t start = ((int64_t)b << S2) - bias;	// residual
			while (j + 1 < nl && lv[j + 1].lo <= start)
				j++;
			tail[b0 + 2 * b + 0] = 0x7fffffffu;
			tail[b0 + 2 * b + 1] = (uint32_t)j;
		}
		for (size_t k = 1; k < nl; k++) {
			const int64_t u = lv[k].lo + bias;
			if ((u & (((int64_t)1 << S2) - 1)) == 0)
				continue;
			tail[b0 + 2 * (size_t)(u >> S2)] = (uint32_t)(u - 1);
		}
		// residual range behind this group, and its bias
		int64_t nmin = 0, nmax = 0;
		first = true;
		for (const Leaf &l : lv) {
			const int64_t a = l.lo - l.off, b = l.hi - 1 - l.off;
			if (first || a < nmin) nmin = a;
			if (first || b > nmax) nmax = b;
			first = false;
		}
		const int64_t nbias = -nmin;
		for (const Leaf &l : lv) {
			tail.push_back(l.pattern);
			// u_next = (r - off) + nbias = u - (off + bias - nbias)
			tail.push_back((uint32_t)(l.off + bias - nbias));
		}
		// LDS the kernel needs for it (cordic_device.h: dt_lds_layout):
		// buckets, aligned to their own size, then the leaf entries
		lds_at = (lds_at + nb2 * 8 - 1) & ~(nb2 * 8 - 1);
		lds_at = (lds_at + nb2 * 8 + 15) & ~(size_t)15;
		lds_at += nl * (size_t)dt_entry_dwords(t) * 4;
		rmin = nmin; rmax = nmax; bias = nbias;
	}
	info.n = ngroups;
	info.bias_last = (uint32_t)bias;
	tail[0] = (uint32_t)ngroups;
	tail[1] = info.bias0;
	tail[2] = info.bias_last;
	if (lds_at > CORDIC_SEED_LDS_BYTES)
		return words;			// no room: seeds only
	if (words + tail.size() > cap)
		return words;
	std::memcpy(buf + words, tail.data(), tail.size() * 4);
	if (dt)
		*dt = info;
	return words + tail.size();
}

// Direction tables for per-sample vectors (cordic_internal.h: dx_*).  The
// octant fold leaves p0 in [-2^29, 2^29) and performs stage 1 itself
// (p1 = p0 -/+ a_0); group g then covers dx_size stages behind dx_first, its
// table indexed by the biased residual u >= 0.
size_t build_dir_table(const cordic_config &c, uint32_t *buf, size_t cap,
		DxInfo *out)
{
	if (out)
		*out = DxInfo{};
	if (c.mode != CORDIC_P2R && c.mode != CORDIC_SP2R)
		return 0;
	if (c.needs_wrap || c.ww > 35 || c.nlive < kDxFirst + 2 || c.nlive > 40)
		return 0;
	const int ngroups = dx_levels(c.nlive);
	if (ngroups < 1 || ngroups > kDxMaxLevels)
		return 0;
	const int lsh = 32 - c.pw;
	// residual range behind stage 1
	const int64_t LO = -((int64_t)1 << 29), HI = (int64_t)1 << 29;
	const int64_t a0 = (int64_t)(c.angle[0] << lsh);
	if (a0 <= 0 || a0 >= HI)
		return 0;
	int64_t rmin = std::min(LO + a0, -a0), rmax = std::max(a0 - 1, HI - 1 - a0);
	int64_t bias = -rmin;
	std::vector<uint32_t> tail(4, 0u);
	DxInfo info;
	info.bias0 = (uint32_t)bias;
	// LDS of rotator_xydir (cordic_xydir.h: dx_lds_layout): the fold's eight
	// 16-byte rows first, then per group its buckets (aligned to their total
	// size) and leaf entries -- the same arithmetic, so that the bound checked
	// below is the figure the launch asks for
	size_t lds_at = 8 * 16;
	for (int g = 0; g < ngroups; g++) {
		const int t = dx_size(c.nlive, g), s0 = dx_first(c.nlive, g);
		uint32_t a2[8];
		for (int i = 0; i < t; i++)
			a2[i] = c.angle[s0 + i] << lsh;
		std::vector<Leaf> lv;
		split(a2, t, 0, rmin, rmax + 1, 0, 0u, lv);
		std::sort(lv.begin(), lv.end(),
			[](const Leaf &a, const Leaf &b) { return a.lo < b.lo; });
		const size_t nl = lv.size();
		if (nl == 0 || nl > 256)
			return 0;
		const int64_t umax = rmax + bias;		// u in [0, umax]
		if (umax >= ((int64_t)1 << 30))
			return 0;
		int S2 = 26;
		std::vector<int> cnt;
		for (; S2 >= 4; S2--) {
			cnt.assign((size_t)(umax >> S2) + 1, 0);
			int worst = 0;
			for (size_t j = 1; j < nl; j++) {
				const int64_t u = lv[j].lo + bias;
				if (u & (((int64_t)1 << S2) - 1))
					worst = std::max(worst, ++cnt[(size_t)(u >> S2)]);
			}
			if (worst <= 1)
				break;
		}
		if (S2 < 4)
			return 0;
		size_t nb2 = 1;
		while (nb2 < (size_t)(umax >> S2) + 1)
			nb2 <<= 1;
		if (nb2 > 4096)
			return 0;
		DtLevel &d = info.lv[g];
		d.t = t; d.shift = S2; d.nb = (int32_t)nb2; d.nl = (int32_t)nl;
		d.word = (int32_t)(tail.size() + 6);
		const uint32_t hdr[6] = {(uint32_t)t, (uint32_t)S2, (uint32_t)nb2,
				(uint32_t)nl, 0u, 0u};
		tail.insert(tail.end(), hdr, hdr + 6);
		const size_t b0 = tail.size();
		tail.resize(b0 + nb2 * 2);
		size_t j = 0;
		for (size_t b = 0; b < nb2; b++) {
			const int64_t start = ((int64_t)b << S2) - bias;	// residual
			while (j + 1 < nl && lv[j + 1].lo <= start)
				j++;
			tail[b0 + 2 * b + 0] = 0x7fffffffu;
			tail[b0 + 2 * b + 1] = (uint32_t)j;
		}
		for (size_t k = 1; k < nl; k++) {
			const int64_t u = lv[k].lo + bias;
			if ((u & (((int64_t)1 << S2) - 1)) == 0)
				continue;
			tail[b0 + 2 * (size_t)(u >> S2)] = (uint32_t)(u - 1);
		}
		int64_t nmin = 0, nmax = 0;
		bool first = true;
		for (const Leaf &l : lv) {
			const int64_t a = l.lo - l.off, b = l.hi - 1 - l.off;
			if (first || a < nmin) nmin = a;
			if (first || b > nmax) nmax = b;
			first = false;
		}
		const int64_t nbias = -nmin;
		for (const Leaf &l : lv) {
			tail.push_back(l.pattern);
			tail.push_back((uint32_t)(l.off + bias - nbias));	// u_next = u - this
		}
		lds_at = (lds_at + nb2 * 8 - 1) & ~(nb2 * 8 - 1);
		lds_at = (lds_at + nb2 * 8 + 15) & ~(size_t)15;
		lds_at += nl * (size_t)dt_entry_dwords(t) * 4;
		rmin = nmin; rmax = nmax; bias = nbias;
	}
	// the recurrence behind the last group carries the residual as a 29-bit
	// signed number scaled by 2^31 (cordic_device.h: RotChainLJ)
	if (dx_rest(c.nlive) > 0 && (rmin < -((int64_t)1 << 28) || rmax >= ((int64_t)1 << 28)))
		return 0;
	if (lds_at > 60 * 1024)		// (the 64 KiB a launch gets without opting in)
		return 0;
	info.n = ngroups;
	info.bias_last = (uint32_t)bias;
	tail[0] = (uint32_t)ngroups;
	tail[1] = info.bias0;
	tail[2] = info.bias_last;
	if (!buf || tail.size() > cap)
		return 0;
	std::memcpy(buf, tail.data(), tail.size() * 4);
	if (out)
		*out = info;
	return tail.size();
}

} // namespace cordic_amd
