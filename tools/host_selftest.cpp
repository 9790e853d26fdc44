// host_selftest.cpp -- the host layer (no HIP) under AddressSanitizer /
// UBSan: random and hostile parameters through every host-only entry of
// cordic_internal.h, with undersized output buffers.
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all \
//       -I include -I cordic_amd/csrc -ffp-contract=off tools/host_selftest.cpp \
//       cordic_amd/csrc/cordic_config.cpp cordic_amd/csrc/cordic_plan.cpp \
//       cordic_amd/csrc/cordic_quadtbl.cpp -o /tmp/host_selftest && /tmp/host_selftest
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "cordic_internal.h"

using namespace cordic_amd;

int main(int argc, char **argv)
{
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 1u);
	auto pick = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
	const char *vocab[] = { "-t", "p2r", "r2p", "sp2r", "sr2p", "tbl", "qtr", "qtbl",
		"bogus", "-i", "-o", "-p", "-n", "-x", "-f", "some/long/path/name.v", "-",
		"-a", "-c", "-r", "-R", "-A", "-v", "-vca", "-h", "13", "0", "-1", "32",
		"33", "64", "65", "999999999", "abc", "", "-z", "-tp2r", "-i13", "-f" };
	const int nvocab = (int)(sizeof vocab / sizeof vocab[0]);
	long built = 0, seeded = 0, quads = 0, parsed = 0;
	for (int it = 0; it < 40000; it++) {
		cordic_config c;
		const int mode = pick(-1, 5), iw = pick(-3, 40), ow = pick(-3, 40);
		const int xtra = pick(-2, 40), pw = pick(-2, 40), ns = pick(-2, 70);
		if (build_from_cli(&c, mode, iw, ow, xtra, pw, ns) == CORDIC_OK) {
			built++;
			for (size_t cap : { (size_t)0, (size_t)1, (size_t)17, (size_t)300, (size_t)4096 }) {
				std::vector<char> buf(cap ? cap : 1);
				(void)write_header(&c, "core", cap ? buf.data() : nullptr, cap);
			}
			for (size_t cap : { (size_t)0, (size_t)3, (size_t)100, (size_t)5000, (size_t)30000 }) {
				std::vector<uint32_t> w(cap ? cap : 1);
				if (build_seed_table(c, CORDIC_SEED_STAGES, cap ? w.data() : nullptr, cap))
					seeded++;
			}
		}
		(void)build_core(&c, mode, ns, iw, ow, xtra, pw);
		cordic_quad_config q;
		if (quad_build_from_cli(&q, iw, ow, xtra, pw) == CORDIC_OK) {
			quads++;
			for (size_t cap : { (size_t)0, (size_t)5, (size_t)q.entries }) {
				std::vector<int32_t> a(cap ? cap : 1), b(cap ? cap : 1), d(cap ? cap : 1);
				(void)quad_fill(q, a.data(), b.data(), d.data(), cap);
			}
			char small[40];
			(void)quad_write_header(&q, "quadtbl", small, sizeof small);
			(void)quad_write_header(&q, "quadtbl", nullptr, 0);
		}
		(void)quad_build_core(&q, pw, ow, xtra);
		for (int kind : { 4, 5, 6 }) {
			cordic_table_config t;
			if (table_derive(&t, kind, iw, ow, pw) == CORDIC_OK && t.entries <= (1 << 16)) {
				std::vector<int32_t> v((size_t)t.entries);
				(void)table_fill(t, v.data(), v.size());
				(void)table_fill(t, v.data(), v.size() / 2);
			}
		}
		std::vector<std::string> own;
		std::vector<const char *> av;
		own.push_back("gencordic");
		const int n = pick(0, 9);
		for (int k = 0; k < n; k++)
			own.push_back(vocab[pick(0, nvocab - 1)]);
		for (auto &s : own) av.push_back(s.c_str());
		char fname[24];		// deliberately short
		int hdr = 0;
		if (parse_args(&c, (int)av.size(), av.data(), fname, sizeof fname, &hdr) == CORDIC_OK)
			parsed++;
		(void)parse_args(&c, (int)av.size(), av.data(), nullptr, 0, nullptr);
		(void)stages_for(pw, iw);
		(void)phase_bits_for(ow);
		(void)phase_variance(ns, pw);
		(void)quantization_variance(ns, xtra, iw);
		(void)next_lg((unsigned)ns);
		(void)status_text(-it % 20);
	}
	printf("host selftest ok: %ld cores, %ld seed tables, %ld quadtbl cores, %ld "
		"command lines parsed\n", built, seeded, quads, parsed);
	return 0;
}
