// gencordic_amd.cpp -- gencordic's command line in front of the MI355X engine.
//
// Accepts the reference generator's flags (sw/main.cpp:57-92,139:
// "aAcf:hi:n:o:p:Rrt:vx:") so that existing make rules such as
//     ./gencordic -vca -f ../rtl/cordic.v -v -i 13 -o 13 -t p2r -x 2 -c
// (sw/Makefile:115-144) can be pointed at this binary.  Where the reference
// writes Verilog, this tool "generates" the GPU core: it derives the same
// parameters and tables, writes the same constants header when -c is given
// (<fname>.v -> <fname>.h, the text between #ifndef and #endif is byte
// identical to the reference's), prints the parameter report with -v, and can
// stream a phase ramp through the core (--run LOG2N) as a smoke test.  It does
// not emit Verilog: RTL text is outside the hot path.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cordic_amd.h"

static void usage()
{
	fprintf(stderr,
"USAGE: gencordic_amd [-aAchrRv] [-f <fname>] [-i <iw>] [-o <ow>]\n"
"\t[-n <stages>] [-p <phasebits>] [-t p2r|r2p|sp2r|sr2p|tbl|qtr|qtbl]\n"
"\t[-x <xtrabits>]\n"
"\t[--run <log2 samples>]\n"
"\n"
"\tSame meaning as the flags of ZipCPU/cordic's gencordic.  -c writes the\n"
"\tconstants header next to <fname>; --run streams a phase ramp through the\n"
"\tgenerated core on the current GPU and reports the rate.\n");
}

static const char *mode_name(int m)
{
	switch (m) {
	case CORDIC_P2R: return "p2r";
	case CORDIC_R2P: return "r2p";
	case CORDIC_SP2R: return "sp2r";
	default: return "sr2p";
	}
}

int main(int argc, char **argv)
{
	if (argc <= 1) {
		usage();
		return EXIT_SUCCESS;
	}
	// split off our own long option, leave the rest to the gencordic parser
	std::vector<const char *> args;
	int run_lg = -1;
	bool verbose = false;
	for (int k = 0; k < argc; k++) {
		if (k > 0 && !strcmp(argv[k], "--run") && k + 1 < argc) {
			run_lg = atoi(argv[++k]);
			continue;
		}
		if (k > 0 && argv[k][0] == '-' && argv[k][1] != '-') {
			if (strchr(argv[k], 'h') && !strpbrk(argv[k], "finoptx")) {
				usage();
				return EXIT_SUCCESS;
			}
			// a bundle of plain flags may carry -v
			bool value_flag = false;
			for (const char *p = argv[k] + 1; *p && !value_flag; p++) {
				if (strchr("finoptx", *p)) value_flag = true;
				else if (*p == 'v') verbose = true;
			}
		}
		args.push_back(argv[k]);
	}

	// -t tbl / -t qtr: the table cores.  Derive PW / OW as gencordic does and
	// write the same <fname>.hex the reference writes (sw/hexfile.cpp:76-88:
	// "@%08x " every 8 entries, (OW+3)/4 hex digits per entry).
	{
		int kind = 0, iw = -1, ow = -1, pw = -1;
		std::string f;
		for (size_t k = 1; k + 1 < args.size() + 1; k++) {
			const char *a = args[k];
			const char *v = (k + 1 < args.size()) ? args[k + 1] : nullptr;
			if (!strcmp(a, "-t") && v) {
				if (!strcmp(v, "tbl")) kind = CORDIC_TBL;
				else if (!strcmp(v, "qtr")) kind = CORDIC_QTR;
			} else if (!strcmp(a, "-i") && v) iw = atoi(v);
			else if (!strcmp(a, "-o") && v) ow = atoi(v);
			else if (!strcmp(a, "-p") && v) pw = atoi(v);
			else if (!strcmp(a, "-f") && v) f = v;
		}
		if (kind) {
			cordic_table_config tc;
			const int trc = cordic_table_config_init(&tc, kind, iw, ow, pw);
			if (trc != CORDIC_OK) {
				fprintf(stderr, "ERR: %s\n", cordic_strerror(trc));
				return EXIT_FAILURE;
			}
			if (f.empty())
				f = (kind == CORDIC_TBL) ? "sintable.v" : "quarterwav.v";
			if (verbose)
				printf("Generated a %s table core: PW %d, OW %d, %d entries\n",
					kind == CORDIC_TBL ? "sine" : "quarter-wave", tc.pw,
					tc.ow, tc.entries);
			std::vector<int32_t> tv((size_t)tc.entries);
			cordic_table_values(&tc, tv.data(), tv.size());
			std::string hexname = f;
			if (hexname.size() > 4 && hexname[hexname.size() - 2] == '.')
				hexname = hexname.substr(0, hexname.size() - 2);
			hexname += ".hex";
			FILE *hf = fopen(hexname.c_str(), "w");
			if (!hf) {
				fprintf(stderr, "ERR: Cannot open %s for writing\n", hexname.c_str());
				return EXIT_FAILURE;
			}
			const int nc = (tc.ow + 3) / 4;
			const long msk = (1l << tc.ow) - 1l;
			for (int k = 0; k < tc.entries; k++) {
				if (0 == (k % 8))
					fprintf(hf, "%s@%08x ", (k != 0) ? "\n" : "", k);
				fprintf(hf, "%0*lx ", nc, (long)tv[k] & msk);
			}
			fprintf(hf, "\n");
			fclose(hf);
			if (run_lg >= 0) {
				if (run_lg > 31) run_lg = 31;
				const size_t n = (size_t)1 << run_lg;
				uint32_t *ph; int32_t *o;
				cordic_table *tb = nullptr;
				if (hipMalloc((void **)&ph, n * 4) != hipSuccess ||
				    hipMalloc((void **)&o, n * 4) != hipSuccess ||
				    cordic_table_create(&tc, &tb) != CORDIC_OK) {
					fprintf(stderr, "ERR: %s\n", cordic_strerror(CORDIC_ERR_DEVICE));
					return EXIT_FAILURE;
				}
				cordic_fill_phase_ramp(ph, n, 0, 0, nullptr);
				cordic_table_lookup(tb, n, ph, o, nullptr);
				(void)hipDeviceSynchronize();
				const auto t0 = std::chrono::steady_clock::now();
				for (int k = 0; k < 10; k++)
					cordic_table_lookup(tb, n, ph, o, nullptr);
				(void)hipDeviceSynchronize();
				const double dt = std::chrono::duration<double>(
						std::chrono::steady_clock::now() - t0).count();
				printf("streamed 10 x 2^%d samples: %.1f Msamples/s\n",
					run_lg, (double)n * 10 / dt / 1e6);
				cordic_table_destroy(tb);
				(void)hipFree(ph); (void)hipFree(o);
			}
			return EXIT_SUCCESS;
		}
	}

	// -t qtbl: the quadratically interpolated sine core.  Writes the three
	// coefficient tables the reference writes (<fname>_ctbl.hex, _ltbl.hex,
	// _qtbl.hex; sw/quadtbl.cpp:243-259) and, with -c, its constants header.
	{
		bool quad = false, hdr = false, aux = false, noreset = false;
		int iw = -1, ow = -1, pw = -1, xtra = 2;
		std::string f;
		for (size_t k = 1; k < args.size(); k++) {
			const char *a = args[k];
			const char *v = (k + 1 < args.size()) ? args[k + 1] : nullptr;
			if (a[0] != '-')
				continue;
			if (!strcmp(a, "-t") && v) quad = !strcmp(v, "qtbl");
			else if (!strcmp(a, "-i") && v) iw = atoi(v);
			else if (!strcmp(a, "-o") && v) ow = atoi(v);
			else if (!strcmp(a, "-p") && v) pw = atoi(v);
			else if (!strcmp(a, "-x") && v) xtra = atoi(v);
			else if (!strcmp(a, "-f") && v) f = v;
			else if (!strpbrk(a, "finoptx")) {	// bundle of plain flags
				if (strchr(a, 'c')) hdr = true;
				if (strchr(a, 'a')) aux = true;
				if (strchr(a, 'R')) noreset = true;
			}
		}
		if (quad) {
			cordic_quad_config qc;
			const int qrc = cordic_quad_config_init(&qc, iw, ow, xtra, pw);
			if (qrc != CORDIC_OK) {
				fprintf(stderr, "ERR: %s\n", cordic_strerror(qrc));
				return EXIT_FAILURE;
			}
			qc.has_aux = aux;
			qc.has_reset = !noreset;
			if (f.empty())
				f = "quadtbl.v";
			std::string stem = f;
			const size_t dot = stem.rfind('.');
			if (dot != std::string::npos)
				stem = stem.substr(0, dot);
			if (verbose)
				printf("Generated a quadratic-interpolation sine core: PW %d, "
					"OW %d, XTRA %d, %d-entry tables, CBITS:LBITS:QBITS = "
					"%d:%d:%d, table error %.6f\n", qc.pw, qc.ow, qc.xtra,
					qc.entries, qc.cbits, qc.lbits, qc.qbits, qc.itbl_err);
			std::vector<int32_t> tc(qc.entries), tl(qc.entries), tq(qc.entries);
			cordic_quad_tables(&qc, tc.data(), tl.data(), tq.data(), tc.size());
			const struct { const char *sfx; const std::vector<int32_t> *v; int bits; }
				files[3] = { { "_ctbl", &tc, qc.cbits }, { "_ltbl", &tl, qc.lbits },
					     { "_qtbl", &tq, qc.qbits } };
			for (const auto &t : files) {
				const std::string hn = stem + t.sfx + ".hex";
				FILE *hf = fopen(hn.c_str(), "w");
				if (!hf) {
					fprintf(stderr, "ERR: Cannot open %s for writing\n", hn.c_str());
					return EXIT_FAILURE;
				}
				const int nc = (t.bits + 3) / 4;
				const long msk = (1l << t.bits) - 1l;
				for (int k = 0; k < qc.entries; k++) {
					if (0 == (k % 8))
						fprintf(hf, "%s@%08x ", (k != 0) ? "\n" : "", k);
					fprintf(hf, "%0*lx ", nc, (long)(*t.v)[k] & msk);
				}
				fprintf(hf, "\n");
				fclose(hf);
			}
			if (hdr) {
				std::string base = stem;
				const size_t sl = base.rfind('/');
				if (sl != std::string::npos)
					base = base.substr(sl + 1);
				char text[2048];
				cordic_quad_write_header(&qc, base.c_str(), text, sizeof text);
				const std::string hn = stem + ".h";
				FILE *hf = fopen(hn.c_str(), "w");
				if (!hf) {
					fprintf(stderr, "WARNING: Could not open %s\n", hn.c_str());
				} else {
					fputs(text, hf);
					fclose(hf);
				}
			}
			if (run_lg >= 0) {
				if (run_lg > 31) run_lg = 31;
				const size_t n = (size_t)1 << run_lg;
				uint32_t *ph; int32_t *o;
				cordic_quad *core = nullptr;
				if (hipMalloc((void **)&ph, n * 4) != hipSuccess ||
				    hipMalloc((void **)&o, n * 4) != hipSuccess ||
				    cordic_quad_create(&qc, &core) != CORDIC_OK) {
					fprintf(stderr, "ERR: %s\n", cordic_strerror(CORDIC_ERR_DEVICE));
					return EXIT_FAILURE;
				}
				cordic_fill_phase_ramp(ph, n, 0, 0, nullptr);
				cordic_quad_lookup(core, n, ph, o, nullptr);
				(void)hipDeviceSynchronize();
				const auto t0 = std::chrono::steady_clock::now();
				for (int k = 0; k < 10; k++)
					cordic_quad_lookup(core, n, ph, o, nullptr);
				(void)hipDeviceSynchronize();
				const double dt = std::chrono::duration<double>(
						std::chrono::steady_clock::now() - t0).count();
				printf("streamed 10 x 2^%d samples: %.1f Msamples/s\n",
					run_lg, (double)n * 10 / dt / 1e6);
				cordic_quad_destroy(core);
				(void)hipFree(ph); (void)hipFree(o);
			}
			return EXIT_SUCCESS;
		}
	}

	cordic_config cfg;
	char fname[512];
	int want_header = 0;
	const int rc = cordic_config_from_args(&cfg, (int)args.size(), args.data(),
			fname, sizeof fname, &want_header);
	if (rc != CORDIC_OK) {
		fprintf(stderr, "ERR: %s\n", cordic_strerror(rc));
		return EXIT_FAILURE;
	}

	if (verbose) {
		printf("Generated a %s core for the MI355X engine:\n"
			"\tOutput file     : %s\n"
			"\tInput  bits     : %2d\n"
			"\tExtra  bits     : %2d (used in computation, dropped when done)\n"
			"\tOutput bits     : %2d\n"
			"\tWorking bits    : %2d\n"
			"\tPhase  bits     : %2d\n"
			"\tNumber of stages: %2d (%d micro-rotations performed)\n",
			mode_name(cfg.mode), fname, cfg.iw, cfg.nextra, cfg.ow, cfg.ww,
			cfg.pw, cfg.nstages, cfg.nlive);
		if (cfg.clocks_per_output)
			printf("\tSequential core : %d clocks per output in RTL\n",
				cfg.clocks_per_output);
	}

	if (want_header) {
		std::string f(fname);
		const size_t slen = f.size();
		if (slen > 2 && f[slen - 1] == 'v' && f[slen - 2] == '.') {
			std::string hname = f.substr(0, slen - 1) + "h";
			// module name: strip the directory and the ".v" (sw/legal.cpp:96-113)
			const size_t slash = f.find_last_of('/');
			std::string mod = f.substr(slash == std::string::npos ? 0 : slash + 1);
			mod = mod.substr(0, mod.size() - 2);
			char text[4096];
			const int n = cordic_config_write_header(&cfg, mod.c_str(), text, sizeof text);
			FILE *fh = (n > 0) ? fopen(hname.c_str(), "w") : nullptr;
			if (!fh) {
				fprintf(stderr, "WARNING: Could not open %s\n", hname.c_str());
			} else {
				fprintf(fh, "// %s -- constants of the %s core generated by\n"
					"//   gencordic_amd", hname.c_str(), mode_name(cfg.mode));
				for (size_t k = 1; k < args.size(); k++)
					fprintf(fh, " %s", args[k]);
				fprintf(fh, "\n//\n");
				fputs(text, fh);
				fclose(fh);
			}
		}
	}

	if (run_lg >= 0) {
		if (run_lg > 31) run_lg = 31;
		const size_t n = (size_t)1 << run_lg;
		int32_t *a, *b; uint32_t *ph;
		if (hipMalloc((void **)&a, n * 4) != hipSuccess ||
		    hipMalloc((void **)&b, n * 4) != hipSuccess ||
		    hipMalloc((void **)&ph, n * 4) != hipSuccess) {
			fprintf(stderr, "ERR: %s\n", cordic_strerror(CORDIC_ERR_DEVICE));
			return EXIT_FAILURE;
		}
		const int shift = cfg.pw > run_lg ? cfg.pw - run_lg : 0;
		int r2 = cordic_fill_phase_ramp(ph, n, 0, shift, nullptr);
		cordic_plan *plan = nullptr;
		if (r2 == CORDIC_OK) r2 = cordic_plan_create(&cfg, &plan);
		const int32_t amp = (int32_t)((1ul << (cfg.iw - 1)) - 1);
		auto once = [&]() {
			if (cfg.mode == CORDIC_P2R || cfg.mode == CORDIC_SP2R)
				return cordic_plan_p2r_const(plan, n, amp, 0, ph, a, b, nullptr);
			// feed the ramp as I, its bit reverse-ish as Q: any data will do
			return cordic_r2p(&cfg, n, (const int32_t *)ph, (const int32_t *)ph,
					a, (uint32_t *)b, nullptr);
		};
		if (r2 == CORDIC_OK) r2 = once();	// warm up
		(void)hipDeviceSynchronize();
		const auto t0 = std::chrono::steady_clock::now();
		const int reps = 10;
		for (int k = 0; k < reps && r2 == CORDIC_OK; k++)
			r2 = once();
		if (hipDeviceSynchronize() != hipSuccess)
			r2 = CORDIC_ERR_DEVICE;
		const double dt = std::chrono::duration<double>(
				std::chrono::steady_clock::now() - t0).count();
		if (r2 != CORDIC_OK) {
			fprintf(stderr, "ERR: %s\n", cordic_strerror(r2));
			return EXIT_FAILURE;
		}
		printf("streamed %d x 2^%d samples: %.1f Msamples/s\n", reps, run_lg,
			(double)n * reps / dt / 1e6);
		cordic_plan_destroy(plan);
		(void)hipFree(a); (void)hipFree(b); (void)hipFree(ph);
	}
	return EXIT_SUCCESS;
}
