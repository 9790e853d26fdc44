// cordic_launch.h -- launchers exported by the instantiation units.
#ifndef CORDIC_LAUNCH_H
#define CORDIC_LAUNCH_H

#include <hip/hip_runtime.h>

#include "cordic_device.h"
#include "cordic_internal.h"

// The unrolled instances that exist (number of live rotations).  Anything
// else runs on the generic kernels (same results, lower throughput).
// (Round 3: every count from 13 / 16 to 29 -- gencordic's own derivation gives
// odd counts, WW for p2r and PW-3 for r2p: 19 live stages for 16-bit sin/cos,
// 27 for 24-bit, 29 for 32-bit and for 24-bit r2p -- and 29 is the most a
// 32-bit phase leaves alive.)
#define CORDIC_ROT_STAGES(X) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) \
	X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29)
#define CORDIC_POL_STAGES(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) \
	X(25) X(26) X(27) X(28) X(29)

namespace cordic_amd {

// stage counts without a static instance run on one instance unrolled to
// kDynStages with a scalar early exit (cordic_device.h: DYN)
constexpr int kDynStages = 40;

// every stage in the GENERAL form
constexpr int kAllGeneral = 1 << 20;

#define CORDIC_ROT_LAUNCHER(NAME) \
	bool NAME(Feed feed, int nlive, int grid, hipStream_t st, \
		const dev::CoreParams &kp, const RotatorJob &j)
#define CORDIC_POL_LAUNCHER(NAME) \
	bool NAME(int nlive, int grid, hipStream_t st, const dev::CoreParams &kp, \
		const int32_t *x, const int32_t *y, int32_t *mag, uint32_t *ph, \
		size_t n)

CORDIC_ROT_LAUNCHER(launch_rot_narrow);		// WW <= 32
CORDIC_ROT_LAUNCHER(launch_rot_lj29);		// WW == 35
CORDIC_ROT_LAUNCHER(launch_rot_lj30);		// WW == 33, 34
CORDIC_ROT_LAUNCHER(launch_rot_lj28);		// WW == 36 .. 40: LJ = 64 - WW,
CORDIC_ROT_LAUNCHER(launch_rot_lj27);		// dynamic-exit instances only
CORDIC_ROT_LAUNCHER(launch_rot_lj26);
CORDIC_ROT_LAUNCHER(launch_rot_lj25);
CORDIC_ROT_LAUNCHER(launch_rot_lj24);
CORDIC_ROT_LAUNCHER(launch_rot_wide2);		// WW <= 35 (kept for A/B)
CORDIC_ROT_LAUNCHER(launch_rot_wide8);		// WW <= 41
CORDIC_ROT_LAUNCHER(launch_rot_wideall);	// WW <= 64
#define CORDIC_SEED_LAUNCHER(NAME) \
	bool NAME(Feed feed, int nlive, int grid, hipStream_t st, \
		const dev::CoreParams &kp, const dev::SeedArgs &sa, \
		const RotatorJob &j, size_t lds_bytes)
CORDIC_SEED_LAUNCHER(launch_seed_narrow);	// WW <= 32
CORDIC_SEED_LAUNCHER(launch_seed_lj29);		// WW == 35
CORDIC_SEED_LAUNCHER(launch_seed_lj30);		// WW == 33, 34
namespace dev { struct DirArgs; }
// per-sample vectors with looked-up directions (cordic_xydir.h)
#define CORDIC_XYDIR_LAUNCHER(NAME) \
	bool NAME(int nlive, int grid, hipStream_t st, const dev::CoreParams &kp, \
		const dev::DirArgs &da, const RotatorJob &j, size_t lds)
CORDIC_XYDIR_LAUNCHER(launch_xydir_lj29);	// WW == 35
CORDIC_XYDIR_LAUNCHER(launch_xydir_lj30);	// WW <= 34
// ... and their tile-reading forms (cordic_jobset: CORDIC_JOBS_P2R_XY / _MIX)
#define CORDIC_XYDIR_JOBS_LAUNCHER(NAME) \
	bool NAME(int nlive, int grid, hipStream_t st, const dev::CoreParams &kp, \
		const dev::DirArgs &da, const TileDescXY *tiles, uint32_t ntiles, \
		size_t lds)
CORDIC_XYDIR_JOBS_LAUNCHER(launch_xydir_jobs_lj29);
CORDIC_XYDIR_JOBS_LAUNCHER(launch_xydir_jobs_lj30);
CORDIC_POL_LAUNCHER(launch_pol_narrow);
// WW <= 32, no reachable overflow: left-justified form, 7 instructions per
// micro-rotation (cordic_device.h: topolar_lj)
CORDIC_POL_LAUNCHER(launch_pol_lj);
// its tile-reading form (cordic_jobset: CORDIC_JOBS_R2P)
bool launch_pol_lj_jobs(int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const TileDescXY *tiles, uint32_t ntiles);
CORDIC_POL_LAUNCHER(launch_pol_ljw);
CORDIC_POL_LAUNCHER(launch_pol_wide8);
CORDIC_POL_LAUNCHER(launch_pol_wideall);
// int16 / uint16 sample arrays (WW <= 32 only: the ports are <= 16 bits);
// one dynamic-exit instance per feed (cordic_inst_io16.hip)
CORDIC_ROT_LAUNCHER(launch_rot_narrow16);
CORDIC_SEED_LAUNCHER(launch_seed_narrow16);
CORDIC_POL_LAUNCHER(launch_pol_narrow16);

} // namespace cordic_amd
#endif
