// cordic_inst_xydir_body.h -- instantiation unit of rotator_xydir (per-sample
// vectors, looked-up directions: cordic_xydir.h).  The including .hip defines
// CORDIC_XYDIR_NAME (the launcher it exports) and CORDIC_XYDIR_LJ (29: WW 35,
// 30: WW <= 34).  Static instances for every live-stage count of
// CORDIC_ROT_STAGES; any other count keeps the phase-recurrence kernel
// (rotator_unrolled).
#include <hip/hip_runtime.h>

#include "cordic_xydir.h"
#include "cordic_launch.h"

namespace cordic_amd {

namespace {
template <int LJ>
bool launch_lj(int nlive, int grid, hipStream_t st, const dev::CoreParams &kp,
		const dev::DirArgs &da, const RotatorJob &j, size_t lds)
{
	using namespace dev;
	switch (nlive) {
#define X(N) case N: \
	if (da.dx.n != dx_levels(N)) \
		return false; \
	hipLaunchKernelGGL((rotator_xydir<LJ, N>), dim3(grid), dim3(kBlock), lds, st, \
		kp, da, (const i32x4g *)j.x, (const i32x4g *)j.y, \
		(const u32x4g *)j.phase, (i32x4g *)j.ox, (i32x4g *)j.oy, j.n / kVec, \
		(const TileDescXY *)nullptr, 0u); \
	return true;
	CORDIC_ROT_STAGES(X)
#undef X
	default:
		return false;
	}
}
} // namespace

bool CORDIC_XYDIR_NAME(int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const dev::DirArgs &da,
		const RotatorJob &j, size_t lds)
{
	return launch_lj<CORDIC_XYDIR_LJ>(nlive, grid, st, kp, da, j, lds);
}

// Job sets (CORDIC_JOBS_P2R_XY / CORDIC_JOBS_MIX): tile-reading instances for
// the stage counts the including unit lists in CORDIC_XYDIR_JOB_STAGES (the
// BASELINE cores and gencordic's own derivations); other counts run their jobs
// one by one (launch_xy_jobs returns CORDIC_ERR_UNSUPPORTED).
bool CORDIC_XYDIR_JOBS_NAME(int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const dev::DirArgs &da,
		const TileDescXY *tiles, uint32_t ntiles, size_t lds)
{
	using namespace dev;
	switch (nlive) {
#define X(N) case N: \
	if (da.dx.n != dx_levels(N)) \
		return false; \
	hipLaunchKernelGGL((rotator_xydir<CORDIC_XYDIR_LJ, N, true>), dim3(grid), \
		dim3(kBlock), lds, st, kp, da, (const i32x4g *)nullptr, \
		(const i32x4g *)nullptr, (const u32x4g *)nullptr, (i32x4g *)nullptr, \
		(i32x4g *)nullptr, (size_t)0, tiles, ntiles); \
	return true;
	CORDIC_XYDIR_JOB_STAGES(X)
#undef X
	default:
		return false;
	}
}

} // namespace cordic_amd
