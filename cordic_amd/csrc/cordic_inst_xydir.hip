// cordic_inst_xydir.hip -- instantiation unit of rotator_xydir (per-sample
// vectors, looked-up directions: cordic_xydir.h).  Static instances for the
// live-stage counts listed in CORDIC_XYDIR_STAGES, in the two left-justified
// containers the WW <= 35 cores run in; every other count keeps the
// phase-recurrence kernel (rotator_unrolled).
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
		(const u32x4g *)j.phase, (i32x4g *)j.ox, (i32x4g *)j.oy, j.n / kVec); \
	return true;
	CORDIC_XYDIR_STAGES(X)
#undef X
	default:
		return false;
	}
}
} // namespace

bool launch_xydir(int lj, int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const dev::DirArgs &da,
		const RotatorJob &j, size_t lds)
{
	if (lj == 29)
		return launch_lj<29>(nlive, grid, st, kp, da, j, lds);
	if (lj == 30)
		return launch_lj<30>(nlive, grid, st, kp, da, j, lds);
	return false;
}

} // namespace cordic_amd
