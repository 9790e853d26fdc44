// cordic_inst_pol_lj.hip -- instances of the left-justified converter
// (cordic_device.h: topolar_lj, topolar_ljw): r2p / sr2p cores with WW <= 34 /
// WW 35 .. 40 whose registers cannot overflow.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "cordic_device.h"
#include "cordic_launch.h"

namespace cordic_amd {

bool launch_pol_lj(int nlive, int grid, hipStream_t st, const dev::CoreParams &kp,
		const int32_t *x, const int32_t *y, int32_t *mag, uint32_t *ph,
		size_t n)
{
	using namespace dev;
	if (nlive < 1 || nlive > kDynStages)
		return false;
	if (kp.post_mul != 0) {		// CORDIC_FLAG_UNIT_GAIN: dynamic-exit instance
		hipLaunchKernelGGL((topolar_lj<kDynStages, true, Io32, true>),
			dim3(grid), dim3(kBlock), 0, st, kp, (const i32x4 *)x,
			(const i32x4 *)y, (i32x4 *)mag, (u32x4 *)ph, n / kVec);
		return true;
	}
	// the static instances are PLAIN: WW <= 32 and rounding at the 2^30 scale
	const bool plain = (32 - kp.iw) - kp.in_shl >= 2 && kp.r >= 2 && kp.r <= 31;
	static const bool force_dyn = [] {	// A/B knob: cordic_inst_body.h
		const char *e = std::getenv("CORDIC_FORCE_DYN");
		return e && e[0] == '1';
	}();
	switch ((plain && !force_dyn) ? nlive : -1) {
#define X(N) case N: \
	hipLaunchKernelGGL((topolar_lj<N, false, Io32, false, true>), dim3(grid), \
		dim3(kBlock), 0, st, kp, (const i32x4 *)x, (const i32x4 *)y, \
		(i32x4 *)mag, (u32x4 *)ph, n / kVec); \
	return true;
	CORDIC_POL_STAGES(X)
#undef X
	default:
		hipLaunchKernelGGL((topolar_lj<kDynStages, true>), dim3(grid),
			dim3(kBlock), 0, st, kp, (const i32x4 *)x, (const i32x4 *)y,
			(i32x4 *)mag, (u32x4 *)ph, n / kVec);
		return true;
	}
}

// Job sets (CORDIC_JOBS_R2P): the 20-stage core of BASELINE config 3 and the
// 29 stages gencordic derives for 24-bit ports on static instances, every
// other count on the dynamic-exit one; unit gain: the jobs one by one.
bool launch_pol_lj_jobs(int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const TileDescXY *tiles, uint32_t ntiles)
{
	using namespace dev;
	if (nlive < 1 || nlive > kDynStages || kp.post_mul != 0)
		return false;
	const bool plain = (32 - kp.iw) - kp.in_shl >= 2 && kp.r >= 2 && kp.r <= 31;
	switch (plain ? nlive : -1) {
	case 20:
		hipLaunchKernelGGL((topolar_lj_jobs<20, false, true>), dim3(grid),
			dim3(kBlock), 0, st, kp, tiles, ntiles);
		return true;
	case 29:
		hipLaunchKernelGGL((topolar_lj_jobs<29, false, true>), dim3(grid),
			dim3(kBlock), 0, st, kp, tiles, ntiles);
		return true;
	default:
		hipLaunchKernelGGL((topolar_lj_jobs<kDynStages, true, false>), dim3(grid),
			dim3(kBlock), 0, st, kp, tiles, ntiles);
		return true;
	}
}

// WW = 35 .. 40: one dynamic-exit instance per width (LJ = 64 - WW)
namespace {
template <int LJ>
void launch_ljw(int grid, hipStream_t st, const dev::CoreParams &kp, const int32_t *x,
		const int32_t *y, int32_t *mag, uint32_t *ph, size_t n)
{
	using namespace dev;
	if (kp.post_mul != 0)
		hipLaunchKernelGGL((topolar_ljw<LJ, kDynStages, Io32, true>), dim3(grid),
			dim3(kBlock), 0, st, kp, (const i32x4 *)x, (const i32x4 *)y,
			(i32x4 *)mag, (u32x4 *)ph, n / kVec);
	else
		hipLaunchKernelGGL((topolar_ljw<LJ, kDynStages>), dim3(grid),
			dim3(kBlock), 0, st, kp, (const i32x4 *)x, (const i32x4 *)y,
			(i32x4 *)mag, (u32x4 *)ph, n / kVec);
}
} // namespace

bool launch_pol_ljw(int nlive, int grid, hipStream_t st, const dev::CoreParams &kp,
		const int32_t *x, const int32_t *y, int32_t *mag, uint32_t *ph,
		size_t n)
{
	using namespace dev;
	if (nlive < 1 || nlive > kDynStages)
		return false;
	switch (kp.iw + kp.in_shl + 2) {	// WW (r2p: in_shl = WW - IW - 2)
	case 35: launch_ljw<29>(grid, st, kp, x, y, mag, ph, n); return true;
	case 36: launch_ljw<28>(grid, st, kp, x, y, mag, ph, n); return true;
	case 37: launch_ljw<27>(grid, st, kp, x, y, mag, ph, n); return true;
	case 38: launch_ljw<26>(grid, st, kp, x, y, mag, ph, n); return true;
	case 39: launch_ljw<25>(grid, st, kp, x, y, mag, ph, n); return true;
	case 40: launch_ljw<24>(grid, st, kp, x, y, mag, ph, n); return true;
	default: return false;
	}
}

} // namespace cordic_amd
