// cordic_inst_pol_lj.hip -- instances of the left-justified converter
// (cordic_device.h: topolar_lj): r2p / sr2p cores with WW <= 32 whose registers
// cannot overflow.
#include <hip/hip_runtime.h>

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
	switch (nlive) {
#define X(N) case N: \
	hipLaunchKernelGGL((topolar_lj<N>), dim3(grid), dim3(kBlock), 0, st, kp, \
		(const i32x4 *)x, (const i32x4 *)y, (i32x4 *)mag, (u32x4 *)ph, \
		n / kVec); \
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

} // namespace cordic_amd
