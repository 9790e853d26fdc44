// cordic_inst_io16.hip -- the kernels on int16 / uint16 sample arrays.
//
// Cores with 16-bit ports (-i 16 -o 16 -p 16) move half
// the bytes per sample when the arrays hold shorts, which is what the seeded
// kernel's HBM bound is made of.  Such cores always fit the 32-bit container
// (WW = 16 + nxtra), so only Narrow32 is instantiated, and only the
// dynamic-exit form (any stage count up to kDynStages).
#include <hip/hip_runtime.h>

#include "cordic_device.h"
#include "cordic_launch.h"

namespace cordic_amd {
namespace {
using namespace dev;

template <Feed FEED>
bool rot16(int nlive, int grid, hipStream_t st, const CoreParams &kp,
		const RotatorJob &j)
{
	if (nlive < 1 || nlive > kDynStages)
		return false;
	if (kp.post_mul != 0)		// CORDIC_FLAG_UNIT_GAIN
		hipLaunchKernelGGL((rotator_unrolled<Narrow32, kDynStages, 0, FEED,
				true, Io16, true>), dim3(grid), dim3(kBlock), 0, st, kp,
			(const i16x4 *)j.x, (const i16x4 *)j.y, (const u16x4 *)j.phase,
			(i16x4 *)j.ox, (i16x4 *)j.oy, j.n / kVec);
	else
		hipLaunchKernelGGL((rotator_unrolled<Narrow32, kDynStages, 0, FEED,
				true, Io16>), dim3(grid), dim3(kBlock), 0, st, kp,
			(const i16x4 *)j.x, (const i16x4 *)j.y, (const u16x4 *)j.phase,
			(i16x4 *)j.ox, (i16x4 *)j.oy, j.n / kVec);
	return true;
}

template <Feed FEED>
bool seed16(int nlive, int grid, hipStream_t st, const CoreParams &kp,
		const SeedArgs &sa, const RotatorJob &j, size_t lds_bytes)
{
	if (nlive < kSeedStages || nlive > kDynStages)
		return false;
	auto kern = (kp.post_mul != 0)
		? rotator_seeded<Narrow32, kDynStages, kSeedStages, FEED, true, Io16,
				true>
		: rotator_seeded<Narrow32, kDynStages, kSeedStages, FEED, true, Io16>;
	// the stage counts 16-bit cores usually have: static instances
	// (-i 16 -o 16 -p 16: 13 live stages; -p 18..20: 16)
	// (the static instances carry the queued sweep only: cordic_inst_body.h)
	const bool queued = sa.queue != nullptr || sa.image_out != nullptr;
	if (kp.post_mul == 0 && nlive == 13 && queued)
		kern = rotator_seeded<Narrow32, 13, kSeedStages, FEED, false, Io16>;
	if (kp.post_mul == 0 && nlive == 16 && queued)
		kern = rotator_seeded<Narrow32, 16, kSeedStages, FEED, false, Io16>;
	hipFuncAttributes attr;	// see cordic_inst_body.h: seeded_kernel_usable
	if (hipFuncGetAttributes(&attr, (const void *)kern) != hipSuccess
			|| attr.sharedSizeBytes != 0
			|| (lds_bytes > 64 * 1024 && hipFuncSetAttribute(
				(const void *)kern,
				hipFuncAttributeMaxDynamicSharedMemorySize,
				(int)lds_bytes) != hipSuccess)) {
		(void)hipGetLastError();
		return false;
	}
	hipLaunchKernelGGL(kern, dim3(grid), dim3(kSeedBlock), lds_bytes, st, kp,
		sa, (const u16x4 *)j.phase, (i16x4 *)j.ox, (i16x4 *)j.oy,
		j.n / kVec);
	return true;
}
} // namespace

bool launch_rot_narrow16(Feed feed, int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const RotatorJob &j)
{
	switch (feed) {
	case Feed::PhaseArray_ConstXY:
		return rot16<Feed::PhaseArray_ConstXY>(nlive, grid, st, kp, j);
	case Feed::PhaseArray_XYArray:
		return rot16<Feed::PhaseArray_XYArray>(nlive, grid, st, kp, j);
	default:
		return rot16<Feed::Nco_ConstXY>(nlive, grid, st, kp, j);
	}
}

bool launch_seed_narrow16(Feed feed, int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const dev::SeedArgs &sa,
		const RotatorJob &j, size_t lds_bytes)
{
	if (feed == Feed::PhaseArray_ConstXY)
		return seed16<Feed::PhaseArray_ConstXY>(nlive, grid, st, kp, sa, j,
				lds_bytes);
	if (feed == Feed::Nco_ConstXY)
		return seed16<Feed::Nco_ConstXY>(nlive, grid, st, kp, sa, j,
				lds_bytes);
	return false;
}

bool launch_pol_narrow16(int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const int32_t *x, const int32_t *y,
		int32_t *mag, uint32_t *ph, size_t n)
{
	if (nlive < 1 || nlive > kDynStages)
		return false;
	if (kp.post_mul != 0)
		hipLaunchKernelGGL((topolar_unrolled<Narrow32, kDynStages, 0, true,
				Io16, true>), dim3(grid), dim3(kBlock), 0, st, kp,
			(const i16x4 *)x, (const i16x4 *)y, (i16x4 *)mag, (u16x4 *)ph,
			n / kVec);
	else
		hipLaunchKernelGGL((topolar_unrolled<Narrow32, kDynStages, 0, true,
				Io16>), dim3(grid), dim3(kBlock), 0, st, kp,
			(const i16x4 *)x, (const i16x4 *)y, (i16x4 *)mag, (u16x4 *)ph,
			n / kVec);
	return true;
}

} // namespace cordic_amd
