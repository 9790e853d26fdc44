// cordic_inst_body.h -- one instantiation unit of the unrolled kernels.
// The including .hip file defines
//   CORDIC_INST_KIND       1 = rotator (p2r), 2 = converter (r2p),
//                          3 = seeded rotator (constant vector)
//   CORDIC_INST_NAME       name of the launcher this unit exports
//   CORDIC_INST_CONTAINER  dev::Narrow32 or dev::Wide64
//   CORDIC_INST_NGEN       leading stages in the GENERAL 64-bit form
// Splitting the instances over several translation units keeps the build
// parallel (each unit compiles in ~15 s).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "cordic_device.h"
#include "cordic_launch.h"

namespace cordic_amd {

// A/B knob (measurement only): CORDIC_FORCE_DYN=1 sends every launch of this
// unit to its dynamic-exit instance, so that a static instance can be weighed
// against it on the same box (profiles/r05/static_vs_dyn.txt decided which
// static instances the library keeps)
static inline bool force_dyn()
{
	static const bool v = [] {
		const char *e = std::getenv("CORDIC_FORCE_DYN");
		return e && e[0] == '1';
	}();
	return v;
}

#if CORDIC_INST_KIND == 1
namespace {
template <Feed FEED>
bool launch_feed(int nlive, int grid, hipStream_t st, const dev::CoreParams &kp,
		const RotatorJob &j)
{
	using namespace dev;
	constexpr int G = CORDIC_INST_NGEN;
	// fused output scaling (CORDIC_FLAG_UNIT_GAIN): its own dynamic-exit
	// instance, so that the static instances carry no trace of it
	if (kp.post_mul != 0) {
		if (nlive < 1 || nlive > kDynStages)
			return false;
		hipLaunchKernelGGL((rotator_unrolled<CORDIC_INST_CONTAINER,
				kDynStages, (G > kDynStages ? kDynStages : G), FEED,
				true, Io32, true>), dim3(grid), dim3(kBlock), 0, st, kp,
			(const i32x4 *)j.x, (const i32x4 *)j.y,
			(const u32x4 *)j.phase, (i32x4 *)j.ox, (i32x4 *)j.oy,
			j.n / kVec);
		return true;
	}
	switch (force_dyn() ? -1 : nlive) {
#ifndef CORDIC_INST_DYN_ONLY	// (units that carry the dynamic-exit instance only)
#define X(N) case N: \
	hipLaunchKernelGGL((rotator_unrolled<CORDIC_INST_CONTAINER, N, \
			(G > N ? N : G), FEED>), dim3(grid), dim3(kBlock), 0, st, \
		kp, (const i32x4 *)j.x, (const i32x4 *)j.y, \
		(const u32x4 *)j.phase, (i32x4 *)j.ox, (i32x4 *)j.oy, j.n / kVec); \
	return true;
	CORDIC_ROT_STAGES(X)
#undef X
#endif
	default:
		// any other count up to kDynStages: the dynamic-exit instance
		if (nlive < 1 || nlive > kDynStages)
			return false;
		hipLaunchKernelGGL((rotator_unrolled<CORDIC_INST_CONTAINER,
				kDynStages, (G > kDynStages ? kDynStages : G), FEED,
				true>), dim3(grid), dim3(kBlock), 0, st, kp,
			(const i32x4 *)j.x, (const i32x4 *)j.y,
			(const u32x4 *)j.phase, (i32x4 *)j.ox, (i32x4 *)j.oy,
			j.n / kVec);
		return true;
	}
}
} // namespace

bool CORDIC_INST_NAME(Feed feed, int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const RotatorJob &j)
{
	switch (feed) {
	case Feed::PhaseArray_ConstXY:
		return launch_feed<Feed::PhaseArray_ConstXY>(nlive, grid, st, kp, j);
	case Feed::PhaseArray_XYArray:
		return launch_feed<Feed::PhaseArray_XYArray>(nlive, grid, st, kp, j);
	default:
		return launch_feed<Feed::Nco_ConstXY>(nlive, grid, st, kp, j);
	}
}
#elif CORDIC_INST_KIND == 3
namespace {
// rotator_seeded addresses LDS by byte offset from 0, which holds as long as
// the kernel has no static LDS in front of its dynamic array.  A build that
// adds some (instrumentation, a debug option) is detected here, once per
// kernel, and the job then runs on the full-recurrence kernel instead of
// trapping on the device.  Also raises the dynamic-LDS limit.
inline bool seeded_kernel_usable(const void *kern, size_t lds_bytes)
{
	// function attributes and LDS limits are per DEVICE: the cache of
	// verified kernels is keyed on (device, kernel), and the table has to
	// fit what this device offers per block (160 KiB on gfx950; a part with
	// less simply runs the full-recurrence kernel)
	int dev = 0, lds_max = 0;
	if (hipGetDevice(&dev) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	constexpr int kCache = 64;
	struct Seen { const void *kern; int dev; int lds_max; };
	thread_local Seen ok[kCache] = {};
	for (int i = 0; i < kCache; i++)
		if (ok[i].kern == kern && ok[i].dev == dev)
			return lds_bytes <= (size_t)ok[i].lds_max;
	hipFuncAttributes attr;
	int optin = 0;
	if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin,
				dev) != hipSuccess) {
		(void)hipGetLastError();
		optin = 0;
	}
	if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock,
				dev) != hipSuccess
			|| lds_bytes > (size_t)(lds_max = lds_max > optin ? lds_max : optin)
			|| hipFuncGetAttributes(&attr, kern) != hipSuccess
			|| attr.sharedSizeBytes != 0
			|| hipFuncSetAttribute(kern,
				hipFuncAttributeMaxDynamicSharedMemorySize, lds_max)
				!= hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	for (int i = 0; i < kCache; i++)
		if (!ok[i].kern) {
			ok[i] = Seen{kern, dev, lds_max};
			break;
		}
	return true;
}

template <Feed FEED>
bool launch_seeded(int nlive, int grid, hipStream_t st, const dev::CoreParams &kp,
		const dev::SeedArgs &sa, const RotatorJob &j, size_t lds_bytes)
{
	using namespace dev;
	if (kp.post_mul != 0) {		// see launch_feed
		if (nlive < kSeedStages || nlive > kDynStages)
			return false;
		auto kern = rotator_seeded<CORDIC_INST_CONTAINER, kDynStages,
				kSeedStages, FEED, true, Io32, true>;
		if (!seeded_kernel_usable((const void *)kern, lds_bytes))
			return false;
		hipLaunchKernelGGL(kern, dim3(grid), dim3(kSeedBlock), lds_bytes, st,
			kp, sa, (const u32x4 *)j.phase, (i32x4 *)j.ox, (i32x4 *)j.oy,
			j.n / kVec);
		return true;
	}
	// Direction tails pay where the lanes of a wave read neighbouring table
	// entries.  A phase ARRAY is judged by the kernel itself, wave by wave; an
	// NCO's phases are known here: a row of 64 lanes x 4 samples spans
	// 256 increments, which has to stay under the 2^kDtCoherentLog2 the
	// kernel's own test allows -- otherwise the plain instance runs, without
	// that test.
	bool tails_pay = true;
	if (FEED == Feed::Nco_ConstXY) {
		const int32_t f = (int32_t)kp.fcw;	// left-justified increment
		tails_pay = kDtCoherentLog2 >= 31	// (A/B builds: every row)
			|| (f < 0 ? -(int64_t)f : (int64_t)f) < ((int64_t)1 << (kDtCoherentLog2 - 8));
	}
	// a batch (cordic_jobset: tile descriptors) runs the dynamic-exit instance,
	// the one whose tile loop reads them
	(void)tails_pay;	// (units that carry the dynamic-exit instance only)
	// ... and so does a launch without a tile queue (the static instances
	// carry the queued sweep only; build mode needs neither)
#ifdef CORDIC_DESC_LOOP_ALL
	const bool batch_needs_dyn = false;	// A/B: every instance reads descriptors
#else
	const bool batch_needs_dyn = sa.tiles != nullptr;
#endif
	auto go = [&](auto kern) {
		if (!seeded_kernel_usable((const void *)kern, lds_bytes))
			return false;
		hipLaunchKernelGGL(kern, dim3(grid), dim3(kSeedBlock), lds_bytes, st, kp, sa,
			(const u32x4 *)j.phase, (i32x4 *)j.ox, (i32x4 *)j.oy, j.n / kVec);
		return true;
	};
#ifdef CORDIC_INST_DESC_STATIC
	// a job set on 16 / 24 stages: the static instance that reads descriptors
	if (sa.tiles != nullptr && !force_dyn() && desc_static(nlive, sa.dt.n)) {
		if (nlive == 16)
			return go(rotator_seeded<CORDIC_INST_CONTAINER, 16, kSeedStages, FEED,
					false, Io32, false, true, true>);
		return go(rotator_seeded<CORDIC_INST_CONTAINER, 24, kSeedStages, FEED,
				false, Io32, false, true, true>);
	}
#endif
	switch ((batch_needs_dyn || (!sa.queue && !sa.image_out) || force_dyn()) ? -1 : nlive) {
#ifndef CORDIC_INST_DYN_ONLY
	// static instances; where the plan carries direction tails for the
	// stages behind the seeds (left-justified cores with kDtMinStages or more
	// of them), the instance that looks their multipliers up.  A phase ARRAY
	// on such a core -- and an NCO where every row looks up (dt_always) -- runs
	// without its tails under CORDIC_FLAG_NO_TAILS only (an A/B flag): no
	// static instance for that, the dynamic-exit one below.
#define X(N) case N: { \
	constexpr bool kTails = (CORDIC_INST_CONTAINER::lj != 0 || !CORDIC_INST_CONTAINER::wide) \
			&& dt_levels(N - kSeedStages) >= 1 \
			&& dt_levels(N - kSeedStages) <= kDtMaxLevels; \
	if constexpr (kTails) { \
		if (sa.dt.n == dt_levels(N - kSeedStages) \
				&& (tails_pay || dt_always(N - kSeedStages))) \
			return go(rotator_seeded<CORDIC_INST_CONTAINER, N, kSeedStages, \
					FEED, false, Io32, false, true>); \
	} \
	if constexpr (kTails && (FEED == Feed::PhaseArray_ConstXY \
				|| dt_always(N - kSeedStages))) \
		break; \
	else \
		return go(rotator_seeded<CORDIC_INST_CONTAINER, N, kSeedStages, FEED>); }
	CORDIC_ROT_STAGES(X)
#undef X
#endif
	default:
		break;
	}
	if (nlive < kSeedStages || nlive > kDynStages)
		return false;
	return go(rotator_seeded<CORDIC_INST_CONTAINER, kDynStages, kSeedStages, FEED, true>);
}
} // namespace

bool CORDIC_INST_NAME(Feed feed, int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const dev::SeedArgs &sa,
		const RotatorJob &j, size_t lds_bytes)
{
	if (feed == Feed::PhaseArray_ConstXY)
		return launch_seeded<Feed::PhaseArray_ConstXY>(nlive, grid, st, kp,
				sa, j, lds_bytes);
	if (feed == Feed::Nco_ConstXY)
		return launch_seeded<Feed::Nco_ConstXY>(nlive, grid, st, kp, sa, j,
				lds_bytes);
	return false;
}
#else
bool CORDIC_INST_NAME(int nlive, int grid, hipStream_t st,
		const dev::CoreParams &kp, const int32_t *x, const int32_t *y,
		int32_t *mag, uint32_t *ph, size_t n)
{
	using namespace dev;
	constexpr int G = CORDIC_INST_NGEN;
	if (kp.post_mul != 0) {		// see launch_feed
		if (nlive < 1 || nlive > kDynStages)
			return false;
		hipLaunchKernelGGL((topolar_unrolled<CORDIC_INST_CONTAINER, kDynStages,
				(G > kDynStages ? kDynStages : G), true, Io32, true>),
			dim3(grid), dim3(kBlock), 0, st, kp, (const i32x4 *)x,
			(const i32x4 *)y, (i32x4 *)mag, (u32x4 *)ph, n / kVec);
		return true;
	}
	switch (force_dyn() ? -1 : nlive) {
#ifndef CORDIC_INST_DYN_ONLY
#define X(N) case N: \
	hipLaunchKernelGGL((topolar_unrolled<CORDIC_INST_CONTAINER, N, \
			(G > N ? N : G)>), dim3(grid), dim3(kBlock), 0, st, kp, \
		(const i32x4 *)x, (const i32x4 *)y, (i32x4 *)mag, (u32x4 *)ph, \
		n / kVec); \
	return true;
	CORDIC_POL_STAGES(X)
#undef X
#endif
	default:
		if (nlive < 1 || nlive > kDynStages)
			return false;
		hipLaunchKernelGGL((topolar_unrolled<CORDIC_INST_CONTAINER, kDynStages,
				(G > kDynStages ? kDynStages : G), true>), dim3(grid),
			dim3(kBlock), 0, st, kp, (const i32x4 *)x, (const i32x4 *)y,
			(i32x4 *)mag, (u32x4 *)ph, n / kVec);
		return true;
	}
}
#endif

} // namespace cordic_amd
