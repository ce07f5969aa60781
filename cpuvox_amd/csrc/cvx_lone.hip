// cvx_lone.hip -- the translation unit of the latency kernel (cvx_lone.h): lone_kernel<false> / lone_kernel<true> and their launcher.
//
// Its own file because it is compiled with its own optimisation level (Makefile: OPT_cvx_lone): the batch kernel is fastest at -Os (one large divergent
// loop, profiles/r05_experiments.md), the latency kernel -- wave-uniform control flow around short vector sections -- at -O3 (profiles/r06_experiments.md).
// The arithmetic flags (no contraction, IEEE division, denormals kept) are the same: the float contract is one.
#undef CVX_PROFILE_SECTIONS /* the section profile, its counters and the per-tile clocks belong to the batch kernel's translation unit */
#undef CVX_PROFILE_COUNTS
#undef CVX_TILE_TIMES
#define CVX_DEVICE_FUNCTIONS_ONLY
#include <cstring>

#include "cvx_context.h"
#include "cvx_lone.h"

namespace cvxi {

void LaunchLone(bool hi, unsigned rays, size_t ldsBytes, hipStream_t stream, const DevFrame *frames, const DevTile *tiles, const DevWorld *world)
{
	if (hi) { // windows of more than 2048 pixels (4K): a second mask register
		hipLaunchKernelGGL((cvxk::lone_kernel<true>), dim3(rays), dim3(CVX_WAVE), ldsBytes, stream, frames, tiles, world);
	} else {
		hipLaunchKernelGGL((cvxk::lone_kernel<false>), dim3(rays), dim3(CVX_WAVE), ldsBytes, stream, frames, tiles, world);
	}
}

} // namespace cvxi

extern "C" {
#ifdef CVX_LONE_STATS /* diagnostic variant only (tools/lone_stats.py): event counts of the latency kernel, accumulated over all launches */
int cvx_debug_lone_stats(uint64_t out[96], int reset)
{
	unsigned long long tmp[48];
	if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(tmp, HIP_SYMBOL(cvxk::g_loneStats), sizeof tmp) != hipSuccess) { return CVX_ERR_HIP; }
	for (int i = 0; i < 48; i++) { out[i] = tmp[i]; }
	if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(cvxk::g_loneLongest), sizeof tmp) != hipSuccess) { return CVX_ERR_HIP; }
	for (int i = 0; i < 48; i++) { out[48 + i] = tmp[i]; }
	if (reset) {
		std::memset(tmp, 0, sizeof tmp);
		if (hipMemcpyToSymbol(HIP_SYMBOL(cvxk::g_loneStats), tmp, sizeof tmp) != hipSuccess) { return CVX_ERR_HIP; }
	}
	return CVX_OK;
}
#endif
} // extern "C"
