// cvx_lone.hip -- the translation unit of the latency kernel (cvx_lone.h): lone_kernel<false> / lone_kernel<true> and their launcher.
//
// Its own file because it is compiled with its own optimisation level (Makefile: OPT_cvx_lone): the batch kernel is fastest at -Os (one large divergent
// loop, profiles/r05_experiments.md), the latency kernel -- wave-uniform control flow around short vector sections -- at -O3 (profiles/r06_experiments.md).
// The arithmetic flags (no contraction, IEEE division, denormals kept) are the same: the float contract is one.
#if defined(CVX_EXPERIMENTS) || defined(CVX_PROFILE_SECTIONS) /* the builds that export include/cpuvox_gpu_diag.h */
#define CVX_LONE_DIAGNOSTICS
#endif
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

#ifdef CVX_LONE_DIAGNOSTICS /* include/cpuvox_gpu_diag.h: the two inline-assembly primitives of the latency kernel on a caller's values (tests/test_gpu_parity.py) */
namespace cvxk {
// op 0: wave w runs lone_crossing_chains from (a[w], a[w]) with the steps (b[w], -b[w]): out[128 w + lane] = X of the lane, out[128 w + 64 + lane] = Z
// op 1: wave w holds a[64 w + lane] in its lanes and writes b[w] into lane (w % 64) with write_lane: out[64 w + lane] = the register afterwards
__global__ __launch_bounds__(CVX_WAVE) void selftest_lone_kernel(int op, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out)
{
	const int w = (int)blockIdx.x, lane = (int)threadIdx.x;
	if (op == 0) {
		float X = a[w], Z = a[w];
		lone_crossing_chains(X, Z, b[w], -b[w]);
		out[(size_t)w * 128 + lane] = X;
		out[(size_t)w * 128 + 64 + lane] = Z;
	} else {
		const uint32_t vec = __float_as_uint(a[(size_t)w * 64 + lane]);
		const uint32_t value = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(b[w]));
		out[(size_t)w * 64 + lane] = __uint_as_float(write_lane(vec, value, w & 63));
	}
}
} // namespace cvxk

extern "C" int cvx_selftest_lone(cvx_context *ctx, int op, int waves, const float *a, const float *b, float *out)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (waves <= 0 || (op != 0 && op != 1) || !a || !b || !out) { return cvxi::Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	if (hipSetDevice(ctx->device) != hipSuccess) { return cvxi::Fail(ctx, CVX_ERR_HIP, "hipSetDevice failed"); }
	const size_t na = op == 0 ? (size_t)waves : (size_t)waves * 64, nb = (size_t)waves, no = op == 0 ? (size_t)waves * 128 : (size_t)waves * 64;
	float *d = nullptr;
	if (hipMalloc((void **)&d, (na + nb + no) * sizeof(float)) != hipSuccess) { return cvxi::Fail(ctx, CVX_ERR_HIP, "selftest buffers: out of memory"); }
	hipError_t e = hipMemcpyAsync(d, a, na * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) { e = hipMemcpyAsync(d + na, b, nb * sizeof(float), hipMemcpyHostToDevice, ctx->stream); }
	if (e == hipSuccess) {
		hipLaunchKernelGGL(cvxk::selftest_lone_kernel, dim3((unsigned)waves), dim3(CVX_WAVE), 0, ctx->stream, op, d, d + na, d + na + nb);
		e = hipGetLastError();
	}
	if (e == hipSuccess) { e = hipMemcpyAsync(out, d + na + nb, no * sizeof(float), hipMemcpyDeviceToHost, ctx->stream); }
	if (e == hipSuccess) { e = hipStreamSynchronize(ctx->stream); }
	(void)hipFree(d);
	return e == hipSuccess ? CVX_OK : cvxi::Fail(ctx, CVX_ERR_HIP, "selftest failed: %s", hipGetErrorString(e));
}
#endif

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
