/* cpuvox_gpu_diag.h -- diagnostics of the EXPERIMENT and PROFILING builds of libcpuvox_gpu (libcpuvox_gpu_exp.so: `make gpu-exp`,
 * -DCVX_EXPERIMENTS; libcpuvox_gpu_prof.so / _count.so: -DCVX_PROFILE_SECTIONS).  NOT part of the drop-in boundary: the product
 * library (libcpuvox_gpu.so, include/cpuvox_gpu.h) neither declares nor exports any of these; the tests of the device arithmetic
 * and the profiling tools load the experiment build for them. */
#ifndef CPUVOX_GPU_DIAG_H
#define CPUVOX_GPU_DIAG_H

#include "cpuvox_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Resident render-kernel workgroups (= waves) per CU the runtime predicts for a given dynamic LDS size. */
int cvx_debug_occupancy(cvx_context *ctx, int64_t ldsBytes, int *blocksPerCU);

/* Diagnostic build only (make gpu-prof, -DCVX_PROFILE_SECTIONS): wave cycles spent per code section of the
 * render kernel (s_memtime stamps), accumulated over all launches.  Sections: 0 prologue/epilogue, 1 phase A
 * (DDA step + header + cull), 2 frustum clip, 3 element walk, 4 side-face setup, 5 side-face pixels,
 * 6 top/bottom setup, 7 top/bottom pixels, 8 skybox pass.  The regular build returns CVX_ERR_NOT_READY. */
int cvx_debug_section_cycles(cvx_context *ctx, uint64_t out[32], int reset); /* [16+i] = cycles/16 * active lanes of section i */
/* Counting diagnostic build only (make gpu-count): how many lanes were active each time a wave executed section i:
 * out[i * 8 + b] = executions with 8b+1 .. 8b+8 active lanes. */
int cvx_debug_section_histogram(cvx_context *ctx, uint64_t out[128], int reset);

/* Arithmetic self-test hook used by tests: evaluates op on n float pairs on
 * the device.  op: 0 a/b, 1 sqrt(a), 2 1/sqrt(a), 3 a*b+c style lerp a+b*(b-a),
 * 4 floor, 5 ceil, 6 round-half-even, 7 (int)a with the x86 rule. */
int cvx_selftest_math(cvx_context *ctx, int op, int n, const float *a, const float *b, float *out);

/* The device prefix sum behind cvx_world_downsample (three launches, cvx_downsample.h) on a caller's array: values[0 .. n) are
 * replaced by their exclusive prefix sums (modulo 2^32), *total receives the 64-bit sum.  For tests of the tail / multi-chunk paths. */
int cvx_selftest_scan(cvx_context *ctx, int n, uint32_t *values, uint64_t *total);

/* The two inline-assembly primitives of the latency kernel (csrc/cvx_lone.h) on a caller's values, one wavefront per case:
 * op 0: the DDA's crossing chains -- out[128 w + n] = a[w] + n additions of b[w], out[128 w + 64 + n] = a[w] + n additions of -b[w] (n = 0 .. 63; a, b: `waves` floats);
 * op 1: v_writelane -- out[64 w + n] = a[64 w + n], except out[64 w + (w % 64)] = b[w] (a: 64 x waves floats, b: waves). */
int cvx_selftest_lone(cvx_context *ctx, int op, int waves, const float *a, const float *b, float *out);

#ifdef __cplusplus
}
#endif
#endif
