// valu_rate.hip -- diagnostic micro-benchmark (not part of the product): what one SIMD of gfx950 pays per wave64
// VALU instruction, by instruction kind and by resident waves per SIMD.  The render kernel is VALU-issue bound
// (DESIGN.md section 8), so these prices -- not instruction counts -- are what its blocks cost.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x

// Every body is 16 instructions on independent registers (v[0..7] read-only sources, v[8..15] destinations unless the
// instruction needs a dependency), repeated ITER times.
#define BODY_LIST(X)                                                                                                                                  \
	X(fma, "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"                                  \
	       "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")                                 \
	X(mul, "v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"                                                  \
	       "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n")                                                 \
	X(add, "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                                                  \
	       "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")                                                 \
	X(pk_mul, "v_pk_mul_f32 %10, %12, %10\n v_pk_mul_f32 %11, %12, %11\n v_pk_mul_f32 %13, %12, %13\n v_pk_mul_f32 %14, %12, %14\n"                        \
	          "v_pk_mul_f32 %10, %12, %10\n v_pk_mul_f32 %11, %12, %11\n v_pk_mul_f32 %13, %12, %13\n v_pk_mul_f32 %14, %12, %14\n")                       \
	X(pk_add, "v_pk_add_f32 %10, %12, %10\n v_pk_add_f32 %11, %12, %11\n v_pk_add_f32 %13, %12, %13\n v_pk_add_f32 %14, %12, %14\n"                        \
	          "v_pk_add_f32 %10, %12, %10\n v_pk_add_f32 %11, %12, %11\n v_pk_add_f32 %13, %12, %13\n v_pk_add_f32 %14, %12, %14\n")                       \
	X(pk_fma, "v_pk_fma_f32 %10, %12, %12, %10\n v_pk_fma_f32 %11, %12, %12, %11\n v_pk_fma_f32 %13, %12, %12, %13\n v_pk_fma_f32 %14, %12, %12, %14\n"    \
	          "v_pk_fma_f32 %10, %12, %12, %10\n v_pk_fma_f32 %11, %12, %12, %11\n v_pk_fma_f32 %13, %12, %12, %13\n v_pk_fma_f32 %14, %12, %12, %14\n")   \
	X(min, "v_min_f32 %0, %8, %0\n v_min_f32 %1, %8, %1\n v_min_f32 %2, %8, %2\n v_min_f32 %3, %8, %3\n"                                                  \
	       "v_min_f32 %4, %8, %4\n v_min_f32 %5, %8, %5\n v_min_f32 %6, %8, %6\n v_min_f32 %7, %8, %7\n")                                                 \
	X(med3, "v_med3_f32 %0, %8, %9, %0\n v_med3_f32 %1, %8, %9, %1\n v_med3_f32 %2, %8, %9, %2\n v_med3_f32 %3, %8, %9, %3\n"                             \
	        "v_med3_f32 %4, %8, %9, %4\n v_med3_f32 %5, %8, %9, %5\n v_med3_f32 %6, %8, %9, %6\n v_med3_f32 %7, %8, %9, %7\n")                            \
	X(cmp_vcc, "v_cmp_lt_f32 vcc, %8, %0\n v_cmp_lt_f32 vcc, %8, %1\n v_cmp_lt_f32 vcc, %8, %2\n v_cmp_lt_f32 vcc, %8, %3\n"                              \
	           "v_cmp_lt_f32 vcc, %8, %4\n v_cmp_lt_f32 vcc, %8, %5\n v_cmp_lt_f32 vcc, %8, %6\n v_cmp_lt_f32 vcc, %8, %7\n")                             \
	X(cmp_sgpr, "v_cmp_lt_f32 s[20:21], %8, %0\n v_cmp_lt_f32 s[22:23], %8, %1\n v_cmp_lt_f32 s[24:25], %8, %2\n v_cmp_lt_f32 s[26:27], %8, %3\n"         \
	            "v_cmp_lt_f32 s[20:21], %8, %4\n v_cmp_lt_f32 s[22:23], %8, %5\n v_cmp_lt_f32 s[24:25], %8, %6\n v_cmp_lt_f32 s[26:27], %8, %7\n")        \
	X(cndmask, "v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n"          \
	           "v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc\n")         \
	X(cmp_cnd, "v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n v_cmp_lt_f32 vcc, %8, %1\n v_cndmask_b32 %1, %8, %1, vcc\n"                    \
	           "v_cmp_lt_f32 vcc, %8, %2\n v_cndmask_b32 %2, %8, %2, vcc\n v_cmp_lt_f32 vcc, %8, %3\n v_cndmask_b32 %3, %8, %3, vcc\n")                   \
	X(rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n") \
	X(div_scale, "v_div_scale_f32 %0, vcc, %8, %9, %8\n v_div_scale_f32 %1, vcc, %8, %9, %8\n v_div_scale_f32 %2, vcc, %8, %9, %8\n v_div_scale_f32 %3, vcc, %8, %9, %8\n" \
	             "v_div_scale_f32 %4, vcc, %8, %9, %8\n v_div_scale_f32 %5, vcc, %8, %9, %8\n v_div_scale_f32 %6, vcc, %8, %9, %8\n v_div_scale_f32 %7, vcc, %8, %9, %8\n") \
	X(div_fmas, "v_div_fmas_f32 %0, %8, %9, %0\n v_div_fmas_f32 %1, %8, %9, %1\n v_div_fmas_f32 %2, %8, %9, %2\n v_div_fmas_f32 %3, %8, %9, %3\n"         \
	            "v_div_fmas_f32 %4, %8, %9, %4\n v_div_fmas_f32 %5, %8, %9, %5\n v_div_fmas_f32 %6, %8, %9, %6\n v_div_fmas_f32 %7, %8, %9, %7\n")        \
	X(div_fixup, "v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n"    \
	             "v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9\n")   \
	X(floor, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n") \
	X(cvt_i32, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n") \
	X(add_u32, "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"                                              \
	           "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n")                                             \
	X(and_b32, "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n"                                              \
	           "v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n")                                             \
	X(lshl_add, "v_lshl_add_u32 %0, %8, 3, %0\n v_lshl_add_u32 %1, %8, 3, %1\n v_lshl_add_u32 %2, %8, 3, %2\n v_lshl_add_u32 %3, %8, 3, %3\n"             \
	            "v_lshl_add_u32 %4, %8, 3, %4\n v_lshl_add_u32 %5, %8, 3, %5\n v_lshl_add_u32 %6, %8, 3, %6\n v_lshl_add_u32 %7, %8, 3, %7\n")            \
	X(mul_lo, "v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %8, %3\n"                                   \
	          "v_mul_lo_u32 %4, %8, %4\n v_mul_lo_u32 %5, %8, %5\n v_mul_lo_u32 %6, %8, %6\n v_mul_lo_u32 %7, %8, %7\n")                                  \
	X(mad_u24, "v_mad_u32_u24 %0, %8, %9, %0\n v_mad_u32_u24 %1, %8, %9, %1\n v_mad_u32_u24 %2, %8, %9, %2\n v_mad_u32_u24 %3, %8, %9, %3\n"              \
	           "v_mad_u32_u24 %4, %8, %9, %4\n v_mad_u32_u24 %5, %8, %9, %5\n v_mad_u32_u24 %6, %8, %9, %6\n v_mad_u32_u24 %7, %8, %9, %7\n")             \
	X(ffbl, "v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7\n") \
	X(mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n") \
	X(salu_and, "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[24:25]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n s_or_b64 s[26:27], s[26:27], s[20:21]\n"        \
	            "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[24:25]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n s_or_b64 s[26:27], s[26:27], s[20:21]\n")       \
	X(salu_valu_mix, "s_and_b64 s[20:21], s[20:21], s[22:23]\n v_add_f32 %0, %8, %0\n s_or_b64 s[22:23], s[22:23], s[24:25]\n v_add_f32 %1, %8, %1\n"                                  \
	                 "s_and_b64 s[24:25], s[24:25], s[26:27]\n v_add_f32 %2, %8, %2\n s_or_b64 s[26:27], s[26:27], s[20:21]\n v_add_f32 %3, %8, %3\n")                                 \
	X(cmp_saveexec, "v_cmp_lt_f32 vcc, %8, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %8, %0\n s_or_b64 exec, exec, s[20:21]\n"                                              \
	                "v_cmp_lt_f32 vcc, %8, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %1, %8, %1\n s_or_b64 exec, exec, s[20:21]\n")                                             \
	X(snop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")                                                                                   \
	X(lshl_add_u64, "v_lshl_add_u64 %10, %12, 3, %10\n v_lshl_add_u64 %11, %12, 3, %11\n v_lshl_add_u64 %13, %12, 3, %13\n v_lshl_add_u64 %14, %12, 3, %14\n" \
	                "v_lshl_add_u64 %10, %12, 3, %10\n v_lshl_add_u64 %11, %12, 3, %11\n v_lshl_add_u64 %13, %12, 3, %13\n v_lshl_add_u64 %14, %12, 3, %14\n")

enum Kind {
#define X(name, text) K_##name,
	BODY_LIST(X)
#undef X
	K_count
};

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, float seed, int iters)
{
	float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
	float m = 1.0000001f + seed * 1e-9f, c = 1e-9f + seed;
	double d0 = a0, d1 = a1, d2 = m, d3 = a3, d4 = a4; // 64-bit register pairs for the packed forms
	for (int i = 0; i < iters; i++) {
#define X(name, text)                                                                                                                              \
	if (KIND == K_##name) {                                                                                                                        \
		asm volatile(text text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                                     \
		             : "v"(m), "v"(c), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4)                                                                  \
		             : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                                                              \
	}
		BODY_LIST(X)
#undef X
	}
	out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d3 + d4);
}

__device__ unsigned long long g_clocks[2];
// the same loop with only the first `active` lanes of every wave alive: does a wave64 instruction cost less when half of the wave is masked off?
__global__ __launch_bounds__(64) void k_masked(float *out, float seed, int iters, int active, int which)
{
	const int lane = threadIdx.x;
	const bool on = which == 0 ? lane < active : (which == 1 ? lane >= 64 - active : (lane % (64 / active)) == 0); // low lanes / high lanes / spread over the wave
	if (!on) { return; }
	const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64(); // shader clock counter / constant 100 MHz counter
	float a0 = seed + lane, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
	float m = 1.0000001f + seed * 1e-9f, c = 1e-9f + seed;
	for (int i = 0; i < iters; i++) {
		asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
		             "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
		             "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
		             "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
		             : "v"(m), "v"(c));
	}
	out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if (blockIdx.x == 0 && lane == (which == 1 ? 63 : 0)) {
		g_clocks[0] = __builtin_readcyclecounter() - c0;
		g_clocks[1] = wall_clock64() - r0;
	}
}

void run_masked(float *d, int cus, double ghz)
{
	const int iters = 20000, instPerIter = 16;
	const char *names[3] = { "low lanes", "high lanes", "spread" };
	for (int wavesPerSimd : { 1, 2, 4, 8 })
	for (int which = 0; which < 3; which += 2) {
		printf("fma, %d w, %-10s", wavesPerSimd, names[which]);
		for (int active : { 64, 32, 16, 8, 4, 2, 1 }) {
			const int blocks = cus * 4 * wavesPerSimd;
			hipEvent_t e0, e1;
			(void)hipEventCreate(&e0);
			(void)hipEventCreate(&e1);
			hipLaunchKernelGGL(k_masked, dim3(blocks), dim3(64), 0, 0, d, 1.0f, 200, active, which);
			(void)hipDeviceSynchronize();
			(void)hipEventRecord(e0);
			hipLaunchKernelGGL(k_masked, dim3(blocks), dim3(64), 0, 0, d, 1.0f, iters, active, which);
			(void)hipEventRecord(e1);
			(void)hipEventSynchronize(e1);
			float ms = 0;
			(void)hipEventElapsedTime(&ms, e0, e1);
			unsigned long long clk[2] = { 0, 0 };
			(void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clocks), sizeof clk);
			printf("  %2d lanes: %5.2f (%4.0f MHz)", active, ms * 1e-3 * ghz * 1e9 / ((double)iters * instPerIter * wavesPerSimd), clk[1] ? (double)clk[0] / (double)clk[1] * 100.0 : 0.0);
		}
		printf("\n");
	}
}

template <int KIND>
void run(const char *name, float *d, int cus, double ghz)
{
	const int iters = 20000, instPerIter = 16;
	printf("%-14s", name);
	for (int wavesPerSimd : { 1, 2, 3, 4, 8 }) {
		const int blocks = cus * 4 * wavesPerSimd;
		hipEvent_t e0, e1;
		(void)hipEventCreate(&e0);
		(void)hipEventCreate(&e1);
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 1.0f, 200); // warm-up
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 1.0f, iters);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		float ms = 0;
		(void)hipEventElapsedTime(&ms, e0, e1);
		const double instPerSimd = (double)iters * instPerIter * wavesPerSimd;
		printf("  %d w: %6.2f", wavesPerSimd, ms * 1e-3 * ghz * 1e9 / instPerSimd);
	}
	printf("\n");
}

int main()
{
	hipDeviceProp_t p;
	(void)hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	const double ghz = 2.4;
	printf("%d CUs; cycles (at a nominal %.1f GHz: the real clock under load is lower, compare rows) per wave64 instruction per SIMD, by resident waves per SIMD\n", cus, ghz);
	float *d;
	(void)hipMalloc(&d, (size_t)cus * 4 * 8 * 64 * 4);
#define X(name, text) run<K_##name>(#name, d, cus, ghz);
	BODY_LIST(X)
#undef X
	printf("\nactive lanes (4 waves per SIMD): cycles per wave64 v_fma_f32 per SIMD\n");
	run_masked(d, cus, ghz);
	return 0;
}
