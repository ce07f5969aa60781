// What a TAKEN scalar branch costs a wave (gfx950): hipcc --offload-arch=gfx950 -O2 -o /tmp/branch_cost tools/branch_cost.hip && /tmp/branch_cost
// Three loops of 64 v_add each per iteration: straight-line; with an s_branch to the very next instruction after every v_add (taken, nothing skipped); with an
// s_cbranch_scc1 that is never taken.  Cycles per iteration by s_memtime, for 1 / 2 / 3 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int KIND>
__global__ void k(unsigned long long *out, int iters, float *sink)
{
	float a = threadIdx.x * 1.0f, b = 1.5f;
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < iters; i++) {
		if (KIND == 0) { asm volatile(REP64("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b)); }
		if (KIND == 1) { asm volatile(REP64("v_add_f32 %0, %0, %1\n s_branch 0\n") : "+v"(a) : "v"(b)); }
		if (KIND == 2) { asm volatile("s_cmp_eq_u32 0, 1\n" REP64("v_add_f32 %0, %0, %1\n s_cbranch_scc1 0\n") : "+v"(a) : "v"(b) : "scc"); }
		if (KIND == 3) { asm volatile(REP64("v_add_f32 %0, %0, %1\n s_nop 0\n") : "+v"(a) : "v"(b)); }
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
	sink[blockIdx.x * 64 + threadIdx.x] = a;
}
int main()
{
	const int iters = 2000;
	unsigned long long *d; float *s;
	for (int wavesPerSimd = 1; wavesPerSimd <= 4; wavesPerSimd++) {
		const int blocks = 256 * 4 * wavesPerSimd;
		hipMalloc(&d, blocks * 8); hipMalloc(&s, blocks * 64 * 4);
		const char *names[4] = { "64 v_add", "64 x (v_add + taken s_branch)", "64 x (v_add + not-taken s_cbranch)", "64 x (v_add + s_nop)" };
		for (int kind = 0; kind < 4; kind++) {
			for (int rep = 0; rep < 2; rep++) {
				if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, iters, s);
				if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, iters, s);
				if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, iters, s);
				if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d, iters, s);
				hipDeviceSynchronize();
			}
			std::vector<unsigned long long> h(blocks);
			hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
			double sum = 0; for (auto v : h) sum += (double)v;
			printf("%d waves per SIMD: %-38s %8.2f clock ticks per v_add (+ its branch)\n", wavesPerSimd, names[kind], sum / blocks / iters / 64.0);
		}
		hipFree(d); hipFree(s);
	}
	return 0;
}
