// fetch_calibration.hip -- diagnostic (not part of the product): what rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* report on gfx950 for the
// access shapes of the render kernel, against byte counts known by construction (MI355X_MICROARCH.md, HBM: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//
// Kernels (each launched once per run, so a --pmc pass gives one row per kernel name):
//   cal_stream16        coalesced 16 B per lane over a buffer far larger than the Infinity Cache (the guide's x 1/2 case: control)
//   cal_gather<32,1>    one random 32-byte record per 128-byte line, every line of the table exactly once (two global_load_dwordx4,
//                       scalar base + 32-bit lane offset: the render kernel's record fetch); L2 hit rate 0 by construction
//   cal_gather<32,4>    all four 32-byte records of every line, each by a different lane at an unrelated time (a permutation of the records)
//   cal_gather<64,1> / <128,1>   one 64-byte half / the whole line per lane (4 / 8 loads)
//   cal_gather<4,1>     one dword per line (the colour loads)
//   cal_walk            64 adjacent "rays" per wave walk a row-major 2048 x 2048 table of 32-byte records with a DDA, one record
//                       per step and lane: the render kernel's own locality (neighbouring lanes and steps share lines)
// Every kernel writes a checksum so that nothing is optimised away.  Sizes: the small table is 134 MB (the 2048^2 record table of
// LOD 0, inside the 256 MiB Infinity Cache), the big one 4 GiB (HBM): the big runs also say how many bytes per record the memory
// system really moves, from the record rate it sustains against ~6 TB/s.
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_cal tools/fetch_calibration.hip
//   /tmp/fetch_cal                        # times + algorithmic bytes per kernel (CSV on stdout)
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/p1 -- /tmp/fetch_cal      (one pass per counter group: tools/fetch_calibration.sh)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))

__global__ __launch_bounds__(256) void cal_stream16(const u32x4 *__restrict__ p, size_t n16, uint32_t *out)
{
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
		const u32x4 v = p[i];
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u) { out[0] = acc; }
}

// BYTES per record read by one lane (4: one dword), PER_LINE records of every 128-byte line are read (by different lanes).
// lines = power of two.  Lane g reads, for r = 0 .. rounds-1, record number perm(g * rounds + r) of the lines * PER_LINE records;
// perm(i) = (i * odd + c) mod 2^k is a bijection, so every record is read exactly once when grid * rounds = lines * PER_LINE.
template <int BYTES, int PER_LINE>
__global__ __launch_bounds__(256) void cal_gather(const uint8_t *__restrict__ table, uint32_t lineMask, int rounds, uint32_t mulOdd, uint32_t *out)
{
	const GLOBAL uint8_t *base = (const GLOBAL uint8_t *)table;
	const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t acc = 0;
	const uint32_t recMask = (lineMask + 1u) * PER_LINE - 1u;
#pragma unroll 4
	for (int r = 0; r < rounds; r++) {
		const uint32_t i = (uint32_t)(g * (uint64_t)rounds + (uint64_t)r);
		const uint32_t rec = (i * mulOdd + 0x9E3779B9u) & recMask;
		const uint32_t hsh = (rec * 0x85EBCA6Bu) >> 27; // 5 random bits
		const uint32_t line = rec / PER_LINE;
		// where in its 128-byte line the record starts: PER_LINE == 4: record rec % 4; else a random BYTES-aligned place
		const uint32_t inLine = PER_LINE == 4 ? (rec % 4u) * 32u : (BYTES >= 128 ? 0u : (hsh * (uint32_t)BYTES) & 127u);
		const uint64_t off = (uint64_t)line * 128u + inLine;
		if (BYTES == 4) {
			acc += *(const GLOBAL uint32_t *)(base + off);
		} else {
#pragma unroll
			for (int k = 0; k < BYTES / 16; k++) {
				const u32x4 v = *(const GLOBAL u32x4 *)(base + (off + (uint64_t)k * 16u));
				acc += v.x ^ v.y ^ v.z ^ v.w;
			}
		}
	}
	if (acc == 0x12345678u) { out[0] = acc; }
}

// a wave = 64 neighbouring rays from one start point; table = dim x dim records of 32 bytes, row-major (index = (x << shift) + z)
__global__ __launch_bounds__(64) void cal_walk(const uint8_t *__restrict__ table, int shift, int steps, uint32_t *out)
{
	const GLOBAL uint8_t *base = (const GLOBAL uint8_t *)table;
	const int dim = 1 << shift;
	uint32_t h = blockIdx.x * 0x9E3779B9u + 0x7F4A7C15u;
	h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
	const float startX = (float)(h & (dim - 1)) + 0.37f, startZ = (float)((h >> 12) & (dim - 1)) + 0.61f;
	const float angle = (float)(h >> 24) * (6.2831853f / 256.0f) + (float)threadIdx.x * (1.0f / 1920.0f); // adjacent rays: about a pixel apart at 1080p
	const float dx = __cosf(angle), dz = __sinf(angle);
	int px = (int)startX, pz = (int)startZ;
	const int sx = dx > 0 ? 1 : -1, sz = dz > 0 ? 1 : -1;
	const float tdx = 1.0f / fmaxf(1e-7f, fabsf(dx)), tdz = 1.0f / fmaxf(1e-7f, fabsf(dz));
	float tmx = (dx > 0 ? 1.0f - (startX - (float)px) : startX - (float)px) * tdx, tmz = (dz > 0 ? 1.0f - (startZ - (float)pz) : startZ - (float)pz) * tdz;
	uint32_t acc = 0;
	for (int s = 0; s < steps; s++) {
		const uint32_t off = ((((uint32_t)px & (dim - 1)) << shift) + ((uint32_t)pz & (dim - 1))) << 5; // wraps around: every step is a fetch
		const u32x4 a = *(const GLOBAL u32x4 *)(base + off);
		const u32x4 b = *(const GLOBAL u32x4 *)(base + off + 16u);
		acc += a.x ^ a.w ^ b.y ^ b.z;
		if (tmx < tmz) { tmx += tdx; px += sx; } else { tmz += tdz; pz += sz; }
	}
	if (acc == 0x12345678u) { out[0] = acc; }
}

static double timed(hipEvent_t e0, hipEvent_t e1)
{
	float ms = 0;
	CK(hipEventSynchronize(e1));
	CK(hipEventElapsedTime(&ms, e0, e1));
	return ms;
}

int main()
{
	const size_t smallBytes = (size_t)2048 * 2048 * 32;   // 134 MB
	const size_t bigBytes = (size_t)4 << 30;               // 4 GiB
	uint8_t *small_, *big;
	uint32_t *out;
	CK(hipMalloc(&small_, smallBytes));
	CK(hipMalloc(&big, bigBytes));
	CK(hipMalloc(&out, 64));
	CK(hipMemset(small_, 0x5A, smallBytes));
	CK(hipMemset(big, 0x5A, bigBytes));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	printf("kernel,table_MB,algorithmic_bytes,ms,GB_per_s,records,Grecords_per_s\n");
	auto report = [&](const char *name, size_t tableBytes, double bytes, double records, double ms) {
		printf("%s,%.0f,%.0f,%.4f,%.1f,%.0f,%.2f\n", name, tableBytes / 1e6, bytes, ms, bytes / ms / 1e6, records, records / ms / 1e6);
		fflush(stdout);
	};
	// between two measured kernels: a 1 GiB stream over the other end of the big buffer pushes the previous table out of L2 / the Infinity Cache
	auto flush = [&]() { hipLaunchKernelGGL(cal_stream16, dim3(4096), dim3(256), 0, 0, (const u32x4 *)(big + bigBytes - ((size_t)1 << 30)), ((size_t)1 << 30) / 16, out); CK(hipDeviceSynchronize()); };

	{ // control: 2 GiB coalesced
		flush();
		const size_t n16 = ((size_t)2 << 30) / 16;
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL(cal_stream16, dim3(8192), dim3(256), 0, 0, (const u32x4 *)big, n16, out);
		CK(hipEventRecord(e1));
		report("cal_stream16", (size_t)2 << 30, (double)n16 * 16, (double)n16, timed(e0, e1));
	}
#define GATHER(BYTES, PER_LINE, TABLE, TBYTES, TAG)                                                                                        \
	{                                                                                                                                      \
		flush();                                                                                                                           \
		const uint32_t lines = (uint32_t)((TBYTES) / 128);                                                                                 \
		const uint64_t records = (uint64_t)lines * PER_LINE;                                                                               \
		const int rounds = 8;                                                                                                              \
		const uint32_t blocks = (uint32_t)(records / rounds / 256);                                                                        \
		CK(hipEventRecord(e0));                                                                                                            \
		hipLaunchKernelGGL((cal_gather<BYTES, PER_LINE>), dim3(blocks), dim3(256), 0, 0, TABLE, lines - 1u, rounds, 0x2545F491u, out);      \
		CK(hipEventRecord(e1));                                                                                                            \
		report("cal_gather<" #BYTES ";" #PER_LINE ">" TAG, TBYTES, (double)records *BYTES, (double)records, timed(e0, e1));                \
	}
	GATHER(32, 1, small_, smallBytes, "_134MB")
	GATHER(32, 4, small_, smallBytes, "_134MB")
	GATHER(64, 1, small_, smallBytes, "_134MB")
	GATHER(128, 1, small_, smallBytes, "_134MB")
	GATHER(4, 1, small_, smallBytes, "_134MB")
	GATHER(32, 1, big, bigBytes, "_4GiB")
	GATHER(32, 4, big, bigBytes, "_4GiB")
	GATHER(64, 1, big, bigBytes, "_4GiB")
	GATHER(128, 1, big, bigBytes, "_4GiB")
	GATHER(4, 1, big, bigBytes, "_4GiB")
	{ // the render kernel's locality: 16 waves per CU, 2000 steps each
		flush();
		const int waves = 256 * 16 * 4, steps = 2000;
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL(cal_walk, dim3(waves), dim3(64), 0, 0, small_, 11, steps, out);
		CK(hipEventRecord(e1));
		report("cal_walk_134MB", smallBytes, (double)waves * 64 * steps * 32, (double)waves * 64 * steps, timed(e0, e1));
	}
	return 0;
}
