// cvx_downsample.h -- World.DownSample (Assets/Code/World.cs:45-127) on the device (gfx950).
//
// The reference builds LOD j by pushing every voxel of the (2^j)^2 source columns under a target column into an
// RLEColumnBuilder with Y >> j (DownSamplePartial, :101-127), then sorting by Y, averaging the colours of voxels that
// share a Y and run-length encoding the result (RLEColumnBuilder.ToFinalColumn, WordBuilder.cs:181-268).  Byte work,
// no floating point.  Here: one wave per target column; the source voxels are accumulated into per-target-Y buckets
// in LDS (sum r, g, b, count with ds_add; the alpha of the FIRST voxel in the reference's insertion order with a
// 64-bit ds_min on {source column, sequence, alpha}), then the buckets are scanned top-down 64 at a time: a ballot
// gives the occupancy word, runs are cut out of it with ctz, colour slots are popcount prefixes.  Two passes over the
// same code: pass 1 counts runs / colours per column, an exclusive scan assigns element offsets in column order (the
// deterministic order of the host build, cvx_world.cpp BuildColumns), pass 2 writes headers, guards, runs, colours
// in the reference's storage layout (World.cs:161-169, 190-209) so the blob is byte-identical to the host's.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cvxk {

#define CVX_DS_BUCKETS 256 /* target-Y buckets held in LDS per pass over the sources (6 KB); taller occupied spans take several passes */

// One solid voxel of a level >= 1 as the NEXT level needs it (round 5: the LOD chain reads LOD 0 once).  World.DownSample always starts from LOD 0
// (UnityManager.cs:328-331), and a colour of level j is (sum of the LOD-0 voxels under the bucket) / (their number) per channel in integers -- which
// cannot be derived from the rounded colours of level j - 1, but CAN from that level's sums: sums and counts of the eight children add up.  The alpha of
// a bucket is that of the voxel the reference inserts FIRST (RLEColumnBuilder keeps the first of equal Ys, WordBuilder.cs:199-214, in the insertion
// order of World.cs:85-94,101-127: source columns ix-major, runs top-down, voxels of a run bottom-up), i.e. of the lexicographically first LOD-0
// column (x, z) that has a voxel in the bucket, its topmost run reaching into the bucket, that run's lowest voxel inside the bucket.  Per bucket:
//   key   = x << 16 | z of that LOD-0 column (all candidates of a target column share their high bits, so the order of keys is the insertion order),
//   alpha = the alpha described above,
//   cont  = that run goes on below the bucket.
// For a parent bucket (children: upper and lower half, four columns): the winner is the smallest key among the present children; if the winner has a
// voxel in the upper half and cont(upper), its first voxel lies in the lower half (same run: alpha(lower), cont(lower)), else in the upper half
// (alpha(upper), no continuation); a winner present in the lower half only: alpha(lower), cont(lower).  (A column that has voxels in a half IS that
// half's winner whenever it is the parent's: a smaller column with a voxel there would have been the parent's.)
struct SumVoxel {
	uint32_t r, g, b, n; // channel sums and number of the LOD-0 voxels under the bucket
	uint32_t key;        // x << 16 | z of the LOD-0 column whose voxel the reference inserts first
	uint32_t af;         // its alpha | cont << 8
};

struct DownsampleParams {
	const uint32_t *srcHeaders;  // 12-byte RLEColumn headers as 3 words each
	const uint32_t *srcElements; // element / colour pool
	const SumVoxel *srcSums;     // SUMS_IN: sums of the source level, indexed like srcElements (only the slots of colours are used)
	int srcLod, extraLods;
	int dimY;
	int srcMulX;        // dimZ >> srcLod
	int targetColumnsZ; // dimZ >> (srcLod + extraLods)
	int targetColumns;
	int chunkBuckets;   // buckets per pass over the sources: min(dimY >> target lod, CVX_DS_BUCKETS)
};

struct DownsampleOut {
	uint32_t *alloc;     // pass 1 out / pass 2 in (after the scan: element offset of the column)
	uint32_t *runCounts; // pass 1 out / pass 2 in
	uint32_t *headers;   // pass 2: 3 words per target column (table zero-initialised by the host)
	uint32_t *elements;  // pass 2
	SumVoxel *sums;      // pass 2, SUMS_OUT: sums of the level being written, indexed like `elements`
	unsigned long long *voxelCount;
	int *error; // 1: a column needs more than 65535 runs (World.cs:193-195)
};

template <bool WRITE>
__global__ __launch_bounds__(256) void downsample_kernel(DownsampleParams P, DownsampleOut O)
{
	// dynamic LDS: `chunk` = min(target height, CVX_DS_BUCKETS) buckets of 24 bytes (the launch passes chunk * 24)
	extern __shared__ unsigned long long dsLds[];
	const int chunk = P.chunkBuckets;
	unsigned long long *first = dsLds;
	uint32_t *sumR = reinterpret_cast<uint32_t *>(dsLds + chunk), *sumG = sumR + chunk, *sumB = sumG + chunk, *count = sumB + chunk;

	// 64 threads per target column for low LODs, 256 for high ones (1024 source columns per target column at LOD 5): all of them
	// gather the sources, the first wave alone writes the column
	const int thread = threadIdx.x, threads = blockDim.x;
	const int lane = thread & 63;
	const bool emitter = thread < 64;
	__shared__ int spanShared[8];
	const int k = blockIdx.x;
	const int targetLod = P.srcLod + P.extraLods;
	const int step = 1 << targetLod, stepSize = 1 << P.srcLod, steps = 1 << P.extraLods;
	const int xStart = (k / P.targetColumnsZ) * step, zStart = (k % P.targetColumnsZ) * step;
	const int srcHeight = P.dimY >> P.srcLod;
	const int topY = (P.dimY >> targetLod) - 1;

	// wave-uniform emit state
	int runLen = 0, runColorsIndex = 0, runs = 0, solid = 0;
	bool runSolid = false;
	int highest = -1, lowest = 0;
	uint32_t elemBase = 0, colourBase = 0;
	if (WRITE) {
		if (O.runCounts[k] == 0u) {
			return; // empty column: the header stays zero
		}
		elemBase = O.alloc[k];
		colourBase = elemBase + O.runCounts[k] + 2u;
	}
	auto flushRun = [&]() {
		if (runLen > 0) {
			if (WRITE && lane == 0) {
				O.elements[elemBase + 1u + (uint32_t)runs] = (uint32_t)(runSolid ? (runColorsIndex & 0xFFFF) : 0xFFFF) | ((uint32_t)runLen << 16);
			}
			runs++;
			runLen = 0;
		}
	};

	// Only the buckets between the lowest and the highest voxel of the source columns can be occupied: everything above is
	// one air run, everything below another.  (Found by walking the runs, not taken from the headers' WorldMin / WorldMax, so
	// the result does not depend on those being consistent.)
	int spanLo = 0x7FFFFFFF, spanHi = -1;
	for (int s = thread; s < steps * steps; s += threads) {
		const int x = xStart + (s / steps) * stepSize, z = zStart + (s % steps) * stepSize;
		const uint32_t *h = P.srcHeaders + 3 * (size_t)((x >> P.srcLod) * P.srcMulX + (z >> P.srcLod));
		const int runCount = (int)(h[1] & 0xFFFFu);
		const uint32_t *guardStart = P.srcElements + h[0];
		int elementBoundsX = srcHeight;
		for (int run = 0; run < runCount; run++) {
			const uint32_t raw = guardStart[run + 1];
			const int length = (int)(int16_t)(raw >> 16);
			elementBoundsX -= length;
			if ((int16_t)(raw & 0xFFFFu) >= 0) {
				spanHi = max(spanHi, (elementBoundsX + length - 1) >> P.extraLods);
				spanLo = min(spanLo, elementBoundsX >> P.extraLods);
			}
		}
	}
	for (int o = 32; o > 0; o >>= 1) {
		spanLo = min(spanLo, __shfl_xor(spanLo, o));
		spanHi = max(spanHi, __shfl_xor(spanHi, o));
	}
	if (threads > 64) { // combine the waves of the block
		if (lane == 0) {
			spanShared[2 * (thread >> 6)] = spanLo;
			spanShared[2 * (thread >> 6) + 1] = spanHi;
		}
		__syncthreads();
		for (int w = 0; w < (threads >> 6); w++) {
			spanLo = min(spanLo, spanShared[2 * w]);
			spanHi = max(spanHi, spanShared[2 * w + 1]);
		}
	}
	if (spanHi < 0) { // every source column is empty
		if (!WRITE && thread == 0) {
			O.alloc[k] = 0u;
			O.runCounts[k] = 0u;
		}
		return;
	}
	if (spanHi > topY) { spanHi = topY; }
	if (spanHi < topY) { // the air above the highest voxel
		runSolid = false;
		runLen = topY - spanHi;
	}

	for (int chunkTop = spanHi; chunkTop >= spanLo; chunkTop -= chunk) {
		const int chunkLo = chunkTop - chunk + 1 > spanLo ? chunkTop - chunk + 1 : spanLo;
		const int buckets = chunkTop - chunkLo + 1;
		for (int b = thread; b < buckets; b += threads) {
			sumR[b] = sumG[b] = sumB[b] = count[b] = 0u;
			first[b] = ~0ull;
		}
		__syncthreads();

		// DownSamplePartial for source column s = ix * steps + iz (the reference's insertion order, World.cs:85-94)
		for (int s = thread; s < steps * steps; s += threads) {
			const int x = xStart + (s / steps) * stepSize, z = zStart + (s % steps) * stepSize;
			const uint32_t *h = P.srcHeaders + 3 * (size_t)((x >> P.srcLod) * P.srcMulX + (z >> P.srcLod));
			const uint32_t offset = h[0];
			const int runCount = (int)(h[1] & 0xFFFFu);
			if (runCount == 0) {
				continue;
			}
			const uint32_t *guardStart = P.srcElements + offset;
			const uint32_t *colours = guardStart + runCount + 2;
			int elementBoundsX = srcHeight;
			uint32_t seq = 0;
			for (int run = 0; run < runCount; run++) {
				const uint32_t raw = guardStart[run + 1];
				const int length = (int)(int16_t)(raw >> 16);
				const int colorsIndex = (int)(int16_t)(raw & 0xFFFFu);
				elementBoundsX -= length;
				if (colorsIndex < 0) {
					continue;
				}
				// voxels i of the run with chunkLo <= (elementBoundsX + i) >> extraLods <= chunkTop
				int iFrom = (chunkLo << P.extraLods) - elementBoundsX;
				int iTo = ((chunkTop + 1) << P.extraLods) - elementBoundsX; // exclusive
				if (iFrom < 0) { iFrom = 0; }
				if (iTo > length) { iTo = length; }
				for (int i = iFrom; i < iTo; i++) {
					const int b = ((elementBoundsX + i) >> P.extraLods) - chunkLo;
					const uint32_t c = colours[colorsIndex + length - i - 1]; // bytes a, r, g, b
					atomicAdd(&sumR[b], (c >> 8) & 0xFFu);
					atomicAdd(&sumG[b], (c >> 16) & 0xFFu);
					atomicAdd(&sumB[b], c >> 24);
					atomicAdd(&count[b], 1u);
					atomicMin(&first[b], ((unsigned long long)s << 40) | ((unsigned long long)(seq + (uint32_t)i) << 8) | (c & 0xFFu));
				}
				seq += (uint32_t)length;
			}
		}
		__syncthreads();

		// ToFinalColumn over this chunk, top-down, 64 buckets per step (lane 0 = highest Y), first wave only
		for (int g = chunkTop; emitter && g >= chunkLo; g -= 64) {
			const int y = g - lane;
			const bool valid = y >= chunkLo;
			const uint32_t n = valid ? count[y - chunkLo] : 0u;
			const unsigned long long occupied = __ballot(n > 0u);
			const int nValid = g - chunkLo + 1 < 64 ? g - chunkLo + 1 : 64;
			if (WRITE && n > 0u) {
				const int b = y - chunkLo;
				const uint32_t slot = (uint32_t)solid + (uint32_t)__popcll(occupied & ((1ull << lane) - 1ull));
				O.elements[colourBase + slot] = (uint32_t)(first[b] & 0xFFull) | ((sumR[b] / n) << 8) | ((sumG[b] / n) << 16) | ((sumB[b] / n) << 24);
			}
			if (occupied != 0ull) {
				if (highest < 0) {
					highest = g - (__ffsll((long long)occupied) - 1);
				}
				lowest = g - (63 - __clzll((long long)occupied));
			}
			int pos = 0;
			while (pos < nValid) { // wave-uniform: cut the occupancy word into maximal runs
				const bool bit = ((occupied >> pos) & 1ull) != 0ull;
				const unsigned long long differing = (bit ? ~occupied : occupied) >> pos;
				int len = differing == 0ull ? 64 - pos : __ffsll((long long)differing) - 1;
				if (len > nValid - pos) { len = nValid - pos; }
				if (runLen > 0 && runSolid != bit) {
					flushRun();
				}
				if (runLen == 0) {
					runSolid = bit;
					runColorsIndex = solid + __popcll(occupied & ((1ull << pos) - 1ull));
				}
				runLen += len;
				pos += len;
			}
			solid += __popcll(occupied);
		}
		__syncthreads();
	}
	if (!emitter) {
		return; // the run / colour state lives in the first wave
	}
	if (spanLo > 0) { // the air below the lowest voxel
		if (runLen > 0 && runSolid) {
			flushRun();
		}
		runSolid = false;
		runLen += spanLo;
	}
	flushRun();

	if (solid == 0) {
		if (!WRITE && lane == 0) {
			O.alloc[k] = 0u;
			O.runCounts[k] = 0u;
		}
		return;
	}
	if (lane == 0) {
		if (!WRITE) {
			if (runs > 65535) {
				*O.error = 1;
			}
			O.alloc[k] = (uint32_t)(runs + solid + 2);
			O.runCounts[k] = (uint32_t)runs;
			atomicAdd(O.voxelCount, (unsigned long long)solid);
		} else {
			const uint32_t voxelScale = 1u << targetLod;
			const uint32_t worldMin = ((uint32_t)lowest * voxelScale) & 0xFFFFu;        // (ushort) casts of World.cs:231-232
			const uint32_t worldMax = ((uint32_t)(highest + 1) * voxelScale) & 0xFFFFu;
			O.headers[3 * (size_t)k + 0] = elemBase;
			O.headers[3 * (size_t)k + 1] = (uint32_t)runs | (worldMin << 16);
			O.headers[3 * (size_t)k + 2] = worldMax;
			O.elements[elemBase] = 0u;                      // element guards, World.cs:205-209
			O.elements[elemBase + 1u + (uint32_t)runs] = 0u;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// Low levels (2 x 2 or 4 x 4 source columns per target column): ONE THREAD per target column.  A workgroup per target
// column (above) leaves 60 of 64 lanes idle at LOD 1, and LOD 1 has a million target columns at 2048^2.  Here a thread
// walks its source columns top-down with one cursor each, bucket by bucket of the target (Y >> extraLods): sums and count of
// the voxels that fall into the bucket, the alpha of the first one in the reference's insertion order (source column s =
// ix * steps + iz ascending, runs top-down, voxels of a run bottom-up: World.cs:85-94,101-127), then the run-length
// encoding of RLEColumnBuilder.ToFinalColumn (WordBuilder.cs:181-268) as a scalar state machine.  Same two passes around the
// same scan, byte-identical output (tests: device blob == host blob).
// ---------------------------------------------------------------------------------------------------------------------
// SUMS_IN (E == 1 only): the source is a level >= 1 with its SumVoxel table instead of LOD 0 with its colours; SUMS_OUT: the write pass also stores the
// SumVoxel of every bucket for the next level (cvx_world_build_lods: LOD 0 is read once, every further level reads the one before it).
template <bool WRITE, int E, bool SUMS_IN = false, bool SUMS_OUT = false>
__global__ __launch_bounds__(64) void downsample_thread_kernel(DownsampleParams P, DownsampleOut O)
{
	static_assert(!SUMS_IN || E == 1, "a level is built from the sums of the level right below it");
	static_assert(!SUMS_OUT || E == 1, "sums are emitted by the 2 x 2 kernel only");
	constexpr int steps = 1 << E, S = steps * steps;
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= P.targetColumns) {
		return;
	}
	const int targetLod = P.srcLod + E;
	const int step = 1 << targetLod, stepSize = 1 << P.srcLod;
	const int xStart = (k / P.targetColumnsZ) * step, zStart = (k % P.targetColumnsZ) * step;
	const int srcHeight = P.dimY >> P.srcLod;
	const int topY = (P.dimY >> targetLod) - 1;

	// one cursor per source column: the run it stands on (solid runs only; air is skipped when advancing)
	uint32_t nextRun[S];   // index (into srcElements) of the next run word to read
	int runsLeft[S];       // run words not yet read
	int bound[S];          // elementBoundsX: bottom of the last run read = top (exclusive) of what is left
	int curLo[S], curLen[S]; // current solid run: source Ys [curLo, curLo + curLen - 1]; curLen == 0: none left
	uint32_t curColour[S];   // index (into srcElements) of the column's colours (behind its end guard)
	int colourIndexOfRun[S]; // ColorsIndex of the current run (its colours are stored top -> bottom)
	int spanLo = 0x7FFFFFFF, spanHi = -1;
#pragma unroll
	for (int s = 0; s < S; s++) {
		const int x = xStart + (s / steps) * stepSize, z = zStart + (s % steps) * stepSize;
		const uint32_t *h = P.srcHeaders + 3 * (size_t)((x >> P.srcLod) * P.srcMulX + (z >> P.srcLod));
		const uint32_t offset = h[0];
		const int runCount = (int)(h[1] & 0xFFFFu);
		// span of the occupied buckets (as the workgroup kernel: from the runs, not from the headers' WorldMin / WorldMax)
		int b = srcHeight;
		for (int run = 0; run < runCount; run++) {
			const uint32_t raw = P.srcElements[offset + 1u + (uint32_t)run];
			const int length = (int)(int16_t)(raw >> 16);
			b -= length;
			if ((int16_t)(raw & 0xFFFFu) >= 0) {
				spanHi = max(spanHi, (b + length - 1) >> E);
				spanLo = min(spanLo, b >> E);
			}
		}
		nextRun[s] = offset + 1u;
		runsLeft[s] = runCount;
		bound[s] = srcHeight;
		curLen[s] = 0;
		curLo[s] = 0;
		curColour[s] = offset + (uint32_t)runCount + 2u; // colours of the column start behind the end guard
	}
	if (spanHi < 0) { // every source column is empty
		if (!WRITE) {
			O.alloc[k] = 0u;
			O.runCounts[k] = 0u;
		}
		return;
	}
	if (spanHi > topY) { spanHi = topY; }

	uint32_t elemBase = 0, colourBase = 0;
	if (WRITE) {
		if (O.runCounts[k] == 0u) {
			return; // empty column: the header stays zero
		}
		elemBase = O.alloc[k];
		colourBase = elemBase + O.runCounts[k] + 2u;
	}
	int runLen = 0, runColorsIndex = 0, runs = 0, solid = 0;
	bool runSolid = false;
	int highest = -1, lowest = 0;
	auto flushRun = [&]() {
		if (runLen > 0) {
			if (WRITE) {
				O.elements[elemBase + 1u + (uint32_t)runs] = (uint32_t)(runSolid ? (runColorsIndex & 0xFFFF) : 0xFFFF) | ((uint32_t)runLen << 16);
			}
			runs++;
			runLen = 0;
		}
	};
	if (spanHi < topY) { // the air above the highest voxel
		runSolid = false;
		runLen = topY - spanHi;
	}
	const uint32_t *colourPool = P.srcElements;
	// every cursor on its first solid run
	auto fetchRun = [&](int s) { // next solid run of source s (air only moves the bound); curLen[s] stays 0 when there is none
		while (curLen[s] == 0 && runsLeft[s] > 0) {
			const uint32_t raw = P.srcElements[nextRun[s]];
			nextRun[s]++;
			runsLeft[s]--;
			const int length = (int)(int16_t)(raw >> 16);
			const int colorsIndex = (int)(int16_t)(raw & 0xFFFFu);
			bound[s] -= length;
			if (colorsIndex >= 0) {
				curLo[s] = bound[s];
				curLen[s] = length;
				colourIndexOfRun[s] = colorsIndex;
			}
		}
	};
#pragma unroll
	for (int s = 0; s < S; s++) { fetchRun(s); }
	int yPrev = spanHi + 1; // the bucket above the next one to emit
	for (int guard = spanHi - spanLo + 2; guard > 0; guard--) { // (every pass emits a lower bucket; the bound only makes termination independent of the data)
		// the highest bucket that still holds a voxel: buckets between it and the previous one are air (a column with a floating layer has
		// hundreds of them -- they are not visited one by one)
		int y = -1;
#pragma unroll
		for (int s = 0; s < S; s++) {
			if (curLen[s] > 0) { y = max(y, (curLo[s] + curLen[s] - 1) >> E); }
		}
		if (y < 0) {
			break;
		}
		if (y > topY) { y = topY; } // (cannot happen for a valid source: its height is srcHeight)
		const int gap = yPrev - 1 - y;
		if (gap > 0) { // `gap` air buckets, as the per-bucket rule below would treat them
			if (runLen > 0 && runSolid) { flushRun(); }
			if (runLen == 0) { runSolid = false; runColorsIndex = solid; }
			runLen += gap;
		}
		yPrev = y;
		const int bLo = y << E, bHi = bLo + steps - 1; // source Ys of this bucket
		uint32_t sumR = 0u, sumG = 0u, sumB = 0u, n = 0u, firstAlpha = 0u;
		uint32_t firstKey = 0xFFFFFFFFu, firstCont = 0u; // SUMS_IN / SUMS_OUT: see SumVoxel
#pragma unroll
		for (int s = 0; s < S; s++) {
			// SUMS_IN: what this source column holds in the upper (Y = 2 y + 1) and the lower (Y = 2 y) half of the bucket
			bool hasU = false;
			uint32_t keyU = 0xFFFFFFFFu, keyL = 0xFFFFFFFFu, afU = 0u, afL = 0u;
			while (curLen[s] > 0) {
				const int lo = curLo[s], hi = lo + curLen[s] - 1;
				if (hi < bLo) {
					break; // the run lies below this bucket: nothing more from this source here
				}
				// voxels of the run inside the bucket: Ys [max(lo, bLo), min(hi, bHi)], visited bottom-up like the reference
				const int yFrom = lo > bLo ? lo : bLo, yTo = hi < bHi ? hi : bHi;
				for (int Y = yFrom; Y <= yTo; Y++) {
					const int i = Y - lo;
					const uint32_t at = curColour[s] + (uint32_t)(colourIndexOfRun[s] + curLen[s] - i - 1);
					if (SUMS_IN) {
						if (WRITE) { // (the counting pass only needs the occupancy)
							const SumVoxel v = P.srcSums[at];
							sumR += v.r;
							sumG += v.g;
							sumB += v.b;
							n += v.n;
							if (Y & 1) { hasU = true; keyU = v.key; afU = v.af; } else { keyL = v.key; afL = v.af; }
						} else {
							n++;
						}
					} else {
						const uint32_t c = colourPool[at]; // bytes a, r, g, b
						if (n == 0u) {
							firstAlpha = c & 0xFFu;
							if (SUMS_OUT) { // this source column is the first in insertion order with a voxel here; `lo < bLo`: its run goes on below the bucket
								const int x = xStart + (s / steps) * stepSize, z = zStart + (s % steps) * stepSize;
								firstKey = ((uint32_t)x << 16) | (uint32_t)z;
								firstCont = lo < bLo ? 1u : 0u;
							}
						}
						sumR += (c >> 8) & 0xFFu;
						sumG += (c >> 16) & 0xFFu;
						sumB += c >> 24;
						n++;
					}
				}
				if (lo >= bLo) {
					curLen[s] = 0; // consumed: the next run of this source may reach into the same bucket
					fetchRun(s);
					continue;
				}
				// the run goes on below this bucket: keep its lower part [lo, bLo - 1] (its colours are stored top -> bottom, so the
				// colour of the part's top voxel is further down the run's colour list by the number of voxels consumed)
				colourIndexOfRun[s] += curLen[s] - (bLo - lo);
				curLen[s] = bLo - lo;
				break;
			}
			if (SUMS_IN && WRITE) { // the parent rule of SumVoxel, applied to this source's winner if it beats the sources before it
				const uint32_t keyS = keyU < keyL ? keyU : keyL; // (0xFFFFFFFF = absent)
				if (keyS < firstKey) {
					firstKey = keyS;
					const bool inU = hasU && keyU == keyS;
					if (inU && ((afU >> 8) & 1u)) { // its topmost run crosses into the lower half: the first voxel is the lower half's
						firstAlpha = afL & 0xFFu;
						firstCont = (afL >> 8) & 1u;
					} else if (inU) {
						firstAlpha = afU & 0xFFu;
						firstCont = 0u;
					} else {
						firstAlpha = afL & 0xFFu;
						firstCont = (afL >> 8) & 1u;
					}
				}
			}
		}
		// (n > 0: the bucket was chosen because a run reaches into it)
		if (highest < 0) { highest = y; }
		lowest = y;
		if (runLen > 0 && !runSolid) {
			flushRun();
		}
		if (runLen == 0) {
			runSolid = true;
			runColorsIndex = solid;
		}
		runLen++;
		if (WRITE) {
			O.elements[colourBase + (uint32_t)solid] = firstAlpha | ((sumR / n) << 8) | ((sumG / n) << 16) | ((sumB / n) << 24);
			if (SUMS_OUT) {
				O.sums[colourBase + (uint32_t)solid] = SumVoxel{ sumR, sumG, sumB, n, firstKey, firstAlpha | (firstCont << 8) };
			}
		}
		solid++;
	}
	if (spanLo > 0) { // the air below the lowest voxel
		if (runLen > 0 && runSolid) {
			flushRun();
		}
		runSolid = false;
		runLen += spanLo;
	}
	flushRun();

	if (solid == 0) {
		if (!WRITE) {
			O.alloc[k] = 0u;
			O.runCounts[k] = 0u;
		}
		return;
	}
	if (!WRITE) {
		if (runs > 65535) {
			*O.error = 1;
		}
		O.alloc[k] = (uint32_t)(runs + solid + 2);
		O.runCounts[k] = (uint32_t)runs;
		atomicAdd(O.voxelCount, (unsigned long long)solid);
	} else {
		const uint32_t voxelScale = 1u << targetLod;
		const uint32_t worldMin = ((uint32_t)lowest * voxelScale) & 0xFFFFu;        // (ushort) casts of World.cs:231-232
		const uint32_t worldMax = ((uint32_t)(highest + 1) * voxelScale) & 0xFFFFu;
		O.headers[3 * (size_t)k + 0] = elemBase;
		O.headers[3 * (size_t)k + 1] = (uint32_t)runs | (worldMin << 16);
		O.headers[3 * (size_t)k + 2] = worldMax;
		O.elements[elemBase] = 0u;                      // element guards, World.cs:205-209
		O.elements[elemBase + 1u + (uint32_t)runs] = 0u;
	}
}

// Exclusive prefix sum of n counts in place; *total receives the sum.  Three launches: every workgroup sums its chunk of CVX_SCAN_CHUNK values
// (coalesced), one workgroup scans the chunk sums, every workgroup scans its chunk again on top of its offset.  (Rounds 1-3: ONE workgroup whose
// threads each walked n / 1024 values with a stride between neighbours: 1.8 ms for the million columns of LOD 1 at 2048^2, a third of that level.)
#define CVX_SCAN_THREADS 256
#define CVX_SCAN_PER_THREAD 16
#define CVX_SCAN_CHUNK (CVX_SCAN_THREADS * CVX_SCAN_PER_THREAD)

__device__ __forceinline__ unsigned long long scan_block_inclusive(unsigned long long v, unsigned long long *partial /* CVX_SCAN_THREADS / 64 + 1 */)
{
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	for (int o = 1; o < 64; o <<= 1) { // inclusive scan inside the wave
		const unsigned long long up = (unsigned long long)__shfl_up((long long)v, o);
		if (lane >= o) { v += up; }
	}
	if (lane == 63) { partial[wave] = v; }
	__syncthreads();
	unsigned long long before = 0;
	for (int w = 0; w < wave; w++) { before += partial[w]; }
	__syncthreads();
	return v + before;
}

__global__ __launch_bounds__(CVX_SCAN_THREADS) void scan_chunk_sums_kernel(const uint32_t *__restrict__ values, int n, unsigned long long *__restrict__ chunkSums)
{
	__shared__ unsigned long long partial[CVX_SCAN_THREADS / 64 + 1];
	const long long base = (long long)blockIdx.x * CVX_SCAN_CHUNK;
	unsigned long long sum = 0;
	for (int i = 0; i < CVX_SCAN_PER_THREAD; i++) { // thread t takes elements t, t + 256, ...: coalesced
		const long long at = base + (long long)i * CVX_SCAN_THREADS + threadIdx.x;
		if (at < n) { sum += values[at]; }
	}
	const unsigned long long inclusive = scan_block_inclusive(sum, partial);
	if (threadIdx.x == CVX_SCAN_THREADS - 1) { chunkSums[blockIdx.x] = inclusive; }
}

// one workgroup: exclusive scan of the chunk sums in place (chunks <= a few thousand), *total = their sum
__global__ __launch_bounds__(CVX_SCAN_THREADS) void scan_chunk_offsets_kernel(unsigned long long *chunkSums, int chunks, unsigned long long *total)
{
	__shared__ unsigned long long partial[CVX_SCAN_THREADS / 64 + 1];
	__shared__ unsigned long long carry;
	if (threadIdx.x == 0) { carry = 0; }
	__syncthreads();
	for (int base = 0; base < chunks; base += CVX_SCAN_THREADS) {
		const int at = base + (int)threadIdx.x;
		const unsigned long long v = at < chunks ? chunkSums[at] : 0ull;
		const unsigned long long inclusive = scan_block_inclusive(v, partial);
		const unsigned long long offset = carry;
		if (at < chunks) { chunkSums[at] = offset + inclusive - v; }
		__syncthreads();
		if (threadIdx.x == CVX_SCAN_THREADS - 1) { carry = offset + inclusive; }
		__syncthreads();
	}
	if (threadIdx.x == 0) { *total = carry; }
}

__global__ __launch_bounds__(CVX_SCAN_THREADS) void scan_apply_kernel(uint32_t *values, int n, const unsigned long long *__restrict__ chunkOffsets)
{
	__shared__ unsigned long long partial[CVX_SCAN_THREADS / 64 + 1];
	// thread t owns the CONSECUTIVE elements [t * 16, t * 16 + 16) of the chunk (64 bytes: four dwordx4) so that its running sum is a local one
	const long long first = (long long)blockIdx.x * CVX_SCAN_CHUNK + (long long)threadIdx.x * CVX_SCAN_PER_THREAD;
	uint32_t v[CVX_SCAN_PER_THREAD];
	unsigned long long sum = 0;
#pragma unroll
	for (int i = 0; i < CVX_SCAN_PER_THREAD; i++) {
		v[i] = first + i < n ? values[first + i] : 0u;
		sum += v[i];
	}
	const unsigned long long inclusive = scan_block_inclusive(sum, partial);
	unsigned long long running = chunkOffsets[blockIdx.x] + inclusive - sum;
#pragma unroll
	for (int i = 0; i < CVX_SCAN_PER_THREAD; i++) {
		if (first + i < n) { values[first + i] = (uint32_t)running; } // callers check *total < 2^31 before using the offsets
		running += v[i];
	}
}

} // namespace cvxk
