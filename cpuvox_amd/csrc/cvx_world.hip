// cvx_world.hip -- libcpuvox_gpu.so, the world side of the C ABI: validation of the reference's storage blobs,
// cvx_world_upload (blob -> device column records) and World.DownSample on the device (cvx_world_downsample,
// cvx_world_build_lods; kernels in cvx_downsample.h).  See include/cpuvox_gpu.h for the contract of every entry point.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cvx_context.h"
#include "cvx_downsample.h"
#include "cpuvox_gpu_diag.h"

using cvxi::Fail;
using cvxi::IsPow2;
using cvxi::ValidateColumn;

namespace cvxi {

// One column of a world blob in the reference's layout (World.cs:161-209): element range inside the pool, both
// guards present, positive run lengths that fit the column height, colours inside the pool.  Everything the kernels
// dereference later is covered here.  *solidRuns receives the number of solid runs, *colourCount (optional) the number of colours the runs address.
int ValidateColumn(cvx_context *ctx, int64_t i, const RefHeader &h, const uint32_t *elements, int64_t elementCount, int maxY, size_t *solidRuns, int64_t *colourCount)
{
	const int64_t off = h.storageOffset;
	if (off < 0 || off + h.runCount + 2 > elementCount) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "column %lld: element range outside the pool", (long long)i);
	}
	if (elements[off] != 0u || elements[off + h.runCount + 1] != 0u) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "column %lld: missing element guards (World.cs:205-209)", (long long)i);
	}
	int64_t colours = 0, total = 0;
	size_t solid = 0;
	for (int r = 0; r < h.runCount; r++) {
		const uint32_t raw = elements[off + 1 + r];
		const int colorsIndex = (int)(int16_t)(raw & 0xFFFFu);
		const int length = (int)(int16_t)(raw >> 16);
		if (length <= 0) {
			return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "column %lld: run %d has length %d", (long long)i, r, length);
		}
		total += length;
		if (colorsIndex >= 0) {
			solid++;
			if (colorsIndex + length > colours) { colours = colorsIndex + length; }
		}
	}
	if (total != maxY || off + h.runCount + 2 + colours > elementCount) {
		// (total == height: RLEColumnBuilder.ToFinalColumn always emits full-height columns, WordBuilder.cs:232-258; for a shorter
		// one the reference's top-down and bottom-up walks would place the same run at different heights)
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "column %lld: runs do not add up to the column height or colours exceed the pool", (long long)i);
	}
	*solidRuns = solid;
	if (colourCount) { *colourCount = colours; }
	return CVX_OK;
}

} // namespace cvxi

int cvx_world_upload(cvx_context *ctx, int lod, const void *storage, int64_t byteLength, int dimX, int dimY, int dimZ, int columnCount)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!storage || lod < 0 || lod >= CVX_LOD_LEVELS) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad storage / lod"); }
	if (!IsPow2(dimX) || !IsPow2(dimY) || !IsPow2(dimZ) || dimY > 65536) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "world dimensions must be powers of two (WordBuilder.cs:30), Y <= 65536");
	}
	if (dimX > 32768 || dimZ > 32768) { // (the column loop keeps a ray's column as x * 65536 + z in one register; 8192 x 8192 columns already fill the 4 GiB the offsets address)
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "world dimensions X, Z must be <= 32768");
	}
	if (lod > 0 && !ctx->levelSet[0]) {
		return Fail(ctx, CVX_ERR_NOT_READY, "upload LOD 0 first: the other levels are checked against its dimensions (World.cs:47)");
	}
	if (lod > 0 && (ctx->hostWorld.dimX != dimX || ctx->hostWorld.dimY != dimY || ctx->hostWorld.dimZ != dimZ)) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "all LODs share the LOD-0 dimensions (World.cs:47)");
	}
	const int64_t usedX = dimX >> lod, usedZ = dimZ >> lod;
	const int64_t usedColumns = usedX * usedZ;
	if (columnCount < usedColumns || (int64_t)columnCount * 12 > byteLength) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "columnCount %d inconsistent with dims/lod/byteLength", columnCount);
	}
	const int64_t elementCount = (byteLength - (int64_t)columnCount * 12) / 4;
	const RefHeader *src = static_cast<const RefHeader *>(storage);
	const uint32_t *elements = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(storage) + (size_t)columnCount * 12);

	// Validate every column (so that nothing the kernel dereferences can leave the pool) and build the table of 16-byte
	// column records (row-major), the run list and the counting build's table (cvx_device.h).  The data goes to the device with the next draw.
	const int maxY = dimY >> lod;
	int rowShift = 0;
	while (((int64_t)1 << rowShift) < usedZ) { rowShift++; }
	const size_t recordCount = (size_t)usedX << rowShift;
	if (elementCount >= ((int64_t)1 << 30)) { // (the colours are a subset of the pool)
		return Fail(ctx, CVX_ERR_CAPACITY, "LOD %d: an element pool of %lld entries (the records address 2^30 colours)", lod, (long long)elementCount);
	}
	// pass 1: validation, and the size of the run list (a block per column whose record cannot hold its runs: an even number of entries)
	struct Shape {
		uint32_t solid;                  // solid runs
		uint32_t bottom[3], top[3];      // spans of the first three, LOD-0 voxels
		uint32_t position[3];            // 1-based position among all elements, top-down
		bool derived;                    // every ColorsIndex is the sum of the lengths of the solid runs above it (what the reference's builder emits)
	};
	const auto shapeOf = [&](const RefHeader &h) {
		Shape sh{};
		sh.derived = true;
		uint32_t start = 0, colours = 0;
		for (int r = 0; r < h.runCount; r++) { // top-down
			const uint32_t raw = elements[h.storageOffset + 1 + r];
			const uint32_t length = raw >> 16;
			if ((int16_t)(raw & 0xFFFFu) >= 0) {
				// the run's span in LOD-0 voxels: [bottomY, topY] with topY = dimY - (start << lod) (<= 65536)
				const uint32_t topY = (uint32_t)dimY - (start << lod), bottomY = topY - (length << lod);
				if (sh.solid < 3) { sh.bottom[sh.solid] = bottomY; sh.top[sh.solid] = topY; sh.position[sh.solid] = (uint32_t)(r + 1); }
				if ((raw & 0xFFFFu) != colours) { sh.derived = false; }
				colours += length;
				sh.solid++;
			}
			start += length;
		}
		return sh;
	};
	// 1 .. 3: the record holds the column's runs (cvx_device.h); 0: they live in the run list
	const auto codeOf = [](const RefHeader &h, const Shape &sh) -> uint32_t {
		if (sh.solid >= 1 && sh.solid <= 3 && sh.derived && h.worldMax == sh.top[0] && h.worldMin == sh.bottom[sh.solid - 1]) { return sh.solid; }
		return 0u;
	};
	size_t listEntries = 0;
	std::vector<uint32_t> colourCounts((size_t)usedColumns, 0u);
	int64_t colourTotal = 0; // colours of all columns: the device keeps them densely packed (the reference's pool interleaves them with the RLE elements, which the kernel never reads)
	for (int64_t i = 0; i < usedColumns; i++) {
		const RefHeader &h = src[i];
		if (h.runCount == 0) {
			continue;
		}
		size_t solid = 0;
		int64_t colours = 0;
		const int rc = ValidateColumn(ctx, i, h, elements, elementCount, maxY, &solid, &colours);
		if (rc != CVX_OK) {
			return rc;
		}
		colourCounts[(size_t)i] = (uint32_t)colours;
		colourTotal += colours;
		if (codeOf(h, shapeOf(h)) == 0u) {
			listEntries += (solid + 1) & ~(size_t)1;
		}
	}
	cvx_context::HostLevel &H = ctx->hostLevel[lod];
	H.records.assign(recordCount, uint4{ 0u, 0u, 0u, 0u });
	H.counts.assign(recordCount, uint2{ 0u, 0u });
	H.runs.assign(listEntries + 4, uint2{ 0u, 0u }); // never empty; the kernel may read two entries at any block
	// Colours: blocks of CVX_COLOR_BLOCK_X x CVX_COLOR_BLOCK_Z columns, colour k of the block's 32 columns in ONE 128-byte line (cvx_device.h).
	// A block is as deep as its column with the most colours: ~2.5 x the colours themselves for a terrain.  A world of a few deep columns among
	// empty ones would pay up to 32 x: beyond 4 x the level keeps its colours column after column instead (stride 1).
	const int64_t blocksX = (usedX + CVX_COLOR_BLOCK_X - 1) / CVX_COLOR_BLOCK_X, blocksZ = (usedZ + CVX_COLOR_BLOCK_Z - 1) / CVX_COLOR_BLOCK_Z;
	std::vector<uint32_t> blockBase((size_t)(blocksX * blocksZ), 0u);
	size_t colourEntries = CVX_COLOR_STRIDE; // (one line of zeros in front: no column's base is 0, cvx_device.h)
	for (int64_t bx = 0; bx < blocksX; bx++) {
		for (int64_t bz = 0; bz < blocksZ; bz++) {
			uint32_t depth = 0;
			for (int64_t cx = bx * CVX_COLOR_BLOCK_X; cx < std::min<int64_t>(usedX, (bx + 1) * CVX_COLOR_BLOCK_X); cx++) {
				for (int64_t cz = bz * CVX_COLOR_BLOCK_Z; cz < std::min<int64_t>(usedZ, (bz + 1) * CVX_COLOR_BLOCK_Z); cz++) {
					depth = std::max(depth, colourCounts[(size_t)(cx * usedZ + cz)]);
				}
			}
			blockBase[(size_t)(bx * blocksZ + bz)] = (uint32_t)colourEntries;
			colourEntries += (size_t)depth * CVX_COLOR_STRIDE;
		}
	}
	const bool blocked = colourEntries <= 4 * (size_t)colourTotal + 65536 && colourEntries <= ((size_t)1 << 29); // (... and at most half of what the 32-bit offsets address)
	const size_t colourStride = blocked ? CVX_COLOR_STRIDE : 1;
	if (!blocked) { colourEntries = CVX_COLOR_STRIDE + (size_t)colourTotal; }
	if (colourEntries + CVX_COLOR_STRIDE >= ((size_t)1 << 30)) {
		return Fail(ctx, CVX_ERR_CAPACITY, "LOD %d: %.2f G colour slots (the records address 2^30)", lod, (double)colourEntries / 1e9);
	}
	H.elements.assign(colourEntries + CVX_COLOR_STRIDE, 0u);
	H.colorShift = blocked ? 7 : 2; // log2 of the bytes between two colours of a column
	size_t denseCursor = CVX_COLOR_STRIDE;
	size_t listCursor = 0;
	H.solidColumns = H.listedColumns = 0;
	for (int64_t cx = 0; cx < usedX; cx++) {
		for (int64_t cz = 0; cz < usedZ; cz++) {
			const int64_t i = cx * usedZ + cz; // World.GetIndexKnownInBounds, World.cs:145-149
			const RefHeader &h = src[i];
			if (h.runCount == 0) {
				continue;
			}
			const int64_t off = h.storageOffset;
			const int n = h.runCount;
			const size_t at = ((size_t)cx << rowShift) + (size_t)cz;
			const Shape sh = shapeOf(h);
			const uint32_t code = codeOf(h, sh);
			// RLEColumn.ColorPointer (World.cs:185) as the slot of the column's first colour (>= 32: a listed column's x is never 0, the empty column's always)
			const uint32_t colorsBase = blocked ? blockBase[(size_t)((cx / CVX_COLOR_BLOCK_X) * blocksZ + cz / CVX_COLOR_BLOCK_Z)] + (uint32_t)((cx % CVX_COLOR_BLOCK_X) * CVX_COLOR_BLOCK_Z + cz % CVX_COLOR_BLOCK_Z)
			                                    : (uint32_t)denseCursor;
			for (uint32_t k = 0; k < colourCounts[(size_t)i]; k++) {
				H.elements[(size_t)colorsBase + (size_t)k * colourStride] = elements[off + n + 2 + k];
			}
			denseCursor += colourCounts[(size_t)i];
			const uint32_t bounds = (uint32_t)h.worldMin | ((uint32_t)h.worldMax << 16);
			uint32_t z = 0, w = 0;
			H.solidColumns++;
			if (code == 0u) {
				H.listedColumns++;
				const size_t block = listCursor;
				uint32_t start = 0;
				for (int r = 0; r < n; r++) { // top-down
					const uint32_t raw = elements[off + 1 + r];
					const uint32_t length = raw >> 16;
					if ((int16_t)(raw & 0xFFFFu) >= 0) {
						const uint32_t topY = (uint32_t)dimY - (start << lod), bottomY = topY - (length << lod);
						H.runs[listCursor++] = uint2{ bottomY | ((topY - 1u) << 16), (raw & 0xFFFFu) | ((uint32_t)(r + 1) << 16) };
					}
					start += length;
				}
				listCursor = (listCursor + 1) & ~(size_t)1;
				z = (uint32_t)block;
				w = sh.solid;
			} else {
				// run 0 = [w.lo, worldMax], run 1 = [z.lo, w.hi + 1], run 2 = [worldMin, z.hi + 1]; the last run's foot is worldMin
				w = (code >= 2u) ? (sh.bottom[0] | ((sh.top[1] - 1u) << 16)) : bounds;
				z = (code == 3u) ? (sh.bottom[1] | ((sh.top[2] - 1u) << 16)) : (uint32_t)h.worldMin;
			}
			H.records[at] = uint4{ (code << 30) | colorsBase, bounds, z, w };
			H.counts[at] = uint2{ (uint32_t)n | (sh.position[0] << 16), sh.position[1] | (sh.position[2] << 16) };
		}
	}
	H.recordsBytes = H.records.size() * sizeof(uint4);
	H.runsBytes = H.runs.size() * sizeof(uint2);
	H.countsBytes = H.counts.size() * sizeof(uint2);
	H.elementsBytes = H.elements.size() * sizeof(uint32_t);
	H.rowShift = rowShift;
	H.pending = true;
	if (lod == 0) {
		if (ctx->hostWorld.dimX != dimX || ctx->hostWorld.dimY != dimY || ctx->hostWorld.dimZ != dimZ) {
			// a LOD 0 of other dimensions starts a new world: the tables of the old LOD 1..5 are indexed with the old
			// dimensions and must not survive (the draw reports CVX_ERR_NOT_READY until all levels are uploaded again)
			for (int l = 1; l < CVX_LOD_LEVELS; l++) {
				ctx->hostLevel[l] = cvx_context::HostLevel();
				ctx->levelSet[l] = false;
			}
		}
		ctx->hostWorld.dimX = dimX;
		ctx->hostWorld.dimY = dimY;
		ctx->hostWorld.dimZ = dimZ;
		ctx->hostWorld.maskX = dimX - 1;
		ctx->hostWorld.maskZ = dimZ - 1;
	}
	ctx->levelSet[lod] = true;
	ctx->worldDirty = true;
	return CVX_OK;
}

namespace {

// Validates a world blob (every column, like cvx_world_upload) and copies it to the device.
// *elementsOfColumns (optional) receives what the columns hold when each is counted by itself -- RunCount runs + two guards + the colours its solid runs
// address, summed over the columns: a number the blob's pool size says nothing about when several headers share one storageOffset.
int UploadSourceBlob(cvx_context *ctx, const void *storage, int64_t byteLength, int dimX, int dimY, int dimZ, int lod, int columnCount, uint8_t **dSrc,
                     int64_t *elementsOfColumns = nullptr)
{
	*dSrc = nullptr;
	if (elementsOfColumns) { *elementsOfColumns = 0; }
	if (!storage) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	if (!IsPow2(dimX) || !IsPow2(dimY) || !IsPow2(dimZ) || dimY > 65536) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "world dimensions must be powers of two (WordBuilder.cs:30), Y <= 65536");
	}
	if (lod < 0 || lod > 15 || (dimX >> lod) < 1 || (dimY >> lod) < 1 || (dimZ >> lod) < 1) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "lod %d out of range for these dimensions", lod);
	}
	const int64_t usedColumns = (int64_t)(dimX >> lod) * (dimZ >> lod);
	if (columnCount < usedColumns || (int64_t)columnCount * 12 > byteLength) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "columnCount %d inconsistent with dims/lod/byteLength", columnCount);
	}
	const int64_t elementCount = (byteLength - (int64_t)columnCount * 12) / 4;
	const RefHeader *src = static_cast<const RefHeader *>(storage);
	const uint32_t *elements = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(storage) + (size_t)columnCount * 12);
	for (int64_t i = 0; i < usedColumns; i++) {
		if (src[i].runCount == 0) { continue; }
		size_t solid = 0;
		int64_t colours = 0;
		const int rc = ValidateColumn(ctx, i, src[i], elements, elementCount, dimY >> lod, &solid, &colours);
		if (rc != CVX_OK) { return rc; }
		if (elementsOfColumns) { *elementsOfColumns += (int64_t)src[i].runCount + 2 + colours; }
	}
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	CVX_HIP(ctx, hipMalloc((void **)dSrc, (size_t)byteLength));
	hipError_t e = hipMemcpyAsync(*dSrc, storage, (size_t)byteLength, hipMemcpyHostToDevice, ctx->stream);
	if (e != hipSuccess) {
		(void)hipFree(*dSrc);
		*dSrc = nullptr;
		return Fail(ctx, CVX_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
	}
	return CVX_OK;
}

// World.DownSample(extraLods) of the validated blob at dSrc (device); see cvx_world_downsample.
// Exclusive prefix sum of n counts in place, *total = their sum: chunk sums, their offsets, the chunks again (cvx_downsample.h).
// chunkSums: (n + CVX_SCAN_CHUNK - 1) / CVX_SCAN_CHUNK words of 8 bytes.
void ExclusiveScan(hipStream_t stream, uint32_t *values, int n, unsigned long long *chunkSums, unsigned long long *total)
{
	const unsigned chunks = (unsigned)(((long long)n + CVX_SCAN_CHUNK - 1) / CVX_SCAN_CHUNK);
	hipLaunchKernelGGL(cvxk::scan_chunk_sums_kernel, dim3(chunks), dim3(CVX_SCAN_THREADS), 0, stream, values, n, chunkSums);
	hipLaunchKernelGGL(cvxk::scan_chunk_offsets_kernel, dim3(1), dim3(CVX_SCAN_THREADS), 0, stream, chunkSums, (int)chunks, total);
	hipLaunchKernelGGL(cvxk::scan_apply_kernel, dim3(chunks), dim3(CVX_SCAN_THREADS), 0, stream, values, n, chunkSums);
}

int DownsampleDevice(cvx_context *ctx, const uint8_t *dSrc, int dimX, int dimY, int dimZ, int lod, int columnCount, int extraLods,
                     void **outStorage, int64_t *outByteLength, int32_t *outColumnCount, int64_t *outVoxelCount, float *outDeviceMs)
{
	*outStorage = nullptr;
	const int targetLod = lod + extraLods;
	if (extraLods < 1 || extraLods > 8 || targetLod > 15 || (dimX >> targetLod) < 1 || (dimY >> targetLod) < 1 || (dimZ >> targetLod) < 1) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "lod %d + extraLods %d out of range for these dimensions", lod, extraLods);
	}
	const int64_t targetColumns = (int64_t)(dimX >> targetLod) * (dimZ >> targetLod);
	const int64_t allocatedColumns = ((int64_t)dimX * dimZ) / ((int64_t)(targetLod + 1) * (targetLod + 1)); // World.ColumnCount, World.cs:17
	if (targetColumns > 0x7FFFFFFF || allocatedColumns > 0x7FFFFFFF || allocatedColumns < targetColumns) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "target LOD has an unsupported column count");
	}

	uint32_t *dAlloc = nullptr, *dHeaders = nullptr, *dElements = nullptr;
	unsigned long long *dScalars = nullptr; // [0] voxel count, [1] element total, [2] error flag
	hipEvent_t evBegin = nullptr, evEnd = nullptr;
	void *host = nullptr;
	int rc = CVX_OK;
	auto release = [&]() {
		if (dAlloc) { (void)hipFree(dAlloc); }
		if (dHeaders) { (void)hipFree(dHeaders); }
		if (dElements) { (void)hipFree(dElements); }
		if (dScalars) { (void)hipFree(dScalars); }
		if (evBegin) { (void)hipEventDestroy(evBegin); }
		if (evEnd) { (void)hipEventDestroy(evEnd); }
	};
#define CVX_DS(call)                                                                                                      \
	do {                                                                                                                  \
		hipError_t e_ = (call);                                                                                           \
		if (e_ != hipSuccess) {                                                                                           \
			rc = Fail(ctx, CVX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);       \
			release();                                                                                                    \
			std::free(host);                                                                                              \
			return rc;                                                                                                    \
		}                                                                                                                 \
	} while (0)
	const size_t headerWords = (size_t)allocatedColumns * 3;
	// (+ the chunk sums of the offset scan behind the two tables, 8-byte aligned)
	const size_t scanChunks = (size_t)((targetColumns + CVX_SCAN_CHUNK - 1) / CVX_SCAN_CHUNK);
	const size_t tableWords = ((size_t)targetColumns * 2 + 1) & ~(size_t)1;
	CVX_DS(hipMalloc((void **)&dAlloc, tableWords * sizeof(uint32_t) + (scanChunks + 1) * sizeof(unsigned long long)));
	unsigned long long *dChunkSums = reinterpret_cast<unsigned long long *>(dAlloc + tableWords);
	CVX_DS(hipMalloc((void **)&dHeaders, headerWords * sizeof(uint32_t)));
	CVX_DS(hipMalloc((void **)&dScalars, 3 * sizeof(unsigned long long)));
	CVX_DS(hipEventCreate(&evBegin));
	CVX_DS(hipEventCreate(&evEnd));
	CVX_DS(hipMemsetAsync(dHeaders, 0, headerWords * sizeof(uint32_t), ctx->stream));
	CVX_DS(hipMemsetAsync(dScalars, 0, 3 * sizeof(unsigned long long), ctx->stream));

	cvxk::DownsampleParams P{};
	P.srcHeaders = reinterpret_cast<const uint32_t *>(dSrc);
	P.srcElements = reinterpret_cast<const uint32_t *>(dSrc + (size_t)columnCount * 12);
	P.srcLod = lod;
	P.extraLods = extraLods;
	P.dimY = dimY;
	P.srcMulX = dimZ >> lod;
	P.targetColumnsZ = dimZ >> targetLod;
	P.targetColumns = (int)targetColumns;
	P.chunkBuckets = std::min(dimY >> targetLod, CVX_DS_BUCKETS); // 6 KB of LDS: the wave count per CU, not LDS, limits residency
	const size_t dsLdsBytes = (size_t)P.chunkBuckets * 24;
	const unsigned dsThreads = extraLods >= 3 ? 256u : 64u; // (2^extraLods)^2 source columns per target column
	cvxk::DownsampleOut O{};
	O.alloc = dAlloc;
	O.runCounts = dAlloc + targetColumns;
	O.headers = dHeaders;
	O.voxelCount = dScalars;
	O.error = reinterpret_cast<int *>(dScalars + 2);

	CVX_DS(hipEventRecord(evBegin, ctx->stream));
	// 2 x 2 / 4 x 4 source columns per target column (LOD 1 / 2: a million / a quarter of a million target columns at 2048^2): one THREAD per target
	// column (cvx_downsample.h); above that a workgroup per target column.  LOD 1: 17.9 -> 5.4 ms, LOD 2: 5.4 -> 3.5 ms.
	const dim3 threadGrid((unsigned)((targetColumns + 63) / 64)), threadBlock(64);
	if (extraLods == 1) {
		hipLaunchKernelGGL((cvxk::downsample_thread_kernel<false, 1>), threadGrid, threadBlock, 0, ctx->stream, P, O);
	} else if (extraLods == 2) {
		hipLaunchKernelGGL((cvxk::downsample_thread_kernel<false, 2>), threadGrid, threadBlock, 0, ctx->stream, P, O);
	} else {
		hipLaunchKernelGGL((cvxk::downsample_kernel<false>), dim3((unsigned)targetColumns), dim3(dsThreads), dsLdsBytes, ctx->stream, P, O);
	}
	ExclusiveScan(ctx->stream, dAlloc, (int)targetColumns, dChunkSums, dScalars + 1); // element counts -> element offsets
	CVX_DS(hipGetLastError());
	unsigned long long scalars[3] = { 0, 0, 0 };
	CVX_DS(hipMemcpyAsync(scalars, dScalars, sizeof scalars, hipMemcpyDeviceToHost, ctx->stream));
	CVX_DS(hipStreamSynchronize(ctx->stream));
	if ((int)scalars[2] != 0) {
		release();
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "a downsampled column needs more than 65535 runs (World.cs:193-195)");
	}
	if (scalars[1] > 0x7FFFFFFFull) {
		release();
		return Fail(ctx, CVX_ERR_CAPACITY, "Only supports up to 2^31 elements (World.cs:355-357)");
	}
	const size_t elementTotal = (size_t)scalars[1];
	CVX_DS(hipMalloc((void **)&dElements, (elementTotal > 0 ? elementTotal : 1) * sizeof(uint32_t)));
	O.elements = dElements;
	if (extraLods == 1) {
		hipLaunchKernelGGL((cvxk::downsample_thread_kernel<true, 1>), threadGrid, threadBlock, 0, ctx->stream, P, O);
	} else if (extraLods == 2) {
		hipLaunchKernelGGL((cvxk::downsample_thread_kernel<true, 2>), threadGrid, threadBlock, 0, ctx->stream, P, O);
	} else {
		hipLaunchKernelGGL((cvxk::downsample_kernel<true>), dim3((unsigned)targetColumns), dim3(dsThreads), dsLdsBytes, ctx->stream, P, O);
	}
	CVX_DS(hipGetLastError());
	CVX_DS(hipEventRecord(evEnd, ctx->stream));
	const size_t outBytes = headerWords * 4 + elementTotal * 4;
	host = std::malloc(outBytes > 0 ? outBytes : 1);
	if (!host) {
		release();
		return Fail(ctx, CVX_ERR_HIP, "out of host memory");
	}
	CVX_DS(hipMemcpyAsync(host, dHeaders, headerWords * 4, hipMemcpyDeviceToHost, ctx->stream));
	if (elementTotal > 0) {
		CVX_DS(hipMemcpyAsync(static_cast<uint8_t *>(host) + headerWords * 4, dElements, elementTotal * 4, hipMemcpyDeviceToHost, ctx->stream));
	}
	CVX_DS(hipStreamSynchronize(ctx->stream));
	float ms = 0.0f;
	CVX_DS(hipEventElapsedTime(&ms, evBegin, evEnd));
#undef CVX_DS
	release();
	*outStorage = host;
	*outByteLength = (int64_t)outBytes;
	*outColumnCount = (int32_t)allocatedColumns;
	if (outVoxelCount) { *outVoxelCount = (int64_t)scalars[0]; }
	if (outDeviceMs) { *outDeviceMs = ms; }
	return CVX_OK;
}

// The whole LOD chain in one go, LOD 0 read ONCE (round 5): level 1 from LOD 0's colours, every further level from the level
// before it through its table of per-voxel sums (cvxk::SumVoxel: channel sums, count, and the key / alpha / continuation that reproduce the
// reference's "first inserted voxel keeps its alpha").  Same kernels, same two passes around the same scan per level as DownsampleDevice, but no host
// round trip between them: the element pools are sized by a bound instead of by the scanned total -- a target column needs at most the elements of its
// four sources + 2 (its solid runs are unions of theirs, one more air run, two guards; halving Y never adds runs), so level j has at most
// elements(LOD 0) + 2 (columns(1) + ... + columns(j)) < elements(LOD 0) + columns(LOD 0) of them, elements(LOD 0) counted COLUMN BY COLUMN (a blob whose
// headers share pool regions holds more elements than its pool) -- and every error is looked at once, at the end.
// Device memory: per level the bound x 4 bytes, + two tables of per-voxel sums of the bound x 24 bytes: ~17 x the LOD 0 blob for five levels, ~19 x for seven.
// Levels above `kMaxChainLevel` (channel sums of 2^24 voxels x 255 no longer fit 32 bits) are left to DownsampleDevice.
constexpr int kMaxChainLevel = 7;

int BuildLodChainDevice(cvx_context *ctx, const uint8_t *dSrc, int64_t elementsOfColumns, int dimX, int dimY, int dimZ, int columnCount, int levelCount,
                        void **outStorage, int64_t *outByteLength, int32_t *outColumnCount, float *outDeviceMs)
{
	struct Level {
		int64_t targetColumns = 0, allocatedColumns = 0;
		uint32_t *dTables = nullptr, *dHeaders = nullptr, *dElements = nullptr;
		unsigned long long *dChunkSums = nullptr;
	};
	std::vector<Level> L((size_t)levelCount + 1);
	cvxk::SumVoxel *dSums[2] = { nullptr, nullptr };
	unsigned long long *dScalars = nullptr; // per level: [0] voxel count, [1] element total, [2] error flag
	hipEvent_t evBegin = nullptr, evEnd = nullptr;
	std::vector<void *> host((size_t)levelCount, nullptr);
	auto release = [&]() {
		for (Level &l : L) {
			if (l.dTables) { (void)hipFree(l.dTables); }
			if (l.dHeaders) { (void)hipFree(l.dHeaders); }
			if (l.dElements) { (void)hipFree(l.dElements); }
		}
		for (cvxk::SumVoxel *p : dSums) { if (p) { (void)hipFree(p); } }
		if (dScalars) { (void)hipFree(dScalars); }
		if (evBegin) { (void)hipEventDestroy(evBegin); }
		if (evEnd) { (void)hipEventDestroy(evEnd); }
	};
	auto fail = [&](int rc) {
		release();
		for (void *p : host) { std::free(p); }
		return rc;
	};
	// (running out of device memory is CVX_ERR_CAPACITY: the caller then builds level by level, which needs a fraction of it; any other HIP error is an error)
#define CVX_CH(call)                                                                                                                          \
	do {                                                                                                                                      \
		hipError_t e_ = (call);                                                                                                               \
		if (e_ != hipSuccess) {                                                                                                               \
			if (e_ == hipErrorOutOfMemory) { (void)hipGetLastError(); }                                                                       \
			return fail(Fail(ctx, e_ == hipErrorOutOfMemory ? CVX_ERR_CAPACITY : CVX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__)); \
		}                                                                                                                                     \
	} while (0)
	const int64_t usedColumns0 = (int64_t)dimX * dimZ;
	// The bound counts every source column by itself (UploadSourceBlob: runs + two guards + colours, summed over the columns), NOT the blob's pool: headers
	// may share one storageOffset and runs of a column may share colours (the reference's loader accepts both, and so does cvx_world_upload), and then a
	// small pool stands for many elements -- every one of which the next level may need a place for.
	const int64_t elementBound = elementsOfColumns + usedColumns0; // (see above; >= 1)
	if (elementBound > 0x7FFFFFFFll) { return Fail(ctx, CVX_ERR_CAPACITY, "the chain's pools would pass 2^31 elements (World.cs:355-357 per level): level by level instead"); }
	for (int j = 1; j <= levelCount; j++) {
		Level &l = L[(size_t)j];
		if ((dimX >> j) < 1 || (dimY >> j) < 1 || (dimZ >> j) < 1) { return fail(Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "LOD %d out of range for these dimensions", j)); }
		l.targetColumns = (int64_t)(dimX >> j) * (dimZ >> j);
		l.allocatedColumns = ((int64_t)dimX * dimZ) / ((int64_t)(j + 1) * (j + 1)); // World.ColumnCount, World.cs:17
		if (l.allocatedColumns > 0x7FFFFFFF || l.allocatedColumns < l.targetColumns) { return fail(Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "target LOD has an unsupported column count")); }
		const size_t scanChunks = (size_t)((l.targetColumns + CVX_SCAN_CHUNK - 1) / CVX_SCAN_CHUNK);
		const size_t tableWords = ((size_t)l.targetColumns * 2 + 1) & ~(size_t)1;
		CVX_CH(hipMalloc((void **)&l.dTables, tableWords * sizeof(uint32_t) + (scanChunks + 1) * sizeof(unsigned long long)));
		l.dChunkSums = reinterpret_cast<unsigned long long *>(l.dTables + tableWords);
		CVX_CH(hipMalloc((void **)&l.dHeaders, (size_t)l.allocatedColumns * 12));
		CVX_CH(hipMalloc((void **)&l.dElements, (size_t)elementBound * sizeof(uint32_t)));
		CVX_CH(hipMemsetAsync(l.dHeaders, 0, (size_t)l.allocatedColumns * 12, ctx->stream));
	}
	for (int i = 0; i < 2 && i < levelCount - 1; i++) { CVX_CH(hipMalloc((void **)&dSums[i], (size_t)elementBound * sizeof(cvxk::SumVoxel))); }
	CVX_CH(hipMalloc((void **)&dScalars, (size_t)levelCount * 3 * sizeof(unsigned long long)));
	CVX_CH(hipMemsetAsync(dScalars, 0, (size_t)levelCount * 3 * sizeof(unsigned long long), ctx->stream));
	CVX_CH(hipEventCreate(&evBegin));
	CVX_CH(hipEventCreate(&evEnd));

	CVX_CH(hipEventRecord(evBegin, ctx->stream));
	for (int j = 1; j <= levelCount; j++) {
		Level &l = L[(size_t)j];
		const bool fromLod0 = j == 1, emitSums = j < levelCount;
		cvxk::DownsampleParams P{};
		P.srcHeaders = fromLod0 ? reinterpret_cast<const uint32_t *>(dSrc) : L[(size_t)j - 1].dHeaders;
		P.srcElements = fromLod0 ? reinterpret_cast<const uint32_t *>(dSrc + (size_t)columnCount * 12) : L[(size_t)j - 1].dElements;
		P.srcSums = fromLod0 ? nullptr : dSums[(j - 1) & 1];
		P.srcLod = j - 1;
		P.extraLods = 1;
		P.dimY = dimY;
		P.srcMulX = dimZ >> (j - 1);
		P.targetColumnsZ = dimZ >> j;
		P.targetColumns = (int)l.targetColumns;
		P.chunkBuckets = 0;
		cvxk::DownsampleOut O{};
		O.alloc = l.dTables;
		O.runCounts = l.dTables + l.targetColumns;
		O.headers = l.dHeaders;
		O.elements = l.dElements;
		O.sums = emitSums ? dSums[j & 1] : nullptr;
		unsigned long long *scalars = dScalars + (size_t)(j - 1) * 3;
		O.voxelCount = scalars;
		O.error = reinterpret_cast<int *>(scalars + 2);
		const dim3 grid((unsigned)((l.targetColumns + 63) / 64)), block(64);
		if (fromLod0) {
			hipLaunchKernelGGL((cvxk::downsample_thread_kernel<false, 1, false, false>), grid, block, 0, ctx->stream, P, O);
		} else {
			hipLaunchKernelGGL((cvxk::downsample_thread_kernel<false, 1, true, false>), grid, block, 0, ctx->stream, P, O);
		}
		ExclusiveScan(ctx->stream, l.dTables, (int)l.targetColumns, l.dChunkSums, scalars + 1); // element counts -> element offsets
		if (fromLod0) {
			if (emitSums) {
				hipLaunchKernelGGL((cvxk::downsample_thread_kernel<true, 1, false, true>), grid, block, 0, ctx->stream, P, O);
			} else {
				hipLaunchKernelGGL((cvxk::downsample_thread_kernel<true, 1, false, false>), grid, block, 0, ctx->stream, P, O);
			}
		} else if (emitSums) {
			hipLaunchKernelGGL((cvxk::downsample_thread_kernel<true, 1, true, true>), grid, block, 0, ctx->stream, P, O);
		} else {
			hipLaunchKernelGGL((cvxk::downsample_thread_kernel<true, 1, true, false>), grid, block, 0, ctx->stream, P, O);
		}
		CVX_CH(hipGetLastError());
	}
	CVX_CH(hipEventRecord(evEnd, ctx->stream));
	std::vector<unsigned long long> scalars((size_t)levelCount * 3, 0ull);
	CVX_CH(hipMemcpyAsync(scalars.data(), dScalars, scalars.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
	CVX_CH(hipStreamSynchronize(ctx->stream));
	for (int j = 1; j <= levelCount; j++) {
		if ((int)scalars[(size_t)(j - 1) * 3 + 2] != 0) { return fail(Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "a downsampled column needs more than 65535 runs (World.cs:193-195)")); }
		if (scalars[(size_t)(j - 1) * 3 + 1] > (unsigned long long)elementBound) { return fail(Fail(ctx, CVX_ERR_CAPACITY, "LOD %d: element total beyond its bound (internal error)", j)); }
	}
	for (int j = 1; j <= levelCount; j++) {
		Level &l = L[(size_t)j];
		const size_t headerBytes = (size_t)l.allocatedColumns * 12, elementTotal = (size_t)scalars[(size_t)(j - 1) * 3 + 1];
		const size_t outBytes = headerBytes + elementTotal * 4;
		host[(size_t)j - 1] = std::malloc(outBytes > 0 ? outBytes : 1);
		if (!host[(size_t)j - 1]) { return fail(Fail(ctx, CVX_ERR_HIP, "out of host memory")); }
		CVX_CH(hipMemcpyAsync(host[(size_t)j - 1], l.dHeaders, headerBytes, hipMemcpyDeviceToHost, ctx->stream));
		if (elementTotal > 0) { CVX_CH(hipMemcpyAsync(static_cast<uint8_t *>(host[(size_t)j - 1]) + headerBytes, l.dElements, elementTotal * 4, hipMemcpyDeviceToHost, ctx->stream)); }
		outByteLength[j - 1] = (int64_t)outBytes;
		outColumnCount[j - 1] = (int32_t)l.allocatedColumns;
	}
	CVX_CH(hipStreamSynchronize(ctx->stream));
	float ms = 0.0f;
	CVX_CH(hipEventElapsedTime(&ms, evBegin, evEnd));
#undef CVX_CH
	release();
	for (int j = 0; j < levelCount; j++) { outStorage[j] = host[(size_t)j]; }
	if (outDeviceMs) { *outDeviceMs = ms; }
	return CVX_OK;
}

} // namespace

/* World.DownSample(extraLods), World.cs:45-127, on the device (cvx_downsample.h). */
int cvx_world_downsample(cvx_context *ctx, const void *storage, int64_t byteLength, int dimX, int dimY, int dimZ, int lod, int columnCount, int extraLods,
                         void **outStorage, int64_t *outByteLength, int32_t *outColumnCount, int64_t *outVoxelCount, float *outDeviceMs)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!outStorage || !outByteLength || !outColumnCount) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	*outStorage = nullptr;
	if (lod != 0) {
		// The reference only ever downsamples LOD 0 (UnityManager.cs:328-331: worldLODs[i] = worldLODs[0].DownSample(i)); for a source
		// of lod > 0 its DownSamplePartial shifts source-LOD heights by lod + extraLods (World.cs:108-124), which squashes the column
		// and is not a result worth reproducing -- refuse instead of returning a blob that differs from the host build.
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "cvx_world_downsample takes the LOD 0 blob (lod = %d given): World.DownSample is only defined for LOD 0 sources", lod);
	}
	uint8_t *dSrc = nullptr;
	int rc = UploadSourceBlob(ctx, storage, byteLength, dimX, dimY, dimZ, lod, columnCount, &dSrc);
	if (rc != CVX_OK) { return rc; }
	rc = DownsampleDevice(ctx, dSrc, dimX, dimY, dimZ, lod, columnCount, extraLods, outStorage, outByteLength, outColumnCount, outVoxelCount, outDeviceMs);
	(void)hipFree(dSrc);
	return rc;
}

/* UnityManager.cs:328-331: worldLODs[i] = worldLODs[0].DownSample(i) for i = 1..levelCount, one validation + one upload of LOD 0. */
int cvx_world_build_lods(cvx_context *ctx, const void *storage, int64_t byteLength, int dimX, int dimY, int dimZ, int columnCount, int levelCount,
                         void **outStorage, int64_t *outByteLength, int32_t *outColumnCount, float *outDeviceMs)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!outStorage || !outByteLength || !outColumnCount || levelCount < 1 || levelCount > 15) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	for (int i = 0; i < levelCount; i++) { outStorage[i] = nullptr; }
	uint8_t *dSrc = nullptr;
	int64_t elementsOfColumns = 0;
	int rc = UploadSourceBlob(ctx, storage, byteLength, dimX, dimY, dimZ, 0, columnCount, &dSrc, &elementsOfColumns);
	if (rc != CVX_OK) { return rc; }
	// levels 1 .. min(levelCount, 7) as one chain that reads LOD 0 once; anything above from LOD 0 directly
	float totalMs = 0.0f;
	const int chained = std::min(levelCount, kMaxChainLevel);
	rc = BuildLodChainDevice(ctx, dSrc, elementsOfColumns, dimX, dimY, dimZ, columnCount, chained, outStorage, outByteLength, outColumnCount, &totalMs);
#ifdef CVX_EXPERIMENTS
	if (std::getenv("CVX_LOD_CHAIN_FAILS")) { // (diagnostics: the level-by-level fallback below, as if the chain's pools had not fitted)
		for (int i = 0; i < chained; i++) { std::free(outStorage[i]); outStorage[i] = nullptr; }
		rc = CVX_ERR_CAPACITY;
	}
#endif
	int first = chained;
	if (rc == CVX_ERR_CAPACITY) { // the chain's pools (~17 x the LOD 0 blob) did not fit the device memory or the 2^31 elements of a level: level by level from LOD 0 instead, like rounds 1-4
		(void)hipGetLastError();
		for (int i = 0; i < levelCount; i++) { outStorage[i] = nullptr; }
		totalMs = 0.0f;
		first = 0;
		rc = CVX_OK;
	}
	for (int i = first; i < levelCount && rc == CVX_OK; i++) {
		float ms = 0.0f;
		rc = DownsampleDevice(ctx, dSrc, dimX, dimY, dimZ, 0, columnCount, i + 1, &outStorage[i], &outByteLength[i], &outColumnCount[i], nullptr, &ms);
		totalMs += ms;
	}
	(void)hipFree(dSrc);
	if (rc != CVX_OK) {
		for (int i = 0; i < levelCount; i++) {
			std::free(outStorage[i]);
			outStorage[i] = nullptr;
		}
		return rc;
	}
	if (outDeviceMs) { *outDeviceMs = totalMs; }
	return CVX_OK;
}

#if defined(CVX_EXPERIMENTS) || defined(CVX_PROFILE_SECTIONS) /* include/cpuvox_gpu_diag.h: not in the product library */
int cvx_selftest_scan(cvx_context *ctx, int n, uint32_t *values, uint64_t *total)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (n <= 0 || !values || !total) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	const size_t chunks = ((size_t)n + CVX_SCAN_CHUNK - 1) / CVX_SCAN_CHUNK;
	uint8_t *d = nullptr;
	const size_t valueBytes = ((size_t)n * 4 + 15) & ~(size_t)15;
	CVX_HIP(ctx, hipMalloc((void **)&d, valueBytes + (chunks + 1) * 8));
	uint32_t *dValues = reinterpret_cast<uint32_t *>(d);
	unsigned long long *dChunkSums = reinterpret_cast<unsigned long long *>(d + valueBytes), *dTotal = dChunkSums + chunks;
	unsigned long long sum = 0;
	hipError_t e = hipMemcpyAsync(dValues, values, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) {
		ExclusiveScan(ctx->stream, dValues, n, dChunkSums, dTotal);
		e = hipGetLastError();
	}
	if (e == hipSuccess) { e = hipMemcpyAsync(values, dValues, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream); }
	if (e == hipSuccess) { e = hipMemcpyAsync(&sum, dTotal, 8, hipMemcpyDeviceToHost, ctx->stream); }
	if (e == hipSuccess) { e = hipStreamSynchronize(ctx->stream); }
	(void)hipFree(d);
	if (e != hipSuccess) { return Fail(ctx, CVX_ERR_HIP, "scan self-test failed: %s", hipGetErrorString(e)); }
	*total = sum;
	return CVX_OK;
}
#endif

void cvx_free(void *p)
{
	std::free(p);
}
