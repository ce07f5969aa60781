// cvx_gpu.hip -- libcpuvox_gpu.so: the C-ABI drop-in for
// RenderManager.DrawSegments (Assets/Code/RenderManager.cs:258-372) on MI355X.
// See include/cpuvox_gpu.h for the contract of every entry point.
#include <hip/hip_runtime.h>

#include <memory>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "cvx_context.h"
#include "cpuvox_gpu_diag.h"
#include "cvx_kernels.h"

namespace {
std::string g_createError; // cvx_last_error(NULL)
}

namespace cvxi {

int Fail(cvx_context *ctx, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (ctx) {
		ctx->error = buf;
	} else {
		g_createError = buf;
	}
	return code;
}

} // namespace cvxi

using cvxi::Fail;
using cvxi::IsPow2;

namespace {

void FreeRaybuffers(cvx_context *ctx)
{
	if (!ctx->poolsExternal) {
		if (ctx->poolBaseTD) { (void)hipFree(ctx->poolBaseTD); }
		if (ctx->poolBaseLR) { (void)hipFree(ctx->poolBaseLR); }
	}
	ctx->poolBaseTD = ctx->poolBaseLR = nullptr;
	ctx->poolsExternal = false;
	ctx->poolTD.clear();
	ctx->poolLR.clear();
	ctx->last.clear();
	if (ctx->screen) { (void)hipFree(ctx->screen); ctx->screen = nullptr; }
	if (ctx->screenBatch) { (void)hipFree(ctx->screenBatch); ctx->screenBatch = nullptr; }
	ctx->screenBatchFrames = 0;
	ctx->resX = ctx->resY = 0;
}

// Mathf.RoundToInt (half to even) then clamp, RenderManager.cs:302-311
int RoundClamp(float v, int lo, int hi)
{
	float r = std::nearbyint(v);
	int i = (r != r || r >= 2147483648.0f || r < -2147483648.0f) ? (int)0x80000000 : (int)r;
	return i < lo ? lo : (i > hi ? hi : i);
}

bool Finite(const float *v, int n)
{
	for (int i = 0; i < n; i++) {
		if (!std::isfinite(v[i])) { return false; }
	}
	return true;
}

// All buffers of one kind live back to back in ONE allocation (buffer b at b * poolBytes), so a consumer can view
// them as one [bufferCount * tiles, width * 64] array (RCCL exchange of tiles).
int EnsurePools(cvx_context *ctx)
{
	if (ctx->poolBaseTD) { return CVX_OK; }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	CVX_HIP(ctx, hipMalloc((void **)&ctx->poolBaseTD, ctx->poolBytesTD * (size_t)ctx->bufferCount));
	CVX_HIP(ctx, hipMalloc((void **)&ctx->poolBaseLR, ctx->poolBytesLR * (size_t)ctx->bufferCount));
	CVX_HIP(ctx, hipMemsetAsync(ctx->poolBaseTD, 0, ctx->poolBytesTD * (size_t)ctx->bufferCount, ctx->stream));
	CVX_HIP(ctx, hipMemsetAsync(ctx->poolBaseLR, 0, ctx->poolBytesLR * (size_t)ctx->bufferCount, ctx->stream));
	for (int i = 0; i < ctx->bufferCount; i++) {
		ctx->poolTD[(size_t)i] = ctx->poolBaseTD + (size_t)i * (ctx->poolBytesTD / 4);
		ctx->poolLR[(size_t)i] = ctx->poolBaseLR + (size_t)i * (ctx->poolBytesLR / 4);
	}
	return CVX_OK;
}

int EnsureScratch(cvx_context *ctx, size_t frames, size_t tiles)
{
	if (frames > ctx->devFramesCap) {
		if (ctx->devFrames) { (void)hipFree(ctx->devFrames); ctx->devFrames = nullptr; }
		size_t cap = frames + frames / 2 + 4;
		CVX_HIP(ctx, hipMalloc((void **)&ctx->devFrames, cap * sizeof(DevFrame)));
		ctx->devFramesCap = cap;
	}
	if (tiles > ctx->devTilesCap) {
		if (ctx->devTiles) { (void)hipFree(ctx->devTiles); ctx->devTiles = nullptr; }
		size_t cap = tiles + tiles / 2 + 64;
		CVX_HIP(ctx, hipMalloc((void **)&ctx->devTiles, cap * sizeof(DevTile)));
		ctx->devTilesCap = cap;
	}
	return CVX_OK;
}

#ifdef CVX_TILE_TIMES /* diagnostic build (tools/lpt_oracle.py): which tile each launched wave belongs to; after a blocking draw the clock ticks of
                         every tile (sum over its waves) are written to $CVX_TILE_TIMES_OUT as one float per tile, in BuildFrame's tile order */
std::vector<uint32_t> g_waveSource;
size_t g_sourceTiles = 0;
unsigned long long *g_tileTimesDev = nullptr;
size_t g_tileTimesCap = 0;
#endif

// Launch-order heuristic only (never affects results): DDA column visits a ray of the segment would make if nothing occluded it = path length
// inside the world's XZ box (capped by far clip) * (|dx| + |dz|); an upper bound of the ray's steps.
float RayColumnVisits(const cvx_context *ctx, const DevFrame &F, const DevSegment &S, int planeRayIndex)
{
	float t = (float)planeRayIndex / (float)(S.rayCount > 0 ? S.rayCount : 1);
	if (t > 1.f) { t = 1.f; }
	float dx = S.rayMinX + t * (S.rayMaxX - S.rayMinX);
	float dz = S.rayMinZ + t * (S.rayMaxZ - S.rayMinZ);
	float len = std::sqrt(dx * dx + dz * dz);
	if (!(len > 0.f)) { return 0.f; }
	dx /= len;
	dz /= len;
	float t0 = 0.f, t1 = F.farClip;
	const float lo[2] = { 0.f, 0.f }, hi[2] = { (float)ctx->hostWorld.dimX, (float)ctx->hostWorld.dimZ };
	const float o[2] = { F.posX, F.posZ }, d[2] = { dx, dz };
	for (int a = 0; a < 2; a++) {
		if (std::fabs(d[a]) < 1e-12f) {
			if (o[a] < lo[a] || o[a] > hi[a]) { return 0.f; }
			continue;
		}
		float ta = (lo[a] - o[a]) / d[a], tb = (hi[a] - o[a]) / d[a];
		if (ta > tb) { float tmp = ta; ta = tb; tb = tmp; }
		if (ta > t0) { t0 = ta; }
		if (tb < t1) { t1 = tb; }
	}
	if (t1 <= t0) { return 0.f; }
	// (weighting the far part by its LOD -- a step at LOD l crosses 2^l voxels -- was measured and orders the launch slightly worse)
	return (t1 - t0) * (std::fabs(dx) + std::fabs(dz));
}

// A wave lives as long as its longest ray, and the path through the world box changes fast from ray to ray near the box's corners and where rays
// start to miss it: the tile's estimate is the LONGER of its two edge rays (the middle ray ranks such tiles far too low, and a long tile that
// starts late is what a launch waits for at its end).  List scheduling of measured tile lives on the 4096 wave slots (profiles/r02_experiments.md):
// middle ray 1.090 x the ideal, longer edge ray 1.015, the measured lives themselves 1.010.  An optional term counts the pixels of the tile's
// window (CVX_TILE_COST_PIXELS; it helped the middle-ray estimate by 1 % and hurts this one).
float EstimateTileCost(const cvx_context *ctx, const DevFrame &F, const DevSegment &S, int tileInSeg)
{
	const int first = tileInSeg * CVX_WAVE;
	const int last = std::min(first + CVX_WAVE - 1, (S.rayCount > 0 ? S.rayCount : 1) - 1);
	const float visits = ctx->tileCostMiddleRay ? RayColumnVisits(ctx, F, S, first + CVX_WAVE / 2)
	                                            : std::max(RayColumnVisits(ctx, F, S, first), RayColumnVisits(ctx, F, S, last));
	return visits + ctx->tileCostPixelWeight * (float)(S.omax - S.omin + 1);
}

// Fills SegmentContext[4] the way DrawSegments does (RenderManager.cs:281-318)
// and appends this frame's tiles.
// placements: optional caller-chosen output address per tile (canonical order: frame-major, segment-major); 0 = this
// context does not render the tile.  placeCursor runs over the whole batch.
int BuildFrame(cvx_context *ctx, const cvx_segment_data segments[4], const cvx_camera_data *camera, int W, int H, const float vp[2],
               int bufferIndex, int frameIndex, DevFrame &F, std::vector<DevTile> &tiles, std::vector<float> &tileCost, std::vector<int> &tileWords, LastDraw &last,
               const uint64_t *placements, int64_t placementCount, int64_t &placeCursor)
{
	if (!Finite(camera->WorldToScreenMatrix, 16) || !Finite(camera->PositionXZ, 2) || !Finite(&camera->PositionY, 1) ||
	    !Finite(&camera->FarClip, 1) || !Finite(camera->LODDistances, CVX_LOD_LEVELS) || !Finite(vp, 2)) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "camera data / vanishing point contains a non-finite value");
	}
	std::memcpy(F.M, camera->WorldToScreenMatrix, sizeof F.M);
	F.posX = camera->PositionXZ[0];
	F.posZ = camera->PositionXZ[1];
	F.posY = camera->PositionY;
	F.farClip = camera->FarClip;
	for (int i = 0; i < CVX_LOD_LEVELS; i++) { F.lod[i] = camera->LODDistances[i]; }
	F.inverse = camera->InverseElementIterationDirection ? 1 : 0;
	F.pad_ = 0;
	F.poolTD = placements ? nullptr : ctx->poolTD[(size_t)bufferIndex];
	F.poolLR = placements ? nullptr : ctx->poolLR[(size_t)bufferIndex];

	last.valid = placements == nullptr; // read-back / blit only know the library's own layout
	last.width = W;
	last.height = H;
	last.vp[0] = vp[0];
	last.vp[1] = vp[1];
	std::memcpy(last.segments, segments, sizeof last.segments);

	int tileIdInFrame = 0;
	for (int s = 0; s < 4; s++) {
		DevSegment &S = F.seg[s];
		std::memset(&S, 0, sizeof S);
		const cvx_segment_data &seg = segments[s];
		int rayCount = seg.RayCount > 0 ? seg.RayCount : 0;
		S.rayCount = rayCount;
		S.axisMappedToY = s > 1 ? 0 : 1;
		S.colLen = s < 2 ? H : W;
		// tiles of segment 1 / 3 follow those of segment 0 / 2 in the pool
		S.tileBase = (s == 1 || s == 3) ? (F.seg[s - 1].tileBase + (F.seg[s - 1].rayCount + CVX_WAVE - 1) / CVX_WAVE) : 0;
		last.tileBase[s] = S.tileBase;
		if (rayCount <= 0) {
			continue;
		}
		if (!Finite(seg.CamLocalPlaneRayMin, 2) || !Finite(seg.CamLocalPlaneRayMax, 2)) {
			return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "segment %d has non-finite plane rays", s);
		}
		S.rayMinX = seg.CamLocalPlaneRayMin[0];
		S.rayMinZ = seg.CamLocalPlaneRayMin[1];
		S.rayMaxX = seg.CamLocalPlaneRayMax[0];
		S.rayMaxZ = seg.CamLocalPlaneRayMax[1];
		if (s == 0) { // top
			S.omin = RoundClamp(vp[1], 0, H - 1);
			S.omax = H - 1;
		} else if (s == 1) { // bottom
			S.omin = 0;
			S.omax = RoundClamp(vp[1], 0, H - 1);
		} else if (s == 3) { // left
			S.omin = 0;
			S.omax = RoundClamp(vp[0], 0, W - 1);
		} else { // right
			S.omin = RoundClamp(vp[0], 0, W - 1);
			S.omax = W - 1;
		}
		int segTiles = (rayCount + CVX_WAVE - 1) / CVX_WAVE;
		int capacity = s < 2 ? ctx->tilesTD : ctx->tilesLR;
		if (S.tileBase + segTiles > capacity) {
			return Fail(ctx, CVX_ERR_CAPACITY, "segment %d: %d rays exceed the raybuffer capacity (RenderManager.cs:35-36)", s, rayCount);
		}
		const int maskWords = (S.omax >> 5) - (S.omin >> 5) + 1;
		for (int t = 0; t < segTiles; t++, tileIdInFrame++) {
			uint32_t *out;
			if (placements) {
				if (placeCursor >= placementCount) {
					return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "tile placement list shorter than the batch's tile count");
				}
				out = reinterpret_cast<uint32_t *>((uintptr_t)placements[placeCursor++]);
				if (!out) { continue; }
			} else {
				if (tileIdInFrame % ctx->shardCount != ctx->shardIndex) {
					continue;
				}
				out = (s < 2 ? F.poolTD : F.poolLR) + ((size_t)(S.tileBase + t) * (size_t)S.colLen) * CVX_WAVE;
			}
			tiles.push_back(DevTile{ frameIndex, s, t, 0, out });
			tileCost.push_back(EstimateTileCost(ctx, F, S, t));
			tileWords.push_back(maskWords);
		}
	}
	// reference capacity check: TopDown holds W+2H rays, LeftRight 2W+H
	if ((int64_t)F.seg[0].rayCount + F.seg[1].rayCount > (int64_t)W + 2 * H || (int64_t)F.seg[2].rayCount + F.seg[3].rayCount > (int64_t)2 * W + H) {
		return Fail(ctx, CVX_ERR_CAPACITY, "ray counts exceed the raybuffer capacity (RenderManager.cs:35-36)");
	}
	return CVX_OK;
}

// Folds all pending event pairs into accumulatedMs (waits for them).
int FoldEvents(cvx_context *ctx)
{
	for (size_t i = 0; i < ctx->evUsed; i++) {
		float ms = 0.f;
		CVX_HIP(ctx, hipEventSynchronize(ctx->evPairs[2 * i + 1]));
		CVX_HIP(ctx, hipEventElapsedTime(&ms, ctx->evPairs[2 * i], ctx->evPairs[2 * i + 1]));
		ctx->accumulatedMs += ms;
		ctx->accumulatedDraws++;
		ctx->lastMs = ms;
	}
	ctx->evUsed = 0;
	return CVX_OK;
}

int NextEventPair(cvx_context *ctx, hipEvent_t &start, hipEvent_t &stop)
{
	if (ctx->evUsed >= 1024) {
		int rc = FoldEvents(ctx);
		if (rc != CVX_OK) { return rc; }
	}
	if (ctx->evPairs.size() < 2 * (ctx->evUsed + 1)) {
		hipEvent_t a = nullptr, b = nullptr;
		CVX_HIP(ctx, hipEventCreate(&a));
		CVX_HIP(ctx, hipEventCreate(&b));
		ctx->evPairs.push_back(a);
		ctx->evPairs.push_back(b);
	}
	start = ctx->evPairs[2 * ctx->evUsed];
	stop = ctx->evPairs[2 * ctx->evUsed + 1];
	ctx->evUsed++;
	return CVX_OK;
}

int SyncWorld(cvx_context *ctx)
{
	if (!ctx->levelSet[0]) {
		return Fail(ctx, CVX_ERR_NOT_READY, "world LOD 0 has not been uploaded");
	}
	if (ctx->worldDirty) {
		// Missing higher LODs alias the last uploaded one only if never reached; require all 6 like the reference's World[6].
		for (int i = 0; i < CVX_LOD_LEVELS; i++) {
			if (!ctx->levelSet[i]) {
				return Fail(ctx, CVX_ERR_NOT_READY, "world LOD %d has not been uploaded (UnityManager.LOD_LEVELS = 6)", i);
			}
		}
		// One arena for all levels: [guard | records | guard | run list | counts | element pool] per level, 256-byte aligned parts, 32-bit offsets.
		// guard = one row of records + 64 bytes that belong to nothing (cvx_device.h: the fetch of a ray that has just left the world).
		DevWorld next = ctx->hostWorld;
		size_t cursor = 0;
		auto place = [&](size_t bytes) { const size_t at = cursor; cursor = (cursor + bytes + 255) & ~(size_t)255; return at; };
		size_t recordsAt[CVX_LOD_LEVELS], runsAt[CVX_LOD_LEVELS], countsAt[CVX_LOD_LEVELS], elementsAt[CVX_LOD_LEVELS];
		for (int i = 0; i < CVX_LOD_LEVELS; i++) {
			const cvx_context::HostLevel &H = ctx->hostLevel[i];
			const size_t guard = (((size_t)16 << H.rowShift) + 64 + 255) & ~(size_t)255;
			recordsAt[i] = place(guard + H.recordsBytes + guard) + guard;
			runsAt[i] = place(H.runsBytes);
			countsAt[i] = place(H.countsBytes);
			elementsAt[i] = place(H.elementsBytes);
		}
		if (cursor >= ((size_t)1 << 32)) {
			return Fail(ctx, CVX_ERR_CAPACITY, "the world needs %.2f GiB of device tables: more than the 4 GiB the 32-bit offsets of the kernel can address", (double)cursor / (double)((size_t)1 << 30));
		}
		CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		uint8_t *arena = nullptr;
		CVX_HIP(ctx, hipMalloc((void **)&arena, cursor));
		hipError_t e = hipSuccess;
		for (int i = 0; i < CVX_LOD_LEVELS && e == hipSuccess; i++) {
			cvx_context::HostLevel &H = ctx->hostLevel[i];
			if (H.pending) {
				e = hipMemcpy(arena + recordsAt[i], H.records.data(), H.recordsBytes, hipMemcpyHostToDevice);
				if (e == hipSuccess) { e = hipMemcpy(arena + runsAt[i], H.runs.data(), H.runsBytes, hipMemcpyHostToDevice); }
				if (e == hipSuccess) { e = hipMemcpy(arena + countsAt[i], H.counts.data(), H.countsBytes, hipMemcpyHostToDevice); }
				if (e == hipSuccess) { e = hipMemcpy(arena + elementsAt[i], H.elements.data(), H.elementsBytes, hipMemcpyHostToDevice); }
			} else { // unchanged level: it lives in the old arena
				const DevWorldLevel &old = ctx->hostWorld.level[i];
				e = hipMemcpy(arena + recordsAt[i], ctx->arena + old.recordsOff, H.recordsBytes, hipMemcpyDeviceToDevice);
				if (e == hipSuccess) { e = hipMemcpy(arena + runsAt[i], ctx->arena + old.runsOff, H.runsBytes, hipMemcpyDeviceToDevice); }
				if (e == hipSuccess) { e = hipMemcpy(arena + countsAt[i], ctx->arena + old.countsOff, H.countsBytes, hipMemcpyDeviceToDevice); }
				if (e == hipSuccess) { e = hipMemcpy(arena + elementsAt[i], ctx->arena + old.elementsOff, H.elementsBytes, hipMemcpyDeviceToDevice); }
			}
		}
		if (e != hipSuccess) {
			(void)hipFree(arena);
			return Fail(ctx, CVX_ERR_HIP, "world upload failed: %s", hipGetErrorString(e));
		}
		for (int i = 0; i < CVX_LOD_LEVELS; i++) {
			cvx_context::HostLevel &H = ctx->hostLevel[i];
			std::vector<uint4>().swap(H.records);
			std::vector<uint2>().swap(H.runs);
			std::vector<uint2>().swap(H.counts);
			std::vector<uint32_t>().swap(H.elements);
			H.pending = false;
			DevWorldLevel &L = next.level[i];
			L.recordsOff = (uint32_t)recordsAt[i];
			L.runsOff = (uint32_t)runsAt[i];
			L.elementsOff = (uint32_t)elementsAt[i]; // (colour indices count from the start of the padded array)
			L.shift = i;
			L.rowShift = H.rowShift;
			L.countsOff = (uint32_t)countsAt[i];
			L.colorShift = H.colorShift;
			L.pad_ = 0;
		}
		if (ctx->arena) { (void)hipFree(ctx->arena); }
		ctx->arena = arena;
		ctx->arenaBytes = cursor;
		next.arena = arena;
		ctx->hostWorld = next;
		if (!ctx->devWorld) {
			CVX_HIP(ctx, hipMalloc((void **)&ctx->devWorld, sizeof(DevWorld)));
		}
		CVX_HIP(ctx, hipMemcpyAsync(ctx->devWorld, &ctx->hostWorld, sizeof(DevWorld), hipMemcpyHostToDevice, ctx->stream));
		CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		ctx->worldDirty = false;
	}
	return CVX_OK;
}

int Launch(cvx_context *ctx, int frameCount, int flags)
{
	size_t nTiles = ctx->hostTiles.size();
	int rc = EnsureScratch(ctx, (size_t)frameCount, nTiles);
	if (rc != CVX_OK) { return rc; }
	// Frame / tile descriptors go through a small ring of pinned staging buffers so that back-to-back asynchronous
	// draws never read host memory the next call is already overwriting.
	{
		cvx_context::UploadSlot &slot = ctx->upload[ctx->uploadNext++ % cvx_context::kUploadSlots];
		const size_t framesBytes = (size_t)frameCount * sizeof(DevFrame), tilesBytes = nTiles * sizeof(DevTile);
		if (!slot.done) { CVX_HIP(ctx, hipEventCreateWithFlags(&slot.done, hipEventDisableTiming)); }
		if (slot.inFlight) { CVX_HIP(ctx, hipEventSynchronize(slot.done)); slot.inFlight = false; }
		if (slot.bytes < framesBytes + tilesBytes) {
			if (slot.pinned) { (void)hipHostFree(slot.pinned); slot.pinned = nullptr; slot.bytes = 0; }
			const size_t want = (framesBytes + tilesBytes) * 3 / 2 + 4096;
			CVX_HIP(ctx, hipHostMalloc(&slot.pinned, want, hipHostMallocDefault));
			slot.bytes = want;
		}
		std::memcpy(slot.pinned, ctx->hostFrames.data(), framesBytes);
		if (tilesBytes) { std::memcpy(static_cast<uint8_t *>(slot.pinned) + framesBytes, ctx->hostTiles.data(), tilesBytes); }
		CVX_HIP(ctx, hipMemcpyAsync(ctx->devFrames, slot.pinned, framesBytes, hipMemcpyHostToDevice, ctx->stream));
		if (tilesBytes) {
			CVX_HIP(ctx, hipMemcpyAsync(ctx->devTiles, static_cast<uint8_t *>(slot.pinned) + framesBytes, tilesBytes, hipMemcpyHostToDevice, ctx->stream));
		}
		CVX_HIP(ctx, hipEventRecord(slot.done, ctx->stream));
		slot.inFlight = true;
	}
	if (ctx->countersEnabled) {
		CVX_HIP(ctx, hipMemsetAsync(ctx->devCounters, 0, sizeof(DevCounters), ctx->stream));
	}
#ifdef CVX_TILE_TIMES
	if (nTiles > g_tileTimesCap) {
		CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (g_tileTimesDev) { (void)hipFree(g_tileTimesDev); }
		g_tileTimesCap = nTiles + nTiles / 2;
		CVX_HIP(ctx, hipMalloc((void **)&g_tileTimesDev, g_tileTimesCap * sizeof(unsigned long long)));
		CVX_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(cvxk::g_tileTimes), &g_tileTimesDev, sizeof g_tileTimesDev));
	}
#endif
	hipEvent_t evStart = nullptr, evStop = nullptr;
	rc = NextEventPair(ctx, evStart, evStop);
	if (rc != CVX_OK) { return rc; }
	CVX_HIP(ctx, hipEventRecord(evStart, ctx->stream));
	if (nTiles) {
		// ONE launch, all frames of the batch in it: the iteration direction (RenderJob.Execute :174-178) is a wave-uniform runtime switch inside the
		// kernel, so the tails of different frames overlap.  Its dynamic-LDS size is the largest mask any of its waves needs (DrawBatch).
		const size_t ldsBytes = (size_t)std::max(ctx->ldsWordsNeeded, ctx->minMaskWords * CVX_WAVE) * sizeof(uint32_t);
		dim3 grid((unsigned)(ctx->launchLone ? nTiles * CVX_WAVE : nTiles)), block(CVX_WAVE);
		const size_t loneLdsBytes = (size_t)(CVX_WAVE + ctx->lonePixels) * sizeof(uint32_t); // lone_kernel: the merge buffer + the ray's pixel row
		if (ctx->countersEnabled) {
			hipLaunchKernelGGL((cvxk::render_kernel<true>), grid, block, ldsBytes, ctx->stream, ctx->devFrames, ctx->devTiles, ctx->devWorld, ctx->devCounters);
		} else if (ctx->launchLone == 1) { // one wave per ray, lanes = columns (cvx_lone.h): the single interactive frame
			cvxi::LaunchLone(false, grid.x, loneLdsBytes, ctx->stream, ctx->devFrames, ctx->devTiles, ctx->devWorld);
		} else if (ctx->launchLone == 2) { // ... windows of more than 2048 pixels (4K)
			cvxi::LaunchLone(true, grid.x, loneLdsBytes, ctx->stream, ctx->devFrames, ctx->devTiles, ctx->devWorld);
		} else {
			hipLaunchKernelGGL((cvxk::render_kernel<false>), grid, block, ldsBytes, ctx->stream, ctx->devFrames, ctx->devTiles, ctx->devWorld, ctx->devCounters);
		}
		CVX_HIP(ctx, hipGetLastError());
	}
	CVX_HIP(ctx, hipEventRecord(evStop, ctx->stream));
	if (!(flags & CVX_DRAW_ASYNC)) {
		CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
#ifdef CVX_TILE_TIMES
		if (const char *path = std::getenv("CVX_TILE_TIMES_OUT")) {
			if (nTiles && !ctx->countersEnabled && g_waveSource.size() == nTiles) {
				std::vector<unsigned long long> ticks(nTiles);
				CVX_HIP(ctx, hipMemcpy(ticks.data(), g_tileTimesDev, nTiles * sizeof(unsigned long long), hipMemcpyDeviceToHost));
				std::vector<float> perTile(g_sourceTiles, 0.0f);
				for (size_t i = 0; i < nTiles; i++) { perTile[g_waveSource[i]] += (float)ticks[i]; }
				if (FILE *fh = std::fopen(path, "wb")) {
					std::fwrite(perTile.data(), sizeof(float), perTile.size(), fh);
					std::fclose(fh);
				}
			}
		}
#endif
	}
	return CVX_OK;
}

} // namespace

extern "C" {

const char *cvx_version(void) { return "cpuvox_gpu 0.2 (gfx950)"; }

const char *cvx_last_error(const cvx_context *ctx) { return ctx ? ctx->error.c_str() : g_createError.c_str(); }

int cvx_create(int device, cvx_context **out)
{
	if (!out) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "out is NULL"); }
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		return Fail(nullptr, CVX_ERR_HIP, "no HIP device available: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
	}
	if (device < 0 || device >= count) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device, count); }
	cvx_context *ctx = new (std::nothrow) cvx_context();
	if (!ctx) { return Fail(nullptr, CVX_ERR_HIP, "out of host memory"); }
	ctx->device = device;
	auto bail = [&](hipError_t err, const char *what) {
		Fail(nullptr, CVX_ERR_HIP, "%s failed: %s", what, hipGetErrorString(err));
		cvx_destroy(ctx);
		return CVX_ERR_HIP;
	};
	if ((e = hipSetDevice(device)) != hipSuccess) { return bail(e, "hipSetDevice"); }
	if ((e = hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking)) != hipSuccess) { return bail(e, "hipStreamCreate"); }
	ctx->stream = ctx->ownStream;
	if ((e = hipMalloc((void **)&ctx->devCounters, sizeof(DevCounters))) != hipSuccess) { return bail(e, "hipMalloc"); }
	{
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
			ctx->splitWaveBudget = prop.multiProcessorCount * 16; // measured optimum on MI355X (256 CUs): at most ~4096 waves per launch
		}
#ifdef CVX_EXPERIMENTS /* diagnostics of the experiment build only (`make gpu-exp`): the product library reads no environment */
		if (const char *v = std::getenv("CVX_TILE_SPLIT")) {
			const int f = std::atoi(v);
			if (f >= 1 && f <= CVX_WAVE && (f & (f - 1)) == 0) { ctx->forcedSplit = f; }
		}
		if (const char *v = std::getenv("CVX_MAX_WAVE_MASK_WORDS")) { // diagnostics: LDS budget per wave in mask words (x 4 bytes)
			const int w = std::atoi(v);
			if (w >= 64 && w <= 40960) { ctx->maxWaveMaskWords = w; ctx->maxWaveMaskWordsAuto = false; }
		}
		if (const char *v = std::getenv("CVX_TILE_COST_MIDDLE_RAY")) { ctx->tileCostMiddleRay = std::atoi(v) != 0; } // diagnostics: the round-1 estimate
		if (const char *v = std::getenv("CVX_TILE_COST_PIXELS")) { // diagnostics
			const float w = (float)std::atof(v);
			if (w >= 0.f && w <= 100.f) { ctx->tileCostPixelWeight = w; }
		}
		if (const char *v = std::getenv("CVX_LONE")) { // diagnostics: 0 = never the one-wave-per-ray kernel, 1 = always (whatever the batch size), unset = by the wave budget
			ctx->loneMode = std::atoi(v) != 0 ? 2 : 0;
		}
		if (const char *v = std::getenv("CVX_LONE_BUDGET")) { // diagnostics: the largest launch (in rays) the one-wave-per-ray kernel is chosen for
			const int w = std::atoi(v);
			if (w >= 0) { ctx->loneWaveBudget = w; }
		}
		if (const char *v = std::getenv("CVX_MIN_MASK_WORDS")) {
			const int w = std::atoi(v);
			if (w >= 1 && w <= 512) { ctx->minMaskWords = w; }
		}
#endif
	}
	*out = ctx;
	return CVX_OK;
}

void cvx_destroy(cvx_context *ctx)
{
	if (!ctx) { return; }
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
	FreeRaybuffers(ctx);
	if (ctx->arena) { (void)hipFree(ctx->arena); }
	if (ctx->devWorld) { (void)hipFree(ctx->devWorld); }
	if (ctx->devFrames) { (void)hipFree(ctx->devFrames); }
	if (ctx->devTiles) { (void)hipFree(ctx->devTiles); }
	if (ctx->devCounters) { (void)hipFree(ctx->devCounters); }
	if (ctx->staging) { (void)hipFree(ctx->staging); }
	if (ctx->blitParamsDev) { (void)hipFree(ctx->blitParamsDev); }
	if (ctx->blitParamsPinned) { (void)hipHostFree(ctx->blitParamsPinned); }
	for (hipEvent_t e : ctx->evPairs) { (void)hipEventDestroy(e); }
	for (auto &slot : ctx->upload) {
		if (slot.pinned) { (void)hipHostFree(slot.pinned); }
		if (slot.done) { (void)hipEventDestroy(slot.done); }
	}
	if (ctx->ownStream) { (void)hipStreamDestroy(ctx->ownStream); }
	delete ctx;
}

int cvx_set_stream(cvx_context *ctx, void *hipStream)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	ctx->stream = hipStream ? (hipStream_t)hipStream : ctx->ownStream;
	return CVX_OK;
}

int cvx_set_buffer_count(cvx_context *ctx, int bufferCount)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (bufferCount < 1 || bufferCount > 4096) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bufferCount must be in [1, 4096]"); }
	if (bufferCount != ctx->bufferCount) {
		CVX_HIP(ctx, hipSetDevice(ctx->device));
		CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		int rx = ctx->resX, ry = ctx->resY;
		FreeRaybuffers(ctx);
		ctx->bufferCount = bufferCount;
		if (rx > 0) { return cvx_set_resolution(ctx, rx, ry); }
	}
	return CVX_OK;
}

int cvx_set_resolution(cvx_context *ctx, int resolutionX, int resolutionY)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (resolutionX <= 0 || resolutionY <= 0 || resolutionX > 16384 || resolutionY > 16384) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "resolution out of range");
	}
	if (resolutionX == ctx->resX && resolutionY == ctx->resY && !ctx->poolTD.empty()) {
		return CVX_OK; // RenderManager.SetResolution returns false when nothing changes (:96,108)
	}
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	FreeRaybuffers(ctx);
	const int W = resolutionX, H = resolutionY;
	// capacities of RenderManager.cs:35-36 in whole tiles, +2 tiles of slack for
	// the per-segment round-up to 64 rays
	ctx->tilesTD = (W + 2 * H + CVX_WAVE - 1) / CVX_WAVE + 2;
	ctx->tilesLR = (2 * W + H + CVX_WAVE - 1) / CVX_WAVE + 2;
	ctx->poolBytesTD = (size_t)ctx->tilesTD * (size_t)H * CVX_WAVE * 4;
	ctx->poolBytesLR = (size_t)ctx->tilesLR * (size_t)W * CVX_WAVE * 4;
	ctx->poolTD.assign((size_t)ctx->bufferCount, nullptr);
	ctx->poolLR.assign((size_t)ctx->bufferCount, nullptr);
	ctx->last.assign((size_t)ctx->bufferCount, LastDraw());
	// the pools themselves are allocated on first use (EnsurePools) so that cvx_bind_raybuffers can supply
	// caller-owned memory without a transient second copy
	CVX_HIP(ctx, hipMalloc((void **)&ctx->screen, (size_t)W * (size_t)H * 4));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	ctx->resX = W;
	ctx->resY = H;
	return CVX_OK;
}

int cvx_bind_raybuffers(cvx_context *ctx, void *topDown, int64_t topDownBytes, void *leftRight, int64_t leftRightBytes)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (ctx->poolTD.empty()) { return Fail(ctx, CVX_ERR_NOT_READY, "call cvx_set_resolution first"); }
	if (!topDown || !leftRight || topDownBytes < (int64_t)(ctx->poolBytesTD * (size_t)ctx->bufferCount) ||
	    leftRightBytes < (int64_t)(ctx->poolBytesLR * (size_t)ctx->bufferCount) || ((uintptr_t)topDown & 255u) || ((uintptr_t)leftRight & 255u)) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "external raybuffers too small or not 256-byte aligned");
	}
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (!ctx->poolsExternal) {
		if (ctx->poolBaseTD) { (void)hipFree(ctx->poolBaseTD); }
		if (ctx->poolBaseLR) { (void)hipFree(ctx->poolBaseLR); }
	}
	ctx->poolsExternal = true;
	ctx->poolBaseTD = static_cast<uint32_t *>(topDown);
	ctx->poolBaseLR = static_cast<uint32_t *>(leftRight);
	for (int i = 0; i < ctx->bufferCount; i++) {
		ctx->poolTD[(size_t)i] = ctx->poolBaseTD + (size_t)i * (ctx->poolBytesTD / 4);
		ctx->poolLR[(size_t)i] = ctx->poolBaseLR + (size_t)i * (ctx->poolBytesLR / 4);
		ctx->last[(size_t)i] = LastDraw();
	}
	return CVX_OK;
}

int cvx_set_latency_kernel(cvx_context *ctx, int mode)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (mode != CVX_LATENCY_AUTO && mode != CVX_LATENCY_NEVER && mode != CVX_LATENCY_ALWAYS) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad latency-kernel mode %d", mode); }
	ctx->loneMode = mode == CVX_LATENCY_AUTO ? 1 : (mode == CVX_LATENCY_NEVER ? 0 : 2);
	return CVX_OK;
}

int cvx_set_shard(cvx_context *ctx, int shardIndex, int shardCount)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (shardCount < 1 || shardIndex < 0 || shardIndex >= shardCount) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad shard %d/%d", shardIndex, shardCount); }
	ctx->shardIndex = shardIndex;
	ctx->shardCount = shardCount;
	return CVX_OK;
}

} // extern "C"

namespace {
// Shared body of cvx_draw_segments_batch (library-owned tile layout) and cvx_draw_segments_placed (caller-chosen
// output address per tile).
int DrawBatch(cvx_context *ctx, int frameCount, const cvx_segment_data *segments, const cvx_camera_data *cameras,
              int screenWidth, int screenHeight, const float *vanishingPoints, int firstBufferIndex,
              const uint64_t *placements, int64_t placementCount, int flags)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (frameCount <= 0 || !segments || !cameras || !vanishingPoints) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad frame arguments"); }
	if (ctx->poolTD.empty() || screenWidth != ctx->resX || screenHeight != ctx->resY) {
		return Fail(ctx, CVX_ERR_NOT_READY, "cvx_set_resolution(%d, %d) has not been called", screenWidth, screenHeight);
	}
	if (!placements && (firstBufferIndex < 0 || firstBufferIndex >= ctx->bufferCount || frameCount > ctx->bufferCount)) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "buffer index %d / %d frames do not fit bufferCount %d", firstBufferIndex, frameCount, ctx->bufferCount);
	}
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	int rc = placements ? CVX_OK : EnsurePools(ctx);
	if (rc != CVX_OK) { return rc; }
	rc = SyncWorld(ctx);
	if (rc != CVX_OK) { return rc; }
	int64_t placeCursor = 0;
	LastDraw placedScratch;
	ctx->hostFrames.assign((size_t)frameCount, DevFrame());
	ctx->hostTiles.clear();
	ctx->hostTileCost.clear();
	ctx->hostTileWords.clear();
	for (int f = 0; f < frameCount; f++) {
		int b = placements ? 0 : (firstBufferIndex + f) % ctx->bufferCount;
		rc = BuildFrame(ctx, segments + (size_t)f * 4, cameras + f, screenWidth, screenHeight, vanishingPoints + (size_t)f * 2, b, f,
		                ctx->hostFrames[(size_t)f], ctx->hostTiles, ctx->hostTileCost, ctx->hostTileWords, placements ? placedScratch : ctx->last[(size_t)b],
		                placements, placementCount, placeCursor);
		if (rc != CVX_OK) { return rc; }
	}
	if (placements && placeCursor != placementCount) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "tile placement list has %lld entries, the batch has %lld tiles", (long long)placementCount, (long long)placeCursor);
	}
	// Longest tiles first: the hardware dispatches workgroups in blockIdx order, so the tail of the launch is made of
	// short tiles (LPT scheduling).  XCD-affine and frame-major orders were measured and were no better / worse.
	{
		const size_t n = ctx->hostTiles.size();
		std::vector<uint32_t> order(n);
		ctx->maskWordsNeeded = 1;
		for (size_t i = 0; i < n; i++) {
			order[i] = (uint32_t)i;
			if (ctx->hostTileWords[i] > ctx->maskWordsNeeded) { ctx->maskWordsNeeded = ctx->hostTileWords[i]; }
		}
#ifdef CVX_TILE_TIMES
		if (const char *path = std::getenv("CVX_TILE_EST_OUT")) { // the library's own estimates, one float per tile
			if (FILE *fh = std::fopen(path, "wb")) {
				std::fwrite(ctx->hostTileCost.data(), sizeof(float), n, fh);
				std::fclose(fh);
			}
		}
		if (const char *path = std::getenv("CVX_TILE_COST_FILE")) { // diagnostic build only (tools/lpt_oracle.py): launch order from measured costs, one float per tile
			if (FILE *fh = std::fopen(path, "rb")) {
				std::vector<float> measured(n);
				if (n > 0 && std::fread(measured.data(), sizeof(float), n, fh) == n && std::fgetc(fh) == EOF) { ctx->hostTileCost = measured; }
				std::fclose(fh);
			}
		}
#endif
		const std::vector<float> &cost = ctx->hostTileCost;
		// descending by cost, ties in tile order (a stable sort): estimates are non-negative floats, whose bit patterns order like their values, so four
		// stable counting passes over the inverted bits do it (30 000 tiles of a 512-frame batch: the host part of the draw call 2.8 -> 1.2 ms against std::stable_sort with an indirect compare; four threads for the frame descriptors on top were slower than one)
		bool radix = n >= 2048;
		for (size_t i = 0; i < n && radix; i++) { radix = cost[i] >= 0.0f; } // (a negative or NaN estimate: the general sort)
		if (radix) {
			std::vector<uint32_t> keys(n), other(n), keys2(n);
			for (size_t i = 0; i < n; i++) {
				uint32_t bits;
				std::memcpy(&bits, &cost[i], 4);
				keys[i] = ~(bits == 0x80000000u ? 0u : bits); // (-0 orders with +0)
			}
			for (int pass = 0; pass < 4; pass++) {
				size_t count[257] = { 0 };
				const int shift = pass * 8;
				for (size_t i = 0; i < n; i++) { count[((keys[i] >> shift) & 0xFFu) + 1]++; }
				for (int b = 0; b < 256; b++) { count[b + 1] += count[b]; }
				for (size_t i = 0; i < n; i++) {
					const size_t at = count[(keys[i] >> shift) & 0xFFu]++;
					keys2[at] = keys[i];
					other[at] = order[i];
				}
				keys.swap(keys2);
				order.swap(other);
			}
		} else {
			std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
		}
#ifdef CVX_EXPERIMENTS
		if (const char *v = std::getenv("CVX_TILE_ORDER")) { // diagnostics: how much the launch order matters
			if (!std::strcmp(v, "reverse")) {
				std::reverse(order.begin(), order.end());
			} else if (!std::strcmp(v, "frame")) {
				for (size_t i = 0; i < n; i++) { order[i] = (uint32_t)i; }
			} else if (!std::strncmp(v, "snake", 5) && std::atoi(v + 5) > 0) { // every other block of N tiles reversed: long and short tiles alternate on the units the blocks land on
				const size_t period = (size_t)std::atoi(v + 5);
				for (size_t at = period; at < n; at += 2 * period) { std::reverse(order.begin() + (ptrdiff_t)at, order.begin() + (ptrdiff_t)std::min(n, at + period)); }
			} else if (!std::strcmp(v, "random")) {
				uint32_t state = 12345u;
				for (size_t i = n; i > 1; i--) {
					state = state * 1664525u + 1013904223u;
					std::swap(order[i - 1], order[(size_t)(state >> 8) % i]);
				}
			}
		}
#endif
		// Small batches (a single interactive frame is ~60 tiles on a chip with 1024 SIMDs): every tile is cut into 2, 4, ...
		// 64 sub-tiles of consecutive rays, one wave each.  A wave's cost per column step is the union of what its rays
		// need, so narrower waves finish sooner; with few waves there are idle SIMDs to run them on (1 frame: 6.7 -> 4.0 ms at
		// 2 rays per wave, 16 frames: 8.6 -> 6.3 ms at 16).  Once the chip is full the fixed per-wave part dominates and splitting loses (256
		// frames: x1.6 slower at split 2), hence the wave budget.
		// A launch of few rays (the reference's call pattern: ONE frame per blocking call, RenderManager.cs:358-363) goes to the latency kernel
		// (cvx_lone.h): one wave per RAY, its lanes the ray's next 64 columns.  Its mask lives in one or two vector registers: windows of up to 4096 pixels.
		ctx->launchLone = 0;
		// AUTO's budget is in rays (= waves).  The crossover against the batch kernel is at ~12 000 - 14 000 rays at 1080p (2 - 3 frames) and between 8 000 and
		// 11 000 at 4K, where a ray has twice the pixels to write and so twice the events (budget sweep and per-pose crossover: profiles/r06_latency.md)
		const size_t loneBudget = (long long)ctx->resX * ctx->resY > 2560ll * 1440ll ? (size_t)ctx->loneWaveBudget * 3 / 4 : (size_t)ctx->loneWaveBudget;
		if (!ctx->countersEnabled && ctx->maskWordsNeeded <= 2 * CVX_WAVE && (ctx->loneMode == 2 || (ctx->loneMode == 1 && n * (size_t)CVX_WAVE <= loneBudget)) && n > 0) {
			// (one workgroup per RAY: the kernel takes tile blockIdx / 64, ray blockIdx % 64 of it -- the tile list goes to the device as it is, longest tiles first)
			std::vector<DevTile> sortedTiles;
			sortedTiles.reserve(n);
			for (size_t i = 0; i < n; i++) { sortedTiles.push_back(ctx->hostTiles[order[i]]); }
			ctx->hostTiles.swap(sortedTiles);
			ctx->launchLone = ctx->maskWordsNeeded > CVX_WAVE ? 2 : 1;
			ctx->lonePixels = 0;
			for (const DevFrame &F : ctx->hostFrames) {
				for (int sIdx = 0; sIdx < 4; sIdx++) {
					if (F.seg[sIdx].rayCount > 0) { ctx->lonePixels = std::max(ctx->lonePixels, F.seg[sIdx].omax - F.seg[sIdx].omin + 1); }
				}
			}
			ctx->ldsWordsNeeded = 0;
			return Launch(ctx, frameCount, flags);
		}
		int split = 1;
		while (split < CVX_WAVE && n * (size_t)split * 2 <= (size_t)ctx->splitWaveBudget) { split *= 2; }
		if (ctx->forcedSplit > 0) { split = ctx->forcedSplit; }
		// LDS: the seen mask of a wave is words * lanes * 4 bytes.  Measured (profiles/r02_occupancy_sweep.txt): the kernel
		// gains steadily up to the 16 waves per CU its 128 VGPRs allow, which needs <= 10 KB of LDS per wave = 40 words at 64
		// lanes.  A tile whose window needs more words (a left / right segment spanning most of a 1920-pixel row, any
		// segment at 4K) is therefore rendered by 2, 4, ... narrower waves, so ONE wide tile no longer takes the
		// occupancy of the whole launch down.
		// The budget itself is chosen per launch: a small one keeps 16 waves resident but cuts wide tiles into narrow waves
		// (which repeat the per-step work of a wave), a large one keeps the tiles whole at fewer resident waves.  Cost model
		// fitted to the sweeps in profiles/r02_occupancy.md: a wave of 64 / 32 / 16 / 8 ... lanes costs 1 / 0.68 / 0.40 / 0.25 ... of
		// a full one, and throughput grows with (resident waves)^0.4 (at most 16 per CU: 128 VGPRs).  1080p picks 10 KB (every
		// top / bottom tile stays whole), 4K picks 17 KB (68-word top / bottom tiles whole, left / right tiles halved).
		// (Measured and archived, tools/patches/exp_lds_classes_and_per_tile_rule.patch + profiles/r04_experiments.md: one launch per mask-size class on
		// streams of their own -- 4K / 2048^3 25.0 against 23.2 ms -- and a split chosen per tile instead of by one budget per draw: +3.6 % at 1080p.)
		static const double laneCost[7] = { 1.0, 0.68, 0.40, 0.25, 0.16, 0.11, 0.08 }; // 64, 32, 16, 8, 4, 2, 1 lanes
		auto residentWaves = [](int waveWords) { return std::max(1, std::min(16, (int)(163840 / ((size_t)waveWords * 4)))); };
		int baseLevel = 0;
		while ((1 << baseLevel) < split) { baseLevel++; }
		int budget = ctx->maxWaveMaskWords; // (diagnostics: CVX_MAX_WAVE_MASK_WORDS pins it)
		if (ctx->maxWaveMaskWordsAuto) {
			static const int candidates[] = { 40 * CVX_WAVE, 48 * CVX_WAVE, 60 * CVX_WAVE, 68 * CVX_WAVE, 80 * CVX_WAVE, 96 * CVX_WAVE, 120 * CVX_WAVE, 160 * CVX_WAVE, 256 * CVX_WAVE };
			double bestCost = 0.0;
			for (int candidate : candidates) {
				const int resident = residentWaves(candidate);
				double work = 0.0;
				for (size_t i = 0; i < n; i++) {
					int s2 = split, level = baseLevel;
					while (s2 < CVX_WAVE && ctx->hostTileWords[i] * (CVX_WAVE / s2) > candidate) { s2 *= 2; level++; }
					work += (double)s2 * laneCost[level];
				}
				const double cost = work / std::pow((double)resident, 0.4);
				if (bestCost == 0.0 || cost < bestCost) { bestCost = cost; budget = candidate; }
			}
		}
		std::vector<DevTile> sorted;
		sorted.reserve(n * (size_t)split);
#ifdef CVX_TILE_TIMES
		g_waveSource.clear();
		g_sourceTiles = n;
#endif
		int ldsWords = 1; // words * lanes of the largest wave
		for (size_t i = 0; i < n; i++) {
			DevTile t = ctx->hostTiles[order[i]];
			const int words = ctx->hostTileWords[order[i]];
			int tileSplit = split;
			while (tileSplit < CVX_WAVE && words * (CVX_WAVE / tileSplit) > budget) { tileSplit *= 2; }
			const int lanesPerWave = CVX_WAVE / tileSplit;
			ldsWords = std::max(ldsWords, words * lanesPerWave);
			if (tileSplit == 1) {
				sorted.push_back(t);
#ifdef CVX_TILE_TIMES
				g_waveSource.push_back(order[i]);
#endif
				continue;
			}
			for (int k = 0; k < tileSplit; k++) {
				// A wave with 8 or fewer active lanes issues vector instructions ~3.6 x slower than one with 16 (tools/valu_rate.hip, gfx950),
				// so the rays of a narrow sub-tile are worked on by 64 / laneCount
				// lanes each: same addresses, same values, same stores from every lane of a group (not with the counters on: they are summed over lanes).
				int dupShift = 0;
				while (!ctx->countersEnabled && (lanesPerWave << dupShift) < CVX_WAVE) { dupShift++; }
				t.lanes = (k * lanesPerWave) | (lanesPerWave << 8) | (dupShift << 16);
				sorted.push_back(t);
#ifdef CVX_TILE_TIMES
				g_waveSource.push_back(order[i]);
#endif
			}
		}
		ctx->ldsWordsNeeded = ldsWords;
		ctx->hostTiles.swap(sorted);
	}
	return Launch(ctx, frameCount, flags);
}
} // namespace

extern "C" {

int cvx_draw_segments_batch(cvx_context *ctx, int frameCount, const cvx_segment_data *segments, const cvx_camera_data *cameras,
                            int screenWidth, int screenHeight, const float *vanishingPoints, int firstBufferIndex, int flags)
{
	return DrawBatch(ctx, frameCount, segments, cameras, screenWidth, screenHeight, vanishingPoints, firstBufferIndex, nullptr, 0, flags);
}

int cvx_draw_segments_placed(cvx_context *ctx, int frameCount, const cvx_segment_data *segments, const cvx_camera_data *cameras,
                             int screenWidth, int screenHeight, const float *vanishingPoints, int64_t tileCount, const uint64_t *tileOut, int flags)
{
	if (ctx && !tileOut) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "tileOut is NULL"); }
	return DrawBatch(ctx, frameCount, segments, cameras, screenWidth, screenHeight, vanishingPoints, 0, tileOut, tileCount, flags);
}

int cvx_draw_segments(cvx_context *ctx, const cvx_segment_data segments[4], const cvx_camera_data *camera, int screenWidth, int screenHeight,
                      const float vanishingPointScreenSpace[2], int bufferIndex, int flags)
{
	return cvx_draw_segments_batch(ctx, 1, segments, camera, screenWidth, screenHeight, vanishingPointScreenSpace, bufferIndex, flags);
}

int cvx_synchronize(cvx_context *ctx)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return CVX_OK;
}

int cvx_clear_raybuffer(cvx_context *ctx, int bufferIndex, int which, uint32_t argb)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (ctx->poolTD.empty() || bufferIndex < 0 || bufferIndex >= ctx->bufferCount || (which != 0 && which != 1)) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad buffer selection");
	}
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	{ int rc = EnsurePools(ctx); if (rc != CVX_OK) { return rc; } }
	uint32_t *p = which == 0 ? ctx->poolTD[(size_t)bufferIndex] : ctx->poolLR[(size_t)bufferIndex];
	size_t bytes = which == 0 ? ctx->poolBytesTD : ctx->poolBytesLR;
	CVX_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)p, (int)argb, bytes / 4, ctx->stream));
	return CVX_OK;
}

int cvx_read_raybuffer(cvx_context *ctx, int bufferIndex, int which, int firstRay, int rayCount, void *dst)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (ctx->poolTD.empty() || bufferIndex < 0 || bufferIndex >= ctx->bufferCount || (which != 0 && which != 1) || !dst) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad buffer selection");
	}
	const int W = ctx->resX, H = ctx->resY;
	const int width = which == 0 ? H : W;
	const int capacity = which == 0 ? W + 2 * H : 2 * W + H;
	if (firstRay < 0 || rayCount < 0 || firstRay + rayCount > capacity) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "ray range outside the buffer"); }
	if (rayCount == 0) { return CVX_OK; }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	{ int rc = EnsurePools(ctx); if (rc != CVX_OK) { return rc; } }
	const size_t bytes = (size_t)rayCount * (size_t)width * 4;
	if (bytes > ctx->stagingBytes) {
		if (ctx->staging) { (void)hipFree(ctx->staging); ctx->staging = nullptr; ctx->stagingBytes = 0; }
		CVX_HIP(ctx, hipMalloc((void **)&ctx->staging, bytes));
		ctx->stagingBytes = bytes;
	}
	const LastDraw &last = ctx->last[(size_t)bufferIndex];
	const int s0 = which == 0 ? 0 : 2;
	const int count0 = last.valid ? (last.segments[s0].RayCount > 0 ? last.segments[s0].RayCount : 0) : capacity;
	const int tileBase1 = last.valid ? last.tileBase[s0 + 1] : 0;
	const uint32_t *pool = which == 0 ? ctx->poolTD[(size_t)bufferIndex] : ctx->poolLR[(size_t)bufferIndex];
	dim3 block(256), grid((unsigned)((width + 255) / 256), (unsigned)rayCount);
	hipLaunchKernelGGL(cvxk::untile_kernel, grid, block, 0, ctx->stream, pool, ctx->staging, firstRay, rayCount, width, count0, tileBase1,
	                   which == 0 ? ctx->tilesTD : ctx->tilesLR);
	CVX_HIP(ctx, hipGetLastError());
	CVX_HIP(ctx, hipMemcpyAsync(dst, ctx->staging, bytes, hipMemcpyDeviceToHost, ctx->stream));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return CVX_OK;
}

// edge functions of the four triangles (VP, MaxScreen, MinScreen), float32 operation by operation, mirrored by the numpy rule the tests compare against (blit_reference)
static void MakeBlitParams(const cvx_segment_data *segments, const float *vp, const int *tileBase, int W, int H, cvxk::BlitParams &p)
{
	const float ax = vp[0], ay = vp[1];
	for (int s = 0; s < 4; s++) {
		const float qx = segments[s].MinScreen[0], qy = segments[s].MinScreen[1];
		const float bx = segments[s].MaxScreen[0], by = segments[s].MaxScreen[1];
		const float den = (by - qy) * (ax - qx) + (qx - bx) * (ay - qy);
		const float inv = 1.0f / den;
		p.qx[s] = qx;
		p.qy[s] = qy;
		p.a0[s] = (by - qy) * inv;
		p.b0[s] = (qx - bx) * inv;
		p.a1[s] = (qy - ay) * inv;
		p.b1[s] = (ax - qx) * inv;
		p.rayCount[s] = segments[s].RayCount;
		p.tileBase[s] = tileBase ? tileBase[s] : 0;
	}
	p.width = W;
	p.height = H;
	p.clearColor = 0u;
}

static void FillBlitParams(const cvx_context *ctx, int bufferIndex, cvxk::BlitParams &p)
{
	const LastDraw &last = ctx->last[(size_t)bufferIndex];
	MakeBlitParams(last.segments, last.vp, last.tileBase, ctx->resX, ctx->resY, p);
}

int cvx_blit_segments_batch(cvx_context *ctx, int firstBufferIndex, int frameCount, void *dstDevice, void **imagesDevice)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (imagesDevice) { *imagesDevice = nullptr; }
	if (ctx->poolTD.empty() || frameCount <= 0 || firstBufferIndex < 0 || firstBufferIndex + frameCount > ctx->bufferCount) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad buffer range %d + %d of %d", firstBufferIndex, frameCount, ctx->bufferCount);
	}
	for (int f = 0; f < frameCount; f++) {
		if (!ctx->last[(size_t)(firstBufferIndex + f)].valid) { return Fail(ctx, CVX_ERR_NOT_READY, "nothing has been drawn into buffer %d", firstBufferIndex + f); }
	}
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	const size_t imageBytes = (size_t)ctx->resX * (size_t)ctx->resY * 4;
	uint32_t *images = static_cast<uint32_t *>(dstDevice);
	if (!images) {
		if (ctx->screenBatchFrames < frameCount) {
			CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
			if (ctx->screenBatch) { (void)hipFree(ctx->screenBatch); ctx->screenBatch = nullptr; ctx->screenBatchFrames = 0; }
			CVX_HIP(ctx, hipMalloc((void **)&ctx->screenBatch, imageBytes * (size_t)frameCount));
			ctx->screenBatchFrames = frameCount;
		}
		images = ctx->screenBatch;
	}
	if (ctx->blitParamsCapacity < frameCount) {
		CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->blitParamsDev) { (void)hipFree(ctx->blitParamsDev); ctx->blitParamsDev = nullptr; }
		if (ctx->blitParamsPinned) { (void)hipHostFree(ctx->blitParamsPinned); ctx->blitParamsPinned = nullptr; }
		ctx->blitParamsCapacity = 0;
		CVX_HIP(ctx, hipMalloc(&ctx->blitParamsDev, sizeof(cvxk::BlitParams) * (size_t)frameCount));
		CVX_HIP(ctx, hipHostMalloc(&ctx->blitParamsPinned, sizeof(cvxk::BlitParams) * (size_t)frameCount, hipHostMallocDefault));
		ctx->blitParamsCapacity = frameCount;
	} else {
		CVX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the staging copy of the previous batch may still be read by its upload
	}
	cvxk::BlitParams *host = static_cast<cvxk::BlitParams *>(ctx->blitParamsPinned);
	for (int f = 0; f < frameCount; f++) { FillBlitParams(ctx, firstBufferIndex + f, host[f]); }
	CVX_HIP(ctx, hipMemcpyAsync(ctx->blitParamsDev, host, sizeof(cvxk::BlitParams) * (size_t)frameCount, hipMemcpyHostToDevice, ctx->stream));
	dim3 block(256); // one workgroup per 64 x 64 pixels (blit_block)
	dim3 grid((unsigned)((ctx->resX + CVX_BLIT_TILE - 1) / CVX_BLIT_TILE), (unsigned)((ctx->resY + CVX_BLIT_TILE - 1) / CVX_BLIT_TILE), (unsigned)frameCount);
	hipLaunchKernelGGL(cvxk::blit_batch_kernel, grid, block, 0, ctx->stream, ctx->poolBaseTD, ctx->poolBaseLR, ctx->poolBytesTD / 4, ctx->poolBytesLR / 4, images,
	                   static_cast<const cvxk::BlitParams *>(ctx->blitParamsDev), firstBufferIndex);
	CVX_HIP(ctx, hipGetLastError());
	if (imagesDevice) { *imagesDevice = images; }
	return CVX_OK;
}

int cvx_blit_segments(cvx_context *ctx, int bufferIndex, void *dstHost)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (ctx->poolTD.empty() || bufferIndex < 0 || bufferIndex >= ctx->bufferCount) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad buffer selection"); }
	const LastDraw &last = ctx->last[(size_t)bufferIndex];
	if (!last.valid) { return Fail(ctx, CVX_ERR_NOT_READY, "nothing has been drawn into buffer %d", bufferIndex); }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	cvxk::BlitParams p;
	FillBlitParams(ctx, bufferIndex, p);
	dim3 block(256), grid((unsigned)((p.width + CVX_BLIT_TILE - 1) / CVX_BLIT_TILE), (unsigned)((p.height + CVX_BLIT_ROWS_SINGLE - 1) / CVX_BLIT_ROWS_SINGLE));
	hipLaunchKernelGGL(cvxk::blit_kernel, grid, block, 0, ctx->stream, ctx->poolTD[(size_t)bufferIndex], ctx->poolLR[(size_t)bufferIndex], ctx->screen, p);
	CVX_HIP(ctx, hipGetLastError());
	if (dstHost) {
		CVX_HIP(ctx, hipMemcpyAsync(dstHost, ctx->screen, (size_t)p.width * (size_t)p.height * 4, hipMemcpyDeviceToHost, ctx->stream));
		CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	return CVX_OK;
}

// ---- multi-GPU image gather (include/cpuvox_gpu.h; kernels and the scheme: cvx_kernels.h, "Multi-GPU image gather") ---------------
struct cvx_image_plan {
	int rank = 0, worldSize = 1, W = 0, H = 0, frameCount = 0;
	int64_t tileCount = 0;         // canonical tiles of the batch (all ranks)
	int64_t localSlots = 0;        // tiles I render
	size_t tileStrideWords = 0;    // words per tile in my compact store: 64 * max(W, H)
	int imagesMine = 0;            // frames I display
	std::vector<cvxk::ImageFrame> frames;
	std::vector<int64_t> firstTile; // canonical index of the first tile of every frame (frameCount + 1)
	std::vector<int64_t> sendStart, recvStart; // N + 1 boundaries, pixels
	cvxk::ImageFrame *devFrames = nullptr;
	int *devRowBase = nullptr;
	long long *devTotals = nullptr;
};

void cvx_image_plan_destroy(cvx_image_plan *plan)
{
	if (!plan) { return; }
	if (plan->devFrames) { (void)hipFree(plan->devFrames); }
	if (plan->devRowBase) { (void)hipFree(plan->devRowBase); }
	if (plan->devTotals) { (void)hipFree(plan->devTotals); }
	delete plan;
}

int cvx_image_plan_create(cvx_context *ctx, int frameCount, const cvx_segment_data *segments, const float *vanishingPoints, int screenWidth, int screenHeight,
                          int rank, int worldSize, cvx_image_plan **out)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!out) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "out is NULL"); }
	*out = nullptr;
	if (frameCount <= 0 || !segments || !vanishingPoints || screenWidth <= 0 || screenHeight <= 0 || screenWidth > 16384 || screenHeight > 16384 ||
	    worldSize < 1 || worldSize > 8 || rank < 0 || rank >= worldSize) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad image plan arguments (at most 8 ranks)");
	}
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	std::unique_ptr<cvx_image_plan, void (*)(cvx_image_plan *)> p(new (std::nothrow) cvx_image_plan(), cvx_image_plan_destroy);
	if (!p) { return Fail(ctx, CVX_ERR_HIP, "out of host memory"); }
	const int N = worldSize, W = screenWidth, H = screenHeight;
	try {
		p->rank = rank; p->worldSize = N; p->W = W; p->H = H; p->frameCount = frameCount;
		p->tileStrideWords = (size_t)CVX_WAVE * (size_t)std::max(W, H);
		p->frames.resize((size_t)frameCount);
		p->firstTile.assign((size_t)frameCount + 1, 0);
		int64_t slots = 0;
		for (int b = 0; b < frameCount; b++) {
			cvxk::ImageFrame &F = p->frames[(size_t)b];
			std::memset(&F, 0, sizeof F);
			MakeBlitParams(segments + (size_t)b * 4, vanishingPoints + (size_t)b * 2, nullptr, W, H, F.p);
			int t = 0;
			for (int s = 0; s < 4; s++) {
				F.tileStart[s] = t;
				const int rays = segments[(size_t)b * 4 + s].RayCount > 0 ? segments[(size_t)b * 4 + s].RayCount : 0;
				t += (rays + CVX_WAVE - 1) / CVX_WAVE;
			}
			F.root = b % N;
			F.imageSlot = b / N;
			F.localSlotBase = (int)slots;
			slots += (t > rank) ? (t - rank + N - 1) / N : 0; // canonical tiles rank, rank + N, ... of this frame
			p->firstTile[(size_t)b + 1] = p->firstTile[(size_t)b] + t;
			if (F.root == rank) { p->imagesMine++; }
		}
		if (slots > 0x7FFFFFFF) { return Fail(ctx, CVX_ERR_CAPACITY, "too many tiles for one image plan"); }
		p->localSlots = slots;
		p->tileCount = p->firstTile[(size_t)frameCount];
	} catch (const std::exception &e) {
		return Fail(ctx, CVX_ERR_HIP, "cvx_image_plan_create: %s", e.what());
	}
	const size_t frameBytes = sizeof(cvxk::ImageFrame) * (size_t)frameCount;
	CVX_HIP(ctx, hipMalloc((void **)&p->devFrames, frameBytes));
	CVX_HIP(ctx, hipMalloc((void **)&p->devRowBase, sizeof(int) * 8 * (size_t)H * (size_t)frameCount));
	CVX_HIP(ctx, hipMalloc((void **)&p->devTotals, sizeof(long long) * 8 * (size_t)frameCount));
	CVX_HIP(ctx, hipMemcpyAsync(p->devFrames, p->frames.data(), frameBytes, hipMemcpyHostToDevice, ctx->stream));
	hipLaunchKernelGGL(cvxk::image_count_kernel, dim3((unsigned)H, (unsigned)frameCount), dim3(CVX_WAVE), 0, ctx->stream, p->devFrames, N, p->devRowBase);
	hipLaunchKernelGGL(cvxk::image_scan_kernel, dim3((unsigned)((frameCount * 8 + 255) / 256)), dim3(256), 0, ctx->stream, frameCount, H, p->devRowBase, p->devTotals);
	CVX_HIP(ctx, hipGetLastError());
	std::vector<long long> totals((size_t)frameCount * 8);
	CVX_HIP(ctx, hipMemcpyAsync(totals.data(), p->devTotals, sizeof(long long) * totals.size(), hipMemcpyDeviceToHost, ctx->stream));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	// streams: (src -> dst) holds the frames dst displays, in frame order, each with src's pixels of it in row-major order
	std::vector<int64_t> sendLen((size_t)N, 0), recvLen((size_t)N, 0);
	for (int b = 0; b < frameCount; b++) {
		const int root = b % N;
		if (root == rank) {
			for (int r = 0; r < N; r++) { if (r != rank) { recvLen[(size_t)r] += totals[(size_t)b * 8 + (size_t)r]; } }
		} else {
			sendLen[(size_t)root] += totals[(size_t)b * 8 + (size_t)rank];
		}
	}
	p->sendStart.assign((size_t)N + 1, 0);
	p->recvStart.assign((size_t)N + 1, 0);
	for (int i = 0; i < N; i++) {
		p->sendStart[(size_t)i + 1] = p->sendStart[(size_t)i] + sendLen[(size_t)i];
		p->recvStart[(size_t)i + 1] = p->recvStart[(size_t)i] + recvLen[(size_t)i];
	}
	std::vector<int64_t> sendAt(p->sendStart.begin(), p->sendStart.end() - 1), recvAt(p->recvStart.begin(), p->recvStart.end() - 1);
	for (int b = 0; b < frameCount; b++) {
		cvxk::ImageFrame &F = p->frames[(size_t)b];
		if (F.root == rank) {
			for (int r = 0; r < N; r++) {
				if (r == rank) { continue; }
				F.recvBase[r] = recvAt[(size_t)r];
				recvAt[(size_t)r] += totals[(size_t)b * 8 + (size_t)r];
			}
		} else {
			F.sendBase = sendAt[(size_t)F.root];
			sendAt[(size_t)F.root] += totals[(size_t)b * 8 + (size_t)rank];
		}
	}
	CVX_HIP(ctx, hipMemcpyAsync(p->devFrames, p->frames.data(), frameBytes, hipMemcpyHostToDevice, ctx->stream));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out = p.release();
	return CVX_OK;
}

int64_t cvx_image_plan_tile_count(const cvx_image_plan *plan) { return plan ? plan->tileCount : 0; }

int cvx_image_plan_sizes(const cvx_image_plan *plan, int64_t *localStoreBytes, int64_t *sendPixels, int64_t *recvPixels, int32_t *imagesDisplayed)
{
	if (!plan) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "plan is NULL"); }
	if (localStoreBytes) { *localStoreBytes = (int64_t)((size_t)plan->localSlots * plan->tileStrideWords * 4); }
	if (sendPixels) { *sendPixels = plan->sendStart.back(); }
	if (recvPixels) { *recvPixels = plan->recvStart.back(); }
	if (imagesDisplayed) { *imagesDisplayed = plan->imagesMine; }
	return CVX_OK;
}

int cvx_image_plan_transfer(const cvx_image_plan *plan, int peer, int64_t *sendPixel, int64_t *sendPixels, int64_t *recvPixel, int64_t *recvPixels)
{
	if (!plan || peer < 0 || peer >= plan->worldSize || !sendPixel || !sendPixels || !recvPixel || !recvPixels) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	*sendPixel = plan->sendStart[(size_t)peer];
	*sendPixels = plan->sendStart[(size_t)peer + 1] - plan->sendStart[(size_t)peer];
	*recvPixel = plan->recvStart[(size_t)peer];
	*recvPixels = plan->recvStart[(size_t)peer + 1] - plan->recvStart[(size_t)peer];
	return CVX_OK;
}

int cvx_image_plan_tile_out(const cvx_image_plan *plan, void *localStore, uint64_t *tileOut)
{
	if (!plan || !tileOut) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	const int N = plan->worldSize;
	for (int b = 0; b < plan->frameCount; b++) {
		const int64_t first = plan->firstTile[(size_t)b], tiles = plan->firstTile[(size_t)b + 1] - first;
		for (int64_t c = 0; c < tiles; c++) {
			tileOut[first + c] = (c % N == plan->rank)
			                         ? (uint64_t)(uintptr_t)localStore + (uint64_t)((size_t)(plan->frames[(size_t)b].localSlotBase + c / N) * plan->tileStrideWords * 4)
			                         : 0;
		}
	}
	return CVX_OK;
}

static int ImageGatherLaunch(cvx_context *ctx, const cvx_image_plan *plan, void *hipStream, int unpack, const void *localStore, void *sendStream, const void *recvStream, void *images)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	// (`images` is written only for frames this rank displays: a rank that displays none -- fewer frames than ranks -- may pass NULL)
	if (!plan || (!images && plan->imagesMine > 0)) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "plan / images missing"); }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t st = hipStream ? (hipStream_t)hipStream : ctx->stream;
	hipLaunchKernelGGL(cvxk::image_gather_kernel, dim3((unsigned)plan->H, (unsigned)plan->frameCount), dim3(CVX_WAVE), 0, st, plan->devFrames, plan->worldSize, plan->rank, unpack,
	                   plan->devRowBase, static_cast<const uint32_t *>(localStore), plan->tileStrideWords, static_cast<uint32_t *>(sendStream),
	                   static_cast<const uint32_t *>(recvStream), static_cast<uint32_t *>(images));
	CVX_HIP(ctx, hipGetLastError());
	return CVX_OK;
}

int cvx_image_pack(cvx_context *ctx, const cvx_image_plan *plan, void *hipStream, const void *localStore, void *sendStream, void *images)
{
	if (plan && (!localStore || (plan->sendStart.back() > 0 && !sendStream))) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "local store / send stream missing"); }
	return ImageGatherLaunch(ctx, plan, hipStream, 0, localStore, sendStream, nullptr, images);
}

int cvx_image_unpack(cvx_context *ctx, const cvx_image_plan *plan, void *hipStream, const void *recvStream, void *images)
{
	if (plan && plan->recvStart.back() > 0 && !recvStream) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "receive stream missing"); }
	if (plan && plan->recvStart.back() == 0) { return CVX_OK; }
	return ImageGatherLaunch(ctx, plan, hipStream, 1, nullptr, nullptr, recvStream, images);
}

int cvx_raybuffer_device_ptr(cvx_context *ctx, int bufferIndex, int which, void **ptr, int64_t *bytes)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (ctx->poolTD.empty() || bufferIndex < 0 || bufferIndex >= ctx->bufferCount || (which != 0 && which != 1) || !ptr) {
		return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad buffer selection");
	}
	{ int rc = EnsurePools(ctx); if (rc != CVX_OK) { return rc; } }
	*ptr = which == 0 ? ctx->poolTD[(size_t)bufferIndex] : ctx->poolLR[(size_t)bufferIndex];
	if (bytes) { *bytes = (int64_t)(which == 0 ? ctx->poolBytesTD : ctx->poolBytesLR); }
	return CVX_OK;
}

int cvx_screen_device_ptr(cvx_context *ctx, void **ptr, int64_t *bytes)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!ctx->screen || !ptr) { return Fail(ctx, CVX_ERR_NOT_READY, "resolution not set"); }
	*ptr = ctx->screen;
	if (bytes) { *bytes = (int64_t)ctx->resX * ctx->resY * 4; }
	return CVX_OK;
}

int cvx_last_draw_ms(cvx_context *ctx, float *ms)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!ms) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "ms is NULL"); }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	int rc = FoldEvents(ctx);
	if (rc != CVX_OK) { return rc; }
	if (ctx->accumulatedDraws == 0) { return Fail(ctx, CVX_ERR_NOT_READY, "no timed draw yet"); }
	*ms = ctx->lastMs;
	return CVX_OK;
}

int cvx_draw_time_stats(cvx_context *ctx, double *totalMs, int *draws, int reset)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	int rc = FoldEvents(ctx);
	if (rc != CVX_OK) { return rc; }
	if (totalMs) { *totalMs = ctx->accumulatedMs; }
	if (draws) { *draws = ctx->accumulatedDraws; }
	if (reset) {
		ctx->accumulatedMs = 0.0;
		ctx->accumulatedDraws = 0;
	}
	return CVX_OK;
}

int cvx_enable_counters(cvx_context *ctx, int enable)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	ctx->countersEnabled = enable != 0;
	return CVX_OK;
}

int cvx_get_counters(cvx_context *ctx, cvx_counters *out)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!out) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "out is NULL"); }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	DevCounters c;
	CVX_HIP(ctx, hipMemcpyAsync(&c, ctx->devCounters, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	out->S = (int64_t)c.S;
	out->E = (int64_t)c.E;
	out->C = (int64_t)c.C;
	out->P = (int64_t)c.P;
	out->R = (int64_t)c.R;
	for (int i = 0; i < CVX_LOD_LEVELS; i++) { out->lodVisits[i] = (int64_t)c.lodVisits[i]; }
	return CVX_OK;
}

int cvx_get_raybuffer_layout(cvx_context *ctx, int which, cvx_raybuffer_layout *out)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!out || (which != 0 && which != 1) || ctx->poolTD.empty()) { return Fail(ctx, CVX_ERR_NOT_READY, "resolution not set"); }
	const int W = ctx->resX, H = ctx->resY;
	out->width = which == 0 ? H : W;
	out->rayCapacity = which == 0 ? W + 2 * H : 2 * W + H;
	out->tileRays = CVX_WAVE;
	out->tileCapacity = which == 0 ? ctx->tilesTD : ctx->tilesLR;
	out->tileBytes = (int64_t)out->width * CVX_WAVE * 4;
	return CVX_OK;
}

int cvx_copy_rows(cvx_context *ctx, void *hipStream, int toPacked, int64_t spanCount, const cvx_row_span *spansDevice, void *packedDevice)
{
	static_assert(sizeof(cvx_row_span) == sizeof(cvxk::RowSpan), "span layout");
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (ctx->poolTD.empty()) { return Fail(ctx, CVX_ERR_NOT_READY, "resolution not set"); }
	if (spanCount < 0 || (spanCount > 0 && (!spansDevice || !packedDevice))) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad span arguments"); }
	if (spanCount == 0) { return CVX_OK; }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	{ int rc = EnsurePools(ctx); if (rc != CVX_OK) { return rc; } }
	hipStream_t st = hipStream ? (hipStream_t)hipStream : ctx->stream;
	hipLaunchKernelGGL(cvxk::copy_rows_kernel, dim3((unsigned)spanCount), dim3(256), 0, st, reinterpret_cast<uint4 *>(ctx->poolBaseTD),
	                   reinterpret_cast<uint4 *>(ctx->poolBaseLR), static_cast<uint4 *>(packedDevice),
	                   reinterpret_cast<const cvxk::RowSpan *>(spansDevice), toPacked);
	CVX_HIP(ctx, hipGetLastError());
	return CVX_OK;
}

#if defined(CVX_EXPERIMENTS) || defined(CVX_PROFILE_SECTIONS) /* include/cpuvox_gpu_diag.h: not in the product library */
int cvx_debug_occupancy(cvx_context *ctx, int64_t ldsBytes, int *blocksPerCU)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!blocksPerCU) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "blocksPerCU is NULL"); }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	const int waveThreads = CVX_WAVE;
	CVX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, cvxk::render_kernel<false>, waveThreads, (size_t)ldsBytes));
	return CVX_OK;
}

int cvx_debug_section_cycles(cvx_context *ctx, uint64_t out[32], int reset)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!out) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "out is NULL"); }
#if defined(CVX_PROFILE_SECTIONS)
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	unsigned long long tmp[32];
	CVX_HIP(ctx, hipMemcpyFromSymbol(tmp, HIP_SYMBOL(cvxk::g_sectionCycles), sizeof tmp));
	for (int i = 0; i < 32; i++) { out[i] = tmp[i]; }
	if (reset) {
		std::memset(tmp, 0, sizeof tmp);
		CVX_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(cvxk::g_sectionCycles), tmp, sizeof tmp));
	}
	return CVX_OK;
#else
	(void)reset;
	for (int i = 0; i < 32; i++) { out[i] = 0; }
	return Fail(ctx, CVX_ERR_NOT_READY, "this is not the section-profile build of the library (make gpu-prof)");
#endif
}

int cvx_debug_section_histogram(cvx_context *ctx, uint64_t out[128], int reset)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!out) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "out is NULL"); }
#if defined(CVX_PROFILE_SECTIONS) && defined(CVX_PROFILE_COUNTS)
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	CVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	unsigned long long tmp[128];
	CVX_HIP(ctx, hipMemcpyFromSymbol(tmp, HIP_SYMBOL(cvxk::g_sectionHist), sizeof tmp));
	for (int i = 0; i < 128; i++) { out[i] = tmp[i]; }
	if (reset) {
		std::memset(tmp, 0, sizeof tmp);
		CVX_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(cvxk::g_sectionHist), tmp, sizeof tmp));
	}
	return CVX_OK;
#else
	(void)reset;
	for (int i = 0; i < 128; i++) { out[i] = 0; }
	return Fail(ctx, CVX_ERR_NOT_READY, "this is not the section-count build of the library (make gpu-count)");
#endif
}

int cvx_selftest_math(cvx_context *ctx, int op, int n, const float *a, const float *b, float *out)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (n <= 0 || !a || !b || !out) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	float *d = nullptr;
	const size_t bytes = (size_t)n * sizeof(float);
	CVX_HIP(ctx, hipMalloc((void **)&d, bytes * 3));
	int rc = CVX_OK;
	hipError_t e = hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) { e = hipMemcpyAsync(d + n, b, bytes, hipMemcpyHostToDevice, ctx->stream); }
	if (e == hipSuccess) {
		hipLaunchKernelGGL(cvxk::selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, n, d, d + n, d + 2 * (size_t)n);
		e = hipGetLastError();
	}
	if (e == hipSuccess) { e = hipMemcpyAsync(out, d + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost, ctx->stream); }
	if (e == hipSuccess) { e = hipStreamSynchronize(ctx->stream); }
	if (e != hipSuccess) { rc = Fail(ctx, CVX_ERR_HIP, "selftest failed: %s", hipGetErrorString(e)); }
	(void)hipFree(d);
	return rc;
}
#endif /* diagnostics */

} // extern "C"
