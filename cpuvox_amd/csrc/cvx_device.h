// cvx_device.h -- device-side data layout of libcpuvox_gpu (gfx950 only).
//
// World: per LOD a dense array of 16-byte column headers (x-major, index
// (x>>lod)*(dimZ>>lod) + (z>>lod) like World.GetIndexKnownInBounds,
// World.cs:145-149) plus the element pool in the reference's own order
// [guard][run 0..n-1][guard][colour 0..s-1] (World.cs:163-165).  The 12-byte
// reference header (World.cs:161-169) is widened to 16 bytes on upload so a
// header is one aligned dwordx4 load.
//
// Raybuffer: tile-major.  A tile is 64 consecutive rays of one segment (one
// wavefront); inside a tile pixel y of lane l lives at (y*64 + l), so a wave
// storing the same pixel row writes 256 contiguous bytes.  Tiles of segment 0
// then segment 1 fill the top-down pool, 2 then 3 the left-right pool.  The
// reference's ray-major rows (RayBuffer.cs:121-128) are reconstructed on
// read-back / in the blit.
#pragma once

#include <stdint.h>

#define CVX_WAVE 64
#define CVX_SKYBOX_ARGB 0x191919FFu /* ColorARGB32(25,25,25): bytes FF 19 19 19 (DrawSegmentRayJob.cs:702) */

struct DevWorldLevel {
	// 32-byte column records, one table per element iteration direction:
	//   [0] = {elemOffset, runCount | worldMin << 16, worldMax, 0}
	//   [1] = the first four pool entries in walk order: entries 1..4 after the start guard (top-down walk,
	//         ITERATION_DIRECTION +1) or entries runCount..runCount-3 (bottom-up walk, -1)
	const uint4 *columnsDown;
	const uint4 *columnsUp;
	// entries 5..8 in walk order per column (read only for columns with more than 3 runs)
	const uint4 *extDown;
	const uint4 *extUp;
	const uint32_t *elements; // RLEElement {int16 ColorsIndex, int16 Length} / ColorARGB32
	int32_t shift;            // lod
	int32_t mulX;             // dimZ >> lod
};

struct DevWorld {
	DevWorldLevel level[6];
	int32_t dimX, dimY, dimZ;
	int32_t maskX, maskZ; // dimensionMaskXZ, World.cs:23
	int32_t pad_;
};

struct DevSegment { // SegmentContext (DrawSegmentRayJob.cs:718-727) as the kernel needs it
	float rayMinX, rayMinZ; // SegmentData.CamLocalPlaneRayMin
	float rayMaxX, rayMaxZ; // SegmentData.CamLocalPlaneRayMax
	int32_t rayCount;
	int32_t omin, omax;     // originalNextFreePixelMin / Max
	int32_t axisMappedToY;
	int32_t tileBase;       // first tile of this segment inside its pool
	int32_t colLen;         // pixels per ray: H (segments 0,1) or W (2,3)
	int32_t pad_[2];
};

struct DevFrame { // DrawContext + CameraData (DrawSegmentRayJob.cs:729-734, CameraData.cs:11-16)
	float M[16];
	float posX, posZ, posY, farClip;
	float lod[6];
	int32_t inverse;
	int32_t pad_;
	uint32_t *poolTD; // tile pools of the raybuffer pair this frame renders into
	uint32_t *poolLR;
	DevSegment seg[4];
};

struct DevTile { // one workgroup (= one wave) of the render kernel
	int32_t frame;
	int32_t seg;
	int32_t tileInSeg;
	int32_t pad_;
	uint32_t *out; // where pixel row 0, lane 0 of this tile lives (pixel y of lane l at out[y*64 + l]); rows outside
	               // [origMin, origMax] are never touched, so `out` may point in front of the caller's slot
};

struct DevCounters { // cvx_counters on the device
	unsigned long long S, E, C, P, R;
	unsigned long long lodVisits[6];
};
