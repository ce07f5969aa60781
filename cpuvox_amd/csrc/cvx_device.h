// cvx_device.h -- device-side data layout of libcpuvox_gpu (gfx950 only).
//
// World: per LOD and per walk direction a dense array of 32-byte column
// records (x-major, index (x>>lod)*(dimZ>>lod) + (z>>lod) like
// World.GetIndexKnownInBounds, World.cs:145-149) built on upload from the
// reference's 12-byte headers (World.cs:161-169), plus the element pool in the
// reference's own order [guard][run 0..n-1][guard][colour 0..s-1]
// (World.cs:163-165), from which the kernel reads the colours.
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
	// 32-byte column records, one table per element iteration direction (ITERATION_DIRECTION +1 walks a column
	// top-down, -1 bottom-up).  The reference walks every RLE element (air runs only move the bounds, World.cs:245-259);
	// the records hold the SOLID runs only, in walk order, with the position the walk would have reached:
	//   [0] = {colorsBase, solidCount | worldMin << 16, worldMax | runCount << 16, overflowBase}
	//         colorsBase   = pool index of the column's first colour (RLEColumn.ColorPointer, World.cs:185)
	//         overflowBase = index into runsDown / runsUp of solid run 2 (valid when solidCount > 2; even = 16-byte aligned)
	//   [1] = solid runs 0 and 1, two words each:
	//         w0 = start | length << 16   start = voxels (of this LOD) between the walk's starting end of the column and the run
	//         w1 = colorsIndex | elementIndex << 16   elementIndex = 1-based position of the run among ALL elements in walk order
	//              (only the counting variant reads it: it restores the reference's element count E)
	const uint4 *columnsDown;
	const uint4 *columnsUp;
	const uint2 *runsDown; // solid runs 2.. of the columns that have more than two
	const uint2 *runsUp;
	const uint32_t *elements; // the reference's element pool (RLEElement / ColorARGB32); the kernel reads colours only
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
	int32_t lanes; // 0: the wave renders all 64 rays of the tile; else firstLane | laneCount << 8: a sub-tile (small batches are
	               // split so that a frame alone still spreads over the chip, cvx_gpu.hip DrawBatch)
	uint32_t *out; // where pixel row 0, lane 0 of this tile lives (pixel y of lane l at out[y*64 + l]); rows outside
	               // [origMin, origMax] are never touched, so `out` may point in front of the caller's slot
};

struct DevCounters { // cvx_counters on the device
	unsigned long long S, E, C, P, R;
	unsigned long long lodVisits[6];
};
