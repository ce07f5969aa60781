// cvx_device.h -- device-side data layout of libcpuvox_gpu (gfx950 only).
//
// World: ONE arena for all six LODs.  Per LOD a table of 32-byte column records in 8 x 8 tiles (a DDA step to a
// neighbouring column stays inside a 2 KB tile most of the time; World.GetIndexKnownInBounds' x-major order, World.cs:
// 145-149, puts x-neighbours dimZ * 32 bytes apart), built on upload from the reference's 12-byte headers (World.cs:161-169),
// an overflow list of solid runs for the few columns with more than two, and the element pool in the reference's own
// order [guard][run 0..n-1][guard][colour 0..s-1] (World.cs:163-165), from which the kernel reads the colours.  Everything
// is addressed with 32-bit byte offsets from the arena base (one scalar register pair for the whole wave; the arena is
// limited to 4 GiB).
//
// Raybuffer: tile-major.  A tile is 64 consecutive rays of one segment (one wavefront); inside a tile pixel y of lane l
// lives at (y*64 + l), so a wave storing the same pixel row writes 256 contiguous bytes.  Tiles of segment 0 then
// segment 1 fill the top-down pool, 2 then 3 the left-right pool.  The reference's ray-major rows (RayBuffer.cs:121-128)
// are reconstructed on read-back / in the blit.
#pragma once

#include <stdint.h>

#define CVX_WAVE 64
#define CVX_SKYBOX_ARGB 0x191919FFu /* ColorARGB32(25,25,25): bytes FF 19 19 19 (DrawSegmentRayJob.cs:702) */

struct DevWorldLevel {
	// One 32-byte record per column.  The reference walks every RLE element of a column (air runs only move the bounds,
	// World.cs:245-259), from the top down (ITERATION_DIRECTION +1) or from the bottom up (-1); the record holds the SOLID
	// runs only, top-down, each with its distance from the top of the column -- upload checks that the runs of a column add up
	// to the column height (the reference's builder always emits such columns, WordBuilder.cs:232-258), so the same numbers
	// give the positions the bottom-up walk accumulates, and one table serves both directions (the kernel walks it backwards):
	//   [0] = {colorsBase, solidCount | worldMin << 16, worldMax | runCount << 16, overflowBase}
	//         colorsBase   = element index (inside this level's pool) of the column's first colour (RLEColumn.ColorPointer, World.cs:185)
	//         overflowBase = index into this level's run list of solid run 2 (valid when solidCount > 2)
	//   [1] = solid runs 0 and 1 (top-down order), two words each:
	//         w0 = bottomY | (topY - 1) << 16      the run's span [bottomY, topY] in LOD-0 voxels (topY = dimY - (voxels of this LOD above the run << lod);
	//              round 5: rounds 1-4 stored start | length << 16 and the kernel shifted / subtracted them into these two numbers three times per drawn column)
	//         w1 = colorsIndex | elementIndex << 16   elementIndex = 1-based position of the run among ALL elements, top-down
	//              (only the counting variant reads it: it restores the reference's element count E)
	// Record of column (x, z) (LOD-0 coordinates): cx = x >> shift, cz = z >> shift; index = (cx << rowShift) + cz -- row-major, so that
	// a DDA step moves the record address by a per-ray constant (+- 32 << rowShift bytes along x, +- 32 along z) and the column loop adds
	// instead of recomputing it (round 2's 8 x 8 tiles cost 14 vector instructions per step and bought nothing: a 128-byte line holds four
	// z-neighbours either way).  A table is preceded and followed by at least one row + 64 bytes of arena that belong to nothing: a ray
	// that leaves the world fetches (and never looks at) the record one step outside.
	uint32_t recordsOff;  // byte offsets from DevWorld::arena
	uint32_t runsOff;     // uint2 per solid run k >= 2
	uint32_t elementsOff; // the reference's element pool (RLEElement / ColorARGB32); the kernel reads colours only
	int32_t shift;        // lod
	int32_t rowShift;     // log2 of the records per row (columns of this level along z)
	int32_t pad_;
};

struct DevWorld {
	const uint8_t *arena;
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
	int32_t lanes; // 0: the wave renders all 64 rays of the tile; else firstLane | laneCount << 8 | dupShift << 16: a sub-tile (small batches
	               // are split so that a frame alone still spreads over the chip, cvx_gpu.hip DrawBatch); 2^dupShift lanes work on every ray of it
	uint32_t *out; // where pixel row 0, lane 0 of this tile lives (pixel y of lane l at out[y*64 + l]); rows outside
	               // [origMin, origMax] are never touched, so `out` may point in front of the caller's slot
};

struct DevCounters { // cvx_counters on the device
	unsigned long long S, E, C, P, R;
	unsigned long long lodVisits[6];
};
