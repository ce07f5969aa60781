// cvx_device.h -- device-side data layout of libcpuvox_gpu (gfx950 only).
//
// World: ONE arena for all six LODs.  Per LOD a row-major table of 16-byte column records (World.GetIndexKnownInBounds' x-major order, World.cs:
// 145-149), built on upload from the reference's 12-byte headers (World.cs:161-169),
// a run list for the few columns whose record cannot hold their runs, and the columns' colours in blocks of 4 x 8 columns (the reference's
// pool interleaves them with the RLE elements, [guard][run 0..n-1][guard][colour 0..s-1], World.cs:163-165: the kernel never reads those).  Everything
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
// Colour blocks (see DevWorldLevel::elementsOff): 4 x 8 columns, colour k of a column CVX_COLOR_STRIDE entries behind its colour k - 1
#ifndef CVX_COLOR_BLOCK_X
#define CVX_COLOR_BLOCK_X 4
#define CVX_COLOR_BLOCK_Z 8
#endif
#define CVX_COLOR_STRIDE (CVX_COLOR_BLOCK_X * CVX_COLOR_BLOCK_Z)
#define CVX_SKYBOX_ARGB 0x191919FFu /* ColorARGB32(25,25,25): bytes FF 19 19 19 (DrawSegmentRayJob.cs:702) */

struct DevWorldLevel {
	// One 16-byte record per column (round 5; rounds 1-4: 32 bytes -- the column's header and its first two solid runs with everything the
	// counting build wants beside them.  Three quarters of the columns of a terrain hold ONE solid run whose span is the header's
	// [worldMin, worldMax], nearly all others two or three: the record now keeps what the RENDERING build reads, a 128-byte line holds eight
	// columns instead of four, and the column loop fetches one dwordx4 per step instead of two).
	// The reference walks every RLE element of a column (air runs only move the bounds, World.cs:245-259), from the top down
	// (ITERATION_DIRECTION +1) or from the bottom up (-1); the kernel iterates the SOLID runs only, top-down numbering, each as its span
	// [bottomY, topY] in LOD-0 voxels (topY = dimY - (voxels of this LOD above the run << lod); upload checks that the runs of a column add
	// up to the column height -- the reference's builder always emits such columns, WordBuilder.cs:232-258 -- so the same numbers are what
	// the bottom-up walk accumulates).
	//   x = code << 30 | colorsBase      colorsBase = slot (inside this level's colour array) of the column's first colour (RLEColumn.ColorPointer, World.cs:185; >= 32)
	//   y = worldMin | worldMax << 16    RLEColumn.WorldMin / WorldMax as the blob has them (World.cs:161-169): what the cull of every column step reads
	//   x == 0: RunCount == 0 (the empty column, all four words 0)
	//   code 1 .. 3 = the number of solid runs, for a column the builder's invariants hold for -- the top run ends at worldMax, the lowest one stands on
	//           worldMin, and every ColorsIndex is the sum of the lengths of the solid runs above it (WordBuilder.cs:181-268 emits nothing else; the
	//           kernel derives the index from the spans).  run 0 = [w.lo, worldMax], run 1 = [z.lo, w.hi + 1], run 2 = [worldMin, z.hi + 1]:
	//           one run:  w = y;   two: w = bottom0 | (top1 - 1) << 16, z = worldMin;   three: w likewise, z = bottom1 | (top2 - 1) << 16
	//   code 0, x != 0: anything else ("listed": more runs, no solid run, bounds or colour indices of another shape): z = index into this level's run
	//           list of the column's block (an even index: 16-byte aligned), w = solidCount.  A block holds ALL solid runs of the column, two words each:
	//           w0 = bottomY | (topY - 1) << 16,  w1 = colorsIndex | elementIndex << 16  (elementIndex = 1-based position of the run among ALL
	//           elements, top-down; only the counting variant reads it).  The kernel may read two entries at any block, also an empty one.
	// `counts` (only the counting variant reads it, it restores the reference's element count E): per column {RunCount | elementIndex of solid
	// run 0 << 16, elementIndex of run 1 | elementIndex of run 2 << 16}.
	// Record of column (x, z) (LOD-0 coordinates): cx = x >> shift, cz = z >> shift; index = (cx << rowShift) + cz -- row-major, so that
	// a DDA step moves the record address by a per-ray constant (+- 16 << rowShift bytes along x, +- 16 along z) and the column loop adds
	// instead of recomputing it (round 2's 8 x 8 tiles cost 14 vector instructions per step and bought nothing).  A table is preceded and
	// followed by at least one row + 64 bytes of arena that belong to nothing: a ray that leaves the world fetches (and never looks at) the
	// record one step outside.
	uint32_t recordsOff;  // byte offsets from DevWorld::arena
	uint32_t runsOff;     // run list: uint2 per solid run of the listed columns
	// The columns' colours (ColorARGB32; the reference's pool interleaves them with the RLE elements, which the kernel never reads), in blocks of 4 x 8
	// columns: colour k of the block's 32 columns fills ONE 128-byte line, colour k of a column lives CVX_COLOR_STRIDE entries behind its colour k - 1,
	// and a block is as deep as its column with the most colours (~2.5 x the colours of a terrain; a level whose blocks would take more than 4 x keeps
	// its colours column after column: `colorShift`).  What the rays of a wave read at a step are the first few colours (the top voxels)
	// of neighbouring columns: one line for all of them, where the column-after-column order spent a line on three columns' full stacks.
	uint32_t elementsOff;
	int32_t shift;        // lod
	int32_t rowShift;     // log2 of the records per row (columns of this level along z)
	uint32_t countsOff;   // uint2 per column, indexed like the records
	int32_t colorShift;   // log2 of the bytes between two colours of a column: 7 (blocks) or 2 (column after column, for worlds whose blocks would waste > 4 x)
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
