// cvx_context.h -- state and helpers shared by the translation units of libcpuvox_gpu.so (cvx_gpu.hip: context, raybuffers,
// draw / read-back / blit; cvx_world.hip: world validation, upload, LOD construction).  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "cpuvox_gpu.h"
#include "cvx_device.h"

struct RefHeader { // World.RLEColumn, World.cs:161-169
	int32_t storageOffset;
	uint16_t runCount;
	uint16_t worldMin;
	uint16_t worldMax;
};
static_assert(sizeof(RefHeader) == 12, "reference header is 12 bytes");

struct LastDraw { // what read-back / blit need to know about a buffer pair
	bool valid = false;
	cvx_segment_data segments[4];
	float vp[2];
	int tileBase[4];
	int width = 0, height = 0;
};


struct cvx_context {
	int device = 0;
	hipStream_t ownStream = nullptr;
	hipStream_t stream = nullptr;
	// one HIP event pair per draw since the last cvx_draw_time_stats(reset): kernel time on ctx->stream
	std::vector<hipEvent_t> evPairs; // 2 * pairs
	size_t evUsed = 0;               // pairs in use
	double accumulatedMs = 0.0;      // folded-in pairs
	int accumulatedDraws = 0;
	float lastMs = 0.f;
	std::string error;

	// world: every level is laid out on the host by cvx_world_upload (records, run list, element pool) and placed into ONE
	// device arena by the next draw (SyncWorld); levels that were not uploaded again are copied over from the old arena
	struct HostLevel {
		std::vector<uint4> records;     // 1 per column, row-major (cvx_device.h); empty once the level lives in the arena
		std::vector<uint2> runs;        // run list: every solid run of the columns whose record cannot hold them
		std::vector<uint2> counts;      // per column, what only the counting build reads
		std::vector<uint32_t> elements; // the columns' colours in blocks of 4 x 8 columns (cvx_device.h), a line of zeros on both sides
		size_t recordsBytes = 0, runsBytes = 0, countsBytes = 0, elementsBytes = 0;
		bool pending = false;           // host vectors hold data that is not in the arena yet
		int rowShift = 0;
		int colorShift = 7;             // log2 of the bytes between two colours of a column (cvx_device.h)
		int64_t solidColumns = 0, listedColumns = 0; // non-empty columns of the level, and how many of them keep their runs in the run list (code 0)
	};
	HostLevel hostLevel[CVX_LOD_LEVELS];
	uint8_t *arena = nullptr;
	size_t arenaBytes = 0;
	bool levelSet[CVX_LOD_LEVELS] = {};
	DevWorld hostWorld{};
	DevWorld *devWorld = nullptr;
	bool worldDirty = true;

	// raybuffers
	int resX = 0, resY = 0;
	int bufferCount = 2; // RenderManager.BUFFER_COUNT, RenderManager.cs:14
	int tilesTD = 0, tilesLR = 0;
	size_t poolBytesTD = 0, poolBytesLR = 0;
	std::vector<uint32_t *> poolTD, poolLR; // per buffer: offsets into one allocation each
	uint32_t *poolBaseTD = nullptr, *poolBaseLR = nullptr;
	bool poolsExternal = false;
	std::vector<LastDraw> last;
	uint32_t *screen = nullptr;
	uint32_t *screenBatch = nullptr; // cvx_blit_segments_batch without a caller buffer: screenBatchFrames images, grown on demand
	int screenBatchFrames = 0;
	void *blitParamsDev = nullptr;   // BlitParams of a batch (device) + their pinned staging copy
	void *blitParamsPinned = nullptr;
	int blitParamsCapacity = 0;
	uint32_t *staging = nullptr;
	size_t stagingBytes = 0;

	// per-draw scratch (grown on demand, reused)
	DevFrame *devFrames = nullptr;
	size_t devFramesCap = 0;
	DevTile *devTiles = nullptr;
	size_t devTilesCap = 0;
	struct UploadSlot {
		void *pinned = nullptr;
		size_t bytes = 0;
		hipEvent_t done = nullptr;
		bool inFlight = false;
	};
	static constexpr int kUploadSlots = 3;
	UploadSlot upload[kUploadSlots];
	unsigned uploadNext = 0;
	std::vector<DevFrame> hostFrames;
	std::vector<DevTile> hostTiles;
	std::vector<float> hostTileCost; // estimated DDA steps of the tile's middle ray (launch order: longest first)
	std::vector<int> hostTileWords;  // LDS mask words per lane the tile needs
	int maskWordsNeeded = 1;         // LDS mask words per lane of the widest [origMin, origMax] window in the current launch
	int ldsWordsNeeded = CVX_WAVE;   // LDS words (mask words x lanes) of the largest wave of the current launch
	int maxWaveMaskWords = 40 * CVX_WAVE; // LDS budget per wave: 10 KB = 16 waves per CU; wider tiles are cut into narrower waves
	bool maxWaveMaskWordsAuto = true;     // ... chosen per launch by DrawBatch's cost model unless CVX_MAX_WAVE_MASK_WORDS pins it

	int lonePixels = 0;              // lone_kernel: pixels of the longest window [origMin, origMax] of the launch (its LDS row)
	int launchLone = 0;              // the current launch goes to lone_kernel (cvx_lone.h): 1 = one mask register, 2 = two (windows of more than 2048 pixels)
#ifndef CVX_LONE_DEFAULT_MODE
#define CVX_LONE_DEFAULT_MODE 1
#endif
	int loneMode = CVX_LONE_DEFAULT_MODE; // 1: lone_kernel for launches of at most loneWaveBudget rays; 0 never, 2 always (experiment build: CVX_LONE=0 / 1; variants: -DCVX_LONE_DEFAULT_MODE)
	int loneWaveBudget = 12288;      // AUTO: launches of up to this many rays (= waves, tiles x 64) go to lone_kernel: 2 - 3 frames at 1080p; three quarters of it above
	                                 // 2560 x 1440 (4K: frames of up to ~9 000 rays).  Budget sweep over launches of 2 - 6 frames, per-pose crossover at 4K: profiles/r06_latency.md

	int shardIndex = 0, shardCount = 1;
	bool countersEnabled = false;
	DevCounters *devCounters = nullptr;
	int splitWaveBudget = 4096;                // DrawBatch cuts tiles into sub-tiles while the launch stays below this many waves
	int forcedSplit = 0;                       // CVX_TILE_SPLIT=1|2|...|64 (diagnostics): fixed split factor
	float tileCostPixelWeight = 0.f;           // launch-order estimate += weight * pixels of the tile's window (CVX_TILE_COST_PIXELS, diagnostics)
	bool tileCostMiddleRay = false;            // CVX_TILE_COST_MIDDLE_RAY=1 (diagnostics): estimate from the tile's middle ray instead of its longer edge ray
	int minMaskWords = 0;                      // CVX_MIN_MASK_WORDS (diagnostics): lower bound of the LDS mask words per lane, i.e. an occupancy cap
};

namespace cvxi {

// Records the message on the context (or for cvx_last_error(NULL) when ctx is null) and returns `code`.
int Fail(cvx_context *ctx, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

inline bool IsPow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// One column of a world blob in the reference's layout (World.cs:161-209): element range inside the pool, both guards
// present, positive run lengths that fit the column height, colours inside the pool.  Everything the kernels dereference
// later is covered here.  *solidRuns receives the number of solid runs, *colourCount (optional) the number of colours the runs address.
int ValidateColumn(cvx_context *ctx, int64_t i, const RefHeader &h, const uint32_t *elements, int64_t elementCount, int maxY, size_t *solidRuns, int64_t *colourCount = nullptr);

// cvx_lone.hip (its own translation unit: the latency kernel is compiled with its own optimisation level, Makefile): launches lone_kernel<hi> with one
// workgroup per ray (rays = 64 x tiles), `ldsBytes` of dynamic LDS (merge buffer + the ray's pixel row)
void LaunchLone(bool hi, unsigned rays, size_t ldsBytes, hipStream_t stream, const DevFrame *frames, const DevTile *tiles, const DevWorld *world);
} // namespace cvxi

#define CVX_HIP(ctx, call)                                                                                              \
	do {                                                                                                                \
		hipError_t e_ = (call);                                                                                         \
		if (e_ != hipSuccess) {                                                                                         \
			return cvxi::Fail(ctx, CVX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
		}                                                                                                               \
	} while (0)
