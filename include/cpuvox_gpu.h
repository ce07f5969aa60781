/*
 * cpuvox_gpu.h -- the drop-in boundary: C ABI of libcpuvox_gpu.so.
 *
 * The reference (pipliz/cpuvox) has no FFI for this path: the boundary is the
 * C# static method RenderManager.DrawSegments (Assets/Code/RenderManager.cs:
 * 258-372) which fills SegmentContext[4]/DrawContext, schedules the four
 * Burst jobs of Assets/Code/Rendering/DrawSegmentRayJob.cs and blocks on
 * render.Complete().  Every entry point below cites the reference interface
 * it replaces; INTEGRATION.md shows the P/Invoke binding a maintainer adds.
 *
 * Plain pointers and sizes only; all structs are blittable (C# sequential
 * layout, Pack = 4).  All functions return CVX_OK (0) or a negative error
 * code; cvx_last_error() gives the text.  The reference has no error returns
 * (Burst cannot throw; managed exceptions propagate to UnityManager.cs:184):
 * the P/Invoke wrapper turns a non-zero code into an exception.
 *
 * Threading: one caller thread per context (as the reference: Unity main
 * thread).  cvx_draw_segments is blocking like DrawSegments unless
 * CVX_DRAW_ASYNC is passed.
 *
 * Two kernels stand behind the draw calls and give the same raybuffers bit for
 * bit: the latency kernel for the reference's own call pattern, one blocking
 * frame at a time (one wavefront per ray), and the batch kernel for many
 * frames per launch (one lane per ray); see cvx_set_latency_kernel.
 */
#ifndef CPUVOX_GPU_H
#define CPUVOX_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVX_LOD_LEVELS 6 /* UnityManager.LOD_LEVELS, UnityManager.cs:42 */

enum {
	CVX_OK = 0,
	CVX_ERR_INVALID_ARGUMENT = -1,
	CVX_ERR_HIP = -2,          /* a HIP runtime call failed (no device, OOM, ...) */
	CVX_ERR_NOT_READY = -3,    /* world / resolution not set */
	CVX_ERR_CAPACITY = -4,     /* ray count exceeds the raybuffer capacity */
	CVX_ERR_TIMEOUT = -5,      /* a peer did not arrive in time (cvx_comm_create) */
};

enum {
	CVX_RAYBUFFER_TOPDOWN = 0,   /* RenderManager.rayBufferTopDown,   RenderManager.cs:16,36 */
	CVX_RAYBUFFER_LEFTRIGHT = 1, /* RenderManager.rayBufferLeftRight, RenderManager.cs:17,35 */
};

enum {
	CVX_DRAW_SYNC = 0,  /* return after the kernels completed (render.Complete(), :363) */
	CVX_DRAW_ASYNC = 1, /* enqueue only; cvx_synchronize() or a read-back completes it */
};

/* RenderManager.SegmentData, RenderManager.cs:503-510 (36 bytes). */
typedef struct cvx_segment_data {
	float MinScreen[2];
	float MaxScreen[2];
	float CamLocalPlaneRayMin[2];
	float CamLocalPlaneRayMax[2];
	int32_t RayCount;
} cvx_segment_data;

/* CameraData, CameraData.cs:11-16 (108 bytes).  WorldToScreenMatrix is
 * column major (c0,c1,c2,c3), the memory order of Unity.Mathematics.float4x4.
 * InverseElementIterationDirection is the C# bool (1 byte) + padding. */
typedef struct cvx_camera_data {
	float WorldToScreenMatrix[16];
	float PositionXZ[2];
	float PositionY;
	uint8_t InverseElementIterationDirection;
	uint8_t pad_[3];
	float FarClip;
	float LODDistances[CVX_LOD_LEVELS];
} cvx_camera_data;

/* Work counters of the last draw (optional instrumentation):
 * algorithmic bytes B = 12*S + 4*E + 4*C + 4*P + 80*R. */
typedef struct cvx_counters {
	int64_t S, E, C, P, R;
	int64_t lodVisits[CVX_LOD_LEVELS];
} cvx_counters;

typedef struct cvx_context cvx_context;

/* Lifetime of what RenderManager owns (ctor RenderManager.cs:25-41, Destroy :43-51).
 * device = HIP device ordinal. */
int cvx_create(int device, cvx_context **out);
void cvx_destroy(cvx_context *ctx);
const char *cvx_last_error(const cvx_context *ctx); /* ctx may be NULL: creation errors */

/* Use an externally owned HIP stream (hipStream_t as void*) for all work of
 * this context; NULL = the context's own stream. */
int cvx_set_stream(cvx_context *ctx, void *hipStream);

/*
 * Replaces `new World(dimensions, lod, data)` (World.cs:36-43) + the
 * `fixed (World* worldPtr = worldLODs)` hand-over (RenderManager.cs:155).
 * storage = WorldAllocator.GetStartPointer() (World.cs:273): columnCount
 * 12-byte RLEColumn headers {int32 offset; uint16 runCount, worldMin,
 * worldMax} followed by the 4-byte RLEElement/ColorARGB32 pool
 * (World.cs:161-169,245-259,304-313); byteLength = GetByteLength()
 * (World.cs:278-283), or the exact used length.  columnCount =
 * World.ColumnCount (World.cs:17) = where the element pool starts.
 * The data is copied to the device (and re-laid-out); the caller keeps
 * ownership of storage.  Dimensions must be powers of two (WordBuilder.cs:30),
 * X and Z at most 32768, Y at most 65536 (8192 x 8192 columns already fill the
 * 4 GiB of device tables the 32-bit offsets of the kernel address).
 * Per level: fewer than 2^30 pool entries (CVX_ERR_CAPACITY beyond); on the device the colours are kept in blocks
 * of 4 x 8 columns, each as deep as its tallest column -- ~2.5 x the colours of a terrain, at most 2^29 slots, and a level
 * whose blocks would take more than 4 x its colours keeps them column after column; all levels together within the 4 GiB arena.
 */
int cvx_world_upload(cvx_context *ctx, int lod, const void *storage, int64_t byteLength,
                     int dimX, int dimY, int dimZ, int columnCount);

/* RenderManager.SetResolution, RenderManager.cs:94-109: (re)allocates the
 * raybuffers: top-down = width H, capacity W+2H rays; left-right = width W,
 * capacity 2W+H rays (RenderManager.cs:35-36), times bufferCount. */
int cvx_set_resolution(cvx_context *ctx, int resolutionX, int resolutionY);

/* RenderManager.BUFFER_COUNT (RenderManager.cs:14) is 2; a GPU pipeline that
 * keeps many frames in flight may ask for more.  Call before cvx_set_resolution. */
int cvx_set_buffer_count(cvx_context *ctx, int bufferCount);

/*
 * RenderManager.DrawSegments, RenderManager.cs:258-372: same argument set
 * (segments[4], camera, screen size, vanishing point); worldLODs are the
 * uploaded ones; bufferIndex selects the raybuffer pair (RenderManager.
 * bufferIndex, :19,53-56).  Runs RaySetupJob, DDASetupJob,
 * TraceToFirstColumnJob and RenderJob (DrawSegmentRayJob.cs:12,49,87,156)
 * as HIP kernels.
 */
int cvx_draw_segments(cvx_context *ctx, const cvx_segment_data segments[4], const cvx_camera_data *camera,
                      int screenWidth, int screenHeight, const float vanishingPointScreenSpace[2],
                      int bufferIndex, int flags);

/* Many independent frames in one launch (frame i -> buffer (firstBufferIndex+i) % bufferCount).
 * segments: frameCount*4 entries; cameras, vanishingPoints (x,y pairs): frameCount entries. */
int cvx_draw_segments_batch(cvx_context *ctx, int frameCount, const cvx_segment_data *segments,
                            const cvx_camera_data *cameras, int screenWidth, int screenHeight,
                            const float *vanishingPoints, int firstBufferIndex, int flags);

/*
 * Same as cvx_draw_segments_batch, but the caller chooses where every 64-ray tile is written: tileOut[i] is the device
 * address of pixel row 0 / lane 0 of the i-th tile of the batch (canonical order: frame by frame, segment 0..3, tile
 * 0.. of the segment = ceil(RayCount / 64) tiles each); pixel y of lane l goes to ((uint32_t*)tileOut[i])[y*64 + l] and
 * only rows [origMin, origMax] of the tile's segment are ever written, so a slot needs (origMax - origMin + 1) * 256
 * bytes starting at tileOut[i] + origMin * 256.  tileOut[i] == 0: this context does not render the tile (another GPU
 * does).  Used by the multi-GPU path to render straight into send / display buffers (cpuvox_amd/dist.py); read-back
 * and blit do not apply to placed draws.  tileCount must equal the batch's tile count.
 */
int cvx_draw_segments_placed(cvx_context *ctx, int frameCount, const cvx_segment_data *segments,
                             const cvx_camera_data *cameras, int screenWidth, int screenHeight,
                             const float *vanishingPoints, int64_t tileCount, const uint64_t *tileOut, int flags);

/* Multi-GPU sharding: this context renders only the 64-ray
 * tiles t with t % shardCount == shardIndex.  Default (0, 1) = everything. */
int cvx_set_shard(cvx_context *ctx, int shardIndex, int shardCount);

/* Which kernel a draw goes to.  The reference's caller issues ONE blocking DrawSegments per frame (UnityManager.cs:182, RenderManager.cs:358-363): a
 * few thousand rays, far too few for the batch kernel (one lane per ray).  Such launches go to the latency kernel (one wavefront per RAY, its lanes the
 * ray's next 64 columns: csrc/cvx_lone.h); large batches go to the batch kernel.  Same raybuffers bit for bit either way.
 *   CVX_LATENCY_AUTO (default): the latency kernel for launches of at most ~12 000 rays (~8000 at 4K) whose pixel windows fit its mask (4096 pixels),
 *   CVX_LATENCY_NEVER / CVX_LATENCY_ALWAYS: pin the choice (ALWAYS still falls back to the batch kernel for windows of more than 4096 pixels and while the
 *   work counters are enabled: the counting variant exists for the batch kernel only). */
enum { CVX_LATENCY_AUTO = 0, CVX_LATENCY_NEVER = 1, CVX_LATENCY_ALWAYS = 2 };
int cvx_set_latency_kernel(cvx_context *ctx, int mode);

int cvx_synchronize(cvx_context *ctx);

/*
 * Multi-GPU frames behind the C ABI.  The reference's only synchronisation is `render.Complete()` (RenderManager.cs:358-363);
 * sharded over N GPUs (one process and one context per GPU) the equivalent is, per batch of frames:
 *     plan = cvx_shard_plan_create(frames, rank, N)             -- host arithmetic, identical on every rank
 *     cvx_shard_plan_tile_out(plan, sendBase, dispBase, out)    -- where this rank's tiles are rendered
 *     cvx_draw_segments_placed(ctx, ..., out, CVX_DRAW_ASYNC)   -- RaySetupJob .. RenderJob for this rank's tiles
 *     cvx_exchange(ctx, plan, comm, stream, sendBase, dispBase) -- grouped ncclSend / ncclRecv, one pair per peer
 * Tile t of frame b (canonical order: segment 0..3, tiles of a segment in ray order) is rendered by rank t % N; frame b is
 * displayed on rank b % N.  Two areas of 256-byte pixel rows (64 pixels of one tile row) per rank:
 *     send area     my tiles of frames displayed elsewhere, one section per destination rank, (frame, tile) order
 *     display area  all tiles of the frames I display, one section per rendering rank, (frame, tile) order -- my own section
 *                   is written by my kernel, the others arrive from the peers and are exactly their send sections for me
 * Only rows [origMin, origMax] of a tile exist in either area.  sendStart / dispStart: N + 1 section boundaries in rows.
 */
typedef struct cvx_shard_plan cvx_shard_plan;
int cvx_shard_plan_create(int frameCount, const cvx_segment_data *segments, const float *vanishingPoints, int screenWidth, int screenHeight,
                          int rank, int worldSize, cvx_shard_plan **out);
void cvx_shard_plan_destroy(cvx_shard_plan *plan);
int64_t cvx_shard_plan_tile_count(const cvx_shard_plan *plan);
int cvx_shard_plan_sections(const cvx_shard_plan *plan, int64_t *sendStart, int64_t *dispStart);
/* What travels between this rank and `peer` (what cvx_exchange sends and receives; a host with its own transport can use it directly):
 * rows [sendRow, sendRow + sendRows) of the send area go to peer, rows [recvRow, recvRow + recvRows) of the display area come from it. */
int cvx_shard_plan_transfer(const cvx_shard_plan *plan, int peer, int64_t *sendRow, int64_t *sendRows, int64_t *recvRow, int64_t *recvRows);
/* tileOut[i] for cvx_draw_segments_placed (0 = another rank renders tile i); sendBase / dispBase: device addresses of the two areas */
int cvx_shard_plan_tile_out(const cvx_shard_plan *plan, void *sendBase, void *dispBase, uint64_t *tileOut);
/* RCCL communicator owned by the library (librccl is loaded on first use): rank 0 makes the 128-byte id, the host passes it to
 * the other ranks over its own channel, every rank calls cvx_comm_create.  A communicator made elsewhere (ncclComm_t) works too. */
int cvx_comm_unique_id(void *id128);
int cvx_comm_create(cvx_context *ctx, const void *id128, int rank, int worldSize, void **comm);
/* ... with an explicit limit on how long to wait for the peers (cvx_comm_create waits 180 s): CVX_ERR_TIMEOUT when a rank of
 * the clique never arrives -- the sharded `render.Complete()` (RenderManager.cs:363) must not hang for ever on a dead peer.
 * After CVX_ERR_TIMEOUT the helper thread that called ncclCommInitRank is still parked inside RCCL (it cannot be cancelled) and
 * owns a half-made communicator: the process must report the failure and EXIT, not go on using the library. */
int cvx_comm_create_timeout(cvx_context *ctx, const void *id128, int rank, int worldSize, double timeoutSeconds, void **comm);
int cvx_comm_destroy(void *comm);
/* The exchange of one batch on hipStream (NULL = the context's stream): returns after enqueueing; order the consumer with the stream. */
int cvx_exchange(cvx_context *ctx, const cvx_shard_plan *plan, void *comm, void *hipStream, void *sendBase, void *dispBase);

/*
 * The other way of putting a sharded frame together: the IMAGE gather.  Every rank runs Phase 2
 * (RenderManager.BlitSegments, RenderManager.cs:199-256 + RayBufferBlit.shader:48-64) for the pixels whose ray lies in a tile it
 * rendered itself, and the display rank of a frame receives W * H pixels in total instead of the other ranks' raybuffer rows (a
 * raybuffer is ~2x the pixels of the screen it produces).  Per batch of frames:
 *     plan = cvx_image_plan_create(ctx, frames, rank, N)        -- counts pixels per rank on the device (one pass over the images)
 *     cvx_image_plan_tile_out(plan, localStore, out)            -- my tiles go into a compact local store (64 * max(W, H) pixels per tile)
 *     cvx_draw_segments_placed(ctx, ..., out, CVX_DRAW_ASYNC)
 *     cvx_image_pack(ctx, plan, stream, localStore, send, images)    -- my pixels: into the images I display, or my send stream
 *     cvx_image_exchange(ctx, plan, comm, stream, send, recv)        -- one ncclSend + one ncclRecv per peer
 *     cvx_image_unpack(ctx, plan, stream, recv, images)              -- the peers' pixels of the frames I display
 * Frame b is displayed by rank b % N as image b / N of `images` (W * H ARGB32 each, cvx_blit_segments' pixel rule); tile t of a frame
 * is rendered by rank t % N (as in the raybuffer gather).  The stream (src -> dst) holds the frames dst displays in frame order,
 * each with src's pixels in row-major order.  At most 8 ranks.
 */
typedef struct cvx_image_plan cvx_image_plan;
int cvx_image_plan_create(cvx_context *ctx, int frameCount, const cvx_segment_data *segments, const float *vanishingPoints, int screenWidth, int screenHeight,
                          int rank, int worldSize, cvx_image_plan **out);
void cvx_image_plan_destroy(cvx_image_plan *plan);
int64_t cvx_image_plan_tile_count(const cvx_image_plan *plan);
/* bytes of the local tile store, pixels (4 bytes each) of the send and receive streams, number of images this rank displays */
int cvx_image_plan_sizes(const cvx_image_plan *plan, int64_t *localStoreBytes, int64_t *sendPixels, int64_t *recvPixels, int32_t *imagesDisplayed);
int cvx_image_plan_transfer(const cvx_image_plan *plan, int peer, int64_t *sendPixel, int64_t *sendPixels, int64_t *recvPixel, int64_t *recvPixels);
int cvx_image_plan_tile_out(const cvx_image_plan *plan, void *localStore, uint64_t *tileOut);
/* (`images`: imagesDisplayed x H x W pixels; may be NULL on a rank that displays no frame of the batch, imagesDisplayed == 0) */
int cvx_image_pack(cvx_context *ctx, const cvx_image_plan *plan, void *hipStream, const void *localStore, void *sendStream, void *images);
int cvx_image_exchange(cvx_context *ctx, const cvx_image_plan *plan, void *comm, void *hipStream, void *sendStream, void *recvStream);
int cvx_image_unpack(cvx_context *ctx, const cvx_image_plan *plan, void *hipStream, const void *recvStream, void *images);

/* RenderManager.ClearRayBuffer, RenderManager.cs:58-92 (fills with one ARGB32
 * value, bytes A,R,G,B in memory order packed little-endian in `argb`). */
int cvx_clear_raybuffer(cvx_context *ctx, int bufferIndex, int which, uint32_t argb);

/*
 * Read back rows of a raybuffer in the reference's logical layout
 * (RayBuffer.Native.GetRayColumn, RayBuffer.cs:121-128): ray r of the buffer
 * = `width` contiguous ARGB32 pixels, rows [firstRay, firstRay+rayCount).
 * dst is host memory of rayCount*width*4 bytes.  Only rows rendered by the
 * last draw into that buffer are defined (others keep the cleared value).
 */
int cvx_read_raybuffer(cvx_context *ctx, int bufferIndex, int which, int firstRay, int rayCount, void *dst);

/*
 * RenderManager.BlitSegments (RenderManager.cs:199-256) + RayBufferBlit.shader
 * frag (Assets/Shaders/RayBufferBlit.shader:48-64): Phase 2, raybuffer ->
 * W x H ARGB32 screen image (row 0 = bottom row, Unity screen space).  Uses
 * the segments / vanishing point of the last draw into bufferIndex.
 * dstHost may be NULL (image stays on the device, see cvx_screen_device_ptr).
 */
int cvx_blit_segments(cvx_context *ctx, int bufferIndex, void *dstHost);

/*
 * Phase 2 for the frames of a batch in ONE launch: BlitSegments (RenderManager.cs:199-256) of the last draws into buffers
 * firstBufferIndex .. firstBufferIndex + frameCount - 1 (what cvx_draw_segments_batch rendered), image f into
 * dstDevice + f * W * H * 4 bytes (device memory; NULL = an array the context owns, grown on demand).  Asynchronous on
 * the context's stream; *imagesDevice (may be NULL) receives the address of image 0.  Same pixels as frameCount calls of
 * cvx_blit_segments.
 */
int cvx_blit_segments_batch(cvx_context *ctx, int firstBufferIndex, int frameCount, void *dstDevice, void **imagesDevice);

/* Use caller-owned device memory for the raybuffers (e.g. torch tensors that a RCCL collective
 * operates on).  Sizes: bufferCount * tileCapacity * tileBytes per kind (cvx_get_raybuffer_layout),
 * 256-byte aligned.  Buffer b of a kind starts at b * tileCapacity * tileBytes.  The context never
 * frees bound memory; call after cvx_set_resolution (which allocates internal ones). */
int cvx_bind_raybuffers(cvx_context *ctx, void *topDown, int64_t topDownBytes, void *leftRight, int64_t leftRightBytes);

/* Multi-GPU tile exchange helper: copies pixel rows (one row = 64 pixels of a tile = 256 bytes) between the
 * raybuffer pools of this context and a contiguous device staging buffer.  poolRow counts 256-byte rows from the start
 * of the pool (all buffers back to back): ((buffer * tileCapacity + tile) * width + pixelRow).  The span array lives in
 * device memory; hipStream NULL = the context's stream.  toPacked != 0: pool -> staging, else staging -> pool. */
typedef struct cvx_row_span {
	int64_t poolRow;
	int64_t packedRow;
	int32_t rows;
	int32_t kind; /* CVX_RAYBUFFER_TOPDOWN / CVX_RAYBUFFER_LEFTRIGHT */
} cvx_row_span;
int cvx_copy_rows(cvx_context *ctx, void *hipStream, int toPacked, int64_t spanCount, const cvx_row_span *spansDevice, void *packedDevice);

/* Device pointers for zero-copy consumers (RCCL gather, torch tensors). */
int cvx_raybuffer_device_ptr(cvx_context *ctx, int bufferIndex, int which, void **ptr, int64_t *bytes);
int cvx_screen_device_ptr(cvx_context *ctx, void **ptr, int64_t *bytes);

/* Timing of the last draw call, measured with HIP events on the context's
 * stream (milliseconds; kernels only, no host setup). */
int cvx_last_draw_ms(cvx_context *ctx, float *ms);
/* Sum and count of the kernel times of all draws since the last reset (each draw is bracketed by its own
 * HIP event pair on the context's stream; waits for pending draws). */
int cvx_draw_time_stats(cvx_context *ctx, double *totalMs, int *draws, int reset);

/* Enable in-kernel work counters (slower); read them after a SYNC draw. */
int cvx_enable_counters(cvx_context *ctx, int enable);
int cvx_get_counters(cvx_context *ctx, cvx_counters *out);

/* Device layout description of the tile-major raybuffer (for gathers/tests). */
typedef struct cvx_raybuffer_layout {
	int32_t width;         /* pixels per ray: H (top-down) or W (left-right) */
	int32_t rayCapacity;   /* W+2H or 2W+H */
	int32_t tileRays;      /* 64 */
	int32_t tileCapacity;  /* tiles allocated */
	int64_t tileBytes;     /* width * 64 * 4 */
} cvx_raybuffer_layout;
int cvx_get_raybuffer_layout(cvx_context *ctx, int which, cvx_raybuffer_layout *out);

/* World.DownSample(extraLods) (Assets/Code/World.cs:45-127: DownSampleColumn :71-96, DownSamplePartial :101-127, with
 * RLEColumnBuilder.ToFinalColumn WordBuilder.cs:181-268 and the RLEColumn constructor World.cs:190-234) as a device
 * kernel: builds LOD extraLods from the LOD 0 blob (same layout as cvx_world_upload takes; `lod` must be 0 -- the reference only
 * downsamples LOD 0, UnityManager.cs:328-331, other values are refused with CVX_ERR_INVALID_ARGUMENT) and returns the new
 * blob in the reference's storage layout, byte-identical to the host build (columns stored in index order), in memory
 * owned by the library: release it with cvx_free.  outColumnCount = World.ColumnCount of the new level (World.cs:17),
 * outVoxelCount (may be NULL) = voxels after deduplication, outDeviceMs (may be NULL) = device time of the two passes
 * and the offset scan. */
int cvx_world_downsample(cvx_context *ctx, const void *storage, int64_t byteLength, int dimX, int dimY, int dimZ, int lod, int columnCount, int extraLods,
                         void **outStorage, int64_t *outByteLength, int32_t *outColumnCount, int64_t *outVoxelCount, float *outDeviceMs);
/* UnityManager.cs:328-331 (`worldLODs[i] = worldLODs[0].DownSample(i)`): LOD 1..levelCount from the LOD 0 blob with one
 * validation and one upload of it.  outStorage / outByteLength / outColumnCount are arrays of levelCount entries ([i] = LOD i+1);
 * each blob is released with cvx_free.  outDeviceMs (may be NULL) = device time of the whole chain.  LOD 0 is read ONCE: level 1
 * is built from its colours, every further level (up to 7) from the level before it through exact per-voxel sums, which gives the
 * bytes of `DownSample(i)` applied to LOD 0 (integer averages of the LOD-0 voxels, the first inserted voxel's alpha); levels above 7
 * are built from LOD 0 directly like cvx_world_downsample does.  Device memory while it runs: per level 4 bytes x (the elements of the LOD 0
 * columns, counted column by column, + its columns) and two tables of per-voxel sums of 24 bytes x the same -- ~17 x the LOD 0 blob for five
 * levels, ~19 x for seven; when that does not fit (or passes 2^31 elements) the levels are built one by one from LOD 0, which needs ~2 x. */
int cvx_world_build_lods(cvx_context *ctx, const void *storage, int64_t byteLength, int dimX, int dimY, int dimZ, int columnCount, int levelCount,
                         void **outStorage, int64_t *outByteLength, int32_t *outColumnCount, float *outDeviceMs);
void cvx_free(void *p);

const char *cvx_version(void);

#ifdef __cplusplus
}
#endif
#endif
