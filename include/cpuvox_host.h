/*
 * cpuvox_host.h -- C ABI of libcpuvox_host.so: the host-side (CPU) mirror of
 * the reference's managed code around the DrawSegments boundary, written in
 * C++ because no C#/.NET toolchain exists in this image (the reference's host
 * language).  Same names, argument meaning and error behaviour as the
 * reference classes:
 *   world set      <- World[] worldLODs + WorldBuilder + WorldSaveFile
 *                     (Assets/Code/World.cs, WordBuilder.cs, WorldSaveFile.cs,
 *                      UnityManager.cs:297-343 "Convert", :245-251 "Load")
 *   camera / frame <- UnityManager.LateUpdate + RenderManager.DrawWorld setup
 *                     (UnityManager.cs:163-201,417-458; RenderManager.cs:111-152,374-510)
 *   render manager <- RenderManager (RenderManager.cs:12-256), bound to
 *                     libcpuvox_gpu.so for DrawSegments / BlitSegments.
 * No GPU is needed for the world / frame functions.
 */
#ifndef CPUVOX_HOST_H
#define CPUVOX_HOST_H

#include <stdint.h>

#include "cpuvox_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cvxh_world_set cvxh_world_set; /* World[LOD_LEVELS], UnityManager.cs:15,55 */

typedef struct cvxh_world_info {
	const void *storage;  /* WorldAllocator.GetStartPointer(), World.cs:273 */
	int64_t byteLength;   /* used bytes: headers + elements */
	int32_t dimX, dimY, dimZ;
	int32_t lod;
	int32_t columnCount;  /* World.ColumnCount, World.cs:17 */
	int64_t elementCount; /* 4-byte elements allocated */
} cvxh_world_info;

/* Thread-local text of the last failure of a cvxh_* call on this thread. */
const char *cvxh_last_error(void);

/* UnityManager "Convert": ObjModel.Import + SimpleMesh.Rescale + WorldBuilder.Import
 * + ToLOD0World + DownSample(1..5) (UnityManager.cs:297-343). */
int cvxh_world_from_obj(const char *path, int maxDimension, int swapYZ, int flipX, int flipY, int flipZ,
                        int threads, cvxh_world_set **out);
/* Seeded procedural heightmap world with the full LOD chain (benchmark configs 3-5). */
int cvxh_world_procedural(int dimX, int dimY, int dimZ, uint32_t seed, int threads, cvxh_world_set **out);
/* WorldSaveFile.Deserialize / Serialize, WorldSaveFile.cs:57,8 */
int cvxh_world_load(const char *path, cvxh_world_set **out);
int cvxh_world_save(const cvxh_world_set *worlds, const char *path);
/* Assemble a world set from storage blobs in the reference's layout, blob i = LOD i (what WorldSaveFile.Deserialize does
 * per world, WorldSaveFile.cs:86-92; e.g. LOD 0 from the host build and LOD 1.. from cvx_world_downsample).  Copies. */
int cvxh_world_from_blobs(int dimX, int dimY, int dimZ, int count, const void *const *blobs, const int64_t *byteLengths, cvxh_world_set **out);
/* World.DownSample(extraLods) of LOD 0 on the host (World.cs:45), result discarded: returns the wall-clock seconds it took
 * with `threads` worker threads (<= 0: all) -- the CPU side of the comparison with cvx_world_downsample. */
int cvxh_world_downsample_seconds(const cvxh_world_set *worlds, int extraLods, int threads, double *outSeconds, int64_t *outVoxelCount);
void cvxh_world_free(cvxh_world_set *worlds);
int cvxh_world_lod_count(const cvxh_world_set *worlds);
int cvxh_world_info_get(const cvxh_world_set *worlds, int lod, cvxh_world_info *out);
int64_t cvxh_world_lod0_voxels(const cvxh_world_set *worlds);

/* WorldBuilder (WordBuilder.cs:14-130) for explicit voxel lists: x,y,z,argb arrays of n entries
 * (argb = bytes A,R,G,B in memory order, little-endian packed). */
typedef struct cvxh_world_builder cvxh_world_builder;
int cvxh_world_builder_create(int dimX, int dimY, int dimZ, cvxh_world_builder **out);
int cvxh_world_builder_set_voxels(cvxh_world_builder *b, int64_t n, const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *argb);
/* ToLOD0World + DownSample(1..5); consumes the builder's voxels. */
int cvxh_world_builder_finish(cvxh_world_builder *b, int threads, cvxh_world_set **out);
void cvxh_world_builder_free(cvxh_world_builder *b);

/* Camera pose as the Unity scene holds it (transform + Camera component). */
typedef struct cvxh_camera_pose {
	float position[3];
	float eulerAngles[3];   /* degrees, Unity order */
	float fieldOfView;      /* vertical degrees; scene default 85 */
	float nearClipPlane;    /* scene default 0.05 */
	int32_t pixelWidth, pixelHeight;
} cvxh_camera_pose;

/* UnityManager.SetupLods, UnityManager.cs:417-458 (also yields farClip = 2*maxDim). */
int cvxh_setup_lods(const cvxh_camera_pose *pose, int worldMaxDimension, int resolutionX, int resolutionY,
                    float lodError, float outLODDistances[CVX_LOD_LEVELS], float *outFarClip);

typedef struct cvxh_frame {
	cvx_segment_data segments[4];
	cvx_camera_data camera;
	float vanishingPointScreenSpace[2];
	float vanishingPointWorldSpace[3];
	float forward[3];
	int32_t totalRays;
} cvxh_frame;

/* LimitRotationHorizon (UnityManager.cs:193-201, when limitHorizon != 0) followed by the
 * DrawWorld setup up to the DrawSegments call (RenderManager.cs:119-152). */
int cvxh_setup_frame(const cvxh_camera_pose *pose, int limitHorizon, float farClip,
                     const float LODDistances[CVX_LOD_LEVELS], int screenWidth, int screenHeight,
                     int worldDimensionY, cvxh_frame *out);

/*
 * RenderManager twin (Assets/Code/RenderManager.cs:12-256): SetResolution :94, SwapBuffers :53, ClearRayBuffer :58,
 * DrawWorld :111 (vanishing point + segments + CameraData on the host, DrawSegments and BlitSegments on the GPU
 * through libcpuvox_gpu.so, loaded from gpuLibraryPath; NULL = the file of that name next to libcpuvox_host.so).  Fails (no CPU fallback) when the library or a HIP device
 * is missing.  renderMode: 0 ScreenBuffer, 1 RayBufferTopDown, 2 RayBufferLeftRight (UnityManager.ERenderMode).
 */
typedef struct cvxh_render_manager cvxh_render_manager;
int cvxh_render_manager_create(int device, int screenWidth, int screenHeight, const char *gpuLibraryPath, cvxh_render_manager **out);
void cvxh_render_manager_destroy(cvxh_render_manager *rm);
int cvxh_render_manager_upload_world(cvxh_render_manager *rm, const cvxh_world_set *worlds);
int cvxh_render_manager_set_resolution(cvxh_render_manager *rm, int resolutionX, int resolutionY, int *changed);
int cvxh_render_manager_swap_buffers(cvxh_render_manager *rm); /* returns the new buffer index */
int cvxh_render_manager_clear_raybuffer(cvxh_render_manager *rm, int renderMode);
/* UnityManager.LateUpdate body (UnityManager.cs:179-182): LimitRotationHorizon + DrawWorld.  screenArgb32: W*H
 * pixels, row 0 = bottom, may be NULL; outFrame (may be NULL) receives the frame setup that was used. */
int cvxh_render_manager_draw_world(cvxh_render_manager *rm, const cvxh_camera_pose *pose, int limitHorizon, float farClip,
                                   const float LODDistances[CVX_LOD_LEVELS], uint32_t *screenArgb32, cvxh_frame *outFrame);
int cvxh_render_manager_read_raybuffer(cvxh_render_manager *rm, int which, int firstRay, int rayCount, uint32_t *dst);

/* BenchmarkPath.anim at clip time t in [0, 1.15], position scaled by world dims (UnityManager.cs:86-87). */
void cvxh_sample_benchmark_path(float t, const float worldDims[3], float outPosition[3], float outEuler[3]);

/* Worker threads the library starts when a `threads` argument is <= 0: the OpenMP default capped by the control group's
 * CPU quota (cgroup v2 cpu.max). */
int cvxh_default_threads(void);

/* Texture2D.LoadImage + GetPixels32 as ObjModel uses them for map_Kd textures (SimpleMesh.cs:186-205): decodes a PNG / JPEG / TGA / PPM
 * file into width * height RGBA8 pixels, row 0 = BOTTOM row.  Call with rgba = NULL to get the size, then again with a buffer of
 * capacityBytes >= width * height * 4. */
int cvxh_image_load(const char *path, int32_t *width, int32_t *height, uint8_t *rgba, int64_t capacityBytes);

const char *cvxh_version(void);

#ifdef __cplusplus
}
#endif
#endif
