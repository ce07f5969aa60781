/*
 * cvx_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's Phase-1 raybuffer renderer
 * (pipliz/cpuvox, RenderManager.DrawSegments and the four Burst jobs it
 * schedules).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path
 * (cpuvox_amd/, libcpuvox_gpu.so) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path and its C#/Burst/Unity toolchain is
 * absent here, so this restatement cannot be checked against outputs of the
 * reference itself.  Fidelity is argued line by line: every function cites
 * the reference file:line it follows.
 *
 * Arithmetic contract: strict IEEE-754 binary32, no FMA contraction
 * (-ffp-contract=off), correctly rounded / and sqrt, round = half-to-even,
 * (int)float = x86 cvttss2si (out-of-range / NaN -> INT_MIN).
 */
#ifndef CVX_ORACLE_H
#define CVX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LOD_LEVELS 6 /* UnityManager.cs:42 */

/* RenderManager.SegmentData, RenderManager.cs:503-510 (36 bytes). */
typedef struct {
	float MinScreen[2];
	float MaxScreen[2];
	float CamLocalPlaneRayMin[2];
	float CamLocalPlaneRayMax[2];
	int32_t RayCount;
} orc_segment_data;

/* CameraData, CameraData.cs:11-16.  Matrix is column major (c0,c1,c2,c3) as
 * Unity.Mathematics.float4x4 stores it. */
typedef struct {
	float WorldToScreenMatrix[16];
	float PositionXZ[2];
	float PositionY;
	uint8_t InverseElementIterationDirection;
	uint8_t pad_[3];
	float FarClip;
	float LODDistances[ORC_LOD_LEVELS];
} orc_camera_data;

/* World (read side), World.cs:8-43.  storage points at the reference's raw
 * allocation: columnCount 12-byte RLEColumn headers followed by the 4-byte
 * RLEElement / ColorARGB32 pool (World.cs:285-293, 304-313). */
typedef struct {
	const void *storage;
	int32_t dimX, dimY, dimZ;
	int32_t lod;
	int32_t columnCount; /* World.ColumnCount: where the element pool starts */
} orc_world;

/* Algorithmic work counters:
 * B = 12*S + 4*E + 4*C + 4*P + 80*R. */
typedef struct {
	int64_t S; /* in-bounds GetVoxelColumn header fetches (DrawSegmentRayJob.cs:245) */
	int64_t E; /* RLE elements dereferenced in the element loop incl. guard (:444) */
	int64_t C; /* colour table reads (:531, :553, :560) */
	int64_t P; /* raybuffer pixels stored incl. skybox (:531, :600, :705, :714) */
	int64_t R; /* rays (RayContext items) */
	int64_t lodVisits[ORC_LOD_LEVELS]; /* S split per LOD level */
	int64_t continuations; /* rays that reached RenderJob */
} orc_counters;

/*
 * RenderManager.DrawSegments (RenderManager.cs:258-372) minus the Unity
 * texture upload.  Raybuffers are ray-major ARGB32 (bytes A,R,G,B per
 * pixel), 256-row partial textures laid out back to back
 * (RayBuffer.cs:121-128), i.e. ray r occupies pixels [r*width, (r+1)*width).
 * topDown: width = screenHeight, capacity screenWidth + 2*screenHeight rays;
 * leftRight: width = screenWidth, capacity 2*screenWidth + screenHeight
 * (RenderManager.cs:35-36).  threads <= 0 means all cores.
 * Returns the number of rays, or -1 on bad arguments.
 */
int orc_draw_segments(const orc_segment_data segments[4],
                      const orc_world worldLODs[ORC_LOD_LEVELS],
                      const orc_camera_data *camera,
                      int screenWidth, int screenHeight,
                      const float vanishingPointScreenSpace[2],
                      uint32_t *rayBufferTopDown,
                      uint32_t *rayBufferLeftRight,
                      int threads,
                      orc_counters *counters /* may be NULL */);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
