/*
 * cvx_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See cvx_oracle.h for the contract.  PARITY UNPINNED (no reference vectors).
 *
 * Restates, in the reference's own structure (four jobs, byte-per-pixel seen
 * cache, ray-major raybuffer rows):
 *   Assets/Code/Rendering/DrawSegmentRayJob.cs   (all)
 *   Assets/Code/Utils/SegmentDDAData.cs          (all)
 *   Assets/Code/Utils/CameraData.cs:39-163
 *   Assets/Code/World.cs:130-149,161-188,245-259,285-293
 *   Assets/Code/Rendering/RayBuffer.cs:121-128
 *   Assets/Code/RenderManager.cs:258-372
 * Third-party arithmetic (com.unity.mathematics 1.2.6, not vendored in the
 * reference) is restated from its published definitions in the m_* helpers.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -fopenmp
 */
#include "cvx_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* Unity.Mathematics scalar semantics (math.cs of 1.2.6)               */
/* ------------------------------------------------------------------ */
typedef struct { float x, y; } f2;
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef struct { int x, y; } i2;

/* math.min/max(float,float): "float.IsNaN(y) || x < y ? x : y". */
static inline float m_min(float x, float y) { return (y != y || x < y) ? x : y; }
static inline float m_max(float x, float y) { return (y != y || x > y) ? x : y; }
static inline int m_mini(int x, int y) { return x < y ? x : y; }
static inline int m_maxi(int x, int y) { return x > y ? x : y; }
static inline int m_clampi(int x, int a, int b) { return m_maxi(a, m_mini(b, x)); }
static inline float m_lerp(float a, float b, float t) { return a + t * (b - a); }
static inline float m_unlerp(float a, float b, float x) { return (x - a) / (b - a); }
static inline float m_select(float a, float b, int c) { return c ? b : a; }
static inline float m_sign(float x) { return (x > 0.0f ? 1.0f : 0.0f) - (x < 0.0f ? 1.0f : 0.0f); }
static inline float m_frac(float x) { return x - floorf(x); }
static inline float m_round(float x) { return nearbyintf(x); } /* Math.Round: half to even */
static inline float m_cmax(f2 v) { return m_max(v.x, v.y); }
static inline float m_cmin(f2 v) { return m_min(v.x, v.y); }

/* C# (int)float under Burst/x86 = cvttss2si: NaN / out of range -> INT_MIN
 * (SURVEY.md Appendix A.10). */
static inline int f2i(float x)
{
	if (x != x || x >= 2147483648.0f || x < -2147483648.0f) {
		return INT_MIN;
	}
	return (int)x;
}

static inline f3 f3_add(f3 a, f3 b) { f3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline f3 f3_sub(f3 a, f3 b) { f3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline f3 f3_muls(f3 a, float s) { f3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline f3 f3_lerp(f3 a, f3 b, float t) { return f3_add(a, f3_muls(f3_sub(b, a), t)); } /* a + t*(b-a) */

/* ------------------------------------------------------------------ */
/* World read side, World.cs                                           */
/* ------------------------------------------------------------------ */
typedef struct { /* World.RLEColumn, World.cs:161-169 (12 bytes) */
	int32_t storageOffset;
	uint16_t runCount;
	uint16_t worldMin;
	uint16_t worldMax;
} rle_column;

typedef struct { /* World.RLEElement, World.cs:245-259 */
	int16_t ColorsIndex;
	int16_t Length;
} rle_element;

typedef struct {
	const rle_column *columns;   /* WorldAllocator.pointer */
	const rle_element *elements; /* WorldAllocator.elementsStart, World.cs:310 */
	i2 dimensionMaskXZ;
	int dimX, dimY, dimZ;
	int lod;
	int indexingMulX;
} world_t;

/* World.GetIndexKnownInBounds, World.cs:145-149 */
static inline int world_index(const world_t *w, i2 p)
{
	return (p.x >> w->lod) * w->indexingMulX + (p.y >> w->lod);
}

/* World.GetVoxelColumn, World.cs:130-142 (REPEAT_WORLD == false) */
static inline int world_get_voxel_column(const world_t *w, i2 position, rle_column *column)
{
	i2 inb = { position.x & w->dimensionMaskXZ.x, position.y & w->dimensionMaskXZ.y };
	if (inb.x != position.x || inb.y != position.y) {
		return -1;
	}
	*column = w->columns[world_index(w, position)];
	return column->runCount;
}

/* ------------------------------------------------------------------ */
/* SegmentDDAData, SegmentDDAData.cs                                   */
/* ------------------------------------------------------------------ */
typedef struct {
	i2 position;
	i2 step;
	f2 start, dir, tDelta, tMax;
	f2 intersectionDistances; /* x = last, y = next */
} dda_t;

/* SegmentDDAData ctor, SegmentDDAData.cs:17-28 */
static dda_t dda_new(f2 start, f2 dir)
{
	dda_t d;
	d.start = start;
	d.dir = dir;
	d.position.x = f2i(floorf(start.x));
	d.position.y = f2i(floorf(start.y));
	d.tDelta.x = 1.0f / m_max(0.0000001f, fabsf(dir.x));
	d.tDelta.y = 1.0f / m_max(0.0000001f, fabsf(dir.y));
	f2 signDir = { m_sign(dir.x), m_sign(dir.y) };
	d.step.x = f2i(signDir.x);
	d.step.y = f2i(signDir.y);
	d.tMax.x = (signDir.x * -m_frac(start.x) + (signDir.x * 0.5f) + 0.5f) * d.tDelta.x;
	d.tMax.y = (signDir.y * -m_frac(start.y) + (signDir.y * 0.5f) + 0.5f) * d.tDelta.y;
	f2 prev = { d.tMax.x - d.tDelta.x, d.tMax.y - d.tDelta.y };
	d.intersectionDistances.x = m_cmax(prev);
	d.intersectionDistances.y = m_cmin(d.tMax);
	return d;
}

/* SegmentDDAData.NextLOD, SegmentDDAData.cs:31-73 */
static void dda_next_lod(dda_t *d, int currentVoxelSize)
{
	i2 remainders = { d->position.x & (currentVoxelSize * 2 - 1), d->position.y & (currentVoxelSize * 2 - 1) };
	f2 tMaxPrevious = { d->tMax.x - d->tDelta.x, d->tMax.y - d->tDelta.y };

	if (d->dir.x >= 0.0f) {
		if (remainders.x < currentVoxelSize) {
			d->tMax.x += d->tDelta.x;
		} else {
			tMaxPrevious.x -= d->tDelta.x;
		}
	} else {
		if (remainders.x < currentVoxelSize) {
			tMaxPrevious.x -= d->tDelta.x;
		} else {
			d->tMax.x += d->tDelta.x;
		}
	}

	if (d->dir.y >= 0.0f) {
		if (remainders.y < currentVoxelSize) {
			d->tMax.y += d->tDelta.y;
		} else {
			tMaxPrevious.y -= d->tDelta.y;
		}
	} else {
		if (remainders.y < currentVoxelSize) {
			tMaxPrevious.y -= d->tDelta.y;
		} else {
			d->tMax.y += d->tDelta.y;
		}
	}

	d->intersectionDistances.x = m_cmax(tMaxPrevious);
	d->intersectionDistances.y = m_cmin(d->tMax);
	d->position.x -= remainders.x;
	d->position.y -= remainders.y;
	d->tDelta.x *= 2.0f;
	d->tDelta.y *= 2.0f;
	d->step.x *= 2;
	d->step.y *= 2;
}

/* SegmentDDAData.StepToWorldIntersection, SegmentDDAData.cs:75-130 */
static int dda_step_to_world_intersection(dda_t *d, f2 dimensions)
{
	f2 inverseDir = { 1.0f / d->dir.x, 1.0f / d->dir.y };
	f2 tmin = { -INFINITY, -INFINITY };
	f2 tmax = { INFINITY, INFINITY };

	if (d->dir.x != 0.0f) {
		float tx1 = -d->start.x * inverseDir.x;
		float tx2 = (dimensions.x - d->start.x) * inverseDir.x;
		tmin.x = m_min(tx1, tx2);
		tmax.x = m_max(tx1, tx2);
	}
	if (d->dir.y != 0.0f) {
		float ty1 = -d->start.y * inverseDir.y;
		float ty2 = (dimensions.y - d->start.y) * inverseDir.y;
		tmin.y = m_min(ty1, ty2);
		tmax.y = m_max(ty1, ty2);
	}

	float tmint = m_cmax(tmin);
	float tmaxt = m_cmin(tmax);

	if (tmaxt < tmint || tmint <= 0.0f) {
		return 0;
	}

	f2 tLast;
	if (tmin.x < tmin.y && tmin.x != -INFINITY) {
		tLast.y = tmin.y;
		float offsetAxisToHit = tmint * d->dir.x;
		float hitPosition = d->start.x + offsetAxisToHit;
		hitPosition = d->dir.x > 0.0f ? floorf(hitPosition) : ceilf(hitPosition);
		offsetAxisToHit = hitPosition - d->start.x;
		tLast.x = offsetAxisToHit / d->dir.x;
	} else {
		tLast.x = tmin.x;
		float offsetAxisToHit = tmint * d->dir.y;
		float hitPosition = d->start.y + offsetAxisToHit;
		hitPosition = d->dir.y > 0.0f ? floorf(hitPosition) : ceilf(hitPosition);
		offsetAxisToHit = hitPosition - d->start.y;
		tLast.y = offsetAxisToHit / d->dir.y;
	}

	d->tMax.x = tLast.x + d->tDelta.x;
	d->tMax.y = tLast.y + d->tDelta.y;
	d->intersectionDistances.x = m_cmax(tLast);
	d->intersectionDistances.y = m_cmin(d->tMax);
	float mid = m_lerp(d->intersectionDistances.x, d->intersectionDistances.y, 0.5f);
	d->position.x = f2i(floorf(d->start.x + mid * d->dir.x));
	d->position.y = f2i(floorf(d->start.y + mid * d->dir.y));
	return 1;
}

/* SegmentDDAData.Step, SegmentDDAData.cs:135-150: true when far clip is hit */
static inline int dda_step(dda_t *d, float farclip)
{
	float crossedBoundaryDistance;
	if (d->tMax.x < d->tMax.y) {
		crossedBoundaryDistance = d->tMax.x;
		d->tMax.x += d->tDelta.x;
		d->position.x += d->step.x;
	} else {
		crossedBoundaryDistance = d->tMax.y;
		d->tMax.y += d->tDelta.y;
		d->position.y += d->step.y;
	}
	d->intersectionDistances.x = crossedBoundaryDistance;
	d->intersectionDistances.y = m_cmin(d->tMax);
	return crossedBoundaryDistance >= farclip;
}

/* SegmentDDAData.IsBeyondFarClip, SegmentDDAData.cs:152-155 */
static inline int dda_is_beyond_far_clip(const dda_t *d, float farClip)
{
	return m_cmin(d->tMax) >= farClip;
}

/* ------------------------------------------------------------------ */
/* CameraData helpers, CameraData.cs                                   */
/* ------------------------------------------------------------------ */
/* math.mul(float4x4, float4) = c0*x + c1*y + c2*z + c3*w, CameraData.cs:39-48 */
static f4 cam_mul(const float *m, f4 v)
{
	f4 r;
	r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
	r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
	r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
	r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
	return r;
}

static inline float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; } /* CameraData.cs:117-120 */

/* local ClipMin, CameraData.cs:101-107 */
static inline float clip_min(f3 pMin, f3 pMax, float frustum)
{
	float frustum_inv = 1.0f / frustum;
	float c0 = cross2(1.0f, frustum_inv, pMax.x, pMax.z);
	float c1 = cross2(1.0f, frustum_inv, pMin.x, pMin.z);
	return 1.0f - (c0 / (c0 - c1));
}

/* local ClipMax, CameraData.cs:109-115 */
static inline float clip_max(f3 pMin, f3 pMax, float frustum)
{
	float frustum_inv = 1.0f / frustum;
	float c0 = cross2(1.0f, frustum_inv, pMax.x, pMax.z);
	float c1 = cross2(1.0f, frustum_inv, pMin.x, pMin.z);
	return c1 / (c1 - c0);
}

/* CameraData.GetWorldBoundsClippingCamSpace, CameraData.cs:51-99 */
static int get_world_bounds_clipping_cam_space(f3 pMin, f3 pMax, float fMin, float fMax, float *minLerp, float *maxLerp)
{
	if (pMin.x > pMin.z * fMax) {
		if (pMax.x > pMax.z * fMax) {
			*minLerp = 0.0f;
			*maxLerp = 1.0f;
			return 1;
		}
		*minLerp = clip_min(pMin, pMax, fMax);
		if (pMax.x < pMax.z * fMin) {
			*maxLerp = clip_max(pMin, pMax, fMin);
		} else {
			*maxLerp = 1.0f;
		}
	} else if (pMax.x > pMax.z * fMax) {
		*maxLerp = clip_max(pMin, pMax, fMax);
		if (pMin.x < pMin.z * fMin) {
			*minLerp = clip_min(pMin, pMax, fMin);
		} else {
			*minLerp = 0.0f;
		}
	} else {
		if (pMin.x < pMin.z * fMin) {
			if (pMax.x < pMax.z * fMin) {
				*minLerp = 0.0f;
				*maxLerp = 1.0f;
				return 1;
			}
			*minLerp = clip_min(pMin, pMax, fMin);
			*maxLerp = 1.0f;
		} else if (pMax.x < pMax.z * fMin) {
			*maxLerp = clip_max(pMin, pMax, fMin);
			*minLerp = 0.0f;
		} else {
			*minLerp = 0.0f;
			*maxLerp = 1.0f;
		}
	}
	return 0;
}

/* CameraData.ClipHomogeneousCameraSpaceLine(a, b), CameraData.cs:124-138 */
static int clip_line(f3 *a, f3 *b)
{
	if (a->y <= 0.0f) {
		if (b->y <= 0.0f) {
			return 0;
		}
		float v = b->y / (b->y - a->y);
		*a = f3_lerp(*b, *a, v);
	} else if (b->y <= 0.0f) {
		float v = a->y / (a->y - b->y);
		*b = f3_lerp(*a, *b, v);
	}
	return 1;
}

/* CameraData.ClipHomogeneousCameraSpaceLine(pA, pB, uA, uB), CameraData.cs:141-157 */
static int clip_line_u(f3 *pA, f3 *pB, float *uA, float *uB)
{
	if (pA->y <= 0.0f) {
		if (pB->y <= 0.0f) {
			return 0;
		}
		float v = pB->y / (pB->y - pA->y);
		*pA = f3_lerp(*pB, *pA, v);
		*uA = m_lerp(*uB, *uA, v);
	} else if (pB->y <= 0.0f) {
		float v = pA->y / (pA->y - pB->y);
		*pB = f3_lerp(*pA, *pB, v);
		*uB = m_lerp(*uA, *uB, v);
	}
	return 1;
}

/* ------------------------------------------------------------------ */
/* Contexts, DrawSegmentRayJob.cs:718-734, RayBuffer.cs:103-128        */
/* ------------------------------------------------------------------ */
#define RAYS_PER_PARTIAL 256 /* RayBuffer.cs:18 */
#define RAYS_SHIFT 8         /* RayBuffer.cs:19 */

typedef struct { /* RayBuffer.Native */
	uint32_t **Partials;
	int PartialWidth;
} raybuffer_native;

/* RayBuffer.Native.GetRayColumn, RayBuffer.cs:121-128 */
static inline uint32_t *get_ray_column(const raybuffer_native *rb, int rayIndex)
{
	int partialIdx = rayIndex >> RAYS_SHIFT;
	int rowIdx = rayIndex & (RAYS_PER_PARTIAL - 1);
	return rb->Partials[partialIdx] + rowIdx * rb->PartialWidth;
}

typedef struct { /* SegmentContext, DrawSegmentRayJob.cs:718-727 */
	raybuffer_native activeRayBufferFull;
	orc_segment_data segment;
	int originalNextFreePixelMin;
	int originalNextFreePixelMax;
	int axisMappedToY;
	int segmentRayIndexOffset;
	int seenPixelCacheLength;
} segment_context;

typedef struct { /* DrawContext, DrawSegmentRayJob.cs:729-734 */
	const world_t *worldLODs;
	const orc_camera_data *camera;
	f2 screen;
} draw_context;

typedef struct { const segment_context *context; int planeRayIndex; } ray_context;                 /* :42-46 */
typedef struct { const segment_context *segment; int planeRayIndex; dda_t ddaRay; } ray_dda_context; /* :79-84 */
typedef struct { /* RayContinuation, :146-153 */
	const segment_context *segment;
	uint32_t *rayColumn;
	int planeRayIndex;
	dda_t ddaRay;
	int lod;
} ray_continuation;

/* ColorARGB32(25,25,25) = bytes A=255,R=25,G=25,B=25 in memory order
 * (Color24.cs:8-19; DrawSegmentRayJob.cs:702,712); little-endian uint32. */
#define SKYBOX_ARGB 0x191919FFu

typedef struct { int64_t S, E, C, P, lod[ORC_LOD_LEVELS]; } tls_counters;

/* DrawSegmentRayJob.WriteSkybox, DrawSegmentRayJob.cs:699-708 */
static void write_skybox(int omin, int omax, uint32_t *rayColumn, const uint8_t *seen, tls_counters *tc)
{
	for (int y = omin; y <= omax; y++) {
		if (seen[y] == 0) {
			rayColumn[y] = SKYBOX_ARGB;
			tc->P++;
		}
	}
}

/* DrawSegmentRayJob.WriteSkyboxFull, DrawSegmentRayJob.cs:710-716 */
static void write_skybox_full(int omin, int omax, uint32_t *rayColumn, tls_counters *tc)
{
	for (int y = omin; y <= omax; y++) {
		rayColumn[y] = SKYBOX_ARGB;
		tc->P++;
	}
}

/* DrawSegmentRayJob.ReducePixelHorizon, DrawSegmentRayJob.cs:660-697 */
static void reduce_pixel_horizon(int omin, int omax, int *rbMin, int *rbMax, int *nfMin, int *nfMax,
                                 const uint8_t *seen, float *frustumBoundsMin, float *frustumBoundsMax)
{
	if (*rbMin <= *nfMin) {
		*rbMin = *nfMin;
		if (*rbMax >= *nfMin) {
			*nfMin = *rbMax + 1;
			while (*nfMin <= omax && seen[*nfMin] > 0) {
				*nfMin += 1;
			}
			*frustumBoundsMin = *nfMin - 0.501f;
		}
	}
	if (*rbMax >= *nfMax) {
		*rbMax = *nfMax;
		if (*rbMin <= *nfMax) {
			*nfMax = *rbMin - 1;
			while (*nfMax >= omin && seen[*nfMax] > 0) {
				*nfMax -= 1;
			}
			*frustumBoundsMax = *nfMax + 0.501f;
		}
	}
}

/* DrawSegmentRayJob.SetupProjectedPlaneParams, DrawSegmentRayJob.cs:622-651 */
static void setup_projected_plane_params(const orc_camera_data *camera, const dda_t *ray, float worldMaxY, int yAxis,
                                         f3 *planeStartBottomProjected, f3 *planeStartTopProjected, f3 *planeRayDirectionProjected)
{
	f2 start = ray->start;
	f4 top = cam_mul(camera->WorldToScreenMatrix, (f4){ start.x, worldMaxY, start.y, 1.0f });
	f4 bottom = cam_mul(camera->WorldToScreenMatrix, (f4){ start.x, 0.0f, start.y, 1.0f });
	f4 dir = cam_mul(camera->WorldToScreenMatrix, (f4){ ray->dir.x, 0.0f, ray->dir.y, 0.0f });
	if (yAxis == 0) {
		*planeStartBottomProjected = (f3){ bottom.x, bottom.z, bottom.w };
		*planeStartTopProjected = (f3){ top.x, top.z, top.w };
		*planeRayDirectionProjected = (f3){ dir.x, dir.z, dir.w };
	} else {
		*planeStartBottomProjected = (f3){ bottom.y, bottom.z, bottom.w };
		*planeStartTopProjected = (f3){ top.y, top.z, top.w };
		*planeRayDirectionProjected = (f3){ dir.y, dir.z, dir.w };
	}
}

#define FLOAT_EPSILON 1.401298464324817e-45f /* C# float.Epsilon: smallest denormal */

static inline void swapf(float *a, float *b) { float t = *a; *a = *b; *b = t; }

/* DrawSegmentRayJob.ExecuteRay, DrawSegmentRayJob.cs:195-620 */
static void execute_ray(const ray_continuation *rayContext, const draw_context *drawContext, int ITERATION_DIRECTION,
                        uint8_t *seenPixelCache, tls_counters *tc)
{
	const segment_context *segmentContext = rayContext->segment;
	dda_t ray = rayContext->ddaRay;
	uint32_t *rayColumn = rayContext->rayColumn;
	const orc_camera_data *camera = drawContext->camera;

	int lod = rayContext->lod;
	int voxelScale = 1 << lod;
	const world_t *world = drawContext->worldLODs + lod;
	float farClip = camera->FarClip;
	rle_column worldColumn;
	memset(&worldColumn, 0, sizeof worldColumn);
	float lodMax = camera->LODDistances[lod];

	memset(seenPixelCache, 0, (size_t)segmentContext->seenPixelCacheLength); /* stackalloc, zeroed (:208) */

	const int omin = segmentContext->originalNextFreePixelMin;
	const int omax = segmentContext->originalNextFreePixelMax;
	int nextFreePixelMin = omin;
	int nextFreePixelMax = omax;

	float worldMaxY = (float)world->dimY;
	float cameraPosYNormalized = camera->PositionY / worldMaxY;

	float frustumBoundsMin = nextFreePixelMin - 0.501f;
	float frustumBoundsMax = nextFreePixelMax + 0.501f;

	float frustumDirMaxWorld = FLOAT_EPSILON;
	float frustumDirMinWorld = FLOAT_EPSILON;

	f3 planeStartBottomProjected, planeStartTopProjected, planeRayDirectionProjected;
	setup_projected_plane_params(camera, &ray, worldMaxY, segmentContext->axisMappedToY,
	                             &planeStartBottomProjected, &planeStartTopProjected, &planeRayDirectionProjected);

	while (1) {
		if (ray.intersectionDistances.x >= lodMax && lod < ORC_LOD_LEVELS - 1) { /* :237-243 (+ the same guard) */
			dda_next_lod(&ray, voxelScale);
			lod++;
			voxelScale *= 2;
			world++;
			lodMax = camera->LODDistances[lod];
		}

		int columnRuns = world_get_voxel_column(world, ray.position, &worldColumn);
		if (columnRuns == -1) {
			write_skybox(omin, omax, rayColumn, seenPixelCache, tc);
			return;
		}
		tc->S++;
		tc->lod[lod]++;
		if (columnRuns == 0) {
			if (dda_step(&ray, farClip)) {
				break;
			}
			continue;
		}

		float worldBoundsMin = 0.0f;
		float worldBoundsMax = worldMaxY;

		if (frustumDirMaxWorld != FLOAT_EPSILON) { /* :261-281 */
			float distTop = m_select(ray.intersectionDistances.x, ray.intersectionDistances.y, frustumDirMaxWorld > 0.0f);
			float distBot = m_select(ray.intersectionDistances.x, ray.intersectionDistances.y, frustumDirMinWorld < 0.0f);
			float newMax = camera->PositionY + frustumDirMaxWorld * distTop;
			float newMin = camera->PositionY + frustumDirMinWorld * distBot;
			if (newMin > worldBoundsMax || newMax < worldBoundsMin) {
				write_skybox(omin, omax, rayColumn, seenPixelCache, tc);
				return;
			}
			if ((float)worldColumn.worldMin > newMax || (float)worldColumn.worldMax < newMin) {
				if (dda_step(&ray, farClip)) {
					break;
				}
				continue;
			}
			worldBoundsMin = newMin;
			worldBoundsMax = newMax;
		}

		/* :289-293 */
		f3 camSpaceMinLast = f3_add(planeStartBottomProjected, f3_muls(planeRayDirectionProjected, ray.intersectionDistances.x));
		f3 camSpaceMinNext = f3_add(planeStartBottomProjected, f3_muls(planeRayDirectionProjected, ray.intersectionDistances.y));
		f3 camSpaceMaxLast = f3_add(planeStartTopProjected, f3_muls(planeRayDirectionProjected, ray.intersectionDistances.x));
		f3 camSpaceMaxNext = f3_add(planeStartTopProjected, f3_muls(planeRayDirectionProjected, ray.intersectionDistances.y));

		if (ray.intersectionDistances.x > 2.0f && frustumDirMaxWorld == FLOAT_EPSILON) { /* :295-422 */
			float clipLastMinLerp, clipLastMaxLerp, clipNextMinLerp, clipNextMaxLerp;
			int clippedLast = get_world_bounds_clipping_cam_space(camSpaceMinLast, camSpaceMaxLast, frustumBoundsMin, frustumBoundsMax,
			                                                      &clipLastMinLerp, &clipLastMaxLerp);
			int clippedNext = get_world_bounds_clipping_cam_space(camSpaceMinNext, camSpaceMaxNext, frustumBoundsMin, frustumBoundsMax,
			                                                      &clipNextMinLerp, &clipNextMaxLerp);

			float camSpaceClippedMin, camSpaceClippedMax;
			if (clippedLast) {
				if (clippedNext) {
					write_skybox(omin, omax, rayColumn, seenPixelCache, tc);
					return;
				} else {
					worldBoundsMin = m_lerp(0.0f, worldMaxY, clipNextMinLerp);
					worldBoundsMax = m_lerp(0.0f, worldMaxY, clipNextMaxLerp);

					frustumDirMaxWorld = (worldBoundsMax - camera->PositionY) / ray.intersectionDistances.y;
					frustumDirMinWorld = (worldBoundsMin - camera->PositionY) / ray.intersectionDistances.y;

					f3 minClip = f3_lerp(camSpaceMinNext, camSpaceMaxNext, clipNextMinLerp);
					f3 maxClip = f3_lerp(camSpaceMinNext, camSpaceMaxNext, clipNextMaxLerp);

					camSpaceClippedMin = minClip.x / minClip.z;
					camSpaceClippedMax = maxClip.x / maxClip.z;
					if (camSpaceClippedMax < camSpaceClippedMin) {
						swapf(&camSpaceClippedMin, &camSpaceClippedMax);
					}
				}
			} else {
				if (clippedNext) {
					worldBoundsMin = m_lerp(0.0f, worldMaxY, clipLastMinLerp);
					worldBoundsMax = m_lerp(0.0f, worldMaxY, clipLastMaxLerp);
					f3 minClip = f3_lerp(camSpaceMinLast, camSpaceMaxLast, clipLastMinLerp);
					f3 maxClip = f3_lerp(camSpaceMinLast, camSpaceMaxLast, clipLastMaxLerp);

					frustumDirMaxWorld = (worldBoundsMax - camera->PositionY) / ray.intersectionDistances.x;
					frustumDirMinWorld = (worldBoundsMin - camera->PositionY) / ray.intersectionDistances.x;

					camSpaceClippedMin = minClip.x / minClip.z;
					camSpaceClippedMax = maxClip.x / maxClip.z;
					if (camSpaceClippedMax < camSpaceClippedMin) {
						swapf(&camSpaceClippedMin, &camSpaceClippedMax);
					}
				} else {
					if (clipLastMinLerp < clipNextMinLerp) {
						worldBoundsMin = m_lerp(0.0f, worldMaxY, clipLastMinLerp);
						frustumDirMinWorld = (worldBoundsMin - camera->PositionY) / ray.intersectionDistances.x;
					} else {
						worldBoundsMin = m_lerp(0.0f, worldMaxY, clipNextMinLerp);
						frustumDirMinWorld = (worldBoundsMin - camera->PositionY) / ray.intersectionDistances.y;
					}

					if (clipLastMaxLerp > clipNextMaxLerp) {
						worldBoundsMax = m_lerp(0.0f, worldMaxY, clipLastMaxLerp);
						frustumDirMaxWorld = (worldBoundsMax - camera->PositionY) / ray.intersectionDistances.x;
					} else {
						worldBoundsMax = m_lerp(0.0f, worldMaxY, clipNextMaxLerp);
						frustumDirMaxWorld = (worldBoundsMax - camera->PositionY) / ray.intersectionDistances.y;
					}

					f3 minClipA = f3_lerp(camSpaceMinLast, camSpaceMaxLast, clipLastMinLerp);
					f3 maxClipA = f3_lerp(camSpaceMinLast, camSpaceMaxLast, clipLastMaxLerp);
					f3 minClipB = f3_lerp(camSpaceMinNext, camSpaceMaxNext, clipNextMinLerp);
					f3 maxClipB = f3_lerp(camSpaceMinNext, camSpaceMaxNext, clipNextMaxLerp);

					float minNext = minClipB.x / minClipB.z;
					float minLast = minClipA.x / minClipA.z;
					float maxNext = maxClipB.x / maxClipB.z;
					float maxLast = maxClipA.x / maxClipA.z;

					if (maxNext < minNext) { swapf(&maxNext, &minNext); }
					if (maxLast < minLast) { swapf(&maxLast, &minLast); }

					camSpaceClippedMin = m_min(minLast, minNext);
					camSpaceClippedMax = m_max(maxLast, maxNext);
				}
			}

			worldBoundsMin = floorf(worldBoundsMin);
			worldBoundsMax = ceilf(worldBoundsMax);

			int writableMinPixel = f2i(floorf(camSpaceClippedMin));
			int writableMaxPixel = f2i(ceilf(camSpaceClippedMax));

			if (writableMaxPixel < nextFreePixelMin || writableMinPixel > nextFreePixelMax) {
				write_skybox(omin, omax, rayColumn, seenPixelCache, tc);
				return;
			}

			if (writableMinPixel > nextFreePixelMin) {
				nextFreePixelMin = writableMinPixel;
				while (nextFreePixelMin <= omax && seenPixelCache[nextFreePixelMin] > 0) {
					nextFreePixelMin += 1;
				}
			}
			if (writableMaxPixel < nextFreePixelMax) {
				nextFreePixelMax = writableMaxPixel;
				while (nextFreePixelMax >= omin && seenPixelCache[nextFreePixelMax] > 0) {
					nextFreePixelMax -= 1;
				}
			}
			if (nextFreePixelMin > nextFreePixelMax) {
				write_skybox(omin, omax, rayColumn, seenPixelCache, tc);
				return;
			}
		}

		float elementBoundsMin;
		float elementBoundsMax;
		const rle_element *elementPointer;

		/* World.cs:175-188: guard start / guard end / colour pointer */
		const rle_element *guardStart = world->elements + worldColumn.storageOffset;
		if (ITERATION_DIRECTION > 0) {
			elementBoundsMin = worldMaxY;
			elementBoundsMax = worldMaxY;
			elementPointer = guardStart;
		} else {
			elementBoundsMin = 0.0f;
			elementBoundsMax = 0.0f;
			elementPointer = guardStart + worldColumn.runCount + 1;
		}
		const uint32_t *worldColumnColors = (const uint32_t *)guardStart + worldColumn.runCount + 2;

		while (1) {
			elementPointer += ITERATION_DIRECTION;

			rle_element element = *elementPointer;
			tc->E++;
			if (element.Length == 0) { /* !IsValid */
				break;
			}

			if (ITERATION_DIRECTION > 0) {
				elementBoundsMax = elementBoundsMin;
				elementBoundsMin = elementBoundsMin - (float)(element.Length * voxelScale);
			} else {
				elementBoundsMin = elementBoundsMax;
				elementBoundsMax = elementBoundsMin + (float)(element.Length * voxelScale);
			}

			if (element.ColorsIndex < 0) { /* IsAir */
				continue;
			}

			if (elementBoundsMin > worldBoundsMax) {
				if (ITERATION_DIRECTION < 0) {
					break;
				} else {
					continue;
				}
			}

			if (elementBoundsMax < worldBoundsMin) {
				if (ITERATION_DIRECTION > 0) {
					break;
				} else {
					continue;
				}
			}

			float portionBottom = m_unlerp(0.0f, worldMaxY, elementBoundsMin);
			float portionTop = m_unlerp(0.0f, worldMaxY, elementBoundsMax);
			f3 camSpaceFrontBottom = f3_lerp(camSpaceMinLast, camSpaceMaxLast, portionBottom);
			f3 camSpaceFrontTop = f3_lerp(camSpaceMinLast, camSpaceMaxLast, portionTop);

			/* side of the run, :484-542 */
			{
				float uA = (float)element.Length;
				float uB = 0.0f;
				/* the clip takes the locals by ref (:489): camSpaceFrontTop/Bottom are reused
				 * below (:555,:562) in their clipped form. */
				if (clip_line_u(&camSpaceFrontBottom, &camSpaceFrontTop, &uA, &uB)) {
					f2 uvA = { 1.0f / camSpaceFrontBottom.z, uA / camSpaceFrontBottom.z };
					f2 uvB = { 1.0f / camSpaceFrontTop.z, uB / camSpaceFrontTop.z };

					/* ProjectClippedToScreen, CameraData.cs:160-163 */
					f2 rayBufferBoundsFloat = { camSpaceFrontBottom.x / camSpaceFrontBottom.z, camSpaceFrontTop.x / camSpaceFrontTop.z };

					if (rayBufferBoundsFloat.x > rayBufferBoundsFloat.y) {
						swapf(&rayBufferBoundsFloat.x, &rayBufferBoundsFloat.y);
						f2 t = uvA; uvA = uvB; uvB = t;
					}

					int rayBufferBoundsMin = f2i(m_round(rayBufferBoundsFloat.x));
					int rayBufferBoundsMax = f2i(m_round(rayBufferBoundsFloat.y));

					if (rayBufferBoundsMax >= nextFreePixelMin && rayBufferBoundsMin <= nextFreePixelMax) {
						reduce_pixel_horizon(omin, omax, &rayBufferBoundsMin, &rayBufferBoundsMax, &nextFreePixelMin, &nextFreePixelMax,
						                     seenPixelCache, &frustumBoundsMin, &frustumBoundsMax);

						for (int y = rayBufferBoundsMin; y <= rayBufferBoundsMax; y++) {
							if (seenPixelCache[y] == 0) {
								frustumDirMaxWorld = FLOAT_EPSILON;
								seenPixelCache[y] = 1;

								float l = m_unlerp(rayBufferBoundsFloat.x, rayBufferBoundsFloat.y, (float)y);
								f2 wu = { m_lerp(uvA.x, uvB.x, l), m_lerp(uvA.y, uvB.y, l) };
								float u = wu.y / wu.x;

								int colorIdx = m_clampi(f2i(floorf(u)), 0, element.Length - 1) + element.ColorsIndex;
								rayColumn[y] = worldColumnColors[colorIdx];
								tc->C++;
								tc->P++;
							}
						}

						if (nextFreePixelMin > nextFreePixelMax) {
							write_skybox(omin, omax, rayColumn, seenPixelCache, tc);
							return;
						}
					}
				}
			}

			/* top / bottom of the run, :544-610 */
			f3 camSpaceSecondaryA;
			f3 camSpaceSecondaryB;
			uint32_t secondaryColor;

			if (portionTop < cameraPosYNormalized) {
				if (elementBoundsMax > worldBoundsMax) {
					continue;
				}
				secondaryColor = worldColumnColors[element.ColorsIndex + 0];
				tc->C++;
				camSpaceSecondaryA = f3_lerp(camSpaceMinNext, camSpaceMaxNext, portionTop);
				camSpaceSecondaryB = camSpaceFrontTop;
			} else if (portionBottom > cameraPosYNormalized) {
				if (elementBoundsMin < worldBoundsMin) {
					continue;
				}
				secondaryColor = worldColumnColors[element.ColorsIndex + element.Length - 1];
				tc->C++;
				camSpaceSecondaryA = f3_lerp(camSpaceMinNext, camSpaceMaxNext, portionBottom);
				camSpaceSecondaryB = camSpaceFrontBottom;
			} else {
				continue;
			}

			if (clip_line(&camSpaceSecondaryA, &camSpaceSecondaryB)) {
				f2 rayBufferBoundsFloat = { camSpaceSecondaryA.x / camSpaceSecondaryA.z, camSpaceSecondaryB.x / camSpaceSecondaryB.z };
				rayBufferBoundsFloat.x = m_round(rayBufferBoundsFloat.x);
				rayBufferBoundsFloat.y = m_round(rayBufferBoundsFloat.y);

				int rayBufferBoundsMin = f2i(rayBufferBoundsFloat.x);
				int rayBufferBoundsMax = f2i(rayBufferBoundsFloat.y);

				if (rayBufferBoundsMin > rayBufferBoundsMax) {
					int t = rayBufferBoundsMin; rayBufferBoundsMin = rayBufferBoundsMax; rayBufferBoundsMax = t;
				}

				if (rayBufferBoundsMax >= nextFreePixelMin && rayBufferBoundsMin <= nextFreePixelMax) {
					reduce_pixel_horizon(omin, omax, &rayBufferBoundsMin, &rayBufferBoundsMax, &nextFreePixelMin, &nextFreePixelMax,
					                     seenPixelCache, &frustumBoundsMin, &frustumBoundsMax);

					for (int y = rayBufferBoundsMin; y <= rayBufferBoundsMax; y++) {
						if (seenPixelCache[y] == 0) {
							frustumDirMaxWorld = FLOAT_EPSILON;
							seenPixelCache[y] = 1;
							rayColumn[y] = secondaryColor;
							tc->P++;
						}
					}

					if (nextFreePixelMin > nextFreePixelMax) {
						write_skybox(omin, omax, rayColumn, seenPixelCache, tc);
						return;
					}
				}
			}
		}

		if (dda_step(&ray, farClip)) {
			break;
		}
	}

	write_skybox(omin, omax, rayColumn, seenPixelCache, tc);
}

/* ------------------------------------------------------------------ */
/* The four jobs                                                       */
/* ------------------------------------------------------------------ */

/* RaySetupJob.Execute, DrawSegmentRayJob.cs:19-39 */
static void ray_setup_job(const segment_context contexts[4], ray_context *rays, int startIndex)
{
	int planeIndex = startIndex;
	for (int j = 0; j < 4; j++) {
		int segmentRays = contexts[j].segment.RayCount;
		if (segmentRays <= 0) {
			continue;
		}
		if (planeIndex >= segmentRays) {
			planeIndex -= segmentRays;
			continue;
		}
		rays[startIndex].context = contexts + j;
		rays[startIndex].planeRayIndex = planeIndex;
		break;
	}
}

/* DDASetupJob.Execute, DrawSegmentRayJob.cs:58-76 */
static void dda_setup_job(const ray_context *raysInput, ray_dda_context *raysOutput, const draw_context *drawContext, int i)
{
	const orc_segment_data *seg = &raysInput[i].context->segment;
	float endRayLerp = raysInput[i].planeRayIndex / (float)seg->RayCount;
	f2 d = {
		m_lerp(seg->CamLocalPlaneRayMin[0], seg->CamLocalPlaneRayMax[0], endRayLerp),
		m_lerp(seg->CamLocalPlaneRayMin[1], seg->CamLocalPlaneRayMax[1], endRayLerp),
	};
	/* math.normalize(float2) = rsqrt(dot(x,x)) * x, rsqrt(x) = 1/sqrt(x) */
	float r = 1.0f / sqrtf(d.x * d.x + d.y * d.y);
	f2 n = { r * d.x, r * d.y };
	f2 start = { drawContext->camera->PositionXZ[0], drawContext->camera->PositionXZ[1] };
	raysOutput[i].segment = raysInput[i].context;
	raysOutput[i].planeRayIndex = raysInput[i].planeRayIndex;
	raysOutput[i].ddaRay = dda_new(start, n);
}

/* TraceToFirstColumnJob.Execute, DrawSegmentRayJob.cs:95-143.
 * Returns 1 when the continuation must be appended to the render list. */
static int trace_to_first_column_job(const ray_dda_context *inRays, const draw_context *drawContext, int index,
                                     ray_continuation *cont, tls_counters *tc)
{
	const ray_dda_context *rayContext = &inRays[index];
	const segment_context *segmentContext = rayContext->segment;

	cont->segment = rayContext->segment;
	cont->ddaRay = rayContext->ddaRay;
	cont->planeRayIndex = rayContext->planeRayIndex;
	cont->rayColumn = get_ray_column(&segmentContext->activeRayBufferFull, rayContext->planeRayIndex + segmentContext->segmentRayIndexOffset);
	cont->lod = 0;

	const world_t *world = drawContext->worldLODs;
	float farClip = drawContext->camera->FarClip;
	float lodMax = drawContext->camera->LODDistances[0];

	i2 dimensions = { world->dimX, world->dimZ };
	i2 startPos = cont->ddaRay.position;
	if (startPos.x < 0 || startPos.y < 0 || startPos.x >= dimensions.x || startPos.y >= dimensions.y) {
		f2 dimsf = { (float)dimensions.x, (float)dimensions.y };
		if (dda_step_to_world_intersection(&cont->ddaRay, dimsf)) {
			/* "cont->lod < 5": memory-safety guard only (the reference would index LODDistances[6]);
			 * a ray that far away is beyond far clip and becomes skybox either way. */
			while (cont->ddaRay.intersectionDistances.x >= lodMax && cont->lod < ORC_LOD_LEVELS - 1) {
				dda_next_lod(&cont->ddaRay, 1 << cont->lod);
				cont->lod++;
				world++;
				lodMax = drawContext->camera->LODDistances[cont->lod];
			}
			if (dda_is_beyond_far_clip(&cont->ddaRay, farClip)) {
				write_skybox_full(segmentContext->originalNextFreePixelMin, segmentContext->originalNextFreePixelMax, cont->rayColumn, tc);
			} else {
				return 1;
			}
		} else {
			write_skybox_full(segmentContext->originalNextFreePixelMin, segmentContext->originalNextFreePixelMax, cont->rayColumn, tc);
		}
		return 0;
	}
	return 1;
}

/* Mathf.RoundToInt = (int)Math.Round(f): half to even */
static inline int round_to_int(float f) { return f2i(nearbyintf(f)); }

/* Worker threads worth starting: OpenMP's default, capped by the CPU-time quota of the control group the process
 * runs in (cgroup v2 cpu.max "quota period"; containers often see every host CPU but may only use a few of them,
 * and more runnable threads than that just take turns). */
int orc_max_threads(void)
{
#ifdef _OPENMP
	int threads = omp_get_max_threads();
	FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
	if (f) {
		long long quota = 0, period = 0;
		if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
			long long cpus = (quota + period - 1) / period;
			if (cpus >= 1 && cpus < threads) { threads = (int)cpus; }
		}
		fclose(f);
	}
	return threads;
#else
	return 1;
#endif
}

/* RenderManager.DrawSegments, RenderManager.cs:258-372 */
int orc_draw_segments(const orc_segment_data segments[4],
                      const orc_world worldLODs[ORC_LOD_LEVELS],
                      const orc_camera_data *camera,
                      int screenWidth, int screenHeight,
                      const float vanishingPointScreenSpace[2],
                      uint32_t *rayBufferTopDown,
                      uint32_t *rayBufferLeftRight,
                      int threads,
                      orc_counters *counters)
{
	if (!segments || !worldLODs || !camera || screenWidth <= 0 || screenHeight <= 0) {
		return -1;
	}
#ifdef _OPENMP
	if (threads <= 0) {
		threads = orc_max_threads();
	}
#else
	threads = 1;
#endif

	world_t worlds[ORC_LOD_LEVELS];
	for (int i = 0; i < ORC_LOD_LEVELS; i++) { /* World ctor, World.cs:36-43 */
		const orc_world *w = &worldLODs[i];
		worlds[i].columns = (const rle_column *)w->storage;
		worlds[i].elements = (const rle_element *)((const rle_column *)w->storage + w->columnCount);
		worlds[i].dimX = w->dimX;
		worlds[i].dimY = w->dimY;
		worlds[i].dimZ = w->dimZ;
		worlds[i].lod = w->lod;
		worlds[i].indexingMulX = w->dimZ >> w->lod;
		worlds[i].dimensionMaskXZ.x = w->dimX - 1;
		worlds[i].dimensionMaskXZ.y = w->dimZ - 1;
	}

	/* RayBuffer sizes, RenderManager.cs:35-36; partial textures back to back. */
	int tdRays = screenWidth + 2 * screenHeight;
	int lrRays = 2 * screenWidth + screenHeight;
	int tdPartials = (tdRays + RAYS_PER_PARTIAL - 1) / RAYS_PER_PARTIAL;
	int lrPartials = (lrRays + RAYS_PER_PARTIAL - 1) / RAYS_PER_PARTIAL;
	uint32_t **tdTab = (uint32_t **)malloc(sizeof(uint32_t *) * (size_t)tdPartials);
	uint32_t **lrTab = (uint32_t **)malloc(sizeof(uint32_t *) * (size_t)lrPartials);
	for (int i = 0; i < tdPartials; i++) {
		tdTab[i] = rayBufferTopDown + (size_t)i * RAYS_PER_PARTIAL * (size_t)screenHeight;
	}
	for (int i = 0; i < lrPartials; i++) {
		lrTab[i] = rayBufferLeftRight + (size_t)i * RAYS_PER_PARTIAL * (size_t)screenWidth;
	}
	raybuffer_native rbTopDown = { tdTab, screenHeight };
	raybuffer_native rbLeftRight = { lrTab, screenWidth };

	draw_context drawContext;
	drawContext.camera = camera;
	drawContext.screen.x = (float)screenWidth;
	drawContext.screen.y = (float)screenHeight;
	drawContext.worldLODs = worlds;

	segment_context segmentContexts[4];
	memset(segmentContexts, 0, sizeof segmentContexts);
	int totalRays = 0;
	for (int segmentIndex = 0; segmentIndex < 4; segmentIndex++) { /* :284-318 */
		segment_context *context = &segmentContexts[segmentIndex];
		context->segment = segments[segmentIndex];
		totalRays += segments[segmentIndex].RayCount;

		if (segments[segmentIndex].RayCount <= 0) {
			continue;
		}

		context->axisMappedToY = (segmentIndex > 1) ? 0 : 1;
		context->segmentRayIndexOffset = 0;
		if (segmentIndex == 1) { context->segmentRayIndexOffset = segments[0].RayCount; }
		if (segmentIndex == 3) { context->segmentRayIndexOffset = segments[2].RayCount; }

		int nfx, nfy;
		if (segmentIndex < 2) {
			context->activeRayBufferFull = rbTopDown;
			if (segmentIndex == 0) {
				nfx = m_clampi(round_to_int(vanishingPointScreenSpace[1]), 0, screenHeight - 1);
				nfy = screenHeight - 1;
			} else {
				nfx = 0;
				nfy = m_clampi(round_to_int(vanishingPointScreenSpace[1]), 0, screenHeight - 1);
			}
		} else {
			context->activeRayBufferFull = rbLeftRight;
			if (segmentIndex == 3) {
				nfx = 0;
				nfy = m_clampi(round_to_int(vanishingPointScreenSpace[0]), 0, screenWidth - 1);
			} else {
				nfx = m_clampi(round_to_int(vanishingPointScreenSpace[0]), 0, screenWidth - 1);
				nfy = screenWidth - 1;
			}
		}
		context->originalNextFreePixelMin = nfx;
		context->originalNextFreePixelMax = nfy;
		context->seenPixelCacheLength = (int)ceilf(context->axisMappedToY ? drawContext.screen.y : drawContext.screen.x);
	}

	if (totalRays <= 0) {
		free(tdTab);
		free(lrTab);
		if (counters) { memset(counters, 0, sizeof *counters); }
		return 0;
	}

	ray_context *rayContext = (ray_context *)malloc(sizeof(ray_context) * (size_t)totalRays);
	ray_dda_context *rayDDAContext = (ray_dda_context *)malloc(sizeof(ray_dda_context) * (size_t)totalRays);
	ray_continuation *rayContinuations = (ray_continuation *)malloc(sizeof(ray_continuation) * (size_t)totalRays);
	int continuationCount = 0;
	int cacheLen = (screenWidth > screenHeight ? screenWidth : screenHeight) + 64;

	tls_counters total;
	memset(&total, 0, sizeof total);

#pragma omp parallel num_threads(threads)
	{
		tls_counters tc;
		memset(&tc, 0, sizeof tc);
		uint8_t *seen = (uint8_t *)malloc((size_t)cacheLen);

		/* Schedule(totalRays, 64), RenderManager.cs:358-359 */
#pragma omp for schedule(dynamic, 64)
		for (int i = 0; i < totalRays; i++) {
			ray_setup_job(segmentContexts, rayContext, i);
		}
#pragma omp for schedule(dynamic, 64)
		for (int i = 0; i < totalRays; i++) {
			dda_setup_job(rayContext, rayDDAContext, &drawContext, i);
		}
		/* Schedule(totalRays, 4), :360; NativeList.ParallelWriter.AddNoResize = atomic append */
#pragma omp for schedule(dynamic, 4)
		for (int i = 0; i < totalRays; i++) {
			ray_continuation cont;
			if (trace_to_first_column_job(rayDDAContext, &drawContext, i, &cont, &tc)) {
				int slot;
#pragma omp atomic capture
				slot = continuationCount++;
				rayContinuations[slot] = cont;
			}
		}
		/* Schedule(totalRays, 1), :361; RenderJob.Execute :164-179 */
#pragma omp for schedule(dynamic, 1)
		for (int i = 0; i < totalRays; i++) {
			if (i >= continuationCount) {
				continue;
			}
			if (camera->InverseElementIterationDirection) {
				execute_ray(&rayContinuations[i], &drawContext, -1, seen, &tc);
			} else {
				execute_ray(&rayContinuations[i], &drawContext, 1, seen, &tc);
			}
		}

		free(seen);
#pragma omp critical
		{
			total.S += tc.S;
			total.E += tc.E;
			total.C += tc.C;
			total.P += tc.P;
			for (int k = 0; k < ORC_LOD_LEVELS; k++) {
				total.lod[k] += tc.lod[k];
			}
		}
	}

	if (counters) {
		counters->S = total.S;
		counters->E = total.E;
		counters->C = total.C;
		counters->P = total.P;
		counters->R = totalRays;
		counters->continuations = continuationCount;
		for (int k = 0; k < ORC_LOD_LEVELS; k++) {
			counters->lodVisits[k] = total.lod[k];
		}
	}

	free(rayContext);
	free(rayDDAContext);
	free(rayContinuations);
	free(tdTab);
	free(lrTab);
	return totalRays;
}
