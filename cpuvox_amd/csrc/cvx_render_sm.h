// cvx_render_sm.h -- the render kernel as a per-wave state machine (gfx950 / CDNA4, wave64).  Included by cvx_kernels.h.
//
// render_kernel (cvx_kernels.h) runs ExecuteRay (DrawSegmentRayJob.cs:195-620) as one structured loop: every wave iteration is
// one column step of ALL its rays, and each block of that loop (frustum clip, element walk, side projection, pixel loops, face)
// is executed with whatever lanes need it at that column -- 24 to 42 of 64 on the benchmark (profiles/r02_section_counts.txt),
// i.e. the wave pays for the union of what its rays need.  Rays are independent of each other, so nothing forces them to stay on
// the same column: here every lane carries a STATE (the block of ExecuteRay its ray has to run next), and the wave repeatedly
// picks a block that many lanes are waiting for and runs it for exactly those lanes.  Lanes whose column is empty or culled run
// ahead, lanes that need a clip wait until enough others need one too.  Each ray still performs the reference's sequence of
// operations, value for value; only the interleaving between rays changes.  The pixels of a ray depend on nothing but that
// sequence (its own seen-mask, its own raybuffer column), so the output is the same bits.
//
// States and the reference lines they cover:
//   ADV   take the prefetched column record, step the DDA to the next column (+ LOD switch) and prefetch its record, then
//         test the current column (:237-287, :613): empty / outside the writable world bounds -> ADV again
//   CLIP  frustum clip of the column, horizon update (:289-422)
//   WALK  the ray's own walk over the solid runs of the column up to the next one that must be projected (:424-475)
//   SIDE  projection of that run's side + ReducePixelHorizon (:477-517)
//   SPIX  textured pixels of the side, two per trip (:519-542)
//   TB    top / bottom face of the run: projection, horizon, flat pixels (:544-610)
// The counting build (render_kernel<true>) stays the structured loop: it reproduces the reference's exit points, which the
// element counter E depends on.
#pragma once

namespace cvxk {

enum : int { ST_ADV = 0, ST_CLIP = 1, ST_WALK = 2, ST_SIDE = 3, ST_SPIX = 4, ST_TB = 5, ST_DONE = 6 };

// A block runs when at least this many lanes wait for it (or, when no block has that many, the fullest one runs).
#ifndef CVX_SM_T_ADV
#define CVX_SM_T_ADV 32
#endif
#ifndef CVX_SM_T_CLIP
#define CVX_SM_T_CLIP 40
#endif
#ifndef CVX_SM_T_WALK
#define CVX_SM_T_WALK 32
#endif
#ifndef CVX_SM_T_SIDE
#define CVX_SM_T_SIDE 40
#endif
#ifndef CVX_SM_T_SPIX
#define CVX_SM_T_SPIX 40
#endif
#ifndef CVX_SM_T_TB
#define CVX_SM_T_TB 40
#endif
#ifndef CVX_SM_SPIX_TRIPS
#define CVX_SM_SPIX_TRIPS 2 /* pixel-pair trips per visit of SPIX */
#endif

#ifdef CVX_SM_STATS /* diagnostic build: [k] executions of block k, [8 + k] lanes that ran it, [6] passes, [7] passes in which nothing reached its threshold, [14] waves */
#ifndef CVX_PROFILE_SECTIONS
__device__ unsigned long long g_sectionCycles[32];
#endif
#define CVX_SM_STAT(k) do { smEx[k]++; smLn[k] += (unsigned)CVX_SM_WAITING(k); } while (0)
#else
#define CVX_SM_STAT(k) ((void)0)
#endif

struct SmParams {
	int threshold; // > 0: overrides every per-block threshold (diagnostics, CVX_SM_THRESHOLD)
};

template <int DIR>
__device__ __forceinline__ void trace_wave_sm(const DevFrame &F, const DevSegment &S, const DevWorld *__restrict__ world, int planeRayIndex, bool active,
                                              uint32_t *seen, int sshift, gptr_tile tileOut, uint32_t laneByteOff, int thresholdOverride)
{
	const int omin = S.omin, omax = S.omax;
	const float farClip = F.farClip;
	const float posY = F.posY;
	const gptr_arena arena = (gptr_arena)world->arena;
	const int maskX = world->maskX, maskZ = world->maskZ;
	const int worldMaxYInt = world->dimY;
	const float worldMaxY = (float)worldMaxYInt;
	const float cameraPosYNormalized = posY / worldMaxY;
	const float invWorldMaxY = 1.0f / worldMaxY; // exact: dimY is a power of two

	int state = ST_DONE;

	// ---- per-ray state that lives across blocks -------------------------------------------------------------------------
	DDA ray;
	ray.px = ray.pz = ray.sx = ray.sz = 0;
	ray.startX = ray.startZ = ray.dirX = ray.dirZ = ray.tDeltaX = ray.tDeltaZ = ray.tMaxX = ray.tMaxZ = ray.distLast = ray.distNext = 0.0f;
	bool dirXNonNegative = false, dirZNonNegative = false;
	int lod = 0;
	float lodMax = 0.0f;
	DevWorldLevel L = world->level[0];
	int nextFreePixelMin = omin, nextFreePixelMax = omax;
	float frustumBoundsMin = (float)omin - 0.501f, frustumBoundsMax = (float)omax + 0.501f;
	float frustumDirMaxWorld = CVX_FLOAT_EPSILON, frustumDirMinWorld = CVX_FLOAT_EPSILON;
	f3 planeStartBottom = { 0.0f, 0.0f, 0.0f }, planeStartTop = { 0.0f, 0.0f, 0.0f }, planeDir = { 0.0f, 0.0f, 0.0f };
	int guardSteps = 0;
	uint4 nextHeader = { 0u, 0u, 0u, 0u }, nextQueue = { 0u, 0u, 0u, 0u }; // record of the column the DDA stands on (in flight)
	bool pendingEnd = false;                                                 // the column just processed was the ray's last one (far clip / world edge)
	// the column being processed
	uint32_t colSolidWorldMin = 0u, colWorldMaxRuns = 0u; // header.y, header.z
	uint4 queue = { 0u, 0u, 0u, 0u };
	float curDistLast = 0.0f, curDistNext = 0.0f;
	int curScale = 1;
	uint32_t worldColumnColorsOff = 0u, columnRunsOff = 0u;
	float worldBoundsMin = 0.0f, worldBoundsMax = 0.0f;
	// the run being projected
	int solidIndex = 0;
	int elementLength = 0, elementColorsIndex = 0;
	float elementBoundsMin = 0.0f, elementBoundsMax = 0.0f;
	// SIDE -> SPIX
	float boundsX = 0.0f, boundsY = 0.0f, uvAx = 0.0f, uvAy = 0.0f, uvBx = 0.0f, uvBy = 0.0f;
	int rbMinS = 0, rbMaxS = 0, wS = 0;
	uint32_t todoS = 0u;
	// SIDE -> TB
	f3 secB = { 0.0f, 0.0f, 0.0f };
	float secBQuotient = 0.0f;
	bool secBHasQuotient = false, faceTop = false, faceWanted = false;
	uint32_t secondaryColor = 0u;

	// ---- DDASetupJob.Execute :58-76, TraceToFirstColumnJob.Execute :95-143, SetupProjectedPlaneParams :622-651 (as trace_ray) ----
	if (active) {
		bool alive = true;
		{
			float endRayLerp = (float)planeRayIndex / (float)S.rayCount;
			float dx = m_lerp(S.rayMinX, S.rayMaxX, endRayLerp);
			float dz = m_lerp(S.rayMinZ, S.rayMaxZ, endRayLerp);
			float r = 1.0f / sqrtf(dx * dx + dz * dz);
			dda_init(ray, F.posX, F.posZ, r * dx, r * dz);
		}
		dirXNonNegative = ray.dirX >= 0.0f;
		dirZNonNegative = ray.dirZ >= 0.0f;
		lodMax = F.lod[0];
		const int dimX = world->dimX, dimZ = world->dimZ;
		if (ray.px < 0 || ray.pz < 0 || ray.px >= dimX || ray.pz >= dimZ) {
			if (!dda_step_to_world_intersection(ray, (float)dimX, (float)dimZ)) {
				alive = false; // WriteSkyboxFull
			} else {
				while (ray.distLast >= lodMax && lod < 5) {
					dda_next_lod(ray, 1 << lod, dirXNonNegative, dirZNonNegative);
					lod++;
					lodMax = F.lod[lod];
				}
				if (m_min(ray.tMaxX, ray.tMaxZ) >= farClip) {
					alive = false;
				}
			}
		}
		if (alive) {
			L = world->level[lod];
			const float *M = F.M;
			const int r0 = S.axisMappedToY ? 1 : 0;
			const float sx = ray.startX, sz = ray.startZ;
			planeStartTop.x = M[0 + r0] * sx + M[4 + r0] * worldMaxY + M[8 + r0] * sz + M[12 + r0] * 1.0f;
			planeStartTop.y = M[2] * sx + M[6] * worldMaxY + M[10] * sz + M[14] * 1.0f;
			planeStartTop.z = M[3] * sx + M[7] * worldMaxY + M[11] * sz + M[15] * 1.0f;
			planeStartBottom.x = M[0 + r0] * sx + M[4 + r0] * 0.0f + M[8 + r0] * sz + M[12 + r0] * 1.0f;
			planeStartBottom.y = M[2] * sx + M[6] * 0.0f + M[10] * sz + M[14] * 1.0f;
			planeStartBottom.z = M[3] * sx + M[7] * 0.0f + M[11] * sz + M[15] * 1.0f;
			planeDir.x = M[0 + r0] * ray.dirX + M[4 + r0] * 0.0f + M[8 + r0] * ray.dirZ + M[12 + r0] * 0.0f;
			planeDir.y = M[2] * ray.dirX + M[6] * 0.0f + M[10] * ray.dirZ + M[14] * 0.0f;
			planeDir.z = M[3] * ray.dirX + M[7] * 0.0f + M[11] * ray.dirZ + M[15] * 0.0f;
			guardSteps = dimX + dimZ + 16; // (see trace_ray)
			// column 0: LOD check (:237-243), bounds test and fetch (World.GetVoxelColumn, World.cs:130-142)
			if (ray.distLast >= lodMax && lod < 5) {
				dda_next_lod(ray, 1 << lod, dirXNonNegative, dirZNonNegative);
				lod++;
				L = world->level[lod];
				lodMax = F.lod[lod];
			}
			if ((ray.px & maskX) != ray.px || (ray.pz & maskZ) != ray.pz) {
				alive = false; // out of world bounds -> WriteSkybox
			} else {
				const uint32_t rec = L.recordsOff + record_offset(ray.px >> L.shift, ray.pz >> L.shift, L.rowShift);
				nextHeader = ld4(arena, rec);
				nextQueue = ld4(arena, rec + 16u);
			}
		}
		if (alive) {
			state = ST_ADV;
		}
	}

	// camSpaceMin / camSpaceMax of the column's near (Last) or far (Next) intersection, :289-293.  Recomputed by the blocks that
	// need them (3 multiplications + 6 additions) instead of being held in 12 registers across all blocks.
	auto camSpacePair = [&](float dist, f3 &cMin, f3 &cMax) {
		cMin = f3_madd(planeStartBottom, planeDir, dist);
		cMax = f3_madd(planeStartTop, planeDir, dist);
	};

	const int tAdv = thresholdOverride > 0 ? thresholdOverride : CVX_SM_T_ADV;
	const int tClip = thresholdOverride > 0 ? thresholdOverride : CVX_SM_T_CLIP;
	const int tWalk = thresholdOverride > 0 ? thresholdOverride : CVX_SM_T_WALK;
	const int tSide = thresholdOverride > 0 ? thresholdOverride : CVX_SM_T_SIDE;
	const int tSpix = thresholdOverride > 0 ? thresholdOverride : CVX_SM_T_SPIX;
	const int tTb = thresholdOverride > 0 ? thresholdOverride : CVX_SM_T_TB;
	int forcedNeed = 65; // <= 64: no block reached its threshold in the last pass, the fullest one(s) run now

#define CVX_SM_WAITING(s) ((int)__popcll(__ballot(state == (s))))
#define CVX_SM_RUNS(s, t) (CVX_SM_WAITING(s) >= min((t), forcedNeed))

#ifdef CVX_SM_STATS
	unsigned smEx[6] = { 0, 0, 0, 0, 0, 0 }, smLn[6] = { 0, 0, 0, 0, 0, 0 }, smPasses = 0, smForced = 0;
#endif
	while (true) {
		bool ran = false;
#ifdef CVX_SM_STATS
		smPasses++;
#endif

		// ================================================================================================ ADV
		if (CVX_SM_RUNS(ST_ADV, tAdv)) {
			ran = true;
			CVX_SM_STAT(ST_ADV);
			if (state == ST_ADV) {
				if (pendingEnd || --guardSteps <= 0) {
					state = ST_DONE; // far clip reached / left the world: WriteSkybox
				} else {
					// the record of the column the DDA stands on has been in flight since the previous ADV of this lane
					const uint4 header = nextHeader;
					queue = nextQueue;
					colSolidWorldMin = header.y;
					colWorldMaxRuns = header.z;
					worldColumnColorsOff = L.elementsOff + header.x * 4u; // ColorPointer, World.cs:185
					columnRunsOff = L.runsOff + header.w * 8u;            // solid run j >= 2 (top-down numbering) lives at entry j - 2
					curDistLast = ray.distLast;
					curDistNext = ray.distNext;
					curScale = 1 << lod;
					// ---- look ahead: Step (:613 / :252 / :273), LOD check of the next iteration (:237-243), fetch
					const bool lastColumn = dda_step(ray, farClip);
					if (ray.distLast >= lodMax && lod < 5) {
						dda_next_lod(ray, 1 << lod, dirXNonNegative, dirZNonNegative);
						lod++;
						L = world->level[lod];
						lodMax = F.lod[lod];
					}
					const bool nextOutside = (ray.px & maskX) != ray.px || (ray.pz & maskZ) != ray.pz;
					const uint32_t rec = L.recordsOff + record_offset((ray.px & maskX) >> L.shift, (ray.pz & maskZ) >> L.shift, L.rowShift); // clamped into the table
					nextHeader = ld4(arena, rec);
					nextQueue = ld4(arena, rec + 16u);
					pendingEnd = lastColumn || nextOutside;

					// ---- the current column: empty (:251-256) or outside the writable world bounds (:261-281)?
					if ((colWorldMaxRuns >> 16) != 0u) {
						bool draw = true;
						worldBoundsMin = 0.0f;
						worldBoundsMax = worldMaxY;
						if (frustumDirMaxWorld != CVX_FLOAT_EPSILON) {
							const float columnWorldMin = (float)(colSolidWorldMin >> 16);
							const float columnWorldMax = (float)(colWorldMaxRuns & 0xFFFFu);
							float distTop = frustumDirMaxWorld > 0.0f ? curDistNext : curDistLast;
							float distBot = frustumDirMinWorld < 0.0f ? curDistNext : curDistLast;
							float newMax = posY + frustumDirMaxWorld * distTop;
							float newMin = posY + frustumDirMinWorld * distBot;
							if (newMin > worldBoundsMax || newMax < worldBoundsMin) {
								state = ST_DONE; // frustum left the world entirely
								draw = false;
							} else if (columnWorldMin > newMax || columnWorldMax < newMin) {
								draw = false;
							} else {
								worldBoundsMin = newMin;
								worldBoundsMax = newMax;
							}
						}
						if (draw) {
							solidIndex = 0;
							state = (curDistLast > 2.0f && frustumDirMaxWorld == CVX_FLOAT_EPSILON) ? ST_CLIP : ST_WALK; // :295
						}
					}
				}
			}
		}

		// ================================================================================================ CLIP (:295-422)
		if (CVX_SM_RUNS(ST_CLIP, tClip)) {
			ran = true;
			CVX_SM_STAT(ST_CLIP);
			if (state == ST_CLIP) {
				f3 camSpaceMinLast, camSpaceMaxLast, camSpaceMinNext, camSpaceMaxNext;
				camSpacePair(curDistLast, camSpaceMinLast, camSpaceMaxLast);
				camSpacePair(curDistNext, camSpaceMinNext, camSpaceMaxNext);
				float clipLastMinLerp, clipLastMaxLerp, clipNextMinLerp, clipNextMaxLerp;
				bool straddlesUnused_;
				const float invFrustumMin = quot_safe(1.0f, recip_safe(frustumBoundsMin)), invFrustumMax = quot_safe(1.0f, recip_safe(frustumBoundsMax));
				const bool clippedLast = clip_world_bounds(camSpaceMinLast, camSpaceMaxLast, frustumBoundsMin, frustumBoundsMax, invFrustumMin, invFrustumMax, clipLastMinLerp, clipLastMaxLerp, straddlesUnused_);
				const bool clippedNext = clip_world_bounds(camSpaceMinNext, camSpaceMaxNext, frustumBoundsMin, frustumBoundsMax, invFrustumMin, invFrustumMax, clipNextMinLerp, clipNextMaxLerp, straddlesUnused_);
				const bool minFromLast = !clippedLast && (clippedNext || clipLastMinLerp < clipNextMinLerp);
				const bool maxFromLast = !clippedLast && (clippedNext || clipLastMaxLerp > clipNextMaxLerp);
				worldBoundsMin = m_lerp(0.0f, worldMaxY, minFromLast ? clipLastMinLerp : clipNextMinLerp);
				worldBoundsMax = m_lerp(0.0f, worldMaxY, maxFromLast ? clipLastMaxLerp : clipNextMaxLerp);
				frustumDirMinWorld = (worldBoundsMin - posY) / (minFromLast ? curDistLast : curDistNext);
				frustumDirMaxWorld = (worldBoundsMax - posY) / (maxFromLast ? curDistLast : curDistNext);
				const f3 minClipA = f3_lerp(camSpaceMinLast, camSpaceMaxLast, clipLastMinLerp);
				const f3 maxClipA = f3_lerp(camSpaceMinLast, camSpaceMaxLast, clipLastMaxLerp);
				const f3 minClipB = f3_lerp(camSpaceMinNext, camSpaceMaxNext, clipNextMinLerp);
				const f3 maxClipB = f3_lerp(camSpaceMinNext, camSpaceMaxNext, clipNextMaxLerp);
				float minNext = minClipB.x / minClipB.z;
				float minLast = minClipA.x / minClipA.z;
				float maxNext = maxClipB.x / maxClipB.z;
				float maxLast = maxClipA.x / maxClipA.z;
				if (maxNext < minNext) { float t = maxNext; maxNext = minNext; minNext = t; }
				if (maxLast < minLast) { float t = maxLast; maxLast = minLast; minLast = t; }
				const float camSpaceClippedMin = clippedLast ? minNext : (clippedNext ? minLast : hw_min(minLast, minNext));
				const float camSpaceClippedMax = clippedLast ? maxNext : (clippedNext ? maxLast : hw_max(maxLast, maxNext));
				worldBoundsMin = floorf(worldBoundsMin);
				worldBoundsMax = ceilf(worldBoundsMax);
				const int writableMinPixel = f2i_floor(camSpaceClippedMin);
				const int writableMaxPixel = f2i(ceilf(camSpaceClippedMax));
				if ((clippedLast && clippedNext) || writableMaxPixel < nextFreePixelMin || writableMinPixel > nextFreePixelMax) {
					state = ST_DONE;
				} else {
					if (writableMinPixel > nextFreePixelMin) {
						nextFreePixelMin = scan_up(seen, sshift, writableMinPixel, omax);
					}
					if (writableMaxPixel < nextFreePixelMax) {
						nextFreePixelMax = scan_down(seen, sshift, writableMaxPixel, omin);
					}
					state = ST_WALK; // (a window closed here is noticed at the end of the column, like in render_kernel<false>)
				}
			}
		}

		// ================================================================================================ WALK (:424-475)
		if (CVX_SM_RUNS(ST_WALK, tWalk)) {
			ran = true;
			CVX_SM_STAT(ST_WALK);
			if (state == ST_WALK) {
				const int solidCount = (int)(colSolidWorldMin & 0xFFFFu);
				bool found = false;
				while (solidIndex < solidCount) {
					const int j = DIR > 0 ? solidIndex : solidCount - 1 - solidIndex;
					uint32_t w0, w1;
					if (j < 2) {
						const bool odd = j != 0;
						w0 = odd ? queue.z : queue.x;
						w1 = odd ? queue.w : queue.y;
					} else {
						const uint2 run = ld2(arena, columnRunsOff + (uint32_t)(j - 2) * 8u);
						w0 = run.x;
						w1 = run.y;
					}
					solidIndex++;
					elementLength = (int)(w0 >> 16);
					elementColorsIndex = (int)(w1 & 0xFFFFu);
					const int top = worldMaxYInt - (int)(w0 & 0xFFFFu) * curScale;
					elementBoundsMax = (float)top;
					elementBoundsMin = (float)(top - elementLength * curScale);
					if (elementBoundsMin > worldBoundsMax) {
						if (DIR < 0) { solidIndex = solidCount + 1; break; } else { continue; }
					}
					if (elementBoundsMax < worldBoundsMin) {
						if (DIR > 0) { solidIndex = solidCount + 1; break; } else { continue; }
					}
					found = true;
					break;
				}
				// no run left: the column is done; :537,606 -- the ray ends once every pixel of its window is written (tested once per column, see trace_ray)
				state = found ? ST_SIDE : (nextFreePixelMin <= nextFreePixelMax ? ST_ADV : ST_DONE);
			}
		}

		// ================================================================================================ SIDE (:477-517)
		if (CVX_SM_RUNS(ST_SIDE, tSide)) {
			ran = true;
			CVX_SM_STAT(ST_SIDE);
			if (state == ST_SIDE) {
				f3 camSpaceMinLast, camSpaceMaxLast;
				camSpacePair(curDistLast, camSpaceMinLast, camSpaceMaxLast);
				const float portionBottom = elementBoundsMin * invWorldMaxY;
				const float portionTop = elementBoundsMax * invWorldMaxY;
				f3 camSpaceFrontBottom = f3_lerp(camSpaceMinLast, camSpaceMaxLast, portionBottom);
				f3 camSpaceFrontTop = f3_lerp(camSpaceMinLast, camSpaceMaxLast, portionTop);
				faceTop = portionTop < cameraPosYNormalized;
				const bool faceBottom = !faceTop && portionBottom > cameraPosYNormalized;
				faceWanted = faceTop ? !(elementBoundsMax > worldBoundsMax) : (faceBottom && !(elementBoundsMin < worldBoundsMin));
				secondaryColor = ld_color(arena, worldColumnColorsOff + (uint32_t)(faceTop ? elementColorsIndex : elementColorsIndex + elementLength - 1) * 4u);

				float uA = (float)elementLength;
				float uB = 0.0f;
				bool visible = true; // ClipHomogeneousCameraSpaceLine with u, CameraData.cs:141-157
				bool nearClipped = false;
				if (camSpaceFrontBottom.y <= 0.0f) {
					if (camSpaceFrontTop.y <= 0.0f) {
						visible = false;
					} else {
						float v = camSpaceFrontTop.y / (camSpaceFrontTop.y - camSpaceFrontBottom.y);
						camSpaceFrontBottom = f3_lerp(camSpaceFrontTop, camSpaceFrontBottom, v);
						uA = m_lerp(uB, uA, v);
						nearClipped = true;
					}
				} else if (camSpaceFrontTop.y <= 0.0f) {
					float v = camSpaceFrontBottom.y / (camSpaceFrontBottom.y - camSpaceFrontTop.y);
					camSpaceFrontTop = f3_lerp(camSpaceFrontBottom, camSpaceFrontTop, v);
					uB = m_lerp(uA, uB, v);
					nearClipped = true;
				}
				secB = faceTop ? camSpaceFrontTop : camSpaceFrontBottom;
				secBHasQuotient = visible;
				bool pixels = false;
				if (visible) {
					float frontBottomQuotient, frontTopQuotient;
					if (!nearClipped && div_safe(camSpaceFrontBottom.z) && div_safe(camSpaceFrontTop.z) && div_safe(camSpaceFrontBottom.x) && div_safe(camSpaceFrontTop.x)) {
						const Recip rb = recip_safe(camSpaceFrontBottom.z), rt = recip_safe(camSpaceFrontTop.z);
						uvAx = quot_safe(1.0f, rb);
						uvAy = quot_safe(uA, rb);
						frontBottomQuotient = quot_safe(camSpaceFrontBottom.x, rb);
						uvBx = quot_safe(1.0f, rt);
						uvBy = __int_as_float(__float_as_int(camSpaceFrontTop.z) & (int)0x80000000); // +0 / z: a zero with the sign of z
						frontTopQuotient = quot_safe(camSpaceFrontTop.x, rt);
					} else {
						uvAx = 1.0f / camSpaceFrontBottom.z;
						uvAy = uA / camSpaceFrontBottom.z;
						uvBx = 1.0f / camSpaceFrontTop.z;
						uvBy = uB / camSpaceFrontTop.z;
						frontBottomQuotient = camSpaceFrontBottom.x / camSpaceFrontBottom.z;
						frontTopQuotient = camSpaceFrontTop.x / camSpaceFrontTop.z;
					}
					secBQuotient = faceTop ? frontTopQuotient : frontBottomQuotient;
					boundsX = frontBottomQuotient;
					boundsY = frontTopQuotient;
					if (boundsX > boundsY) {
						float t = boundsX; boundsX = boundsY; boundsY = t;
						t = uvAx; uvAx = uvBx; uvBx = t;
						t = uvAy; uvAy = uvBy; uvBy = t;
					}
					int rbMin = f2i(rintf(boundsX));
					int rbMax = f2i(rintf(boundsY));
					if (rbMax >= nextFreePixelMin && rbMin <= nextFreePixelMax) {
						reduce_pixel_horizon(seen, sshift, omin, omax, rbMin, rbMax, nextFreePixelMin, nextFreePixelMax, frustumBoundsMin, frustumBoundsMax);
						// first mask word of [rbMin, rbMax] that has an unseen pixel (:519)
						int w = rbMin >> 5;
						const int wEnd = rbMax >> 5;
						uint32_t todo = 0u;
						while (w <= wEnd) {
							const uint32_t range = range_mask(w, rbMin, rbMax);
							const uint32_t m = seen[w << sshift];
							todo = ~m & range;
							if (todo != 0u) {
								seen[w << sshift] = m | range;
								frustumDirMaxWorld = CVX_FLOAT_EPSILON;
								break;
							}
							w++;
						}
						if (todo != 0u) {
							pixels = true;
							rbMinS = rbMin;
							rbMaxS = rbMax;
							wS = w;
							todoS = todo;
						}
					}
				}
				state = pixels ? ST_SPIX : (faceWanted ? ST_TB : ST_WALK);
			}
		}

		// ================================================================================================ SPIX (:519-542)
		if (CVX_SM_RUNS(ST_SPIX, tSpix)) {
			ran = true;
			CVX_SM_STAT(ST_SPIX);
			if (state == ST_SPIX) {
				auto colourOffset = [&](int y) -> uint32_t { // perspective-correct colour of pixel y of the run's side, :524-531
					float l = ((float)y - boundsX) / (boundsY - boundsX); // unlerp
					float wux = m_lerp(uvAx, uvBx, l);
					float wuy = m_lerp(uvAy, uvBy, l);
					float u = wuy / wux;
					int colorIdx = m_clampi(f2i_floor(u), 0, elementLength - 1) + elementColorsIndex;
					return worldColumnColorsOff + (uint32_t)colorIdx * 4u;
				};
				int trips = CVX_SM_SPIX_TRIPS;
				const int wEnd = rbMaxS >> 5;
				while (true) {
					// two pixels per trip (both colour loads in flight before the first store waits for its colour)
					const int y0 = (wS << 5) + (__ffs((int)todoS) - 1);
					todoS &= todoS - 1u;
					const uint32_t c0 = ld_color(arena, colourOffset(y0));
					const bool second = todoS != 0u;
					int y1 = y0;
					uint32_t c1 = 0u;
					if (second) {
						y1 = (wS << 5) + (__ffs((int)todoS) - 1);
						todoS &= todoS - 1u;
						c1 = ld_color(arena, colourOffset(y1));
					}
					st_pixel_loop(tileOut, laneByteOff, y0, c0);
					if (second) {
						st_pixel_loop(tileOut, laneByteOff, y1, c1);
					}
					if (todoS == 0u) { // next mask word with an unseen pixel
						wS++;
						while (wS <= wEnd) {
							const uint32_t range = range_mask(wS, rbMinS, rbMaxS);
							const uint32_t m = seen[wS << sshift];
							todoS = ~m & range;
							if (todoS != 0u) {
								seen[wS << sshift] = m | range;
								break;
							}
							wS++;
						}
						if (todoS == 0u) {
							state = faceWanted ? ST_TB : ST_WALK;
							break;
						}
					}
					if (--trips <= 0) {
						break;
					}
				}
			}
		}

		// ================================================================================================ TB (:544-610)
		if (CVX_SM_RUNS(ST_TB, tTb)) {
			ran = true;
			CVX_SM_STAT(ST_TB);
			if (state == ST_TB) {
				f3 camSpaceMinNext, camSpaceMaxNext;
				camSpacePair(curDistNext, camSpaceMinNext, camSpaceMaxNext);
				const float portion = (faceTop ? elementBoundsMax : elementBoundsMin) * invWorldMaxY;
				f3 secA = f3_lerp(camSpaceMinNext, camSpaceMaxNext, portion);
				f3 sB = secB;
				bool visible = true; // ClipHomogeneousCameraSpaceLine, CameraData.cs:124-138
				bool secBKept = secBHasQuotient;
				if (secA.y <= 0.0f) {
					if (sB.y <= 0.0f) {
						visible = false;
					} else {
						float v = sB.y / (sB.y - secA.y);
						secA = f3_lerp(sB, secA, v);
					}
				} else if (sB.y <= 0.0f) {
					float v = secA.y / (secA.y - sB.y);
					sB = f3_lerp(secA, sB, v);
					secBKept = false;
				}
				if (visible) {
					float bx = rintf(secA.x / secA.z);
					float q = secBQuotient;
					if (!secBKept) {
						q = sB.x / sB.z;
					}
					float by = rintf(q);
					int rbMin = f2i(bx);
					int rbMax = f2i(by);
					if (rbMin > rbMax) {
						int t = rbMin; rbMin = rbMax; rbMax = t;
					}
					if (rbMax >= nextFreePixelMin && rbMin <= nextFreePixelMax) {
						reduce_pixel_horizon(seen, sshift, omin, omax, rbMin, rbMax, nextFreePixelMin, nextFreePixelMax, frustumBoundsMin, frustumBoundsMax);
						for (int w = rbMin >> 5; w <= (rbMax >> 5); w++) { // :595-603
							const uint32_t range = range_mask(w, rbMin, rbMax);
							const uint32_t m = seen[w << sshift];
							uint32_t todo = ~m & range;
							if (todo != 0u) {
								seen[w << sshift] = m | range;
								frustumDirMaxWorld = CVX_FLOAT_EPSILON;
								do {
									const int y = (w << 5) + (__ffs((int)todo) - 1);
									todo &= todo - 1u;
									st_pixel_loop(tileOut, laneByteOff, y, secondaryColor);
								} while (todo != 0u);
							}
						}
					}
				}
				state = ST_WALK;
			}
		}

		// ---- scheduling: nothing reached its threshold -> the fullest block runs in the next pass ----
		if (ran) {
			forcedNeed = 65;
		} else {
			int m = CVX_SM_WAITING(ST_ADV);
			m = max(m, CVX_SM_WAITING(ST_CLIP));
			m = max(m, CVX_SM_WAITING(ST_WALK));
			m = max(m, CVX_SM_WAITING(ST_SIDE));
			m = max(m, CVX_SM_WAITING(ST_SPIX));
			m = max(m, CVX_SM_WAITING(ST_TB));
			if (m == 0) {
				break; // every ray of the wave is finished
			}
			forcedNeed = m;
#ifdef CVX_SM_STATS
			smForced++;
#endif
		}
	}
#ifdef CVX_SM_STATS
	if ((threadIdx.x & 63) == 0) {
		for (int k = 0; k < 6; k++) { atomicAdd(&g_sectionCycles[k], (unsigned long long)smEx[k]); atomicAdd(&g_sectionCycles[8 + k], (unsigned long long)smLn[k]); }
		atomicAdd(&g_sectionCycles[6], (unsigned long long)smPasses);
		atomicAdd(&g_sectionCycles[7], (unsigned long long)smForced);
		atomicAdd(&g_sectionCycles[14], 1ull);
	}
#endif
#undef CVX_SM_WAITING
#undef CVX_SM_RUNS
}

// ---------------------------------------------------------------------------
// render kernel, state-machine form: same grid, LDS and tile conventions as render_kernel
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(CVX_WAVE, CVX_WAVES_PER_SIMD) void render_sm_kernel(const DevFrame *__restrict__ frames, const DevTile *__restrict__ tiles,
                                                                               const DevWorld *__restrict__ world, SmParams params)
{
	extern __shared__ uint32_t lds[];
	const int lane = threadIdx.x;
	const DevTile tile = tiles[blockIdx.x];
	const DevFrame &F = frames[tile.frame];
	const DevSegment &S = F.seg[tile.seg];

	const int omin = S.omin, omax = S.omax;
	const int wordBase = omin >> 5;
	const int words = (omax >> 5) - wordBase + 1;
	const int firstLane = tile.lanes & 0xFF, laneCount = tile.lanes ? (tile.lanes >> 8) & 0xFF : CVX_WAVE;
	const int sshift = 31 - __clz(laneCount);
	const int planeRayIndex = tile.tileInSeg * CVX_WAVE + firstLane + lane;
	const bool active = lane < laneCount && planeRayIndex < S.rayCount;
	if (lane < laneCount) {
		for (int w = 0; w < words; w++) {
			lds[(w << sshift) + lane] = 0u;
		}
	}
	const gptr_tile tileOut = (gptr_tile)tile.out;
	const uint32_t laneByteOff = (uint32_t)(firstLane + lane) * 4u;
	uint32_t *seen = lds + lane - (wordBase << sshift);

	if (F.inverse) {
		trace_wave_sm<-1>(F, S, world, planeRayIndex, active, seen, sshift, tileOut, laneByteOff, params.threshold);
	} else {
		trace_wave_sm<1>(F, S, world, planeRayIndex, active, seen, sshift, tileOut, laneByteOff, params.threshold);
	}

	// WriteSkybox / WriteSkyboxFull (:699-716) for the whole wave
	for (int w = omin >> 5; w <= (omax >> 5); w++) {
		uint32_t todo = 0u;
		if (active) { todo = ~seen[w << sshift] & range_mask(w, omin, omax); }
		const int base = w << 5;
#pragma unroll 4
		for (int b = 0; b < 32; b++) {
			if ((todo >> b) & 1u) {
				st_pixel(tileOut, laneByteOff, base + b, CVX_SKYBOX_ARGB);
			}
		}
	}
}

} // namespace cvxk
