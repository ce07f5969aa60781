// cvx_lone.h -- the LATENCY kernel of libcpuvox_gpu (gfx950 / CDNA4, wave64): one wavefront per RAY, lanes = the next 64 COLUMNS of that ray.
//
// The reference's caller issues one blocking DrawSegments per frame (UnityManager.cs:182, RenderManager.cs:358-363): 6000 rays at 1080p on a chip with
// 1024 SIMDs.  render_kernel (cvx_kernels.h, lanes = rays) is built for batches; for a single frame every wave ends up with one or two rays and
// executes the whole column loop -- DDA step, cull, clip, run projection, pixel loops -- once per column with 62 idle (duplicate) lanes.  A frame then
// takes as long as the instruction stream of its longest ray.
//
// Here the 64 lanes of a wave hold 64 CONSECUTIVE COLUMNS of one ray (a "window"), and everything of ExecuteRay (DrawSegmentRayJob.cs:195-620) that
// does not depend on the ray's evolving pixel state is computed for the whole window at once:
//   * the DDA sequence (SegmentDDAData.Step, SegmentDDAData.cs:135-150): the crossings of the x planes and of the z planes are two chains of the reference's
//     own additions (lane n holds the n-th sum of each: 63 DPP wave-shift additions per chain), merged by rank (7 ds_bpermute steps); lane k gets column k;
//   * the 64 column records of the window: ONE round trip to memory instead of one per column;
//   * the Q corners (:289-293) and, for each of the up to three solid runs a record holds, the whole side / face projection (:478-502, :566-578): near
//     clip, the six quotients, the rounded pixel bounds and the texture generators -- pure functions of the column's two distances and the run's span;
//   * the colour of each run's top / bottom face (:553,560).
// What remains serial is the part that reads and writes the ray's state -- the seen mask, nextFreePixelMin / Max, frustumBounds, the cached frustum
// directions (:214-221) -- and it is organised by EVENTS, not by columns:
//   * while the frustum directions are valid (no pixel written since the last clip, :261), the cull (:261-281) and the overlap tests of all runs
//     (:461-475, :505, :581) are evaluated for ALL remaining columns of the window in one pass; a ballot + s_ff1 finds the first column that can
//     touch the state; the columns before it provably change nothing (they are culled, or none of their runs overlaps [nextFreePixelMin, Max]);
//   * that column is processed exactly as the reference does, run by run, with wave-uniform scalars read from its lane (v_readlane);
//   * after a pixel write the directions are invalid (:522,598): the next non-empty column clips (:295-422) -- its four division chains in the four lanes
//     of a quad -- and the pass over the remaining columns is repeated with the new directions.
// Boolean algebra of the pass and the clip is written on BALLOTS (64-bit scalars): selects and merges of per-lane booleans would be made in vector registers.
// The seen mask (:208) is ONE WORD PER LANE in a vector register (two for windows of more than 2048 pixels): horizon scans (:407-414, :678-692) are a
// ballot over "my word has an unseen bit in range" + s_ff1 / s_flbit + one v_readlane (first: one look at the word the scan usually ends in); the pixel
// loops (:519-533, :595-603) run with lane = PIXEL; a range inside one mask word (the usual few pixels) takes its word out with v_readlane and puts it back
// with v_writelane; the mask clear is free and the skybox is the initial value of the ray's pixel row in LDS (written out once, at the end).
//
// Arithmetic: the SAME device functions as render_kernel (cvx_kernels.h) on the same values in the same order -- the contract (IEEE binary32, no
// contraction, x86 cvttss2si) is shared; results are bit-identical to render_kernel and to the CPU oracle (tests/test_gpu_parity.py, tools/soak.py).
#pragma once

#include <type_traits>

#include "cvx_kernels.h"

namespace cvxk {

typedef unsigned long long lanemask_t;

// diagnostic build only (-DCVX_LONE_STATS, tools/lone_stats.py): how often each part of the latency kernel runs per launch
#ifdef CVX_LONE_MARK /* static accounting only (tools/lone_static.py): comment markers in the assembly around the kernel's parts */
#define CVX_LMARK(name) asm volatile("; LMARK " name ::: "memory")
#else
#define CVX_LMARK(name) ((void)0)
#endif
#ifdef CVX_LONE_STATS
__device__ unsigned long long g_loneStats[48];
__device__ unsigned long long g_loneLongest[48];
#define CVX_LSTAT(n) (stat_[n]++)
#define CVX_LSTAT_ADD(n, v) (stat_[n] += (unsigned int)(v))
// section clock (s_memtime): the cycles since the last switch go to the section that was current; stat_[32 + section]
#ifdef CVX_LONE_TIMES
#define CVX_LSEC(n) do { const unsigned int t_ = (unsigned int)__builtin_amdgcn_s_memtime(); stat_[32 + stat_[31]] += t_ - stat_[30]; stat_[30] = t_; stat_[31] = (n); } while (0)
#else
#define CVX_LSEC(n) ((void)0)
#endif
#if defined(CVX_LONE_TIMES) && CVX_LONE_TIMES >= 2 /* ... stamps inside the events too (they cost as much as what they measure: shares only) */
#define CVX_LSECE(n) CVX_LSEC(n)
#else
#define CVX_LSECE(n) ((void)0)
#endif
#else
#define CVX_LSTAT(n) ((void)0)
#define CVX_LSTAT_ADD(n, v) ((void)0)
#define CVX_LSEC(n) ((void)0)
#define CVX_LSECE(n) ((void)0)
#endif

__device__ __forceinline__ float rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rli(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t rlu(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// vec with lane l replaced by the wave-uniform value: v_writelane_b32 (no compiler builtin in this toolchain).  A gfx9 vector instruction reads ONE scalar
// register, so the lane select goes through M0 (written by a scalar instruction: no wait state -- those are for lane selects written by vector instructions)
// (M0 is the compiler's own: it loads M0 immediately before each of its uses -- here the LDS address of global_load_lds --, and the clobber tells it M0 is gone)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ uint32_t write_lane(uint32_t vec, uint32_t value, int l)
{
	asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(vec) : "s"(value), "s"(l) : "m0");
	return vec;
}
#pragma clang diagnostic pop
__device__ __forceinline__ lanemask_t lanes_from(int l) { return ~0ull << l; } // lanes >= l (l in 0..63)
// A wave-uniform value the compiler cannot PROVE uniform (the result of an inline-asm instruction -- f2i, hw_min -- counts as divergent, and everything
// computed from it, and every branch on that: exec-mask loops instead of scalar branches): read from the first lane, it is uniform by construction.
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// The DDA's two crossing chains (lone_trace_ray), one addition per step and chain for the whole wave: `v_add_f32 wave_shr:1` gives lane n the sum of lane n - 1's
// value and the step (lane 0, without a source lane, keeps its value), so after step s lane n holds its start value + min(n, s) additions of the step -- the
// reference's own sequence of rounded sums (SegmentDDAData.Step, SegmentDDAData.cs:135-150).  Called with X = tMax.x, Z = tMax.y in every lane.  (Written as
// lane >= k ? X + step : X the compiler keeps the 63 lane masks in scalar registers for the whole ray: 126 of them, spilled, two restores per step.)  A DPP
// operand needs two wait states after the vector write of its register: the other chain's addition and one s_nop are in between.  Needs all 64 lanes active.
__device__ __forceinline__ void lone_crossing_chains(float &X, float &Z, float stepX, float stepZ)
{
#define CVX_CHAIN_STEP "v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
#define CVX_CHAIN_STEPS_7 CVX_CHAIN_STEP CVX_CHAIN_STEP CVX_CHAIN_STEP CVX_CHAIN_STEP CVX_CHAIN_STEP CVX_CHAIN_STEP CVX_CHAIN_STEP
	asm("s_nop 1\n\t" CVX_CHAIN_STEPS_7 CVX_CHAIN_STEPS_7 CVX_CHAIN_STEPS_7 CVX_CHAIN_STEPS_7 CVX_CHAIN_STEPS_7 CVX_CHAIN_STEPS_7 CVX_CHAIN_STEPS_7 CVX_CHAIN_STEPS_7 CVX_CHAIN_STEPS_7
	    : "+v"(X), "+v"(Z)
	    : "v"(stepX), "v"(stepZ));
#undef CVX_CHAIN_STEPS_7
#undef CVX_CHAIN_STEP
}

// bits of mask word `w` (absolute word index) that fall inside the pixel range [lo, hi]; any w, lo, hi (empty ranges give 0)
__device__ __forceinline__ uint32_t range_mask_any(int w, int lo, int hi)
{
	const int base = w << 5;
	const int a = max(lo - base, 0), b = min(hi - base, 31); // first / last bit of the word inside the range
	const uint32_t m = (0xFFFFFFFFu << (a & 31)) & (0xFFFFFFFFu >> ((31 - b) & 31));
	return a > b ? 0u : m;
}

// The ray's seen-pixel mask: word (wordBase + lane) in w0, word (wordBase + 64 + lane) in w1 (HI instances only)
struct LoneSeen {
	uint32_t w0, w1;
	int wordBase, lane;
};

// mask word `i` (0 .. 127, relative to wordBase) as a wave-uniform scalar; words the instance does not hold read as "all seen"
template <bool HI>
__device__ __forceinline__ uint32_t lone_word(const LoneSeen &s, int i)
{
	const uint32_t a = rlu(s.w0, i & 63);
	if (!HI) { return i < 64 ? a : 0xFFFFFFFFu; }
	const uint32_t b = rlu(s.w1, i & 63);
	return i < 64 ? a : (i < 128 ? b : 0xFFFFFFFFu);
}

// ... the word of a pixel inside [omin, omax]: held by the instance by construction (cvx_gpu.hip picks HI for windows of more than 64 words), no clamp
template <bool HI>
__device__ __forceinline__ uint32_t lone_word_of_pixel(const LoneSeen &s, int y)
{
	const int i = (y >> 5) - s.wordBase;
	if (!HI) { return rlu(s.w0, i); }
	return lone_word<HI>(s, i);
}

// first unseen pixel >= start, or omax + 1; start unchanged when start > omax (the reference's while loop at :407 / :678 does not run then)
template <bool HI>
__device__ __forceinline__ int lone_scan_up(const LoneSeen &s, int start, int omax)
{
	if (CVX_RARE(start > omax)) { return start; }
	if (CVX_USUAL(((lone_word_of_pixel<HI>(s, start) >> (start & 31)) & 1u) == 0u)) { return start; } // (the usual case: the pixel right above the run is unseen)
	const uint32_t m0 = ~s.w0 & range_mask_any(s.wordBase + s.lane, start, omax);
	const lanemask_t b0 = __ballot(m0 != 0u);
	if (b0 != 0ull) {
		const int l = __ffsll((long long)b0) - 1;
		return ((s.wordBase + l) << 5) + (__ffs((int)rlu(m0, l)) - 1);
	}
	if (HI) {
		const uint32_t m1 = ~s.w1 & range_mask_any(s.wordBase + 64 + s.lane, start, omax);
		const lanemask_t b1 = __ballot(m1 != 0u);
		if (b1 != 0ull) {
			const int l = __ffsll((long long)b1) - 1;
			return ((s.wordBase + 64 + l) << 5) + (__ffs((int)rlu(m1, l)) - 1);
		}
	}
	return omax + 1;
}

// last unseen pixel <= start, or omin - 1; start unchanged when start < omin (:413 / :690)
template <bool HI>
__device__ __forceinline__ int lone_scan_down(const LoneSeen &s, int start, int omin)
{
	if (CVX_RARE(start < omin)) { return start; }
	if (CVX_USUAL(((lone_word_of_pixel<HI>(s, start) >> (start & 31)) & 1u) == 0u)) { return start; } // (the usual case: the pixel right below the run is unseen)
	if (HI) {
		const uint32_t m1 = ~s.w1 & range_mask_any(s.wordBase + 64 + s.lane, omin, start);
		const lanemask_t b1 = __ballot(m1 != 0u);
		if (b1 != 0ull) {
			const int l = 63 - __clzll((long long)b1);
			return ((s.wordBase + 64 + l) << 5) + (31 - __clz((int)rlu(m1, l)));
		}
	}
	const uint32_t m0 = ~s.w0 & range_mask_any(s.wordBase + s.lane, omin, start);
	const lanemask_t b0 = __ballot(m0 != 0u);
	if (b0 != 0ull) {
		const int l = 63 - __clzll((long long)b0);
		return ((s.wordBase + l) << 5) + (31 - __clz((int)rlu(m0, l)));
	}
	return omin - 1;
}

// ReducePixelHorizon, DrawSegmentRayJob.cs:660-697 (the scalar form of reduce_pixel_horizon in cvx_kernels.h: every operand is wave-uniform)
template <bool HI>
__device__ __forceinline__ void lone_reduce_pixel_horizon(const LoneSeen &s, int omin, int omax, int &rbMin, int &rbMax, int &nfMin, int &nfMax, float &frustumBoundsMin,
                                                          float &frustumBoundsMax)
{
	// (the callers have tested the overlap, rbMax >= nfMin && rbMin <= nfMax (:505 / :581), and a ray whose window is empty has ended: nfMin <= nfMax; so of the
	// reference's two conjunctions, `rbMin <= nfMin && rbMax >= nfMin` and `rbMax >= nfMax && max(rbMin, nfMin) <= nfMax`, the second halves are true)
	const bool raiseMin = rbMin <= nfMin;
	rbMin = max(rbMin, nfMin);
	if (raiseMin) {
		nfMin = lone_scan_up<HI>(s, rbMax + 1, omax);
		frustumBoundsMin = (float)nfMin - 0.501f;
	}
	const bool lowerMax = rbMax >= nfMax;
	rbMax = min(rbMax, nfMax);
	if (CVX_RARE(lowerMax)) { // (10 - 16 % of the reduces of the benchmark world; raiseMin: 41 - 45 %)
		nfMax = lone_scan_down<HI>(s, rbMin - 1, omin);
		frustumBoundsMax = (float)nfMax + 0.501f;
	}
}

// the pixels yb .. yb + 63 that are NOT yet seen, as a lane mask (bit p = pixel yb + p); yb >= omin
template <bool HI>
__device__ __forceinline__ lanemask_t lone_unseen64(const LoneSeen &s, int yb)
{
	const int i = (yb >> 5) - s.wordBase, sh = yb & 31;
	const unsigned long long a = lone_word<HI>(s, i), b = lone_word<HI>(s, i + 1), c = lone_word<HI>(s, i + 2);
	const unsigned long long lo = (a | (b << 32)) >> sh, hi = (b | (c << 32)) >> sh;
	return ~((lo & 0xFFFFFFFFull) | (hi << 32));
}

// ... of the n <= 64 pixels yb .. yb + n - 1 (the usual case, a few pixels inside one mask word, reads that one word)
template <bool HI>
__device__ __forceinline__ lanemask_t lone_unseen(const LoneSeen &s, int yb, int n)
{
	const int sh = yb & 31;
	if (CVX_USUAL(sh + n <= 32)) { return (lanemask_t)((~lone_word_of_pixel<HI>(s, yb) >> sh) & (0xFFFFFFFFu >> (32 - n))); }
	return lone_unseen64<HI>(s, yb) & (n >= CVX_WAVE ? ~0ull : ((1ull << n) - 1ull));
}

// marks the pixels [lo, hi] as seen (every word at once: lane = word)
template <bool HI>
__device__ __forceinline__ void lone_mark(LoneSeen &s, int lo, int hi)
{
	s.w0 |= range_mask_any(s.wordBase + s.lane, lo, hi);
	if (HI) { s.w1 |= range_mask_any(s.wordBase + 64 + s.lane, lo, hi); }
}

// ---- projection of one solid run of one column: everything of :478-502 (side) and :566-578 (face) that depends on nothing but the column's
// corners and the run's span.  The same operations in the same order as drawColumn (cvx_kernels.h); called once per run index for the 64 columns
// of a window (every operand per lane), and with wave-uniform operands for the columns of the run list.
struct RunProj {
	float boundsX, boundsY, uvAx, uvBx, uvAy, uvBy; // the side's two ends after the swap of :496-499 (rayBufferBoundsFloat, uvA, uvB)
	int rbMinS, rbMaxS;                             // :501-502
	int rbMinF, rbMaxF;                             // the face's pixel bounds, ordered (:570-578)
	bool sideVisible, faceNear;                     // the near-plane clips (CameraData.cs:124-157) left something of the side / the face
	bool faceTop, faceBottom;                       // :549-565 (which face the camera can see; `wanted` also needs the world bounds)
};

__device__ __forceinline__ RunProj project_run(f3 camSpaceMinLast, f3 camSpaceMaxLast, f3 camSpaceMinNext, f3 camSpaceMaxNext, float elementBoundsMin, float elementBoundsMax,
                                               int elementLength, float cameraPosYNormalized, float invWorldMaxY)
{
	RunProj P;
	const float portionBottom = elementBoundsMin * invWorldMaxY;
	const float portionTop = elementBoundsMax * invWorldMaxY;
	f3 camSpaceFrontBottom = f3_lerp(camSpaceMinLast, camSpaceMaxLast, portionBottom);
	f3 camSpaceFrontTop = f3_lerp(camSpaceMinLast, camSpaceMaxLast, portionTop);
	const bool faceTop = portionTop < cameraPosYNormalized;
	const bool faceBottom = ((int)!faceTop & (int)(portionBottom > cameraPosYNormalized)) != 0;
	f3 secA = f3_lerp(camSpaceMinNext, camSpaceMaxNext, faceTop ? portionTop : portionBottom);
	float frontBottomQuotient, frontTopQuotient, secBQuotient;
	bool faceVisible = true;
	float uA = (float)elementLength;
	float uB = 0.0f;
	bool visible = true;
	float uvAx, uvAy, uvBx, uvBy;
	{
		const Recip rb = recip_safe(camSpaceFrontBottom.z), rt = recip_safe(camSpaceFrontTop.z);
		uvAx = quot_safe(1.0f, rb);
		uvAy = quot_safe(uA, rb);
		frontBottomQuotient = quot_safe(camSpaceFrontBottom.x, rb);
		uvBx = quot_safe(1.0f, rt);
		uvBy = __int_as_float(__float_as_int(camSpaceFrontTop.z) & (int)0x80000000); // +0 / z
		frontTopQuotient = quot_safe(camSpaceFrontTop.x, rt);
	}
	secBQuotient = faceTop ? frontTopQuotient : frontBottomQuotient;
	const bool ordinary = ((int)!(camSpaceFrontBottom.y <= 0.0f) & (int)!(camSpaceFrontTop.y <= 0.0f) & (int)!(secA.y <= 0.0f) & (int)div_safe(camSpaceFrontBottom.z) & (int)div_safe(camSpaceFrontTop.z) &
	                       (int)div_safe(camSpaceFrontBottom.x) & (int)div_safe(camSpaceFrontTop.x)) != 0;
	if (CVX_RARE(!ordinary)) {
		if (camSpaceFrontBottom.y <= 0.0f) {
			if (camSpaceFrontTop.y <= 0.0f) {
				visible = false;
			} else {
				float v = camSpaceFrontTop.y / (camSpaceFrontTop.y - camSpaceFrontBottom.y);
				camSpaceFrontBottom = f3_lerp(camSpaceFrontTop, camSpaceFrontBottom, v);
				uA = m_lerp(uB, uA, v);
			}
		} else if (camSpaceFrontTop.y <= 0.0f) {
			float v = camSpaceFrontBottom.y / (camSpaceFrontBottom.y - camSpaceFrontTop.y);
			camSpaceFrontTop = f3_lerp(camSpaceFrontBottom, camSpaceFrontTop, v);
			uB = m_lerp(uA, uB, v);
		}
		uvAx = 1.0f / camSpaceFrontBottom.z;
		uvAy = uA / camSpaceFrontBottom.z;
		uvBx = 1.0f / camSpaceFrontTop.z;
		uvBy = uB / camSpaceFrontTop.z;
		frontBottomQuotient = camSpaceFrontBottom.x / camSpaceFrontBottom.z;
		frontTopQuotient = camSpaceFrontTop.x / camSpaceFrontTop.z;
		f3 secB = faceTop ? camSpaceFrontTop : camSpaceFrontBottom;
		if (secA.y <= 0.0f) {
			if (secB.y <= 0.0f) {
				faceVisible = false;
			} else {
				float v = secB.y / (secB.y - secA.y);
				secA = f3_lerp(secB, secA, v);
			}
		} else if (secB.y <= 0.0f) {
			float v = secA.y / (secA.y - secB.y);
			secB = f3_lerp(secA, secB, v);
		}
		secBQuotient = secB.x / secB.z;
	}
	{
		float boundsX = frontBottomQuotient;
		float boundsY = frontTopQuotient;
		const bool sw = boundsX > boundsY;
		P.boundsX = sw ? boundsY : boundsX;
		P.boundsY = sw ? boundsX : boundsY;
		P.uvAx = sw ? uvBx : uvAx;
		P.uvBx = sw ? uvAx : uvBx;
		P.uvAy = sw ? uvBy : uvAy;
		P.uvBy = sw ? uvAy : uvBy;
		P.rbMinS = f2i(rintf(P.boundsX));
		P.rbMaxS = f2i(rintf(P.boundsY));
	}
	{
		const float bx = rintf(secA.x / secA.z);
		const float by = rintf(secBQuotient);
		const int a = f2i(bx), b = f2i(by);
		P.rbMinF = min(a, b);
		P.rbMaxF = max(a, b);
	}
	P.sideVisible = visible;
	P.faceNear = faceVisible;
	P.faceTop = faceTop;
	P.faceBottom = faceBottom;
	return P;
}

// ---------------------------------------------------------------------------
// One ray: TraceToFirstColumnJob + ExecuteRay by one wave.
// ---------------------------------------------------------------------------
template <int DIR, bool HI>
__device__ __forceinline__ void lone_trace_ray(const DevFrame &F, const DevSegment &S, const DevWorld *__restrict__ world, int planeRayIndex,
                                               LoneSeen &seen, uint32_t *merged /* LDS: 64 words, then the ray's pixel row: pixel y at merged[64 + y - omin] */, unsigned int *stat_)
{
	(void)stat_;
	const int lane = seen.lane;
	const int omin = S.omin, omax = S.omax;
	const float farClip = F.farClip;
	const float posY = F.posY;

	// ---- DDASetupJob.Execute, :58-76 (wave-uniform)
	DDA ray;
	{
		float endRayLerp = (float)planeRayIndex / (float)S.rayCount;
		float dx = m_lerp(S.rayMinX, S.rayMaxX, endRayLerp);
		float dz = m_lerp(S.rayMinZ, S.rayMaxZ, endRayLerp);
		float r = 1.0f / sqrtf(dx * dx + dz * dz);
		dda_init(ray, F.posX, F.posZ, r * dx, r * dz);
	}
	const bool dirXNonNegative = ray.dirX >= 0.0f, dirZNonNegative = ray.dirZ >= 0.0f;

	// ---- TraceToFirstColumnJob.Execute, :95-143
	int lod = 0;
	float lodMax = F.lod[0];
	const int dimX = world->dimX, dimZ = world->dimZ;
	if (ray.px < 0 || ray.pz < 0 || ray.px >= dimX || ray.pz >= dimZ) {
		if (!dda_step_to_world_intersection(ray, (float)dimX, (float)dimZ)) {
			return; // WriteSkyboxFull
		}
		while (ray.distLast >= lodMax) {
			dda_next_lod(ray, 1 << lod, dirXNonNegative, dirZNonNegative);
			lod++;
			{ const float next_ = F.lod[min(lod, 5)]; lodMax = lod < 5 ? next_ : __builtin_inff(); }
		}
		if (m_min(ray.tMaxX, ray.tMaxZ) >= farClip) {
			return; // WriteSkyboxFull
		}
	}

	// (the ray's state is wave-uniform: say so, see uni())
	ray.px = uni(ray.px); ray.pz = uni(ray.pz); ray.sx = uni(ray.sx); ray.sz = uni(ray.sz);
	ray.startX = uni(ray.startX); ray.startZ = uni(ray.startZ); ray.dirX = uni(ray.dirX); ray.dirZ = uni(ray.dirZ);
	ray.tDeltaX = uni(ray.tDeltaX); ray.tDeltaZ = uni(ray.tDeltaZ); ray.tMaxX = uni(ray.tMaxX); ray.tMaxZ = uni(ray.tMaxZ);
	ray.distLast = uni(ray.distLast); ray.distNext = uni(ray.distNext);
	lod = uni(lod);
	lodMax = uni(lodMax);
	const int dirFlags = uni((dirXNonNegative ? 1 : 0) | (dirZNonNegative ? 2 : 0)); // (one scalar register instead of two lane masks for the life of the ray)

	// ---- ExecuteRay, :195-620
	int voxelScale = 1 << lod;
	DevWorldLevel L = world->level[lod];
	const gptr_arena arena = (gptr_arena)world->arena;
	const int maskX = world->maskX, maskZ = world->maskZ;
	const int worldMaxYInt = world->dimY;
	const float worldMaxY = (float)worldMaxYInt;
	const float cameraPosYNormalized = posY / worldMaxY;
	const float invWorldMaxY = 1.0f / worldMaxY;

	int nextFreePixelMin = omin;
	int nextFreePixelMax = omax;
	float frustumBoundsMin = (float)nextFreePixelMin - 0.501f;
	float frustumBoundsMax = (float)nextFreePixelMax + 0.501f;
	float frustumDirMaxWorld = CVX_FLOAT_EPSILON;
	float frustumDirMinWorld = CVX_FLOAT_EPSILON;

	f3 planeStartBottom, planeStartTop, planeDir; // SetupProjectedPlaneParams, :622-651
	{
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
	}

	// column 0: LOD check (:237-243), bounds test (World.GetVoxelColumn, World.cs:130-142)
	if (ray.distLast >= lodMax) {
		dda_next_lod(ray, voxelScale, (dirFlags & 1) != 0, (dirFlags & 2) != 0);
		lod++;
		voxelScale *= 2;
		L = world->level[lod];
		{ const float next_ = F.lod[min(lod, 5)]; lodMax = lod < 5 ? next_ : __builtin_inff(); }
	}
	if ((ray.px & maskX) != ray.px || (ray.pz & maskZ) != ray.pz) {
		return; // out of world bounds -> WriteSkybox
	}
	// the column as ONE integer x * 65536 + z and what a step adds to it (ColumnCursor of cvx_kernels.h, without the record address: lanes compute their own)
	int pos = ray.px * 65536 + ray.pz;
	int posStepX = ray.sx * 65536, posStepZ = ray.sz;
	const int outsideBits = ~((maskX << 16) | maskZ);
	// A DDA walk is monotone in x and z: it leaves the world after at most dimX + dimZ columns; the cap only keeps a wave from spinning on non-finite camera data
	int guardSteps = 2 * (dimX + dimZ + 16);

#ifdef CVX_LONE_STATS
	float lastClipBoundsMin_ = -1.0f, lastClipBoundsMax_ = -1.0f;
#endif
	bool alive = true; // false: the ray is finished (every exit of the reference is WriteSkybox, which the caller does)
	while (alive) {
		// ================= the window: the ray's next (up to) 64 columns at this level =================
		// SegmentDDAData.Step (:135-150) advances the axis with the smaller tMax by its tDelta: the crossing distances of the x planes, X[n] = tMax.x + n additions
		// of tDelta.x, and of the z planes, Z[m], are two chains that do not depend on each other; the DDA merges them (ties: z first, `tMax.x < tMax.y` :138).
		// So: lane n computes X[n] and Z[n] -- the reference's own additions in the reference's order, lane n simply stops after n of them --, a binary
		// search over the other chain gives every crossing its rank in the merged order, and the crossing of rank k is the one column k is LEFT through:
		// IntersectionDistances of column k = (crossing k - 1, crossing k) (:139-147).  ~7 instructions per column instead of the ~45 of stepping one by one.
		float wDistLast, wDistNext;
		int wPos;
		int count;
		int endCode = 0; // 0: 64 columns, more at this level; 1: the ray ends after this window (far clip :613 / left the world :246); 2: LOD boundary (:237-243)
		{
			CVX_LMARK("dda_begin");
			CVX_LSEC(1);
			float X = ray.tMaxX, Z = ray.tMaxZ;
			lone_crossing_chains(X, Z, ray.tDeltaX, ray.tDeltaZ);
			// rank of X[lane] = lane + #{m : Z[m] <= X[lane]}; rank of Z[lane] = lane + #{n : X[n] < Z[lane]}
			int belowX = 0, belowZ = 0;
#pragma unroll
			for (int step = CVX_WAVE; step >= 1; step >>= 1) {
				const int ix = belowX + step - 1, iz = belowZ + step - 1;
				const float zv = __int_as_float(__builtin_amdgcn_ds_bpermute((ix & 63) << 2, __float_as_int(Z)));
				const float xv = __int_as_float(__builtin_amdgcn_ds_bpermute((iz & 63) << 2, __float_as_int(X)));
				belowX += (ix < CVX_WAVE && zv <= X) ? step : 0;
				belowZ += (iz < CVX_WAVE && xv < Z) ? step : 0;
			}
			const int rankX = lane + belowX, rankZ = lane + belowZ;
			// the crossings in merged order: slot k of the wave's 64 words of LDS receives crossing k (an x crossing with its sign bit set: distances are >= 0)
			if (rankX < CVX_WAVE) { merged[rankX] = __float_as_uint(X) | 0x80000000u; }
			if (rankZ < CVX_WAVE) { merged[rankZ] = __float_as_uint(Z); }
			const uint32_t crossing = merged[lane];
			const bool alongX = (int)crossing < 0;
			const float C = __uint_as_float(crossing & 0x7FFFFFFFu);
			const lanemask_t xBits = __ballot(alongX);
			const int nxBefore = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(xBits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)xBits, 0u)); // x steps among the crossings before this lane's
			wPos = pos + nxBefore * posStepX + (lane - nxBefore) * posStepZ;
			const int posAfter = wPos + (alongX ? posStepX : posStepZ);
			wDistNext = C;
			wDistLast = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ray.distLast), __float_as_int(C), 0x138, 0xF, 0xF, false)); // wave_shr:1, lane 0 keeps the entry distance
			// the window ends with the first column whose exit crossing reaches the far clip / this level's LOD distance, or leads out of the world
			const float stopDistance = m_min(farClip, lodMax);
			const lanemask_t stops = __ballot(C >= stopDistance || (posAfter & outsideBits) != 0);
			count = stops != 0ull ? __ffsll((long long)stops) : CVX_WAVE;
			const int last = count - 1;
			const float exitDistance = rlf(C, last);
			const int exitPos = rli(posAfter, last);
			if (stops != 0ull) { endCode = (exitDistance >= farClip || (exitPos & outsideBits) != 0) ? 1 : 2; }
			// the DDA after `count` steps
			const int stepsX = __popcll(count < CVX_WAVE ? (xBits & ~lanes_from(count)) : xBits), stepsZ = count - stepsX;
			const float lastX = rlf(X, min(stepsX, CVX_WAVE - 1)), lastZ = rlf(Z, min(stepsZ, CVX_WAVE - 1));
			ray.tMaxX = stepsX < CVX_WAVE ? lastX : lastX + ray.tDeltaX;
			ray.tMaxZ = stepsZ < CVX_WAVE ? lastZ : lastZ + ray.tDeltaZ;
			ray.distLast = exitDistance;
			ray.distNext = ray.tMaxX < ray.tMaxZ ? ray.tMaxX : ray.tMaxZ; // cmin(tMax) (:147): finite sums of positive terms
			pos = exitPos;
			CVX_LMARK("dda_end");
		}
		guardSteps -= count;
		if (guardSteps <= 0) { endCode = 1; }
		CVX_LSTAT(0);
		CVX_LSTAT_ADD(1, count);

		CVX_LSEC(2);
		CVX_LMARK("winsetup_begin");
		// ---- the window's records: one load per lane, all in flight together
		const bool valid = lane < count;
		uint4 rec = uint4{ 0u, 0u, 0u, 0u }; // (a lane without a column: the empty column's record)
		if (valid) {
			const int px = wPos >> 16, pz = wPos & 0xFFFF;
			rec = ld4(arena, L.recordsOff + record_offset(px >> L.shift, pz >> L.shift, L.rowShift));
		}
#ifdef CVX_LONE_TIMES /* how long the wave waits for the window's records (diagnostic build only) */
		CVX_LSEC(12);
		__builtin_amdgcn_s_waitcnt(0x0F70);
		CVX_LSEC(2);
#endif
		// :289-293 (per lane)
		const f3 camSpaceMinLast = f3_madd(planeStartBottom, planeDir, wDistLast);
		const f3 camSpaceMinNext = f3_madd(planeStartBottom, planeDir, wDistNext);
		const f3 camSpaceMaxLast = f3_madd(planeStartTop, planeDir, wDistLast);
		const f3 camSpaceMaxNext = f3_madd(planeStartTop, planeDir, wDistNext);

		// ---- the runs of the records (cvx_device.h) and their projections
		const bool nonEmpty = rec.x != 0u;
		const bool listed = ((int)nonEmpty & (int)(rec.x < 0x40000000u)) != 0;
		const int code = (int)(rec.x >> 30); // solid runs in the record (0 for empty and listed columns)
		const lanemask_t nonEmptyMask = __ballot(nonEmpty);
		const uint32_t colorsOff = L.elementsOff + (rec.x & 0x3FFFFFFFu) * 4u; // ColorPointer, World.cs:185
		// run r = [runB[r], runT[r]] in LOD-0 voxels, RLEElement.Length and ColorsIndex (the lengths of the runs above)
		int runB[3], runT[3], runLen[3], runCidx[3];
		runB[0] = (int)(rec.w & 0xFFFFu);
		runT[0] = (int)(rec.y >> 16);
		runB[1] = (int)(rec.z & 0xFFFFu);
		runT[1] = (int)(rec.w >> 16) + 1;
		runB[2] = (int)(rec.y & 0xFFFFu);
		runT[2] = (int)(rec.z >> 16) + 1;
#pragma unroll
		for (int r = 0; r < 3; r++) { runLen[r] = (runT[r] - runB[r]) >> lod; }
		runCidx[0] = 0;
		runCidx[1] = runLen[0];
		runCidx[2] = runLen[0] + runLen[1];
		RunProj P[3];
		uint32_t faceColor[3];
		const lanemask_t listedMask = __ballot(listed);              // columns whose runs live in the run list: looked at when their turn comes (processColumn)
		const uint32_t listedBit = listed ? 0x80000000u : 0u; // ... as bit 31 of the lane's `todo` word
		lanemask_t topMask[3];                          // per run index: the columns whose run shows its top face (the others its bottom face, if any)
		int runsInWindow = 1; // how many run indices the window's columns use (wave-uniform)
#pragma unroll
		for (int r = 0; r < 3; r++) {
			const bool exists = r < code;
			faceColor[r] = 0u;
			P[r] = RunProj{};
			topMask[r] = 0ull;
			const bool anyRun = r == 0 || __ballot(exists) != 0ull;
			if (anyRun) { CVX_LSTAT(2);
				runsInWindow = r + 1; // (wave-uniform: a window without a second / third run anywhere skips their projections)
				P[r] = project_run(camSpaceMinLast, camSpaceMaxLast, camSpaceMinNext, camSpaceMaxNext, (float)runB[r], (float)runT[r], runLen[r], cameraPosYNormalized, invWorldMaxY);
				if (exists) {
					// :553,560: the run's first colour for a top face, its last for a bottom face (read for every run: the address is a colour of this run either way)
					faceColor[r] = ld_color(arena, colorsOff + ((uint32_t)(P[r].faceTop ? runCidx[r] : runCidx[r] + runLen[r] - 1) << L.colorShift));
				}
				topMask[r] = __ballot(P[r].faceTop);
			}
			// What can never draw gets an EMPTY pixel range, so the overlap tests of the pass (:505 / :581) need no flags: a run the record does not hold, a
			// side the near plane removed (:484), a face it removed (:566) or that the camera cannot see (neither top nor bottom, :549-565)
			const bool sideCan = exists && P[r].sideVisible;
			const bool faceCan = exists && P[r].faceNear && (P[r].faceTop || P[r].faceBottom);
			P[r].rbMinS = sideCan ? P[r].rbMinS : 0x7FFFFFFF;
			P[r].rbMaxS = sideCan ? P[r].rbMaxS : (int)0x80000000;
			P[r].rbMinF = faceCan ? P[r].rbMinF : 0x7FFFFFFF;
			P[r].rbMaxF = faceCan ? P[r].rbMaxF : (int)0x80000000;
		}

		CVX_LMARK("winsetup_end");
		lanemask_t todoMask = 0ull; // the columns with any such bit, and the columns of the run list
		uint32_t todo = 0u; // (runTests, below) which runs of the lane's column can touch the ray's state: bit 2r = the side of run r, bit 2r + 1 = its face; bit 31: a column of the run list
		float wbMin = 0.0f, wbMax = worldMaxY; // worldBoundsMin / Max of the lane's column (:283-284, narrowed by the cull :277-280 or set by the clip :392-393)

		// ---- pixel loops (lane = pixel) ------------------------------------------------------------------------------------------------
		// A pixel range inside ONE mask word (the usual case: a few pixels): the word comes out of its lane once (v_readlane), gives the unseen pixels of the
		// range, and goes back with the range marked (v_writelane) -- no pass over the 64 words.  Returns false when the range spans words (the general loops).
		auto oneWordRange = [&](int rbMin, int rbMax, lanemask_t &unseen) -> bool {
			const int i = (rbMin >> 5) - seen.wordBase;
			if (CVX_RARE(i != (rbMax >> 5) - seen.wordBase)) { return false; }
			const int first = rbMin & 31;
			const uint32_t range = ((2u << (rbMax - rbMin)) - 1u) << first;
			if (!HI || i < CVX_WAVE) {
				const uint32_t word = rlu(seen.w0, i);
				unseen = (lanemask_t)((range & ~word) >> first);
				if (unseen != 0ull) { seen.w0 = write_lane(seen.w0, word | range, i); }
			} else {
				const uint32_t word = rlu(seen.w1, i - CVX_WAVE);
				unseen = (lanemask_t)((range & ~word) >> first);
				if (unseen != 0ull) { seen.w1 = write_lane(seen.w1, word | range, i - CVX_WAVE); }
			}
			return true;
		};
		// side of a run, :519-533: the unseen pixels of [rbMin, rbMax] get the run's perspective-correct colour
		// (`operands` delivers the side's wave-uniform values -- v_readlanes of the column's lane -- and is only asked when a pixel is there to be written: more
		// than a third of the sides that reach this point hold no unseen pixel)
		struct SideOperands {
			float boundsX, boundsY, uvAx, uvBx, uvAy, uvBy;
			int elementLength, elementColorsIndex;
			uint32_t columnColorsOff;
		};
		auto sidePixels = [&](int rbMin, int rbMax, auto operands) {
			auto trip = [&](int yb, lanemask_t todo) {
				CVX_LSTAT(3);
				CVX_LSTAT_ADD(4, __popcll(todo));
				frustumDirMaxWorld = CVX_FLOAT_EPSILON; // :522
				const SideOperands o = operands();
				const TexRun texRun = tex_run(o.boundsX, o.boundsY, o.uvAx, o.uvBx, o.uvAy, o.uvBy);
				if (__builtin_amdgcn_inverse_ballot_w64(todo)) {
					const int y = yb + lane;
					bool certain;
					int row = tex_row_cheap(y, o.boundsX, o.uvAx, o.uvAy, texRun, certain);
					if (CVX_RARE(!certain)) { row = tex_row_exact(y, o.boundsX, o.boundsY, o.uvAx, o.uvBx, o.uvAy, o.uvBy); }
					const int colorIdx = m_clampi(row, 0, o.elementLength - 1) + o.elementColorsIndex;
					// the colour goes from memory STRAIGHT into the ray's row in LDS (global_load_lds_dword: lane p's dword lands at the LDS base + 4 p, and lane
					// p IS pixel yb + p): no register, so nothing waits for the load until the row is read out at the end of the ray
					__builtin_amdgcn_global_load_lds((const CVX_GLOBAL uint32_t *)(arena + (o.columnColorsOff + ((uint32_t)colorIdx << L.colorShift))),
					                                 (__attribute__((address_space(3))) uint32_t *)(merged + (CVX_WAVE + (yb - omin))), 4, 0, 0);
				}
			};
			lanemask_t unseen;
			if (CVX_USUAL(oneWordRange(rbMin, rbMax, unseen))) {
				if (unseen != 0ull) { trip(rbMin, unseen); }
				return;
			}
			for (int yb = rbMin; yb <= rbMax; yb += CVX_WAVE) {
				const int n = min(CVX_WAVE, rbMax - yb + 1);
				const lanemask_t todo = lone_unseen<HI>(seen, yb, n);
				if (CVX_RARE(todo == 0ull)) { continue; }
				trip(yb, todo);
			}
			lone_mark<HI>(seen, rbMin, rbMax);
		};
		// top / bottom of a run, :595-603
		auto facePixels = [&](int rbMin, int rbMax, uint32_t color) {
			auto trip = [&](int yb, lanemask_t todo) {
				CVX_LSTAT(5);
				CVX_LSTAT_ADD(6, __popcll(todo));
				frustumDirMaxWorld = CVX_FLOAT_EPSILON; // :598
				if (__builtin_amdgcn_inverse_ballot_w64(todo)) { merged[CVX_WAVE + (yb - omin) + lane] = color; }
			};
			lanemask_t unseen;
			if (CVX_USUAL(oneWordRange(rbMin, rbMax, unseen))) {
				if (unseen != 0ull) { trip(rbMin, unseen); }
				return;
			}
			for (int yb = rbMin; yb <= rbMax; yb += CVX_WAVE) {
				const int n = min(CVX_WAVE, rbMax - yb + 1);
				const lanemask_t todo = lone_unseen<HI>(seen, yb, n);
				if (CVX_RARE(todo == 0ull)) { continue; }
				trip(yb, todo);
			}
			lone_mark<HI>(seen, rbMin, rbMax);
		};

		// Can a pixel range change the ray's state?  It has to overlap the window [nextFreePixelMin, Max] (:505 / :581) -- and to hold an UNSEEN pixel:
		// ReducePixelHorizon (:660-697) moves a bound only when the range covers it (and the bounds are unseen pixels), the pixel loops only write unseen
		// pixels.  More than two thirds of the overlapping runs of the benchmark world hold none (the far side of a floating slab whose near side is drawn).
		// Per lane (a column's run in runTests, a run of the run list in processColumn): the mask word of the first pixel of the clamped range [lo, hi] comes
		// from the lane that holds it (ds_bpermute); a range that goes on into a second word counts as writable (the run's turn makes the exact tests).
		auto windowIsClean = [&]() -> bool { // no seen pixel inside the window: every overlapping range is writable (one ballot instead of a gather per range)
			bool clean = __ballot((seen.w0 & range_mask_any(seen.wordBase + lane, nextFreePixelMin, nextFreePixelMax)) != 0u) == 0ull;
			if (HI) { clean = clean && __ballot((seen.w1 & range_mask_any(seen.wordBase + 64 + lane, nextFreePixelMin, nextFreePixelMax)) != 0u) == 0ull; }
			return clean;
		};
		auto gatherWord = [&](int lo) -> uint32_t { // the mask word that holds pixel lo, from the lane that has it (ds_bpermute reads the lane from address bits [7:2])
			const int i = (lo >> 5) - seen.wordBase;
			uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute(i << 2, (int)seen.w0);
			if (HI) {
				const uint32_t word1 = (uint32_t)__builtin_amdgcn_ds_bpermute(i << 2, (int)seen.w1);
				word = i < CVX_WAVE ? word : word1;
			}
			return word;
		};
		auto holdsUnseen = [&](uint32_t word, int lo, int hi) -> lanemask_t { // (as a ballot)
			const int first = lo & 31;
			const int more = min(hi - lo, 31 - first); // pixels of the range in this word, less one
			return __ballot((hi - lo) > more) | __ballot(((~word >> first) << (31 - more)) != 0u);
		};

		// ---- element loop (:424-611) of column j, run by run in the reference's walk order, with the ray's current state
		auto processColumn = [&](int j) {
			CVX_LSTAT(9);
			CVX_LMARK("process_begin");
			CVX_LSECE(6);
#ifdef CVX_LONE_STATS
			{ // how often the window [nextFreePixelMin, Max] holds no seen pixel when a column is processed ("clean": every scan / unseen test is then trivial)
				int n_ = __popc(seen.w0 & range_mask_any(seen.wordBase + lane, nextFreePixelMin, nextFreePixelMax)) + (HI ? __popc(seen.w1 & range_mask_any(seen.wordBase + 64 + lane, nextFreePixelMin, nextFreePixelMax)) : 0);
				if (__ballot(n_ != 0) == 0ull) { CVX_LSTAT(17); }
			}
#endif
			const uint32_t bits = rlu(todo, j);
			if (CVX_RARE((int)bits < 0)) {
				CVX_LSTAT(10);
				// A column of the run list (cvx_device.h: a few per thousand of a built terrain, most columns of a model world such as mill.obj -- ten thin runs
				// and more per column --, every column of a foreign blob).  Here the lanes are the column's RUNS, 64 at a time in the reference's walk order
				// (:428-437): one load brings them all, one pass projects them all (project_run with the column's corners, wave-uniform, and a run per lane),
				// one more fetches their face colours; then the runs that can touch the state take their turn one after the other, their operands read from their lanes.
				const float worldBoundsMin = rlf(wbMin, j), worldBoundsMax = rlf(wbMax, j);
				const uint32_t columnColorsOff = L.elementsOff + (rlu(rec.x, j) & 0x3FFFFFFFu) * 4u;
				const uint32_t columnRunsOff = L.runsOff + rlu(rec.z, j) * 8u;
				const int solidCount = (int)rlu(rec.w, j);
				const float dL = rlf(wDistLast, j), dN = rlf(wDistNext, j); // :289-293 again, from the column's two distances
				const f3 qMinLast = f3_madd(planeStartBottom, planeDir, dL), qMinNext = f3_madd(planeStartBottom, planeDir, dN);
				const f3 qMaxLast = f3_madd(planeStartTop, planeDir, dL), qMaxNext = f3_madd(planeStartTop, planeDir, dN);
				for (int first = 0; first < solidCount && alive; first += CVX_WAVE) {
					const int position = first + lane; // in walk order
					const bool have = position < solidCount;
					uint2 run = uint2{ 0u, 0u };
					if (have) { run = ld2(arena, columnRunsOff + (uint32_t)(DIR > 0 ? position : solidCount - 1 - position) * 8u); }
					const float elementBoundsMin = (float)(run.x & 0xFFFFu), elementBoundsMax = (float)(run.x >> 16) + 1.0f;
					const int elementLength = (int)(((run.x >> 16) + 1u - (run.x & 0xFFFFu)) >> lod);
					const int elementColorsIndex = (int)(run.y & 0xFFFFu);
					const bool in = ((int)have & (int)!(elementBoundsMin > worldBoundsMax) & (int)!(elementBoundsMax < worldBoundsMin)) != 0; // :461-475
					const RunProj R = project_run(qMinLast, qMaxLast, qMinNext, qMaxNext, elementBoundsMin, elementBoundsMax, elementLength, cameraPosYNormalized, invWorldMaxY);
					uint32_t secondaryColor = 0u;
					if (in) { secondaryColor = ld_color(arena, columnColorsOff + ((uint32_t)(R.faceTop ? elementColorsIndex : elementColorsIndex + elementLength - 1) << L.colorShift)); }
					const int loS = max(R.rbMinS, nextFreePixelMin), hiS = min(R.rbMaxS, nextFreePixelMax), loF = max(R.rbMinF, nextFreePixelMin), hiF = min(R.rbMaxF, nextFreePixelMax);
					// (on ballots, as in the pass) the run's side / face can touch the state: inside the world bounds (:461-475), visible (:484 / :566), wanted (:549-565),
					// overlapping the window (:505 / :581) -- and holding an unseen pixel, side and face each by its own test: three quarters of a model world's
					// candidates hold one in one of the two only
					const lanemask_t inBits = __ballot(in);
					lanemask_t sideBits = inBits & __ballot(R.sideVisible) & __ballot(loS <= hiS);
					const lanemask_t wantedBits = (__ballot(R.faceTop) & __ballot(!(elementBoundsMax > worldBoundsMax))) | (__ballot(R.faceBottom) & __ballot(!(elementBoundsMin < worldBoundsMin)));
					lanemask_t faceBits = inBits & wantedBits & __ballot(R.faceNear) & __ballot(loF <= hiF);
					if (!windowIsClean()) {
						const uint32_t wordS = gatherWord(loS), wordF = gatherWord(loF);
						sideBits &= holdsUnseen(wordS, loS, hiS);
						faceBits &= holdsUnseen(wordF, loF, hiF);
					}
					lanemask_t candidates = sideBits | faceBits;
					// the candidates take their turn in walk order, each with its operands read from its lane when (and if) they are needed
					while (candidates != 0ull && alive) {
						const int l = __ffsll((long long)candidates) - 1;
						candidates &= candidates - 1ull;
						if (((sideBits >> l) & 1ull) != 0ull) { // side :484-542
							int rbMin = rli(R.rbMinS, l), rbMax = rli(R.rbMaxS, l);
							if (rbMax >= nextFreePixelMin && rbMin <= nextFreePixelMax) { // :505
								CVX_LSTAT(7);
								lone_reduce_pixel_horizon<HI>(seen, omin, omax, rbMin, rbMax, nextFreePixelMin, nextFreePixelMax, frustumBoundsMin, frustumBoundsMax);
								if (rbMin <= rbMax) {
									sidePixels(rbMin, rbMax, [&]() {
										return SideOperands{ rlf(R.boundsX, l), rlf(R.boundsY, l), rlf(R.uvAx, l), rlf(R.uvBx, l), rlf(R.uvAy, l), rlf(R.uvBy, l), rli(elementLength, l), rli(elementColorsIndex, l), columnColorsOff };
									});
								}
								if (nextFreePixelMin > nextFreePixelMax) { alive = false; return; } // :535-539
							}
						}
						if (((faceBits >> l) & 1ull) != 0ull) { // face :544-610
							int rbMin = rli(R.rbMinF, l), rbMax = rli(R.rbMaxF, l);
							if (rbMax >= nextFreePixelMin && rbMin <= nextFreePixelMax) { // :581
								CVX_LSTAT(8);
								lone_reduce_pixel_horizon<HI>(seen, omin, omax, rbMin, rbMax, nextFreePixelMin, nextFreePixelMax, frustumBoundsMin, frustumBoundsMax);
								if (rbMin <= rbMax) { facePixels(rbMin, rbMax, rlu(secondaryColor, l)); }
								if (nextFreePixelMin > nextFreePixelMax) { alive = false; return; } // :604-608
							}
						}
					}
				}
				return;
			}
			// the runs the last pass over the window found able to touch the state (runTests: inside the column's world bounds :461-475, a visible side whose
			// pixels overlapped [nextFreePixelMin, Max] :505, a wanted visible face that did :549-565,581), in the reference's walk order; the window can only
			// have shrunk since, so each is tested again against the current one
			auto oneRun = [&](auto runConstant) -> bool { // side and face of run r of the column; false: the ray has ended
				constexpr int r = decltype(runConstant)::value;
				if ((bits & (1u << (2 * r))) != 0u) {
					int rbMin = rli(P[r].rbMinS, j), rbMax = rli(P[r].rbMaxS, j);
					if (CVX_USUAL(rbMax >= nextFreePixelMin && rbMin <= nextFreePixelMax)) { // :505
						CVX_LSTAT(7);
#ifdef CVX_LONE_STATS
						if (rbMin <= nextFreePixelMin) { CVX_LSTAT(29); }
						if (rbMax >= nextFreePixelMax) { CVX_LSTAT(46); }
#endif
						CVX_LMARK("sidereduce_begin");
						CVX_LSECE(7);
						lone_reduce_pixel_horizon<HI>(seen, omin, omax, rbMin, rbMax, nextFreePixelMin, nextFreePixelMax, frustumBoundsMin, frustumBoundsMax);
						CVX_LMARK("sidereduce_end");
						CVX_LSECE(8);
						if (CVX_USUAL(rbMin <= rbMax)) {
							sidePixels(rbMin, rbMax, [&]() {
								return SideOperands{ rlf(P[r].boundsX, j), rlf(P[r].boundsY, j), rlf(P[r].uvAx, j), rlf(P[r].uvBx, j), rlf(P[r].uvAy, j), rlf(P[r].uvBy, j), rli(runLen[r], j), rli(runCidx[r], j),
								                     rlu(colorsOff, j) };
							});
						}
						if (CVX_RARE(nextFreePixelMin > nextFreePixelMax)) { alive = false; return false; } // :535-539
						CVX_LMARK("sidepixels_end");
						CVX_LSECE(6);
					}
				}
				if ((bits & (2u << (2 * r))) != 0u) {
					int rbMin = rli(P[r].rbMinF, j), rbMax = rli(P[r].rbMaxF, j);
					if (CVX_USUAL(rbMax >= nextFreePixelMin && rbMin <= nextFreePixelMax)) { // :581
						CVX_LSTAT(8);
#ifdef CVX_LONE_STATS
						if (rbMin <= nextFreePixelMin) { CVX_LSTAT(29); }
						if (rbMax >= nextFreePixelMax) { CVX_LSTAT(46); }
#endif
						CVX_LMARK("facereduce_begin");
						CVX_LSECE(9);
						lone_reduce_pixel_horizon<HI>(seen, omin, omax, rbMin, rbMax, nextFreePixelMin, nextFreePixelMax, frustumBoundsMin, frustumBoundsMax);
						CVX_LMARK("facereduce_end");
						CVX_LSECE(10);
						if (CVX_USUAL(rbMin <= rbMax)) { facePixels(rbMin, rbMax, rlu(faceColor[r], j)); }
						if (CVX_RARE(nextFreePixelMin > nextFreePixelMax)) { alive = false; return false; } // :604-608
						CVX_LMARK("facepixels_end");
						CVX_LSECE(6);
					}
				}
				return true;
			};
			if (CVX_USUAL((bits & ~3u) == 0u)) { // the usual column: its only work is in its first run -- one test (the inverse walk, which starts at the last run, paid two to get there)
				oneRun(std::integral_constant<int, 0>{});
				return;
			}
			// (one test for a run with nothing to do, and one to leave when the runs still to come have nothing: a column with one run pays two tests, not six)
			if (DIR > 0) { // the walk starts at the top (ITERATION_DIRECTION +1) or at the bottom (-1), :428-437
				if ((bits & 3u) != 0u) { if (!oneRun(std::integral_constant<int, 0>{})) { return; } }
				if ((bits >> 2) == 0u) { return; }
				if ((bits & 12u) != 0u) { if (!oneRun(std::integral_constant<int, 1>{})) { return; } }
				if ((bits >> 4) == 0u) { return; }
				oneRun(std::integral_constant<int, 2>{});
			} else {
				if ((bits & 48u) != 0u) { if (!oneRun(std::integral_constant<int, 2>{})) { return; } }
				if ((bits & 15u) == 0u) { return; }
				if ((bits & 12u) != 0u) { if (!oneRun(std::integral_constant<int, 1>{})) { return; } }
				if ((bits & 3u) == 0u) { return; }
				oneRun(std::integral_constant<int, 0>{});
			}
		};

		// ---- frustum clip of column j (:295-422), computed in the column's own lane (the other lanes compute along: their results are not looked at)
		// The clip's eight division chains are the same four computations -- (Last | Next intersection) x (lower | upper window bound) -- on different operands:
		// they run in FOUR LANES at once (lanes = operations; every quad of the wave computes the same four, so no lane has to be switched off).  Lane role
		// c = lane & 3: bit 0 = the upper bound ("max": clip_max against frustumBoundsMax in the ordinary case), bit 1 = the Next intersection.  Partners are
		// exchanged inside the quad (DPP quad_perm).  Every value is computed by the reference's operations on the reference's operands, as clip_min / clip_max /
		// clip_world_bounds (cvx_kernels.h) do -- one lane each instead of one after the other.
		const bool roleMax = (lane & 1) != 0, roleNext = (lane & 2) != 0;
		auto quadSwapNext = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }; // quad_perm:[2,3,0,1]: Last <-> Next
		auto quadSwapMax = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); };  // quad_perm:[1,0,3,2]: min <-> max
		auto clipColumn = [&](int j) {
			CVX_LSTAT(11);
#ifdef CVX_LONE_STATS
			if (frustumBoundsMin == lastClipBoundsMin_ && frustumBoundsMax == lastClipBoundsMax_) { CVX_LSTAT(15); } // (how often a clip sees the window bounds of the clip before it)
			lastClipBoundsMin_ = frustumBoundsMin;
			lastClipBoundsMax_ = frustumBoundsMax;
#endif
			CVX_LMARK("clip_begin");
			CVX_LSECE(3);
			const float dL = rlf(wDistLast, j), dN = rlf(wDistNext, j);
			const float dist = roleNext ? dN : dL;
			// :289-293 for this lane's intersection (x and z: the clip never looks at y)
			const float px_ = planeDir.x * dist, pz_ = planeDir.z * dist;
			const float pMinX = planeStartBottom.x + px_, pMinZ = planeStartBottom.z + pz_, pMaxX = planeStartTop.x + px_, pMaxZ = planeStartTop.z + pz_;
			// GetWorldBoundsClippingCamSpace, CameraData.cs:51-99 (the flattened form of clip_world_bounds)
			// (the boolean algebra on the ballots, 64-bit scalars: a select between per-lane booleans would be made in vector registers)
			const lanemask_t roleMaxBits = 0xAAAAAAAAAAAAAAAAull;
			const lanemask_t a1 = __ballot(pMinX > pMinZ * frustumBoundsMax);
			const lanemask_t a2 = __ballot(pMaxX > pMaxZ * frustumBoundsMax);
			const lanemask_t b1 = __ballot(pMinX < pMinZ * frustumBoundsMin);
			const lanemask_t b2 = __ballot(pMaxX < pMaxZ * frustumBoundsMin);
			const lanemask_t straddleBits = ~a1 & b1 & a2;
			const lanemask_t needBits = (roleMaxBits & ((a1 & b2) | (~a1 & (a2 | b2)))) | (~roleMaxBits & (a1 | b1));
			const lanemask_t againstMaxBits = (roleMaxBits & ~a1 & a2) | (~roleMaxBits & a1); // which window bound this lane's clip is made against
			const lanemask_t clippedBits = (a1 & a2) | (~a1 & ~a2 & b1 & b2);
			const bool need = __builtin_amdgcn_inverse_ballot_w64(needBits);
			const bool againstMax = __builtin_amdgcn_inverse_ballot_w64(againstMaxBits);
			const float finv = quot_safe(1.0f, recip_safe(againstMax ? frustumBoundsMax : frustumBoundsMin)); // CameraData.cs:103,111
			const float c0 = 1.0f * pMaxZ - finv * pMaxX;
			const float c1 = 1.0f * pMinZ - finv * pMinX;
			const float num = roleMax ? c1 : c0, oth = roleMax ? c0 : c1;
			const float q = num / (num - oth);
			const float clipT = roleMax ? q : 1.0f - q; // clip_max: c1 / (c1 - c0); clip_min: 1 - c0 / (c0 - c1)
			const float lerpT = need ? clipT : (roleMax ? 1.0f : 0.0f);
			const bool clippedLast = (clippedBits & 1ull) != 0ull, clippedNext = (clippedBits & 4ull) != 0ull;
			if (!(straddleBits & 1ull) || !(straddleBits & 4ull)) { CVX_LSTAT(12); }
			// :300-390: each bound from the Last or the Next intersection
			const float otherT = quadSwapNext(lerpT);
			const float lastT = roleNext ? otherT : lerpT, nextT = roleNext ? lerpT : otherT;
			const lanemask_t closerBits = (roleMaxBits & __ballot(lastT > nextT)) | (~roleMaxBits & __ballot(lastT < nextT));
			const bool fromLast = __builtin_amdgcn_inverse_ballot_w64(clippedLast ? 0ull : (clippedNext ? ~0ull : closerBits));
			float worldBounds = m_lerp(0.0f, worldMaxY, fromLast ? lastT : nextT);
			const float dir = (worldBounds - posY) / (fromLast ? dL : dN);
			worldBounds = roleMax ? ceilf(worldBounds) : floorf(worldBounds);
			frustumDirMinWorld = rlf(dir, 0);
			frustumDirMaxWorld = rlf(dir, 1);
			const float clipWbMin = rlf(worldBounds, 0), clipWbMax = rlf(worldBounds, 1);
			{ // the column's world bounds go into its lane's slot
				const bool mine = lane == j;
				wbMin = mine ? clipWbMin : wbMin;
				wbMax = mine ? clipWbMax : wbMax;
			}
			// the clipped point of this lane (x, z) and "window untouched" (cvx_kernels.h, drawColumn): both ends straddle the window and every clipped point lies
			// on its bound -- then :337-421 change nothing
			const float ptX = pMinX + (pMaxX - pMinX) * lerpT, ptZ = pMinZ + (pMaxZ - pMinZ) * lerpT;
			const bool onBound = fabsf(ptX - (roleMax ? frustumBoundsMax : frustumBoundsMin) * ptZ) < 0.4f * fabsf(ptZ);
			const bool windowUntouched = (straddleBits & __ballot(onBound) & 0xFull) == 0xFull;
			CVX_LMARK("clip_end");
			if (!CVX_USUAL(windowUntouched)) {
				CVX_LSTAT(13);
				CVX_LSECE(4);
				const float mine = ptX / ptZ; // lanes: minLast, maxLast, minNext, maxNext
				const float partner = quadSwapMax(mine);
				const bool sw = __builtin_amdgcn_inverse_ballot_w64((roleMaxBits & __ballot(mine < partner)) | (~roleMaxBits & __ballot(partner < mine))); // :339-346: max < min -> swap
				const float v = sw ? partner : mine;
				const float o = quadSwapNext(v);
				const float lastV = roleNext ? o : v, nextV = roleNext ? v : o;
				const float bothMin = hw_min(lastV, nextV), bothMax = hw_max(lastV, nextV);
				const float both = roleMax ? bothMax : bothMin;
				const float camSpaceClipped = clippedLast ? nextV : (clippedNext ? lastV : both);
				const int pixelFloor = f2i_floor(camSpaceClipped), pixelCeil = f2i(ceilf(camSpaceClipped));
				const int writableMinPixel = rli(pixelFloor, 0);
				const int writableMaxPixel = rli(pixelCeil, 1);
				if ((clippedLast && clippedNext) || writableMaxPixel < nextFreePixelMin || writableMinPixel > nextFreePixelMax) { // :297-299, :399-403
					alive = false;
					return;
				}
				if (writableMinPixel > nextFreePixelMin) { nextFreePixelMin = lone_scan_up<HI>(seen, writableMinPixel, omax); }   // :405-410
				if (writableMaxPixel < nextFreePixelMax) { nextFreePixelMax = lone_scan_down<HI>(seen, writableMaxPixel, omin); } // :411-416
				if (nextFreePixelMin > nextFreePixelMax) { alive = false; } // :417-421
				CVX_LMARK("cliptouched_end");
			}
		};

		// ---- one pass over the lanes >= from with the ray's current state: the cull of every column (:261-281) and whether any of its runs can touch
		// the state (:461-475, :505, :549-565, :581).  `clipped` >= 0: that lane's column was just clipped (it is drawn with the clip's world bounds).
		// runTests: which runs of the lane's column can touch the ray's state, given the column's world bounds (wbMin / wbMax) and the current window -- `todo`:
		// bit 2r = the side of run r, bit 2r + 1 = its face, bit 31 = a column of the run list (looked at when its turn comes).
		// A run's side / face has to overlap the window (:505 / :581) and to hold an unseen pixel (windowIsClean / holdsUnseen above: one test for the run's
		// side and face together; the gathers of all runs are issued before the first is looked at: one LDS round trip per pass, not one per run).
		// The boolean algebra is written on the ballots -- 64-bit scalars --: selects and merges of per-lane booleans would be made in vector registers.
		auto runTestsOf = [&](auto runsConstant) {
			constexpr int runs = decltype(runsConstant)::value;
			const bool windowClean = windowIsClean();
			int loS[runs], hiS[runs], loF[runs], hiF[runs];
			lanemask_t sideBits[runs], faceBits[runs];
#pragma unroll
			for (int r = 0; r < runs; r++) {
				const float b = (float)runB[r], t = (float)runT[r];
				const lanemask_t in = __ballot(!(b > wbMax)) & __ballot(!(t < wbMin)); // :461-475
				loS[r] = max(P[r].rbMinS, nextFreePixelMin);
				hiS[r] = min(P[r].rbMaxS, nextFreePixelMax);
				loF[r] = max(P[r].rbMinF, nextFreePixelMin);
				hiF[r] = min(P[r].rbMaxF, nextFreePixelMax);
				sideBits[r] = in & __ballot(loS[r] <= hiS[r]);                                                                  // :484, :505 (an invisible side has an empty range)
				const lanemask_t wanted = (topMask[r] & __ballot(!(t > wbMax))) | (~topMask[r] & __ballot(!(b < wbMin))); // :549-565 (a face that is neither has an empty range)
				faceBits[r] = in & wanted & __ballot(loF[r] <= hiF[r]);                                                         // :566, :581
			}
			if (!windowClean) {
				int lo[runs], hi[runs];
				uint32_t word[runs];
#pragma unroll
				for (int r = 0; r < runs; r++) {
					const bool side = __builtin_amdgcn_inverse_ballot_w64(sideBits[r]), face = __builtin_amdgcn_inverse_ballot_w64(faceBits[r]);
					lo[r] = side ? (face ? min(loS[r], loF[r]) : loS[r]) : loF[r];
					hi[r] = side ? (face ? max(hiS[r], hiF[r]) : hiS[r]) : hiF[r];
					word[r] = gatherWord(lo[r]);
				}
#pragma unroll
				for (int r = 0; r < runs; r++) {
					const lanemask_t writable = holdsUnseen(word[r], lo[r], hi[r]);
					sideBits[r] &= writable;
					faceBits[r] &= writable;
				}
			}
			todo = listedBit;
#pragma unroll
			for (int r = 0; r < runs; r++) {
				todo |= (__builtin_amdgcn_inverse_ballot_w64(sideBits[r]) ? 1u << (2 * r) : 0u) | (__builtin_amdgcn_inverse_ballot_w64(faceBits[r]) ? 2u << (2 * r) : 0u);
			}
			todoMask = listedMask;
#pragma unroll
			for (int r = 0; r < runs; r++) { todoMask |= sideBits[r] | faceBits[r]; }
		};
		auto runTests = [&]() { // (wave-uniform: how many run indices the window's columns use)
			if (runsInWindow == 3) {
				runTestsOf(std::integral_constant<int, 3>{});
			} else if (runsInWindow == 2) {
				runTestsOf(std::integral_constant<int, 2>{});
			} else {
				runTestsOf(std::integral_constant<int, 1>{});
			}
		};
		lanemask_t hits = 0ull, leftWorldMask = 0ull;
		auto cullAndFilter = [&](int from, int clipped) {
			CVX_LSTAT(14);
#ifdef CVX_LONE_STATS
			if (runsInWindow >= 2) { CVX_LSTAT(24); if ((__ballot(code >= 2) & lanes_from(from)) == 0ull) { CVX_LSTAT(25); } } // (passes over two-run windows / ... whose remaining columns have one run)
			if (runsInWindow >= 3) { CVX_LSTAT(26); if ((__ballot(code >= 3) & lanes_from(from)) == 0ull) { CVX_LSTAT(27); } } // (... three-run windows / ... whose remaining columns have at most two)
#endif
			CVX_LMARK("filter_begin");
			CVX_LSECE(5);
			const float columnWorldMin = (float)(rec.y & 0xFFFFu);
			const float columnWorldMax = (float)(rec.y >> 16);
			const float newMax = posY + hw_max(frustumDirMaxWorld * wDistNext, frustumDirMaxWorld * wDistLast);
			const float newMin = posY + hw_min(frustumDirMinWorld * wDistNext, frustumDirMinWorld * wDistLast);
			const lanemask_t ownBits = clipped >= 0 ? 1ull << clipped : 0ull;
			const bool own = __builtin_amdgcn_inverse_ballot_w64(ownBits);
			const lanemask_t leftWorldBits = nonEmptyMask & ~ownBits & (__ballot(newMin > worldMaxY) | __ballot(newMax < 0.0f));
			const lanemask_t noOverlapBits = ~ownBits & (__ballot(columnWorldMin > newMax) | __ballot(columnWorldMax < newMin));
			wbMin = own ? wbMin : newMin;
			wbMax = own ? wbMax : newMax;
			runTests();
			const lanemask_t range = lanes_from(from);
			hits = nonEmptyMask & ~leftWorldBits & ~noOverlapBits & todoMask & range;
			leftWorldMask = leftWorldBits & range;
			CVX_LMARK("filter_end");
			CVX_LSECE(0);
		};

		// The window's face colours have to be there before the events start: left to the compiler, the wait (vmcnt(0): it cannot count what is in flight
		// across the loop) sits in front of every face's colour read -- and there it also drains the side colours the column's side trip has just sent on
		// their way into the LDS row, a full memory round trip per face.  Here it costs the tail of ONE round trip per window.
		CVX_LSEC(13);
		__builtin_amdgcn_s_waitcnt(0x0F70);
		CVX_LSEC(0);
		// `frustumDirMaxWorld == float.Epsilon` (:261): the sentinel is the one positive float with the bit pattern 1 (denormals are not flushed), so the
		// wave-uniform comparison can be a scalar integer one (gfx9 has no scalar float compare)
		auto directionsGone = [&]() -> bool { return __float_as_int(frustumDirMaxWorld) == 1; };
		// ---- the events of the window, in column order
		int next = 0;         // first lane not yet looked at
		bool hitsValid = false;
		while (alive) {
			if (directionsGone()) {
				// no valid frustum directions (:261 false): no cull -- the next non-empty column is drawn (:251-256), clipped first when it lies beyond 2 units (:295)
				const lanemask_t rem = next < CVX_WAVE ? (nonEmptyMask & lanes_from(next)) : 0ull;
				if (rem == 0ull) { break; }
				const int j = __ffsll((long long)rem) - 1;
				if (rlf(wDistLast, j) > 2.0f) {
					clipColumn(j);
					CVX_LSECE(0);
					if (!alive) { break; }
					cullAndFilter(j, j);
					hitsValid = true;
					// the clipped column itself goes on to its element loop whatever the clip computed (also a direction that happens to BE the sentinel)
					if ((hits >> j) & 1ull) {
						CVX_LSTAT(21);
						processColumn(j);
#ifdef CVX_LONE_STATS
						if (!directionsGone()) { CVX_LSTAT(28); }
#endif
						CVX_LSECE(0);
						if (directionsGone()) { hitsValid = false; }
					}
					next = j + 1;
				} else {
					{ const bool mine = lane == j; wbMin = mine ? 0.0f : wbMin; wbMax = mine ? worldMaxY : wbMax; }
					runTests();
					processColumn(j);
					CVX_LSECE(0);
					next = j + 1;
					hitsValid = false;
				}
			} else {
				if (!hitsValid) {
					if (next >= CVX_WAVE) { break; }
					cullAndFilter(next, -1);
					hitsValid = true;
				}
				const lanemask_t h = next < CVX_WAVE ? (hits & lanes_from(next)) : 0ull;
				const lanemask_t l = next < CVX_WAVE ? (leftWorldMask & lanes_from(next)) : 0ull;
				if ((h | l) == 0ull) { break; }
				const int fh = h != 0ull ? __ffsll((long long)h) - 1 : CVX_WAVE, fl = l != 0ull ? __ffsll((long long)l) - 1 : CVX_WAVE;
				if (fl < fh) { alive = false; break; } // :265-269: the frustum left the world
				processColumn(fh);
#ifdef CVX_LONE_STATS
				if (!directionsGone()) { CVX_LSTAT(28); } // (a hit of the pass that wrote no pixel)
#endif
				CVX_LSECE(0);
				next = fh + 1;
				if (directionsGone()) { hitsValid = false; } // a pixel was written: the directions are gone (:522,598)
			}
		}
		if (!alive || endCode == 1) { break; }
		if (endCode == 2) { // NextLOD (:237-243) for the column the ray stands on
			ray.px = pos >> 16;
			ray.pz = pos & 0xFFFF;
			ray.sx = posStepX >> 16;
			ray.sz = posStepZ;
			dda_next_lod(ray, voxelScale, (dirFlags & 1) != 0, (dirFlags & 2) != 0);
			lod++;
			voxelScale *= 2;
			L = world->level[lod];
			{ const float next_ = F.lod[min(lod, 5)]; lodMax = lod < 5 ? next_ : __builtin_inff(); }
			pos = ray.px * 65536 + ray.pz;
			posStepX = ray.sx * 65536;
			posStepZ = ray.sz;
		}
	}
}

// ---------------------------------------------------------------------------
// lone kernel: grid = 64 x tiles (workgroup b renders ray b % 64 of tile b / 64; cvx_gpu.hip DrawBatch), block = 64 (one wave).
// HI: windows of more than 2048 pixels ([origMin, origMax] spans more than 64 mask words: 4K) carry a second mask register.
// ---------------------------------------------------------------------------
template <bool HI>
#ifndef CVX_LONE_WAVES_PER_SIMD
#define CVX_LONE_WAVES_PER_SIMD 3
#endif
__global__ __launch_bounds__(CVX_WAVE, CVX_LONE_WAVES_PER_SIMD) void lone_kernel(const DevFrame *__restrict__ frames, const DevTile *__restrict__ tiles, const DevWorld *__restrict__ world)
{
	extern __shared__ uint32_t lds[]; // [0, 64): the DDA's crossings in merged order (lone_trace_ray); [64, 64 + omax - omin]: the ray's pixel row
	uint32_t *merged = lds;
#ifdef CVX_LONE_STATS
	const unsigned long long waveStart_ = __builtin_amdgcn_s_memtime();
#endif
	const DevTile tile = tiles[blockIdx.x >> 6];
	const DevFrame &F = frames[tile.frame];
	const DevSegment &S = F.seg[tile.seg];
	const int firstLane = (int)(blockIdx.x & 63u);
	const int planeRayIndex = tile.tileInSeg * CVX_WAVE + firstLane; // RaySetupJob (:19-39)
	if (planeRayIndex >= S.rayCount) { return; }
#ifdef CVX_LONE_PRIO
	if ((int)blockIdx.x < CVX_LONE_PRIO) { __builtin_amdgcn_s_setprio(3); } // (experiment: the longest rays of the launch first in their SIMD's issue arbitration)
#endif
	const int omin = S.omin, omax = S.omax;
	LoneSeen seen;
	seen.w0 = seen.w1 = 0u; // stackalloc is zero-initialised, :208
	seen.wordBase = omin >> 5;
	seen.lane = (int)threadIdx.x;
	const gptr_tile tileOut = (gptr_tile)tile.out;
	const uint32_t laneByteOff = (uint32_t)firstLane * 4u;
	// The ray's pixel row [omin, omax] is staged in LDS and written out once, at the end: gfx9 counts loads and stores in ONE counter (vmcnt), so a
	// pixel store in the column loop would make every later wait for a colour load also wait for the store's acknowledgement from memory.  Staged, the
	// loop's only vector-memory operations are loads, and the row's stores are issued back to back with nothing waiting for them.  Every pixel starts
	// as the skybox colour (WriteSkybox / WriteSkyboxFull, :699-716: whatever is not written by a run).
	uint32_t *pix = lds + CVX_WAVE - omin;
	for (int y = omin + seen.lane; y <= omax; y += CVX_WAVE) { pix[y] = CVX_SKYBOX_ARGB; }
#ifdef CVX_LONE_STATS
	unsigned int stat_[48];
	for (int i = 0; i < 48; i++) { stat_[i] = 0u; }
	stat_[30] = (unsigned int)__builtin_amdgcn_s_memtime();
#else
	unsigned int *stat_ = nullptr;
#endif
	if (F.inverse) { // RenderJob.Execute :174-178
		lone_trace_ray<-1, HI>(F, S, world, planeRayIndex, seen, merged, stat_);
	} else {
		lone_trace_ray<1, HI>(F, S, world, planeRayIndex, seen, merged, stat_);
	}
#ifdef CVX_LONE_STATS
	stat_[16]++;
	CVX_LSEC(0);
	if (threadIdx.x == 0) {
		const unsigned long long life_ = __builtin_amdgcn_s_memtime() - waveStart_;
		for (int i = 0; i < 48; i++) { if (i != 18 && i != 19 && i != 20) { atomicAdd(&g_loneStats[i], (unsigned long long)stat_[i]); } }
		atomicAdd(&g_loneStats[18], life_);                 // sum of the waves' lives (clock ticks)
		atomicMax(&g_loneStats[19], life_);                 // the longest
		if (life_ == atomicMax(&g_loneStats[19], 0ull)) { // (the counters of the longest wave so far: racy, diagnostic only)
			g_loneStats[20] = stat_[1];
			for (int i = 0; i < 48; i++) { g_loneLongest[i] = stat_[i]; }
		}
	}
#endif
	// the row goes out: pixel y of this ray at tile row y (256 bytes per row, cvx_device.h); first every colour still on its way into the row has to be there
	__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) (gfx9 encoding, see cvx_kernels.h)
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	CVX_LSEC(11);
	for (int y = omin + seen.lane; y <= omax; y += CVX_WAVE) { st_pixel(tileOut, laneByteOff, y, pix[y]); }
}

} // namespace cvxk
