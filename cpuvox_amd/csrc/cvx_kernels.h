// cvx_kernels.h -- HIP kernels of libcpuvox_gpu (gfx950 / CDNA4, wave64).
//
// render kernel: one wavefront (64-thread workgroup) per tile of 64
// consecutive rays of one segment, one lane per raybuffer column.  It fuses the
// reference's four Burst jobs (Assets/Code/Rendering/DrawSegmentRayJob.cs):
//   RaySetupJob :19-39, DDASetupJob :58-76, TraceToFirstColumnJob :95-143,
//   RenderJob -> ExecuteRay :164-620
// The per-ray "seen pixel" byte cache (:208) is a per-lane bitmask in LDS,
// word-interleaved across lanes (word w of lane l at w*64+l: conflict-free);
// horizon scans (:407-414, :678-692) are ffs/clz over mask words; the skybox
// fill (:699-716) is deferred to one wave-uniform pass over the pixel rows at
// the end so its stores are 256-byte coalesced rows of the tile.
//
// Arithmetic contract (shared with the CPU oracle, checked by tests): IEEE
// binary32, no FMA contraction (-ffp-contract=off), correctly rounded / and
// sqrt, denormals preserved, (int)float with the x86 cvttss2si rule.
#pragma once

#include <hip/hip_runtime.h>

#include "cvx_device.h"

namespace cvxk {

// World tables, element pools and raybuffer tiles are reached through pointers that the kernel reads from memory
// (DevWorld / DevTile), so the compiler cannot tell that they are global and emits FLAT loads and stores -- and a flat
// access may come back out of order, so every wait on one is s_waitcnt vmcnt(0) lgkmcnt(0): the look-ahead fetch of the
// next column's record would be waited for by the first colour load or run-list load of the current column.  Saying
// "global" (address space 1) gives global_load / global_store, which return in order and can be waited for with counted
// vmcnt(N), leaving the younger look-ahead loads in flight.
#define CVX_GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); // (HIP's uint4 / uint2 classes cannot be copied out of a qualified address space)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const CVX_GLOBAL uint8_t *gptr_arena;
typedef CVX_GLOBAL uint32_t *gptr_out;
// World data: wave-uniform arena base (a scalar register pair) + 32-bit per-lane byte offset -- the "saddr" form of the
// global_load instructions; a per-lane 64-bit pointer would cost two vector registers for every table the column loop touches.
__device__ __forceinline__ uint4 ld4(gptr_arena arena, uint32_t byteOff) { const u32x4 v = *(const CVX_GLOBAL u32x4 *)(arena + byteOff); return uint4{ v.x, v.y, v.z, v.w }; }
__device__ __forceinline__ uint2 ld2(gptr_arena arena, uint32_t byteOff) { const u32x2 v = *(const CVX_GLOBAL u32x2 *)(arena + byteOff); return uint2{ v.x, v.y }; }
__device__ __forceinline__ uint32_t ld1(gptr_arena arena, uint32_t byteOff) { return *(const CVX_GLOBAL uint32_t *)(arena + byteOff); }
// Raybuffer tile: wave-uniform tile base + 32-bit byte offset (pixel row y of the lane's column: y * 256 + lane * 4)
typedef CVX_GLOBAL uint8_t *gptr_tile;
__device__ __forceinline__ void st_pixel(gptr_tile tile, uint32_t laneByteOff, int y, uint32_t argb) { *(CVX_GLOBAL uint32_t *)(tile + ((uint32_t)y * (CVX_WAVE * 4u) + laneByteOff)) = argb; }
// The skybox pass (half of all pixels: whole 256-byte rows of the tile, written once, read by nobody in this launch) stores non-temporally: the rows
// stream past the L2 instead of evicting world records from it.  Round 5, A/B on one box, three contexts per build: 13.30 .. 13.39 -> 13.18 .. 13.28 ms per 256 frames (-1.1 %).  The scattered 4-byte
// stores of the pixel loops must NOT: they rely on the L2 to combine neighbours into whole lines (`nt` there: +7.8 %).
__device__ __forceinline__ void st_pixel_stream(gptr_tile tile, uint32_t laneByteOff, int y, uint32_t argb) { __builtin_nontemporal_store(argb, (CVX_GLOBAL uint32_t *)(tile + ((uint32_t)y * (CVX_WAVE * 4u) + laneByteOff))); }
// (timing builds of rounds 2-4 -- no stores, no colour loads, constant texture index, lane-major tiles, flat addressing, no block-layout hints, no
// drain at the end of a drawn column -- are archived in tools/patches/exp_timing_switches.patch with their numbers in profiles/r02..r04_experiments.md)
#define st_pixel_loop st_pixel
#define ld_color ld1

// Byte offset (inside the level's table) of the 16-byte record of LOD column (cx, cz): row-major, cvx_device.h
__device__ __forceinline__ uint32_t record_offset(int cx, int cz, int rowShift)
{
	return (((uint32_t)cx << rowShift) + (uint32_t)cz) << 4;
}

// ---- Unity.Mathematics scalar semantics (math.cs 1.2.6) --------------------
__device__ __forceinline__ float m_min(float x, float y) { return (y != y || x < y) ? x : y; }
__device__ __forceinline__ float m_max(float x, float y) { return (y != y || x > y) ? x : y; }
__device__ __forceinline__ float m_lerp(float a, float b, float t) { return a + t * (b - a); }
__device__ __forceinline__ float m_sign(float x) { return (x > 0.0f ? 1.0f : 0.0f) - (x < 0.0f ? 1.0f : 0.0f); }
__device__ __forceinline__ int m_clampi(int x, int a, int b) { return max(a, min(b, x)); }

// C# (int)float compiled by Burst for x86 = cvttss2si: NaN / out of range give
// INT_MIN.  v_cvt_i32_f32 saturates instead, so the rule is explicit.
__device__ __forceinline__ int f2i(float x)
{
	// v_cvt_i32_f32 truncates, saturates and maps NaN to 0: only NaN and x >= 2^31 differ from the x86 rule
	int r;
	asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
	return (x < 2147483648.0f) ? r : (int)0x80000000;
}

// (int)floorf(x) with the same rule: v_cvt_flr_i32_f32 floors and converts in one instruction (saturating, NaN -> 0)
__device__ __forceinline__ int f2i_floor(float x)
{
	int r;
	asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
	return (x < 2147483648.0f) ? r : (int)0x80000000;
}

// Unity.Mathematics min / max as ONE instruction, for call sites where no operand can be a negative zero (or where the sign of
// a zero result cannot be observed): v_min_f32 / v_max_f32 return the other operand when one is a NaN, exactly like m_min /
// m_max above, and differ from them only in min(-0, +0) = -0 (m_min: +0), max(+0, -0) = +0 (m_max: -0) and for a SIGNALLING NaN
// operand (returned quieted; arithmetic only ever produces quiet NaNs and non-finite inputs are rejected by the library).  m_min costs two
// compares and a select, all three half-rate instructions on gfx950 (tools/valu_rate.hip).
__device__ __forceinline__ float hw_min(float x, float y)
{
	float r;
	asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
	return r;
}
__device__ __forceinline__ float hw_max(float x, float y)
{
	float r;
	asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
	return r;
}

// ---- exact f32 division without the scaling steps ---------------------------------------------------------------------
// The compiler expands a / b into v_div_scale (b), v_rcp, fma, fma [refined reciprocal r], v_div_scale (a), mul, fma, fma, fma
// [q = a * r, two residual corrections and a last residual], v_div_fmas, v_div_fixup.  When the magnitudes of a and b lie
// in [2^-30, 2^30], none of v_div_scale's cases applies (exponent difference < 96, nothing denormal, numerator not tiny): both
// return their operand unchanged with VCC = 0, v_div_fmas is then a plain fma and v_div_fixup passes its input through (finite
// non-zero operands, quotient in [2^-60, 2^60]).  What is left is rcp + 2 fma per DENOMINATOR and mul + 4 fma per NUMERATOR:
// the same operations on the same values in the same order, hence the same bits as `a / b` -- by construction, and checked
// against IEEE division by cvx_selftest_math ops 10 / 11.  A block that divides several numerators by one denominator pays
// the reciprocal once; operands outside the range (zero, denormal, huge, infinite, NaN) take the ordinary division.
struct Recip {
	float r, negd;
};
__device__ __forceinline__ bool div_safe(float x)
{
	return __builtin_amdgcn_fmed3f(fabsf(x), 0x1p-30f, 0x1p30f) == fabsf(x); // false for 0, denormals, NaN, infinities
}
__device__ __forceinline__ Recip recip_safe(float d)
{
	float r = __builtin_amdgcn_rcpf(d);
	const float e = __builtin_fmaf(-d, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	return { r, -d };
}
__device__ __forceinline__ float quot_safe(float n, Recip R)
{
	float q = n * R.r;
	float e = __builtin_fmaf(R.negd, q, n);
	q = __builtin_fmaf(e, R.r, q);
	e = __builtin_fmaf(R.negd, q, n);
	return __builtin_fmaf(e, R.r, q);
}

struct f3 {
	float x, y, z;
};
__device__ __forceinline__ f3 f3_madd(f3 a, f3 d, float s) { return { a.x + d.x * s, a.y + d.y * s, a.z + d.z * s }; } // a + d*s
__device__ __forceinline__ f3 f3_lerp(f3 a, f3 b, float t) { return { a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t }; }

// Block layout: conditions that hold for (almost) no / (almost) every lane on every kind of input -- exits, near-plane clipping, non-finite or
// denormal operands, LOD switches, columns with more than two solid runs -- are marked so that the common path is the fall-through one.  The
// column loop executes ~47 branches per step and is bound by its instruction stream: -1.7 % (32.8 -> 32.2 ms per 512 frames, round 2).
#define CVX_RARE(x) __builtin_expect(!!(x), 0)
#define CVX_USUAL(x) __builtin_expect(!!(x), 1)
// boolean algebra on lane masks in clip_world_bounds.  Written with `&&` / `||` on purpose: for these flags (every operand a plain compare result, no side
// effect to guard) the compiler turns the short-circuit forms into one s_and_b64 / s_or_b64 each -- the same code `&` / `|` on the converted
// ints give, measured (r03_experiments.md: -0.8 %); what must be avoided between lane masks is `?:`, which became a divergent branch.
#define CVX_AND(a, b) ((a) && (b))
#define CVX_OR(a, b) ((a) || (b))
#define CVX_FLOAT_EPSILON 1.401298464324817e-45f /* C# float.Epsilon (denormal), DrawSegmentRayJob.cs:220 */

// ---- SegmentDDAData (Assets/Code/Utils/SegmentDDAData.cs) ------------------
struct DDA {
	int px, pz;   // position
	int sx, sz;   // step
	float startX, startZ, dirX, dirZ, tDeltaX, tDeltaZ, tMaxX, tMaxZ;
	float distLast, distNext; // intersectionDistances.x / .y
};

__device__ __forceinline__ void dda_init(DDA &d, float startX, float startZ, float dirX, float dirZ) // :17-28
{
	d.startX = startX; d.startZ = startZ; d.dirX = dirX; d.dirZ = dirZ;
	d.px = f2i(floorf(startX));
	d.pz = f2i(floorf(startZ));
	d.tDeltaX = 1.0f / m_max(0.0000001f, fabsf(dirX));
	d.tDeltaZ = 1.0f / m_max(0.0000001f, fabsf(dirZ));
	float sgx = m_sign(dirX), sgz = m_sign(dirZ);
	d.sx = f2i(sgx);
	d.sz = f2i(sgz);
	d.tMaxX = (sgx * -(startX - floorf(startX)) + (sgx * 0.5f) + 0.5f) * d.tDeltaX;
	d.tMaxZ = (sgz * -(startZ - floorf(startZ)) + (sgz * 0.5f) + 0.5f) * d.tDeltaZ;
	d.distLast = m_max(d.tMaxX - d.tDeltaX, d.tMaxZ - d.tDeltaZ);
	d.distNext = m_min(d.tMaxX, d.tMaxZ);
}

// dirXNonNegative / dirZNonNegative = (dir.x >= 0), (dir.y >= 0): per-ray constants the caller evaluates once (they live in scalar
// lane masks, so the direction itself need not stay in vector registers for the whole column loop)
__device__ __forceinline__ void dda_next_lod(DDA &d, int currentVoxelSize, bool dirXNonNegative, bool dirZNonNegative) // :31-73
{
	int remX = d.px & (currentVoxelSize * 2 - 1);
	int remZ = d.pz & (currentVoxelSize * 2 - 1);
	float prevX = d.tMaxX - d.tDeltaX;
	float prevZ = d.tMaxZ - d.tDeltaZ;
	if (dirXNonNegative == (remX < currentVoxelSize)) { d.tMaxX += d.tDeltaX; } else { prevX -= d.tDeltaX; }
	if (dirZNonNegative == (remZ < currentVoxelSize)) { d.tMaxZ += d.tDeltaZ; } else { prevZ -= d.tDeltaZ; }
	d.distLast = m_max(prevX, prevZ);
	d.distNext = m_min(d.tMaxX, d.tMaxZ);
	d.px -= remX;
	d.pz -= remZ;
	d.tDeltaX *= 2.0f;
	d.tDeltaZ *= 2.0f;
	d.sx *= 2;
	d.sz *= 2;
}

__device__ __forceinline__ bool dda_step_to_world_intersection(DDA &d, float dimX, float dimZ) // :75-130
{
	const float inf = __builtin_inff();
	float invX = 1.0f / d.dirX, invZ = 1.0f / d.dirZ;
	float tminX = -inf, tminZ = -inf, tmaxX = inf, tmaxZ = inf;
	if (d.dirX != 0.0f) {
		float t1 = -d.startX * invX;
		float t2 = (dimX - d.startX) * invX;
		tminX = m_min(t1, t2);
		tmaxX = m_max(t1, t2);
	}
	if (d.dirZ != 0.0f) {
		float t1 = -d.startZ * invZ;
		float t2 = (dimZ - d.startZ) * invZ;
		tminZ = m_min(t1, t2);
		tmaxZ = m_max(t1, t2);
	}
	float tmint = m_max(tminX, tminZ);
	float tmaxt = m_min(tmaxX, tmaxZ);
	if (tmaxt < tmint || tmint <= 0.0f) {
		return false;
	}
	float tLastX, tLastZ;
	if (tminX < tminZ && tminX != -inf) {
		tLastZ = tminZ;
		float hit = d.startX + tmint * d.dirX;
		hit = d.dirX > 0.0f ? floorf(hit) : ceilf(hit);
		tLastX = (hit - d.startX) / d.dirX;
	} else {
		tLastX = tminX;
		float hit = d.startZ + tmint * d.dirZ;
		hit = d.dirZ > 0.0f ? floorf(hit) : ceilf(hit);
		tLastZ = (hit - d.startZ) / d.dirZ;
	}
	d.tMaxX = tLastX + d.tDeltaX;
	d.tMaxZ = tLastZ + d.tDeltaZ;
	d.distLast = m_max(tLastX, tLastZ);
	d.distNext = m_min(d.tMaxX, d.tMaxZ);
	float mid = m_lerp(d.distLast, d.distNext, 0.5f);
	d.px = f2i(floorf(d.startX + mid * d.dirX));
	d.pz = f2i(floorf(d.startZ + mid * d.dirZ));
	return true;
}

// The column loop's form of a ray's position: the column as ONE integer x * 65536 + z (LOD-0 coordinates, multiples of the voxel size of the
// current level; world dimensions <= 32768, checked at upload) and the byte offset of its record in the level's row-major table, both moved by
// per-ray constants when the DDA steps (cvx_device.h) -- two instructions each per step instead of a position pair and an address computation.
// "Outside the world" (:613 / World.cs:130-142, (p & dimensionMask) != p for x or z) is one mask test: a z below 0 borrows from x and leaves
// 65536 - voxel size in the low half, a z of dimZ sets the bit above maskZ, an x below 0 makes the sum negative, an x of dimX sets the bit
// above maskX << 16.
struct ColumnCursor {
	int pos, posStepX, posStepZ;      // x * 65536 + z and what a step along x / z adds to it
	uint32_t rec;                     // arena byte offset of the column's record
	int recStepX, recStepZ;           // +- (16 << rowShift), +- 16
};
__device__ __forceinline__ void cursor_set(ColumnCursor &c, const DDA &d, const DevWorldLevel &level, int maskX, int maskZ)
{
	c.pos = d.px * 65536 + d.pz;
	c.posStepX = d.sx * 65536;
	c.posStepZ = d.sz;
	c.rec = level.recordsOff + record_offset((d.px & maskX) >> level.shift, (d.pz & maskZ) >> level.shift, level.rowShift); // clamped into the table
	c.recStepX = (d.sx >> level.shift) * (16 << level.rowShift); // (sx = +- voxel size, or 0 for a ray that never steps along x)
	c.recStepZ = (d.sz >> level.shift) * 16;
}

__device__ __forceinline__ bool dda_step(DDA &d, float farClip) // :135-150
{
	// branch-free form of "if (tMax.x < tMax.y) step x else step z" (same values, no exec-mask juggling)
	const bool stepX = d.tMaxX < d.tMaxZ;
	const float crossed = stepX ? d.tMaxX : d.tMaxZ;
	const float nextX = d.tMaxX + d.tDeltaX, nextZ = d.tMaxZ + d.tDeltaZ;
	d.tMaxX = stepX ? nextX : d.tMaxX;
	d.tMaxZ = stepX ? d.tMaxZ : nextZ;
	d.px += stepX ? d.sx : 0;
	d.pz += stepX ? 0 : d.sz;
	d.distLast = crossed;
	d.distNext = hw_min(d.tMaxX, d.tMaxZ); // tMax values are sums of positive terms: no negative zero, m_min == v_min_f32
	return crossed >= farClip;
}

// the same step for the column loop: the position moves in its cursor form (d.px / d.pz / d.sx / d.sz are not touched)
__device__ __forceinline__ bool dda_step_cursor(DDA &d, ColumnCursor &c, float stopDistance) // true: the distance at which the column loop has to look up (far clip or LOD boundary) is reached
{
	const bool stepX = d.tMaxX < d.tMaxZ;
	// the distance crossed is the smaller tMax (:139, :145) -- which is what distNext already holds: every place that sets tMax sets distNext to the
	// minimum of the two (here, dda_init, dda_next_lod, dda_step_to_world_intersection; tMax values are positive sums, so min and "x < z ? x : z" agree)
	const float crossed = d.distNext;
	const float nextX = d.tMaxX + d.tDeltaX, nextZ = d.tMaxZ + d.tDeltaZ;
	d.tMaxX = stepX ? nextX : d.tMaxX;
	d.tMaxZ = stepX ? d.tMaxZ : nextZ;
	c.pos += stepX ? c.posStepX : c.posStepZ;
	c.rec += (uint32_t)(stepX ? c.recStepX : c.recStepZ);
	d.distLast = crossed;
	d.distNext = hw_min(d.tMaxX, d.tMaxZ);
	return crossed >= stopDistance;
}

// ---- CameraData helpers (Assets/Code/Utils/CameraData.cs) ------------------
// `finv` = 1 / frustum, which the reference computes inside these two (:103,111): the frustum is one of the two window bounds, so
// the caller divides once per bound and passes the quotient in.
__device__ __forceinline__ float clip_min(f3 pMin, f3 pMax, float finv) // :101-107
{
	float c0 = 1.0f * pMax.z - finv * pMax.x;
	float c1 = 1.0f * pMin.z - finv * pMin.x;
	return 1.0f - (c0 / (c0 - c1));
}
__device__ __forceinline__ float clip_max(f3 pMin, f3 pMax, float finv) // :109-115
{
	float c0 = 1.0f * pMax.z - finv * pMax.x;
	float c1 = 1.0f * pMin.z - finv * pMin.x;
	return c1 / (c1 - c0);
}

// GetWorldBoundsClippingCamSpace, :51-99.  true = entirely outside (the lerps are then unused by the caller).
// Flattened form of the reference's if-tree: with a1/a2 = pMin/pMax above fMax and b1/b2 = pMin/pMax below fMin the tree
// evaluates at most one "min" clip (against fMax when a1, else against fMin when b1) and at most one "max" clip (against
// fMax when a2 and not a1, else against fMin when b2); the arithmetic of each is the tree's own, so the values are the
// same, but divergent lanes of a wave now share one instance of each division sequence instead of six.
// `straddles` = pMin lies below the window and pMax above it (the ordinary view of a world column: its foot under the lowest free
// pixel, its top over the highest): minLerp then comes from the clip against fMin, maxLerp from the clip against fMax, nothing is clipped away.
__device__ __forceinline__ bool clip_world_bounds(f3 pMin, f3 pMax, float fMin, float fMax, float invFMin, float invFMax, float &minLerp, float &maxLerp, bool &straddles)
{
	const bool a1 = pMin.x > pMin.z * fMax;
	const bool a2 = pMax.x > pMax.z * fMax;
	const bool b1 = pMin.x < pMin.z * fMin;
	const bool b2 = pMax.x < pMax.z * fMin;
	// (CVX_AND / CVX_OR, never `?:` between flags: a ternary of two lane masks is compiled into a divergent branch -- saveexec, a handful of
	// scalar instructions, restore -- to compute one bit)
	const bool n1 = !a1, n2 = !a2;
	straddles = CVX_AND(CVX_AND(n1, b1), a2);
	const bool needMin = CVX_OR(a1, b1);
	const bool needMax = CVX_OR(CVX_AND(a1, b2), CVX_AND(n1, CVX_OR(a2, b2)));
	const float lo = clip_min(pMin, pMax, a1 ? invFMax : invFMin);
	const float hi = clip_max(pMin, pMax, CVX_AND(n1, a2) ? invFMax : invFMin);
	minLerp = needMin ? lo : 0.0f;
	maxLerp = needMax ? hi : 1.0f;
	return CVX_OR(CVX_AND(a1, a2), CVX_AND(CVX_AND(n1, n2), CVX_AND(b1, b2)));
}

// ---- texture row of a side pixel (DrawSegmentRayJob.cs:524-531) -------------------------------------------------------
// The reference: l = unlerp(bounds, y), u = lerp(uvA.y, uvB.y, l) / lerp(uvA.x, uvB.x, l), row = (int)floor(u) -- two IEEE divisions per pixel.
__device__ __forceinline__ int tex_row_exact(int y, float boundsX, float boundsY, float uvAx, float uvBx, float uvAy, float uvBy)
{
	float l = ((float)y - boundsX) / (boundsY - boundsX); // unlerp
	float wux = m_lerp(uvAx, uvBx, l);
	float wuy = m_lerp(uvAy, uvBy, l);
	float u = wuy / wux;
	return f2i_floor(u);
}
// Round 5: only floor(u) is ever used, so u is first computed the cheap way (hardware reciprocals, fused multiply-adds: u') together with a bound D on
// |u - u'| that covers every rounding of both computations; where u' lies farther than D from the nearest integer, u and u' have the same floor
// (`certain`).  The other pixels (a few per thousand) take the reference's divisions.  Derivation (e = 2^-24; l, wux, wuy as the reference rounds them
// against the primed ones here; n = y - boundsX and d = boundsY - boundsX are the same floats in both):
//   l = fl(n / d), l' = fl(n * rcp(d)), rcp within one ulp                              =>  |l - l'| <= 2^-22 |n / d|
//   wux = fl(uvA.x + fl(l * a1)), wux' = fma(l', a1, uvA.x), a1 = fl(uvB.x - uvA.x)       =>  |wux - wux'| <= 9 e (|uvA.x| + |l'| |a1|) =: Ex;  Ey likewise
//   if Ex <= |wux'| / 2:  |wuy / wux - wuy' / wux'| <= 2 (Ey + |wuy' / wux'| Ex) / |wux'|;  the final division against rcp + multiply: 4.2 e |u'| more
//   D = (Sy + (|u'| + 1) Sx) |rcp(wux')| + 8 e |u'|  with Sx = 40 e (|uvA.x| + |l'| |a1|) >= 2.2 x (2 Ex), Sy likewise.  The "+ 1" makes D >= 1 (no
//   pixel is certain) whenever Ex > |wux'| / 2; |u'| >= 2^21 gives D > 1/2 too, so a certain u' is far inside the int range (where the x86 rule of
//   f2i_floor is the plain floor); a NaN or an infinity anywhere makes the comparison false.  Outside the normal range: |d| > 2^100 makes every pixel of the
//   run uncertain, 2^-110 |wux'| in D does the same for |wux'| > 2^126 (denormal reciprocals), and Sx >= 2^-140 covers what underflowing products lose.
//   cvx_selftest_math op 16 compares the two forms on the device (tests: 2^24 samples incl. float soup, no certain row may differ).
struct TexRun { // per run: what the cheap row needs of (boundsX, boundsY, uvA, uvB)
	float rd, a1, a2, a1s, a2s, bxs, bys;
};
__device__ __forceinline__ TexRun tex_run(float boundsX, float boundsY, float uvAx, float uvBx, float uvAy, float uvBy)
{
	TexRun T;
	const float d = boundsY - boundsX;
	T.rd = fabsf(d) <= 0x1p100f ? __builtin_amdgcn_rcpf(d) : __builtin_nanf(""); // (a reciprocal below 2^-126 is not "within one ulp": no pixel of such a run is certain)
	T.a1 = uvBx - uvAx;
	T.a2 = uvBy - uvAy;
	const float k = 40.0f * 0x1p-24f;
	T.a1s = k * fabsf(T.a1);
	T.a2s = k * fabsf(T.a2);
	T.bxs = __builtin_fmaf(k, fabsf(uvAx), 0x1p-140f); // (the floor covers what an underflowing product can lose: 2^-149 per operation)
	T.bys = k * fabsf(uvAy);
	return T;
}
__device__ __forceinline__ int tex_row_cheap(int y, float boundsX, float uvAx, float uvAy, const TexRun &T, bool &certain)
{
	const float n = (float)y - boundsX;
	const float lq = n * T.rd;
	const float wx = __builtin_fmaf(lq, T.a1, uvAx), wy = __builtin_fmaf(lq, T.a2, uvAy);
	const float r = __builtin_amdgcn_rcpf(wx);
	const float uq = wy * r;
	const float sx = __builtin_fmaf(fabsf(lq), T.a1s, T.bxs), sy = __builtin_fmaf(fabsf(lq), T.a2s, T.bys);
	// (the |wux'| term: beyond 2^126 its reciprocal is denormal, not "within one ulp" -- the bound then exceeds every distance)
	const float bound = __builtin_fmaf(__builtin_fmaf(fabsf(uq), sx, sy + sx), fabsf(r), __builtin_fmaf(fabsf(wx), 0x1p-110f, 0x1p-21f * fabsf(uq)));
	certain = fabsf(uq - rintf(uq)) > bound;
	int row;
	asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(row) : "v"(uq));
	return row;
}

// ---- seen-pixel bitmask in LDS ---------------------------------------------
// Word w of a lane lives at seen[w << sshift]: the words of the wave's lanes are interleaved with a stride of 64, or of
// the lane count of a sub-tile (a tile whose window needs many words is rendered by narrower waves, so that every
// launch stays within 10 KB of LDS per wave = 16 resident waves per CU, see DrawBatch).
// first unseen pixel >= start, or omax+1; start unchanged when start > omax
// (the reference's while loop at :407 / :678 does not run then).
__device__ __forceinline__ int scan_up(const uint32_t *seen, int sshift, int start, int omax)
{
	// (one way through: the word index clamped into the window, the special results as selects)
	const int wend = omax >> 5;
	int w = min(start >> 5, wend);
	uint32_t m = ~seen[w << sshift] & (0xFFFFFFFFu << (start & 31));
	if (CVX_RARE(((int)(m == 0u) & (int)(w < wend)) != 0)) { // (walking on to further words is the exception: laid out off the common path)
		do {
			w++;
			m = ~seen[w << sshift];
		} while (m == 0u && w < wend);
	}
	const int pos = (w << 5) + (__ffs((int)m) - 1);
	const int found = (m == 0u || pos > omax) ? omax + 1 : pos;
	return start > omax ? start : found;
}

// last unseen pixel <= start, or omin-1; start unchanged when start < omin (:413 / :690).
__device__ __forceinline__ int scan_down(const uint32_t *seen, int sshift, int start, int omin)
{
	const int wbeg = omin >> 5;
	int w = max(start >> 5, wbeg);
	uint32_t m = ~seen[w << sshift] & (0xFFFFFFFFu >> (31 - (start & 31)));
	if (CVX_RARE(((int)(m == 0u) & (int)(w > wbeg)) != 0)) {
		do {
			w--;
			m = ~seen[w << sshift];
		} while (m == 0u && w > wbeg);
	}
	const int pos = (w << 5) + (31 - __clz((int)m));
	const int found = (m == 0u || pos < omin) ? omin - 1 : pos;
	return start < omin ? start : found;
}

// bits of word w that fall inside [lo, hi]
__device__ __forceinline__ uint32_t range_mask(int w, int lo, int hi)
{
	int base = w << 5;
	uint32_t m = 0xFFFFFFFFu;
	if (lo > base) { m &= 0xFFFFFFFFu << (lo - base); }
	if (hi < base + 31) { m &= 0xFFFFFFFFu >> (base + 31 - hi); }
	return m;
}

// ReducePixelHorizon, DrawSegmentRayJob.cs:660-697
__device__ __forceinline__ void reduce_pixel_horizon(const uint32_t *seen, int sshift, int omin, int omax, int &rbMin, int &rbMax, int &nfMin, int &nfMax,
                                                     float &frustumBoundsMin, float &frustumBoundsMax)
{
	// (each clamp as a max / min and ONE divergent region per side instead of two nested ones)
	const bool raiseMin = ((int)(rbMin <= nfMin) & (int)(rbMax >= nfMin)) != 0;
	rbMin = max(rbMin, nfMin);
	if (raiseMin) {
		nfMin = scan_up(seen, sshift, rbMax + 1, omax);
		frustumBoundsMin = (float)nfMin - 0.501f;
	}
	const bool lowerMax = ((int)(rbMax >= nfMax) & (int)(rbMin <= nfMax)) != 0;
	rbMax = min(rbMax, nfMax);
	if (lowerMax) {
		nfMax = scan_down(seen, sshift, rbMin - 1, omin);
		frustumBoundsMax = (float)nfMax + 0.501f;
	}
}

// ---- diagnostic build only (-DCVX_PROFILE_SECTIONS): per-lane cycle accounting of code sections ----
// CVX_BEGIN() stamps, CVX_END(n) adds the cycles since the last stamp to section n -- in every lane that is active
// at both points.  Sections never nest.  Cost: one s_memtime + two VALU ops per mark (the build is only read for
// shares).  Reported per section: max over the lanes of a wave (~ wave time in the section), summed over waves, and
// the lane sum (lane-cycles; / (64 * wave time) = lane utilisation).
#ifdef CVX_PROFILE_SECTIONS
#define CVX_NSEC 16
__device__ unsigned long long g_sectionCycles[32]; // [n] wave cycles, [16+n] lane cycles / 64
struct ProfLane {
	unsigned int last;
	unsigned int acc[CVX_NSEC];
#ifdef CVX_PROFILE_COUNTS
	unsigned int lanes[CVX_NSEC]; // how often THIS lane was active at CVX_COUNT(n); summed over lanes -> [16 + n]
#endif
};
#ifdef CVX_PROFILE_COUNTS
// counting variant: [n] = number of times a wave executed the code at CVX_COUNT(n) (exactly one lane of the executing
// wave increments), no time stamps
// plus a histogram of the number of active lanes per execution, 8 buckets of 8 lanes: g_sectionHist[n * 8 + (active - 1) / 8]
__device__ unsigned long long g_sectionHist[CVX_NSEC * 8];
#define CVX_COUNT(n) do { prof.lanes[n]++; const unsigned long long b_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)b_) - 1) { prof.acc[n]++; atomicAdd(&g_sectionHist[(n) * 8 + (__popcll(b_) - 1) / 8], 1ull); } } while (0)
#define CVX_BEGIN() ((void)0)
#define CVX_END(n) ((void)0)
#define CVX_WAITPROBE(n) ((void)0)
#else
#define CVX_COUNT(n) ((void)0)
#define CVX_BEGIN() (prof.last = (unsigned int)__builtin_amdgcn_s_memtime())
#define CVX_END(n) do { const unsigned int t_ = (unsigned int)__builtin_amdgcn_s_memtime(); prof.acc[n] += t_ - prof.last; prof.last = t_; } while (0)
// drains the vector-memory queue and books the time it took under section n (how long the wave would wait for memory here)
#define CVX_WAITPROBE(n) do { CVX_END(15); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CVX_END(n); } while (0)
#endif
#else
#define CVX_COUNT(n) ((void)0)
struct ProfLane {};
#define CVX_BEGIN() ((void)0)
#define CVX_END(n) ((void)0)
#define CVX_WAITPROBE(n) ((void)0)
#endif

#ifdef CVX_TILE_TIMES /* diagnostic build: clock ticks each launched wave lived, for the launch-order experiment (tools/lpt_oracle.py) */
__device__ unsigned long long *g_tileTimes;
#endif

struct LaneCounters {
	unsigned int S, E, C, P;
	unsigned int lod[6];
};

// ---------------------------------------------------------------------------
// One ray: TraceToFirstColumnJob + ExecuteRay.  Writes colour pixels to
// out[y*64] and marks them in seen[]; every exit path of the reference ends in
// WriteSkybox/WriteSkyboxFull, which the caller performs for the whole wave.
// ---------------------------------------------------------------------------
template <int DIR, bool COUNT>
__device__ __forceinline__ void trace_ray(const DevFrame &F, const DevSegment &S, const DevWorld *__restrict__ world, int planeRayIndex,
                                          uint32_t *seen /* &lds[lane] */, int sshift /* log2 of the mask word stride */, gptr_tile tileOut, uint32_t laneByteOff, LaneCounters &cnt, ProfLane &prof)
{
	(void)prof;
	const int omin = S.omin, omax = S.omax;
	const float farClip = F.farClip;
	const float posY = F.posY;

	// ---- DDASetupJob.Execute, :58-76
	DDA ray;
	{
		float endRayLerp = (float)planeRayIndex / (float)S.rayCount;
		float dx = m_lerp(S.rayMinX, S.rayMaxX, endRayLerp);
		float dz = m_lerp(S.rayMinZ, S.rayMaxZ, endRayLerp);
		float r = 1.0f / sqrtf(dx * dx + dz * dz); // math.normalize = rsqrt(dot) * v, rsqrt = 1/sqrt
		dda_init(ray, F.posX, F.posZ, r * dx, r * dz);
	}

	const bool dirXNonNegative = ray.dirX >= 0.0f, dirZNonNegative = ray.dirZ >= 0.0f; // SegmentDDAData.cs:41,57

	// ---- TraceToFirstColumnJob.Execute, :95-143
	int lod = 0;
	float lodMax = F.lod[0];
	const int dimX = world->dimX, dimZ = world->dimZ;
	if (ray.px < 0 || ray.pz < 0 || ray.px >= dimX || ray.pz >= dimZ) {
		if (!dda_step_to_world_intersection(ray, (float)dimX, (float)dimZ)) {
			return; // WriteSkyboxFull
		}
		// (the threshold of the last level is taken as +infinity -- `lod < 5` of :237 / :101, a memory-safety guard only: such a ray is beyond far clip
		// anyway -- so the test of every column step below is ONE compare)
		while (ray.distLast >= lodMax) {
			dda_next_lod(ray, 1 << lod, dirXNonNegative, dirZNonNegative);
			lod++;
			{ const float next_ = F.lod[min(lod, 5)]; lodMax = lod < 5 ? next_ : __builtin_inff(); }
		}
		if (m_min(ray.tMaxX, ray.tMaxZ) >= farClip) { // IsBeyondFarClip, SegmentDDAData.cs:152
			return; // WriteSkyboxFull
		}
	}

	// ---- ExecuteRay, :195-620
	int voxelScale = 1 << lod;
	DevWorldLevel L = world->level[lod];
	const gptr_arena arena = (gptr_arena)world->arena;
	const int maskX = world->maskX, maskZ = world->maskZ;
	const int worldMaxYInt = world->dimY;
	const float worldMaxY = (float)worldMaxYInt;
	const float cameraPosYNormalized = posY / worldMaxY;
	const float invWorldMaxY = 1.0f / worldMaxY; // exact: dimY is a power of two

	int nextFreePixelMin = omin;
	int nextFreePixelMax = omax;
	float frustumBoundsMin = (float)nextFreePixelMin - 0.501f;
	float frustumBoundsMax = (float)nextFreePixelMax + 0.501f;
	float frustumDirMaxWorld = CVX_FLOAT_EPSILON;
	float frustumDirMinWorld = CVX_FLOAT_EPSILON;

	// SetupProjectedPlaneParams, :622-651: rows (x or y), z, w of M applied to
	// (start.x, 0 | worldMaxY, start.z, 1) and (dir.x, 0, dir.z, 0).
	f3 planeStartBottom, planeStartTop, planeDir;
	{
		const float *M = F.M;
		const int r0 = S.axisMappedToY ? 1 : 0;
		const float sx = ray.startX, sz = ray.startZ;
		// mul(float4x4, float4) = c0*x + c1*y + c2*z + c3*w, left to right
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

	// A DDA walk is monotone in x and z, so it leaves the world after at most
	// dimX + dimZ column visits; the cap can never bind on valid input and only
	// keeps a wave from spinning on non-finite camera data.
	int guardSteps = dimX + dimZ + 16;

	// The column being processed ("cur") and the values of the DDA / LOD state that belong to it; `ray` itself
	// already stands on the NEXT column, whose 16-byte record is in flight while this one is processed.
	uint4 rec;                        // 16-byte record of the current column (cvx_device.h)
	uint4 recB;                       // ... and of the next one (the column loop alternates between the two: no copy at the end of a step)
	float curDistLast, curDistNext;   // ray.IntersectionDistances of the current column
	unsigned int consumed = 0u;       // counting variant: elements the reference's walk has dereferenced in this column
	float worldBoundsMin, worldBoundsMax;

	// Clip, element walk and pixel writes of ExecuteRay (:289-611) for the current column;
	// false = the ray is finished (every such exit is WriteSkybox).
	auto drawColumn = [&](const uint4 &rec, uint32_t recAddr) -> bool {
		(void)recAddr;
		// codes 1 .. 3: the record holds the column's solid runs; code 0 (of a column that is not empty): they live in its block of the run list (cvx_device.h)
		const bool listed = rec.x < 0x40000000u;
		int solidCount = (int)(rec.x >> 30);
		if (COUNT) { solidCount = listed ? (int)rec.w : solidCount; }
		CVX_BEGIN();
		if (COUNT) { consumed = 0u; }
		const uint32_t columnRunsOff = L.runsOff + rec.z * 8u; // solid run j (top-down numbering) of a listed column lives at entry j of its block (only listed lanes use it)
uint2 countInfo = uint2{ 0u, 0u }; // counting build: {RunCount | elementIndex of run 0 << 16, elementIndex of run 1 | elementIndex of run 2 << 16}
		if (COUNT) { countInfo = ld2(arena, L.countsOff + ((recAddr - L.recordsOff) >> 1)); }
		// :289-293
		const f3 camSpaceMinLast = f3_madd(planeStartBottom, planeDir, curDistLast);
		const f3 camSpaceMinNext = f3_madd(planeStartBottom, planeDir, curDistNext);
		const f3 camSpaceMaxLast = f3_madd(planeStartTop, planeDir, curDistLast);
		const f3 camSpaceMaxNext = f3_madd(planeStartTop, planeDir, curDistNext);

		CVX_COUNT(8);
		bool windowClosed = false; // :399-403: the clipped column lies outside the free pixel window -> the ray is finished
		if (curDistLast > 2.0f && frustumDirMaxWorld == CVX_FLOAT_EPSILON) { // :295-422
			CVX_COUNT(2);
			float clipLastMinLerp, clipLastMaxLerp, clipNextMinLerp, clipNextMaxLerp;
			// CameraData.cs:103,111.  frustumBounds = (integer pixel in [-1, 16385]) -/+ 0.501: magnitude in [0.499, 16386], always "safe"
			const float invFrustumMin = quot_safe(1.0f, recip_safe(frustumBoundsMin)), invFrustumMax = quot_safe(1.0f, recip_safe(frustumBoundsMax));
			bool straddlesLast, straddlesNext;
			bool clippedLast, clippedNext;
			{
				const auto straddle = [&](f3 pMin, f3 pMax) { return ((int)!(pMin.x > pMin.z * frustumBoundsMax) & (int)(pMin.x < pMin.z * frustumBoundsMin) & (int)(pMax.x > pMax.z * frustumBoundsMax)) != 0; };
				const bool both = ((int)straddle(camSpaceMinLast, camSpaceMaxLast) & (int)straddle(camSpaceMinNext, camSpaceMaxNext)) != 0;
				// Round 5: when EVERY lane that clips here sees the column's foot below and its top above the window at both ends -- clip_world_bounds' straddle case, the
				// ordinary view of a world column -- the four lerps are its clip_min against frustumBoundsMin and clip_max against frustumBoundsMax, nothing is
				// clipped away, and the wave skips the selects and the flag algebra of the general form (same divisions on the same operands: -0.9 %)
				if (__ballot(!both) == 0ull) {
					clipLastMinLerp = clip_min(camSpaceMinLast, camSpaceMaxLast, invFrustumMin);
					clipLastMaxLerp = clip_max(camSpaceMinLast, camSpaceMaxLast, invFrustumMax);
					clipNextMinLerp = clip_min(camSpaceMinNext, camSpaceMaxNext, invFrustumMin);
					clipNextMaxLerp = clip_max(camSpaceMinNext, camSpaceMaxNext, invFrustumMax);
					clippedLast = clippedNext = false;
					straddlesLast = straddlesNext = true;
				} else {
					clippedLast = clip_world_bounds(camSpaceMinLast, camSpaceMaxLast, frustumBoundsMin, frustumBoundsMax, invFrustumMin, invFrustumMax, clipLastMinLerp, clipLastMaxLerp, straddlesLast);
					clippedNext = clip_world_bounds(camSpaceMinNext, camSpaceMaxNext, frustumBoundsMin, frustumBoundsMax, invFrustumMin, invFrustumMax, clipNextMinLerp, clipNextMaxLerp, straddlesNext);
				}
			}

			// (:297-299 leaves here when both ends are outside the window; that exit is taken together with the next one below --
			// nothing in between has an effect that survives the end of the ray)
			// :300-390, the three cases (only Next visible / only Last visible / both) folded into selects: each
			// bound comes from the Last or the Next intersection, chosen exactly as the reference's branches do.
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
			worldBoundsMin = floorf(worldBoundsMin);
			worldBoundsMax = ceilf(worldBoundsMax);

			// :337-421 project the four clipped points (x / z), order them, and take floor(min) / ceil(max) as the writable pixel range, which
			// can end the ray (:399-403) or move nextFreePixelMin / Max inwards (:405-416).  Only those two INTEGERS are ever used.  In the
			// ordinary case -- both ends straddle the window, so the min points were clipped against frustumBoundsMin = k - 0.501 and the max
			// points against frustumBoundsMax = m + 0.501 (k <= nextFreePixelMin <= nextFreePixelMax <= m, integers: ReducePixelHorizon :682,694) --
			// a clipped point lies ON its bound up to rounding.  If the residual |x - f * z| (as computed, error < 2^-23 |f z| <= 0.002 |z|) is
			// below 0.4 |z|, then x / z = f + r / z lies within 0.402 of f, the correctly rounded quotient too (rounding is monotone), hence
			// floor(min) = k - 1 and ceil(max) = m + 1 whatever the exact quotients are: min < max (no swap, :339-346), the range contains
			// [nextFreePixelMin, nextFreePixelMax] (no exit, nothing moves), and neither end was clipped away.  So nothing of :337-421 has any
			// effect and the four divisions are not needed.  Any lane for which this cannot be shown takes the reference's path below.
			const auto onBound = [](f3 p, float f) { return fabsf(p.x - f * p.z) < 0.4f * fabsf(p.z); };
			// (`&`, not `&&`: six short tests evaluated straight-line instead of a chain of divergent branches)
			const bool windowUntouched = !COUNT && ((int)straddlesLast & (int)straddlesNext & (int)onBound(minClipA, frustumBoundsMin) & (int)onBound(minClipB, frustumBoundsMin) &
			                                        (int)onBound(maxClipA, frustumBoundsMax) & (int)onBound(maxClipB, frustumBoundsMax)) != 0;
			if (!CVX_USUAL(windowUntouched)) {
				float minNext = minClipB.x / minClipB.z;
				float minLast = minClipA.x / minClipA.z;
				float maxNext = maxClipB.x / maxClipB.z;
				float maxLast = maxClipA.x / maxClipA.z;
				if (maxNext < minNext) { float t = maxNext; maxNext = minNext; minNext = t; }
				if (maxLast < minLast) { float t = maxLast; maxLast = minLast; minLast = t; }
				// (hw_min / hw_max: the results only go through floor / ceil and (int), which map -0 and +0 to the same 0)
				const float bothMin = hw_min(minLast, minNext), bothMax = hw_max(maxLast, maxNext); // (computed ahead of the selects: an asm inside a select becomes a branch)
				const float camSpaceClippedMin = clippedLast ? minNext : (clippedNext ? minLast : bothMin);
				const float camSpaceClippedMax = clippedLast ? maxNext : (clippedNext ? maxLast : bothMax);

				const int writableMinPixel = f2i_floor(camSpaceClippedMin);
				const int writableMaxPixel = f2i(ceilf(camSpaceClippedMax));

				if (CVX_RARE((clippedLast && clippedNext) || writableMaxPixel < nextFreePixelMin || writableMinPixel > nextFreePixelMax)) {
					if (COUNT) { return false; }
					windowClosed = true; // (rendering build: no early return out of the lambda -- the element loop below gets nothing to do)
				}
				if (writableMinPixel > nextFreePixelMin) {
					nextFreePixelMin = scan_up(seen, sshift, writableMinPixel, omax);
				}
				if (writableMaxPixel < nextFreePixelMax) {
					nextFreePixelMax = scan_down(seen, sshift, writableMaxPixel, omin);
				}
				if (COUNT && nextFreePixelMin > nextFreePixelMax) {
					return false; // :419 (the rendering build notices at the end of the column)
				}
			}
		}

		CVX_END(2);
		// ---- element loop, :424-611
		// The reference walks all RLE elements of the column, starting from worldMaxY downwards (ITERATION_DIRECTION +1)
		// or from 0 upwards (-1), :428-455; air runs only move the bounds (:457-459).  The records list the solid runs
		// with the distance the walk has covered when it reaches them (an integer number of voxels, so the float
		// bounds the reference accumulates are exactly these integers), so only solid runs are iterated here.
		float elementBoundsMin, elementBoundsMax;
		int solidIndex = 0;
		const uint32_t worldColumnColorsOff = L.elementsOff + (rec.x & 0x3FFFFFFFu) * 4u; // ColorPointer, World.cs:185

		// Rendering build: which solid runs the walk below would project is decided without walking.  A run is projected iff it is
		// neither entirely above worldBoundsMax (:461-467) nor entirely below worldBoundsMin (:468-475) -- the reference's early
		// `break`s only skip runs that these two tests would skip anyway (the runs of a column are sorted by height), and the
		// world bounds do not change inside the element loop.  So the column's first two runs are tested here, straight-line,
		// and the loop below just takes the visible ones in walk order; further runs (3-8 % of the columns have a third one, which
		// its record holds as well; a few per thousand more, in the run list) are scanned when their turn comes.  Same runs, same order per lane as the walk of the counting build (and the oracle).
		bool vis0 = false, vis1 = false;
		int ovNext = 0; // next run beyond the first two to look at: 2, 3, ... (top-down walk) or solidCount - 1, ... 2 (bottom-up walk)
		// world-space span of a run word {bottomY | (topY - 1) << 16} (cvx_device.h): LOD-0 voxels, integers below 2^17 -- exact as floats, and the
		// numbers the reference accumulates (see the walk below)
		auto runSpan = [&](uint32_t w0, float &bottom, float &top) {
			bottom = (float)(w0 & 0xFFFFu);
			top = (float)(w0 >> 16) + 1.0f;
		};
		// RLEElement.Length of that run: its height in voxels of this LOD
		auto runLength = [&](uint32_t w0) -> int { return (int)(((w0 >> 16) + 1u - (w0 & 0xFFFFu)) >> lod); };
		// spans and ColorsIndex of the column's first two solid runs.  A record with its runs (cvx_device.h): run 0 = [w.lo, worldMax], run 1 = [z.lo, w.hi + 1],
		// ColorsIndex = the lengths of the solid runs above (run 0: 0, run 1: Length of run 0); a listed column brings all of that in its block of the run list.
		float b0 = 0.f, t0 = 0.f, b1 = 0.f, t1 = 0.f;
		const int length0 = (int)((rec.y >> 16) - (rec.w & 0xFFFFu)) >> lod; // RLEElement.Length of run 0
		uint32_t colorsIndex0 = 0u, colorsIndex1 = (uint32_t)length0;
		if (!COUNT) {
			b0 = (float)(rec.w & 0xFFFFu);
			t0 = (float)(rec.y >> 16);
			b1 = (float)(rec.z & 0xFFFFu);
			t1 = (float)(rec.w >> 16) + 1.0f;
			if (CVX_RARE(listed)) { // (a handful of columns per thousand: their first two runs are fetched where they are needed)
				const uint4 listHead = ld4(arena, columnRunsOff);
				solidCount = (int)rec.w;
				runSpan(listHead.x, b0, t0);
				runSpan(listHead.z, b1, t1);
				colorsIndex0 = listHead.y & 0xFFFFu;
				colorsIndex1 = listHead.w & 0xFFFFu;
				if (solidCount < 1) { b0 = __builtin_inff(); } // a column of air runs only (RunCount > 0, nothing solid): nothing to project
			}
			const bool in0 = ((int)!(b0 > worldBoundsMax) & (int)!(t0 < worldBoundsMin)) != 0, in1 = ((int)(solidCount > 1) & (int)!(b1 > worldBoundsMax) & (int)!(t1 < worldBoundsMin)) != 0;
			vis0 = !windowClosed && in0; // (a column that is drawn has at least one solid run)
			vis1 = !windowClosed && in1;
			ovNext = DIR > 0 ? 2 : solidCount - 1;
		}
		bool ovPending = !COUNT && !windowClosed && solidCount > 2; // runs beyond the first two still to be looked at

		// Element loop :441-611, realigned for SIMT: every lane first walks its own elements (cheap: decode,
		// bounds bookkeeping, air / world-bounds culls :445-475) up to its next run that has to be projected;
		// the expensive projection + pixel code below then runs once for all lanes that found one, instead of
		// once per element index with whatever lanes happen to hold a solid run at that index.  The sequence of
		// elements each lane consumes is unchanged.
		// (rendering build: a lane stays in the loop exactly as long as it has a visible run left -- no trailing "nothing found" pass)
		while (COUNT || vis0 || vis1 || CVX_RARE(ovPending)) {
			int elementColorsIndex, elementLength; // (no initial values: read only once `found`, which every path that sets it writes them before)
			bool found = false;
			CVX_BEGIN();
			CVX_WAITPROBE(10);
			// runs beyond the first two: rare, so the scan sits behind one branch (bottom-up they come first, top-down last)
			auto scanOverflow = [&]() {
				while (DIR > 0 ? ovNext < solidCount : ovNext >= 2) {
					// (a record with three runs: run 2 = [worldMin, z.hi + 1], its ColorsIndex = Length of run 0 + Length of run 1; no memory access)
					uint2 run = uint2{ (rec.y & 0xFFFFu) | (rec.z & 0xFFFF0000u), (uint32_t)(length0 + ((int)((rec.w >> 16) + 1u - (rec.z & 0xFFFFu)) >> lod)) };
					if (CVX_RARE(listed)) { run = ld2(arena, columnRunsOff + (uint32_t)ovNext * 8u); }
					ovNext += DIR > 0 ? 1 : -1;
					runSpan(run.x, elementBoundsMin, elementBoundsMax);
					if (!(elementBoundsMin > worldBoundsMax) && !(elementBoundsMax < worldBoundsMin)) {
						elementLength = runLength(run.x);
						elementColorsIndex = (int)(run.y & 0xFFFFu);
						found = true;
						break;
					}
				}
				ovPending = DIR > 0 ? ovNext < solidCount : ovNext >= 2;
			};
			if (!COUNT) {
				if (DIR < 0 && CVX_RARE(ovPending)) {
					scanOverflow();
				}
				{ // the next visible run of the record, as selects (no divergent region: a lane without one keeps what it has)
					const bool free = !found; // (top-down: always -- the scan of the runs beyond the first two comes after this)
					const bool take0 = ((int)free & (int)(DIR > 0 ? vis0 : (vis0 && !vis1))) != 0;
					const bool take1 = ((int)free & (int)(DIR > 0 ? (vis1 && !vis0) : vis1)) != 0;
					const bool take = ((int)take0 | (int)take1) != 0;
					const uint32_t w1 = take0 ? colorsIndex0 : colorsIndex1;
					const float spanMin = take0 ? b0 : b1, spanMax = take0 ? t0 : t1;
					const int spanLength = (int)(spanMax - spanMin) >> lod; // RLEElement.Length: integers below 2^17, the difference is exact
					if (DIR > 0) { // (nothing found yet: a lane that takes nothing here either scans on below or leaves the loop)
						elementLength = spanLength;
						elementColorsIndex = (int)w1;
						elementBoundsMin = spanMin;
						elementBoundsMax = spanMax;
					} else {
						elementLength = take ? spanLength : elementLength;
						elementColorsIndex = take ? (int)w1 : elementColorsIndex;
						elementBoundsMin = take ? spanMin : elementBoundsMin;
						elementBoundsMax = take ? spanMax : elementBoundsMax;
					}
					found = ((int)found | (int)take) != 0;
					vis0 = vis0 && !take0;
					vis1 = vis1 && !take1;
				}
			}
			while (COUNT && solidIndex < solidCount) {
				CVX_COUNT(3);
				// the walk's k-th solid run is run k of the record's top-down list, or run solidCount - 1 - k for the bottom-up walk
				const int j = DIR > 0 ? solidIndex : solidCount - 1 - solidIndex;
				uint32_t w0, w1;
				if (CVX_USUAL(!listed)) {
					// (the record's runs as the run list would hold them: span word {bottomY | (topY - 1) << 16}, colorsIndex | elementIndex << 16)
					const uint32_t top0 = rec.y - 0x10000u;
					const uint32_t p0 = (rec.w & 0xFFFFu) | (top0 & 0xFFFF0000u), p1 = (rec.z & 0xFFFFu) | (rec.w & 0xFFFF0000u), p2 = (rec.y & 0xFFFFu) | (rec.z & 0xFFFF0000u);
					const uint32_t l0 = (uint32_t)runLength(p0), l1 = (uint32_t)runLength(p1);
					const uint32_t c0 = countInfo.x & 0xFFFF0000u, c1 = l0 | (countInfo.y << 16), c2 = (l0 + l1) | (countInfo.y & 0xFFFF0000u);
					w0 = j == 0 ? p0 : (j == 1 ? p1 : p2);
					w1 = j == 0 ? c0 : (j == 1 ? c1 : c2);
				} else { // a listed column: fetched when the walk gets there
					const uint2 run = ld2(arena, columnRunsOff + (uint32_t)j * 8u);
					w0 = run.x;
					w1 = run.y;
				}
				solidIndex++;
				elementLength = runLength(w0);
				elementColorsIndex = (int)(w1 & 0xFFFFu);
				if (COUNT) { consumed = DIR > 0 ? (w1 >> 16) : (countInfo.x & 0xFFFFu) + 1u - (w1 >> 16); } // position among all elements in walk order
				// The run's world-space span.  Top-down the reference accumulates it from worldMaxY (:429-431,449-451), bottom-up from 0
				// (:433-435,453-455): integer sums either way, and the runs of a column add up to its height, so both are these.
				runSpan(w0, elementBoundsMin, elementBoundsMax);
				if (elementBoundsMin > worldBoundsMax) {
					if (DIR < 0) { solidIndex = solidCount + 1; break; } else { continue; }
				}
				if (elementBoundsMax < worldBoundsMin) {
					if (DIR > 0) { solidIndex = solidCount + 1; break; } else { continue; }
				}
				found = true;
				break;
			}
			CVX_END(3);
			if (COUNT ? !found : CVX_RARE(!found)) { // (rendering build: only a lane that has nothing visible left in the record)
				if (COUNT && solidIndex == solidCount) { consumed = (countInfo.x & 0xFFFFu) + 1u; } // walked on to the end guard
				if (COUNT) { break; }
				if (DIR > 0) { // top-down the runs beyond the first two come last (here, behind the same rare branch)
					scanOverflow();
				}
				if (!found) {
					continue; // (nothing is pending any more, so the loop condition ends it -- one way out of the loop instead of two)
				}
			}

			// unlerp(0, worldMaxY, x) = (x - 0) / (worldMaxY - 0); worldMaxY is a power of two (enforced at upload), so
			// multiplying by its exact reciprocal gives the identical correctly rounded quotient
			const float portionBottom = elementBoundsMin * invWorldMaxY;
			const float portionTop = elementBoundsMax * invWorldMaxY;
			f3 camSpaceFrontBottom = f3_lerp(camSpaceMinLast, camSpaceMaxLast, portionBottom);
			f3 camSpaceFrontTop = f3_lerp(camSpaceMinLast, camSpaceMaxLast, portionTop);

			// Which face follows the side (:549-565), decided here so that its colour (the run's first colour for a top face, its
			// last for a bottom face, :553,560) is in flight while the side is projected and drawn.  The reference reads it after
			// the side; a load has no side effect, and the counting build counts it where the reference reads it.
			const bool faceTop = portionTop < cameraPosYNormalized;
			// (`&` / `|`: four compares and three scalar ands / ors, straight-line; the short-circuit forms are compiled into three nested divergent regions)
			const bool faceBottom = ((int)!faceTop & (int)(portionBottom > cameraPosYNormalized)) != 0;
			const bool faceWanted = (((int)faceTop & (int)!(elementBoundsMax > worldBoundsMax)) | ((int)faceBottom & (int)!(elementBoundsMin < worldBoundsMin))) != 0; // (faceBottom implies !faceTop)
			// (unconditional: both addresses are colours of this run, and a branch around one load costs more than the load)
			const uint32_t secondaryColor = ld_color(arena, worldColumnColorsOff +((uint32_t)(faceTop ? elementColorsIndex : elementColorsIndex + elementLength - 1) << L.colorShift));

			// side of the run, :484-542
			CVX_COUNT(4);
			// x / z of the front end the face shares with the side (:570-571 project the same point again): kept from the side
			// block unless the face's own near clip moves the point
			float frontBottomQuotient, frontTopQuotient;
			// The face's two ends (:566-571) ahead of the side: secA = the back end (on the Next intersection), secB = the front end the face shares with the
			// side, whose x / z (:570-571 project the same point again) is kept from the side block.  Pure arithmetic, and with it ONE rare branch serves the
			// near-plane clips of both blocks.
			f3 secA = f3_lerp(camSpaceMinNext, camSpaceMaxNext, faceTop ? portionTop : portionBottom);
			float secBQuotient;
			bool faceVisible = faceWanted; // ClipHomogeneousCameraSpaceLine of the face, CameraData.cs:124-138 (a face that is not wanted runs through its block as "not visible")
			{
				float uA = (float)elementLength;
				float uB = 0.0f;
				bool visible = true; // ClipHomogeneousCameraSpaceLine with u, CameraData.cs:141-157
				// uvA = (1, uA) / bottom.z, uvB = (1, uB) / top.z (:490-493) and ProjectClippedToScreen (CameraData.cs:160) of both ends:
				// three numerators per denominator.  Ordinary case (nothing near-clipped, so uA = the run length in [1, 65535] and
				// uB = 0, and all of x, z of both ends within [2^-30, 2^30]): one refined reciprocal per end (see quot_safe).  Every lane
				// computes that; ONE rare branch then redoes the lanes that are not ordinary -- an end behind the near plane (:141-157) or an
				// operand outside the range -- with the clip and the plain divisions (an if / else costs two branches, a test per case one each).
				float uvAx, uvAy, uvBx, uvBy;
				{
					const Recip rb = recip_safe(camSpaceFrontBottom.z), rt = recip_safe(camSpaceFrontTop.z);
					uvAx = quot_safe(1.0f, rb);
					uvAy = quot_safe(uA, rb);
					frontBottomQuotient = quot_safe(camSpaceFrontBottom.x, rb);
					uvBx = quot_safe(1.0f, rt);
					uvBy = __int_as_float(__float_as_int(camSpaceFrontTop.z) & (int)0x80000000); // +0 / z: a zero with the sign of z
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
					// the face's clip (:124-138) with the front end as the side's clip left it
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
					secBQuotient = secB.x / secB.z; // (an unclipped secB: the operands of the front quotient above, the same quotient)
				}
				{ // (a side entirely behind the near plane runs through the rest with whatever it holds and fails the overlap test below: no branch of its own)
					CVX_COUNT(9);
					float boundsX = frontBottomQuotient;
					float boundsY = frontTopQuotient;
					{ // :496-499, the swap of the two ends as selects
						const bool sw = boundsX > boundsY;
						const float bx_ = sw ? boundsY : boundsX, by_ = sw ? boundsX : boundsY;
						const float ax_ = sw ? uvBx : uvAx, bxx_ = sw ? uvAx : uvBx;
						const float ay_ = sw ? uvBy : uvAy, byy_ = sw ? uvAy : uvBy;
						boundsX = bx_; boundsY = by_; uvAx = ax_; uvBx = bxx_; uvAy = ay_; uvBy = byy_;
					}
					int rbMin = f2i(rintf(boundsX));
					int rbMax = f2i(rintf(boundsY));
					if (CVX_USUAL(((int)visible & (int)(rbMax >= nextFreePixelMin) & (int)(rbMin <= nextFreePixelMax)) != 0)) {
						CVX_COUNT(10);
						reduce_pixel_horizon(seen, sshift, omin, omax, rbMin, rbMax, nextFreePixelMin, nextFreePixelMax, frustumBoundsMin, frustumBoundsMax);
						CVX_END(4);
						for (int w = rbMin >> 5; w <= (rbMax >> 5); w++) { // pixel loop :519-533 over unseen bits
							const uint32_t range = range_mask(w, rbMin, rbMax);
							const uint32_t m = seen[w << sshift];
							uint32_t todo = ~m & range;
							if (todo != 0u) {
								seen[w << sshift] = m | range;
								frustumDirMaxWorld = CVX_FLOAT_EPSILON;
								// perspective-correct colour of pixel y of the run's side, :524-531
								// the texture row of pixel y: the cheap form where it is certain, the reference's two divisions where not (tex_row_cheap above)
								const TexRun texRun = COUNT ? TexRun{} : tex_run(boundsX, boundsY, uvAx, uvBx, uvAy, uvBy);
								auto textureRow = [&](int y) -> int {
									if (COUNT) { return tex_row_exact(y, boundsX, boundsY, uvAx, uvBx, uvAy, uvBy); }
									bool certain;
									int row = tex_row_cheap(y, boundsX, uvAx, uvAy, texRun, certain);
									if (CVX_RARE(!certain)) { row = tex_row_exact(y, boundsX, boundsY, uvAx, uvBx, uvAy, uvBy); }
									return row;
								};
								auto colourOffset = [&](int y) -> uint32_t {
									int colorIdx = m_clampi(textureRow(y), 0, elementLength - 1) + elementColorsIndex;
									return worldColumnColorsOff + ((uint32_t)colorIdx << L.colorShift); // (colour k of a column lives one 128-byte line behind its colour k - 1, cvx_device.h)
								};
								// Up to four pixels per trip: all their colour loads are in flight before the first store waits for its colour (a load's
								// latency is what a trip costs, not its arithmetic).  Same pixels, same order of stores per lane.  (Two per trip until
								// round 5; four: -1.7 % at 3840 x 2160, -0.6 % at 1080p, where 35 % of the trips have a second pixel.)
								do {
									CVX_COUNT(5);
									const int y0 = (w << 5) + (__ffs((int)todo) - 1);
									todo &= todo - 1u;
									const uint32_t c0 = ld_color(arena, colourOffset(y0));
									const bool second = todo != 0u;
									// (no initial values: each of these is written and read under the same condition, and an initialiser is seven moves per trip
									// for the lanes that never look at it)
									int y1, y2, y3;
									uint32_t c1, c2, c3;
									bool third = false, fourth = false;
									if (second) {
										y1 = (w << 5) + (__ffs((int)todo) - 1);
										todo &= todo - 1u;
										c1 = ld_color(arena, colourOffset(y1));
										third = todo != 0u;
										if (third) {
											y2 = (w << 5) + (__ffs((int)todo) - 1);
											todo &= todo - 1u;
											c2 = ld_color(arena, colourOffset(y2));
											fourth = todo != 0u;
											if (fourth) {
												y3 = (w << 5) + (__ffs((int)todo) - 1);
												todo &= todo - 1u;
												c3 = ld_color(arena, colourOffset(y3));
											}
										}
									}
									st_pixel_loop(tileOut, laneByteOff, y0, c0);
									if (second) {
										st_pixel_loop(tileOut, laneByteOff, y1, c1);
										if (third) {
											st_pixel_loop(tileOut, laneByteOff, y2, c2);
											if (fourth) { st_pixel_loop(tileOut, laneByteOff, y3, c3); }
										}
									}
									if (COUNT) { const unsigned int n_ = 1u + (second ? 1u : 0u) + (second && third ? 1u : 0u) + (second && third && fourth ? 1u : 0u); cnt.C += n_; cnt.P += n_; }
								} while (todo != 0u);
							}
						}
						CVX_END(5);
						if (COUNT && nextFreePixelMin > nextFreePixelMax) {
							return false; // (the rendering build tests this once per column, below: nothing can be drawn in between)
						}
					}
				}
			}

			// top / bottom of the run, :544-610
			CVX_END(4); // (remainder of) the side block
			// (a run seen from the side, or whose face lies outside the world bounds (:551,558,564), goes through the face block as "not visible": the
			// block is straight-line up to the overlap test, and some lane of the wave needs it anyway)
			if (COUNT && faceWanted) { cnt.C++; }
			CVX_COUNT(6);
			{
				CVX_COUNT(11);
				float bx = rintf(secA.x / secA.z);
				float by = rintf(secBQuotient);
				int rbMin = f2i(bx);
				int rbMax = f2i(by);
				if (rbMin > rbMax) {
					int t = rbMin; rbMin = rbMax; rbMax = t;
				}
				if (CVX_USUAL(((int)faceVisible & (int)(rbMax >= nextFreePixelMin) & (int)(rbMin <= nextFreePixelMax)) != 0)) {
					CVX_COUNT(12);
					reduce_pixel_horizon(seen, sshift, omin, omax, rbMin, rbMax, nextFreePixelMin, nextFreePixelMax, frustumBoundsMin, frustumBoundsMax);
					CVX_END(6);
					for (int w = rbMin >> 5; w <= (rbMax >> 5); w++) { // :595-603
						const uint32_t range = range_mask(w, rbMin, rbMax);
						const uint32_t m = seen[w << sshift];
						uint32_t todo = ~m & range;
						if (todo != 0u) {
							seen[w << sshift] = m | range;
							frustumDirMaxWorld = CVX_FLOAT_EPSILON;
							do {
								CVX_COUNT(7);
								const int y = (w << 5) + (__ffs((int)todo) - 1);
								todo &= todo - 1u;
								st_pixel_loop(tileOut, laneByteOff, y, secondaryColor);
								if (COUNT) { cnt.P++; }
							} while (todo != 0u);
						}
					}
					CVX_END(7);
					if (COUNT && nextFreePixelMin > nextFreePixelMax) {
						return false;
					}
				}
			}
		}

		// :537,606: the ray ends once every pixel of its window has been written.  The reference (and the counting build, whose
		// element count depends on it) leaves right after the pixel loop that closed the window; the rendering build looks once
		// per column: with the window closed nothing more can be drawn (every pixel of [origMin, origMax] is marked seen), so the
		// rest of the column changes no pixel.  It touches no mask word outside the window's either: a run that still passes the
		// overlap test is clamped to rbMin = nextFreePixelMin > rbMax = nextFreePixelMax, and its word loop runs only if both lie
		// in the same word -- which then holds pixels of the window; the horizon scans never start beyond [origMin, origMax].
		return COUNT || (!windowClosed && nextFreePixelMin <= nextFreePixelMax);
	};

	// column 0: LOD check (:237-243), bounds test and fetch (World.GetVoxelColumn, World.cs:130-142)
	if (ray.distLast >= lodMax) {
		dda_next_lod(ray, voxelScale, dirXNonNegative, dirZNonNegative);
		lod++;
		voxelScale *= 2;
		L = world->level[lod];
		{ const float next_ = F.lod[min(lod, 5)]; lodMax = lod < 5 ? next_ : __builtin_inff(); }
	}
	if ((ray.px & maskX) != ray.px || (ray.pz & maskZ) != ray.pz) {
		return; // out of world bounds -> WriteSkybox
	}
	ColumnCursor cur;
	cursor_set(cur, ray, L, maskX, maskZ);
	const int outsideBits = ~((maskX << 16) | maskZ); // bits of a cursor position that are set only outside the world
	rec = ld4(arena, cur.rec);

	// ONE way out of the column loop (`alive`): every early `return` out of a divergent loop costs the structuriser a flag that is
	// merged at every join on the way out.
	bool alive = true; // false: the ray is finished
	bool go = true;    // false: out of the column loop -- finished, or at its stop distance (far clip / LOD boundary)
	// The world's edge is the third thing the stop distance stands for: the DDA leaves the world with the crossing number n of an axis, n = cells
	// between the ray and the edge, at distance tMax + (n - 1) * tDelta up to the rounding of its additions; three crossings short of that the ray
	// is certainly still inside, so the column loop needs no position test ((p & dimensionMask) != p, :613 / World.cs:130-142) below that distance.
	// Beyond it the loop is left after EVERY column and the exact test is made below the loop (the last few columns of a ray).
	auto edgeDistance = [&]() -> float {
		const int px = cur.pos >> 16, pz = cur.pos & 0xFFFF;
		const int sx = cur.posStepX, sz = cur.posStepZ;
		const int nx = sx > 0 ? (dimX - px) >> lod : (px >> lod) + 1, nz = sz > 0 ? (dimZ - pz) >> lod : (pz >> lod) + 1;
		const float ex = sx != 0 ? ray.tMaxX + (float)(nx - 4) * ray.tDeltaX : __builtin_inff();
		const float ez = sz != 0 ? ray.tMaxZ + (float)(nz - 4) * ray.tDeltaZ : __builtin_inff();
		return m_min(ex, ez);
	};
	bool nearEdge = false; // beyond edgeDistance(): every column comes up for the exact test
	// ... and a checkpoint: the three crossings of slack cover the rounding of the DDA's own additions (tMax += tDelta, n of them drift by <= n^2 2^-24
	// tDelta) only for a few thousand crossings, and the library accepts worlds of up to 32768 columns a side with LOD distances of the caller's choice.
	// So the untested stretch is at most 1024 crossings of the faster axis long (drift <= 0.07 crossings); a ray that reaches the checkpoint comes up
	// below the loop, takes a fresh edge distance from where it stands and goes on -- once per ~1000 columns.
	auto stopDistance = [&]() -> float {
		const float checkpoint = ray.distLast + 1024.0f * hw_min(ray.tDeltaX, ray.tDeltaZ);
		return m_min(m_min(m_min(farClip, lodMax), edgeDistance()), checkpoint);
	};
	float stopDist = stopDistance();
	// one column step: `rec` = the record of the column to process (already fetched), `nextRec` receives the look-ahead
	auto columnStep = [&](const uint4 &rec, uint4 &nextRec, auto guardTag) {
		constexpr bool GUARD = decltype(guardTag)::value;
		CVX_BEGIN();
		CVX_WAITPROBE(9);
		CVX_COUNT(1);
		// ---- look ahead: move the DDA to the next column (Step :613 / :252 / :273, then the LOD check of the
		// next iteration :237-243) and start fetching its record; nothing below touches `ray` again.
		curDistLast = ray.distLast;
		curDistNext = ray.distNext;
		// `stopDist` = min(far clip, LOD distance of this level): ONE compare per step serves both (:613 / :273 and :237-243).  A ray that reaches it
		// leaves the column loop after this column; below the loop it either ends (far clip) or changes level and comes back.  (The record fetched here
		// for the next column is then the old level's: fetched again after the switch -- twice per ray.)
		const uint32_t recAddr = cur.rec; // (where `rec` came from: the counting build finds the column's entry of the counts table with it)
		const bool stopReached = dda_step_cursor(ray, cur, stopDist);
		// (the fetch is done for every lane, also one that stops after this column: its state is dead, and an unconditional load from
		// inside the arena is cheaper than branching around it)
		nextRec = ld4(arena, cur.rec);

		// ---- the current column, exactly as the reference processes it
		if (COUNT) {
			cnt.S++;
#pragma unroll
			for (int k = 0; k < 6; k++) { cnt.lod[k] += (lod == k) ? 1u : 0u; }
		}
		{
			bool draw = true;
			worldBoundsMin = 0.0f;
			worldBoundsMax = worldMaxY;
			{ // :251-256 (RunCount > 0: not an empty column) and :261-281, straight-line: the tests as flags (`&` / `|`: no nested divergent regions), the bounds as selects
				const bool nonEmpty = rec.x != 0u; // RunCount > 0 (the empty column's record is all zeros, cvx_device.h)
				const bool cull = ((int)nonEmpty & (int)(frustumDirMaxWorld != CVX_FLOAT_EPSILON)) != 0;
				const float columnWorldMin = (float)(rec.y & 0xFFFFu);
				const float columnWorldMax = (float)(rec.y >> 16);
				// :264-269: the upper bound at the farther distance when the direction rises, at the nearer one otherwise (the lower bound likewise) -- written
				// as the larger / smaller of the two products (curDistNext >= curDistLast >= 0 and rounding is monotone, so the product the reference
				// selects IS the larger / smaller one): a max / min instead of a compare + select, same value
				const float newMax = posY + hw_max(frustumDirMaxWorld * curDistNext, frustumDirMaxWorld * curDistLast);
				const float newMin = posY + hw_min(frustumDirMinWorld * curDistNext, frustumDirMinWorld * curDistLast);
				const bool leftWorld = ((int)cull & ((int)(newMin > worldBoundsMax) | (int)(newMax < worldBoundsMin))) != 0; // frustum left the world entirely
				const bool noOverlap = ((int)cull & ((int)(columnWorldMin > newMax) | (int)(columnWorldMax < newMin))) != 0; // this column does not overlap the writable world bounds
				const bool narrowed = ((int)cull & (int)!leftWorld & (int)!noOverlap) != 0;
				alive = !leftWorld;
				draw = ((int)nonEmpty & (int)!leftWorld & (int)!noOverlap) != 0;
				worldBoundsMin = narrowed ? newMin : worldBoundsMin;
				worldBoundsMax = narrowed ? newMax : worldBoundsMax;
			}
			CVX_END(1);
			if (draw) {
				alive = drawColumn(rec, recAddr);
				if (COUNT) { cnt.E += consumed; }
				// Every vector-memory operation of the drawn column is complete from here on.  Without this the compiler does not know what the pixel loops left in
				// flight (a colour load whose pixel loop never ran, stores) when the paths of a drawn and a skipped column join below, and drains the queue
				// there with s_waitcnt vmcnt(0) -- in EVERY step, also a skipped column's, whose only pending loads are the look-ahead record issued ~30
				// instructions earlier: the one-step look-ahead was being waited for in the step that issued it (rounds 1-3).  With the drain on the drawn
				// path only (where a colour load, being younger, has waited for the look-ahead anyway), the join knows that nothing but the look-ahead can
				// be pending and the wait for it moves to its first use, the top of the next step.
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "the s_waitcnt immediate below is the gfx9 encoding (vmcnt in bits 3:0 + 15:14, expcnt 6:4, lgkmcnt 11:8); this library is written for gfx950"
#endif
				__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
			}
			CVX_BEGIN();
		}

		// ---- next column (far clip reached / left the world: WriteSkybox; the step guard never binds on valid input)
		if (GUARD) { guardSteps--; } // (counted in one of the two copies of the step: the cap is generous by more than a factor of two)
		alive = alive && (!GUARD || guardSteps > 0); // false: the ray is finished (WriteSkybox)
		go = alive && !stopReached;                                  // false: out of the column loop (finished, far clip, or a LOD boundary)
		CVX_END(1);
	};
	// The loop alternates between two register sets for the record pair, so the look-ahead record never has to be copied into "the
	// current one" at the end of a step (8 v_mov per step); the code of a step exists twice for it.
	for (;;) {
		while (go) {
			columnStep(rec, recB, std::true_type{});
			if (go) {
				columnStep(recB, rec, std::false_type{});
			}
		}
		// NextLOD (:237-243) for the column the ray stands on, if that is why it left the loop: alive, at or beyond this level's LOD distance, not
		// beyond far clip (the last level's distance is +infinity: lod < 5 here)
		if (!alive || ray.distLast >= farClip || (cur.pos & outsideBits) != 0) {
			break; // finished: window closed / frustum left the world / step guard, far clip (:273), left the world (:613)
		}
		if (!(ray.distLast >= lodMax)) {
			// neither: a checkpoint, or the ray is within a few columns of the world's edge.  A fresh stop distance beyond where the ray stands means the
			// edge is still more than three crossings away: on with the loop.  Else it comes up after every column from here on (stopDist below every
			// distance).  Either way the record of the column it stands on was fetched into the other register pair: fetched again (a handful of times per ray)
			const float fresh = nearEdge ? -__builtin_inff() : stopDistance();
			nearEdge = !(fresh > ray.distLast);
			stopDist = nearEdge ? -__builtin_inff() : fresh;
			rec = ld4(arena, cur.rec);
			go = true;
			continue;
		}
		// back to the DDA's own form of the position (exact: the ray is inside the world)
		ray.px = cur.pos >> 16;
		ray.pz = cur.pos & 0xFFFF;
		ray.sx = cur.posStepX >> 16;
		ray.sz = cur.posStepZ;
		dda_next_lod(ray, voxelScale, dirXNonNegative, dirZNonNegative);
		lod++;
		voxelScale *= 2;
		L = world->level[lod];
		{ const float next_ = F.lod[min(lod, 5)]; lodMax = lod < 5 ? next_ : __builtin_inff(); }
		cursor_set(cur, ray, L, maskX, maskZ);
		stopDist = nearEdge ? -__builtin_inff() : stopDistance();
		rec = ld4(arena, cur.rec);
		go = true;
	}
}

#ifndef CVX_DEVICE_FUNCTIONS_ONLY /* cvx_lone.hip includes this header for its device functions: the kernels live in ONE translation unit (cvx_gpu.hip) */
// ---------------------------------------------------------------------------
// render kernel: grid = tiles, block = 64 (one wave).  LDS: words*64 uint32.
// ---------------------------------------------------------------------------
template <bool COUNT>
#ifndef CVX_WAVES_PER_SIMD
#define CVX_WAVES_PER_SIMD 4
#endif
__global__ __launch_bounds__(CVX_WAVE, CVX_WAVES_PER_SIMD) void render_kernel(const DevFrame *__restrict__ frames, const DevTile *__restrict__ tiles,
                                                          const DevWorld *__restrict__ world, DevCounters *__restrict__ counters)
{
	extern __shared__ uint32_t lds[];
	const int lane = threadIdx.x;
#ifdef CVX_TILE_TIMES
	const unsigned long long tileStart_ = __builtin_amdgcn_s_memtime();
#endif
	const DevTile tile = tiles[blockIdx.x];
	const DevFrame &F = frames[tile.frame];
	const DevSegment &S = F.seg[tile.seg];

	// The mask only covers the words that hold pixels [omin, omax]; `seen` is biased so that the
	// absolute word index w of a pixel addresses seen[w * 64].
	const int omin = S.omin, omax = S.omax;
	const int wordBase = omin >> 5;
	const int words = (omax >> 5) - wordBase + 1;
	// RaySetupJob (:19-39): tile -> (segment, planeRayIndex)
	const int firstLane = tile.lanes & 0xFF, laneCount = tile.lanes ? (tile.lanes >> 8) & 0xFF : CVX_WAVE;
	const int sshift = 31 - __clz(laneCount); // laneCount is a power of two
	// 2^dupShift physical lanes per ray of a narrow sub-tile (cvx_gpu.hip DrawBatch: a wave with <= 8 active lanes issues ~3.6 x slower); the lanes of a
	// group hold the same values all the way, read and write the same mask words and store the same pixels
	const int dupShift = (tile.lanes >> 16) & 7;
	const int vlane = lane >> dupShift;
	const bool leader = (lane & ((1 << dupShift) - 1)) == 0; // one lane per ray writes the skybox pixels below (64 stores to one address are not free)
	const int planeRayIndex = tile.tileInSeg * CVX_WAVE + firstLane + vlane;
	const bool active = vlane < laneCount && planeRayIndex < S.rayCount;
	if (vlane < laneCount) {
		for (int w = 0; w < words; w++) {
			lds[(w << sshift) + vlane] = 0u; // stackalloc is zero-initialised, :208
		}
	}
	const gptr_tile tileOut = (gptr_tile)tile.out;
	const uint32_t laneByteOff = (uint32_t)(firstLane + vlane) * 4u;
	uint32_t *seen = lds + vlane - (wordBase << sshift);
	ProfLane prof;
#ifdef CVX_PROFILE_SECTIONS
	for (int i = 0; i < CVX_NSEC; i++) { prof.acc[i] = 0u; }
#ifdef CVX_PROFILE_COUNTS
	for (int i = 0; i < CVX_NSEC; i++) { prof.lanes[i] = 0u; }
#endif
	CVX_BEGIN();
#endif

	LaneCounters cnt;
	if (COUNT) {
		cnt.S = cnt.E = cnt.C = cnt.P = 0;
		for (int i = 0; i < 6; i++) { cnt.lod[i] = 0; }
	}

	if (active) {
		// RenderJob.Execute :174-178: the iteration direction is a per-frame (wave-uniform) constant
		if (F.inverse) {
			trace_ray<-1, COUNT>(F, S, world, planeRayIndex, seen, sshift, tileOut, laneByteOff, cnt, prof);
		} else {
			trace_ray<1, COUNT>(F, S, world, planeRayIndex, seen, sshift, tileOut, laneByteOff, cnt, prof);
		}
	}

	// WriteSkybox / WriteSkyboxFull (:699-716) for the whole wave: every pixel
	// of [omin, omax] not marked seen gets the skybox colour.
	CVX_BEGIN();
	unsigned int skyPixels = 0;
	for (int w = omin >> 5; w <= (omax >> 5); w++) {
		uint32_t todo = 0u;
		if (active && leader) { todo = ~seen[w << sshift] & range_mask(w, omin, omax); }
		const int base = w << 5;
		if (!COUNT && __ballot(todo != 0u) == 0ull) { continue; } // (a word every ray of the tile has filled -- the ground half of a frame: 3 instructions instead of 32 bit tests; round 5: -0.5 %)
#pragma unroll 4
		for (int b = 0; b < 32; b++) {
			if ((todo >> b) & 1u) {
				st_pixel_stream(tileOut, laneByteOff, base + b, CVX_SKYBOX_ARGB);
			}
		}
		if (COUNT) { skyPixels += (unsigned int)__popc(todo); }
	}

#ifdef CVX_PROFILE_SECTIONS
	CVX_END(8);
	for (int i = 0; i < CVX_NSEC; i++) {
#ifdef CVX_PROFILE_COUNTS
		unsigned long long tot = prof.acc[i], act = prof.lanes[i];
		for (int o = 32; o > 0; o >>= 1) { tot += (unsigned long long)__shfl_xor((long long)tot, o); act += (unsigned long long)__shfl_xor((long long)act, o); }
		if (lane == 0) { atomicAdd(&g_sectionCycles[i], tot); atomicAdd(&g_sectionCycles[16 + i], act); }
#else
		unsigned int mx = prof.acc[i], sum = prof.acc[i] >> 6;
		for (int o = 32; o > 0; o >>= 1) {
			mx = max(mx, (unsigned int)__shfl_xor((int)mx, o));
			sum += (unsigned int)__shfl_xor((int)sum, o);
		}
		if (lane == 0) {
			atomicAdd(&g_sectionCycles[i], (unsigned long long)mx);
			atomicAdd(&g_sectionCycles[16 + i], (unsigned long long)sum);
		}
#endif
	}
#endif
#ifdef CVX_TILE_TIMES
	if (!COUNT && lane == 0 && g_tileTimes) { g_tileTimes[blockIdx.x] = __builtin_amdgcn_s_memtime() - tileStart_; }
#endif
	if (COUNT) {
		cnt.P += skyPixels;
		atomicAdd(&counters->S, (unsigned long long)cnt.S);
		atomicAdd(&counters->E, (unsigned long long)cnt.E);
		atomicAdd(&counters->C, (unsigned long long)cnt.C);
		atomicAdd(&counters->P, (unsigned long long)cnt.P);
		atomicAdd(&counters->R, active ? 1ull : 0ull);
		for (int i = 0; i < 6; i++) {
			atomicAdd(&counters->lodVisits[i], (unsigned long long)cnt.lod[i]);
		}
	}
}

} // namespace cvxk
namespace cvxk {

// ---------------------------------------------------------------------------
// untile: tile-major pool -> the reference's ray-major rows (RayBuffer.cs:121-128)
// rows [firstRay, firstRay+rayCount) of buffer `which`; count0 = RayCount of the
// pool's first segment (segment 0 or 2), tileBase1 = first tile of the second.
// ---------------------------------------------------------------------------
__global__ void untile_kernel(const uint32_t *__restrict__ pool, uint32_t *__restrict__ dst, int firstRay, int rayCount, int width,
                              int count0, int tileBase1, int tileCapacity)
{
	const int y = blockIdx.x * blockDim.x + threadIdx.x;
	const int row = blockIdx.y;
	if (y >= width || row >= rayCount) {
		return;
	}
	const int r = firstRay + row;
	int plane, tileBase;
	if (r < count0) {
		plane = r;
		tileBase = 0;
	} else {
		plane = r - count0;
		tileBase = tileBase1;
	}
	const int tile = tileBase + (plane >> 6);
	uint32_t v = 0u;
	if (tile < tileCapacity) {
		v = pool[((size_t)tile * (size_t)width + (size_t)y) * CVX_WAVE + (plane & 63)];
	}
	dst[(size_t)row * (size_t)width + (size_t)y] = v;
}

// ---------------------------------------------------------------------------
// copy_rows: pack / unpack pixel rows of tiles (one row = 64 pixels = 256 bytes) between the tile pools and a
// contiguous staging buffer -- the payload of the multi-GPU tile exchange is only the rows [origMin, origMax]
// a segment can write.  One workgroup per span, 16 bytes per thread.
// ---------------------------------------------------------------------------
struct RowSpan { // == cvx_row_span
	long long poolRow;   // first row inside the pool (in 256-byte rows)
	long long packedRow; // first row inside the staging buffer
	int rows;
	int kind;            // 0 top-down pool, 1 left-right pool
};

__global__ void copy_rows_kernel(uint4 *__restrict__ poolTD, uint4 *__restrict__ poolLR, uint4 *__restrict__ packed,
                                 const RowSpan *__restrict__ spans, int toPacked)
{
	const RowSpan sp = spans[blockIdx.x];
	uint4 *pool = (sp.kind == 0 ? poolTD : poolLR) + sp.poolRow * 16;
	uint4 *stage = packed + sp.packedRow * 16;
	const long long n = (long long)sp.rows * 16; // uint4 per span
	for (long long i = threadIdx.x; i < n; i += blockDim.x) {
		if (toPacked) {
			stage[i] = pool[i];
		} else {
			pool[i] = stage[i];
		}
	}
}

// ---------------------------------------------------------------------------
// Phase 2: RenderManager.BlitSegments (RenderManager.cs:199-256) +
// RayBufferBlit.shader frag (:48-64), evaluated at pixel centres.  Rule (ours,
// Unity's rasteriser is not restated; the test-side numpy rule (blit_reference in tests/) is the same
// arithmetic in numpy, float32 operation by operation): for segment s with
// triangle (VP = a, MaxScreen = b, MinScreen = q) the host computes ONCE per frame
//     den = (b.y - q.y) * (a.x - q.x) + (q.x - b.x) * (a.y - q.y),  inv = 1 / den
//     A0 = (b.y - q.y) * inv, B0 = (q.x - b.x) * inv      (weight of VP)
//     A1 = (q.y - a.y) * inv, B1 = (a.x - q.x) * inv      (weight of MaxScreen)
// and a pixel centre c evaluates two edge functions per segment, no division:
//     wVp = A0 * (c.x - q.x) + B0 * (c.y - q.y),  wMax = A1 * (c.x - q.x) + B1 * (c.y - q.y),  wMin = 1 - wVp - wMax
// first segment whose weights are all >= 0 wins; x = wMax / (wMax + wMin) (uv.x / (uv.x + uv.y), RayBufferBlit.shader:56);
// ray = clamp(floor(x * RayCount), 0, RayCount - 1).
// ---------------------------------------------------------------------------
struct BlitParams {
	float qx[4], qy[4];                 // MinScreen
	float a0[4], b0[4], a1[4], b1[4];   // edge functions, see above
	int rayCount[4];
	int tileBase[4];
	int width, height;
	uint32_t clearColor;
};

// Barycentric weights of pixel centre (px, py) in the triangle of segment s (see above)
__device__ __forceinline__ void blit_weights(const BlitParams &p, int s, int px, int py, float &wVp, float &wMax, float &wMin)
{
	const float dx = ((float)px + 0.5f) - p.qx[s], dy = ((float)py + 0.5f) - p.qy[s];
	wVp = p.a0[s] * dx + p.b0[s] * dy;
	wMax = p.a1[s] * dx + p.b1[s] * dy;
	wMin = 1.0f - wVp - wMax;
}
__device__ __forceinline__ int blit_ray(int rc, float wMax, float wMin)
{
	const float x = wMax / (wMax + wMin);
	float rf = floorf(x * (float)rc);
	rf = fminf(fmaxf(rf, 0.0f), (float)(rc - 1));
	return (rf == rf) ? (int)rf : 0;
}
// Which segment owns pixel (px, py) and which ray of it: segment 0..3 (-1: none) and the ray index.
__device__ __forceinline__ int blit_classify(const BlitParams &p, int px, int py, int &ray)
{
	ray = 0;
#pragma unroll // (the four segments' parameters are then fetched once, ahead of the tests, instead of per iteration)
	for (int s = 0; s < 4; s++) {
		const int rc = p.rayCount[s];
		if (rc <= 0) {
			continue;
		}
		float wVp, wMax, wMin;
		blit_weights(p, s, px, py, wVp, wMax, wMin);
		if (wVp >= 0.0f && wMax >= 0.0f && wMin >= 0.0f) {
			ray = blit_ray(rc, wMax, wMin);
			return s;
		}
	}
	return -1;
}
__device__ __forceinline__ uint32_t blit_fetch_td(const uint32_t *__restrict__ poolTD, const BlitParams &p, int s, int ray, int py)
{
	return poolTD[((uint32_t)(p.tileBase[s] + (ray >> 6)) * (uint32_t)p.height + (uint32_t)py) * CVX_WAVE + (uint32_t)(ray & 63)]; // (a pool is < 2^32 pixels: cvx_set_resolution)
}
__device__ __forceinline__ uint32_t blit_fetch_lr(const uint32_t *__restrict__ poolLR, const BlitParams &p, int s, int ray, int px)
{
	return poolLR[((uint32_t)(p.tileBase[s] + (ray >> 6)) * (uint32_t)p.width + (uint32_t)px) * CVX_WAVE + (uint32_t)(ray & 63)];
}

// One workgroup (256 threads) per 64 x 64 block of the screen.
// * Pixels of the top / bottom segments read raybuffer pixel `py` of their ray: lanes along x touch neighbouring rays of one 256-byte
//   tile row.  Pixels of the left / right segments read raybuffer pixel `px` of their ray, and there the neighbouring rays belong to
//   pixels ABOVE each other: read with lanes along x every lane would hit a row of its own (a 64-byte sector per 4 bytes used).  So
//   those pixels are gathered with lanes along y into an LDS tile first (row stride 65 words: conflict-free both ways), and the
//   x-major pass that stores the screen rows takes them from there.
// * Most blocks lie inside ONE segment (the four triangles only meet along the diagonals through the vanishing point), and then the
//   per-pixel search over the segments is not needed.  A block is taken as owned by segment s when, at its four corner pixels, all three
//   weights of s are >= 1e-3 and for every earlier segment one and the same weight is <= -1e-3 (or it has no rays): the weights are
//   linear in the pixel position up to rounding (|products| < 1000 is checked, so a computed weight is within ~1e-4 of the exact linear
//   form), hence inside the block all weights of s stay > 0 and that weight of each earlier segment stays < 0 -- the search would pick
//   s for every pixel.  Any other block searches per pixel.  Same pixels either way.
#define CVX_BLIT_TILE 64
// TH = rows per workgroup: 64 for a batch of frames (a 64 x 64 block: longest contiguous raybuffer spans either way), 16 for a single
// frame (four times the workgroups: one 1080p frame alone is only 510 blocks of 64 x 64 on a chip with 256 CUs)
template <int TH>
__device__ __forceinline__ void blit_block(const uint32_t *__restrict__ poolTD, const uint32_t *__restrict__ poolLR, uint32_t *__restrict__ screen, const BlitParams &p)
{
	__shared__ uint32_t tile[TH * (CVX_BLIT_TILE + 1)];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6; // 4 waves
	const int x0 = blockIdx.x * CVX_BLIT_TILE, y0 = blockIdx.y * TH;
	const int x1 = min(x0 + CVX_BLIT_TILE, p.width) - 1, y1 = min(y0 + TH, p.height) - 1;

	// ---- owner of the whole block, if provable: lane = corner + 4 * segment (every wave computes the same)
	int owner = -1;
	{
		const int c = lane & 3, sg = (lane >> 2) & 3;
		float wVp, wMax, wMin;
		blit_weights(p, sg, (c & 1) ? x1 : x0, (c & 2) ? y1 : y0, wVp, wMax, wMin);
		const float dxm = fmaxf(fabsf((float)x0 - p.qx[sg]), fabsf((float)x1 + 1.0f - p.qx[sg])), dym = fmaxf(fabsf((float)y0 - p.qy[sg]), fabsf((float)y1 + 1.0f - p.qy[sg]));
		const bool bounded = fmaxf(fmaxf(fabsf(p.a0[sg]), fabsf(p.a1[sg])) * dxm, fmaxf(fabsf(p.b0[sg]), fabsf(p.b1[sg])) * dym) < 1000.0f; // false for NaN / inf
		const bool use = lane < 16 && p.rayCount[sg] > 0;
		const unsigned long long in = __ballot(use && bounded && wVp >= 1e-3f && wMax >= 1e-3f && wMin >= 1e-3f);
		const unsigned long long outA = __ballot(lane < 16 && (!use || (bounded && wVp <= -1e-3f)));
		const unsigned long long outB = __ballot(lane < 16 && (!use || (bounded && wMax <= -1e-3f)));
		const unsigned long long outC = __ballot(lane < 16 && (!use || (bounded && wMin <= -1e-3f)));
		bool earlierOut = true;
#pragma unroll
		for (int s = 0; s < 4; s++) {
			const unsigned int m = 15u << (4 * s);
			if (owner < 0 && earlierOut && ((unsigned int)in & m) == m) { owner = s; }
			earlierOut = earlierOut && ((((unsigned int)outA & m) == m) || (((unsigned int)outB & m) == m) || (((unsigned int)outC & m) == m));
		}
	}
	const uint32_t W = (uint32_t)p.width;

	if (owner >= 0 && owner < 2) { // top / bottom: straight from the raybuffer, x-major
		const int px = x0 + lane, rc = p.rayCount[owner];
		if (px <= x1) {
#pragma unroll 4 // (four independent fetches in flight per lane)
			for (int py = y0 + wave; py <= y1; py += 4) {
				float wVp, wMax, wMin;
				blit_weights(p, owner, px, py, wVp, wMax, wMin);
				screen[(uint32_t)py * W + (uint32_t)px] = blit_fetch_td(poolTD, p, owner, blit_ray(rc, wMax, wMin), py);
			}
		}
		return;
	}
	// y-major gather of the left / right pixels: lane = (row inside the block, one of 64 / TH columns); a wave takes every fourth column group
	{
		const int py = y0 + (lane & (TH - 1));
		if (py <= y1) {
#pragma unroll 4
			for (int c = wave * (64 / TH) + lane / TH; x0 + c <= x1; c += 4 * (64 / TH)) {
				const int px = x0 + c;
				int ray, s = owner;
				if (owner >= 0) {
					float wVp, wMax, wMin;
					blit_weights(p, owner, px, py, wVp, wMax, wMin);
					ray = blit_ray(p.rayCount[owner], wMax, wMin);
				} else {
					s = blit_classify(p, px, py, ray);
				}
				if (s >= 2) {
					tile[(lane & (TH - 1)) * (CVX_BLIT_TILE + 1) + c] = blit_fetch_lr(poolLR, p, s, ray, px);
				}
			}
		}
	}
	__syncthreads();
	// x-major: wave w takes rows w, w + 4, ...; lane = column inside the block
	{
		const int px = x0 + lane;
		if (px <= x1) {
#pragma unroll 4
			for (int r = wave; y0 + r <= y1; r += 4) {
				const int py = y0 + r;
				uint32_t color = p.clearColor;
				if (owner >= 2) {
					color = tile[r * (CVX_BLIT_TILE + 1) + lane];
				} else {
					int ray;
					const int s = blit_classify(p, px, py, ray);
					if (s >= 2) {
						color = tile[r * (CVX_BLIT_TILE + 1) + lane];
					} else if (s >= 0) {
						color = blit_fetch_td(poolTD, p, s, ray, py);
					}
				}
				screen[(uint32_t)py * W + (uint32_t)px] = color;
			}
		}
	}
}

#define CVX_BLIT_ROWS_SINGLE 16
__global__ __launch_bounds__(256) void blit_kernel(const uint32_t *__restrict__ poolTD, const uint32_t *__restrict__ poolLR, uint32_t *__restrict__ screen, BlitParams p)
{
	blit_block<CVX_BLIT_ROWS_SINGLE>(poolTD, poolLR, screen, p);
}

// Phase 2 of a whole batch in one launch (blockIdx.z = frame): buffer firstBuffer + f -> image f of `screens`.  All raybuffers of a
// kind are one allocation, strideTD / strideLR = uint32 words per buffer.
__global__ __launch_bounds__(256) void blit_batch_kernel(const uint32_t *__restrict__ poolBaseTD, const uint32_t *__restrict__ poolBaseLR, size_t strideTD, size_t strideLR,
                                                         uint32_t *__restrict__ screens, const BlitParams *__restrict__ params, int firstBuffer)
{
	const int f = blockIdx.z;
	const BlitParams &p = params[f];
	const size_t b = (size_t)(firstBuffer + f);
	blit_block<CVX_BLIT_TILE>(poolBaseTD + b * strideTD, poolBaseLR + b * strideLR, screens + (size_t)f * (size_t)p.height * (size_t)p.width, p);
}

// ---------------------------------------------------------------------------
// Multi-GPU image gather (the alternative to the raybuffer gather): every rank blits only the pixels whose ray lies
// in a 64-ray tile it rendered itself, and the display rank of a frame receives W x H pixels IN TOTAL instead of the raybuffer
// rows of the other ranks' tiles (~2x the screen).  A pixel's owner is a pure function of the frame's segments: canonical tile
// c = tileStart[segment] + ray / 64 of the pixel (blit_classify), owner = c % N.  Sender and receiver walk the image in the same
// order (row by row, left to right), so rank r's pixels of a frame form one packed stream whose positions both sides compute
// from per-row counts: image_count_kernel (counts per row and rank), image_scan_kernel (row offsets), image_gather_kernel (pack on
// the rendering rank, unpack on the display rank).  One wave per image row.
// ---------------------------------------------------------------------------
struct ImageFrame {
	BlitParams p;       // tileBase[] is unused here
	int tileStart[4];   // canonical index (inside the frame) of the first tile of each segment
	int root;           // display rank of the frame
	int localSlotBase;  // first slot of this frame in my compact tile store (slot of canonical tile c that I own: localSlotBase + c / N)
	int imageSlot;      // index of the frame among the frames its root displays
	int pad;
	long long sendBase;     // first pixel of (me -> root) for this frame inside my send stream
	long long recvBase[8];  // root == me: first pixel of (rank r -> me) for this frame inside my receive stream
};

__device__ __forceinline__ int image_owner(const ImageFrame &F, int N, int px, int py, int &tileCanonical, int &ray, int &seg)
{
	seg = blit_classify(F.p, px, py, ray);
	tileCanonical = seg >= 0 ? F.tileStart[seg] + (ray >> 6) : 0;
	return seg >= 0 ? tileCanonical % N : -1;
}

// rowCount[(frame * H + row) * 8 + r] = pixels of the row owned by rank r
__global__ __launch_bounds__(64) void image_count_kernel(const ImageFrame *__restrict__ frames, int N, int *__restrict__ rowCount)
{
	const ImageFrame &F = frames[blockIdx.y];
	const int row = blockIdx.x, lane = threadIdx.x;
	int mine = 0; // lane r: count of rank r
	for (int x0 = 0; x0 < F.p.width; x0 += 64) {
		const int px = x0 + lane;
		int c, ray, seg;
		const int owner = px < F.p.width ? image_owner(F, N, px, row, c, ray, seg) : -1;
		for (int r = 0; r < N; r++) {
			const int n = __popcll(__ballot(owner == r));
			if (lane == r) { mine += n; }
		}
	}
	if (lane < 8) {
		rowCount[((size_t)blockIdx.y * (size_t)F.p.height + (size_t)row) * 8 + lane] = lane < N ? mine : 0;
	}
}

// exclusive scan over the rows of every (frame, rank): rowCount -> row offsets, totals[frame * 8 + r] = pixels of the frame owned by r
__global__ void image_scan_kernel(int frameCount, int H, int *__restrict__ rowCount, long long *__restrict__ totals)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= frameCount * 8) {
		return;
	}
	const int f = i >> 3, r = i & 7;
	int running = 0;
	for (int row = 0; row < H; row++) {
		int *cell = rowCount + ((size_t)f * (size_t)H + (size_t)row) * 8 + r;
		const int n = *cell;
		*cell = running;
		running += n;
	}
	totals[i] = running;
}

// unpack == 0 (every rank, every frame): my pixels -> the image (frames I display) or my send stream; pixels of no segment belong to the root
// unpack == 1 (frames I display): the other ranks' pixels, from my receive stream -> the image
__global__ __launch_bounds__(64) void image_gather_kernel(const ImageFrame *__restrict__ frames, int N, int me, int unpack, const int *__restrict__ rowBase,
                                                         const uint32_t *__restrict__ localStore, size_t tileStrideWords, uint32_t *__restrict__ sendStream,
                                                         const uint32_t *__restrict__ recvStream, uint32_t *__restrict__ images)
{
	const ImageFrame &F = frames[blockIdx.y];
	if (unpack && F.root != me) {
		return;
	}
	const int row = blockIdx.x, lane = threadIdx.x, W = F.p.width, H = F.p.height;
	uint32_t *image = images + (size_t)F.imageSlot * (size_t)W * (size_t)H;
	const int *base = rowBase + ((size_t)blockIdx.y * (size_t)H + (size_t)row) * 8;
	int running = lane < 8 ? base[lane] : 0; // lane r: position of rank r's next pixel of this frame
	for (int x0 = 0; x0 < W; x0 += 64) {
		const int px = x0 + lane;
		int c, ray, seg;
		const int owner = px < W ? image_owner(F, N, px, row, c, ray, seg) : -2;
		int position = 0;
		for (int r = 0; r < N; r++) {
			const unsigned long long b = __ballot(owner == r);
			const int at = __shfl(running, r);
			if (owner == r) { position = at + __popcll(b & ((1ull << lane) - 1ull)); }
			if (lane == r) { running += __popcll(b); }
		}
		if (px >= W) {
			continue;
		}
		if (!unpack) {
			if (owner == me) {
				const int colLen = seg < 2 ? H : W, y = seg < 2 ? row : px;
				(void)colLen;
				const uint32_t v = localStore[(size_t)(F.localSlotBase + c / N) * tileStrideWords + (size_t)y * CVX_WAVE + (size_t)(ray & 63)];
				if (F.root == me) {
					image[(size_t)row * (size_t)W + (size_t)px] = v;
				} else {
					sendStream[F.sendBase + position] = v;
				}
			} else if (owner == -1 && F.root == me) {
				image[(size_t)row * (size_t)W + (size_t)px] = F.p.clearColor;
			}
		} else if (owner >= 0 && owner != me) {
			image[(size_t)row * (size_t)W + (size_t)px] = recvStream[F.recvBase[owner] + position];
		}
	}
}

#if defined(CVX_EXPERIMENTS) || defined(CVX_PROFILE_SECTIONS) /* include/cpuvox_gpu_diag.h */
// ---------------------------------------------------------------------------
// arithmetic self-test (cvx_selftest_math): pins the device float contract.
// ---------------------------------------------------------------------------
__global__ void selftest_math_kernel(int op, int n, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) {
		return;
	}
	if (op == 16) { // texture row: groups of four floats {y, boundsX, boundsY, uvA.x} / {uvB.x, uvA.y, uvB.y, -} -> {exact row, cheap row, certain, 0} (int bits)
		if ((i & 3) == 0 && i + 3 < n) {
			const int py = (int)a[i];
			bool certain;
			const TexRun T = tex_run(a[i + 1], a[i + 2], a[i + 3], b[i], b[i + 1], b[i + 2]);
			const int cheap = tex_row_cheap(py, a[i + 1], a[i + 3], b[i + 1], T, certain);
			out[i] = __int_as_float(tex_row_exact(py, a[i + 1], a[i + 2], a[i + 3], b[i], b[i + 1], b[i + 2]));
			out[i + 1] = __int_as_float(cheap);
			out[i + 2] = __int_as_float(certain ? 1 : 0);
			out[i + 3] = 0.0f;
		}
		return;
	}
	const float x = a[i], y = b[i];
	float r;
	switch (op) {
	case 0: r = x / y; break;
	case 1: r = sqrtf(x); break;
	case 2: r = 1.0f / sqrtf(x); break;
	case 3: r = x + y * (y - x); break;
	case 4: r = floorf(x); break;
	case 5: r = ceilf(x); break;
	case 6: r = rintf(x); break;
	case 7: r = __int_as_float(f2i(x)); break;
	case 8: r = x * y; break;
	case 9: r = x + y; break;
	case 10: r = (div_safe(x) && div_safe(y)) ? quot_safe(x, recip_safe(y)) : x / y; break; // the short division form, guarded as in the kernel
	case 11: r = div_safe(y) ? quot_safe(1.0f, recip_safe(y)) : 1.0f / y; break;
	case 12: r = __int_as_float(f2i_floor(x)); break;
	case 13: r = hw_min(x, y); break;
	case 14: r = hw_max(x, y); break;
	case 15: r = div_safe(x) ? 1.0f : 0.0f; break;
	default: r = 0.0f; break;
	}
	out[i] = r;
}

#endif

#endif // CVX_DEVICE_FUNCTIONS_ONLY

} // namespace cvxk
