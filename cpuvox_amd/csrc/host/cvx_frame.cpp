// cvx_frame.cpp -- see cvx_frame.h.  float arithmetic in the order the
// reference's managed code performs it; Unity engine conventions restated from
// SURVEY.md Appendix B (not under /root/reference, hence unpinned).
#include "cvx_frame.h"

#include <cmath>
#include <cstring>

namespace cvx {

// ---------------------------------------------------------------------------
// Camera (UnityEngine.Camera / Transform conventions)
// ---------------------------------------------------------------------------
void Camera::SetEuler(float x, float y, float z)
{
	// R = Ry(y) * Rx(x) * Rz(z)  (Unity applies z, then x, then y)
	float sx = std::sin(x * kDeg2Rad), cx = std::cos(x * kDeg2Rad);
	float sy = std::sin(y * kDeg2Rad), cy = std::cos(y * kDeg2Rad);
	float sz = std::sin(z * kDeg2Rad), cz = std::cos(z * kDeg2Rad);
	// Rx * Rz
	float a[9] = {
		cz, -sz, 0.f,
		cx * sz, cx * cz, -sx,
		sx * sz, sx * cz, cx,
	};
	// Ry * (Rx * Rz)
	rot[0] = cy * a[0] + sy * a[6]; rot[1] = cy * a[1] + sy * a[7]; rot[2] = cy * a[2] + sy * a[8];
	rot[3] = a[3];                  rot[4] = a[4];                  rot[5] = a[5];
	rot[6] = -sy * a[0] + cy * a[6]; rot[7] = -sy * a[1] + cy * a[7]; rot[8] = -sy * a[2] + cy * a[8];
}

void Camera::SetForward(float3 v)
{
	float3 z = normalize(v);
	float3 x = normalize(cross(float3(0.f, 1.f, 0.f), z));
	float3 y = cross(z, x);
	rot[0] = x.x; rot[1] = y.x; rot[2] = z.x;
	rot[3] = x.y; rot[4] = y.y; rot[5] = z.y;
	rot[6] = x.z; rot[7] = y.z; rot[8] = z.z;
}

mat4 Camera::worldToCameraMatrix() const
{
	float3 r = right(), u = up(), f = forward();
	mat4 v = mat4::identity();
	v.at(0, 0) = r.x; v.at(0, 1) = r.y; v.at(0, 2) = r.z; v.at(0, 3) = -dot(r, position);
	v.at(1, 0) = u.x; v.at(1, 1) = u.y; v.at(1, 2) = u.z; v.at(1, 3) = -dot(u, position);
	v.at(2, 0) = -f.x; v.at(2, 1) = -f.y; v.at(2, 2) = -f.z; v.at(2, 3) = dot(f, position); // Scale(1,1,-1)
	return v;
}

mat4 Camera::nonJitteredProjectionMatrix() const
{
	float aspect = (float)pixelWidth / (float)pixelHeight;
	float t = std::tan(fieldOfView * kDeg2Rad * 0.5f);
	float n = nearClipPlane, f = farClipPlane;
	mat4 p{};
	for (int i = 0; i < 16; i++) { p.m[i] = 0.f; }
	p.at(0, 0) = 1.f / (aspect * t);
	p.at(1, 1) = 1.f / t;
	p.at(2, 2) = -(f + n) / (f - n);
	p.at(2, 3) = -2.f * f * n / (f - n);
	p.at(3, 2) = -1.f;
	return p;
}

void LimitRotationHorizon(Camera &camera)
{
	float3 forward = camera.forward();
	if (std::fabs(forward.y) < 0.001f) {
		forward.y = Sign(forward.y) * 0.001f;
		camera.SetForward(forward);
	}
}

void SetupLods(Camera &cam, int worldMaxDimension, int resolutionX, int resolutionY, float lodError, float out[CVX_LOD_LEVELS])
{
	const int clipMultiplier = 2; // World.REPEAT_WORLD == false
	float clipMax = (float)(worldMaxDimension * clipMultiplier);
	cam.farClipPlane = clipMax;

	float pixelW = (1.f / resolutionX) * cam.pixelWidth;
	float pixelH = (1.f / resolutionY) * cam.pixelHeight;
	int middleWidth = cam.pixelWidth / 2;
	int middleHeight = cam.pixelHeight / 2;

	// ScreenPointToRay directions in camera space (rotation does not change
	// |a.dir - b.dir|, the only thing used below).
	float aspect = (float)cam.pixelWidth / (float)cam.pixelHeight;
	float t = std::tan(cam.fieldOfView * kDeg2Rad * 0.5f);
	auto rayDir = [&](float px, float py) {
		float nx = 2.f * px / cam.pixelWidth - 1.f;
		float ny = 2.f * py / cam.pixelHeight - 1.f;
		return normalize(float3(nx * aspect * t, ny * t, 1.f));
	};
	float3 a = rayDir((float)middleWidth, (float)middleHeight);
	float3 b = rayDir(middleWidth + pixelW, middleHeight + pixelH);

	bool have[CVX_LOD_LEVELS] = {};
	float lods[CVX_LOD_LEVELS] = {};
	float pixelWidth = 1.41f / lodError;

	for (float p = 0.f; p < 1.f; p += 0.0001f) {
		float rayDist = p * clipMax;
		float pAB = length(a * rayDist - b * rayDist);
		for (int j = 0; j < CVX_LOD_LEVELS; j++) {
			if (!have[j] && pAB > pixelWidth * (float)(2 << j)) {
				have[j] = true;
				lods[j] = p;
			}
		}
	}
	have[CVX_LOD_LEVELS - 1] = true;
	lods[CVX_LOD_LEVELS - 1] = 2.f; // the last LOD is never exited
	for (int i = 0; i < CVX_LOD_LEVELS; i++) {
		out[i] = std::ceil((have[i] ? lods[i] : 2.f) * clipMax);
	}
}

cvx_camera_data MakeCameraData(const Camera &camera, const float LODDistances[CVX_LOD_LEVELS], float screenX, float screenY)
{
	cvx_camera_data d;
	std::memset(&d, 0, sizeof d);
	d.FarClip = camera.farClipPlane;
	d.PositionXZ[0] = camera.position.x;
	d.PositionXZ[1] = camera.position.z;
	d.PositionY = camera.position.y;
	mat4 m = mul(camera.nonJitteredProjectionMatrix(), camera.worldToCameraMatrix());
	m = mul(mat4::scale(0.5f, 0.5f, 1.f), m);       // -1..1 -> -0.5..0.5
	m = mul(mat4::translate(0.5f, 0.5f, 1.f), m);   // -> 0..1
	m = mul(mat4::scale(screenX, screenY, 1.f), m); // -> 0..screen
	std::memcpy(d.WorldToScreenMatrix, m.m, sizeof m.m);
	d.InverseElementIterationDirection = camera.forward().y >= 0.f ? 1 : 0;
	for (int i = 0; i < CVX_LOD_LEVELS; i++) { d.LODDistances[i] = LODDistances[i]; }
	return d;
}

namespace {

// Matrix4x4.LookAt(Vector3.zero, forward, up): rotation with columns right, up', forward.
mat4 LookAtRotation(float3 forward, float3 up)
{
	float3 z = normalize(forward);
	float3 x = normalize(cross(up, z));
	float3 y = cross(z, x);
	mat4 m = mat4::identity();
	m.at(0, 0) = x.x; m.at(1, 0) = x.y; m.at(2, 0) = x.z;
	m.at(0, 1) = y.x; m.at(1, 1) = y.y; m.at(2, 1) = y.z;
	m.at(0, 2) = z.x; m.at(1, 2) = z.y; m.at(2, 2) = z.z;
	return m;
}

// Vector2.SignedAngle
float SignedAngle(float2 from, float2 to)
{
	float denominator = std::sqrt((from.x * from.x + from.y * from.y) * (to.x * to.x + to.y * to.y));
	float angle = 0.f;
	if (denominator >= 1e-15f) {
		float d = (from.x * to.x + from.y * to.y) / denominator;
		d = d < -1.f ? -1.f : (d > 1.f ? 1.f : d);
		angle = std::acos(d) * kRad2Deg;
	}
	return angle * Sign(from.x * to.y - from.y * to.x);
}

inline float msign(float x) { return (x > 0.f ? 1.f : 0.f) - (x < 0.f ? 1.f : 0.f); } // math.sign

// RenderManager.GetGenericSegmentParameters, RenderManager.cs:402-501
cvx_segment_data GetGenericSegmentParameters(const Camera &camera, float2 screen, float2 vpScreen, float distToOtherEnd,
                                             float2 neutral, int primaryAxis)
{
	cvx_segment_data segment;
	std::memset(&segment, 0, sizeof segment);
	int secondaryAxis = 1 - primaryAxis;

	float2 simpleCaseMin(vpScreen[secondaryAxis] - distToOtherEnd, vpScreen[secondaryAxis] - distToOtherEnd);
	float2 simpleCaseMax(vpScreen[secondaryAxis] + distToOtherEnd, vpScreen[secondaryAxis] + distToOtherEnd);
	float a = vpScreen[primaryAxis] + distToOtherEnd * msign(neutral[primaryAxis]);
	simpleCaseMin[primaryAxis] = a;
	simpleCaseMax[primaryAxis] = a;

	if (simpleCaseMax[secondaryAxis] <= 0.f || simpleCaseMin[secondaryAxis] >= screen[secondaryAxis]) {
		return segment; // the 45 degree rays are not on screen
	}

	float2 MinScreen, MaxScreen;
	if (vpScreen.x >= 0.f && vpScreen.y >= 0.f && vpScreen.x <= screen.x && vpScreen.y <= screen.y) {
		MinScreen = simpleCaseMin;
		MaxScreen = simpleCaseMax;
	} else {
		float2 dirSimpleMiddle = lerp(simpleCaseMin, simpleCaseMax, 0.5f) - vpScreen;
		float angleLeft = 90.f, angleRight = -90.f;
		float2 dirRight, dirLeft;
		float2 vectors[4] = { float2(0.f, 0.f), float2(0.f, screen[1]), float2(screen[0], 0.f), screen };
		for (int i = 0; i < 4; i++) {
			float2 dir = vectors[i] - vpScreen;
			float2 scaledEnd = dir * (distToOtherEnd / std::fabs(dir[primaryAxis]));
			float angle = SignedAngle(neutral, dir);
			if (angle < angleLeft) {
				angleLeft = angle;
				dirLeft = scaledEnd;
			}
			if (angle > angleRight) {
				angleRight = angle;
				dirRight = scaledEnd;
			}
		}
		float2 cornerLeft = dirLeft + vpScreen;
		float2 cornerRight = dirRight + vpScreen;
		if (angleLeft < -45.f) {
			cornerLeft = SignedAngle(dirSimpleMiddle, simpleCaseMax) > 0.f ? simpleCaseMin : simpleCaseMax;
		}
		if (angleRight > 45.f) {
			cornerRight = SignedAngle(dirSimpleMiddle, simpleCaseMax) < 0.f ? simpleCaseMin : simpleCaseMax;
		}
		bool swap = cornerLeft[secondaryAxis] > cornerRight[secondaryAxis];
		MinScreen = swap ? cornerRight : cornerLeft;
		MaxScreen = swap ? cornerLeft : cornerRight;
	}

	// TransformPixel, RenderManager.cs:487-500: screen -> camera-local world-axis XZ on the far plane
	mat4 matrix = inverse(camera.nonJitteredProjectionMatrix());
	matrix = mul(inverse(mat4::scale(1.f, 1.f, -1.f)), matrix);
	matrix = mul(LookAtRotation(camera.forward(), camera.up()), matrix);
	auto TransformPixel = [&](float2 pixel) {
		float2 ndc((pixel.x / (float)camera.pixelWidth - 0.5f) * 2.f, (pixel.y / (float)camera.pixelHeight - 0.5f) * 2.f);
		float4 val = mul(matrix, float4(ndc.x, ndc.y, 1.f, 1.f));
		return float2(val.x / val.w, val.z / val.w);
	};
	float2 rmin = TransformPixel(MinScreen);
	float2 rmax = TransformPixel(MaxScreen);

	segment.MinScreen[0] = MinScreen.x; segment.MinScreen[1] = MinScreen.y;
	segment.MaxScreen[0] = MaxScreen.x; segment.MaxScreen[1] = MaxScreen.y;
	segment.CamLocalPlaneRayMin[0] = rmin.x; segment.CamLocalPlaneRayMin[1] = rmin.y;
	segment.CamLocalPlaneRayMax[0] = rmax.x; segment.CamLocalPlaneRayMax[1] = rmax.y;
	int rayCount = RoundToInt(MaxScreen[secondaryAxis] - MinScreen[secondaryAxis]);
	segment.RayCount = rayCount > 0 ? rayCount : 0;
	return segment;
}

} // namespace

FrameSetup SetupFrame(const Camera &camera, const float LODDistances[CVX_LOD_LEVELS], int screenWidth, int screenHeight, int worldDimensionY)
{
	(void)worldDimensionY; // passed through by the reference (RenderManager.cs:129) but unused there
	FrameSetup out;
	std::memset(&out, 0, sizeof out);

	// CalculateVanishingPointWorld, RenderManager.cs:374-378.  Unity's Euler
	// extraction gives sin(eulerAngles.x) = -forward.y for any yaw / roll.
	float3 forward = camera.forward();
	float sinPitch = -forward.y;
	float3 vpWorld = camera.position + float3(0.f, 1.f, 0.f) * (-camera.nearClipPlane / sinPitch);

	// ProjectVanishingPointScreenToWorld, RenderManager.cs:380-394
	mat4 lookMatrix = LookAtRotation(forward, camera.up());
	mat4 viewMatrix = mul(mat4::scale(1.f, 1.f, -1.f), inverse(lookMatrix));
	mat4 localToScreenMatrix = mul(camera.nonJitteredProjectionMatrix(), viewMatrix);
	float3 localPos = vpWorld - camera.position;
	float4 camPos = mul(localToScreenMatrix, float4(localPos.x, localPos.y, localPos.z, 1.f));
	float2 vp((camPos.x / camPos.w * 0.5f + 0.5f) * (float)camera.pixelWidth, (camPos.y / camPos.w * 0.5f + 0.5f) * (float)camera.pixelHeight);

	float2 screen((float)screenWidth, (float)screenHeight);
	if (vp.y < screenHeight) { // RenderManager.cs:128-142
		out.segments[0] = GetGenericSegmentParameters(camera, screen, vp, screenHeight - vp.y, float2(0.f, 1.f), 1);
	}
	if (vp.y > 0.f) {
		out.segments[1] = GetGenericSegmentParameters(camera, screen, vp, vp.y, float2(0.f, -1.f), 1);
	}
	if (vp.x < screenWidth) {
		out.segments[2] = GetGenericSegmentParameters(camera, screen, vp, screenWidth - vp.x, float2(1.f, 0.f), 0);
	}
	if (vp.x > 0.f) {
		out.segments[3] = GetGenericSegmentParameters(camera, screen, vp, vp.x, float2(-1.f, 0.f), 0);
	}

	out.camera = MakeCameraData(camera, LODDistances, screen.x, screen.y);
	out.vanishingPointScreenSpace[0] = vp.x;
	out.vanishingPointScreenSpace[1] = vp.y;
	out.vanishingPointWorldSpace[0] = vpWorld.x;
	out.vanishingPointWorldSpace[1] = vpWorld.y;
	out.vanishingPointWorldSpace[2] = vpWorld.z;
	for (int i = 0; i < 4; i++) { out.totalRays += out.segments[i].RayCount; }
	return out;
}

// ---------------------------------------------------------------------------
// Benchmark path, Assets/Code/BenchmarkPath.anim:16-148
// ---------------------------------------------------------------------------
namespace {

struct Key3 {
	float time;
	float value[3];
	float slope[3]; // inSlope == outSlope for every key of the clip
};

const Key3 kEulerKeys[] = {
	{ 0.f, { 0.f, 45.f, 0.f }, { 0.f, 0.f, 0.f } },
	{ 0.25f, { 0.f, -45.f, 0.f }, { 0.f, -360.f, 0.f } },
	{ 0.5f, { -16.2f, -135.f, 0.f }, { 0.f, 0.f, 0.f } },
	{ 0.75f, { 59.12f, -135.f, 0.f }, { 0.f, 0.f, 0.f } },
	{ 0.875f, { 59.12f, -135.f, 180.f }, { 0.f, 0.f, 1440.f } },
	{ 1.f, { 59.12f, -135.f, 360.f }, { 0.f, 0.f, 0.f } },
	{ 1.15f, { 85.f, -225.5f, 360.f }, { 0.f, 0.f, 0.f } },
};
const Key3 kPositionKeys[] = {
	{ 0.f, { -0.1f, 0.5f, -0.1f }, { 0.f, 0.f, 0.f } },
	{ 0.25f, { 1.1f, 0.5f, -0.1f }, { 0.f, 0.f, 0.f } },
	{ 0.5f, { 0.9f, 0.3f, 0.9f }, { 0.f, 0.f, 0.f } },
	{ 0.75f, { 0.9f, 0.95f, 0.9f }, { 0.f, 0.f, 0.f } },
	{ 1.f, { 0.9f, 0.95f, 0.9f }, { 0.f, 0.f, 0.f } },
	{ 1.15f, { 0.427f, 0.95f, 0.52f }, { 0.f, 0.f, 0.f } },
};

// Unity AnimationCurve evaluation for unweighted keys = cubic Hermite.
template <int N>
void EvaluateCurve(const Key3 (&keys)[N], float t, float out[3])
{
	if (t <= keys[0].time) {
		for (int k = 0; k < 3; k++) { out[k] = keys[0].value[k]; }
		return;
	}
	if (t >= keys[N - 1].time) {
		for (int k = 0; k < 3; k++) { out[k] = keys[N - 1].value[k]; }
		return;
	}
	int i = 0;
	while (i + 1 < N && keys[i + 1].time <= t) { i++; }
	const Key3 &k0 = keys[i];
	const Key3 &k1 = keys[i + 1];
	float dt = k1.time - k0.time;
	float s = (t - k0.time) / dt;
	float s2 = s * s, s3 = s2 * s;
	float h00 = 2.f * s3 - 3.f * s2 + 1.f;
	float h10 = s3 - 2.f * s2 + s;
	float h01 = -2.f * s3 + 3.f * s2;
	float h11 = s3 - s2;
	for (int k = 0; k < 3; k++) {
		out[k] = h00 * k0.value[k] + h10 * dt * k0.slope[k] + h01 * k1.value[k] + h11 * dt * k1.slope[k];
	}
}

} // namespace

void SampleBenchmarkPath(float t, const float worldDims[3], float outPosition[3], float outEuler[3])
{
	float p[3];
	EvaluateCurve(kPositionKeys, t, p);
	EvaluateCurve(kEulerKeys, t, outEuler);
	for (int k = 0; k < 3; k++) { outPosition[k] = p[k] * worldDims[k]; } // UnityManager.cs:87
}

} // namespace cvx
