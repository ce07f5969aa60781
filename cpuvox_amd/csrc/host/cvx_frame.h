// cvx_frame.h -- host-side per-frame setup: everything RenderManager.DrawWorld
// (Assets/Code/RenderManager.cs:111-194, 374-510) and UnityManager.LateUpdate
// (Assets/Code/UnityManager.cs:163-201, 417-458) compute on the managed side
// before the DrawSegments boundary: camera matrices (Unity conventions, no
// Unity), vanishing point, the 4 clamped segment triangles, CameraData, LOD
// distances, the benchmark fly-through path.
#pragma once

#include "cpuvox_gpu.h"
#include "cvx_host_math.h"

namespace cvx {

// A UnityEngine.Camera + Transform reduced to what the path reads.
struct Camera {
	float3 position;
	float rot[9];            // row-major 3x3 rotation; columns = right, up, forward
	float fieldOfView = 85.f; // vertical, degrees (Assets/Scenes/SampleScene.unity:176-178)
	float nearClipPlane = 0.05f;
	float farClipPlane = 1000.f;
	int pixelWidth = 0, pixelHeight = 0;

	float3 right() const { return { rot[0], rot[3], rot[6] }; }
	float3 up() const { return { rot[1], rot[4], rot[7] }; }
	float3 forward() const { return { rot[2], rot[5], rot[8] }; }

	// transform.eulerAngles = (x, y, z) degrees; Unity order: z, then x, then y.
	void SetEuler(float x, float y, float z);
	// transform.forward = v  (Quaternion.LookRotation(v, Vector3.up))
	void SetForward(float3 v);

	mat4 worldToCameraMatrix() const;        // Appendix B: Scale(1,1,-1) * inverse(TRS)
	mat4 nonJitteredProjectionMatrix() const; // OpenGL-style perspective
};

// UnityManager.LimitRotationHorizon, UnityManager.cs:193-201
void LimitRotationHorizon(Camera &camera);

// UnityManager.SetupLods, UnityManager.cs:417-458.  Also sets camera.farClipPlane.
void SetupLods(Camera &camera, int worldMaxDimension, int resolutionX, int resolutionY, float lodError, float out[CVX_LOD_LEVELS]);

// new CameraData(camera, LODDistances, screen), CameraData.cs:18-36
cvx_camera_data MakeCameraData(const Camera &camera, const float LODDistances[CVX_LOD_LEVELS], float screenX, float screenY);

struct FrameSetup {
	cvx_segment_data segments[4];
	cvx_camera_data camera;
	float vanishingPointScreenSpace[2];
	float vanishingPointWorldSpace[3];
	int totalRays;
};

// The part of RenderManager.DrawWorld before DrawSegments, RenderManager.cs:119-152
FrameSetup SetupFrame(const Camera &camera, const float LODDistances[CVX_LOD_LEVELS], int screenWidth, int screenHeight, int worldDimensionY);

// BenchmarkPath.anim sampled at normalised clip time t in [0, 1.15]
// (UnityManager.cs:86-87): position (multiplied by world dims) + euler angles.
void SampleBenchmarkPath(float t, const float worldDims[3], float outPosition[3], float outEuler[3]);
constexpr float kBenchmarkPathLength = 1.15f; // BenchmarkPath.anim:179 m_StopTime

} // namespace cvx
