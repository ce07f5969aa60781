// RenderManagerHost.cs -- managed, no-Unity twin of the host side of the reference's RenderManager.DrawWorld
// (Assets/Code/RenderManager.cs:111-194, 374-510) and UnityManager.LimitRotationHorizon / SetupLods
// (Assets/Code/UnityManager.cs:193-201, 417-458), driving libcpuvox_gpu through CpuVoxGpu.cs.
// System.Numerics only.  NOT compiled in this repository's image (no .NET toolchain there); the same logic, in the
// same order of float operations, is what cpuvox_amd/csrc/host/cvx_frame.cpp implements and tests cover.
using System;
using System.Numerics;

namespace CpuVox.Gpu
{
	/// <summary>UnityEngine.Camera + Transform reduced to what the path reads (Unity conventions, SURVEY.md Appendix B).</summary>
	public sealed class HostCamera
	{
		public Vector3 Position;
		public Matrix4x4 Rotation = Matrix4x4.Identity; // rows: right, up, forward (System.Numerics is row-vector based)
		public float FieldOfView = 85f, NearClipPlane = 0.05f, FarClipPlane = 1000f;
		public int PixelWidth, PixelHeight;

		public Vector3 Right => new Vector3(Rotation.M11, Rotation.M12, Rotation.M13);
		public Vector3 Up => new Vector3(Rotation.M21, Rotation.M22, Rotation.M23);
		public Vector3 Forward => new Vector3(Rotation.M31, Rotation.M32, Rotation.M33);

		/// <summary>transform.eulerAngles = (x, y, z) degrees; Unity applies z, then x, then y.</summary>
		public void SetEuler(float x, float y, float z)
		{
			const float d2r = 0.017453292519943295f;
			// column-vector R = Ry * Rx * Rz  ==  row-vector Rz * Rx * Ry
			Rotation = Matrix4x4.CreateRotationZ(z * d2r) * Matrix4x4.CreateRotationX(x * d2r) * Matrix4x4.CreateRotationY(y * d2r);
		}

		/// <summary>transform.forward = v (Quaternion.LookRotation(v, Vector3.up)).</summary>
		public void SetForward(Vector3 v)
		{
			Vector3 z = Vector3.Normalize(v);
			Vector3 x = Vector3.Normalize(Vector3.Cross(Vector3.UnitY, z));
			Vector3 y = Vector3.Cross(z, x);
			Rotation = new Matrix4x4(x.X, x.Y, x.Z, 0, y.X, y.Y, y.Z, 0, z.X, z.Y, z.Z, 0, 0, 0, 0, 1);
		}
	}

	public static unsafe class FrameSetup
	{
		/// <summary>UnityManager.LimitRotationHorizon, UnityManager.cs:193-201 (Mathf.Sign(0) == 1).</summary>
		public static void LimitRotationHorizon(HostCamera camera)
		{
			Vector3 forward = camera.Forward;
			if (MathF.Abs(forward.Y) < 0.001f) {
				forward.Y = (forward.Y >= 0f ? 1f : -1f) * 0.001f;
				camera.SetForward(forward);
			}
		}

		// column-major 4x4 helpers in the memory order of Unity.Mathematics.float4x4 (m[c * 4 + r])
		static float[] Mul(float[] a, float[] b)
		{
			var r = new float[16];
			for (int c = 0; c < 4; c++) {
				for (int row = 0; row < 4; row++) {
					r[c * 4 + row] = a[0 + row] * b[c * 4 + 0] + a[4 + row] * b[c * 4 + 1] + a[8 + row] * b[c * 4 + 2] + a[12 + row] * b[c * 4 + 3];
				}
			}
			return r;
		}

		static float[] Scale(float x, float y, float z) => new float[] { x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1 };
		static float[] Translate(float x, float y, float z) => new float[] { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, x, y, z, 1 };

		/// <summary>camera.worldToCameraMatrix: Scale(1,1,-1) * inverse(TRS(pos, rot, 1)).</summary>
		public static float[] WorldToCamera(HostCamera cam)
		{
			Vector3 r = cam.Right, u = cam.Up, f = cam.Forward, p = cam.Position;
			return new float[] {
				r.X, u.X, -f.X, 0,
				r.Y, u.Y, -f.Y, 0,
				r.Z, u.Z, -f.Z, 0,
				-Vector3.Dot(r, p), -Vector3.Dot(u, p), Vector3.Dot(f, p), 1 };
		}

		/// <summary>camera.nonJitteredProjectionMatrix: OpenGL-style perspective.</summary>
		public static float[] Projection(HostCamera cam)
		{
			float aspect = (float)cam.PixelWidth / cam.PixelHeight;
			float t = MathF.Tan(cam.FieldOfView * 0.017453292519943295f * 0.5f);
			float n = cam.NearClipPlane, f = cam.FarClipPlane;
			var m = new float[16];
			m[0] = 1f / (aspect * t);
			m[5] = 1f / t;
			m[10] = -(f + n) / (f - n);
			m[14] = -2f * f * n / (f - n);
			m[11] = -1f;
			return m;
		}

		/// <summary>new CameraData(camera, LODDistances, screen), CameraData.cs:18-36.</summary>
		public static CameraData MakeCameraData(HostCamera cam, float[] lodDistances, float screenX, float screenY)
		{
			float[] m = Mul(Projection(cam), WorldToCamera(cam));
			m = Mul(Scale(0.5f, 0.5f, 1f), m);
			m = Mul(Translate(0.5f, 0.5f, 1f), m);
			m = Mul(Scale(screenX, screenY, 1f), m);
			CameraData d = default;
			for (int i = 0; i < 16; i++) { d.WorldToScreenMatrix[i] = m[i]; }
			d.PositionXZ[0] = cam.Position.X;
			d.PositionXZ[1] = cam.Position.Z;
			d.PositionY = cam.Position.Y;
			d.InverseElementIterationDirection = (byte)(cam.Forward.Y >= 0f ? 1 : 0);
			d.FarClip = cam.FarClipPlane;
			for (int i = 0; i < 6; i++) { d.LODDistances[i] = lodDistances[i]; }
			return d;
		}

		/// <summary>CalculateVanishingPointWorld + ProjectVanishingPointScreenToWorld, RenderManager.cs:374-394.
		/// sin(eulerAngles.x) == -forward.y for any yaw / roll, so the VP is position + up * (near / forward.y).</summary>
		public static Vector2 VanishingPointScreen(HostCamera cam)
		{
			Vector3 f = cam.Forward, r = cam.Right, u = cam.Up;
			Vector3 local = new Vector3(0f, -cam.NearClipPlane / -f.Y, 0f);
			// view = Scale(1,1,-1) * inverse(LookAt(0, forward, up)); camera-space coordinates of `local`
			Vector3 v = new Vector3(Vector3.Dot(r, local), Vector3.Dot(u, local), -Vector3.Dot(f, local));
			float[] p = Projection(cam);
			float cx = p[0] * v.X, cy = p[5] * v.Y, cw = -v.Z;
			return new Vector2((cx / cw * 0.5f + 0.5f) * cam.PixelWidth, (cy / cw * 0.5f + 0.5f) * cam.PixelHeight);
		}
		// GetGenericSegmentParameters (RenderManager.cs:402-501) is restated in cvx_frame.cpp::GetGenericSegmentParameters;
		// a managed host calls the same function through libcpuvox_host (cvxh_setup_frame) or ports it 1:1 from there.
	}

	/// <summary>
	/// What UnityManager.LateUpdate + RenderManager.DrawWorld do per frame, with the Burst jobs replaced by the GPU:
	///   SwapBuffers -> SetResolution -> LimitRotationHorizon -> (VP, segments, CameraData) -> DrawSegments -> BlitSegments.
	/// Frame setup comes from libcpuvox_host (cvxh_setup_frame); the draw is GpuRenderer.DrawSegments.
	/// </summary>
	public sealed unsafe class RenderManagerHost : IDisposable
	{
		[System.Runtime.InteropServices.DllImport("cpuvox_host")] static extern int cvxh_setup_frame(CameraPose* pose, int limitHorizon, float farClip, float* lodDistances, int screenWidth, int screenHeight, int worldDimensionY, Frame* outFrame);
		[System.Runtime.InteropServices.DllImport("cpuvox_host")] static extern int cvxh_setup_lods(CameraPose* pose, int worldMaxDimension, int resolutionX, int resolutionY, float lodError, float* outLods, float* outFarClip);

		[System.Runtime.InteropServices.StructLayout(System.Runtime.InteropServices.LayoutKind.Sequential)]
		public struct CameraPose { public fixed float Position[3]; public fixed float EulerAngles[3]; public float FieldOfView, NearClipPlane; public int PixelWidth, PixelHeight; }

		[System.Runtime.InteropServices.StructLayout(System.Runtime.InteropServices.LayoutKind.Sequential, Pack = 4)]
		public struct Frame { public SegmentData S0, S1, S2, S3; public CameraData Camera; public fixed float VanishingPointScreenSpace[2]; public fixed float VanishingPointWorldSpace[3]; public fixed float Forward[3]; public int TotalRays; }

		const int BUFFER_COUNT = 2; // RenderManager.cs:14
		readonly GpuRenderer gpu;
		int bufferIndex, screenWidth = -1, screenHeight = -1;
		readonly float[] lodDistances = new float[6];
		float farClip;

		public RenderManagerHost(int device, int screenWidth, int screenHeight) { gpu = new GpuRenderer(device); SetResolution(screenWidth, screenHeight); }

		public void SwapBuffers() { bufferIndex = (bufferIndex + 1) % BUFFER_COUNT; } // RenderManager.cs:53-56

		public bool SetResolution(int resolutionX, int resolutionY) // RenderManager.cs:94-109
		{
			if (screenWidth == resolutionX && screenHeight == resolutionY) { return false; }
			gpu.SetResolution(resolutionX, resolutionY);
			screenWidth = resolutionX;
			screenHeight = resolutionY;
			return true;
		}

		public void SetupLods(CameraPose pose, int worldMaxDimension, float lodError = 1f) // UnityManager.cs:417-458
		{
			fixed (float* l = lodDistances) {
				float far;
				if (cvxh_setup_lods(&pose, worldMaxDimension, screenWidth, screenHeight, lodError, l, &far) != 0) { throw new CvxException(-1, "cvxh_setup_lods"); }
				farClip = far;
			}
		}

		/// <summary>RenderManager.DrawWorld, RenderManager.cs:111-194; screen = W*H ARGB32, row 0 = bottom (may be null).</summary>
		public void DrawWorld(CameraPose pose, int worldDimensionY, void* screen)
		{
			Frame frame;
			fixed (float* l = lodDistances) {
				if (cvxh_setup_frame(&pose, 1, farClip, l, screenWidth, screenHeight, worldDimensionY, &frame) != 0) { throw new CvxException(-1, "cvxh_setup_frame"); }
			}
			gpu.DrawSegments(&frame.S0, &frame.Camera, screenWidth, screenHeight, frame.VanishingPointScreenSpace[0], frame.VanishingPointScreenSpace[1], bufferIndex);
			gpu.BlitSegments(bufferIndex, screen);
		}

		public GpuRenderer Gpu => gpu;
		public void Dispose() { gpu.Dispose(); }
	}
}
