// CpuVoxGpu.cs -- P/Invoke binding of libcpuvox_gpu.so (include/cpuvox_gpu.h) for the reference's C# host.
// No Unity, no Burst.  NOT compiled in this repository's image (no dotnet/mono/csc there); kept dependency-free
// (System.Runtime.InteropServices only) so that `dotnet build` on any machine with a .NET SDK picks it up.
// Struct layouts are the reference's own blittable structs:
//   SegmentData  <- RenderManager.SegmentData (Assets/Code/RenderManager.cs:503-510)
//   CameraData   <- CameraData (Assets/Code/Utils/CameraData.cs:11-16)
using System;
using System.Runtime.InteropServices;

namespace CpuVox.Gpu
{
	[StructLayout(LayoutKind.Sequential, Pack = 4)]
	public unsafe struct SegmentData
	{
		public fixed float MinScreen[2];
		public fixed float MaxScreen[2];
		public fixed float CamLocalPlaneRayMin[2];
		public fixed float CamLocalPlaneRayMax[2];
		public int RayCount;
	}

	[StructLayout(LayoutKind.Sequential, Pack = 4)]
	public unsafe struct CameraData
	{
		public fixed float WorldToScreenMatrix[16]; // column major: c0, c1, c2, c3 (Unity.Mathematics.float4x4)
		public fixed float PositionXZ[2];
		public float PositionY;
		public byte InverseElementIterationDirection; // C# bool in the reference struct
		fixed byte pad[3];
		public float FarClip;
		public fixed float LODDistances[6];
	}

	[StructLayout(LayoutKind.Sequential)]
	public unsafe struct Counters
	{
		public long S, E, C, P, R;
		public fixed long LodVisits[6];
	}

	[StructLayout(LayoutKind.Sequential, Pack = 4)]
	public struct RaybufferLayout
	{
		public int Width, RayCapacity, TileRays, TileCapacity;
		public long TileBytes;
	}

	[StructLayout(LayoutKind.Sequential, Pack = 4)]
	public struct RowSpan
	{
		public long PoolRow, PackedRow;
		public int Rows, Kind;
	}

	public sealed class CvxException : Exception
	{
		public readonly int Code;
		public CvxException(int code, string message) : base(message) { Code = code; }
	}

	/// <summary>Raw entry points, one per declaration of include/cpuvox_gpu.h.</summary>
	public static unsafe class Native
	{
		const string Lib = "cpuvox_gpu"; // libcpuvox_gpu.so

		[DllImport(Lib)] public static extern int cvx_create(int device, out IntPtr ctx);
		[DllImport(Lib)] public static extern void cvx_destroy(IntPtr ctx);
		[DllImport(Lib)] public static extern IntPtr cvx_last_error(IntPtr ctx);
		[DllImport(Lib)] public static extern int cvx_set_stream(IntPtr ctx, IntPtr hipStream);
		[DllImport(Lib)] public static extern int cvx_world_upload(IntPtr ctx, int lod, void* storage, long byteLength, int dimX, int dimY, int dimZ, int columnCount);
		[DllImport(Lib)] public static extern int cvx_world_downsample(IntPtr ctx, void* storage, long byteLength, int dimX, int dimY, int dimZ, int lod, int columnCount, int extraLods,
		                                                               out IntPtr outStorage, out long outByteLength, out int outColumnCount, out long outVoxelCount, out float outDeviceMs);
		[DllImport(Lib)] public static extern int cvx_world_build_lods(IntPtr ctx, void* storage, long byteLength, int dimX, int dimY, int dimZ, int columnCount, int levelCount,
		                                                               IntPtr* outStorage, long* outByteLength, int* outColumnCount, out float outDeviceMs);
		[DllImport(Lib)] public static extern void cvx_free(IntPtr p);
		[DllImport(Lib)] public static extern int cvx_set_resolution(IntPtr ctx, int resolutionX, int resolutionY);
		[DllImport(Lib)] public static extern int cvx_set_buffer_count(IntPtr ctx, int bufferCount);
		[DllImport(Lib)] public static extern int cvx_draw_segments(IntPtr ctx, SegmentData* segments, CameraData* camera, int screenWidth, int screenHeight, float* vanishingPointScreenSpace, int bufferIndex, int flags);
		[DllImport(Lib)] public static extern int cvx_draw_segments_batch(IntPtr ctx, int frameCount, SegmentData* segments, CameraData* cameras, int screenWidth, int screenHeight, float* vanishingPoints, int firstBufferIndex, int flags);
		[DllImport(Lib)] public static extern int cvx_set_shard(IntPtr ctx, int shardIndex, int shardCount);
		public const int CVX_LATENCY_AUTO = 0, CVX_LATENCY_NEVER = 1, CVX_LATENCY_ALWAYS = 2;
		[DllImport(Lib)] public static extern int cvx_set_latency_kernel(IntPtr ctx, int mode);
		[DllImport(Lib)] public static extern int cvx_synchronize(IntPtr ctx);
		[DllImport(Lib)] public static extern int cvx_clear_raybuffer(IntPtr ctx, int bufferIndex, int which, uint argb);
		[DllImport(Lib)] public static extern int cvx_read_raybuffer(IntPtr ctx, int bufferIndex, int which, int firstRay, int rayCount, void* dst);
		[DllImport(Lib)] public static extern int cvx_blit_segments(IntPtr ctx, int bufferIndex, void* dstHost);
		[DllImport(Lib)] public static extern int cvx_blit_segments_batch(IntPtr ctx, int firstBufferIndex, int frameCount, void* dstDevice, out IntPtr imagesDevice);
		[DllImport(Lib)] public static extern int cvx_bind_raybuffers(IntPtr ctx, void* topDown, long topDownBytes, void* leftRight, long leftRightBytes);
		[DllImport(Lib)] public static extern int cvx_raybuffer_device_ptr(IntPtr ctx, int bufferIndex, int which, out IntPtr ptr, out long bytes);
		[DllImport(Lib)] public static extern int cvx_screen_device_ptr(IntPtr ctx, out IntPtr ptr, out long bytes);
		[DllImport(Lib)] public static extern int cvx_last_draw_ms(IntPtr ctx, out float ms);
		[DllImport(Lib)] public static extern int cvx_draw_time_stats(IntPtr ctx, out double totalMs, out int draws, int reset);
		[DllImport(Lib)] public static extern int cvx_enable_counters(IntPtr ctx, int enable);
		[DllImport(Lib)] public static extern int cvx_get_counters(IntPtr ctx, out Counters counters);
		[DllImport(Lib)] public static extern IntPtr cvx_version();
		[DllImport(Lib)] public static extern int cvx_draw_segments_placed(IntPtr ctx, int frameCount, SegmentData* segments, CameraData* cameras, int screenWidth, int screenHeight, float* vanishingPoints, long tileCount, ulong* tileOut, int flags);
		[DllImport(Lib)] public static extern int cvx_get_raybuffer_layout(IntPtr ctx, int which, out RaybufferLayout layout);
		[DllImport(Lib)] public static extern int cvx_copy_rows(IntPtr ctx, IntPtr hipStream, int toPacked, long spanCount, RowSpan* spansDevice, void* packedDevice);
		// multi-GPU: shard plan (host arithmetic), library-owned RCCL communicator, tile exchange (RenderManager.cs:358-363 sharded)
		[DllImport(Lib)] public static extern int cvx_shard_plan_create(int frameCount, SegmentData* segments, float* vanishingPoints, int screenWidth, int screenHeight, int rank, int worldSize, out IntPtr plan);
		[DllImport(Lib)] public static extern void cvx_shard_plan_destroy(IntPtr plan);
		[DllImport(Lib)] public static extern long cvx_shard_plan_tile_count(IntPtr plan);
		[DllImport(Lib)] public static extern int cvx_shard_plan_sections(IntPtr plan, long* sendStart, long* dispStart);
		[DllImport(Lib)] public static extern int cvx_shard_plan_transfer(IntPtr plan, int peer, out long sendRow, out long sendRows, out long recvRow, out long recvRows);
		[DllImport(Lib)] public static extern int cvx_shard_plan_tile_out(IntPtr plan, void* sendBase, void* dispBase, ulong* tileOut);
		[DllImport(Lib)] public static extern int cvx_comm_unique_id(void* id128);
		[DllImport(Lib)] public static extern int cvx_comm_create(IntPtr ctx, void* id128, int rank, int worldSize, out IntPtr comm);
		[DllImport(Lib)] public static extern int cvx_comm_create_timeout(IntPtr ctx, void* id128, int rank, int worldSize, double timeoutSeconds, out IntPtr comm);
		[DllImport(Lib)] public static extern int cvx_comm_destroy(IntPtr comm);
		[DllImport(Lib)] public static extern int cvx_exchange(IntPtr ctx, IntPtr plan, IntPtr comm, IntPtr hipStream, void* sendBase, void* dispBase);
		// multi-GPU, image gather: every rank blits its own tiles' pixels, the display rank receives W * H pixels per frame
		[DllImport(Lib)] public static extern int cvx_image_plan_create(IntPtr ctx, int frameCount, SegmentData* segments, float* vanishingPoints, int screenWidth, int screenHeight, int rank, int worldSize, out IntPtr plan);
		[DllImport(Lib)] public static extern void cvx_image_plan_destroy(IntPtr plan);
		[DllImport(Lib)] public static extern long cvx_image_plan_tile_count(IntPtr plan);
		[DllImport(Lib)] public static extern int cvx_image_plan_sizes(IntPtr plan, out long localStoreBytes, out long sendPixels, out long recvPixels, out int imagesDisplayed);
		[DllImport(Lib)] public static extern int cvx_image_plan_transfer(IntPtr plan, int peer, out long sendPixel, out long sendPixels, out long recvPixel, out long recvPixels);
		[DllImport(Lib)] public static extern int cvx_image_plan_tile_out(IntPtr plan, void* localStore, ulong* tileOut);
		[DllImport(Lib)] public static extern int cvx_image_pack(IntPtr ctx, IntPtr plan, IntPtr hipStream, void* localStore, void* sendStream, void* images);
		[DllImport(Lib)] public static extern int cvx_image_exchange(IntPtr ctx, IntPtr plan, IntPtr comm, IntPtr hipStream, void* sendStream, void* recvStream);
		[DllImport(Lib)] public static extern int cvx_image_unpack(IntPtr ctx, IntPtr plan, IntPtr hipStream, void* recvStream, void* images);
	}

	/// <summary>
	/// Owns the device-side state RenderManager owns in the reference (RenderManager.cs:12-56): uploaded world LODs and
	/// the raybuffer pairs.  DrawSegments has the reference's signature minus the Unity texture wrappers.
	/// </summary>
	public sealed unsafe class GpuRenderer : IDisposable
	{
		IntPtr ctx;

		public GpuRenderer(int device = 0)
		{
			int rc = Native.cvx_create(device, out ctx);
			if (rc != 0) { throw new CvxException(rc, Marshal.PtrToStringAnsi(Native.cvx_last_error(IntPtr.Zero))); }
		}

		void Check(int rc)
		{
			if (rc != 0) { throw new CvxException(rc, Marshal.PtrToStringAnsi(Native.cvx_last_error(ctx))); }
		}

		/// <summary>Replaces `fixed (World* worldPtr = worldLODs)` (RenderManager.cs:155): hand over each World's raw storage.</summary>
		public void UploadWorld(int lod, void* storageStartPointer, long byteLength, int dimX, int dimY, int dimZ, int columnCount)
		{
			Check(Native.cvx_world_upload(ctx, lod, storageStartPointer, byteLength, dimX, dimY, dimZ, columnCount));
		}

		/// <summary>World.DownSample(extraLods) (World.cs:45) on the GPU: returns the new level's storage blob (headers + elements, the
		/// layout WorldSaveFile writes), to be wrapped exactly like WorldSaveFile.Deserialize wraps a file's world (WorldSaveFile.cs:86-92).
		/// The caller copies it into its own allocation and calls <see cref="Native.cvx_free"/> on the pointer.</summary>
		public IntPtr DownSample(void* storageStartPointer, long byteLength, int dimX, int dimY, int dimZ, int lod, int columnCount, int extraLods,
		                         out long outByteLength, out int outColumnCount, out long outVoxelCount)
		{
			Check(Native.cvx_world_downsample(ctx, storageStartPointer, byteLength, dimX, dimY, dimZ, lod, columnCount, extraLods,
			                                  out IntPtr blob, out outByteLength, out outColumnCount, out outVoxelCount, out float _));
			return blob;
		}

		/// <summary>RenderManager.SetResolution (RenderManager.cs:94-109).</summary>
		public void SetResolution(int resolutionX, int resolutionY) { Check(Native.cvx_set_resolution(ctx, resolutionX, resolutionY)); }

		/// <summary>Which kernel a draw runs on: Native.CVX_LATENCY_AUTO (default: one blocking frame -> the latency kernel), _NEVER, _ALWAYS.</summary>
		public void SetLatencyKernel(int mode) { Check(Native.cvx_set_latency_kernel(ctx, mode)); }

		/// <summary>RenderManager.DrawSegments (RenderManager.cs:258-372); blocks like render.Complete().</summary>
		public void DrawSegments(SegmentData* segments4, CameraData* camera, int screenWidth, int screenHeight, float vpX, float vpY, int bufferIndex)
		{
			float* vp = stackalloc float[2];
			vp[0] = vpX;
			vp[1] = vpY;
			Check(Native.cvx_draw_segments(ctx, segments4, camera, screenWidth, screenHeight, vp, bufferIndex, 0));
		}

		/// <summary>RenderManager.BlitSegments + RayBufferBlit.shader: W*H ARGB32 pixels, row 0 = bottom.</summary>
		public void BlitSegments(int bufferIndex, void* dstArgb32) { Check(Native.cvx_blit_segments(ctx, bufferIndex, dstArgb32)); }

		/// <summary>Phase 2 of a batch in one launch; the images stay in device memory (returns the address of image 0).</summary>
		public IntPtr BlitSegmentsBatch(int firstBufferIndex, int frameCount, void* dstDevice = null) { Check(Native.cvx_blit_segments_batch(ctx, firstBufferIndex, frameCount, dstDevice, out IntPtr images)); return images; }

		/// <summary>Rows of a raybuffer in the reference's layout (RayBuffer.Native.GetRayColumn, RayBuffer.cs:121-128).</summary>
		public void ReadRayBuffer(int bufferIndex, int which, int firstRay, int rayCount, void* dst) { Check(Native.cvx_read_raybuffer(ctx, bufferIndex, which, firstRay, rayCount, dst)); }

		/// <summary>One frame sharded over worldSize GPUs (one process and one GpuRenderer per GPU): this rank renders its tiles straight into
		/// the send / display areas and the exchange completes the frames this rank displays -- the multi-GPU form of
		/// `render.Complete()` (RenderManager.cs:358-363).  comm: cvx_comm_create (the 128-byte id travels over the host's own channel).</summary>
		public void DrawSegmentsSharded(int frameCount, SegmentData* segments, CameraData* cameras, float* vanishingPoints, int screenWidth, int screenHeight,
		                                int rank, int worldSize, IntPtr comm, void* sendArea, void* displayArea)
		{
			int rc = Native.cvx_shard_plan_create(frameCount, segments, vanishingPoints, screenWidth, screenHeight, rank, worldSize, out IntPtr plan);
			if (rc != 0) { throw new CvxException(rc, Marshal.PtrToStringAnsi(Native.cvx_last_error(IntPtr.Zero))); }
			try {
				long tiles = Native.cvx_shard_plan_tile_count(plan);
				ulong[] tileOut = new ulong[Math.Max(1, tiles)];
				fixed (ulong* p = tileOut) {
					Check(Native.cvx_shard_plan_tile_out(plan, sendArea, displayArea, p));
					Check(Native.cvx_draw_segments_placed(ctx, frameCount, segments, cameras, screenWidth, screenHeight, vanishingPoints, tiles, p, 1));
				}
				Check(Native.cvx_exchange(ctx, plan, comm, IntPtr.Zero, sendArea, displayArea));
				Check(Native.cvx_synchronize(ctx));
			} finally {
				Native.cvx_shard_plan_destroy(plan);
			}
		}

		public void Dispose()
		{
			if (ctx != IntPtr.Zero) { Native.cvx_destroy(ctx); ctx = IntPtr.Zero; }
		}
	}
}
