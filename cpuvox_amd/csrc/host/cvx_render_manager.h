// cvx_render_manager.h -- no-Unity twin of the reference's RenderManager (Assets/Code/RenderManager.cs:12-256):
// same public surface (SetResolution, SwapBuffers, ClearRayBuffer, DrawWorld), with the Burst jobs behind
// DrawSegments replaced by libcpuvox_gpu (bound at run time with dlopen, so libcpuvox_host itself stays loadable on
// machines without a GPU; constructing a RenderManager without the library or a device throws).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "cpuvox_gpu.h"
#include "cvx_frame.h"
#include "cvx_world.h"

namespace cvx {

enum class ERenderMode { ScreenBuffer, RayBufferTopDown, RayBufferLeftRight }; // UnityManager.cs:485-490

class RenderManager {
public:
	static constexpr int BUFFER_COUNT = 2; // RenderManager.cs:14

	RenderManager(int device, int screenWidth, int screenHeight, const std::string &gpuLibraryPath);
	~RenderManager(); // RenderManager.Destroy, :43-51
	RenderManager(const RenderManager &) = delete;
	RenderManager &operator=(const RenderManager &) = delete;

	// `fixed (World* worldPtr = worldLODs)` happens per DrawWorld call in the reference (:155); the device copy is
	// made once here instead.
	void UploadWorld(const std::vector<World> &worldLODs);

	void SwapBuffers() { bufferIndex = (bufferIndex + 1) % BUFFER_COUNT; } // :53-56
	void ClearRayBuffer(ERenderMode renderMode);                            // :58-92 (pink 255,20,147)
	bool SetResolution(int resolutionX, int resolutionY);                   // :94-109: true when it changed

	// RenderManager.DrawWorld, :111-194: vanishing point + segment setup, DrawSegments (GPU), BlitSegments.
	// `screenArgb32` (W*H uint32, row 0 = bottom) may be null: the image then stays on the device.
	void DrawWorld(const Camera &camera, const float LODDistances[CVX_LOD_LEVELS], uint32_t *screenArgb32);

	int BufferIndex() const { return bufferIndex; }
	int ScreenWidth() const { return screenWidth; }
	int ScreenHeight() const { return screenHeight; }
	const FrameSetup &LastFrame() const { return lastFrame; }
	float LastDrawMs() const;
	cvx_context *Context() const { return ctx; }
	// Raw rows of a raybuffer in the reference layout (RayBuffer.Native.GetRayColumn, RayBuffer.cs:121-128).
	void ReadRayBuffer(int which, int firstRay, int rayCount, uint32_t *dst) const;

private:
	struct Api;
	void Check(int rc, const char *what) const;

	Api *api = nullptr;
	void *library = nullptr;
	cvx_context *ctx = nullptr;
	int bufferIndex = 0;
	int screenWidth = -1, screenHeight = -1;
	int worldDimensionY = 0;
	FrameSetup lastFrame{};
};

} // namespace cvx
