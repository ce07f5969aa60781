// cvx_render_manager.cpp -- see cvx_render_manager.h.
#include "cvx_render_manager.h"

#include <dlfcn.h>

#include <stdexcept>

namespace cvx {

struct RenderManager::Api {
	decltype(&cvx_create) create;
	decltype(&cvx_destroy) destroy;
	decltype(&cvx_last_error) last_error;
	decltype(&cvx_world_upload) world_upload;
	decltype(&cvx_set_resolution) set_resolution;
	decltype(&cvx_draw_segments) draw_segments;
	decltype(&cvx_clear_raybuffer) clear_raybuffer;
	decltype(&cvx_read_raybuffer) read_raybuffer;
	decltype(&cvx_blit_segments) blit_segments;
	decltype(&cvx_last_draw_ms) last_draw_ms;
};

namespace {
template <typename F>
void Bind(void *lib, const char *name, F &out)
{
	out = reinterpret_cast<F>(dlsym(lib, name));
	if (!out) {
		throw std::runtime_error(std::string("libcpuvox_gpu: missing symbol ") + name);
	}
}
} // namespace

void RenderManager::Check(int rc, const char *what) const
{
	if (rc != CVX_OK) {
		throw std::runtime_error(std::string(what) + ": " + (api && api->last_error ? api->last_error(ctx) : "?"));
	}
}

RenderManager::RenderManager(int device, int screenWidth_, int screenHeight_, const std::string &gpuLibraryPath)
{
	library = dlopen(gpuLibraryPath.c_str(), RTLD_NOW | RTLD_LOCAL);
	if (!library) {
		throw std::runtime_error(std::string("cannot load the HIP library (no CPU fallback exists): ") + dlerror());
	}
	api = new Api();
	try {
		Bind(library, "cvx_create", api->create);
		Bind(library, "cvx_destroy", api->destroy);
		Bind(library, "cvx_last_error", api->last_error);
		Bind(library, "cvx_world_upload", api->world_upload);
		Bind(library, "cvx_set_resolution", api->set_resolution);
		Bind(library, "cvx_draw_segments", api->draw_segments);
		Bind(library, "cvx_clear_raybuffer", api->clear_raybuffer);
		Bind(library, "cvx_read_raybuffer", api->read_raybuffer);
		Bind(library, "cvx_blit_segments", api->blit_segments);
		Bind(library, "cvx_last_draw_ms", api->last_draw_ms);
		int rc = api->create(device, &ctx);
		if (rc != CVX_OK) {
			throw std::runtime_error(std::string("cvx_create: ") + api->last_error(nullptr));
		}
		SetResolution(screenWidth_, screenHeight_);
	} catch (...) {
		if (ctx) { api->destroy(ctx); ctx = nullptr; }
		delete api;
		api = nullptr;
		dlclose(library);
		library = nullptr;
		throw;
	}
}

RenderManager::~RenderManager()
{
	if (ctx) { api->destroy(ctx); }
	delete api;
	if (library) { dlclose(library); }
}

void RenderManager::UploadWorld(const std::vector<World> &worldLODs)
{
	for (size_t i = 0; i < worldLODs.size(); i++) {
		const World &w = worldLODs[i];
		Check(api->world_upload(ctx, (int)i, w.StoragePointer(), w.StorageByteLength(), w.Dimensions().x, w.Dimensions().y, w.Dimensions().z, w.ColumnCount()),
		      "cvx_world_upload");
	}
	worldDimensionY = worldLODs.empty() ? 0 : worldLODs[0].Dimensions().y;
}

void RenderManager::ClearRayBuffer(ERenderMode renderMode)
{
	const uint32_t pink = 0x9314FFFFu; // ColorARGB32(255, 20, 147): bytes A=FF R=FF G=14 B=93 (RenderManager.cs:64)
	if (renderMode == ERenderMode::RayBufferLeftRight) {
		Check(api->clear_raybuffer(ctx, bufferIndex, CVX_RAYBUFFER_LEFTRIGHT, pink), "cvx_clear_raybuffer");
	} else if (renderMode == ERenderMode::RayBufferTopDown) {
		Check(api->clear_raybuffer(ctx, bufferIndex, CVX_RAYBUFFER_TOPDOWN, pink), "cvx_clear_raybuffer");
	}
}

bool RenderManager::SetResolution(int resolutionX, int resolutionY)
{
	if (screenWidth != resolutionX || screenHeight != resolutionY) {
		Check(api->set_resolution(ctx, resolutionX, resolutionY), "cvx_set_resolution");
		screenWidth = resolutionX;
		screenHeight = resolutionY;
		return true;
	}
	return false;
}

void RenderManager::DrawWorld(const Camera &camera, const float LODDistances[CVX_LOD_LEVELS], uint32_t *screenArgb32)
{
	// "Setup VP" + "Setup segment params" + new CameraData (RenderManager.cs:119-152)
	lastFrame = SetupFrame(camera, LODDistances, screenWidth, screenHeight, worldDimensionY);
	// "Draw planes" (:154-168): the GPU replacement of DrawSegments
	Check(api->draw_segments(ctx, lastFrame.segments, &lastFrame.camera, screenWidth, screenHeight, lastFrame.vanishingPointScreenSpace, bufferIndex, CVX_DRAW_SYNC),
	      "cvx_draw_segments");
	// "Apply textures" (:173-176) has no counterpart: the raybuffers never leave HBM.  "Blit raybuffer" (:178-190):
	Check(api->blit_segments(ctx, bufferIndex, screenArgb32), "cvx_blit_segments");
}

float RenderManager::LastDrawMs() const
{
	float ms = 0.f;
	Check(api->last_draw_ms(ctx, &ms), "cvx_last_draw_ms");
	return ms;
}

void RenderManager::ReadRayBuffer(int which, int firstRay, int rayCount, uint32_t *dst) const
{
	Check(api->read_raybuffer(ctx, bufferIndex, which, firstRay, rayCount, dst), "cvx_read_raybuffer");
}

} // namespace cvx
