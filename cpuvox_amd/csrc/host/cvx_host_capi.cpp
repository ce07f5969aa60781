// cvx_host_capi.cpp -- extern "C" surface of libcpuvox_host.so (include/cpuvox_host.h).
#include "cpuvox_host.h"

#include <cstring>
#include <exception>
#include <dlfcn.h>

#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "cvx_frame.h"
#include "cvx_mesh.h"
#include "cvx_render_manager.h"
#include "cvx_image.h"
#include "cvx_world.h"

struct cvxh_world_set {
	std::vector<cvx::World> worlds;
	int64_t lod0Voxels = 0;
};

struct cvxh_world_builder {
	std::unique_ptr<cvx::WorldBuilder> builder;
};

namespace {
thread_local std::string g_error;

int Fail(const std::string &msg)
{
	g_error = msg;
	return CVX_ERR_INVALID_ARGUMENT;
}

cvx::Camera CameraFromPose(const cvxh_camera_pose &pose)
{
	cvx::Camera cam;
	cam.position = cvx::float3(pose.position[0], pose.position[1], pose.position[2]);
	cam.SetEuler(pose.eulerAngles[0], pose.eulerAngles[1], pose.eulerAngles[2]);
	cam.fieldOfView = pose.fieldOfView;
	cam.nearClipPlane = pose.nearClipPlane;
	cam.pixelWidth = pose.pixelWidth;
	cam.pixelHeight = pose.pixelHeight;
	return cam;
}
} // namespace

extern "C" {

const char *cvxh_last_error(void) { return g_error.c_str(); }
int cvxh_default_threads(void) { return cvx::DefaultThreads(); }

const char *cvxh_version(void) { return "cpuvox_host 0.1"; }

int cvxh_world_from_obj(const char *path, int maxDimension, int swapYZ, int flipX, int flipY, int flipZ, int threads, cvxh_world_set **out)
{
	if (!path || !out || maxDimension <= 0) { return Fail("bad argument"); }
	auto set = std::make_unique<cvxh_world_set>();
	std::string error;
	if (!cvx::BuildWorldFromObj(path, maxDimension, swapYZ != 0, flipX != 0, flipY != 0, flipZ != 0, threads, set->worlds, &set->lod0Voxels, &error)) {
		return Fail(error);
	}
	*out = set.release();
	return CVX_OK;
}

int cvxh_world_procedural(int dimX, int dimY, int dimZ, uint32_t seed, int threads, cvxh_world_set **out)
{
	if (!out) { return Fail("bad argument"); }
	try {
		auto set = std::make_unique<cvxh_world_set>();
		set->worlds = cvx::BuildProceduralWorld(dimX, dimY, dimZ, seed, threads, &set->lod0Voxels);
		*out = set.release();
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

int cvxh_world_load(const char *path, cvxh_world_set **out)
{
	if (!path || !out) { return Fail("bad argument"); }
	auto set = std::make_unique<cvxh_world_set>();
	std::string error;
	if (!cvx::DeserializeWorlds(path, set->worlds, &error)) { return Fail(error); }
	*out = set.release();
	return CVX_OK;
}

int cvxh_world_save(const cvxh_world_set *worlds, const char *path)
{
	if (!worlds || !path) { return Fail("bad argument"); }
	std::string error;
	if (!cvx::SerializeWorlds(worlds->worlds, path, &error)) { return Fail(error); }
	return CVX_OK;
}

void cvxh_world_free(cvxh_world_set *worlds) { delete worlds; }

int cvxh_world_from_blobs(int dimX, int dimY, int dimZ, int count, const void *const *blobs, const int64_t *byteLengths, cvxh_world_set **out)
{
	if (!out || !blobs || !byteLengths || count <= 0 || dimX <= 0 || dimY <= 0 || dimZ <= 0) { return Fail("bad argument"); }
	try {
		auto set = std::make_unique<cvxh_world_set>();
		cvx::int3 dims; dims.x = dimX; dims.y = dimY; dims.z = dimZ;
		for (int i = 0; i < count; i++) {
			const int64_t headerBytes = ((int64_t)dimX * dimZ) / ((int64_t)(i + 1) * (i + 1)) * 12; // World.ColumnCount, World.cs:17
			if (!blobs[i] || byteLengths[i] < headerBytes) { return Fail("blob " + std::to_string(i) + " is shorter than its column table"); }
			std::string why;
			if (!cvx::World::ValidateBlob(dims, i, blobs[i], byteLengths[i], &why)) { return Fail(why); }
			set->worlds.emplace_back(dims, i, blobs[i], byteLengths[i]);
		}
		*out = set.release();
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

int cvxh_image_load(const char *path, int32_t *width, int32_t *height, uint8_t *rgba, int64_t capacityBytes)
{
	if (!path || !width || !height) { return Fail("bad argument"); }
	try {
		cvx::Image img;
		std::string why;
		if (!cvx::LoadImageFile(path, img, &why)) { return Fail(why); }
		*width = img.width;
		*height = img.height;
		if (rgba) {
			if (capacityBytes < (int64_t)img.rgba.size()) { return Fail("image buffer too small"); }
			std::memcpy(rgba, img.rgba.data(), img.rgba.size());
		}
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

int cvxh_world_downsample_seconds(const cvxh_world_set *worlds, int extraLods, int threads, double *outSeconds, int64_t *outVoxelCount)
{
	if (!worlds || worlds->worlds.empty() || extraLods < 1 || !outSeconds) { return Fail("bad argument"); }
	try {
		int64_t voxels = 0;
		const auto t0 = std::chrono::steady_clock::now();
		cvx::World lod = worlds->worlds[0].DownSample(extraLods, &voxels, threads);
		*outSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (outVoxelCount) { *outVoxelCount = voxels; }
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

int cvxh_world_lod_count(const cvxh_world_set *worlds) { return worlds ? (int)worlds->worlds.size() : 0; }

int64_t cvxh_world_lod0_voxels(const cvxh_world_set *worlds) { return worlds ? worlds->lod0Voxels : 0; }

int cvxh_world_info_get(const cvxh_world_set *worlds, int lod, cvxh_world_info *out)
{
	if (!worlds || !out || lod < 0 || lod >= (int)worlds->worlds.size()) { return Fail("bad argument"); }
	const cvx::World &w = worlds->worlds[(size_t)lod];
	out->storage = w.StoragePointer();
	out->byteLength = w.StorageByteLength();
	out->dimX = w.Dimensions().x;
	out->dimY = w.Dimensions().y;
	out->dimZ = w.Dimensions().z;
	out->lod = w.Lod();
	out->columnCount = w.ColumnCount();
	out->elementCount = w.ElementCount();
	return CVX_OK;
}

int cvxh_world_builder_create(int dimX, int dimY, int dimZ, cvxh_world_builder **out)
{
	if (!out) { return Fail("bad argument"); }
	try {
		auto b = std::make_unique<cvxh_world_builder>();
		b->builder = std::make_unique<cvx::WorldBuilder>(dimX, dimY, dimZ);
		*out = b.release();
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

int cvxh_world_builder_set_voxels(cvxh_world_builder *b, int64_t n, const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *argb)
{
	if (!b || !b->builder || n < 0 || (n > 0 && (!x || !y || !z || !argb))) { return Fail("bad argument"); }
	cvx::int3 d = b->builder->Dimensions();
	for (int64_t i = 0; i < n; i++) {
		if (x[i] < 0 || y[i] < 0 || z[i] < 0 || x[i] >= d.x || y[i] >= d.y || z[i] >= d.z) { return Fail("voxel out of bounds"); }
		cvx::ColorARGB32 c;
		std::memcpy(&c, &argb[i], 4);
		b->builder->SetVoxel(x[i], y[i], z[i], c);
	}
	return CVX_OK;
}

int cvxh_world_builder_finish(cvxh_world_builder *b, int threads, cvxh_world_set **out)
{
	if (!b || !b->builder || !out) { return Fail("bad argument"); }
	try {
		auto set = std::make_unique<cvxh_world_set>();
		set->worlds.push_back(b->builder->ToLOD0World(&set->lod0Voxels, threads));
		for (int j = 1; j < cvx::LOD_LEVELS; j++) {
			set->worlds.push_back(set->worlds[0].DownSample(j, nullptr, threads));
		}
		*out = set.release();
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

void cvxh_world_builder_free(cvxh_world_builder *b) { delete b; }

int cvxh_setup_lods(const cvxh_camera_pose *pose, int worldMaxDimension, int resolutionX, int resolutionY, float lodError,
                    float outLODDistances[CVX_LOD_LEVELS], float *outFarClip)
{
	if (!pose || !outLODDistances || resolutionX <= 0 || resolutionY <= 0 || lodError <= 0.f) { return Fail("bad argument"); }
	cvx::Camera cam = CameraFromPose(*pose);
	cvx::SetupLods(cam, worldMaxDimension, resolutionX, resolutionY, lodError, outLODDistances);
	if (outFarClip) { *outFarClip = cam.farClipPlane; }
	return CVX_OK;
}

int cvxh_setup_frame(const cvxh_camera_pose *pose, int limitHorizon, float farClip, const float LODDistances[CVX_LOD_LEVELS],
                     int screenWidth, int screenHeight, int worldDimensionY, cvxh_frame *out)
{
	if (!pose || !LODDistances || !out || screenWidth <= 0 || screenHeight <= 0) { return Fail("bad argument"); }
	cvx::Camera cam = CameraFromPose(*pose);
	cam.pixelWidth = screenWidth; // fakeCamera.pixelRect = (0,0,resX,resY), UnityManager.cs:180
	cam.pixelHeight = screenHeight;
	cam.farClipPlane = farClip;
	if (limitHorizon) { cvx::LimitRotationHorizon(cam); }
	cvx::FrameSetup fs = cvx::SetupFrame(cam, LODDistances, screenWidth, screenHeight, worldDimensionY);
	std::memcpy(out->segments, fs.segments, sizeof fs.segments);
	out->camera = fs.camera;
	out->vanishingPointScreenSpace[0] = fs.vanishingPointScreenSpace[0];
	out->vanishingPointScreenSpace[1] = fs.vanishingPointScreenSpace[1];
	std::memcpy(out->vanishingPointWorldSpace, fs.vanishingPointWorldSpace, sizeof fs.vanishingPointWorldSpace);
	cvx::float3 f = cam.forward();
	out->forward[0] = f.x; out->forward[1] = f.y; out->forward[2] = f.z;
	out->totalRays = fs.totalRays;
	return CVX_OK;
}

struct cvxh_render_manager {
	std::unique_ptr<cvx::RenderManager> rm;
};

int cvxh_render_manager_create(int device, int screenWidth, int screenHeight, const char *gpuLibraryPath, cvxh_render_manager **out)
{
	if (!out) { return Fail("bad argument"); }
	try {
		std::string path = gpuLibraryPath ? gpuLibraryPath : "";
		if (path.empty()) { // default: libcpuvox_gpu.so in the directory this library was loaded from
			Dl_info self{};
			if (dladdr(reinterpret_cast<const void *>(&cvxh_render_manager_create), &self) && self.dli_fname) {
				path = self.dli_fname;
				const size_t slash = path.find_last_of('/');
				path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/libcpuvox_gpu.so";
			} else {
				path = "libcpuvox_gpu.so";
			}
		}
		auto h = std::make_unique<cvxh_render_manager>();
		h->rm = std::make_unique<cvx::RenderManager>(device, screenWidth, screenHeight, path);
		*out = h.release();
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

void cvxh_render_manager_destroy(cvxh_render_manager *rm) { delete rm; }

int cvxh_render_manager_upload_world(cvxh_render_manager *rm, const cvxh_world_set *worlds)
{
	if (!rm || !worlds) { return Fail("bad argument"); }
	try {
		rm->rm->UploadWorld(worlds->worlds);
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

int cvxh_render_manager_set_resolution(cvxh_render_manager *rm, int resolutionX, int resolutionY, int *changed)
{
	if (!rm) { return Fail("bad argument"); }
	try {
		bool c = rm->rm->SetResolution(resolutionX, resolutionY);
		if (changed) { *changed = c ? 1 : 0; }
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

int cvxh_render_manager_swap_buffers(cvxh_render_manager *rm)
{
	if (!rm) { return Fail("bad argument"); }
	rm->rm->SwapBuffers();
	return rm->rm->BufferIndex();
}

int cvxh_render_manager_clear_raybuffer(cvxh_render_manager *rm, int renderMode)
{
	if (!rm || renderMode < 0 || renderMode > 2) { return Fail("bad argument"); }
	try {
		rm->rm->ClearRayBuffer(static_cast<cvx::ERenderMode>(renderMode));
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

int cvxh_render_manager_draw_world(cvxh_render_manager *rm, const cvxh_camera_pose *pose, int limitHorizon, float farClip,
                                   const float LODDistances[CVX_LOD_LEVELS], uint32_t *screenArgb32, cvxh_frame *outFrame)
{
	if (!rm || !pose || !LODDistances) { return Fail("bad argument"); }
	try {
		cvx::Camera cam = CameraFromPose(*pose);
		cam.pixelWidth = rm->rm->ScreenWidth(); // fakeCamera.pixelRect, UnityManager.cs:180
		cam.pixelHeight = rm->rm->ScreenHeight();
		cam.farClipPlane = farClip;
		if (limitHorizon) { cvx::LimitRotationHorizon(cam); }
		rm->rm->DrawWorld(cam, LODDistances, screenArgb32);
		if (outFrame) {
			const cvx::FrameSetup &fs = rm->rm->LastFrame();
			std::memcpy(outFrame->segments, fs.segments, sizeof fs.segments);
			outFrame->camera = fs.camera;
			outFrame->vanishingPointScreenSpace[0] = fs.vanishingPointScreenSpace[0];
			outFrame->vanishingPointScreenSpace[1] = fs.vanishingPointScreenSpace[1];
			std::memcpy(outFrame->vanishingPointWorldSpace, fs.vanishingPointWorldSpace, sizeof fs.vanishingPointWorldSpace);
			cvx::float3 f = cam.forward();
			outFrame->forward[0] = f.x; outFrame->forward[1] = f.y; outFrame->forward[2] = f.z;
			outFrame->totalRays = fs.totalRays;
		}
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

int cvxh_render_manager_read_raybuffer(cvxh_render_manager *rm, int which, int firstRay, int rayCount, uint32_t *dst)
{
	if (!rm || !dst) { return Fail("bad argument"); }
	try {
		rm->rm->ReadRayBuffer(which, firstRay, rayCount, dst);
	} catch (const std::exception &e) {
		return Fail(e.what());
	}
	return CVX_OK;
}

void cvxh_sample_benchmark_path(float t, const float worldDims[3], float outPosition[3], float outEuler[3])
{
	cvx::SampleBenchmarkPath(t, worldDims, outPosition, outEuler);
}

} // extern "C"
