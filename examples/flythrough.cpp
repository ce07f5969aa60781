// flythrough.cpp -- the reference's built-in benchmark fly-through (UnityManager.cs:79-97, BenchmarkPath.anim) driven through
// the two C ABIs only: libcpuvox_host (world building, camera, the RenderManager twin) and, behind it, libcpuvox_gpu.
//
//   flythrough <model.obj | file.world | proc:<dim>> [frames=60] [width=1280] [height=720] [out-prefix]
//
// Renders `frames` poses of the path one at a time (the interactive use of the reference: DrawWorld per frame, Phase 1 + Phase 2),
// prints the frame rate, and writes the first, middle and last image as <out-prefix>_<n>.ppm when a prefix is given.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cpuvox_host.h"

static int Die(const char *what)
{
	std::fprintf(stderr, "%s: %s\n", what, cvxh_last_error());
	return 1;
}

static void WritePpm(const std::string &path, const std::vector<uint32_t> &argb, int W, int H)
{
	FILE *f = std::fopen(path.c_str(), "wb");
	if (!f) { return; }
	std::fprintf(f, "P6\n%d %d\n255\n", W, H);
	for (int y = H - 1; y >= 0; y--) { // row 0 of the screen is the bottom row (Unity screen space)
		for (int x = 0; x < W; x++) {
			const uint32_t p = argb[(size_t)y * W + x]; // bytes in memory: A, R, G, B
			const unsigned char rgb[3] = { (unsigned char)(p >> 8), (unsigned char)(p >> 16), (unsigned char)(p >> 24) };
			std::fwrite(rgb, 1, 3, f);
		}
	}
	std::fclose(f);
}

int main(int argc, char **argv)
{
	if (argc < 2) {
		std::fprintf(stderr, "usage: %s <model.obj | file.world | proc:<dim>> [frames] [width] [height] [out-prefix]\n", argv[0]);
		return 2;
	}
	const std::string source = argv[1];
	const int frames = argc > 2 ? std::atoi(argv[2]) : 60;
	const int W = argc > 3 ? std::atoi(argv[3]) : 1280, H = argc > 4 ? std::atoi(argv[4]) : 720;
	const std::string prefix = argc > 5 ? argv[5] : "";

	cvxh_world_set *worlds = nullptr;
	int rc;
	if (source.rfind("proc:", 0) == 0) {
		const int dim = std::atoi(source.c_str() + 5);
		rc = cvxh_world_procedural(dim, dim, dim, 0x5EED2048u, 0, &worlds);
	} else if (source.size() > 6 && source.compare(source.size() - 6, 6, ".world") == 0) {
		rc = cvxh_world_load(source.c_str(), &worlds);
	} else {
		rc = cvxh_world_from_obj(source.c_str(), 512, 0, 1, 0, 0, 0, &worlds); // UnityManager defaults: max dimension GUI value, X flipped
	}
	if (rc != 0) { return Die("world"); }
	cvxh_world_info info;
	cvxh_world_info_get(worlds, 0, &info);
	std::printf("world %dx%dx%d, %lld voxels at LOD 0, %d LODs\n", info.dimX, info.dimY, info.dimZ, (long long)cvxh_world_lod0_voxels(worlds), cvxh_world_lod_count(worlds));

	cvxh_render_manager *rm = nullptr;
	if (cvxh_render_manager_create(0, W, H, nullptr, &rm) != 0) { return Die("render manager (is libcpuvox_gpu.so next to libcpuvox_host.so, and a HIP device present?)"); }
	if (cvxh_render_manager_upload_world(rm, worlds) != 0) { return Die("upload"); }
	int changed = 0;
	if (cvxh_render_manager_set_resolution(rm, W, H, &changed) != 0) { return Die("resolution"); }

	cvxh_camera_pose pose{};
	pose.fieldOfView = 85.0f; // Assets/Scenes/SampleScene.unity:176-178
	pose.nearClipPlane = 0.05f;
	pose.pixelWidth = W;
	pose.pixelHeight = H;
	const int maxDim = info.dimX > info.dimY ? (info.dimX > info.dimZ ? info.dimX : info.dimZ) : (info.dimY > info.dimZ ? info.dimY : info.dimZ);
	float lods[CVX_LOD_LEVELS], farClip = 0.0f;
	if (cvxh_setup_lods(&pose, maxDim, W, H, 1.0f, lods, &farClip) != 0) { return Die("lods"); }

	const float dims[3] = { (float)info.dimX, (float)info.dimY, (float)info.dimZ };
	std::vector<uint32_t> screen((size_t)W * H);
	const auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < frames; i++) {
		const float t = 1.15f * (float)i / (float)(frames > 1 ? frames - 1 : 1);
		cvxh_sample_benchmark_path(t, dims, pose.position, pose.eulerAngles);
		cvxh_render_manager_swap_buffers(rm);
		if (cvxh_render_manager_draw_world(rm, &pose, 1, farClip, lods, screen.data(), nullptr) != 0) { return Die("draw"); }
		if (!prefix.empty() && (i == 0 || i == frames / 2 || i == frames - 1)) {
			WritePpm(prefix + "_" + std::to_string(i) + ".ppm", screen, W, H);
		}
	}
	const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	std::printf("%d frames at %dx%d in %.3f s: %.1f fps (Phase 1 + Phase 2 + image read-back, one frame in flight)\n", frames, W, H, seconds, frames / seconds);

	cvxh_render_manager_destroy(rm);
	cvxh_world_free(worlds);
	return 0;
}
