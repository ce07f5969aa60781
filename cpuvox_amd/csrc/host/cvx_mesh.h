// cvx_mesh.h -- host-side OBJ import, rescale and triangle voxelizer: the
// preprocessing that turns datasets/mill.obj into a world (configs 1-2).
// Mirrors Assets/Code/Utils/ObjModel.cs, Utils/SimpleMesh.cs and
// VoxelizerHelper.cs: v x y z [r g b] / vt / f a b c in all v/vt/vn index forms,
// mtllib + usemtl with map_Kd diffuse textures (PNG, TGA, PPM: cvx_image.h).
#pragma once

#include <string>
#include <vector>

#include "cvx_host_math.h"
#include "cvx_image.h"
#include "cvx_world.h"

namespace cvx {

struct Color32 {
	uint8_t r, g, b, a;
};

// SimpleMesh.Vertex, SimpleMesh.cs:108-114
struct MeshVertex {
	float3 Position;
	Color32 Color;
	float2 UV;
	int MaterialIndex;
};

// SimpleMesh.Material, SimpleMesh.cs:116-135
struct MeshMaterial {
	std::string Name;
	int MaterialIndex = 0;
	Image Diffuse; // map_Kd; empty when the material has none (the reference would throw on GetDiffusePixel then: treated as white here)

	// GetDiffusePixel, SimpleMesh.cs:130-134: pixel = floor(uv * (size - 1)), no filtering.  The reference indexes the array
	// unchecked (uv outside [0, 1] throws); the pixel is clamped into the texture here.  rgba in [0, 1] like UnityEngine.Color.
	void GetDiffusePixel(float2 uv, float rgba[4]) const;
};

// SimpleMesh, SimpleMesh.cs:11-106 (indices are always 0..n-1, ObjModel.cs:164-167)
struct SimpleMesh {
	std::vector<MeshVertex> Vertices;
	std::vector<int> Indices;
	std::vector<MeshMaterial> Materials; // MaterialLib.Materials, SimpleMesh.cs:137-139

	// SimpleMesh.Rescale -> Remap_Internal, SimpleMesh.cs:49-106
	int3 Rescale(float maxDimension, float3 dimensionFlips);
};

// ObjModel.Import, ObjModel.cs:10-171
bool ImportObj(const std::string &path, bool swapYZ, SimpleMesh &mesh, std::string *error);

// WorldBuilder.Import (WordBuilder.cs:39-97) on top of VoxelizerHelper.GetVoxelsInternal
// (VoxelizerHelper.cs:28-132).
void VoxelizeMesh(const SimpleMesh &mesh, WorldBuilder &builder, int threads = 0);

// UnityManager "Convert" (UnityManager.cs:297-343): import, rescale (X flipped
// by default, UnityManager.cs:27), voxelize, LOD0 + 5 down-sampled levels.
bool BuildWorldFromObj(const std::string &path, int maxDimension, bool swapYZ, bool flipX, bool flipY, bool flipZ,
                       int threads, std::vector<World> &worlds, int64_t *lod0Voxels, std::string *error);

} // namespace cvx
