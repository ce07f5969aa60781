// cvx_mesh.cpp -- see cvx_mesh.h.  CPU preprocessing only.
#include "cvx_mesh.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace cvx {

namespace {

// UnityEngine Color -> Color32: (byte)Mathf.Round(Mathf.Clamp01(c) * 255f)
inline uint8_t ToByte(float c)
{
	float v = c < 0.f ? 0.f : (c > 1.f ? 1.f : c);
	return (uint8_t)std::nearbyint(v * 255.f);
}

// ObjModel.ParseFaceIndex, ObjModel.cs:173-196 (returns zero-based index)
int ParseFaceIndex(const std::string &line, size_t &index)
{
	int result = 0;
	int sign = 1;
	if (index >= line.size()) { return -1; }
	char c = line[index];
	if (c == '-') {
		sign = -1;
		if (++index >= line.size()) { return -1; }
		c = line[index];
	}
	while (c >= '0' && c <= '9') {
		result = result * 10 + (c - '0');
		if (++index == line.size()) {
			break;
		}
		c = line[index];
	}
	return result * sign - 1;
}

bool StartsWith(const std::string &s, const char *p) { return s.compare(0, std::strlen(p), p) == 0; }

std::string DirectoryOf(const std::string &path)
{
	const size_t slash = path.find_last_of("/\\");
	return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}

// MaterialLib.ParseFromObj, SimpleMesh.cs:152-218: newmtl starts a material, map_Kd (with an optional "-bm <x>" prefix,
// :187-196) loads its diffuse texture; every other statement is ignored like in the reference.
bool ParseMaterialLib(const std::string &objPath, const std::string &relativeFilePath, std::vector<MeshMaterial> &materials, std::string *error)
{
	const std::string libPath = DirectoryOf(objPath) + "/" + relativeFilePath;
	std::ifstream file(libPath);
	if (!file) {
		if (error) { *error = "cannot open mtllib " + libPath; }
		return false;
	}
	materials.clear();
	std::string line;
	while (std::getline(file, line)) {
		if (!line.empty() && line.back() == '\r') { line.pop_back(); }
		if (line.empty() || line[0] == '#') { continue; }
		if (StartsWith(line, "newmtl ")) {
			MeshMaterial material;
			material.MaterialIndex = (int)materials.size();
			material.Name = line.substr(7);
			materials.push_back(material);
		} else if (StartsWith(line, "map_Kd ")) {
			if (materials.empty()) {
				if (error) { *error = libPath + ": map_Kd before newmtl"; }
				return false;
			}
			size_t idx = 7;
			if (idx + 2 < line.size() && line[idx] == '-' && line[idx + 1] == 'b' && line[idx + 2] == 'm') {
				idx += 4;
				while (idx < line.size() && line[idx] != ' ') { idx++; } // skip the -bm {x}
				idx++;
			}
			if (idx >= line.size()) {
				if (error) { *error = libPath + ": map_Kd without a path"; }
				return false;
			}
			std::string relativeMapPath = line.substr(idx);
			for (char &c : relativeMapPath) { if (c == '\\') { c = '/'; } }
			if (!LoadImageFile(DirectoryOf(libPath) + "/" + relativeMapPath, materials.back().Diffuse, error)) {
				return false;
			}
		}
	}
	return true;
}

} // namespace

bool ImportObj(const std::string &path, bool swapYZ, SimpleMesh &mesh, std::string *error)
{
	std::ifstream file(path);
	if (!file) {
		if (error) { *error = "cannot open " + path; }
		return false;
	}
	std::vector<float3> positionsLUT;
	std::vector<Color32> colorsLUT;
	std::vector<float2> uvLookupTable;
	mesh.Vertices.clear();
	mesh.Indices.clear();
	mesh.Materials.clear();
	int activeMaterial = -1;

	auto GatherVertex = [&](int positionIndex, int textureIndex) -> bool {
		if (positionIndex < 0 || positionIndex >= (int)positionsLUT.size()) { return false; }
		MeshVertex vertex{};
		vertex.Color = colorsLUT[(size_t)positionIndex];
		vertex.Position = positionsLUT[(size_t)positionIndex];
		if (textureIndex >= 0 && textureIndex < (int)uvLookupTable.size()) {
			vertex.UV = uvLookupTable[(size_t)textureIndex];
		}
		vertex.MaterialIndex = activeMaterial; // activeMaterial?.MaterialIndex ?? -1, ObjModel.cs:144
		mesh.Vertices.push_back(vertex);
		return true;
	};

	std::string line;
	while (std::getline(file, line)) {
		if (!line.empty() && line.back() == '\r') { line.pop_back(); }
		if (line.empty()) { continue; }
		if (StartsWith(line, "v ")) { // ParsePositionLine, ObjModel.cs:59-80
			std::istringstream ss(line.substr(2));
			std::vector<float> v;
			float f;
			while (ss >> f) { v.push_back(f); }
			if (v.size() < 3) { continue; }
			float3 pos(v[0], v[1], v[2]);
			if (swapYZ) { std::swap(pos.y, pos.z); }
			positionsLUT.push_back(pos);
			Color32 color;
			if (v.size() >= 6) { // subs.Length > 6 counts the leading "v"
				color = { ToByte(v[3]), ToByte(v[4]), ToByte(v[5]), 255 };
			} else {
				color = { 255, 255, 255, 255 };
			}
			colorsLUT.push_back(color);
		} else if (StartsWith(line, "vt ")) { // ParseUVLine, :52-57
			std::istringstream ss(line.substr(3));
			float2 uv;
			ss >> uv.x >> uv.y;
			uvLookupTable.push_back(uv);
		} else if (StartsWith(line, "f ")) { // ParseFaceLine, :87-137
			size_t index = 2;
			int entriesPerIndex = 1;
			for (size_t i = index; i < line.size() && line[i] != ' '; i++) {
				if (line[i] == '/') { entriesPerIndex++; }
			}
			bool ok = true;
			for (int i = 0; i < 3 && ok; i++) {
				int v = ParseFaceIndex(line, index);
				int vt = -1;
				if (entriesPerIndex >= 2) {
					index++; // skip '/'
					if (entriesPerIndex == 2 || (index < line.size() && line[index] != '/')) {
						vt = ParseFaceIndex(line, index);
					}
					if (entriesPerIndex == 3) {
						index++; // skip second '/'
						(void)ParseFaceIndex(line, index); // vn, unused
					}
				}
				index++; // skip the space between the indices
				ok = GatherVertex(v, vt);
			}
			if (!ok) {
				if (error) { *error = "bad face index in " + path; }
				return false;
			}
		}
		else if (StartsWith(line, "mtllib ")) { // ObjModel.cs:44-45
			if (!ParseMaterialLib(path, line.substr(7), mesh.Materials, error)) {
				return false;
			}
			activeMaterial = -1;
		} else if (StartsWith(line, "usemtl ")) { // GetByName, :48-49 / SimpleMesh.cs:141-149; unknown names give "no material"
			const std::string name = line.substr(7);
			activeMaterial = -1;
			for (const MeshMaterial &m : mesh.Materials) {
				if (m.Name == name) {
					activeMaterial = m.MaterialIndex;
					break;
				}
			}
		}
		// vn / o / s: ignored (ObjModel.cs:42-47)
	}
	mesh.Indices.resize(mesh.Vertices.size());
	for (size_t i = 0; i < mesh.Indices.size(); i++) { mesh.Indices[i] = (int)i; }
	if (mesh.Vertices.empty()) {
		if (error) { *error = "no faces in " + path; }
		return false;
	}
	return true;
}

void MeshMaterial::GetDiffusePixel(float2 uv, float rgba[4]) const
{
	if (Diffuse.width <= 0 || Diffuse.height <= 0) {
		rgba[0] = rgba[1] = rgba[2] = rgba[3] = 1.f;
		return;
	}
	int px = (int)std::floor(uv.x * (float)(Diffuse.width - 1));
	int py = (int)std::floor(uv.y * (float)(Diffuse.height - 1));
	px = std::max(0, std::min(Diffuse.width - 1, px));
	py = std::max(0, std::min(Diffuse.height - 1, py));
	const uint8_t *p = &Diffuse.rgba[((size_t)px + (size_t)py * (size_t)Diffuse.width) * 4];
	const float inv255 = 1.f / 255.f; // Color32 -> Color
	for (int i = 0; i < 4; i++) { rgba[i] = p[i] * inv255; }
}

// Remap_Internal, SimpleMesh.cs:64-106
int3 SimpleMesh::Rescale(float maxDimension, float3 dimensionFlips)
{
	float3 minimum = Vertices[0].Position;
	float3 maximum = Vertices[0].Position;
	for (size_t i = 1; i < Vertices.size(); i++) {
		minimum = min3(Vertices[i].Position, minimum);
		maximum = max3(Vertices[i].Position, maximum);
	}
	float3 size = maximum - minimum;
	float scale = maxDimension / std::fmax(size.x, std::fmax(size.y, size.z));

	int3 result;
	result.x = NextPowerOfTwo((int)(size.x * scale));
	result.y = NextPowerOfTwo((int)(size.y * scale));
	result.z = NextPowerOfTwo((int)(size.z * scale));

	for (auto &v : Vertices) {
		v.Position = (v.Position - minimum) * scale;
	}
	float3 flipScales((float)result.x, (float)result.y, (float)result.z);
	if (dimensionFlips.x < 1.f) { for (auto &v : Vertices) { v.Position.x = flipScales.x - v.Position.x; } }
	if (dimensionFlips.y < 1.f) { for (auto &v : Vertices) { v.Position.y = flipScales.y - v.Position.y; } }
	if (dimensionFlips.z < 1.f) { for (auto &v : Vertices) { v.Position.z = flipScales.z - v.Position.z; } }
	return result;
}

namespace {

constexpr int VOXELIZE_BUFFER_MAX = 1024 * 256; // WordBuilder.cs:37

inline int ClampI(int v, int lo, int hi) { return std::max(lo, std::min(hi, v)); }

// VoxelizerHelper.GetVoxelsInternal, VoxelizerHelper.cs:28-132, writing
// straight into the builder columns (WordBuilder.cs:76-88, no materials).
struct EmittedVoxel { // VoxelizerHelper.VoxelizedPosition after the material step (WordBuilder.cs:76-88)
	int32_t x, z;
	int16_t y;
	ColorARGB32 color;
};

void VoxelizeTriangle(const SimpleMesh &mesh, int indexStart, int3 dims, std::vector<EmittedVoxel> &out)
{
	const MeshVertex &v0 = mesh.Vertices[(size_t)mesh.Indices[(size_t)indexStart]];
	const MeshVertex &v1 = mesh.Vertices[(size_t)mesh.Indices[(size_t)indexStart + 1]];
	const MeshVertex &v2 = mesh.Vertices[(size_t)mesh.Indices[(size_t)indexStart + 2]];

	float3 a = v0.Position, b = v1.Position, c = v2.Position;

	float3 normalTri;
	{
		float3 normalCross = cross(b - a, c - a);
		float normalCrossLengthSqrd = dot(normalCross, normalCross);
		if (normalCrossLengthSqrd == 0.f) {
			return;
		}
		normalTri = normalCross * (1.0f / std::sqrt(normalCrossLengthSqrd));
	}

	// extend the triangle by half a voxel along the corner-middle directions (:52-57)
	float3 middle = (a + b + c) / 3.f;
	a = a + normalize(a - middle) * 0.5f;
	b = b + normalize(b - middle) * 0.5f;
	c = c + normalize(c - middle) * 0.5f;

	float3 minf = min3(a, min3(b, c));
	float3 maxf = max3(a, max3(b, c));
	int3 maxDimensions; maxDimensions.x = dims.x - 1; maxDimensions.y = dims.y - 1; maxDimensions.z = dims.z - 1;
	int3 mini, maxi;
	mini.x = ClampI((int)std::floor(minf.x), 0, maxDimensions.x);
	mini.y = ClampI((int)std::floor(minf.y), 0, maxDimensions.y);
	mini.z = ClampI((int)std::floor(minf.z), 0, maxDimensions.z);
	maxi.x = ClampI((int)std::ceil(maxf.x), 0, maxDimensions.x);
	maxi.y = ClampI((int)std::ceil(maxf.y), 0, maxDimensions.y);
	maxi.z = ClampI((int)std::ceil(maxf.z), 0, maxDimensions.z);

	int written = 0;
	const int materialIndex = (int)(int8_t)v0.MaterialIndex; // (sbyte)v0.MaterialIndex, VoxelizerHelper.cs:120
	const float inv255 = 1.f / 255.f; // Color32 -> Color
	float c0[3] = { v0.Color.r * inv255, v0.Color.g * inv255, v0.Color.b * inv255 };
	float c1[3] = { v1.Color.r * inv255, v1.Color.g * inv255, v1.Color.b * inv255 };
	float c2[3] = { v2.Color.r * inv255, v2.Color.g * inv255, v2.Color.b * inv255 };

	for (int x = mini.x; x <= maxi.x; x++) {
		for (int z = mini.z; z <= maxi.z; z++) {
			for (int y = mini.y; y <= maxi.y; y++) {
				float3 voxel((float)x + 0.5f, (float)y + 0.5f, (float)z + 0.5f);
				float normalDistToTriangle = dot(voxel - a, normalTri);
				if (std::fabs(normalDistToTriangle) > 0.5f) {
					continue;
				}
				float3 p = voxel - normalTri * normalDistToTriangle;
				float3 p0 = b - a;
				float3 p1 = c - a;
				float3 p2 = p - a;
				float d00 = dot(p0, p0);
				float d01 = dot(p0, p1);
				float d11 = dot(p1, p1);
				float d20 = dot(p2, p0);
				float d21 = dot(p2, p1);
				float denom = 1.f / (d00 * d11 - d01 * d01);
				float by = (d11 * d20 - d01 * d21) * denom;
				float bz = (d00 * d21 - d01 * d20) * denom;
				float bx = 1.0f - by - bz;
				if (bx < 0.f || by < 0.f || bz < 0.f || bx > 1.f || by > 1.f || bz > 1.f) {
					continue;
				}
				float rgb[3] = { c0[0] * bx + c1[0] * by + c2[0] * bz, c0[1] * bx + c1[1] * by + c2[1] * bz, c0[2] * bx + c1[2] * by + c2[2] * bz };
				bool keep = true;
				if (materialIndex >= 0 && materialIndex < (int)mesh.Materials.size()) { // WordBuilder.cs:78-84
					float2 uv;
					uv.x = v0.UV.x * bx + v1.UV.x * by + v2.UV.x * bz;
					uv.y = v0.UV.y * bx + v1.UV.y * by + v2.UV.y * bz;
					float albedo[4];
					mesh.Materials[(size_t)materialIndex].GetDiffusePixel(uv, albedo);
					if (albedo[3] < 1.f) {
						keep = false; // not fully opaque texels leave no voxel
					}
					for (int k = 0; k < 3; k++) { rgb[k] *= albedo[k]; }
				}
				if (keep) {
					ColorARGB32 color;
					color.r = ToByte(rgb[0]);
					color.g = ToByte(rgb[1]);
					color.b = ToByte(rgb[2]);
					color.a = 255;
					out.push_back({ x, z, (int16_t)y, color });
				}
				if (++written == VOXELIZE_BUFFER_MAX) {
					return; // buffer full, the triangle must have been huge (:124-126)
				}
			}
		}
	}
}

} // namespace

void VoxelizeMesh(const SimpleMesh &mesh, WorldBuilder &builder, int threads)
{
	// The reference splits the triangles over Environment.ProcessorCount tasks that append to the columns under a lock
	// (WordBuilder.cs:44-95).  The finished world does not depend on the order of insertion (ToFinalColumn sorts by Y and
	// averages voxels that share one), so here every thread voxelises its triangles into a private list and the lists are
	// appended to the columns one thread at a time.
	const int64_t triangles = (int64_t)(mesh.Indices.size() / 3);
	const int3 dims = builder.Dimensions();
#ifdef _OPENMP
	if (threads <= 0) { threads = DefaultThreads(); }
#else
	threads = 1;
#endif
#pragma omp parallel num_threads(threads)
	{
		std::vector<EmittedVoxel> local;
#pragma omp for schedule(dynamic, 64) nowait
		for (int64_t t = 0; t < triangles; t++) {
			VoxelizeTriangle(mesh, (int)(t * 3), dims, local);
		}
#pragma omp critical(cvx_voxelize_merge)
		for (const EmittedVoxel &v : local) {
			builder.SetVoxel(v.x, v.y, v.z, v.color);
		}
	}
}

bool BuildWorldFromObj(const std::string &path, int maxDimension, bool swapYZ, bool flipX, bool flipY, bool flipZ,
                       int threads, std::vector<World> &worlds, int64_t *lod0Voxels, std::string *error)
{
	worlds.clear();
	SimpleMesh mesh;
	if (!ImportObj(path, swapYZ, mesh, error)) {
		return false;
	}
	try {
		int3 worldDimensions = mesh.Rescale((float)maxDimension, float3(flipX ? -1.f : 1.f, flipY ? -1.f : 1.f, flipZ ? -1.f : 1.f));
		if (worldDimensions.x <= 0 || worldDimensions.y <= 0 || worldDimensions.z <= 0) {
			// Mathf.NextPowerOfTwo(0) == 0 (SimpleMesh.cs:78-80): a model without extent along an axis has no voxel grid
			throw std::runtime_error("model has no extent along at least one axis at this max dimension (world dimensions " +
			                         std::to_string(worldDimensions.x) + "x" + std::to_string(worldDimensions.y) + "x" + std::to_string(worldDimensions.z) + ")");
		}
		WorldBuilder builder(worldDimensions.x, worldDimensions.y, worldDimensions.z);
		VoxelizeMesh(mesh, builder, threads);
		worlds.push_back(builder.ToLOD0World(lod0Voxels, threads));
		for (int j = 1; j < LOD_LEVELS; j++) {
			worlds.push_back(worlds[0].DownSample(j, nullptr, threads));
		}
	} catch (const std::exception &e) {
		worlds.clear();
		if (error) { *error = e.what(); }
		return false;
	}
	return true;
}

} // namespace cvx
