// cvx_world.h -- host-side world data model: the reference's RLE column store
// (Assets/Code/World.cs), column builder (Assets/Code/WordBuilder.cs), LOD
// down-sampling (World.cs:45-127), .world file format (WorldSaveFile.cs) and a
// seeded procedural heightmap generator for the 2048^3 / 4096^3 benchmark
// worlds (ours; the reference ships no large dataset).
//
// The in-memory layout of a World's storage is the reference's, byte for byte
// (ColumnCount 12-byte RLEColumn headers, then the 4-byte element/colour pool),
// so the blob can be handed straight to cvx_world_upload() -- or produced by
// the reference's own WorldSaveFile.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "cvx_host_math.h"

namespace cvx {

int DefaultThreads(); // worker threads used when a `threads` argument is <= 0 (cgroup-quota aware)

constexpr int LOD_LEVELS = 6; // UnityManager.cs:42

// Color24.cs:6-19: bytes a, r, g, b in memory order.
struct ColorARGB32 {
	uint8_t a, r, g, b;
};

// World.RLEElement, World.cs:245-259
struct RLEElement {
	int16_t ColorsIndex; // < 0: air
	int16_t Length;      // 0: guard
};

// World.RLEColumn, World.cs:161-169 (12 bytes, 4-aligned)
struct RLEColumn {
	int32_t storageOffset; // WorldAllocator.StoragePointer.Offset, in 4-byte elements
	uint16_t runCount;
	uint16_t worldMin;
	uint16_t worldMax;
};
static_assert(sizeof(RLEColumn) == 12, "RLEColumn must match the reference's 12-byte header");
static_assert(sizeof(RLEElement) == 4 && sizeof(ColorARGB32) == 4, "element / colour are 4 bytes");

// One voxel appended to a column before compression (WordBuilder.cs:137-146).
struct ColumnVoxel {
	int16_t Y;
	ColorARGB32 Color;
};

// Output of RLEColumnBuilder::ToFinalColumn before it is placed in a World.
struct FinalColumn {
	std::vector<RLEElement> runs;    // top -> bottom
	std::vector<ColorARGB32> colors; // top -> bottom, one per solid voxel
	int worldMin = 0, worldMax = 0;  // LOD-0 units (World.cs:232-233)
};

// WorldBuilder.RLEColumnBuilder, WordBuilder.cs:135-269
class RLEColumnBuilder {
public:
	void Clear() { voxels.clear(); }
	void SetVoxel(int Y, ColorARGB32 color) { voxels.push_back({ (int16_t)Y, color }); }
	bool Empty() const { return voxels.empty(); }
	// Returns false for an empty column (reference returns default(RLEColumn)).
	bool ToFinalColumn(int voxelScale, int16_t topY, FinalColumn &out, int64_t &totalVoxels);

private:
	std::vector<ColumnVoxel> voxels;
};

class World {
public:
	World() = default;
	World(int3 dimensions, int lod);                                           // World.cs:27-34
	World(int3 dimensions, int lod, const void *data, int64_t byteLength);     // World.cs:36-43 (copies)

	int3 Dimensions() const { return dimensions; }
	int Lod() const { return lod; }
	bool Exists() const { return !storage.empty(); }
	int MaxDimension() const { return dimensions.x > dimensions.y ? (dimensions.x > dimensions.z ? dimensions.x : dimensions.z) : (dimensions.y > dimensions.z ? dimensions.y : dimensions.z); }
	// World.cs:17 -- note the reference's (lod+1)^2 divisor (over-allocates for lod >= 2).
	int ColumnCount() const { return (dimensions.x * dimensions.z) / ((lod + 1) * (lod + 1)); }
	// World.cs:145-149
	int GetIndexKnownInBounds(int x, int z) const { return (x >> lod) * indexingMulX + (z >> lod); }

	const RLEColumn *Columns() const { return reinterpret_cast<const RLEColumn *>(storage.data()); }
	RLEColumn *Columns() { return reinterpret_cast<RLEColumn *>(storage.data()); }
	const RLEElement *Elements() const { return reinterpret_cast<const RLEElement *>(storage.data() + (size_t)ColumnCount() * sizeof(RLEColumn)); }

	// WorldAllocator.GetStartPointer / GetByteLength (World.cs:273-283); the
	// length here is the used length (header table + allocated elements).
	const void *StoragePointer() const { return storage.data(); }
	int64_t StorageByteLength() const { return (int64_t)storage.size(); }
	int64_t ElementCount() const { return elementAllocationCount; }

	// World.SetVoxelColumn (World.cs:151-159) fused with the RLEColumn ctor
	// (World.cs:190-234) and WorldAllocator.AllocateElements (World.cs:340-373).
	void StoreColumn(int index, const FinalColumn &column);

	// World.DownSample, World.cs:45-69
	World DownSample(int extraLods, int64_t *voxelCount, int threads) const;

	// Checks a blob that came from outside (a .world file, cvxh_world_from_blobs) before anything walks it: power-of-two
	// dimensions, the column table inside the blob, and for every used column the element range, both guards
	// (World.cs:205-209), positive run lengths that fit the column height and colours inside the pool.
	static bool ValidateBlob(int3 dimensions, int lod, const void *data, int64_t byteLength, std::string *error);

private:
	void DownSamplePartial(int x, int z, int extraLods, RLEColumnBuilder &builder) const; // World.cs:101-127

	std::vector<uint8_t> storage;
	int3 dimensions{};
	int lod = 0;
	int indexingMulX = 0;
	int64_t elementAllocationCount = 0;
};

// A per-column voxel source used to stream columns into ToLOD0World without
// keeping x*z growable lists alive (procedural worlds).
using ColumnSource = void (*)(void *user, int x, int z, RLEColumnBuilder &out);

// WorldBuilder, WordBuilder.cs:14-130
class WorldBuilder {
public:
	WorldBuilder(int x, int y, int z);
	int3 Dimensions() const { return dimensions; }
	void SetVoxel(int x, int y, int z, ColorARGB32 color) { WorldColumns[(size_t)x * dimensions.z + z].SetVoxel(y, color); }
	RLEColumnBuilder &Column(int xzIndex) { return WorldColumns[(size_t)xzIndex]; }
	// ToLOD0World, WordBuilder.cs:99-130
	World ToLOD0World(int64_t *voxelCount, int threads);
	// Same, columns produced on the fly by `source`.
	static World ToLOD0World(int x, int y, int z, ColumnSource source, void *user, int64_t *voxelCount, int threads);

private:
	int3 dimensions;
	std::vector<RLEColumnBuilder> WorldColumns;
};

// WorldSaveFile.Serialize / Deserialize, WorldSaveFile.cs:8-104
bool SerializeWorlds(const std::vector<World> &worlds, const std::string &filePath, std::string *error);
bool DeserializeWorlds(const std::string &filePath, std::vector<World> &worlds, std::string *error);

// Procedural benchmark world: seeded integer-hash
// value-noise heightmap, visible shell only, floating runs in ~20 % of the
// columns.  Builds all LOD_LEVELS levels.
std::vector<World> BuildProceduralWorld(int dimX, int dimY, int dimZ, uint32_t seed, int threads, int64_t *lod0Voxels);

} // namespace cvx
