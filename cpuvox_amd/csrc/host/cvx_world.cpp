// cvx_world.cpp -- see cvx_world.h.  Host-side (CPU) preprocessing only; the
// device never sees these classes, only the storage blob through the C ABI.
#include "cvx_world.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace cvx {

// Worker threads for `threads <= 0`: OpenMP's default capped by the control group's CPU quota (cgroup v2 cpu.max): a
// container usually sees every host CPU but may only use a few, and more runnable threads than that just take turns.
int DefaultThreads()
{
#ifdef _OPENMP
	int threads = omp_get_max_threads();
	if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
		long long quota = 0, period = 0;
		if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
			const long long cpus = (quota + period - 1) / period;
			if (cpus >= 1 && cpus < threads) { threads = (int)cpus; }
		}
		std::fclose(f);
	}
	return threads;
#else
	return 1;
#endif
}

// ---------------------------------------------------------------------------
// RLEColumnBuilder.ToFinalColumn, WordBuilder.cs:181-268
// ---------------------------------------------------------------------------
bool RLEColumnBuilder::ToFinalColumn(int voxelScale, int16_t topY, FinalColumn &out, int64_t &totalVoxels)
{
	out.runs.clear();
	out.colors.clear();
	if (voxels.empty()) {
		return false;
	}

	// sort the randomly ordered voxels in descending order (:188)
	std::stable_sort(voxels.begin(), voxels.end(), [](const ColumnVoxel &a, const ColumnVoxel &b) { return a.Y > b.Y; });

	// dedupe voxels that share a Y; average their colours (:192-228)
	int16_t dedupedCount = 0;
	{
		int r = 0, g = 0, b = 0, weight = 1;
		auto AddWeightsToPrevious = [&]() {
			ColumnVoxel previous = voxels[(size_t)dedupedCount - 1];
			previous.Color.r = (uint8_t)((previous.Color.r + r) / weight);
			previous.Color.g = (uint8_t)((previous.Color.g + g) / weight);
			previous.Color.b = (uint8_t)((previous.Color.b + b) / weight);
			voxels[(size_t)dedupedCount - 1] = previous;
			r = g = b = 0;
			weight = 1;
		};
		int lastY = -1;
		for (size_t i = 0; i < voxels.size(); i++) {
			ColumnVoxel voxel = voxels[i];
			if (voxel.Y == lastY) {
				r += voxel.Color.r;
				g += voxel.Color.g;
				b += voxel.Color.b;
				weight++;
			} else {
				if (weight > 1) {
					AddWeightsToPrevious();
				}
				voxels[(size_t)dedupedCount++] = voxel;
				lastY = voxel.Y;
			}
		}
		if (weight > 1) {
			AddWeightsToPrevious();
		}
	}

	totalVoxels += dedupedCount;

	// compress the sorted, deduped voxels into runs (:232-258)
	for (int16_t i = 0; i < dedupedCount;) {
		int16_t voxelY = voxels[(size_t)i].Y;
		int16_t airFromTop = (int16_t)(topY - voxelY);
		if (airFromTop > 0) {
			out.runs.push_back({ (int16_t)-1, airFromTop });
			topY = (int16_t)(topY - airFromTop);
		}

		int16_t runLength = 1;
		for (int16_t j = (int16_t)(i + 1); j < dedupedCount; j++) {
			if (topY - (j - i) == voxels[(size_t)j].Y) {
				runLength++;
			} else {
				break;
			}
		}

		out.runs.push_back({ i, runLength });
		topY = (int16_t)(topY - runLength);
		i = (int16_t)(i + runLength);
	}

	if (topY >= 0) {
		out.runs.push_back({ (int16_t)-1, (int16_t)(topY + 1) });
	}

	// RLEColumn ctor: world-space bounds of the solid part (World.cs:211-233)
	int worldMin = INT_MAX;
	int worldMax = INT_MIN;
	int elementBoundsMin = 0;
	int elementBoundsMax = 0;
	for (int i = (int)out.runs.size() - 1; i >= 0; i--) {
		RLEElement element = out.runs[(size_t)i];
		elementBoundsMin = elementBoundsMax;
		elementBoundsMax = elementBoundsMin + element.Length;
		if (element.ColorsIndex < 0) {
			continue;
		}
		worldMin = std::min(worldMin, elementBoundsMin);
		worldMax = std::max(worldMax, elementBoundsMax);
	}
	if (worldMin == INT_MAX) {
		throw std::runtime_error("only air elements in the RLE"); // World.cs:228-230
	}
	out.worldMin = (uint16_t)(worldMin * voxelScale);
	out.worldMax = (uint16_t)(worldMax * voxelScale);

	out.colors.resize((size_t)dedupedCount);
	for (int i = 0; i < dedupedCount; i++) { // :263-266
		out.colors[(size_t)i] = voxels[(size_t)i].Color;
	}
	return true;
}

// ---------------------------------------------------------------------------
// World
// ---------------------------------------------------------------------------
World::World(int3 dimensions_, int lod_) : dimensions(dimensions_), lod(lod_)
{
	indexingMulX = dimensions.z >> lod;
	storage.assign((size_t)ColumnCount() * sizeof(RLEColumn), 0);
}

World::World(int3 dimensions_, int lod_, const void *data, int64_t byteLength) : dimensions(dimensions_), lod(lod_)
{
	indexingMulX = dimensions.z >> lod;
	storage.assign((const uint8_t *)data, (const uint8_t *)data + byteLength);
	int64_t headerBytes = (int64_t)ColumnCount() * (int64_t)sizeof(RLEColumn);
	elementAllocationCount = byteLength > headerBytes ? (byteLength - headerBytes) / 4 : 0;
}

bool World::ValidateBlob(int3 dims, int lod, const void *data, int64_t byteLength, std::string *error)
{
	auto fail = [&](const std::string &what) {
		if (error) { *error = what; }
		return false;
	};
	auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
	if (!data || lod < 0 || lod > 15 || !pow2(dims.x) || !pow2(dims.y) || !pow2(dims.z) || dims.y > 65536 ||
	    (dims.x >> lod) < 1 || (dims.y >> lod) < 1 || (dims.z >> lod) < 1 || (int64_t)dims.x * dims.z > (int64_t)INT_MAX) {
		return fail("world dimensions must be powers of two that the LOD divides (WordBuilder.cs:30)");
	}
	const int64_t columnCount = ((int64_t)dims.x * dims.z) / ((int64_t)(lod + 1) * (lod + 1)); // World.cs:17
	const int64_t usedColumns = (int64_t)(dims.x >> lod) * (dims.z >> lod);
	if (byteLength < columnCount * 12 || usedColumns > columnCount) {
		return fail("LOD " + std::to_string(lod) + ": blob shorter than its column table");
	}
	const int64_t elementCount = (byteLength - columnCount * 12) / 4;
	const RLEColumn *columns = static_cast<const RLEColumn *>(data);
	const RLEElement *elements = reinterpret_cast<const RLEElement *>(static_cast<const uint8_t *>(data) + columnCount * 12);
	const int maxY = dims.y >> lod;
	for (int64_t i = 0; i < usedColumns; i++) {
		const RLEColumn &c = columns[i];
		if (c.runCount == 0) { continue; }
		const int64_t off = c.storageOffset;
		if (off < 0 || off + c.runCount + 2 > elementCount) {
			return fail("LOD " + std::to_string(lod) + " column " + std::to_string(i) + ": element range outside the pool");
		}
		if (elements[off].Length != 0 || elements[off].ColorsIndex != 0 || elements[off + c.runCount + 1].Length != 0 || elements[off + c.runCount + 1].ColorsIndex != 0) {
			return fail("LOD " + std::to_string(lod) + " column " + std::to_string(i) + ": missing element guards (World.cs:205-209)");
		}
		int64_t total = 0, colours = 0;
		for (int r = 0; r < c.runCount; r++) {
			const RLEElement e = elements[off + 1 + r];
			if (e.Length <= 0) {
				return fail("LOD " + std::to_string(lod) + " column " + std::to_string(i) + ": run with non-positive length");
			}
			total += e.Length;
			if (e.ColorsIndex >= 0 && (int64_t)e.ColorsIndex + e.Length > colours) { colours = (int64_t)e.ColorsIndex + e.Length; }
		}
		// total == height, the rule cvx_world_upload enforces as well (RLEColumnBuilder.ToFinalColumn always emits full-height columns,
		// WordBuilder.cs:232-258; the device walks ONE run table in both directions, which needs it): a blob this library loads is a
		// blob the GPU library accepts.
		if (total != maxY || off + c.runCount + 2 + colours > elementCount) {
			return fail("LOD " + std::to_string(lod) + " column " + std::to_string(i) + ": runs do not add up to the column height or colours exceed the pool");
		}
	}
	return true;
}

void World::StoreColumn(int index, const FinalColumn &column)
{
	RLEColumn *pointer = Columns() + index;
	if (pointer->runCount > 0) {
		throw std::runtime_error("column set twice"); // World.cs:155-157
	}
	int runCount = (int)column.runs.size();
	int solidCount = (int)column.colors.size();
	if (runCount <= 0 || runCount > 65535) {
		throw std::out_of_range("runCount"); // World.cs:193-195
	}
	int64_t allocationElementCount = (int64_t)runCount + solidCount + 2;
	int64_t oldCount = elementAllocationCount; // AllocateElements, World.cs:340-373
	if (oldCount + allocationElementCount > (int64_t)INT_MAX) {
		throw std::runtime_error("Only supports up to 2^31 elements"); // World.cs:355-357
	}
	elementAllocationCount += allocationElementCount;
	size_t headerBytes = (size_t)ColumnCount() * sizeof(RLEColumn);
	size_t needed = headerBytes + (size_t)elementAllocationCount * 4;
	if (storage.size() < needed) {
		if (storage.capacity() < needed) {
			storage.reserve(std::max(needed, storage.capacity() * 2));
		}
		storage.resize(needed, 0);
	}
	pointer = Columns() + index; // storage may have moved
	RLEElement *startPointer = reinterpret_cast<RLEElement *>(storage.data() + headerBytes) + oldCount;
	startPointer[0] = { 0, 0 }; // element guards, World.cs:205-209
	std::memcpy(startPointer + 1, column.runs.data(), (size_t)runCount * sizeof(RLEElement));
	startPointer[runCount + 1] = { 0, 0 };
	std::memcpy(startPointer + runCount + 2, column.colors.data(), (size_t)solidCount * sizeof(ColorARGB32));
	pointer->storageOffset = (int32_t)oldCount;
	pointer->runCount = (uint16_t)runCount;
	pointer->worldMin = (uint16_t)column.worldMin;
	pointer->worldMax = (uint16_t)column.worldMax;
}

// World.DownSamplePartial, World.cs:101-127
void World::DownSamplePartial(int x, int z, int extraLods, RLEColumnBuilder &columnBuilder) const
{
	const RLEColumn column = Columns()[GetIndexKnownInBounds(x, z)];
	if (column.runCount <= 0) {
		return;
	}
	int elementBoundsX = dimensions.y >> lod;
	int nextLod = lod + extraLods;
	const RLEElement *guardStart = Elements() + column.storageOffset;
	const ColorARGB32 *colorPointer = reinterpret_cast<const ColorARGB32 *>(guardStart) + column.runCount + 2;

	for (int run = 0; run < column.runCount; run++) {
		RLEElement element = guardStart[run + 1];
		elementBoundsX = elementBoundsX - element.Length;
		if (element.ColorsIndex < 0) {
			continue;
		}
		for (int i = 0; i < element.Length; i++) {
			int Y = elementBoundsX + i;
			int colorIdx = element.ColorsIndex + element.Length - i - 1;
			columnBuilder.SetVoxel(Y >> nextLod, colorPointer[colorIdx]);
		}
	}
}

namespace {

// Build `count` target columns with `make(i, builder)` in parallel blocks and
// store them sequentially in index order (deterministic element offsets; the
// reference's order depends on thread timing, World.cs:53, WordBuilder.cs:113).
template <typename MakeFn, typename IndexFn>
void BuildColumns(World &target, int64_t count, int voxelScale, int16_t topY, int threads, int64_t *voxelCount, MakeFn make, IndexFn indexOf)
{
	const int64_t kBlock = 1 << 16;
	std::vector<FinalColumn> block((size_t)std::min(count, kBlock));
	std::vector<uint8_t> present(block.size());
	int64_t total = 0;
#ifdef _OPENMP
	if (threads <= 0) { threads = DefaultThreads(); }
#else
	threads = 1;
#endif
	for (int64_t base = 0; base < count; base += kBlock) {
		int64_t n = std::min(kBlock, count - base);
		int64_t blockTotal = 0;
		std::string failure; // an exception must not leave the parallel region: the first one is re-thrown after it
#pragma omp parallel num_threads(threads) reduction(+ : blockTotal)
		{
			RLEColumnBuilder builder;
#pragma omp for schedule(dynamic, 256)
			for (int64_t k = 0; k < n; k++) {
				try {
					builder.Clear();
					make(base + k, builder);
					int64_t v = 0;
					present[(size_t)k] = builder.ToFinalColumn(voxelScale, topY, block[(size_t)k], v) ? 1 : 0;
					blockTotal += v;
				} catch (const std::exception &e) {
					present[(size_t)k] = 0;
#pragma omp critical(cvx_build_columns_failure)
					if (failure.empty()) { failure = e.what(); }
				}
			}
		}
		if (!failure.empty()) {
			throw std::runtime_error(failure);
		}
		total += blockTotal;
		for (int64_t k = 0; k < n; k++) {
			if (present[(size_t)k]) {
				target.StoreColumn(indexOf(base + k), block[(size_t)k]);
			}
		}
	}
	if (voxelCount) { *voxelCount = total; }
}

} // namespace

// World.DownSample + DownSampleColumn, World.cs:45-96
World World::DownSample(int extraLods, int64_t *voxelCount, int threads) const
{
	World subWorld(dimensions, lod + extraLods);
	const int step = 1 << subWorld.lod;
	const int stepSize = 1 << lod;
	const int steps = 1 << extraLods;
	const int nextVoxelCountY = (dimensions.y >> (lod + extraLods)) - 1;
	const int columnsZ = dimensions.z / step;
	const int64_t count = (int64_t)(dimensions.x / step) * columnsZ;
	const World *self = this;
	BuildColumns(
		subWorld, count, 1 << (lod + extraLods), (int16_t)nextVoxelCountY, threads, voxelCount,
		[=](int64_t k, RLEColumnBuilder &builder) {
			int xStart = (int)(k / columnsZ) * step;
			int zStart = (int)(k % columnsZ) * step;
			for (int ix = 0; ix < steps; ix++) {
				int x = xStart + ix * stepSize;
				for (int iz = 0; iz < steps; iz++) {
					int z = zStart + iz * stepSize;
					self->DownSamplePartial(x, z, extraLods, builder);
				}
			}
		},
		[&](int64_t k) { return subWorld.GetIndexKnownInBounds((int)(k / columnsZ) * step, (int)(k % columnsZ) * step); });
	return subWorld;
}

// ---------------------------------------------------------------------------
// WorldBuilder
// ---------------------------------------------------------------------------
WorldBuilder::WorldBuilder(int x, int y, int z)
{
	dimensions.x = x; dimensions.y = y; dimensions.z = z;
	if (((x - 1) & x) != 0 || ((z - 1) & z) != 0 || x <= 0 || z <= 0) {
		throw std::invalid_argument("Expected x/z to be powers of two"); // WordBuilder.cs:30-32
	}
	WorldColumns.resize((size_t)x * z);
}

World WorldBuilder::ToLOD0World(int64_t *voxelCount, int threads)
{
	World world(dimensions, 0);
	int16_t maxY = (int16_t)(dimensions.y - 1);
	std::vector<RLEColumnBuilder> &cols = WorldColumns;
	BuildColumns(
		world, (int64_t)cols.size(), 1, maxY, threads, voxelCount,
		[&](int64_t i, RLEColumnBuilder &builder) { std::swap(builder, cols[(size_t)i]); },
		[&](int64_t i) { return world.GetIndexKnownInBounds((int)(i / dimensions.z), (int)(i % dimensions.z)); });
	WorldColumns.clear();
	return world;
}

World WorldBuilder::ToLOD0World(int x, int y, int z, ColumnSource source, void *user, int64_t *voxelCount, int threads)
{
	int3 dims; dims.x = x; dims.y = y; dims.z = z;
	World world(dims, 0);
	BuildColumns(
		world, (int64_t)x * z, 1, (int16_t)(y - 1), threads, voxelCount,
		[=](int64_t i, RLEColumnBuilder &builder) { source(user, (int)(i / z), (int)(i % z), builder); },
		[&](int64_t i) { return world.GetIndexKnownInBounds((int)(i / z), (int)(i % z)); });
	return world;
}

// ---------------------------------------------------------------------------
// WorldSaveFile, WorldSaveFile.cs:8-104
// ---------------------------------------------------------------------------
namespace {
struct SaveHeader { // WorldSaveFile.cs:96-103
	int64_t EmptyBytes;
	int32_t DimensionX, DimensionY, DimensionZ;
	int32_t WorldCount;
};
static_assert(sizeof(SaveHeader) == 24, "header is 24 bytes");
} // namespace

bool SerializeWorlds(const std::vector<World> &worlds, const std::string &filePath, std::string *error)
{
	if (worlds.empty()) {
		if (error) { *error = "no worlds"; }
		return false;
	}
	FILE *f = std::fopen(filePath.c_str(), "wb");
	if (!f) {
		if (error) { *error = "cannot open " + filePath; }
		return false;
	}
	SaveHeader header{};
	header.DimensionX = worlds[0].Dimensions().x;
	header.DimensionY = worlds[0].Dimensions().y;
	header.DimensionZ = worlds[0].Dimensions().z;
	header.WorldCount = (int32_t)worlds.size();
	std::vector<int64_t> offsets(worlds.size() * 2);
	int64_t offsetToStartOfWorld = (int64_t)sizeof(SaveHeader) + (int64_t)offsets.size() * 8;
	for (size_t i = 0; i < worlds.size(); i++) {
		offsets[i * 2] = offsetToStartOfWorld;
		offsets[i * 2 + 1] = worlds[i].StorageByteLength();
		offsetToStartOfWorld += worlds[i].StorageByteLength();
	}
	bool ok = std::fwrite(&header, sizeof header, 1, f) == 1;
	ok = ok && std::fwrite(offsets.data(), 8, offsets.size(), f) == offsets.size();
	for (size_t i = 0; ok && i < worlds.size(); i++) {
		size_t n = (size_t)worlds[i].StorageByteLength();
		ok = std::fwrite(worlds[i].StoragePointer(), 1, n, f) == n;
	}
	std::fclose(f);
	if (!ok && error) { *error = "short write to " + filePath; }
	return ok;
}

bool DeserializeWorlds(const std::string &filePath, std::vector<World> &worlds, std::string *error)
{
	worlds.clear();
	FILE *f = std::fopen(filePath.c_str(), "rb");
	if (!f) {
		if (error) { *error = "cannot open " + filePath; }
		return false;
	}
	SaveHeader header{};
	std::string detail;
	int64_t fileSize = 0;
	if (std::fseek(f, 0, SEEK_END) == 0) { fileSize = (int64_t)std::ftell(f); }
	std::rewind(f);
	bool ok = std::fread(&header, sizeof header, 1, f) == 1 && header.WorldCount > 0 && header.WorldCount <= 64;
	std::vector<int64_t> offsets;
	if (ok) {
		offsets.resize((size_t)header.WorldCount * 2);
		ok = std::fread(offsets.data(), 8, offsets.size(), f) == offsets.size();
	}
	int3 dims; dims.x = header.DimensionX; dims.y = header.DimensionY; dims.z = header.DimensionZ;
	for (int i = 0; ok && i < header.WorldCount; i++) {
		int64_t offset = offsets[(size_t)i * 2];
		int64_t count = offsets[(size_t)i * 2 + 1];
		if (offset < 0 || count < 0 || count > fileSize || offset > fileSize - count) { // the table is untrusted input
			ok = false;
			break;
		}
		std::vector<uint8_t> blob((size_t)count);
		ok = std::fseek(f, (long)offset, SEEK_SET) == 0 && std::fread(blob.data(), 1, blob.size(), f) == blob.size();
		if (ok && !World::ValidateBlob(dims, i, blob.data(), count, &detail)) {
			ok = false;
		}
		if (ok) {
			worlds.emplace_back(dims, i, blob.data(), count);
		}
	}
	std::fclose(f);
	if (!ok) {
		worlds.clear();
		if (error) { *error = "malformed world file " + filePath + (detail.empty() ? "" : ": " + detail); }
	}
	return ok;
}

// ---------------------------------------------------------------------------
// Procedural benchmark world (ours).  Integer arithmetic only, so the world
// is identical on every machine for a given (dims, seed).
// ---------------------------------------------------------------------------
namespace {

inline uint32_t Hash32(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}
inline uint32_t Hash3(uint32_t a, uint32_t b, uint32_t c)
{
	return Hash32(a * 0x9E3779B1u ^ Hash32(b * 0x85EBCA77u ^ Hash32(c + 0xC2B2AE3Du)));
}

// 16.16 fixed-point value noise in [0, 65535], lattice spacing `cell`.
uint32_t ValueNoise(int x, int z, int cell, uint32_t seed)
{
	int cx = x / cell, cz = z / cell;
	uint64_t tx = (uint64_t)(x % cell) * 65536u / (uint32_t)cell;
	uint64_t tz = (uint64_t)(z % cell) * 65536u / (uint32_t)cell;
	uint64_t sx = (tx * tx * (3u * 65536u - 2u * tx)) >> 32; // smoothstep, 0..65536
	uint64_t sz = (tz * tz * (3u * 65536u - 2u * tz)) >> 32;
	int64_t v00 = Hash3((uint32_t)cx, (uint32_t)cz, seed) & 0xFFFF;
	int64_t v10 = Hash3((uint32_t)cx + 1, (uint32_t)cz, seed) & 0xFFFF;
	int64_t v01 = Hash3((uint32_t)cx, (uint32_t)cz + 1, seed) & 0xFFFF;
	int64_t v11 = Hash3((uint32_t)cx + 1, (uint32_t)cz + 1, seed) & 0xFFFF;
	int64_t a = v00 + (((v10 - v00) * (int64_t)sx) >> 16);
	int64_t b = v01 + (((v11 - v01) * (int64_t)sx) >> 16);
	return (uint32_t)(a + (((b - a) * (int64_t)sz) >> 16));
}

struct ProceduralContext {
	int dimX, dimY, dimZ;
	uint32_t seed;
	std::vector<uint16_t> height; // x-major
	int lo, hi;
};

inline int HeightAt(const ProceduralContext &c, int x, int z)
{
	if (x < 0 || z < 0 || x >= c.dimX || z >= c.dimZ) {
		return 0; // world edge: the shell reaches the ground, so the rim is a wall
	}
	return c.height[(size_t)x * c.dimZ + z];
}

inline uint8_t ClampByte(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

ColorARGB32 TerrainColor(const ProceduralContext &c, int x, int y, int z)
{
	// height ramp (grass -> rock -> grey -> snow), perturbed per voxel
	static const int ramp[5][3] = { { 52, 110, 48 }, { 96, 150, 70 }, { 128, 104, 80 }, { 150, 150, 152 }, { 240, 242, 250 } };
	int span = c.hi - c.lo;
	int t = (y - c.lo) * 1024 / (span > 0 ? span : 1); // 0..1024
	t = t < 0 ? 0 : (t > 1023 ? 1023 : t);
	int seg = t / 256, f = t % 256;
	uint32_t h = Hash3((uint32_t)x, (uint32_t)y * 2654435761u + (uint32_t)z, c.seed ^ 0xC0105EEDu);
	int jitter[3] = { (int)(h & 31) - 16, (int)((h >> 5) & 31) - 16, (int)((h >> 10) & 31) - 16 };
	ColorARGB32 col;
	col.a = 255;
	int rgb[3];
	for (int k = 0; k < 3; k++) {
		rgb[k] = ramp[seg][k] + ((ramp[seg + 1][k] - ramp[seg][k]) * f) / 256 + jitter[k];
	}
	col.r = ClampByte(rgb[0]); col.g = ClampByte(rgb[1]); col.b = ClampByte(rgb[2]);
	return col;
}

void ProceduralColumn(void *user, int x, int z, RLEColumnBuilder &out)
{
	const ProceduralContext &c = *static_cast<const ProceduralContext *>(user);
	int h = HeightAt(c, x, z);
	int lowest = h;
	lowest = std::min(lowest, HeightAt(c, x - 1, z));
	lowest = std::min(lowest, HeightAt(c, x + 1, z));
	lowest = std::min(lowest, HeightAt(c, x, z - 1));
	lowest = std::min(lowest, HeightAt(c, x, z + 1));
	int bottom = std::max(0, lowest - 1);
	for (int y = bottom; y <= h; y++) {
		out.SetVoxel(y, TerrainColor(c, x, y, z));
	}
	// floating slabs ("clouds"): second noise layer, top ~20 % of its range
	int maxDim = std::max(c.dimX, c.dimZ);
	uint32_t cloud = (ValueNoise(x, z, std::max(8, maxDim / 16), c.seed ^ 0xF10A7u) * 3u + ValueNoise(x, z, std::max(4, maxDim / 64), c.seed ^ 0xBEEFu)) / 4u;
	const uint32_t threshold = 40000u;
	if (cloud > threshold) {
		int base = c.dimY * 27 / 32 + (int)((cloud - threshold) >> 9);
		int thickness = 2 + (int)((cloud - threshold) >> 11);
		int top = std::min(c.dimY - 2, base + thickness);
		uint32_t hh = Hash3((uint32_t)x, (uint32_t)z, c.seed ^ 0x51AB5u);
		for (int y = std::max(base, h + 2); y <= top; y++) {
			int shade = 200 + (int)((hh >> (y & 15)) & 31);
			out.SetVoxel(y, ColorARGB32{ 255, ClampByte(shade), ClampByte(shade), ClampByte(shade + 20) });
		}
		if ((hh & 7u) == 0u) { // a second, thin run a little higher
			int y2 = std::min(c.dimY - 2, top + 3 + (int)((hh >> 8) & 7));
			if (y2 > top + 1) {
				out.SetVoxel(y2, ColorARGB32{ 255, 250, 210, 160 });
			}
		}
	}
}

} // namespace

std::vector<World> BuildProceduralWorld(int dimX, int dimY, int dimZ, uint32_t seed, int threads, int64_t *lod0Voxels)
{
	if (dimX <= 0 || dimY <= 0 || dimZ <= 0 || (dimX & (dimX - 1)) || (dimZ & (dimZ - 1)) || (dimY & (dimY - 1)) || dimY > 32768) {
		throw std::invalid_argument("procedural world dimensions must be powers of two, Y <= 32768");
	}
	ProceduralContext c;
	c.dimX = dimX; c.dimY = dimY; c.dimZ = dimZ; c.seed = seed;
	c.lo = dimY / 32;
	c.hi = dimY * 25 / 32;
	c.height.resize((size_t)dimX * dimZ);
	int maxDim = std::max(dimX, dimZ);
#ifdef _OPENMP
	if (threads <= 0) { threads = DefaultThreads(); }
#endif
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
	for (int x = 0; x < dimX; x++) {
		for (int z = 0; z < dimZ; z++) {
			uint64_t sum = 0;
			uint32_t weight = 16, total = 0;
			for (int o = 0; o < 5; o++) {
				int cell = std::max(2, (maxDim / 4) >> o);
				sum += (uint64_t)ValueNoise(x, z, cell, seed + (uint32_t)o * 0x1234567u) * weight;
				total += weight;
				weight >>= 1;
			}
			uint32_t n = (uint32_t)(sum / total); // 0..65535
			// mild shaping: emphasise valleys and peaks
			uint64_t shaped = ((uint64_t)n * n) >> 16;
			uint32_t mixed = (uint32_t)((n + shaped) / 2);
			c.height[(size_t)x * dimZ + z] = (uint16_t)(c.lo + (int)(((uint64_t)(c.hi - c.lo) * mixed) >> 16));
		}
	}

	std::vector<World> worlds;
	worlds.reserve(LOD_LEVELS);
	worlds.push_back(WorldBuilder::ToLOD0World(dimX, dimY, dimZ, &ProceduralColumn, &c, lod0Voxels, threads));
	for (int j = 1; j < LOD_LEVELS; j++) { // UnityManager.cs:328-331
		worlds.push_back(worlds[0].DownSample(j, nullptr, threads));
	}
	return worlds;
}

} // namespace cvx
