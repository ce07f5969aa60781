// cvx_shard.hip -- libcpuvox_gpu.so, multi-GPU side of the C ABI: the shard plan (which rank renders which 64-ray tile and
// where its pixel rows have to end up) and the RCCL exchange of the rendered tiles.  See include/cpuvox_gpu.h.
//
// The reference has one synchronisation point per frame, `render.Complete()` (RenderManager.cs:363); sharded over N GPUs
// the equivalent is "every rank draws its tiles (cvx_draw_segments_placed) + one exchange".  The plan is pure host
// arithmetic (testable without a GPU); RCCL is loaded on first use (dlopen), so a single-GPU host needs no librccl.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
// RCCL's own header types the entry points (a prototype that drifts is a compile error).  A host that has the runtime library but not the
// development headers still builds the product: the handful of declarations this file uses, as RCCL 2.x publishes them.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
const char *ncclGetErrorString(ncclResult_t result);
}
#endif

#include <cmath>
#include <chrono>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <new>
#include <vector>

#include "cvx_context.h"

using cvxi::Fail;

struct cvx_shard_plan {
	int rank = 0, worldSize = 1;
	int64_t tileCount = 0;
	std::vector<int64_t> sendStart, dispStart; // worldSize + 1 entries each, in 256-byte rows
	struct MyTile {
		int64_t index; // canonical tile index in the batch
		int area;      // 0 = send area, 1 = display area
		int section;   // destination rank (send) / rendering rank (display)
		int64_t row;   // row offset inside the section
		int omin;      // first pixel row of the tile that exists in the area
	};
	std::vector<MyTile> myTiles;
};

namespace {

// Mathf.RoundToInt (half to even) then clamp, RenderManager.cs:302-311 -- the same rule BuildFrame (cvx_gpu.hip) applies
int RoundClamp(float v, int lo, int hi)
{
	float r = std::nearbyint(v);
	int i = (r != r || r >= 2147483648.0f || r < -2147483648.0f) ? (int)0x80000000 : (int)r;
	return i < lo ? lo : (i > hi ? hi : i);
}

// ---- RCCL, resolved at run time -------------------------------------------------------------------------------------
// librccl is dlopen()ed on first use (a single-GPU host needs none); the function-pointer types come from RCCL's own header,
// so a prototype that drifts from the installed library is a compile error, not a silent ABI mismatch.
typedef decltype(&ncclGetUniqueId) FnGetUniqueId;
typedef decltype(&ncclCommInitRank) FnCommInitRank;
typedef decltype(&ncclCommDestroy) FnCommDestroy;
typedef decltype(&ncclCommAbort) FnCommAbort;
typedef decltype(&ncclGroupStart) FnGroup;
typedef decltype(&ncclSend) FnSend;
typedef decltype(&ncclRecv) FnRecv;
typedef decltype(&ncclGetErrorString) FnGetErrorString;
static_assert(sizeof(ncclUniqueId) == 128, "cvx_comm_unique_id hands out 128 bytes");

struct Rccl {
	void *handle = nullptr;
	FnGetUniqueId getUniqueId = nullptr;
	FnCommInitRank commInitRank = nullptr;
	FnCommDestroy commDestroy = nullptr;
	FnCommAbort commAbort = nullptr;
	FnGroup groupStart = nullptr, groupEnd = nullptr;
	FnSend send = nullptr;
	FnRecv recv = nullptr;
	FnGetErrorString errorString = nullptr;
	bool ok = false;
};

Rccl &LoadRccl()
{
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		for (const char *name : { "librccl.so.1", "librccl.so" }) {
			r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (r.handle) { break; }
		}
		if (!r.handle) { return; }
		r.getUniqueId = (FnGetUniqueId)dlsym(r.handle, "ncclGetUniqueId");
		r.commInitRank = (FnCommInitRank)dlsym(r.handle, "ncclCommInitRank");
		r.commDestroy = (FnCommDestroy)dlsym(r.handle, "ncclCommDestroy");
		r.commAbort = (FnCommAbort)dlsym(r.handle, "ncclCommAbort");
		r.groupStart = (FnGroup)dlsym(r.handle, "ncclGroupStart");
		r.groupEnd = (FnGroup)dlsym(r.handle, "ncclGroupEnd");
		r.send = (FnSend)dlsym(r.handle, "ncclSend");
		r.recv = (FnRecv)dlsym(r.handle, "ncclRecv");
		r.errorString = (FnGetErrorString)dlsym(r.handle, "ncclGetErrorString");
		r.ok = r.getUniqueId && r.commInitRank && r.commDestroy && r.groupStart && r.groupEnd && r.send && r.recv;
	});
	return r;
}

int NcclFail(cvx_context *ctx, Rccl &r, const char *what, ncclResult_t rc)
{
	return Fail(ctx, CVX_ERR_HIP, "%s failed: %s", what, r.errorString ? r.errorString(rc) : "RCCL error");
}

} // namespace

extern "C" {

int cvx_shard_plan_create(int frameCount, const cvx_segment_data *segments, const float *vanishingPoints, int screenWidth, int screenHeight,
                          int rank, int worldSize, cvx_shard_plan **out)
{
	if (!out) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "out is NULL"); }
	*out = nullptr;
	if (frameCount <= 0 || !segments || !vanishingPoints || screenWidth <= 0 || screenHeight <= 0 || worldSize < 1 || rank < 0 || rank >= worldSize) {
		return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "bad shard plan arguments");
	}
	// (the vectors below grow: nothing may escape across the C boundary, and a half-built plan is not leaked)
	std::unique_ptr<cvx_shard_plan> p(new (std::nothrow) cvx_shard_plan());
	if (!p) { return Fail(nullptr, CVX_ERR_HIP, "out of host memory"); }
	try {
	const int N = worldSize, W = screenWidth, H = screenHeight;
	p->rank = rank;
	p->worldSize = N;
	std::vector<int64_t> sendRows((size_t)N, 0), dispRows((size_t)N, 0);
	int64_t index = 0;
	for (int b = 0; b < frameCount; b++) {
		const cvx_segment_data *seg = segments + (size_t)b * 4;
		const float *vp = vanishingPoints + (size_t)b * 2;
		const int vx = RoundClamp(vp[0], 0, W - 1), vy = RoundClamp(vp[1], 0, H - 1);
		// originalNextFreePixelMin / Max of the four segments, RenderManager.cs:298-316
		const int lo[4] = { vy, 0, vx, 0 }, hi[4] = { H - 1, vy, W - 1, vx };
		const int root = b % N; // display rank of the frame
		int t = 0;              // tile index inside the frame (segment-major: the order DrawBatch numbers them)
		for (int s = 0; s < 4; s++) {
			const int rays = seg[s].RayCount > 0 ? seg[s].RayCount : 0;
			const int tiles = (rays + CVX_WAVE - 1) / CVX_WAVE;
			const int64_t rows = hi[s] - lo[s] + 1;
			for (int k = 0; k < tiles; k++, t++, index++) {
				const int owner = t % N; // rendering rank
				if (root == rank) {
					if (owner == rank) { p->myTiles.push_back({ index, 1, rank, dispRows[(size_t)rank], lo[s] }); }
					dispRows[(size_t)owner] += rows;
				} else if (owner == rank) {
					p->myTiles.push_back({ index, 0, root, sendRows[(size_t)root], lo[s] });
					sendRows[(size_t)root] += rows;
				}
			}
		}
	}
	p->tileCount = index;
	p->sendStart.assign((size_t)N + 1, 0);
	p->dispStart.assign((size_t)N + 1, 0);
	for (int i = 0; i < N; i++) {
		p->sendStart[(size_t)i + 1] = p->sendStart[(size_t)i] + sendRows[(size_t)i];
		p->dispStart[(size_t)i + 1] = p->dispStart[(size_t)i] + dispRows[(size_t)i];
	}
	} catch (const std::exception &e) {
		return Fail(nullptr, CVX_ERR_HIP, "cvx_shard_plan_create: %s", e.what());
	}
	*out = p.release();
	return CVX_OK;
}

void cvx_shard_plan_destroy(cvx_shard_plan *plan) { delete plan; }

int64_t cvx_shard_plan_tile_count(const cvx_shard_plan *plan) { return plan ? plan->tileCount : 0; }

int cvx_shard_plan_sections(const cvx_shard_plan *plan, int64_t *sendStart, int64_t *dispStart)
{
	if (!plan || !sendStart || !dispStart) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	std::memcpy(sendStart, plan->sendStart.data(), plan->sendStart.size() * sizeof(int64_t));
	std::memcpy(dispStart, plan->dispStart.data(), plan->dispStart.size() * sizeof(int64_t));
	return CVX_OK;
}

int cvx_shard_plan_transfer(const cvx_shard_plan *plan, int peer, int64_t *sendRow, int64_t *sendRows, int64_t *recvRow, int64_t *recvRows)
{
	if (!plan || peer < 0 || peer >= plan->worldSize || !sendRow || !sendRows || !recvRow || !recvRows) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	const bool self = peer == plan->rank; // my own section of the display area is written by my kernel: nothing travels
	*sendRow = plan->sendStart[(size_t)peer];
	*sendRows = self ? 0 : plan->sendStart[(size_t)peer + 1] - plan->sendStart[(size_t)peer];
	*recvRow = plan->dispStart[(size_t)peer];
	*recvRows = self ? 0 : plan->dispStart[(size_t)peer + 1] - plan->dispStart[(size_t)peer];
	return CVX_OK;
}

int cvx_shard_plan_tile_out(const cvx_shard_plan *plan, void *sendBase, void *dispBase, uint64_t *tileOut)
{
	if (!plan || !tileOut) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "bad arguments"); }
	for (int64_t i = 0; i < plan->tileCount; i++) { tileOut[i] = 0; }
	const uint64_t base[2] = { (uint64_t)(uintptr_t)sendBase, (uint64_t)(uintptr_t)dispBase };
	for (const cvx_shard_plan::MyTile &t : plan->myTiles) {
		const int64_t start = t.area == 0 ? plan->sendStart[(size_t)t.section] : plan->dispStart[(size_t)t.section];
		// the address of pixel row 0: rows below omin do not exist in the area, cvx_draw_segments_placed never touches them
		tileOut[t.index] = base[t.area] + (uint64_t)((start + t.row - t.omin) * (int64_t)(CVX_WAVE * 4));
	}
	return CVX_OK;
}

int cvx_comm_unique_id(void *id128)
{
	if (!id128) { return Fail(nullptr, CVX_ERR_INVALID_ARGUMENT, "id128 is NULL"); }
	Rccl &r = LoadRccl();
	if (!r.ok) { return Fail(nullptr, CVX_ERR_NOT_READY, "librccl could not be loaded"); }
	ncclUniqueId id;
	const ncclResult_t rc = r.getUniqueId(&id);
	if (rc != ncclSuccess) { return NcclFail(nullptr, r, "ncclGetUniqueId", rc); }
	std::memcpy(id128, &id, sizeof id);
	return CVX_OK;
}

// ncclCommInitRank blocks until every rank of the clique has arrived; a peer that never comes (crashed, wrong id, no route)
// would leave the caller hanging for as long as its job scheduler allows.  The call therefore runs on a helper thread and
// is given up after `timeoutSeconds`: the caller gets CVX_ERR_TIMEOUT and is expected to end the process (the helper thread
// stays parked inside RCCL; there is no portable way to cancel it).
int cvx_comm_create_timeout(cvx_context *ctx, const void *id128, int rank, int worldSize, double timeoutSeconds, void **comm)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!id128 || !comm || worldSize < 1 || rank < 0 || rank >= worldSize || !(timeoutSeconds > 0.0)) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "bad communicator arguments"); }
	*comm = nullptr;
	Rccl &r = LoadRccl();
	if (!r.ok) { return Fail(ctx, CVX_ERR_NOT_READY, "librccl could not be loaded"); }
	struct Job {
		ncclUniqueId id;
		ncclComm_t comm = nullptr;
		std::promise<ncclResult_t> done;
	};
	auto job = std::make_shared<Job>();
	std::memcpy(&job->id, id128, sizeof job->id);
	std::future<ncclResult_t> fut = job->done.get_future();
	const int device = ctx->device;
	const FnCommInitRank init = r.commInitRank;
	try {
		std::thread([job, init, device, rank, worldSize] {
			ncclResult_t rc = ncclUnhandledCudaError;
			if (hipSetDevice(device) == hipSuccess) { rc = init(&job->comm, worldSize, job->id, rank); }
			job->done.set_value(rc);
		}).detach();
	} catch (const std::exception &e) {
		return Fail(ctx, CVX_ERR_HIP, "cvx_comm_create: %s", e.what());
	}
	if (fut.wait_for(std::chrono::duration<double>(timeoutSeconds)) != std::future_status::ready) {
		return Fail(ctx, CVX_ERR_TIMEOUT, "ncclCommInitRank(rank %d of %d) did not return within %.0f s: a peer is missing or unreachable", rank, worldSize, timeoutSeconds);
	}
	const ncclResult_t rc = fut.get();
	if (rc != ncclSuccess) { return NcclFail(ctx, r, "ncclCommInitRank", rc); }
	*comm = job->comm;
	return CVX_OK;
}

int cvx_comm_create(cvx_context *ctx, const void *id128, int rank, int worldSize, void **comm)
{
	return cvx_comm_create_timeout(ctx, id128, rank, worldSize, 180.0, comm);
}

int cvx_comm_destroy(void *comm)
{
	if (!comm) { return CVX_OK; }
	Rccl &r = LoadRccl();
	if (!r.ok) { return CVX_ERR_NOT_READY; }
	return r.commDestroy((ncclComm_t)comm) == ncclSuccess ? CVX_OK : CVX_ERR_HIP;
}

int cvx_exchange(cvx_context *ctx, const cvx_shard_plan *plan, void *comm, void *hipStream, void *sendBase, void *dispBase)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!plan) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "plan is NULL"); }
	const int N = plan->worldSize;
	if (N == 1) { return CVX_OK; } // nothing travels
	if (!comm || !sendBase || !dispBase) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "communicator / buffers missing"); }
	Rccl &r = LoadRccl();
	if (!r.ok) { return Fail(ctx, CVX_ERR_NOT_READY, "librccl could not be loaded"); }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t st = hipStream ? (hipStream_t)hipStream : ctx->stream;
	const size_t rowBytes = CVX_WAVE * 4;
	// One ncclSend + one ncclRecv per peer, grouped: every pair rides its own xGMI link, no ring; the received rows land in
	// the display area exactly where the blit / read-back expects them (the peer's send section for me has my layout).
	ncclResult_t rc = r.groupStart();
	if (rc != ncclSuccess) { return NcclFail(ctx, r, "ncclGroupStart", rc); }
	const ncclComm_t c = (ncclComm_t)comm;
	for (int peer = 0; peer < N && rc == ncclSuccess; peer++) {
		int64_t s0, sn, r0, rn;
		(void)cvx_shard_plan_transfer(plan, peer, &s0, &sn, &r0, &rn); // (0 rows for peer == rank)
		if (sn > 0) { rc = r.send(static_cast<uint8_t *>(sendBase) + (size_t)s0 * rowBytes, (size_t)sn * rowBytes, ncclInt8, peer, c, st); }
		if (rc == ncclSuccess && rn > 0) { rc = r.recv(static_cast<uint8_t *>(dispBase) + (size_t)r0 * rowBytes, (size_t)rn * rowBytes, ncclInt8, peer, c, st); }
	}
	const ncclResult_t rcEnd = r.groupEnd();
	if (rc != ncclSuccess) { return NcclFail(ctx, r, "ncclSend / ncclRecv", rc); }
	if (rcEnd != ncclSuccess) { return NcclFail(ctx, r, "ncclGroupEnd", rcEnd); }
	return CVX_OK;
}

int cvx_image_exchange(cvx_context *ctx, const cvx_image_plan *plan, void *comm, void *hipStream, void *sendStream, void *recvStream)
{
	if (!ctx) { return CVX_ERR_INVALID_ARGUMENT; }
	if (!plan) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "plan is NULL"); }
	int64_t total[2] = { 0, 0 };
	(void)cvx_image_plan_sizes(plan, nullptr, &total[0], &total[1], nullptr);
	if (total[0] == 0 && total[1] == 0) { return CVX_OK; } // one rank: nothing travels
	if (!comm || (total[0] > 0 && !sendStream) || (total[1] > 0 && !recvStream)) { return Fail(ctx, CVX_ERR_INVALID_ARGUMENT, "communicator / streams missing"); }
	Rccl &r = LoadRccl();
	if (!r.ok) { return Fail(ctx, CVX_ERR_NOT_READY, "librccl could not be loaded"); }
	CVX_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t st = hipStream ? (hipStream_t)hipStream : ctx->stream;
	ncclResult_t rc = r.groupStart();
	if (rc != ncclSuccess) { return NcclFail(ctx, r, "ncclGroupStart", rc); }
	const ncclComm_t c = (ncclComm_t)comm;
	for (int peer = 0; rc == ncclSuccess; peer++) {
		int64_t s0, sn, r0, rn;
		if (cvx_image_plan_transfer(plan, peer, &s0, &sn, &r0, &rn) != CVX_OK) { break; } // past the last rank
		if (sn > 0) { rc = r.send(static_cast<uint8_t *>(sendStream) + (size_t)s0 * 4, (size_t)sn * 4, ncclInt8, peer, c, st); }
		if (rc == ncclSuccess && rn > 0) { rc = r.recv(static_cast<uint8_t *>(recvStream) + (size_t)r0 * 4, (size_t)rn * 4, ncclInt8, peer, c, st); }
	}
	const ncclResult_t rcEnd = r.groupEnd();
	if (rc != ncclSuccess) { return NcclFail(ctx, r, "ncclSend / ncclRecv", rc); }
	if (rcEnd != ncclSuccess) { return NcclFail(ctx, r, "ncclGroupEnd", rcEnd); }
	return CVX_OK;
}

} // extern "C"
