// smm_dist.hip -- the SpMV + Krylov loop row-partitioned over the GPUs of one node, behind the C ABI.
//
// One process per GPU.  Rank g owns a contiguous range of rows of A (values / positions / start exactly as in the reference's
// CSRMatrix, ref:1243-1259, with GLOBAL column numbers) and the matching slices of every vector.  The reference has no
// counterpart (single process, shared memory: SURVEY.md section 2.1); what is kept is the arithmetic of ref:2191-2283 (BiCGStab)
// and ref:2316-2398 (ConjugateGradient): the same update expressions in the same order, with the dot products completed across
// ranks.
//
//   * halo: a rank needs x only on the column range its rows touch.  The parts owned by other ranks are received point to point
//     (RCCL ncclSend / ncclRecv over xGMI) straight into a halo-extended vector [left halo | owned | right halo].
//   * the local rows are split ON THE DEVICE into A_loc (owned columns) and A_rem (halo columns): A_loc x runs on the caller's
//     stream while the exchange is in flight on the communicator's side stream; A_rem x is added in place when it has landed,
//     with the dot products of the freshly computed vector fused into that launch and completed by its last workgroup
//     (lastBlockSums, smm_device.h); the sums an update kernel leaves behind are added by a one-workgroup launch (distFinishSums).
//   * the 1-2 scalars of each reduction point are all-reduced in place (ncclAllReduce) on the side stream; every workgroup of
//     the next update kernel reads the all-reduced totals and forms alpha / omega / beta itself, so there are no scalar launches:
//     8 kernels per BiCGStab iteration (4 SpMV launches, s, r, x, p updates) + 1 finishing launch, 6 + 1 per CG iteration.  The x update does not depend
//     on the last all-reduce of an iteration (||r||^2, r.r0) and runs beside it.
//   * nothing in the loop synchronises with the host; the `done` flag is polled through a pinned mailbox (DonePoller).
//   * r05 / r06: between processes the halo and the scalars travel PEER TO PEER by default (smm_p2p.h: pushes into IPC-mapped landing areas with
//     relay ranks, slot reductions -- no collective inside the loop), and a rank's SpMV runs as ONE launch over A_loc and A_rem
//     (smm_spmv_split.hip: local half, a bounded wait for the word the land kernel raises, remote half).  The collectives above are the
//     fall-back (SMM_HIP_P2P=0, a failed self-test, ranks that are threads of one process).
//
// Communicators: RCCL (librccl resolved with dlopen at run time, so single-GPU users need no RCCL), or host callbacks (the
// caller moves the bytes: used by the tests to run several ranks on one GPU, where RCCL cannot, and by the gloo rehearsal of
// bench.py), or none (world size 1).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <thread>

#include <rccl/rccl.h>  // types and prototypes only: every call goes through dlsym'd pointers
#include <rocprim/device/device_scan.hpp>

#include <unistd.h>

#include "smm_device.h"
#include "smm_internal.h"
#include "smm_p2p.h"
#include "smm_solver_scal.h"

namespace smm {

constexpr int TPB = 256;

// ---------------------------------------------------------------------------------------------------------
// RCCL through dlopen
// ---------------------------------------------------------------------------------------------------------
struct RcclApi {
	void* handle = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	// optional (older libraries): resolved when present
	decltype(&ncclCommAbort) CommAbort = nullptr;
	decltype(&ncclCommCount) CommCount = nullptr;
};

static RcclApi* rccl() {
	static RcclApi api;
	static std::mutex mu;
	std::lock_guard<std::mutex> lock(mu);
	if (api.handle) return &api;
	// the copy PyTorch already mapped (same SONAME) wins, so one RCCL serves both; otherwise the system ROCm one
	const char* candidates[] = {getenv("SMM_HIP_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
	for (const char* c : candidates) {
		if (h) break;
		if (c && *c) h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
	}
	if (!h) {
		setError("RCCL not found (dlopen librccl.so.1: %s); set SMM_HIP_RCCL_PATH", dlerror());
		return nullptr;
	}
#define SMM_RCCL_SYM(NAME)                                                        \
	api.NAME = reinterpret_cast<decltype(api.NAME)>(dlsym(h, "nccl" #NAME));      \
	if (!api.NAME) {                                                              \
		setError("RCCL symbol nccl" #NAME " missing");                           \
		return nullptr;                                                           \
	}
	SMM_RCCL_SYM(GetUniqueId)
	SMM_RCCL_SYM(CommInitRank)
	SMM_RCCL_SYM(CommDestroy)
	SMM_RCCL_SYM(AllReduce)
	SMM_RCCL_SYM(Send)
	SMM_RCCL_SYM(Recv)
	SMM_RCCL_SYM(GroupStart)
	SMM_RCCL_SYM(GroupEnd)
	SMM_RCCL_SYM(GetErrorString)
#undef SMM_RCCL_SYM
	api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(h, "ncclCommAbort"));
	api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(h, "ncclCommCount"));
	api.handle = h;
	return &api;
}

static int rcclFail(ncclResult_t r, const char* what) {
	RcclApi* api = rccl();
	setError("RCCL error %d (%s) in %s", static_cast<int>(r), api && api->GetErrorString ? api->GetErrorString(r) : "?", what);
	return SMM_HIP_ERR_COMM;
}
#define SMM_RCCL_TRY(expr)                                  \
	do {                                                    \
		ncclResult_t _r = (expr);                           \
		if (_r != ncclSuccess) return rcclFail(_r, #expr);  \
	} while (0)

struct Seg {  // a contiguous piece of a halo-extended vector exchanged with one peer
	int peer, offset, count;
};

}  // namespace smm

struct smm_hip_comm {
	int rank = 0, world = 1, kind = SMM_COMM_SELF;
	ncclComm_t nccl = nullptr;
	smm_hip_host_allreduce_fn hostAllreduce = nullptr;
	smm_hip_host_sendrecv_fn hostSendrecv = nullptr;
	void* user = nullptr;
	hipStream_t stream = nullptr;  // side stream of the collectives (high priority: a pending exchange must get its few workgroups
	                               // before the persistent SpMV grid of the caller's stream fills every CU)
	std::vector<hipEvent_t> events;
	size_t nextEvent = 0;
	char* pinned = nullptr;  // host staging of the callback kind
	size_t pinnedBytes = 0;
	long long* d_i64 = nullptr;  // set-up reductions
	size_t i64Count = 0;
	bool broken = false;  // aborted after a failure or a time-out: every further call fails at once
};

namespace smm {

static hipEvent_t takeEvent(smm_hip_comm* c) {
	if (c->events.empty()) {
		// between two drains of the pipeline (every CHECK_EVERY = 16 iterations) a BiCGStab loop records ~10 events per iteration; an event
		// re-recorded while an earlier wait on it is still queued is legal (a wait refers to the record that preceded it), but the pool is
		// sized so that it does not happen
		c->events.resize(256);
		for (auto& e : c->events) {
			if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
		}
	}
	hipEvent_t e = c->events[c->nextEvent];
	c->nextEvent = (c->nextEvent + 1) % c->events.size();
	return e;
}

// ---- failure containment (a collective that a peer never joins would otherwise block this rank for ever) ---------------------
// Every wait of the row-partitioned code on a stream that may carry RCCL work goes through boundedSync: the stream is POLLED, and
// after SMM_HIP_COMM_TIMEOUT_S seconds (default 180) the communicator is aborted (ncclCommAbort: queued collectives of this rank are
// torn down instead of waiting for a peer that is gone) and the call fails with SMM_HIP_ERR_COMM.  Any other failure inside a
// distributed call aborts the communicator too (guardComm), so that the peers run into THEIR bounded wait instead of hanging in the
// next collective.  The process is expected to exit then (bench.py does, with a non-zero status).
static double commTimeoutSeconds() {
	static const double t = [] {
		const char* env = getenv("SMM_HIP_COMM_TIMEOUT_S");
		const double v = env ? atof(env) : 180.0;
		return v > 0 ? v : 180.0;
	}();
	return t;
}

static void commAbort(smm_hip_comm* c) {
	if (!c || c->broken) return;
	c->broken = true;
	if (c->kind == SMM_COMM_RCCL && c->nccl) {
		RcclApi* api = rccl();
		if (api && api->CommAbort) api->CommAbort(c->nccl);
		c->nccl = nullptr;  // (without ncclCommAbort the handle is leaked rather than destroyed: ncclCommDestroy would wait for the peers)
	}
}

static int boundedSync(smm_hip_comm* c, hipStream_t s) {
	if (!c || c->kind != SMM_COMM_RCCL) {
		SMM_HIP_TRY(hipStreamSynchronize(s));
		return SMM_HIP_OK;
	}
	const auto t0 = std::chrono::steady_clock::now();
	for (unsigned spins = 0;; ++spins) {
		const hipError_t q = hipStreamQuery(s);
		if (q == hipSuccess) return SMM_HIP_OK;
		if (q != hipErrorNotReady) {
			commAbort(c);
			return hipFail(q, "hipStreamQuery", __FILE__, __LINE__);
		}
		if (spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));  // (the first polls spin: short waits stay short)
		if ((spins & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > commTimeoutSeconds()) {
			commAbort(c);
			setError("comm: rank %d waited %.0f s for a collective to complete (a peer missing or failed?); communicator aborted", c->rank,
			         commTimeoutSeconds());
			return SMM_HIP_ERR_COMM;
		}
	}
}

// Only failures that can leave the ranks out of step tear the communicator down: a HIP or RCCL failure somewhere inside the call.  A
// caller's mistake that is found before anything was enqueued (SMM_HIP_ERR_INVALID: a wrong dtype, a null vector, a foreign
// preconditioner; SMM_HIP_ERR_PRECOND) is returned as it is and the communicator stays usable -- every rank makes the same call with
// the same kind of arguments, so every rank gets the same refusal.
static int guardComm(smm_hip_comm* c, int rc) {
	if ((rc == SMM_HIP_ERR_HIP || rc == SMM_HIP_ERR_COMM || rc == SMM_HIP_ERR_NOMEM) && c && c->kind == SMM_COMM_RCCL) commAbort(c);
	return rc;
}

static int commUsable(const smm_hip_comm* c) {
	if (c && c->broken) {
		setError("comm: the communicator was aborted after an earlier failure");
		return SMM_HIP_ERR_COMM;
	}
	return SMM_HIP_OK;
}

// everything enqueued on `from` so far happens before whatever is enqueued on `to` from now on
static int orderAfter(smm_hip_comm* c, hipStream_t from, hipStream_t to) {
	if (from == to) return SMM_HIP_OK;
	hipEvent_t e = takeEvent(c);
	if (!e) {
		setError("comm: event pool exhausted");
		return SMM_HIP_ERR_HIP;
	}
	SMM_HIP_TRY(hipEventRecord(e, from));
	SMM_HIP_TRY(hipStreamWaitEvent(to, e, 0));
	return SMM_HIP_OK;
}

static int ensurePinned(smm_hip_comm* c, size_t bytes) {
	if (c->pinnedBytes >= bytes) return SMM_HIP_OK;
	if (c->pinned) hipHostFree(c->pinned);
	c->pinned = nullptr;
	c->pinnedBytes = 0;
	SMM_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->pinned), bytes, hipHostMallocDefault));
	c->pinnedBytes = bytes;
	return SMM_HIP_OK;
}

template <typename T>
static ncclDataType_t ncclTypeOf() { return sizeof(T) == 4 ? ncclFloat32 : ncclFloat64; }

// in-place sum of d_buf[0..count) over the ranks, enqueued on `s` (the callback kind blocks until `s` has drained)
template <typename T>
static int commAllreduce(smm_hip_comm* c, T* d_buf, int count, hipStream_t s) {
	if (c->kind == SMM_COMM_SELF) return SMM_HIP_OK;
	if (c->kind == SMM_COMM_RCCL) {
		SMM_RCCL_TRY(rccl()->AllReduce(d_buf, d_buf, static_cast<size_t>(count), ncclTypeOf<T>(), ncclSum, c->nccl, s));
		return SMM_HIP_OK;
	}
	SMM_TRY(ensurePinned(c, count * sizeof(T)));
	SMM_HIP_TRY(hipMemcpyAsync(c->pinned, d_buf, count * sizeof(T), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	if (c->hostAllreduce(c->user, c->pinned, count, dtypeOf<T>()) != 0) {
		setError("comm: host all-reduce callback failed");
		return SMM_HIP_ERR_COMM;
	}
	SMM_HIP_TRY(hipMemcpyAsync(d_buf, c->pinned, count * sizeof(T), hipMemcpyHostToDevice, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));  // the staging buffer is reused by the next call
	return SMM_HIP_OK;
}

// blocking sum of host 64-bit integers (set-up only)
static int commAllreduceI64(smm_hip_comm* c, long long* h, int count) {
	if (c->kind == SMM_COMM_SELF) return SMM_HIP_OK;
	if (c->kind == SMM_COMM_HOST) {
		if (c->hostAllreduce(c->user, h, count, SMM_DTYPE_I64) != 0) {
			setError("comm: host all-reduce callback failed");
			return SMM_HIP_ERR_COMM;
		}
		return SMM_HIP_OK;
	}
	if (c->i64Count < static_cast<size_t>(count)) {
		if (c->d_i64) hipFree(c->d_i64);
		c->d_i64 = nullptr;
		SMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_i64), count * sizeof(long long)));
		c->i64Count = count;
	}
	SMM_HIP_TRY(hipMemcpyAsync(c->d_i64, h, count * sizeof(long long), hipMemcpyHostToDevice, c->stream));
	SMM_RCCL_TRY(rccl()->AllReduce(c->d_i64, c->d_i64, static_cast<size_t>(count), ncclInt64, ncclSum, c->nccl, c->stream));
	SMM_HIP_TRY(hipMemcpyAsync(h, c->d_i64, count * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
	SMM_TRY(boundedSync(c, c->stream));
	return SMM_HIP_OK;
}

// halo exchange of the halo-extended vector `ext`, enqueued on `s`
template <typename T>
static int commExchange(smm_hip_comm* c, T* ext, const std::vector<Seg>& sends, const std::vector<Seg>& recvs, hipStream_t s) {
	if (sends.empty() && recvs.empty()) return SMM_HIP_OK;
	if (c->kind == SMM_COMM_SELF) {
		setError("comm: a single-rank communicator cannot exchange halos");
		return SMM_HIP_ERR_INVALID;
	}
	if (c->kind == SMM_COMM_RCCL) {
		RcclApi* api = rccl();
		SMM_RCCL_TRY(api->GroupStart());
		for (const Seg& g : recvs) SMM_RCCL_TRY(api->Recv(ext + g.offset, static_cast<size_t>(g.count), ncclTypeOf<T>(), g.peer, c->nccl, s));
		for (const Seg& g : sends) SMM_RCCL_TRY(api->Send(ext + g.offset, static_cast<size_t>(g.count), ncclTypeOf<T>(), g.peer, c->nccl, s));
		SMM_RCCL_TRY(api->GroupEnd());
		return SMM_HIP_OK;
	}
	size_t total = 0;
	for (const Seg& g : sends) total += g.count * sizeof(T);
	for (const Seg& g : recvs) total += g.count * sizeof(T);
	SMM_TRY(ensurePinned(c, total));
	std::vector<int> sp, rp;
	std::vector<void*> sb, rb;
	std::vector<size_t> sn, rn;
	char* at = c->pinned;
	for (const Seg& g : sends) {
		SMM_HIP_TRY(hipMemcpyAsync(at, ext + g.offset, g.count * sizeof(T), hipMemcpyDeviceToHost, s));
		sp.push_back(g.peer);
		sb.push_back(at);
		sn.push_back(g.count * sizeof(T));
		at += g.count * sizeof(T);
	}
	char* recvBase = at;
	for (const Seg& g : recvs) {
		rp.push_back(g.peer);
		rb.push_back(at);
		rn.push_back(g.count * sizeof(T));
		at += g.count * sizeof(T);
	}
	SMM_HIP_TRY(hipStreamSynchronize(s));
	if (c->hostSendrecv(c->user, static_cast<int>(sp.size()), sp.data(), sb.data(), sn.data(), static_cast<int>(rp.size()), rp.data(), rb.data(), rn.data()) != 0) {
		setError("comm: host send/recv callback failed");
		return SMM_HIP_ERR_COMM;
	}
	at = recvBase;
	for (const Seg& g : recvs) {
		SMM_HIP_TRY(hipMemcpyAsync(ext + g.offset, at, g.count * sizeof(T), hipMemcpyHostToDevice, s));
		at += g.count * sizeof(T);
	}
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// splitting the local rows into A_loc / A_rem on the device
// ---------------------------------------------------------------------------------------------------------
__global__ void colRangeKernel(long long nnz, const int* __restrict__ positions, int* __restrict__ minmax) {
	int lo = 0x7fffffff, hi = -1;
	for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < nnz; i += static_cast<long long>(gridDim.x) * blockDim.x) {
		const int c = positions[i];
		lo = min(lo, c);
		hi = max(hi, c);
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		lo = min(lo, __shfl_xor(lo, o, WAVE));
		hi = max(hi, __shfl_xor(hi, o, WAVE));
	}
	if ((threadIdx.x & 63) == 0 && hi >= 0) {
		atomicMin(minmax, lo);
		atomicMax(minmax + 1, hi);
	}
}

// counts[row] = entries of the row with an owned column; counts[nLocal + 1 + row] = the others.  Both arrays have nLocal + 1
// slots so that an exclusive scan over nLocal + 1 elements ends with the total.
// labWindow (measurements with ONE rank, smm_hip_dist_csr::labWindow).  > 0: an owned column farther than that from its row counts as remote too
// (every row gets a remote part: a rank of MANY); < 0: the last |labWindow| columns of the range count as remote (only rows near that end get a
// remote part: a rank of a FEW)
__device__ __forceinline__ bool labLocal(int c, int row, int ownLo, int ownHi, int labWindow) {
	if (labWindow == 0) return true;
	if (labWindow > 0) return abs(c - row) < labWindow;
	return c < ownHi + labWindow;  // (the last |labWindow| columns are "the neighbour's": the first rank of two)
}

__global__ void splitCountKernel(int nLocal, const int* __restrict__ start, const int* __restrict__ positions, int ownLo, int ownHi, int labWindow,
                                 int* __restrict__ cntLoc, int* __restrict__ cntRem) {
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row <= nLocal; row += gridDim.x * blockDim.x) {
		int nl = 0, nr = 0;
		if (row < nLocal) {
			const int e = start[row + 1];
			for (int k = start[row]; k < e; ++k) {
				const int c = positions[k];
				if (c >= ownLo && c < ownHi && labLocal(c, ownLo + row, ownLo, ownHi, labWindow)) ++nl;
				else ++nr;
			}
		}
		cntLoc[row] = nl;
		cntRem[row] = nr;
	}
}

constexpr int THIN_SHARE = 8;  // a remote block is thin when rows with a remote entry x THIN_SHARE <= rows
// rows that hold a remote entry: a 0 / 1 flag per row (scanned into the row's index in the list), then the list
__global__ void thinFlagKernel(int nLocal, const int* __restrict__ startRem, int* __restrict__ flag) {
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row <= nLocal; row += gridDim.x * blockDim.x) {
		flag[row] = row < nLocal && startRem[row + 1] > startRem[row] ? 1 : 0;
	}
}
__global__ void thinListKernel(int nLocal, const int* __restrict__ startRem, const int* __restrict__ index, int* __restrict__ rows) {
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nLocal; row += gridDim.x * blockDim.x) {
		if (startRem[row + 1] > startRem[row]) rows[index[row]] = row;
	}
}

template <typename T>
__global__ void splitScatterKernel(int nLocal, const int* __restrict__ start, const int* __restrict__ positions, const T* __restrict__ values, int ownLo,
                                   int ownHi, int labWindow, int cmin, const int* __restrict__ startLoc, const int* __restrict__ startRem, int* __restrict__ posLoc,
                                   T* __restrict__ valLoc, int* __restrict__ posRem, T* __restrict__ valRem) {
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nLocal; row += gridDim.x * blockDim.x) {
		int il = startLoc[row], ir = startRem[row];
		const int e = start[row + 1];
		for (int k = start[row]; k < e; ++k) {  // order inside a row is preserved (columns stay ascending, ref:1247-1249)
			const int c = positions[k];
			const T v = values[k];
			if (c >= ownLo && c < ownHi && labLocal(c, ownLo + row, ownLo, ownHi, labWindow)) {
				posLoc[il] = c - ownLo;
				valLoc[il] = v;
				++il;
			} else {
				posRem[ir] = c - cmin;
				valRem[ir] = v;
				++ir;
			}
		}
	}
}

// piece of the halo a column of A_rem (numbered from the first column of the halo-extended vector) falls into: segment [off, off + cnt)
// is cut at off + floor(k cnt / K), k = 0 .. K; at most MAX_CHUNK_SEGS segments (more: the matrix keeps one piece)
constexpr int MAX_CHUNK_SEGS = 16;
constexpr int MAX_HALO_CHUNKS = 4;
struct ChunkSegs {
	int n, K;
	int off[MAX_CHUNK_SEGS], cnt[MAX_CHUNK_SEGS];
};
__host__ __device__ inline int chunkBound(int cnt, int k, int K) { return static_cast<int>(static_cast<long long>(cnt) * k / K); }
__device__ __forceinline__ int chunkOfColumn(const ChunkSegs& g, int col) {
	for (int i = 0; i < g.n; ++i) {
		const int rel = col - g.off[i];
		if (rel >= 0 && rel < g.cnt[i]) {
			int k = static_cast<int>(static_cast<long long>(rel) * g.K / g.cnt[i]);
			while (k + 1 < g.K && rel >= chunkBound(g.cnt[i], k + 1, g.K)) ++k;  // (the floor of the inverse can land one piece early)
			while (k > 0 && rel < chunkBound(g.cnt[i], k, g.K)) --k;
			return k;
		}
	}
	return g.K - 1;  // (a column outside every received segment cannot occur: A_rem's columns are what the segments were made from)
}

// counts[k * (nLocal + 1) + row] = entries of A_rem's row in piece k
__global__ void chunkCountKernel(int nLocal, const int* __restrict__ start, const int* __restrict__ positions, ChunkSegs segs, int* __restrict__ counts) {
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row <= nLocal; row += gridDim.x * blockDim.x) {
		int n[MAX_HALO_CHUNKS] = {0, 0, 0, 0};
		if (row < nLocal) {
			const int e = start[row + 1];
			for (int k = start[row]; k < e; ++k) ++n[chunkOfColumn(segs, positions[k])];
		}
		for (int k = 0; k < segs.K; ++k) counts[static_cast<size_t>(k) * (nLocal + 1) + row] = n[k];
	}
}

struct ChunkOut {
	int* pos[MAX_HALO_CHUNKS];
	void* val[MAX_HALO_CHUNKS];
};
template <typename T>
__global__ void chunkScatterKernel(int nLocal, const int* __restrict__ start, const int* __restrict__ positions, const T* __restrict__ values, ChunkSegs segs,
                                   const int* __restrict__ starts, ChunkOut out) {
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nLocal; row += gridDim.x * blockDim.x) {
		int at[MAX_HALO_CHUNKS];
		for (int k = 0; k < segs.K; ++k) at[k] = starts[static_cast<size_t>(k) * (nLocal + 1) + row];
		const int e = start[row + 1];
		for (int i = start[row]; i < e; ++i) {  // order inside a row is preserved
			const int k = chunkOfColumn(segs, positions[i]);
			out.pos[k][at[k]] = positions[i];
			static_cast<T*>(out.val[k])[at[k]] = values[i];
			++at[k];
		}
	}
}

static int exclusiveScan(int* d_inout, int count, hipStream_t s) {
	size_t tempBytes = 0;
	SMM_HIP_TRY(rocprim::exclusive_scan(nullptr, tempBytes, d_inout, d_inout, 0, static_cast<size_t>(count), rocprim::plus<int>(), s));
	void* temp = nullptr;
	SMM_TRY(devAlloc(&temp, tempBytes ? tempBytes : 1));
	const hipError_t e = rocprim::exclusive_scan(temp, tempBytes, d_inout, d_inout, 0, static_cast<size_t>(count), rocprim::plus<int>(), s);
	if (e == hipSuccess) (void)hipStreamSynchronize(s);
	devFree(temp);
	SMM_HIP_TRY(e);
	return SMM_HIP_OK;
}

// recurrence state of the distributed loops
template <typename T>
struct DistScal {
	T rrPing[2];  // BiCGStab: rr0 / CG: ||r||^2, double-buffered by iteration parity (workgroup 0 writes the next while others read)
	T alpha, omega, res;
	int done, iters, status, pad;  // pad: `done` as distCgR found it (what the launches behind it test: workgroup 0 of one of them may set `done` while others still start)
	T alphaRing[LAZY_M];  // CG with the deferred x update (distCgLazyP): alpha of the last LAZY_M iterations
	int flushIter, pad2;  // CG with the direction formed inside the SpMV: the iteration whose SpMV launch found its predecessor converged (-1: none)
};

// up to four [lo, hi) runs of rows, passed to the update kernels by value
struct RowRanges {
	int n = 0;
	int lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
	long long rows() const {
		long long t = 0;
		for (int i = 0; i < n; ++i) t += hi[i] - lo[i];
		return t;
	}
};

struct P2PState {
	int world = 1, rank = 0;
	bool haloOn = true;      // false: the halo stays with the communicator's grouped send / receive, only the scalars go through the slots (the hybrid)
	int relays = 0;          // relay ranks per segment (0: the direct path only)
	double directShare = 1;  // share of a segment that takes the direct path
	void* block = nullptr;   // this rank's symmetric block (fine-grained device memory)
	size_t bytes = 0;
	std::vector<char*> peer;   // every rank's block as mapped here (peer[rank] == block)
	std::vector<bool> opened;  // mapped with hipIpcOpenMemHandle (to be closed)
	size_t landOff[P2P_KINDS] = {}, stageOff[P2P_KINDS] = {};  // byte offsets inside MY block
	long long landElems = 0, stageElems = 0;
	P2PJob* d_push = nullptr;
	int nPush = 0;
	P2PJob* d_fwd = nullptr;
	int nFwd = 0;
	P2PLandSeg* d_land = nullptr;
	int nLand = 0;
	unsigned* d_counters = nullptr;
	P2PPeers peers{};
	unsigned long long haloSeq[P2P_KINDS] = {}, redSeq[P2P_RED_POINTS] = {};
	long long ticks = 0;
	P2PHeader* hdr() const { return reinterpret_cast<P2PHeader*>(block); }
};

}  // namespace smm

struct smm_hip_dist_csr {
	smm_hip_comm* comm = nullptr;
	int dtype = 0;
	int nGlobal = 0, rowBegin = 0, rowEnd = 0, nLocal = 0;
	int cmin = 0, cmaxExcl = 0, extLen = 1, ownOffset = 0;
	long long nnzLoc = 0, nnzRem = 0;
	int haloElements = 0;
	smm_hip_csr* aLoc = nullptr;
	smm_hip_csr* aRem = nullptr;
	void* arrays[6] = {};  // startLoc, posLoc, valLoc, startRem, posRem, valRem (owned)
	std::vector<smm::Seg> sends, recvs;
	bool remEmpty = true;
	// the halo in `chunks` pieces (SMM_HIP_HALO_CHUNKS, default 1 = one exchange, one remote block): piece k of every segment travels in
	// exchange k, and A_rem is cut by columns into aRemK[k] = its entries in piece k of any received segment -- the part of A_rem that
	// needs only piece k runs while the later pieces are still in flight
	int chunks = 1;
	std::vector<std::vector<smm::Seg>> sendsK, recvsK;
	std::vector<smm_hip_csr*> aRemK;
	std::vector<long long> nnzRemK;  // entries of each piece (a piece without entries is not launched unless it carries the epilogue)
	std::vector<void*> chunkArrays;  // start / positions / values of the pieces (owned)
	// workspace of the solvers, kept across solves
	void *r = nullptr, *r0 = nullptr, *ap = nullptr, *as = nullptr, *scratch = nullptr;
	void *pExt = nullptr, *sExt = nullptr, *xExt = nullptr;
	void* lazyExt[smm::LAZY_M] = {};  // CG with the deferred x update: LAZY_M more halo-extended direction vectors (allocated by the first such solve)
	void* rExt = nullptr;  // CG with the direction formed inside the SpMV: the residual, halo-extended (its halo travels instead of the direction's)
	long long cgFused = 0;  // SpMVs of ConjugateGradient that formed the direction themselves (smm_hip_dist_csr_cg_fused)
	void *partsA = nullptr, *partsB = nullptr, *partsC = nullptr;  // finishing buffers (PARTS_LEN): totals are all-reduced in place
	void* sc = nullptr;
	// every rank's column range [cmin, cmaxExcl) and the row bounds (global knowledge: any rank can derive any rank's halo plan)
	std::vector<long long> needs;
	std::vector<int> bounds;
	// halo first (r05): the rows of an updated vector that some peer receives -- merged, clipped, aligned ranges in owned-local numbering --
	// are produced by a small launch of their own, the exchange is posted right behind it, and the bulk launch takes the complement
	smm::RowRanges boundary, bulk;
	bool haloFirst = false;
	// the exchange begun by distExchangeBegin and not yet consumed by distMatvecCompute
	struct Pending {
		bool active = false;
		int kind = 0;
		unsigned long long seq = 0;
		unsigned long long landSeq = 0;  // what splitSync[kind] will read once this exchange has landed (0: no word is raised for it)
		hipEvent_t landed[smm::MAX_HALO_CHUNKS] = {};
		int waitSlot[smm::MAX_HALO_CHUNKS] = {-1, -1, -1, -1};
		bool async = false;
	} pending;
	smm::P2PState* p2p = nullptr;  // peer-to-peer data movement (smm_p2p.h); null: collectives of the communicator
	// the SpMV in ONE launch (smm_spmv_split.hip): splitSync[kind] is raised -- on the stream the exchange ran on, behind it -- to the sequence
	// number of the exchange whose halo has landed in the vector of that kind; splitSync[P2P_KINDS] is the error word of an expired wait
	// (with the peer-to-peer transport the block's own error word is used instead: one word for the host to read); splitSync[P2P_KINDS + 1]
	// accumulates the ticks workgroup 0 of every one-launch SpMV waited for its word (smm_hip_dist_csr_split_wait)
	unsigned long long* splitSync = nullptr;
	unsigned long long landSeq[smm::P2P_KINDS] = {};
	long long matvecsSplit = 0, matvecsTwo = 0;  // how many SpMVs with a halo ran in one launch / in two (smm_hip_dist_csr_matvec_forms)
	int splitSumsLdsMax = 16384;  // SMM_HIP_SPLIT_SUMS_LDS at create time: bytes of LDS the one-launch SpMV's row sums may take (0: always through out[]; tests)
	bool splitForced = false;
	bool sharedGpu = false;  // two ranks of the communicator run on one card (seen in the peer-to-peer set-up's table)
	bool splitAllowed = true;  // SMM_HIP_SPLIT_SPMV=0 at create time: this matrix keeps the two launches (A/B measurements, the bit-equality tests)
	bool reducedInKernel = false;  // the SpMV just launched ran its reduction point itself (the one-launch form with the slots): allreduceTotals has nothing to do
	int labWindow = 0;  // measurements on ONE GPU (SMM_HIP_LAB_SELF_SPLIT, single-rank communicator only): entries |column - row| >= window count as "remote"
	// a THIN remote block (r06): at most an eighth of the rows hold a remote entry (the slabs of a 3-D grid: two planes) -- thinRows lists them, and
	// the second half of the SpMV is a launch over those rows only (thinRemoteKernel) instead of a pass over every row of `out`
	int* thinRows = nullptr;
	int nThin = 0;
	void* partsThin = nullptr;  // finishing buffer of that launch's share of the dot products
	bool totalsFinal = false;   // the SpMV just launched left the rank's TOTALS in the finishing buffer (the thin form): the slot kernel must not add partials again
	long long matvecsThin = 0;
};

namespace smm {

template <typename T>
static int distWorkspace(smm_hip_dist_csr* D) {
	if (D->r) return SMM_HIP_OK;
	const size_t vb = static_cast<size_t>(std::max(1, D->nLocal)) * sizeof(T);
	const size_t eb = static_cast<size_t>(std::max(1, D->extLen)) * sizeof(T);
	SMM_TRY(devAlloc(&D->r, vb));
	SMM_TRY(devAlloc(&D->r0, vb));
	SMM_TRY(devAlloc(&D->ap, vb));
	SMM_TRY(devAlloc(&D->as, vb));
	SMM_TRY(devAlloc(&D->pExt, eb));
	SMM_TRY(devAlloc(&D->sExt, eb));
	SMM_TRY(devAlloc(&D->xExt, eb));
	SMM_TRY(devAlloc(&D->partsA, PARTS_LEN * sizeof(T)));
	SMM_TRY(devAlloc(&D->partsB, PARTS_LEN * sizeof(T)));
	SMM_TRY(devAlloc(&D->partsC, PARTS_LEN * sizeof(T)));
	SMM_TRY(devAlloc(&D->sc, sizeof(DistScal<T>)));
	SMM_TRY(devAlloc(reinterpret_cast<void**>(&D->splitSync), (P2P_KINDS + 2) * sizeof(unsigned long long)));
	preloadSplitUnit();  // (its code object is built for the device now, not inside the first iteration)
	hipStream_t s = libStream();
	SMM_HIP_TRY(hipMemsetAsync(D->splitSync, 0, (P2P_KINDS + 2) * sizeof(unsigned long long), s));
	// halo slots outside every recv segment are never read by A_rem; zero keeps them finite.  Tickets start at 0.
	SMM_HIP_TRY(hipMemsetAsync(D->pExt, 0, eb, s));
	SMM_HIP_TRY(hipMemsetAsync(D->sExt, 0, eb, s));
	SMM_HIP_TRY(hipMemsetAsync(D->xExt, 0, eb, s));
	SMM_HIP_TRY(hipMemsetAsync(D->partsA, 0, PARTS_LEN * sizeof(T), s));
	SMM_HIP_TRY(hipMemsetAsync(D->partsB, 0, PARTS_LEN * sizeof(T), s));
	SMM_HIP_TRY(hipMemsetAsync(D->partsC, 0, PARTS_LEN * sizeof(T), s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// halo first: which rows of an updated vector some peer receives
// ---------------------------------------------------------------------------------------------------------
// The send segments in owned-local numbering, merged, clipped and widened to multiples of 16 bytes (the update kernels' fast path needs
// 16-byte aligned runs; a few extra rows in the small launch cost nothing).  No split when (almost) every row is a boundary row.
template <typename T>
static void planHaloFirst(smm_hip_dist_csr* D) {
	D->haloFirst = false;
	D->boundary = RowRanges{};
	D->bulk = RowRanges{};
	const int n = D->nLocal;
	D->bulk.n = 1;
	D->bulk.lo[0] = 0;
	D->bulk.hi[0] = n;
	static const bool allowed = [] {
		const char* env = getenv("SMM_HIP_HALO_FIRST");
		return env ? atoi(env) != 0 : true;
	}();
	if (!allowed || D->sends.empty() || n <= 0) return;
	constexpr int VEC = 16 / sizeof(T);
	std::vector<std::pair<int, int>> runs;
	for (const Seg& g : D->sends) {
		int lo = g.offset - D->ownOffset, hi = lo + g.count;
		lo = std::max(0, lo / VEC * VEC);
		hi = std::min(n, (hi + VEC - 1) / VEC * VEC);
		if (lo < hi) runs.emplace_back(lo, hi);
	}
	std::sort(runs.begin(), runs.end());
	std::vector<std::pair<int, int>> merged;
	for (const auto& r : runs) {
		if (!merged.empty() && r.first <= merged.back().second) merged.back().second = std::max(merged.back().second, r.second);
		else merged.push_back(r);
	}
	long long total = 0;
	for (const auto& r : merged) total += r.second - r.first;
	if (merged.empty() || merged.size() > 3 || total * 4 > static_cast<long long>(n) * 3) return;  // (too scattered, or nothing left for a bulk launch)
	RowRanges b{}, rest{};
	int at = 0;
	for (const auto& r : merged) {
		b.lo[b.n] = r.first;
		b.hi[b.n] = r.second;
		++b.n;
		if (at < r.first) {
			rest.lo[rest.n] = at;
			rest.hi[rest.n] = r.first;
			++rest.n;
		}
		at = r.second;
	}
	if (at < n) {
		rest.lo[rest.n] = at;
		rest.hi[rest.n] = n;
		++rest.n;
	}
	D->boundary = b;
	D->bulk = rest;
	D->haloFirst = true;
}

// ---------------------------------------------------------------------------------------------------------
// peer-to-peer set-up (smm_p2p.h)
// ---------------------------------------------------------------------------------------------------------
// The halo plan of ANY rank from what every rank knows (the column ranges and the row bounds): the segments rank q receives, in
// ascending source order, with their positions in q's landing area.
struct PlanSeg {
	int src, dst;
	long long extOff;   // element offset in dst's halo-extended vector
	long long landOff;  // element offset in dst's landing area
	int count;
};
static std::vector<PlanSeg> planRecvs(int world, const long long* needs, const int* bounds, int q) {
	std::vector<PlanSeg> out;
	long long at = 0;
	for (int p = 0; p < world; ++p) {
		if (p == q) continue;
		const long long lo = std::max<long long>(needs[2 * static_cast<size_t>(q)], bounds[static_cast<size_t>(p)]);
		const long long hi = std::min<long long>(needs[2 * static_cast<size_t>(q) + 1], bounds[static_cast<size_t>(p) + 1]);
		if (lo < hi) {
			out.push_back({p, q, lo - needs[2 * static_cast<size_t>(q)], at, static_cast<int>(hi - lo)});
			at += hi - lo;
		}
	}
	return out;
}

// relay ranks of the segment p -> q: the ranks farthest (on the ring of ranks) from both ends first -- halo partners are near neighbours in
// a row partition, so those are the ranks whose links to p and to q carry nothing else
static std::vector<int> planRelays(int world, int p, int q, int want) {
	std::vector<int> cand;
	for (int r = 0; r < world; ++r) {
		if (r != p && r != q) cand.push_back(r);
	}
	auto ring = [world](int a, int b) {
		const int d = std::abs(a - b);
		return std::min(d, world - d);
	};
	std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return std::min(ring(a, p), ring(a, q)) > std::min(ring(b, p), ring(b, q)); });
	if (static_cast<int>(cand.size()) > want) cand.resize(static_cast<size_t>(std::max(0, want)));
	return cand;
}

// the parts of a segment of `count` elements: part 0 is the direct one, part k >= 1 goes through relay k - 1; cuts at multiples of 4
// elements.  bounds has parts + 1 entries.
static std::vector<int> planParts(int count, int relays, double directShare) {
	std::vector<int> cut(static_cast<size_t>(relays) + 2, 0);
	cut[static_cast<size_t>(relays) + 1] = count;
	if (count < 64 * (relays + 1)) {  // a short segment travels whole on the direct path (no empty part in front of a non-empty one, ever)
		for (int k = 1; k <= relays; ++k) cut[static_cast<size_t>(k)] = count;
		return cut;
	}
	const double rest = relays > 0 ? (1.0 - directShare) / relays : 0.0;
	double cum = directShare;
	for (int k = 1; k <= relays; ++k) {
		int c = static_cast<int>(count * cum) & ~3;
		c = std::max(cut[static_cast<size_t>(k) - 1], std::min(count, c));
		cut[static_cast<size_t>(k)] = c;
		cum += rest;
	}
	return cut;
}

// Every part of every segment of every rank, and what each relay stages where: the SAME table on every rank, from global knowledge only.
struct P2PRoute {
	PlanSeg seg;
	int part, relay, a, b;  // elements [a, b) of the segment; relay < 0: the direct part
	int job;                // index among the relay's jobs
	long long stagePos;     // element offset in the relay's staging area
};
struct P2PPlan {
	std::vector<P2PRoute> routes;
	std::vector<long long> stageTotal;  // elements of staging area per rank
	std::vector<int> jobCount;          // relay jobs per rank
};
static P2PPlan p2pMakePlan(int world, const long long* needs, const int* bounds, int relays, double directShare, bool* tooMany) {
	P2PPlan plan;
	plan.stageTotal.assign(static_cast<size_t>(world), 0);
	plan.jobCount.assign(static_cast<size_t>(world), 0);
	*tooMany = false;
	for (int q = 0; q < world; ++q) {
		for (const PlanSeg& g : planRecvs(world, needs, bounds, q)) {
			const std::vector<int> via = planRelays(world, g.src, q, relays);
			const std::vector<int> cut = planParts(g.count, static_cast<int>(via.size()), directShare);
			for (size_t k = 0; k + 1 < cut.size(); ++k) {
				P2PRoute rt{g, static_cast<int>(k), k == 0 ? -1 : via[k - 1], cut[k], cut[k + 1], -1, 0};
				if (rt.relay >= 0) {
					rt.job = plan.jobCount[static_cast<size_t>(rt.relay)]++;
					rt.stagePos = plan.stageTotal[static_cast<size_t>(rt.relay)];
					plan.stageTotal[static_cast<size_t>(rt.relay)] += (rt.b - rt.a + 3) & ~3;
					if (rt.job >= P2P_MAX_JOBS) *tooMany = true;
				}
				plan.routes.push_back(rt);
			}
		}
	}
	return plan;
}

// Collective (every rank tears its matrix down in the same order as it built it): a rank's block must outlive every remote write into it
// -- a peer's land kernel acknowledges into the SOURCE's block after the source may have returned from its solve --, so every rank first
// drains its own device (its remote writes are done), the ranks meet, the mappings are closed, the ranks meet again, and only then is the
// block freed.  A communicator that was aborted cannot meet: the mappings are closed and the block freed at once (the process is expected
// to exit).
static void p2pTeardown(smm_hip_dist_csr* D) {
	P2PState* P = D->p2p;
	if (!P) return;
	smm_hip_comm* c = D->comm;
	auto meet = [c]() {
		if (!c || c->broken || c->kind == SMM_COMM_SELF) return;
		long long one = 1;
		if (commAllreduceI64(c, &one, 1) != SMM_HIP_OK) c->broken = true;
	};
	(void)hipDeviceSynchronize();
	meet();
	for (size_t q = 0; q < P->peer.size(); ++q) {
		if (P->opened[q] && P->peer[q]) (void)hipIpcCloseMemHandle(P->peer[q]);
	}
	meet();
	for (void* p : {static_cast<void*>(P->d_push), static_cast<void*>(P->d_fwd), static_cast<void*>(P->d_land), static_cast<void*>(P->d_counters)}) devFree(p);
	if (P->block) (void)hipFree(P->block);
	(void)hipGetLastError();
	delete P;
	D->p2p = nullptr;
}

static long long p2pTicks() {
	static const long long t = [] {
		const char* env = getenv("SMM_HIP_P2P_TIMEOUT_S");
		const double sec = env && atof(env) > 0 ? atof(env) : 20.0;
		return static_cast<long long>(sec * 1.0e8);  // wall_clock64 counts at 100 MHz on gfx9
	}();
	return t;
}

// sum of one 0 / 1 vote per rank == world?
static int p2pAllAgree(smm_hip_comm* c, bool mine, bool* all) {
	std::vector<long long> v(1, mine ? 1 : 0);
	SMM_TRY(commAllreduceI64(c, v.data(), 1));
	*all = v[0] == c->world;
	return SMM_HIP_OK;
}

template <typename T>
static int p2pHaloLaunch(smm_hip_dist_csr* D, T* ext, int kind, hipStream_t cs);
template <typename T>
static int p2pLandLaunch(smm_hip_dist_csr* D, T* ext, int kind, unsigned long long seq, hipStream_t s, unsigned long long* word = nullptr, unsigned long long wordSeq = 0);
template <typename T>
static int p2pAllreduceLaunch(smm_hip_dist_csr* D, int point, T* totals, int count, const int* doneFlag, hipStream_t s, const T* parts = nullptr);

// Collective.  Leaves D->p2p null (the communicator's collectives are used) unless EVERY rank asked for the peer-to-peer path, could
// allocate and export its block, map every peer's and pass the self-test through every path.
template <typename T>
static int p2pSetup(smm_hip_dist_csr* D) {
	smm_hip_comm* c = D->comm;
	const int world = c->world, rank = c->rank;
	// (SMM_HIP_LAB_SELF_P2P=1, single-rank communicator: the slots with ONE rank -- what a rank of a many-GPU run launches per iteration in this
	// transport, measured on one GPU; tools/lab/rank_loop_streams.py)
	const bool labSelf = world == 1 && c->kind == SMM_COMM_SELF && getenv("SMM_HIP_LAB_SELF_P2P") && atoi(getenv("SMM_HIP_LAB_SELF_P2P")) != 0;
	if ((world < 2 && !labSelf) || world > P2P_MAX_WORLD) return SMM_HIP_OK;
	// r06: ON unless a rank says SMM_HIP_P2P=0 -- the transport is taken whenever EVERY rank can allocate, export and map the blocks and passes
	// the self-test below; anything less leaves all ranks with the communicator's collectives (read at every create, like SMM_HIP_HALO_CHUNKS)
	const char* env = getenv("SMM_HIP_P2P");
	bool all = false;
	SMM_TRY(p2pAllAgree(c, !(env && atoi(env) == 0) && D->chunks == 1, &all));
	if (!all) return SMM_HIP_OK;
	std::unique_ptr<P2PState> owner(new P2PState());
	P2PState* P = owner.get();
	P->world = world;
	P->rank = rank;
	P->ticks = p2pTicks();
	{
		const char* r = getenv("SMM_HIP_P2P_RELAYS");
		P->relays = r ? atoi(r) : std::max(0, world - 4);  // 8 ranks: 4 relays per segment -- half of it direct, an eighth through each relay
		P->relays = std::max(0, std::min(std::min(P->relays, world - 2), P2P_MAX_PATHS - 1));
		// the direct link carries its share once; a relay's links carry up to two shares in each of the two stages: d = 4 rho, d + R rho = 1
		P->directShare = 4.0 / (P->relays + 4.0);
		if (const char* d = getenv("SMM_HIP_P2P_DIRECT_SHARE")) P->directShare = std::min(1.0, std::max(0.05, atof(d)));
		if (P->relays == 0) P->directShare = 1.0;
	}
	// ---- the plan, from global knowledge: what I receive, what I send (directly / staged at a relay), what I forward
	const std::vector<PlanSeg> mine = planRecvs(world, D->needs.data(), D->bounds.data(), rank);
	bool tooMany = false;
	const P2PPlan plan = p2pMakePlan(world, D->needs.data(), D->bounds.data(), P->relays, P->directShare, &tooMany);
	const std::vector<P2PRoute>& routes = plan.routes;
	const std::vector<long long>& stageTotal = plan.stageTotal;
	using Route = P2PRoute;
	// ---- my block: header | landing x 3 | staging x 3, element-aligned to 256 bytes
	long long landElems = 0;
	for (const PlanSeg& g : mine) landElems += g.count;
	P->landElems = landElems;
	P->stageElems = stageTotal[static_cast<size_t>(rank)];
	auto up = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
	size_t at = up(sizeof(P2PHeader));
	for (int k = 0; k < P2P_KINDS; ++k) {
		P->landOff[k] = at;
		at += up(static_cast<size_t>(std::max<long long>(1, landElems)) * sizeof(T) + 64);
	}
	for (int k = 0; k < P2P_KINDS; ++k) {
		P->stageOff[k] = at;
		at += up(static_cast<size_t>(std::max<long long>(1, P->stageElems)) * sizeof(T) + 64);
	}
	P->bytes = at;
	bool ok = !tooMany;
	if (ok) {
		hipError_t e = hipExtMallocWithFlags(&P->block, P->bytes, hipDeviceMallocFinegrained);
		if (e != hipSuccess) {
			(void)hipGetLastError();
			P->block = nullptr;
			e = hipExtMallocWithFlags(&P->block, P->bytes, hipDeviceMallocUncached);
		}
		if (e != hipSuccess) {
			(void)hipGetLastError();
			P->block = nullptr;
			ok = false;
		} else {
			ok = hipMemset(P->block, 0, P->bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
		}
	}
	// ---- every rank learns every rank's handle, process, address and layout
	constexpr int WORDS = 8 + 2 + 2 * P2P_KINDS + 2;  // handle | pid, address | landing, staging offsets | ok | which GPU (a hash of its PCI bus id)
	long long myGpu = 0;
	{
		int dev = 0;
		char bus[64] = {};
		if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetPCIBusId(bus, sizeof(bus), dev) == hipSuccess) {
			unsigned long long hsh = 1469598103934665603ull;  // FNV-1a
			for (const char* ch = bus; *ch; ++ch) hsh = (hsh ^ static_cast<unsigned char>(*ch)) * 1099511628211ull;
			myGpu = static_cast<long long>(hsh >> 1) | 1;
		} else {
			(void)hipGetLastError();
		}
	}
	std::vector<long long> table(static_cast<size_t>(world) * WORDS, 0);
	long long* me = table.data() + static_cast<size_t>(rank) * WORDS;
	hipIpcMemHandle_t handle{};
	static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
	if (ok && hipIpcGetMemHandle(&handle, P->block) != hipSuccess) {
		(void)hipGetLastError();
		ok = false;
	}
	memcpy(me, &handle, 64);
	me[8] = static_cast<long long>(getpid());
	me[9] = static_cast<long long>(reinterpret_cast<uintptr_t>(P->block));
	for (int k = 0; k < P2P_KINDS; ++k) {
		me[10 + k] = static_cast<long long>(P->landOff[k]);
		me[10 + P2P_KINDS + k] = static_cast<long long>(P->stageOff[k]);
	}
	me[10 + 2 * P2P_KINDS] = ok ? 1 : 0;
	me[WORDS - 1] = myGpu;
	SMM_TRY(commAllreduceI64(c, table.data(), world * WORDS));
	for (int q = 0; q < world; ++q) ok = ok && table[static_cast<size_t>(q) * WORDS + 10 + 2 * P2P_KINDS] == 1;
	// Ranks that SHARE a GPU (a rehearsal of several ranks on one card) keep the two-launch SpMV: the one-launch form waits inside a grid that fills
	// the chip but for one CU per XCD's worth of slots, and on a shared card a peer's grid takes exactly those slots -- the land and push kernels
	// the grids are waiting for then find no room until the bounded wait expires (seen with 2 x 2.5 M rows on one MI355X, r06).  One rank per
	// GPU -- the deployment -- leaves the room to the rank's own copy kernels.  Every rank sees the same table: every rank decides alike.
	for (int q = 0; q < world; ++q) {
		for (int r2 = q + 1; r2 < world; ++r2) {
			const long long a = table[static_cast<size_t>(q) * WORDS + WORDS - 1], b = table[static_cast<size_t>(r2) * WORDS + WORDS - 1];
			if (a == 0 || b == 0 || a == b) D->sharedGpu = true;
		}
	}
	P->peer.assign(static_cast<size_t>(world), nullptr);
	P->opened.assign(static_cast<size_t>(world), false);
	if (ok) {
		for (int q = 0; q < world && ok; ++q) {
			const long long* row = table.data() + static_cast<size_t>(q) * WORDS;
			if (q == rank) {
				P->peer[static_cast<size_t>(q)] = static_cast<char*>(P->block);
			} else if (row[8] == static_cast<long long>(getpid())) {
				// A rank of this very process (ranks as threads: the tests).  REFUSED since r06: this transport makes kernels of one rank wait for
				// kernels of another, and inside ONE process HIP gives no control over which hardware queue a stream's dispatches take -- a kernel
				// trace of the thread-rank test under GPU_MAX_HW_QUEUES=32 shows 15 of 59 streams dispatched on more than one queue and the side
				// streams of two live ranks on the same one (profiles/r06/p2p_thread_rank_queues.txt): a waiting kernel then sits in front of the
				// kernel it waits for until its bound expires (the intermittent time-outs of r05 / r06).  Ranks in separate processes -- one per
				// GPU, the deployment -- have queues of their own.  Every rank sees the same table, so every rank votes alike: the collectives.
				ok = false;
			} else {
				hipIpcMemHandle_t h{};
				memcpy(&h, row, 64);
				void* ptr = nullptr;
				if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
					(void)hipGetLastError();
					ok = false;
				} else {
					P->peer[static_cast<size_t>(q)] = static_cast<char*>(ptr);
					P->opened[static_cast<size_t>(q)] = true;
				}
			}
		}
	}
	// ---- device tables
	std::vector<P2PJob> push, fwd;
	std::vector<P2PLandSeg> land;
	if (ok) {
		auto hdrOf = [&](int q) { return reinterpret_cast<P2PHeader*>(P->peer[static_cast<size_t>(q)]); };
		auto rowOf = [&](int q) { return table.data() + static_cast<size_t>(q) * WORDS; };
		for (const Route& rt : routes) {
			const int n = rt.b - rt.a;
			if (n <= 0) continue;
			const int q = rt.seg.dst;
			if (rt.seg.src == rank) {  // I push this part: straight into q's landing area, or into the relay's staging area
				P2PJob j{};
				j.srcOff = (static_cast<long long>(D->bounds[static_cast<size_t>(rank)]) - D->cmin) +
				           (D->needs[2 * static_cast<size_t>(q)] + rt.seg.extOff - D->bounds[static_cast<size_t>(rank)]) + rt.a;
				j.count = n;
				j.waitJob = -1;
				j.ackFrom = q;
				for (int k = 0; k < P2P_KINDS; ++k) {
					if (rt.relay < 0) {
						j.dst[k] = P->peer[static_cast<size_t>(q)] + rowOf(q)[10 + k] + (rt.seg.landOff + rt.a) * static_cast<long long>(sizeof(T));
						j.flag[k] = &hdrOf(q)->haloFlag[k][rank][0];
					} else {
						j.dst[k] = P->peer[static_cast<size_t>(rt.relay)] + rowOf(rt.relay)[10 + P2P_KINDS + k] + rt.stagePos * static_cast<long long>(sizeof(T));
						j.flag[k] = &hdrOf(rt.relay)->stageFlag[k][rt.job];
					}
				}
				push.push_back(j);
			}
			if (rt.relay == rank) {  // I forward this part from my staging area into q's landing area
				P2PJob j{};
				j.srcOff = rt.stagePos;
				j.count = n;
				j.waitJob = rt.job;
				j.ackFrom = q;
				for (int k = 0; k < P2P_KINDS; ++k) {
					j.dst[k] = P->peer[static_cast<size_t>(q)] + rowOf(q)[10 + k] + (rt.seg.landOff + rt.a) * static_cast<long long>(sizeof(T));
					j.flag[k] = &hdrOf(q)->haloFlag[k][rt.seg.src][rt.part];
				}
				fwd.push_back(j);
			}
		}
		for (const PlanSeg& g : mine) {
			int paths = 0;
			for (const Route& rt : routes) {
				if (rt.seg.dst == rank && rt.seg.src == g.src && rt.b > rt.a) paths = std::max(paths, rt.part + 1);
			}
			// (a part without elements is never signalled: the land kernel waits for the parts that exist -- they are the first `paths` ones only when
			// no part in between is empty, which planParts guarantees except for tiny segments, where every part but the direct one may be empty)
			bool gaps = false;
			for (const Route& rt : routes) {
				if (rt.seg.dst == rank && rt.seg.src == g.src && rt.b <= rt.a && rt.part < paths) gaps = true;
			}
			if (gaps) ok = false;
			P2PLandSeg ls{g.landOff, g.extOff, g.count, g.src, paths, {}};
			for (int k = 0; k < P2P_KINDS; ++k) ls.ack[k] = &hdrOf(g.src)->ackFlag[k][rank];
			land.push_back(ls);
		}
		P->nPush = static_cast<int>(push.size());
		P->nFwd = static_cast<int>(fwd.size());
		P->nLand = static_cast<int>(land.size());
		for (int q = 0; q < world; ++q) P->peers.hdr[q] = hdrOf(q);
		auto upload = [&](auto& vec, auto** d) -> int {
			using E = typename std::remove_reference<decltype(vec)>::type::value_type;
			void* p = nullptr;
			SMM_TRY(devAlloc(&p, std::max<size_t>(1, vec.size()) * sizeof(E)));
			*d = static_cast<E*>(p);
			if (!vec.empty()) SMM_HIP_TRY(hipMemcpy(p, vec.data(), vec.size() * sizeof(E), hipMemcpyHostToDevice));
			return SMM_HIP_OK;
		};
		// (a failure here must not return: the other ranks are about to vote, and a rank that left would leave them alone in the collective)
		void* cnt = nullptr;
		const size_t nCounters = static_cast<size_t>(P->nPush + P->nFwd + P->nLand + 1);  // (+ 1: finished segments of a land launch)
		if (upload(push, &P->d_push) != SMM_HIP_OK || upload(fwd, &P->d_fwd) != SMM_HIP_OK || upload(land, &P->d_land) != SMM_HIP_OK ||
		    devAlloc(&cnt, nCounters * sizeof(unsigned)) != SMM_HIP_OK || hipMemset(cnt, 0, nCounters * sizeof(unsigned)) != hipSuccess) {
			(void)hipGetLastError();
			ok = false;
		}
		P->d_counters = static_cast<unsigned*>(cnt);
	}
	SMM_TRY(p2pAllAgree(c, ok, &all));
	D->p2p = owner.release();  // (torn down below unless everything checks out; the peers' mappings must be closed either way)
	if (!all) {
		p2pTeardown(D);
		return SMM_HIP_OK;
	}
	// ---- self-test, in two parts with a vote each (no early return between votes -- ADVICE r05: a rank that left would tear its block down in
	// p2pTeardown, whose "meet" all-reduces would pair with the peers' votes; every failure is a `pass = false`).
	// (on the communicator's own stream: kernels of this rank WAIT for kernels of its peers)
	hipStream_t s = c->stream;
	noteStream(s);
	// Part 1, the scalars: every reduction point once -- rank + 1 and 2 (rank + 1) through the slots must give world (world + 1) / 2 and twice that
	// on every rank.  A failure here leaves every rank with the collectives for everything.
	bool pass = true;
	{
		DevBuf<T> tot;
		T mineTot[2 * P2P_RED_POINTS], got[2 * P2P_RED_POINTS] = {};
		for (int k = 0; k < P2P_RED_POINTS; ++k) {
			mineTot[2 * k] = static_cast<T>(rank + 1 + k);
			mineTot[2 * k + 1] = static_cast<T>(2 * (rank + 1 + k));
		}
		unsigned long long err = 0;
		bool okHost = tot.alloc(2 * P2P_RED_POINTS) == SMM_HIP_OK;
		if (okHost) okHost = hipMemcpyAsync(tot, mineTot, sizeof(mineTot), hipMemcpyHostToDevice, s) == hipSuccess;
		int rc = SMM_HIP_OK;
		for (int k = 0; k < P2P_RED_POINTS && okHost && rc == SMM_HIP_OK; ++k) rc = p2pAllreduceLaunch<T>(D, k, tot.p + 2 * k, 2, nullptr, s);
		if (okHost) okHost = hipMemcpyAsync(got, tot, sizeof(got), hipMemcpyDeviceToHost, s) == hipSuccess;
		okHost = hipMemcpyAsync(&err, &P->hdr()->err, sizeof(err), hipMemcpyDeviceToHost, s) == hipSuccess && okHost;
		okHost = hipStreamSynchronize(s) == hipSuccess && okHost;
		if (!okHost) (void)hipGetLastError();
		pass = okHost && rc == SMM_HIP_OK && err == 0;
		for (int k = 0; k < P2P_RED_POINTS && pass; ++k) {
			const T want = static_cast<T>(0.5 * world * (world + 1) + static_cast<double>(k) * world);
			pass = got[2 * k] == want && got[2 * k + 1] == 2 * want;
		}
	}
	SMM_TRY(p2pAllAgree(c, pass, &all));
	if (!all) {
		if (!pass) fprintf(stderr, "libsmm_hip: rank %d: the peer-to-peer self-test (scalars) failed; every rank stays with the communicator's collectives\n", rank);
		p2pTeardown(D);
		return SMM_HIP_OK;
	}
	// Part 2, the halo: the halo-extended vector carries (global column + 7 round) mod 8191 in every owned element; after an exchange every halo
	// element of every rank must hold ITS column's number -- once through the areas of each vector kind, then twice more back to back through
	// the last kind's (a push then waits for the acknowledgement of the exchange before).  A failure here leaves the HALO with the
	// communicator's grouped send / receive and keeps the scalars in the slots (the hybrid: a reduction point costs ~3 us instead of ~23).
	const char* haloEnv = getenv("SMM_HIP_P2P_HALO");  // 0: do not try (tests of the hybrid)
	pass = !(haloEnv && atoi(haloEnv) == 0);
	{
		T* xExt = static_cast<T*>(D->xExt);
		std::vector<T> host(static_cast<size_t>(D->extLen), T(-1));
		unsigned long long err = 0;
		const int kinds[5] = {0, 1, 2, 2, 2};
		for (int round = 0; round < 5 && pass; ++round) {
			std::fill(host.begin(), host.end(), T(-1));
			for (int i = 0; i < D->nLocal; ++i) host[static_cast<size_t>(D->ownOffset + i)] = static_cast<T>((D->rowBegin + i + 7LL * round) % 8191);
			bool okHost = hipMemcpyAsync(xExt, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, s) == hipSuccess;
			// (whatever fails on the host, the launches go out if they can at all: the peers' kernels are waiting for this rank's parts)
			int rc = p2pHaloLaunch<T>(D, xExt, kinds[round], s);
			if (rc == SMM_HIP_OK) rc = p2pLandLaunch<T>(D, xExt, kinds[round], P->haloSeq[kinds[round]], s);
			okHost = hipMemcpyAsync(host.data(), xExt, host.size() * sizeof(T), hipMemcpyDeviceToHost, s) == hipSuccess && okHost;
			okHost = hipMemcpyAsync(&err, &P->hdr()->err, sizeof(err), hipMemcpyDeviceToHost, s) == hipSuccess && okHost;
			okHost = hipStreamSynchronize(s) == hipSuccess && okHost;
			if (!okHost) (void)hipGetLastError();
			pass = okHost && rc == SMM_HIP_OK && err == 0;
			for (const PlanSeg& g : mine) {
				for (int i = 0; i < g.count && pass; ++i) {
					const long long col = D->cmin + g.extOff + i;
					pass = host[static_cast<size_t>(g.extOff + i)] == static_cast<T>((col + 7LL * round) % 8191);
				}
			}
		}
		if (hipMemsetAsync(xExt, 0, host.size() * sizeof(T), s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) (void)hipGetLastError();
	}
	SMM_TRY(p2pAllAgree(c, pass, &all));
	if (!all) {
		if (!pass && !(haloEnv && atoi(haloEnv) == 0)) {
			fprintf(stderr, "libsmm_hip: rank %d: the peer-to-peer self-test (halo) failed; the halo stays with the communicator's send / receive on every rank\n", rank);
		}
		// every rank is past its own waits (each synchronised its stream above, and the vote was a collective): the error words of the halo part
		// are history, the slots go on
		P->haloOn = false;
		P->relays = 0;
		P->directShare = 1.0;
		if (hipMemset(&P->hdr()->err, 0, sizeof(unsigned long long)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
		bool cleared = false;
		SMM_TRY(p2pAllAgree(c, true, &cleared));  // (nobody enters a reduction before every rank has cleared its word)
	}
	return SMM_HIP_OK;
}

// the exchange of `ext`'s boundary slices: push kernel (direct parts + the shares staged at relays), then this rank's forwards
template <typename T>
static int p2pHaloLaunch(smm_hip_dist_csr* D, T* ext, int kind, hipStream_t cs) {
	P2PState* P = D->p2p;
	const unsigned long long seq = ++P->haloSeq[kind];
	if (P->nPush > 0) {
		p2pCopyKernel<T, false><<<dim3(P2P_BLOCKS_PER_JOB, static_cast<unsigned>(P->nPush)), P2P_TPB, 0, cs>>>(P->d_push, P->d_counters, ext, kind, seq, P->hdr(), P->ticks);
	}
	if (P->nFwd > 0) {
		const T* staging = reinterpret_cast<const T*>(static_cast<char*>(P->block) + P->stageOff[kind]);
		p2pCopyKernel<T, true><<<dim3(P2P_BLOCKS_PER_JOB, static_cast<unsigned>(P->nFwd)), P2P_TPB, 0, cs>>>(P->d_fwd, P->d_counters + P->nPush, staging, kind, seq, P->hdr(),
		                                                                                              P->ticks);
	}
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T>
static int p2pLandLaunch(smm_hip_dist_csr* D, T* ext, int kind, unsigned long long seq, hipStream_t s, unsigned long long* word, unsigned long long wordSeq) {
	P2PState* P = D->p2p;
	if (P->nLand == 0) {
		if (word) launchSplitSignal(word, wordSeq, s);  // (nothing lands here, yet a launch is waiting for the word)
		return SMM_HIP_OK;
	}
	const T* landing = reinterpret_cast<const T*>(static_cast<char*>(P->block) + P->landOff[kind]);
	p2pLandKernel<T><<<dim3(16, static_cast<unsigned>(P->nLand)), P2P_TPB, 0, s>>>(P->d_land, P->d_counters + P->nPush + P->nFwd, landing, ext, kind, seq, P->hdr(), P->ticks,
	                                                                               word, wordSeq, P->d_counters + P->nPush + P->nFwd + P->nLand);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T>
static int p2pAllreduceLaunch(smm_hip_dist_csr* D, int point, T* totals, int count, const int* doneFlag, hipStream_t s, const T* parts) {
	P2PState* P = D->p2p;
	const unsigned long long seq = ++P->redSeq[point];
	p2pAllreduceKernel<T><<<1, P2P_TPB, 0, s>>>(P->peers, P->world, P->rank, point, seq, count, totals, parts, NPART, P->ticks, doneFlag);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

// the error word of the peer-to-peer path, read wherever the host reads `done` (stream `s` is synchronised by the caller afterwards)
static int p2pPostErrRead(smm_hip_dist_csr* D, unsigned long long* err, hipStream_t s) {
	*err = 0;
	// (without the peer-to-peer transport the one-launch SpMV's own word -- smm_hip_dist_csr::splitSync -- is the only bounded device-side wait)
	const unsigned long long* word = D->p2p ? &D->p2p->hdr()->err : D->splitSync ? D->splitSync + P2P_KINDS : nullptr;
	if (!word) return SMM_HIP_OK;
	SMM_HIP_TRY(hipMemcpyAsync(err, word, sizeof(*err), hipMemcpyDeviceToHost, s));
	return SMM_HIP_OK;
}
static int p2pFailIf(smm_hip_dist_csr* D, unsigned long long err) {
	if (!err) return SMM_HIP_OK;
	const unsigned what = static_cast<unsigned>(err >> 32);
	static const char* const names[] = {"?", "a push for the destination's acknowledgement of the previous exchange (vector, destination rank)",
	                                    "a forward for its staged share (vector, relay job)", "the land kernel for a part (vector, source rank, path)",
	                                    "a reduction for a rank's slot (reduction point, rank)", "the one-launch SpMV for the halo of its exchange"};
	setError("dist: rank %d waited longer than SMM_HIP_P2P_TIMEOUT_S for a peer (%s): %s = (%u, %u, %u), sequence number %llu; the communicator "
	         "is unusable", D->comm->rank, D->p2p ? "peer-to-peer path" : "collectives", names[std::min(5u, what >> 12)], (what >> 8) & 0xFu, (what >> 4) & 0xFu, what & 0xFu,
	         err & 0xFFFFFFFFull);
	fprintf(stderr, "libsmm_hip: %s\n", smm_hip_last_error());
	D->comm->broken = true;
	return SMM_HIP_ERR_COMM;
}

template <typename T>
static int distCreate(smm_hip_comm* comm, int nGlobal, const int* bounds, const int* d_start, const int* d_positions, const T* d_values,
                      smm_hip_dist_csr** out) {
	if (!out) {
		setError("dist_csr_create: out is null");
		return SMM_HIP_ERR_INVALID;
	}
	*out = nullptr;
	if (!comm || !bounds || !d_start || nGlobal < 0) {
		setError("dist_csr_create: null argument");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	const int world = comm->world, rank = comm->rank;
	if (bounds[0] != 0 || bounds[world] != nGlobal) {
		setError("dist_csr_create: bounds must run from 0 to n_global");
		return SMM_HIP_ERR_INVALID;
	}
	for (int q = 0; q < world; ++q) {
		if (bounds[q] > bounds[q + 1]) {
			setError("dist_csr_create: bounds must be non-decreasing");
			return SMM_HIP_ERR_INVALID;
		}
	}
	struct Guard {
		smm_hip_dist_csr* d;
		~Guard() {
			if (d) smm_hip_dist_csr_destroy(d);
		}
	} guard{new smm_hip_dist_csr()};
	smm_hip_dist_csr* D = guard.d;
	D->comm = comm;
	D->dtype = dtypeOf<T>();
	D->nGlobal = nGlobal;
	D->rowBegin = bounds[rank];
	D->rowEnd = bounds[rank + 1];
	const int nLocal = D->nLocal = D->rowEnd - D->rowBegin;
	hipStream_t s = libStream();
	SMM_HIP_TRY(hipDeviceSynchronize());  // the caller's arrays may still be being written on one of its streams (set-up path)
	int nnz = 0;
	SMM_HIP_TRY(hipMemcpyAsync(&nnz, d_start + nLocal, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	if (nnz < 0 || (nnz > 0 && (!d_positions || !d_values))) {
		setError("dist_csr_create: bad start[] or null positions / values");
		return SMM_HIP_ERR_INVALID;
	}
	// column range this rank's rows touch
	int hmm[2] = {0x7fffffff, -1};
	if (nnz > 0) {
		DevBuf<int> mm;
		SMM_TRY(mm.alloc(2));
		SMM_HIP_TRY(hipMemcpyAsync(mm, hmm, sizeof(hmm), hipMemcpyHostToDevice, s));
		colRangeKernel<<<std::min(2048, (nnz + 255) / 256), 256, 0, s>>>(nnz, d_positions, mm);
		SMM_HIP_TRY(hipMemcpyAsync(hmm, mm, sizeof(hmm), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
		if (hmm[0] < 0 || hmm[1] >= nGlobal) {
			setError("dist_csr_create: column %d outside [0, %d)", hmm[0] < 0 ? hmm[0] : hmm[1], nGlobal);
			return SMM_HIP_ERR_INVALID;
		}
	}
	D->cmin = nnz > 0 ? std::min(hmm[0], D->rowBegin) : D->rowBegin;
	D->cmaxExcl = nnz > 0 ? std::max(hmm[1] + 1, D->rowEnd) : D->rowEnd;
	D->extLen = std::max(1, D->cmaxExcl - D->cmin);
	D->ownOffset = D->rowBegin - D->cmin;
	// every rank learns every rank's range (an all-gather written as a sum of disjoint contributions)
	std::vector<long long> needs(2 * static_cast<size_t>(world), 0);
	needs[2 * rank] = D->cmin;
	needs[2 * rank + 1] = D->cmaxExcl;
	SMM_TRY(commAllreduceI64(comm, needs.data(), 2 * world));
	D->needs = needs;
	D->bounds.assign(bounds, bounds + world + 1);
	for (int q = 0; q < world; ++q) {
		if (q == rank) continue;
		long long lo = std::max<long long>(needs[2 * rank], bounds[q]), hi = std::min<long long>(needs[2 * rank + 1], bounds[q + 1]);
		if (lo < hi) {
			D->recvs.push_back({q, static_cast<int>(lo - D->cmin), static_cast<int>(hi - lo)});
			D->haloElements += static_cast<int>(hi - lo);
		}
		lo = std::max<long long>(needs[2 * q], D->rowBegin);
		hi = std::min<long long>(needs[2 * q + 1], D->rowEnd);
		if (lo < hi) D->sends.push_back({q, static_cast<int>(lo - D->cmin), static_cast<int>(hi - lo)});
	}
	// split
	int* cnt = nullptr;
	SMM_TRY(devAlloc(reinterpret_cast<void**>(&cnt), 2 * (static_cast<size_t>(nLocal) + 1) * sizeof(int)));
	D->arrays[0] = cnt;  // becomes startLoc | startRem
	int* startLoc = cnt;
	int* startRem = cnt + nLocal + 1;
	const int grid = std::max(1, std::min(8192, (nLocal + 256) / 256));
	{
		const char* env = getenv("SMM_HIP_SPLIT_SPMV");  // (read at every create, like SMM_HIP_HALO_CHUNKS: a property of the matrix)
		D->splitAllowed = env ? atoi(env) != 0 : true;
		D->splitForced = env && atoi(env) == 2;  // (2: also between ranks that share a GPU -- the tests, whose matrices leave the card half empty)
		const char* lds = getenv("SMM_HIP_SPLIT_SUMS_LDS");
		if (lds) D->splitSumsLdsMax = std::max(0, atoi(lds));
	}
	if (comm->kind == SMM_COMM_SELF) {
		const char* env = getenv("SMM_HIP_LAB_SELF_SPLIT");  // (measurement hook: what a rank of a many-GPU run computes, on one GPU -- tools/lab/rank_loop_streams.py)
		D->labWindow = env ? atoi(env) : 0;
	}
	splitCountKernel<<<grid, 256, 0, s>>>(nLocal, d_start, d_positions, D->rowBegin, D->rowEnd, D->labWindow, startLoc, startRem);
	SMM_HIP_TRY(hipGetLastError());
	SMM_TRY(exclusiveScan(startLoc, nLocal + 1, s));
	SMM_TRY(exclusiveScan(startRem, nLocal + 1, s));
	int totals[2] = {0, 0};
	SMM_HIP_TRY(hipMemcpyAsync(&totals[0], startLoc + nLocal, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipMemcpyAsync(&totals[1], startRem + nLocal, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	D->nnzLoc = totals[0];
	D->nnzRem = totals[1];
	D->remEmpty = totals[1] == 0;
	SMM_TRY(devAlloc(&D->arrays[1], std::max<size_t>(1, totals[0]) * sizeof(int)));
	SMM_TRY(devAlloc(&D->arrays[2], std::max<size_t>(1, totals[0]) * sizeof(T)));
	SMM_TRY(devAlloc(&D->arrays[4], std::max<size_t>(1, totals[1]) * sizeof(int)));
	SMM_TRY(devAlloc(&D->arrays[5], std::max<size_t>(1, totals[1]) * sizeof(T)));
	if (nLocal > 0) {
		splitScatterKernel<T><<<grid, 256, 0, s>>>(nLocal, d_start, d_positions, d_values, D->rowBegin, D->rowEnd, D->labWindow, D->cmin, startLoc, startRem,
		                                          static_cast<int*>(D->arrays[1]), static_cast<T*>(D->arrays[2]), static_cast<int*>(D->arrays[4]),
		                                          static_cast<T*>(D->arrays[5]));
		SMM_HIP_TRY(hipGetLastError());
	}
	SMM_HIP_TRY(hipStreamSynchronize(s));
	if (dtypeOf<T>() == SMM_DTYPE_F32) {
		SMM_TRY(smm_hip_csr_create_dev_f32(nLocal, nLocal, startLoc, static_cast<int*>(D->arrays[1]), static_cast<float*>(D->arrays[2]), &D->aLoc));
		SMM_TRY(smm_hip_csr_create_dev_f32(nLocal, D->extLen, startRem, static_cast<int*>(D->arrays[4]), static_cast<float*>(D->arrays[5]), &D->aRem));
	} else {
		SMM_TRY(smm_hip_csr_create_dev_f64(nLocal, nLocal, startLoc, static_cast<int*>(D->arrays[1]), static_cast<double*>(D->arrays[2]), &D->aLoc));
		SMM_TRY(smm_hip_csr_create_dev_f64(nLocal, D->extLen, startRem, static_cast<int*>(D->arrays[4]), static_cast<double*>(D->arrays[5]), &D->aRem));
	}
	// ---- a thin remote block: the rows with a remote entry, listed (each rank for itself: nothing collective depends on it)
	{
		const char* env = getenv("SMM_HIP_THIN_REMOTE");  // (read at every create: a property of the matrix; 0: the general second launch)
		const bool allowed = env ? atoi(env) != 0 : true;
		if (allowed && nLocal > 0 && totals[1] > 0) {
			DevBuf<int> flag;
			SMM_TRY(flag.alloc(static_cast<size_t>(nLocal) + 1));
			thinFlagKernel<<<grid, 256, 0, s>>>(nLocal, startRem, flag);
			SMM_HIP_TRY(hipGetLastError());
			SMM_TRY(exclusiveScan(flag, nLocal + 1, s));
			int nThin = 0;
			SMM_HIP_TRY(hipMemcpyAsync(&nThin, flag.p + nLocal, sizeof(int), hipMemcpyDeviceToHost, s));
			SMM_HIP_TRY(hipStreamSynchronize(s));
			if (nThin > 0 && static_cast<long long>(nThin) * THIN_SHARE <= nLocal) {
				SMM_TRY(devAlloc(reinterpret_cast<void**>(&D->thinRows), static_cast<size_t>(nThin) * sizeof(int)));
				thinListKernel<<<grid, 256, 0, s>>>(nLocal, startRem, flag, D->thinRows);
				SMM_HIP_TRY(hipGetLastError());
				SMM_TRY(devAlloc(&D->partsThin, PARTS_LEN * sizeof(T)));
				SMM_HIP_TRY(hipMemsetAsync(D->partsThin, 0, PARTS_LEN * sizeof(T), s));
				SMM_HIP_TRY(hipStreamSynchronize(s));
				D->nThin = nThin;
			}
		}
	}
	// ---- the halo in pieces (opt-in): every rank must cut alike, so the ranks agree on the smallest request
	{
		const char* env = getenv("SMM_HIP_HALO_CHUNKS");  // (read at every create: a property of the matrix, not of the process)
		const int wanted = env ? std::max(1, std::min(MAX_HALO_CHUNKS, atoi(env))) : 1;
		std::vector<long long> votes(static_cast<size_t>(world), 0);
		// (a rank whose segments do not fit the cut's table asks for one piece)
		votes[static_cast<size_t>(rank)] = D->recvs.size() > static_cast<size_t>(MAX_CHUNK_SEGS) ? 1 : wanted;
		SMM_TRY(commAllreduceI64(comm, votes.data(), world));
		int K = MAX_HALO_CHUNKS;
		for (long long v : votes) K = std::min<long long>(K, std::max<long long>(1, v));
		D->chunks = K;
	}
	if (D->chunks > 1) {
		const int K = D->chunks;
		auto piece = [K](const Seg& g, int k) {
			const int a = chunkBound(g.count, k, K), b = chunkBound(g.count, k + 1, K);
			return Seg{g.peer, g.offset + a, b - a};
		};
		D->sendsK.assign(static_cast<size_t>(K), {});
		D->recvsK.assign(static_cast<size_t>(K), {});
		for (int k = 0; k < K; ++k) {
			for (const Seg& g : D->sends) {
				const Seg q = piece(g, k);
				if (q.count > 0) D->sendsK[static_cast<size_t>(k)].push_back(q);
			}
			for (const Seg& g : D->recvs) {
				const Seg q = piece(g, k);
				if (q.count > 0) D->recvsK[static_cast<size_t>(k)].push_back(q);
			}
		}
		ChunkSegs segs{};
		segs.n = static_cast<int>(D->recvs.size());
		segs.K = K;
		for (int i = 0; i < segs.n; ++i) {
			segs.off[i] = D->recvs[static_cast<size_t>(i)].offset;
			segs.cnt[i] = D->recvs[static_cast<size_t>(i)].count;
		}
		int* starts = nullptr;
		SMM_TRY(devAlloc(reinterpret_cast<void**>(&starts), static_cast<size_t>(K) * (nLocal + 1) * sizeof(int)));
		D->chunkArrays.push_back(starts);
		chunkCountKernel<<<grid, 256, 0, s>>>(nLocal, startRem, static_cast<const int*>(D->arrays[4]), segs, starts);
		SMM_HIP_TRY(hipGetLastError());
		ChunkOut outs{};
		std::vector<int> totalsK(static_cast<size_t>(K), 0);
		for (int k = 0; k < K; ++k) SMM_TRY(exclusiveScan(starts + static_cast<size_t>(k) * (nLocal + 1), nLocal + 1, s));
		for (int k = 0; k < K; ++k) {
			SMM_HIP_TRY(hipMemcpyAsync(&totalsK[static_cast<size_t>(k)], starts + static_cast<size_t>(k) * (nLocal + 1) + nLocal, sizeof(int), hipMemcpyDeviceToHost, s));
		}
		SMM_HIP_TRY(hipStreamSynchronize(s));
		for (int k = 0; k < K; ++k) {
			void *pp = nullptr, *pv = nullptr;
			SMM_TRY(devAlloc(&pp, std::max<size_t>(1, totalsK[static_cast<size_t>(k)]) * sizeof(int)));
			D->chunkArrays.push_back(pp);
			SMM_TRY(devAlloc(&pv, std::max<size_t>(1, totalsK[static_cast<size_t>(k)]) * sizeof(T)));
			D->chunkArrays.push_back(pv);
			outs.pos[k] = static_cast<int*>(pp);
			outs.val[k] = pv;
		}
		if (nLocal > 0) {
			chunkScatterKernel<T><<<grid, 256, 0, s>>>(nLocal, startRem, static_cast<const int*>(D->arrays[4]), static_cast<const T*>(D->arrays[5]), segs, starts, outs);
			SMM_HIP_TRY(hipGetLastError());
		}
		SMM_HIP_TRY(hipStreamSynchronize(s));
		D->aRemK.assign(static_cast<size_t>(K), nullptr);
		D->nnzRemK.assign(totalsK.begin(), totalsK.end());
		for (int k = 0; k < K; ++k) {
			int* st = starts + static_cast<size_t>(k) * (nLocal + 1);
			if (dtypeOf<T>() == SMM_DTYPE_F32) {
				SMM_TRY(smm_hip_csr_create_dev_f32(nLocal, D->extLen, st, outs.pos[k], static_cast<float*>(outs.val[k]), &D->aRemK[static_cast<size_t>(k)]));
			} else {
				SMM_TRY(smm_hip_csr_create_dev_f64(nLocal, D->extLen, st, outs.pos[k], static_cast<double*>(outs.val[k]), &D->aRemK[static_cast<size_t>(k)]));
			}
		}
	}
	SMM_TRY(distWorkspace<T>(D));
	planHaloFirst<T>(D);
	SMM_TRY(p2pSetup<T>(D));
	guard.d = nullptr;
	*out = D;
	return SMM_HIP_OK;
}

// ---- one row-partitioned SpMV in two halves ---------------------------------------------------------------------------------------
// distExchangeBegin: the boundary slices of `ext`'s owned part are final on `s` -- post the halo exchange (side stream; peer-to-peer
// pushes or the communicator's grouped send / receive, in one piece or in `chunks`).  Called right behind the small "boundary" launch
// of a vector update (halo first), so that the transfer starts before the bulk of the update has run.
// distMatvecCompute: out = op(lhs, A ext) on the owned rows: the local block while the halo is in flight, then -- once it has landed --
// the remote block(s), with the dot products of the freshly computed vector (dotMode / w1 / parts as in launchSpmv; parts a finishing
// buffer) and the Jacobi division in the epilogue of the launch that completes a row.
static bool p2pHaloOn(const smm_hip_dist_csr* D) { return D->p2p && D->p2p->haloOn; }

// the device word the one-launch SpMV polls, and where an expired wait of it is recorded
static unsigned long long* splitErrWord(smm_hip_dist_csr* D) {
	if (D->p2p) return &D->p2p->hdr()->err;
	return D->splitSync ? D->splitSync + P2P_KINDS : nullptr;
}
// The one-launch SpMV is taken behind the peer-to-peer halo (the kernels it waits for are this library's own few workgroups, for which its grid
// leaves room) and wherever the exchange has completed before the launch is enqueued (the host-callback communicator: same stream).  Behind an
// RCCL exchange (the collectives, the hybrid) the two launches stay: a grid that waits inside the kernel must not depend on a library kernel
// whose resource needs are not ours to know finding room beside it.
static bool splitWordWanted(const smm_hip_dist_csr* D) {
	if (!D->splitAllowed || !D->splitSync || D->remEmpty || D->chunks != 1) return false;
	if (D->p2p && D->p2p->haloOn) return !D->sharedGpu || D->splitForced;  // (ranks that share a card: the two launches, see p2pSetup)
	return D->labWindow != 0 || D->comm->kind != SMM_COMM_RCCL;
}

template <typename T>
static int distExchangeBegin(smm_hip_dist_csr* D, T* ext, int kind, hipStream_t s) {
	smm_hip_comm* c = D->comm;
	auto& pend = D->pending;
	pend = smm_hip_dist_csr::Pending{};
	const bool exchange = !D->sends.empty() || !D->recvs.empty();
	if (D->labWindow != 0 && !D->remEmpty) {
		// one rank, "remote" entries by distance (measurements): nothing travels, but the word is raised from the communicator's stream behind
		// the update -- the one-launch SpMV takes exactly the path it takes behind a real exchange
		hipStream_t cs = c->stream;
		noteStream(cs);
		pend.active = true;
		pend.kind = kind;
		pend.async = true;
		SMM_TRY(orderAfter(c, s, cs));
		if (splitWordWanted(D)) {
			pend.landSeq = ++D->landSeq[kind];
			launchSplitSignal(D->splitSync + kind, pend.landSeq, cs);
		}
		pend.landed[0] = takeEvent(c);
		SMM_HIP_TRY(hipEventRecord(pend.landed[0], cs));
		pend.waitSlot[0] = profWaitAwaited(cs);
		return SMM_HIP_OK;
	}
	if (p2pHaloOn(D) && !exchange) {
		// A rank that neither sends nor receives may still RELAY (planRelays picks by ring distance alone: a decoupled diagonal block in a
		// world >= 3): its forwards must run, and its sequence numbers must advance, with every exchange of the peers -- r05 returned here
		// before looking at nFwd, and every solve of such a world expired.  Nothing of `ext` is read or written: no ordering
		// against `s`, no event for `s` to wait for, nothing pending.
		hipStream_t cs = c->stream;
		noteStream(cs);
		return p2pHaloLaunch<T>(D, ext, kind, cs);
	}
	if (!exchange) return SMM_HIP_OK;
	pend.active = true;
	pend.kind = kind;
	if (p2pHaloOn(D)) {
		hipStream_t cs = c->stream;
		noteStream(cs);
		SMM_TRY(orderAfter(c, s, cs));
		SMM_TRY(p2pHaloLaunch<T>(D, ext, kind, cs));
		pend.seq = D->p2p->haloSeq[kind];
		// the land kernel (waits for every part of every segment, then landing area -> halo of `ext`) runs on the side stream as well, beside
		// the local block: the solver's stream only waits for its event, as it waits for a collective exchange
		// (r06: the land kernel raises the word the one-launch SpMV polls itself -- write-through stores, drained, then the word: no launch behind it)
		if (splitWordWanted(D)) pend.landSeq = ++D->landSeq[kind];
		SMM_TRY(p2pLandLaunch<T>(D, ext, kind, pend.seq, cs, pend.landSeq ? D->splitSync + kind : nullptr, pend.landSeq));
		pend.landed[0] = takeEvent(c);
		SMM_HIP_TRY(hipEventRecord(pend.landed[0], cs));
		pend.waitSlot[0] = profWaitAwaited(cs);
		pend.async = true;
		return SMM_HIP_OK;
	}
	hipStream_t cs = c->kind == SMM_COMM_RCCL ? c->stream : s;  // the callback kind blocks anyway
	noteStream(cs);
	SMM_TRY(orderAfter(c, s, cs));
	pend.async = cs != s;
	if (D->chunks > 1) {
		// (`chunks` is agreed by all ranks at create time and decides ALONE which form of the exchange runs: a rank whose own A_rem is empty --
		// one-sided / upwind stencils, any structurally non-symmetric matrix: it sends but receives nothing -- must still post the K piece-sized
		// sends its peers' K receives are waiting for; r04 tested `!remEmpty` as well and such a rank fell through to ONE full-count exchange:
		// mismatched RCCL call sequences, a hang or a corrupt halo.  ADVICE r04.)
		// The halo in pieces: every piece is an exchange of its own on the communicator's stream (all of them enqueued at once, so the links
		// never idle between them)
		for (int k = 0; k < D->chunks; ++k) {
			SMM_TRY(commExchange<T>(c, ext, D->sendsK[static_cast<size_t>(k)], D->recvsK[static_cast<size_t>(k)], cs));
			if (cs != s) {
				pend.landed[k] = takeEvent(c);
				SMM_HIP_TRY(hipEventRecord(pend.landed[k], cs));
				pend.waitSlot[k] = profWaitAwaited(cs);
			}
		}
		return SMM_HIP_OK;
	}
	SMM_TRY(commExchange<T>(c, ext, D->sends, D->recvs, cs));
	if (splitWordWanted(D)) {  // (behind the grouped receive / the staged copies, on the stream they ran on)
		pend.landSeq = ++D->landSeq[kind];
		launchSplitSignal(D->splitSync + kind, pend.landSeq, cs);
	}
	if (cs != s) {
		pend.landed[0] = takeEvent(c);
		SMM_HIP_TRY(hipEventRecord(pend.landed[0], cs));
		pend.waitSlot[0] = profWaitAwaited(cs);
	}
	return SMM_HIP_OK;
}

// The second half of a row-partitioned SpMV whose remote block is THIN (smm_hip_dist_csr::thinRows): out[row] (+|-)= A_rem[row] . ext for the
// listed rows only -- one lane per row, the row's entries in order: the sum the one-lane kernels form -- where the general form passes over
// every row of out[] to add nothing to most of them and to read them for the dot products.  The dot products were taken by the LOCAL
// launch over the half-finished vector (its totals are in `totals`); this launch adds what its rows change: with o = a + d,
//   o . w = a . w + d . w          o . o = a . a + d (a + o)
// (the same numbers up to the order of the additions: the partial sums of a rank are not the reference's in any form).  The workgroups'
// shares go through a buffer of their own; thinFinishKernel adds them to `totals` in a fixed order.
constexpr int THIN_MAX_GRID = NPART;
// FORM (ConjugateGradient with the direction formed inside the SpMV, smm_spmv_march.hip MarchFuse): the local block's
// launch reads the previous direction and r and forms p = beta p_old + r for every element it touches; what it cannot form is the HALO of
// the new direction -- so the halo of r travels instead of p's (posted right behind distCgR, while ||r||^2 is reduced) and THIS launch forms
// the halo of p from it, twice: the entries it gathers are formed on the fly, and the halo is stored for the next iteration (no lane reads
// what another stores) -- fma(beta, p_old, r), the owner's expression on the owner's operands, beta from the same all-reduced total and the
// same rrPing on every rank: the owner's bits.  Runs behind the fused launch (whose workgroup 0 did iteration i - 1's bookkeeping: rrPing[par]
// is untouched, `done` is final).
struct HaloSegs {
	int n = 0;
	int off[P2P_MAX_WORLD] = {}, cnt[P2P_MAX_WORLD] = {};
};
template <typename T>
struct ThinForm {
	const T* pOldExt = nullptr;
	const T* rExt = nullptr;
	T* pNewExt = nullptr;
	const DistScal<T>* sc = nullptr;
	const T* totalsC = nullptr;
	int par = 0;
	HaloSegs segs{};
};
template <typename T, bool FORM>
__global__ __launch_bounds__(TPB) void thinRemoteKernel(int nThin, const int* __restrict__ rowsList, const int* __restrict__ start, const int* __restrict__ positions,
                                                        const T* __restrict__ values, const T* __restrict__ ext, int subtract, T* out, int dotMode,
                                                        const T* __restrict__ w1, T* __restrict__ partsThin, const int* __restrict__ doneFlag, ThinForm<T> fm) {
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	T beta = T(0);
	if constexpr (FORM) {
		beta = fm.totalsC[0] / fm.sc->rrPing[fm.par];
		for (int g = 0; g < fm.segs.n; ++g) {
			const int base = fm.segs.off[g];
			for (int i = blockIdx.x * TPB + threadIdx.x; i < fm.segs.cnt[g]; i += gridDim.x * TPB) {
				fm.pNewExt[base + i] = smmFma(beta, fm.pOldExt[base + i], fm.rExt[base + i]);
			}
		}
	}
	T acc0 = T(0), acc1 = T(0);
	for (int k = blockIdx.x * TPB + threadIdx.x; k < nThin; k += gridDim.x * TPB) {
		const int row = rowsList[k];
		const int e = start[row + 1];
		T sum = T(0);
		for (int j = start[row]; j < e; ++j) {
			const int c = positions[j];
			T xv;
			if constexpr (FORM) xv = smmFma(beta, fm.pOldExt[c], fm.rExt[c]);
			else xv = ext[c];
			sum = smmFma(values[j], xv, sum);
		}
		const T a = out[row];
		const T o = subtract ? a - sum : a + sum;
		out[row] = o;
		if (dotMode) {
			const T dlt = subtract ? -sum : sum;
			if (dotMode == 2) acc0 += dlt * (a + o);
			acc1 += dlt * w1[row];
		}
	}
	if (!dotMode) return;
	if (dotMode == 2) {
		const T s0 = blockSum256(acc0, red);
		if (threadIdx.x == 0) partsThin[blockIdx.x] = s0;
	}
	const T s1 = blockSum256(acc1, red);
	if (threadIdx.x == 0) partsThin[(dotMode == 2 ? NPART : 0) + blockIdx.x] = s1;
}
// (a launch of its own, started without a gap, instead of a ticket taken by every workgroup of a kernel that lasts microseconds: as distFinishSums)
template <typename T>
__global__ __launch_bounds__(TPB) void thinFinishKernel(const T* __restrict__ partsThin, int groups, int nsets, T* totals, const int* __restrict__ doneFlag) {
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	for (int k = 0; k < nsets; ++k) {
		T acc = T(0);
		for (int i = threadIdx.x; i < groups; i += TPB) acc += partsThin[(nsets == 2 ? k * NPART : 0) + i];
		const T s = blockSum256(acc, red);
		if (threadIdx.x == 0) totals[k] += s;
	}
}

// whether the thin form serves this matrix now (the remote block's kernel configuration is settled first: it decides the lanes per row)
static int thinUsable(smm_hip_dist_csr* D, hipStream_t s, bool* usable) {
	*usable = false;
	if (D->nThin <= 0 || D->chunks != 1) return SMM_HIP_OK;
	SMM_TRY(ensureCsrReady(D->aRem, s, true));
	*usable = D->aRem->lanes() == 1;
	return SMM_HIP_OK;
}
// out[listed rows] (+|-)= A_rem . ext, the launch's share of the dot products added to the totals the local launch left in `parts`; form: the
// halo of `ext` is formed here (fused ConjugateGradient), not read
template <typename T>
static int launchThinRemote(smm_hip_dist_csr* D, const T* ext, bool subtract, T* out, int dotMode, const T* w1, T* parts, const int* doneFlag, hipStream_t s,
                            const ThinForm<T>* form = nullptr) {
	const int g = std::max(1, std::min(THIN_MAX_GRID, (D->nThin + TPB - 1) / TPB));
	const int* startRem = static_cast<const int*>(D->arrays[0]) + D->nLocal + 1;
	if (form) {
		thinRemoteKernel<T, true><<<g, TPB, 0, s>>>(D->nThin, D->thinRows, startRem, static_cast<const int*>(D->arrays[4]), static_cast<const T*>(D->arrays[5]), ext,
		                                           subtract ? 1 : 0, out, dotMode, w1, static_cast<T*>(D->partsThin), doneFlag, *form);
	} else {
		thinRemoteKernel<T, false><<<g, TPB, 0, s>>>(D->nThin, D->thinRows, startRem, static_cast<const int*>(D->arrays[4]), static_cast<const T*>(D->arrays[5]), ext,
		                                            subtract ? 1 : 0, out, dotMode, w1, static_cast<T*>(D->partsThin), doneFlag, ThinForm<T>{});
	}
	if (dotMode) thinFinishKernel<T><<<1, TPB, 0, s>>>(static_cast<const T*>(D->partsThin), g, dotMode == 2 ? 2 : 1, parts + PARTS_TOTALS, doneFlag);
	SMM_HIP_TRY(hipGetLastError());
	D->totalsFinal = dotMode != 0;
	++D->matvecsThin;
	return SMM_HIP_OK;
}

// locFlags: flags of the LOCAL block's launch beyond those formed here (SPMV_HALF_TILES: ConjugateGradient's launches, smm_internal.h)
template <typename T>
static int distMatvecCompute(smm_hip_dist_csr* D, T* ext, int op, const T* lhs, T* out, int dotMode, const T* w1, T* parts, const int* doneFlag, hipStream_t s,
                             const T* jacobiDiag = nullptr, int slotPoint = -1, int locFlags = 0) {
	const T* own = ext + D->ownOffset;
	auto& pend = D->pending;
	const bool exchange = pend.active;
	pend.active = false;
	const int finish = dotMode && !D->p2p ? SPMV_FINISH : 0;  // (peer to peer: the slot kernel adds the partials itself)
	// jacobiDiag (op must be SMM_OP_ASSIGN): out = (A x) / diag with the division folded into the launch that completes a row -- the
	// local block's when nothing is remote, else the remote block's epilogue ("add, then divide"): the loop then has the kernel count
	// of the unpreconditioned one, and the dot products of the divided vector ride in the same epilogue
	if (D->remEmpty && !exchange) {
		if (jacobiDiag) return launchSpmv<T>(D->aLoc, op, jacobiDiag, own, out, dotMode, w1, parts, doneFlag, s, finish | SPMV_DIV_LHS | locFlags);
		return launchSpmv<T>(D->aLoc, op, lhs, own, out, dotMode, w1, parts, doneFlag, s, finish | locFlags);
	}
	// ONE launch for both halves where the blocks allow it (smm_spmv_split.hip; r06): the local half of the workgroup's rows, a bounded wait for the
	// word this exchange raises, the remote half -- the arithmetic of the two launches below, bit for bit, without the second ramp and tail, the
	// second pass over out[] and the kernel boundary behind the exchange
	if (exchange && pend.landSeq != 0 && !D->remEmpty) {
		SMM_TRY(ensureCsrReady(D->aLoc, s, true));
		SMM_TRY(ensureCsrReady(D->aRem, s, true));
		// peer-to-peer scalars: the launch finishes its dot products and runs their reduction point in its last workgroup (slotPoint: which one)
		P2PSlotArgs slots{};
		const bool fuseSlots = D->p2p && dotMode && slotPoint >= 0;
		if (fuseSlots) {
			P2PState* P = D->p2p;
			slots.peers = P->peers;
			slots.world = P->world;
			slots.me = P->rank;
			slots.point = slotPoint;
			slots.count = dotMode == 2 ? 2 : 1;
			slots.seq = P->redSeq[slotPoint] + 1;
			slots.ticks = P->ticks;
		}
		const int st = launchSpmvSplit<T>(D->aLoc, D->aRem, op, lhs, jacobiDiag, own, ext, out, dotMode, w1, parts, doneFlag,
		                                  (fuseSlots ? SPMV_FINISH : finish) | (jacobiDiag ? SPMV_ADD_DIV : 0), D->splitSync + pend.kind, pend.landSeq, splitErrWord(D),
		                                  p2pTicks(), s, fuseSlots ? &slots : nullptr, D->splitSumsLdsMax, D->splitSync + P2P_KINDS + 1);
		if (st == SMM_HIP_OK) {
			if (fuseSlots) {
				++D->p2p->redSeq[slotPoint];
				D->reducedInKernel = true;
			}
			++D->matvecsSplit;
			return SMM_HIP_OK;  // (the kernel itself waited for the halo: `s` needs no event of the communicator's stream)
		}
		if (st < 0) return st;
	}
	// a THIN remote block (at most an eighth of the rows hold a remote entry; one lane per row; no Jacobi division in the epilogue): the local
	// launch takes the dot products and finishes them, the remote half is a launch over the listed rows that adds its share to the totals
	bool thin = false;
	if (!jacobiDiag) SMM_TRY(thinUsable(D, s, &thin));
	if (thin) {
		SMM_TRY(launchSpmv<T>(D->aLoc, op, lhs, own, out, dotMode, w1, parts, doneFlag, s,
		                      (dotMode ? SPMV_FINISH : 0) | (exchange && pend.async ? SPMV_LEAVE_ROOM : 0) | locFlags));
		if (exchange && pend.landed[0]) {
			profWaitWaiting(pend.waitSlot[0], s);
			SMM_HIP_TRY(hipStreamWaitEvent(s, pend.landed[0], 0));
		}
		return launchThinRemote<T>(D, ext, op == SMM_OP_SUB, out, dotMode, w1, parts, doneFlag, s);
	}
	if (exchange && !D->remEmpty) ++D->matvecsTwo;
	// the local block runs while the halo is in flight; the exchange is itself a kernel (a few workgroups per peer), and the persistent
	// SpMV grid would otherwise take every workgroup slot of the chip until it ends: it leaves one CU per XCD's worth free
	SMM_TRY(launchSpmv<T>(D->aLoc, op, lhs, own, out, 0, nullptr, nullptr, doneFlag, s, (exchange && pend.async ? SPMV_LEAVE_ROOM : 0) | locFlags));
	const int remOp = op == SMM_OP_SUB ? SMM_OP_SUB : SMM_OP_ADD;
	if (exchange && D->chunks > 1 && !D->p2p) {
		// part k of A_rem reads only piece k and starts as soon as THAT piece has landed: out = ((A_loc x + A_rem,0 x) + A_rem,1 x) + ... -- each
		// part a row sum of its own, added in piece order (deterministic; differs from the one-piece form only in where the row sum is cut)
		const int K = D->chunks;
		for (int k = 0; k < K; ++k) {
			if (pend.landed[k]) {
				profWaitWaiting(pend.waitSlot[k], s);
				SMM_HIP_TRY(hipStreamWaitEvent(s, pend.landed[k], 0));
			}
			const bool last = k == K - 1;
			if (!last) {
				if (D->nnzRemK[static_cast<size_t>(k)] == 0) continue;  // nothing of A_rem reads this piece (the last part always runs: it carries the epilogue)
				SMM_TRY(launchSpmv<T>(D->aRemK[static_cast<size_t>(k)], remOp, out, ext, out, 0, nullptr, nullptr, doneFlag, s, 0));
			} else if (jacobiDiag) {
				return launchSpmv<T>(D->aRemK[static_cast<size_t>(k)], SMM_OP_ADD, out, ext, out, dotMode, w1, parts, doneFlag, s, finish | SPMV_ADD_DIV, jacobiDiag);
			} else {
				return launchSpmv<T>(D->aRemK[static_cast<size_t>(k)], remOp, out, ext, out, dotMode, w1, parts, doneFlag, s, finish);
			}
		}
	} else if (exchange && pend.landed[0]) {
		profWaitWaiting(pend.waitSlot[0], s);  // (profiling on: how long A_rem waits for the halo after A_loc has ended -- the exposed part of the exchange)
		SMM_HIP_TRY(hipStreamWaitEvent(s, pend.landed[0], 0));
	}
	if (jacobiDiag) return launchSpmv<T>(D->aRem, SMM_OP_ADD, out, ext, out, dotMode, w1, parts, doneFlag, s, finish | SPMV_ADD_DIV, jacobiDiag);
	return launchSpmv<T>(D->aRem, remOp, out, ext, out, dotMode, w1, parts, doneFlag, s, finish);
}

// both halves back to back (the set-up SpMVs, the stand-alone distributed SpMV)
template <typename T>
static int distMatvec(smm_hip_dist_csr* D, T* ext, int kind, int op, const T* lhs, T* out, int dotMode, const T* w1, T* parts, const int* doneFlag, hipStream_t s,
                      const T* jacobiDiag = nullptr) {
	SMM_TRY(distExchangeBegin<T>(D, ext, kind, s));
	return distMatvecCompute<T>(D, ext, op, lhs, out, dotMode, w1, parts, doneFlag, s, jacobiDiag);
}

// all-reduce of the totals of a finishing buffer on the side stream; *joined = event the caller's stream must wait for (null when
// nothing is pending)
template <typename T>
static int allreduceTotals(smm_hip_dist_csr* D, T* parts, int count, hipStream_t s, hipEvent_t* joined, int point = 0, const int* doneFlag = nullptr) {
	smm_hip_comm* c = D->comm;
	*joined = nullptr;
	const bool finished = D->totalsFinal;  // (the thin form: the rank's totals are in place)
	D->totalsFinal = false;
	if (D->reducedInKernel) {  // (the one-launch SpMV in front ran this point in its last workgroup)
		D->reducedInKernel = false;
		return SMM_HIP_OK;
	}
	if (c->kind == SMM_COMM_SELF && !D->p2p) return SMM_HIP_OK;  // (one rank with the slots: the lab's SMM_HIP_LAB_SELF_P2P)
	T* totals = parts + PARTS_TOTALS;
	// peer to peer: ONE single-workgroup kernel on the solver's own stream writes this rank's totals into every rank's slot, waits for all
	// slots of this sequence number and adds them in rank order (smm_p2p.h) -- no collective launch, no cross-stream events
	if (D->p2p) return p2pAllreduceLaunch<T>(D, point, totals, count, doneFlag, s, finished ? nullptr : parts);  // (else it adds the rank's partials itself: nobody finished them)
	if (c->kind == SMM_COMM_HOST) return commAllreduce<T>(c, totals, count, s);
	noteStream(c->stream);
	SMM_TRY(orderAfter(c, s, c->stream));
	SMM_TRY(commAllreduce<T>(c, totals, count, c->stream));
	*joined = takeEvent(c);
	SMM_HIP_TRY(hipEventRecord(*joined, c->stream));
	return SMM_HIP_OK;
}

static int join(hipStream_t s, hipEvent_t e) {
	if (e) SMM_HIP_TRY(hipStreamWaitEvent(s, e, 0));
	return SMM_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// kernels of the loops: each reads the all-reduced totals itself (no scalar launches)
// ---------------------------------------------------------------------------------------------------------
// totals[0] = a.b, totals[1] = a.c (nsets 2); NPART workgroups, finished by the last one
// The totals of a reduction whose partials an UPDATE kernel left behind (distDots, distBicgR, distCgR): one workgroup adds the NPART
// partials of each quantity in the order the fused hand-off uses (lastBlockSums, smm_device.h: i = t, t + 256, ...; then blockSum256) --
// the same bits -- and leaves them where the all-reduce expects them.  r01-r04 had the LAST workgroup of the update kernel do this: every
// workgroup then ends with two dependent trips to memory (publish the partial, take a ticket), which for a 5 us kernel over a rank's
// 1.25 M rows was most of its time (distBicgR 28.6 us; this launch: 4.7-6.3 us, started without a gap; profiles/r05/rank_loop_gaps.txt).
// The SpMV kernels keep the fused hand-off (SPMV_FINISH): there it costs what this launch would.
template <typename T>
__global__ __launch_bounds__(TPB) void distFinishSums(const T* __restrict__ parts, int nsets, T* totals, const int* __restrict__ doneFlag) {
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	for (int k = 0; k < nsets; ++k) {
		T acc = T(0);
		for (int i = threadIdx.x; i < NPART; i += TPB) acc += parts[k * NPART + i];
		const T s = blockSum256(acc, red);
		if (threadIdx.x == 0) totals[k] = s;
	}
}

template <typename T>
__global__ __launch_bounds__(TPB) void distDots(int n, const T* a, const T* b, const T* cvec, int nsets, T* parts, const int* __restrict__ doneFlag) {
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	T acc0 = T(0), acc1 = T(0);
	if (nsets == 2) {
		const T* const in[3] = {a, b, cvec};
		streamMap<T, false, 3, 0>(n, in, nullptr, [&](const T(&v)[3], T(&)[1]) {
			acc0 += v[0] * v[1];
			acc1 += v[0] * v[2];
		});
	} else {
		const T* const in[2] = {a, b};
		streamMap<T, false, 2, 0>(n, in, nullptr, [&](const T(&v)[2], T(&)[1]) { acc0 += v[0] * v[1]; });
	}
	const T s0 = blockSum256(acc0, red);
	if (threadIdx.x == 0) parts[blockIdx.x] = s0;
	if (nsets == 2) {
		const T s1 = blockSum256(acc1, red);
		if (threadIdx.x == 0) parts[NPART + blockIdx.x] = s1;
	}
}

template <typename T>
__global__ void distBicgInit(const T* __restrict__ totals, DistScal<T>* sc) {
	sc->rrPing[0] = totals[0];  // rr0 = r.r0, ref:2231
	sc->res = T(0);
	sc->iters = 0;
	sc->done = 0;
	sc->status = SMM_SOLVER_SUCCESS;
}

// an element-wise map over up to four runs of rows (the boundary rows some peer receives / the rest: halo first)
template <typename T, int NIN, int NOUT, bool NT = false, typename F>
__device__ __forceinline__ void rangesMap(const RowRanges& rg, const T* const* in, T* const* out, F&& f) {
	for (int i = 0; i < rg.n; ++i) {
		const T* in2[NIN];
		T* out2[NOUT];
#pragma unroll
		for (int k = 0; k < NIN; ++k) in2[k] = in[k] + rg.lo[i];
#pragma unroll
		for (int k = 0; k < NOUT; ++k) out2[k] = out[k] + rg.lo[i];
		streamMap<T, NT, NIN, NOUT>(rg.hi[i] - rg.lo[i], in2, out2, f);
	}
}

// alpha = rr0 / (ap.r0) ; s = -alpha ap + r   (ref:2243-2247).  book: this launch records alpha (the other launch of the pair -- boundary
// rows first, then the bulk -- forms the same alpha from the same operands and only uses it)
template <typename T>
__global__ __launch_bounds__(TPB) void distBicgS(RowRanges rg, int book, DistScal<T>* sc, int par, const T* __restrict__ totalsA, const T* ap, const T* r, T* sv) {
	if (sc->done) return;
	const T alpha = sc->rrPing[par] / totalsA[0];
	if (book && blockIdx.x == 0 && threadIdx.x == 0) sc->alpha = alpha;
	const T* const in[2] = {ap, r};
	T* const out[1] = {sv};
	rangesMap<T, 2, 1>(rg, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(-alpha, v[0], v[1]); });
}

// omega = (as.s) / (as.as) ; r = -omega as + s ; local ||r||^2 and r.r0   (ref:2259-2261, 2265)
template <typename T>
__global__ __launch_bounds__(TPB) void distBicgR(int n, DistScal<T>* sc, const T* __restrict__ totalsB, const T* sv, const T* as, const T* r0, T* r,
                                                 T* partsC) {
	__shared__ T red[4];
	if (sc->done) return;
	const T omega = totalsB[1] / totalsB[0];
	if (blockIdx.x == 0 && threadIdx.x == 0) sc->omega = omega;
	T acc0 = T(0), acc1 = T(0);
	const T* const in[3] = {sv, as, r0};
	T* const out[1] = {r};
	streamMap<T, false, 3, 1>(n, in, out, [&](const T(&v)[3], T(&o)[1]) {
		const T ri = smmFma(-omega, v[1], v[0]);
		o[0] = ri;
		acc0 += ri * ri;
		acc1 += ri * v[2];
	});
	const T s0 = blockSum256(acc0, red);
	const T s1 = blockSum256(acc1, red);
	if (threadIdx.x == 0) {
		partsC[blockIdx.x] = s0;
		partsC[NPART + blockIdx.x] = s1;
	}
}

// The same with x = alpha p + (omega s + x) (ref:2264) folded in -- the single-GPU loop's bicgFusedXR: s is read once for both, one launch
// instead of two.  Taken when the scalars go through the per-rank slots (the reduction behind this kernel is then a single-workgroup launch on
// the solver's own stream: there is no side-stream collective for a separate x update to run beside).  Same expressions: same bits.
template <typename T>
__global__ __launch_bounds__(TPB) void distBicgRX(int n, DistScal<T>* sc, const T* __restrict__ totalsB, const T* sv, const T* as, const T* r0, const T* p, T* r,
                                                  T* x, T* partsC) {
	__shared__ T red[4];
	if (sc->done) return;
	const T omega = totalsB[1] / totalsB[0];
	const T alpha = sc->alpha;
	if (blockIdx.x == 0 && threadIdx.x == 0) sc->omega = omega;
	T acc0 = T(0), acc1 = T(0);
	const T* const in[5] = {sv, as, r0, p, x};
	T* const out[2] = {r, x};
	streamMap<T, false, 5, 2>(n, in, out, [&](const T(&v)[5], T(&o)[2]) {
		const T ri = smmFma(-omega, v[1], v[0]);
		o[0] = ri;
		o[1] = smmFma(alpha, v[3], smmFma(omega, v[0], v[4]));
		acc0 += ri * ri;
		acc1 += ri * v[2];
	});
	const T s0 = blockSum256(acc0, red);
	const T s1 = blockSum256(acc1, red);
	if (threadIdx.x == 0) {
		partsC[blockIdx.x] = s0;
		partsC[NPART + blockIdx.x] = s1;
	}
}

// x = alpha p + (omega s + x)   (ref:2264) -- independent of the all-reduce of ||r||^2, r.r0 and runs beside it
template <typename T>
__global__ __launch_bounds__(TPB) void distBicgX(int n, const DistScal<T>* __restrict__ sc, const T* p, const T* sv, T* x) {
	if (sc->done) return;
	const T alpha = sc->alpha, omega = sc->omega;
	const T* const in[3] = {sv, x, p};
	T* const out[1] = {x};
	streamMap<T, false, 3, 1>(n, in, out, [&](const T(&v)[3], T(&o)[1]) { o[0] = smmFma(alpha, v[2], smmFma(omega, v[0], v[1])); });
}

// resL2Norm, loop test, beta, p = beta (-omega ap + p) + r   (ref:2268-2277).  book: this launch does the scalar bookkeeping (once per pair)
template <typename T>
__global__ __launch_bounds__(TPB) void distBicgP(RowRanges rg, int book, DistScal<T>* sc, int par, const T* __restrict__ totalsC, T eps, const T* ap, const T* r, T* p) {
	if (sc->done) return;
	const T rr = totalsC[0];
	const T newRR0 = totalsC[1];
	const T res = sizeof(T) == 4 ? static_cast<T>(__fsqrt_rn(static_cast<float>(rr))) : static_cast<T>(__dsqrt_rn(static_cast<double>(rr)));
	const T alpha = sc->alpha, omega = sc->omega, rr0 = sc->rrPing[par];
	const bool leave = !(res > eps);
	if (book && blockIdx.x == 0 && threadIdx.x == 0) {
		sc->res = res;
		sc->rrPing[par ^ 1] = newRR0;
		sc->iters += 1;
		if (leave) sc->done = 1;
	}
	if (leave) return;
	const T beta = (newRR0 * alpha) / (rr0 * omega);  // ref:2271
	const T* const in[3] = {ap, p, r};
	T* const out[1] = {p};
	rangesMap<T, 3, 1>(rg, in, out, [&](const T(&v)[3], T(&o)[1]) { o[0] = smmFma(beta, smmFma(-omega, v[0], v[1]), v[2]); });
}

template <typename T>
__global__ void distCgInit(const T* __restrict__ totals, DistScal<T>* sc, T eps) {
	const T rr = totals[0];  // ref:2341
	sc->rrPing[0] = rr;
	sc->res = rr;
	sc->iters = 0;
	sc->done = 0;
	sc->status = SMM_SOLVER_MAX_ITERATIONS_REACHED;
	sc->flushIter = -1;
	if (eps * eps > rr) {  // ref:2342-2344: x stays untouched
		sc->done = 1;
		sc->status = SMM_SOLVER_SUCCESS;
	}
	sc->pad = sc->done;
}

// alpha = rr / (Ap.p) ; r = -alpha Ap + r ; local ||r||^2   (ref:2354-2375)
// (NT in the CG kernels: non-temporal loads / stores when the vectors cannot stay in the Infinity Cache anyway -- updateNT, as in smm_solvers.hip)
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void distCgR(int n, DistScal<T>* sc, int par, const T* __restrict__ totalsA, const T* Ap, T* r, T* partsC, int alphaSlot) {
	__shared__ T red[4];
	const int done = sc->done;
	if (blockIdx.x == 0 && threadIdx.x == 0) sc->pad = done;
	if (done) return;
	const T alpha = sc->rrPing[par] / totalsA[0];
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		sc->alpha = alpha;
		sc->alphaRing[alphaSlot] = alpha;
	}
	T acc = T(0);
	const T* const in[2] = {Ap, r};
	T* const out[1] = {r};
	streamMap<T, NT, 2, 1>(n, in, out, [&](const T(&v)[2], T(&o)[1]) {
		const T ri = smmFma(-alpha, v[0], v[1]);
		o[0] = ri;
		acc += ri * ri;
	});
	const T s0 = blockSum256(acc, red);
	if (threadIdx.x == 0) partsC[blockIdx.x] = s0;
}

// x = alpha p + xcur   (ref:2372)
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void distCgX(int n, const DistScal<T>* __restrict__ sc, const T* p, const T* xcur, T* x) {
	if (sc->done) return;
	const T alpha = sc->alpha;
	const T* const in[2] = {p, xcur};
	T* const out[1] = {x};
	streamMap<T, NT, 2, 1>(n, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(alpha, v[0], v[1]); });
}

// convergence test, beta, p = beta p + r   (ref:2377-2394).  book: as in distBicgP
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void distCgP(RowRanges rg, int book, DistScal<T>* sc, int par, const T* __restrict__ totalsC, T eps, T* p, const T* r) {
	if (sc->done) return;
	const T rrNew = totalsC[0];
	const T rrOld = sc->rrPing[par];
	const bool converged = eps * eps > rrNew;
	if (book && blockIdx.x == 0 && threadIdx.x == 0) {
		sc->iters += 1;
		sc->res = rrNew;
		if (converged) {
			sc->done = 1;
			sc->status = SMM_SOLVER_SUCCESS;
		} else {
			sc->rrPing[par ^ 1] = rrNew;
		}
	}
	if (converged) return;
	const T beta = rrNew / rrOld;
	const T* const in[2] = {p, r};
	T* const out[1] = {p};
	rangesMap<T, 2, 1, NT>(rg, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(beta, v[0], v[1]); });
}

// The deferred x update of the single-GPU loop (smm_solvers.hip, cgLazyXP) in the row-partitioned one: nothing inside the loop reads x
// (ref:2362-2366), so the last LAZY_M directions stay in a ring of halo-extended vectors and x is brought up to date every LAZY_M-th
// iteration, in the last planned one and by the launch that finds the iteration converged -- x = alpha_k p_k + (... + (alpha_{k-M+1}
// p_{k-M+1} + x)): the reference's roundings in the reference's order.  11 vector passes per iteration become 9.125.  Range-aware like
// distCgP (boundary rows first, then the bulk: each launch flushes its own rows).
template <typename T>
struct DistRing {
	T* p[LAZY_M + 1];  // the owned part of every ring vector
};

template <typename T, bool NT, int PENDING, bool PUPD>
__device__ __forceinline__ void distLazyFlush(const RowRanges& rg, const DistRing<T>& ring, int cur, const T (&alpha)[LAZY_M], T beta, const T* r, const T* xcur, T* x) {
	const T* in[PENDING + 2];
	in[0] = xcur;
#pragma unroll
	for (int k = 0; k < PENDING; ++k) in[1 + k] = ring.p[(cur + (LAZY_M + 1) - (PENDING - 1 - k)) % (LAZY_M + 1)];
	in[PENDING + 1] = PUPD ? r : xcur;  // (without the p update r is not read: any valid vector)
	T* out[2] = {x, ring.p[(cur + 1) % (LAZY_M + 1)]};
	rangesMap<T, PENDING + 2, PUPD ? 2 : 1, NT>(rg, in, out, [&](const T(&v)[PENDING + 2], T(&o)[PUPD ? 2 : 1]) {
		T xv = v[0];
#pragma unroll
		for (int k = 0; k < PENDING; ++k) xv = smmFma(alpha[LAZY_M - PENDING + k], v[1 + k], xv);  // oldest direction first, ref:2372
		o[0] = xv;
		if constexpr (PUPD) o[1] = smmFma(beta, v[PENDING], v[PENDING + 1]);
	});
}

// convergence test, beta, p_next = beta p_cur + r (ref:2377-2394); x completed when `flush` (host: every LAZY_M-th iteration and the last
// planned one) or when the iteration converged.  pending: directions not yet in x, this iteration's included.  book: as in distCgP
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void distCgLazyP(RowRanges rg, int book, DistScal<T>* sc, int par, const T* __restrict__ totalsC, T eps, DistRing<T> ring, int cur,
                                                   int pending, int flush, int alphaSlot, const T* r, const T* xcur, T* x) {
	if (sc->pad) return;  // (`done` as distCgR found it: workgroup 0 of THIS launch may be setting it while others start -- and their rows still need the flush)
	const T rrNew = totalsC[0];
	const T rrOld = sc->rrPing[par];
	const bool converged = eps * eps > rrNew;
	T alpha[LAZY_M];  // alpha[LAZY_M - 1] = this iteration's, alpha[LAZY_M - 2] the one before, ...
#pragma unroll
	for (int k = 0; k < LAZY_M; ++k) alpha[LAZY_M - 1 - k] = sc->alphaRing[(alphaSlot + LAZY_M - k) % LAZY_M];
	if (book && blockIdx.x == 0 && threadIdx.x == 0) {
		sc->iters += 1;
		sc->res = rrNew;
		if (converged) {
			sc->done = 1;
			sc->status = SMM_SOLVER_SUCCESS;
		} else {
			sc->rrPing[par ^ 1] = rrNew;
		}
	}
	const T beta = rrNew / rrOld;
	if (!(flush || converged)) {
		const T* const in[2] = {ring.p[cur], r};
		T* const out[1] = {ring.p[(cur + 1) % (LAZY_M + 1)]};
		rangesMap<T, 2, 1, NT>(rg, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(beta, v[0], v[1]); });
		return;
	}
	static_assert(LAZY_M == 8, "the cases below");
#define SMM_DIST_LAZY_CASE(P)                                                                  \
	case P:                                                                                    \
		if (converged) distLazyFlush<T, NT, P, false>(rg, ring, cur, alpha, beta, r, xcur, x);     \
		else distLazyFlush<T, NT, P, true>(rg, ring, cur, alpha, beta, r, xcur, x);                \
		break;
	switch (pending) {
		SMM_DIST_LAZY_CASE(1)
		SMM_DIST_LAZY_CASE(2)
		SMM_DIST_LAZY_CASE(3)
		SMM_DIST_LAZY_CASE(4)
		SMM_DIST_LAZY_CASE(5)
		SMM_DIST_LAZY_CASE(6)
		SMM_DIST_LAZY_CASE(7)
		SMM_DIST_LAZY_CASE(8)
	default: break;
	}
#undef SMM_DIST_LAZY_CASE
}

// what is left of distCgLazyP once the direction is formed in the SpMV: x, brought up to date when scheduled (every LAZY_M-th iteration)
// or when the SpMV launch in front found its predecessor converged (flushIter == iter) -- cgLazyFlushOnly of smm_solvers.hip
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void distCgFlushOnly(RowRanges rg, const DistScal<T>* __restrict__ sc, DistRing<T> ring, int cur, int pending, int scheduled, int iter,
                                                       int alphaSlot, const T* xcur, T* x) {
	const bool converged = sc->flushIter == iter;
	if (!converged && (!scheduled || sc->done)) return;
	T alpha[LAZY_M];
#pragma unroll
	for (int k = 0; k < LAZY_M; ++k) alpha[LAZY_M - 1 - k] = sc->alphaRing[(alphaSlot + LAZY_M - k) % LAZY_M];
	static_assert(LAZY_M == 8, "the cases below");
	switch (pending) {
	case 1: distLazyFlush<T, NT, 1, false>(rg, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 2: distLazyFlush<T, NT, 2, false>(rg, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 3: distLazyFlush<T, NT, 3, false>(rg, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 4: distLazyFlush<T, NT, 4, false>(rg, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 5: distLazyFlush<T, NT, 5, false>(rg, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 6: distLazyFlush<T, NT, 6, false>(rg, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 7: distLazyFlush<T, NT, 7, false>(rg, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 8: distLazyFlush<T, NT, 8, false>(rg, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	default: break;
	}
}

#define SMM_DIST_UPDATE(KERNEL, NTFLAG, GRID, STREAM, ...)                       \
	do {                                                                        \
		if (NTFLAG) KERNEL<T, true><<<(GRID), TPB, 0, (STREAM)>>>(__VA_ARGS__);  \
		else KERNEL<T, false><<<(GRID), TPB, 0, (STREAM)>>>(__VA_ARGS__);        \
	} while (0)

static int gridFor(long long n) { return static_cast<int>(std::max<long long>(1, std::min<long long>((n + TPB - 1) / TPB, NPART))); }
// Leaving the loop early must be the SAME decision on every rank (a rank that stops issuing iterations while another goes on leaves the
// other one alone in its next collective).  The `done` flag is formed from all-reduced -- hence identical -- numbers at the same
// iteration on every rank, so a BLOCKING read of it at fixed iteration numbers is consistent; the asynchronous mailbox of the single-GPU
// loops (whose answer depends on how far the host has run ahead) is not.  One pipeline drain every CHECK_EVERY iterations.
constexpr int CHECK_EVERY = 16;
static int readDone(smm_hip_dist_csr* D, const int* d_done, hipStream_t s, int* done) {
	unsigned long long err = 0;
	SMM_HIP_TRY(hipMemcpyAsync(done, d_done, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_TRY(p2pPostErrRead(D, &err, s));
	SMM_TRY(boundedSync(D->comm, s));
	return p2pFailIf(D, err);
}

// A vector update whose result feeds the next SpMV, halo first: the boundary rows some peer receives in a small launch, the exchange posted
// right behind it, the bulk in a second launch -- the same expressions on the same operands, the same bits as one launch.
// LAUNCH(ranges, book) enqueues the update kernel on `s` for those rows; `book`: that launch does the kernel's scalar bookkeeping.
template <typename T, typename Launch>
static int updateThenExchange(smm_hip_dist_csr* D, T* ext, int kind, hipStream_t s, Launch&& launch) {
	if (D->haloFirst) {
		launch(D->boundary, 0);
		SMM_TRY(distExchangeBegin<T>(D, ext, kind, s));
		launch(D->bulk, 1);
	} else {
		RowRanges all{};
		all.n = 1;
		all.hi[0] = D->nLocal;
		launch(all, 1);
		SMM_TRY(distExchangeBegin<T>(D, ext, kind, s));
	}
	return SMM_HIP_OK;
}

// an exchange that was posted but whose SpMV never runs (the loop was left): complete in itself on the side stream (the peer-to-peer land
// kernel empties the landing area and acknowledges there); the solver's stream only has to stay behind it
template <typename T>
static int distExchangeDrain(smm_hip_dist_csr* D, T* ext, hipStream_t s) {
	auto& pend = D->pending;
	if (!pend.active) return SMM_HIP_OK;
	pend.active = false;
	(void)ext;
	for (hipEvent_t e : pend.landed) {
		if (e) SMM_HIP_TRY(hipStreamWaitEvent(s, e, 0));  // (the next use of `ext` on `s` must not overtake the receive)
	}
	return SMM_HIP_OK;
}

template <typename T>
static int checkDist(const smm_hip_dist_csr* D, const char* who) {
	if (!D || D->dtype != dtypeOf<T>()) {
		setError("%s: null distributed matrix or dtype mismatch", who);
		return SMM_HIP_ERR_INVALID;
	}
	return SMM_HIP_OK;
}

template <typename T>
static int distSpmv(smm_hip_dist_csr* D, int op, const T* lhs, const T* xOwn, T* out, hipStream_t s) {
	SMM_TRY(checkDist<T>(D, "dist_spmv"));
	SMM_TRY(ensureInit());
	if (op < SMM_OP_ASSIGN || op > SMM_OP_SUB || (D->nLocal > 0 && (!xOwn || !out || (op != SMM_OP_ASSIGN && !lhs)))) {
		setError("dist_spmv: bad op or null vector");
		return SMM_HIP_ERR_INVALID;
	}
	T* xExt = static_cast<T*>(D->xExt);
	if (D->nLocal > 0) SMM_HIP_TRY(hipMemcpyAsync(xExt + D->ownOffset, xOwn, sizeof(T) * D->nLocal, hipMemcpyDeviceToDevice, s));
	return distMatvec<T>(D, xExt, 2, op, lhs, out, 0, nullptr, nullptr, nullptr, s);
}

template <typename T>
static int distBicgstab(smm_hip_dist_csr* D, const T* b, T* x, int maxIterations, T eps, const smm_hip_precond* M, hipStream_t s, int* status,
                        int* iterations, T* resnorm) {
	SMM_TRY(checkDist<T>(D, "dist_bicgstab"));
	SMM_TRY(ensureInit());
	const int n = D->nLocal;
	if (n > 0 && (!b || !x)) {
		setError("dist_bicgstab: null vector");
		return SMM_HIP_ERR_INVALID;
	}
	// block-Jacobi by rank: M is this rank's preconditioner of its diagonal block A_loc, applied without
	// communication; with one rank it is the single-GPU preconditioner, with more a weaker one (iteration counts differ)
	const bool pre = M != nullptr && M->kind != SMM_PRECOND_NONE;
	if (pre && (M->a != D->aLoc || M->kind == SMM_PRECOND_IC0)) {
		setError("dist_bicgstab: preconditioner must be JACOBI / ILU0 / SGS / BLOCK_ILU0 / BLOCK_SGS of the local diagonal block (smm_hip_dist_csr_local_block)");
		return SMM_HIP_ERR_INVALID;
	}
	D->reducedInKernel = false;  // (a call that failed between an SpMV and its reduction point must not leave the mark behind)
	D->totalsFinal = false;
	maxIterations = std::min(maxIterations, D->nGlobal);  // ref:2200
	if (maxIterations == -1) maxIterations = D->nGlobal;  // ref:2201-2203
	// many SpMVs ahead: both local blocks may take the index-free family (each rank decides for its own blocks; no collective involved)
	// (the blocks' own configuration first: a block nobody has multiplied with yet still carries the handle's initial word, and the adoption
	// below only moves a block that is on the STREAM family -- r05: the FIRST solve of a distributed matrix ran on the CSR kernels)
	if (D->aLoc) SMM_TRY(ensureCsrReady(D->aLoc, s, true));
	if (D->aRem) SMM_TRY(ensureCsrReady(D->aRem, s, true));
	if (D->aLoc) SMM_TRY(adoptPatternForSolver(D->aLoc, maxIterations, s));
	if (D->aRem) SMM_TRY(adoptPatternForSolver(D->aRem, maxIterations, s));
	for (smm_hip_csr* piece : D->aRemK) SMM_TRY(adoptPatternForSolver(piece, maxIterations, s));
	T *r = static_cast<T*>(D->r), *r0 = static_cast<T*>(D->r0), *ap = static_cast<T*>(D->ap), *as = static_cast<T*>(D->as);
	T *pExt = static_cast<T*>(D->pExt), *sExt = static_cast<T*>(D->sExt), *xExt = static_cast<T*>(D->xExt);
	T *p = pExt + D->ownOffset, *sv = sExt + D->ownOffset;
	T *partsA = static_cast<T*>(D->partsA), *partsB = static_cast<T*>(D->partsB), *partsC = static_cast<T*>(D->partsC);
	auto* sc = static_cast<DistScal<T>*>(D->sc);
	if (pre && !D->scratch) SMM_TRY(devAlloc(&D->scratch, static_cast<size_t>(std::max(1, n)) * sizeof(T)));
	T* scratch = static_cast<T*>(D->scratch);
	hipEvent_t ev = nullptr;
	// Jacobi: x = rhs / diag is folded into the rows of the SpMV (smm_solvers.hip does the same on one GPU): no apply launch, no dot
	// launch -- 8 kernels per iteration like the unpreconditioned loop
	const T* jacobiDiag = pre && M->kind == SMM_PRECOND_JACOBI ? static_cast<const T*>(M->d_values) : nullptr;

	// r = b - A x (ref:2215) [; r = M^-1 r, ref:2217-2224]
	if (n > 0) SMM_HIP_TRY(hipMemcpyAsync(xExt + D->ownOffset, x, sizeof(T) * n, hipMemcpyDeviceToDevice, s));
	if (pre) {
		SMM_TRY(distMatvec<T>(D, xExt, 2, SMM_OP_SUB, b, scratch, 0, nullptr, nullptr, nullptr, s));
		SMM_TRY(precondApplyDev<T>(M, scratch, r, nullptr, s));
	} else {
		SMM_TRY(distMatvec<T>(D, xExt, 2, SMM_OP_SUB, b, r, 0, nullptr, nullptr, nullptr, s));
	}
	SMM_TRY(launchCopy2<T>(n, r, r0, p, s));                              // r0 = p = r, ref:2225-2226
	SMM_TRY(distExchangeBegin<T>(D, pExt, 0, s));                         // the halo of p travels while r.r0 is reduced
	distDots<T><<<NPART, TPB, 0, s>>>(n, r, r, nullptr, 1, partsC, nullptr);  // r.r0 with r0 == r, ref:2231
	if (!D->p2p) distFinishSums<T><<<1, TPB, 0, s>>>(partsC, 1, partsC + PARTS_TOTALS, nullptr);
	SMM_TRY(allreduceTotals<T>(D, partsC, 1, s, &ev, 2));
	SMM_TRY(join(s, ev));
	distBicgInit<T><<<1, 1, 0, s>>>(partsC + PARTS_TOTALS, sc);

	const int* doneFlag = &sc->done;
	const int planned = std::max(1, maxIterations);  // do { } while: the body always runs once (ref:2232, 2277)
	for (int i = 0; i < planned; ++i) {
		if (i > 0 && i % CHECK_EVERY == 0) {
			int seen = 0;
			SMM_TRY(readDone(D, doneFlag, s, &seen));
			if (seen) break;
		}
		const int par = i & 1;
		// ap = [M^-1] A p ; ap.r0 (ref:2233-2243) -- the exchange of p was posted behind its update
		if (jacobiDiag) {
			SMM_TRY(distMatvecCompute<T>(D, pExt, SMM_OP_ASSIGN, nullptr, ap, 1, r0, partsA, doneFlag, s, jacobiDiag, 0));
		} else if (pre) {
			SMM_TRY(distMatvecCompute<T>(D, pExt, SMM_OP_ASSIGN, nullptr, scratch, 0, nullptr, nullptr, doneFlag, s));
			SMM_TRY(precondApplyDev<T>(M, scratch, ap, doneFlag, s));
			distDots<T><<<NPART, TPB, 0, s>>>(n, ap, r0, nullptr, 1, partsA, doneFlag);
			if (!D->p2p) distFinishSums<T><<<1, TPB, 0, s>>>(partsA, 1, partsA + PARTS_TOTALS, doneFlag);
		} else {
			SMM_TRY(distMatvecCompute<T>(D, pExt, SMM_OP_ASSIGN, nullptr, ap, 1, r0, partsA, doneFlag, s, nullptr, 0));
		}
		SMM_TRY(allreduceTotals<T>(D, partsA, 1, s, &ev, 0, doneFlag));
		SMM_TRY(join(s, ev));
		SMM_TRY((updateThenExchange<T>(D, sExt, 1, s, [&](const RowRanges& rg, int book) {
			distBicgS<T><<<gridFor(rg.rows()), TPB, 0, s>>>(rg, book, sc, par, partsA + PARTS_TOTALS, ap, r, sv);
		})));
		// as = [M^-1] A s ; as.as, as.s (ref:2249-2261)
		if (jacobiDiag) {
			SMM_TRY(distMatvecCompute<T>(D, sExt, SMM_OP_ASSIGN, nullptr, as, 2, sv, partsB, doneFlag, s, jacobiDiag, 1));
		} else if (pre) {
			SMM_TRY(distMatvecCompute<T>(D, sExt, SMM_OP_ASSIGN, nullptr, scratch, 0, nullptr, nullptr, doneFlag, s));
			SMM_TRY(precondApplyDev<T>(M, scratch, as, doneFlag, s));
			distDots<T><<<NPART, TPB, 0, s>>>(n, as, as, sv, 2, partsB, doneFlag);
			if (!D->p2p) distFinishSums<T><<<1, TPB, 0, s>>>(partsB, 2, partsB + PARTS_TOTALS, doneFlag);
		} else {
			SMM_TRY(distMatvecCompute<T>(D, sExt, SMM_OP_ASSIGN, nullptr, as, 2, sv, partsB, doneFlag, s, nullptr, 1));
		}
		SMM_TRY(allreduceTotals<T>(D, partsB, 2, s, &ev, 1, doneFlag));
		SMM_TRY(join(s, ev));
		if (D->p2p) {
			distBicgRX<T><<<NPART, TPB, 0, s>>>(n, sc, partsB + PARTS_TOTALS, sv, as, r0, p, r, x, partsC);  // (r and x in one pass over s)
			SMM_TRY(allreduceTotals<T>(D, partsC, 2, s, &ev, 2, doneFlag));
		} else {
			distBicgR<T><<<NPART, TPB, 0, s>>>(n, sc, partsB + PARTS_TOTALS, sv, as, r0, r, partsC);
			distFinishSums<T><<<1, TPB, 0, s>>>(partsC, 2, partsC + PARTS_TOTALS, doneFlag);
			SMM_TRY(allreduceTotals<T>(D, partsC, 2, s, &ev, 2, doneFlag));  // (the communicator's collectives: on the side stream ...
			distBicgX<T><<<gridFor(n), TPB, 0, s>>>(n, sc, p, sv, x);         // ... while x is updated)
		}
		SMM_TRY(join(s, ev));
		if (i + 1 < planned) {
			SMM_TRY((updateThenExchange<T>(D, pExt, 0, s, [&](const RowRanges& rg, int book) {
				distBicgP<T><<<gridFor(rg.rows()), TPB, 0, s>>>(rg, book, sc, par, partsC + PARTS_TOTALS, eps, ap, r, p);
			})));
		} else {
			RowRanges all{};
			all.n = 1;
			all.hi[0] = n;
			distBicgP<T><<<gridFor(n), TPB, 0, s>>>(all, 1, sc, par, partsC + PARTS_TOTALS, eps, ap, r, p);  // the last pass: no SpMV follows
		}
	}
	SMM_TRY(distExchangeDrain<T>(D, pExt, s));  // (left early: the exchange posted for the pass that never ran)
	SMM_HIP_TRY(hipGetLastError());
	DistScal<T> h;
	unsigned long long p2pErr = 0;
	SMM_HIP_TRY(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, s));
	SMM_TRY(p2pPostErrRead(D, &p2pErr, s));
	SMM_TRY(boundedSync(D->comm, s));
	SMM_TRY(p2pFailIf(D, p2pErr));
	if (status) *status = h.iters > maxIterations ? SMM_SOLVER_MAX_ITERATIONS_REACHED : SMM_SOLVER_SUCCESS;  // ref:2279-2282
	if (iterations) *iterations = h.iters;
	if (resnorm) *resnorm = h.res;
	return pre ? precondTakeError(M, s) : SMM_HIP_OK;
}

template <typename T>
static int distCg(smm_hip_dist_csr* D, const T* b, const T* x0, T* x, int maxIterations, T eps, hipStream_t s, int* status, int* iterations,
                  T* resnorm2) {
	SMM_TRY(checkDist<T>(D, "dist_cg"));
	SMM_TRY(ensureInit());
	const int n = D->nLocal;
	if (n > 0 && (!b || !x0 || !x)) {
		setError("dist_cg: null vector");
		return SMM_HIP_ERR_INVALID;
	}
	D->reducedInKernel = false;
	D->totalsFinal = false;
	if (maxIterations == -1) maxIterations = D->nGlobal;  // ref:2345-2347 (no clamp otherwise)
	// (the blocks' own configuration first: a block nobody has multiplied with yet still carries the handle's initial word, and the adoption
	// below only moves a block that is on the STREAM family -- r05: the FIRST solve of a distributed matrix ran on the CSR kernels)
	if (D->aLoc) SMM_TRY(ensureCsrReady(D->aLoc, s, true));
	if (D->aRem) SMM_TRY(ensureCsrReady(D->aRem, s, true));
	if (D->aLoc) SMM_TRY(adoptPatternForSolver(D->aLoc, maxIterations, s));
	if (D->aRem) SMM_TRY(adoptPatternForSolver(D->aRem, maxIterations, s));
	for (smm_hip_csr* piece : D->aRemK) SMM_TRY(adoptPatternForSolver(piece, maxIterations, s));
	T *r = static_cast<T*>(D->r), *ap = static_cast<T*>(D->ap);
	T *pExt = static_cast<T*>(D->pExt), *xExt = static_cast<T*>(D->xExt);
	T* p = pExt + D->ownOffset;
	T *partsA = static_cast<T*>(D->partsA), *partsC = static_cast<T*>(D->partsC);
	auto* sc = static_cast<DistScal<T>*>(D->sc);
	hipEvent_t ev = nullptr;
	// vectors beyond the caches: x deferred through a ring of LAZY_M more direction vectors (distCgLazyP); when they cannot be had, the eager loop
	bool lazy = n > 0 && static_cast<long long>(n) * static_cast<long long>(sizeof(T)) >= cgLazyMinBytes();
	T* ringExt[LAZY_M + 1] = {pExt};
	DistRing<T> ring{};
	ring.p[0] = p;
	if (lazy) {
		const size_t eb = static_cast<size_t>(std::max(1, D->extLen)) * sizeof(T);
		for (int k = 0; k < LAZY_M && lazy; ++k) {
			if (!D->lazyExt[k]) {
				if (devAlloc(&D->lazyExt[k], eb) != SMM_HIP_OK) {
					(void)hipGetLastError();
					D->lazyExt[k] = nullptr;
					lazy = false;
					break;
				}
				SMM_HIP_TRY(hipMemsetAsync(D->lazyExt[k], 0, eb, s));  // (halo slots outside every recv segment are never read by A_rem; zero keeps them finite)
			}
			ringExt[k + 1] = static_cast<T*>(D->lazyExt[k]);
			ring.p[k + 1] = ringExt[k + 1] + D->ownOffset;
		}
	}
	// ... and the next direction formed inside the local block's SpMV where that block is served by the 2.5-D constant-diagonal kernel and the
	// remote one is empty or thin (the slabs of a grid): the halo of r travels instead of p's, in a halo-extended r.  What travels differs from
	// the other loop forms, so the ranks must AGREE: decided once per solve by a vote (one small all-reduce; nothing is in flight yet)
	bool fuse = false;
	if (lazy && D->aLoc && constMarchFusable(D->aLoc, sizeof(T)) && D->recvs.size() <= static_cast<size_t>(P2P_MAX_WORLD) && D->chunks == 1) {
		bool thin = false;
		if (!D->remEmpty) SMM_TRY(thinUsable(D, s, &thin));
		fuse = D->remEmpty || thin;
	}
	if (fuse && !D->rExt) {
		const size_t eb = static_cast<size_t>(std::max(1, D->extLen)) * sizeof(T);
		if (devAlloc(&D->rExt, eb) != SMM_HIP_OK) {
			(void)hipGetLastError();
			D->rExt = nullptr;
			fuse = false;
		} else {
			SMM_HIP_TRY(hipMemsetAsync(D->rExt, 0, eb, s));
		}
	}
	if (getenv("SMM_HIP_DIST_DEBUG")) {
		const smm_hip_csr* m = D->aLoc;
		fprintf(stderr, "libsmm_hip: dist_cg rank %d: lazy %d, local block fusable %d (family %d lanes %d state %d encoding %d const %d const_off %d clusters %d), remote empty %d thin rows %d, candidate %d\n",
		        D->comm->rank, lazy ? 1 : 0, m && constMarchFusable(m, sizeof(T)) ? 1 : 0, m ? m->family() : -1, m ? m->lanes() : -1, m ? static_cast<int>(m->pat_state.load()) : -1,
		        m ? m->pat_encoding : -1, m ? static_cast<int>(m->pat_const) : -1, m ? static_cast<int>(m->pat_const_off) : -1, m ? static_cast<int>(m->march_clusters) : -1,
		        D->remEmpty ? 1 : 0, D->nThin, fuse ? 1 : 0);
	}
	if (D->comm->world > 1) {
		long long votes = fuse ? 1 : 0;
		SMM_TRY(commAllreduceI64(D->comm, &votes, 1));
		fuse = votes == D->comm->world;
	}
	T* rExt = nullptr;
	if (fuse) {
		rExt = static_cast<T*>(D->rExt);
		r = rExt + D->ownOffset;
	}
	if (n > 0) SMM_HIP_TRY(hipMemcpyAsync(xExt + D->ownOffset, x0, sizeof(T) * n, hipMemcpyDeviceToDevice, s));
	SMM_TRY(distMatvec<T>(D, xExt, 2, SMM_OP_SUB, b, r, 0, nullptr, nullptr, nullptr, s));  // r = b - A x0, ref:2337
	SMM_TRY(launchCopy2<T>(n, r, p, nullptr, s));                                           // p = r, ref:2340
	distDots<T><<<NPART, TPB, 0, s>>>(n, r, r, nullptr, 1, partsC, nullptr);
	if (!D->p2p) distFinishSums<T><<<1, TPB, 0, s>>>(partsC, 1, partsC + PARTS_TOTALS, nullptr);
	SMM_TRY(allreduceTotals<T>(D, partsC, 1, s, &ev, 2));
	SMM_TRY(join(s, ev));
	distCgInit<T><<<1, 1, 0, s>>>(partsC + PARTS_TOTALS, sc, eps);
	const int* doneFlag = &sc->done;
	T* lastExt = pExt;  // the vector the last posted exchange fills
	const bool nt3 = updateNT(n, sizeof(T), 3);
	HaloSegs segs{};
	for (const Seg& g : D->recvs) {
		if (segs.n < P2P_MAX_WORLD) {
			segs.off[segs.n] = g.offset;
			segs.cnt[segs.n++] = g.count;
		}
	}
	RowRanges allRows{};
	allRows.n = 1;
	allRows.hi[0] = n;
	for (int i = 0; i < maxIterations; ++i) {
		if (i % CHECK_EVERY == 0) {  // i == 0: the early exit of ref:2342-2344 costs nothing more than this read
			int seen = 0;
			SMM_TRY(readDone(D, doneFlag, s, &seen));
			if (seen) break;
		}
		const int par = i & 1;
		const int cur = lazy ? i % (LAZY_M + 1) : 0, next = lazy ? (cur + 1) % (LAZY_M + 1) : 0;
		T* const curExt = ringExt[cur];
		T* const pc = ring.p[cur];
		if (fuse) {
			// the loop of smm_solvers.hip's cgDev with the direction formed INSIDE the SpMV: SpMV' (iteration i - 1's bookkeeping, p_i, A_loc p_i, the local
			// share of p.Ap) while r's halo is in flight, the thin remote block -- which forms the halo of p_i -- behind it, the flush of x (scheduled every LAZY_M-th
			// iteration, or because SpMV' found iteration i - 1 converged), the r update -- and r's halo on its way while ||r||^2 is reduced
			if (i == 0) {
				SMM_TRY(distExchangeBegin<T>(D, curExt, 0, s));
				SMM_TRY(distMatvecCompute<T>(D, curExt, SMM_OP_ASSIGN, nullptr, ap, 1, pc, partsA, doneFlag, s, nullptr, 0, SPMV_HALF_TILES));
			} else {
				const int prev = (i - 1) % (LAZY_M + 1);
				auto& pend = D->pending;
				const bool exchange = pend.active;
				pend.active = false;
				const CgFuseBook<T> bk{&sc->pad, sc->rrPing, &sc->res, &sc->iters, &sc->done, &sc->status, &sc->flushIter};
				CgFuseArgs<T> f{r, pc, bk, nullptr, eps, (i - 1) & 1, i};
				f.totalsC = partsC + PARTS_TOTALS;
				f.extraFlags = SPMV_FINISH | (exchange && pend.async ? SPMV_LEAVE_ROOM : 0);
				if (!launchConstMarchFusedP<T>(D->aLoc, ring.p[prev], ap, partsA, doneFlag, f, s)) {
					setError("dist_cg: the fused SpMV could not be launched");
					return SMM_HIP_ERR_HIP;
				}
				++D->cgFused;
				if (exchange && pend.landed[0]) {  // (also a rank that only sends: r's boundary rows must not be rewritten before they have left)
					profWaitWaiting(pend.waitSlot[0], s);
					SMM_HIP_TRY(hipStreamWaitEvent(s, pend.landed[0], 0));
				}
				D->totalsFinal = true;
				if (!D->remEmpty) {  // the thin remote block, forming the halo of p_i as it goes (and storing it for iteration i + 1)
					ThinForm<T> fm;
					fm.pOldExt = ringExt[prev];
					fm.rExt = rExt;
					fm.pNewExt = curExt;
					fm.sc = sc;
					fm.totalsC = partsC + PARTS_TOTALS;
					fm.par = (i - 1) & 1;
					fm.segs = segs;
					SMM_TRY(launchThinRemote<T>(D, curExt, false, ap, 1, pc, partsA, doneFlag, s, &fm));
				}
				SMM_DIST_UPDATE(distCgFlushOnly, updateNT(n, sizeof(T), 5), gridFor(n), s, allRows, sc, ring, prev, (i - 1) % LAZY_M + 1, i % LAZY_M == 0 ? 1 : 0, i,
				                (i - 1) % LAZY_M, i <= LAZY_M ? x0 : x, x);
			}
			SMM_TRY(allreduceTotals<T>(D, partsA, 1, s, &ev, 0, doneFlag));
			SMM_TRY(join(s, ev));
			SMM_DIST_UPDATE(distCgR, nt3, NPART, s, n, sc, par, partsA + PARTS_TOTALS, ap, r, partsC, i % LAZY_M);
			if (!D->p2p) distFinishSums<T><<<1, TPB, 0, s>>>(partsC, 1, partsC + PARTS_TOTALS, doneFlag);
			SMM_TRY(allreduceTotals<T>(D, partsC, 1, s, &ev, 2, doneFlag));
			const bool lastPass = i + 1 >= maxIterations;
			if (!lastPass) {
				SMM_TRY(distExchangeBegin<T>(D, rExt, 1, s));  // r's halo travels behind the reduction of ||r||^2 and beside the next SpMV'
				lastExt = rExt;
			}
			SMM_TRY(join(s, ev));
			if (lastPass) {  // the last planned iteration has no SpMV' behind it: its bookkeeping and the rest of x
				SMM_DIST_UPDATE(distCgLazyP, nt3, gridFor(n), s, allRows, 1, sc, par, partsC + PARTS_TOTALS, eps, ring, cur, i % LAZY_M + 1, 1, i % LAZY_M, r,
				                i < LAZY_M ? x0 : x, x);
			}
			continue;
		}
		if (i == 0) SMM_TRY(distExchangeBegin<T>(D, curExt, 0, s));  // (later passes: posted behind the update of p)
		SMM_TRY(distMatvecCompute<T>(D, curExt, SMM_OP_ASSIGN, nullptr, ap, 1, pc, partsA, doneFlag, s, nullptr, 0, SPMV_HALF_TILES));  // Ap = A p ; p.Ap, ref:2353-2354
		SMM_TRY(allreduceTotals<T>(D, partsA, 1, s, &ev, 0, doneFlag));
		SMM_TRY(join(s, ev));
		SMM_DIST_UPDATE(distCgR, nt3, NPART, s, n, sc, par, partsA + PARTS_TOTALS, ap, r, partsC, i % LAZY_M);
		if (!D->p2p) distFinishSums<T><<<1, TPB, 0, s>>>(partsC, 1, partsC + PARTS_TOTALS, doneFlag);
		SMM_TRY(allreduceTotals<T>(D, partsC, 1, s, &ev, 2, doneFlag));           // (the communicator's collectives: on the side stream ...
		if (!lazy) SMM_DIST_UPDATE(distCgX, nt3, gridFor(n), s, n, sc, pc, i == 0 ? x0 : x, x);  // ... while x is updated; ref:2351, 2395)
		SMM_TRY(join(s, ev));
		const bool last = i + 1 >= maxIterations;
		RowRanges all{};
		all.n = 1;
		all.hi[0] = n;
		if (lazy) {
			const int pending = i % LAZY_M + 1;
			const int flush = (pending == LAZY_M || last) ? 1 : 0;
			const T* xc = i < LAZY_M ? x0 : x;  // (until the first scheduled flush x has not been written: ref:2351, 2395)
			auto launch = [&](const RowRanges& rg, int book) {
				SMM_DIST_UPDATE(distCgLazyP, nt3, gridFor(rg.rows()), s, rg, book, sc, par, partsC + PARTS_TOTALS, eps, ring, cur, pending, flush, i % LAZY_M, r, xc, x);
			};
			if (!last) {
				SMM_TRY((updateThenExchange<T>(D, ringExt[next], 0, s, launch)));
				lastExt = ringExt[next];
			} else {
				launch(all, 1);
			}
		} else if (!last) {
			SMM_TRY((updateThenExchange<T>(D, pExt, 0, s, [&](const RowRanges& rg, int book) {
				SMM_DIST_UPDATE(distCgP, nt3, gridFor(rg.rows()), s, rg, book, sc, par, partsC + PARTS_TOTALS, eps, p, r);
			})));
		} else {
			SMM_DIST_UPDATE(distCgP, nt3, gridFor(n), s, all, 1, sc, par, partsC + PARTS_TOTALS, eps, p, r);
		}
	}
	SMM_TRY(distExchangeDrain<T>(D, lastExt, s));
	SMM_HIP_TRY(hipGetLastError());
	DistScal<T> h;
	unsigned long long p2pErr = 0;
	SMM_HIP_TRY(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, s));
	SMM_TRY(p2pPostErrRead(D, &p2pErr, s));
	SMM_TRY(boundedSync(D->comm, s));
	SMM_TRY(p2pFailIf(D, p2pErr));
	if (status) *status = h.status;
	if (iterations) *iterations = h.iters;
	if (resnorm2) *resnorm2 = h.res;
	return SMM_HIP_OK;
}

}  // namespace smm

using namespace smm;

extern "C" {

int smm_hip_comm_unique_id(void* id) {
	if (!id) {
		setError("comm_unique_id: null buffer");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	RcclApi* api = rccl();
	if (!api) return SMM_HIP_ERR_COMM;
	static_assert(sizeof(ncclUniqueId) == SMM_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
	ncclUniqueId u;
	SMM_RCCL_TRY(api->GetUniqueId(&u));
	memcpy(id, &u, sizeof(u));
	return SMM_HIP_OK;
}

static int commFinish(smm_hip_comm* c, smm_hip_comm** out) {
	int least = 0, greatest = 0;
	hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
	if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest);
	if (e != hipSuccess) {
		smm_hip_comm_destroy(c);
		return hipFail(e, "comm stream", __FILE__, __LINE__);
	}
	*out = c;
	return SMM_HIP_OK;
}

int smm_hip_comm_create_self(smm_hip_comm** out) {
	if (!out) {
		setError("comm_create: out is null");
		return SMM_HIP_ERR_INVALID;
	}
	*out = nullptr;
	SMM_TRY(ensureInit());
	auto* c = new smm_hip_comm();
	return commFinish(c, out);
}

int smm_hip_comm_create_rccl(int rank, int world, const void* id, smm_hip_comm** out) {
	if (!out || !id || world < 1 || rank < 0 || rank >= world) {
		setError("comm_create_rccl: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	*out = nullptr;
	SMM_TRY(ensureInit());
	RcclApi* api = rccl();
	if (!api) return SMM_HIP_ERR_COMM;
	auto* c = new smm_hip_comm();
	c->rank = rank;
	c->world = world;
	c->kind = SMM_COMM_RCCL;
	ncclUniqueId u;
	memcpy(&u, id, sizeof(u));
	// ncclCommInitRank blocks until every rank has arrived.  It runs on a helper thread so that this rank can give up after
	// SMM_HIP_COMM_TIMEOUT_S seconds (a peer that never starts, a wrong unique id): the helper is then left behind -- it cannot be
	// cancelled -- and the caller is expected to exit; nothing else is shared with it but the state block below.
	struct InitState {
		std::mutex mu;
		std::condition_variable cv;
		bool done = false;
		bool abandoned = false;  // the waiter gave up: a communicator that arrives later belongs to nobody and is aborted by the helper
		ncclResult_t result = ncclSuccess;
		ncclComm_t comm = nullptr;
	};
	auto state = std::make_shared<InitState>();
	int device = 0;
	SMM_HIP_TRY(hipGetDevice(&device));
	std::thread([state, api, world, u, rank, device]() {
		(void)hipSetDevice(device);
		ncclComm_t comm = nullptr;
		const ncclResult_t r = api->CommInitRank(&comm, world, u, rank);
		std::lock_guard<std::mutex> lock(state->mu);
		if (state->abandoned) {  // (a peer that arrived after the time-out: nothing may leak into a process that went on without it)
			if (r == ncclSuccess && comm && api->CommAbort) api->CommAbort(comm);
			return;
		}
		state->result = r;
		state->comm = comm;
		state->done = true;
		state->cv.notify_all();
	}).detach();
	{
		std::unique_lock<std::mutex> lock(state->mu);
		if (!state->cv.wait_for(lock, std::chrono::duration<double>(commTimeoutSeconds()), [&] { return state->done; })) {
			state->abandoned = true;
			delete c;
			setError("comm_create_rccl: rank %d of %d waited %.0f s in ncclCommInitRank (is every rank running, with the same unique id?)", rank, world,
			         commTimeoutSeconds());
			return SMM_HIP_ERR_COMM;
		}
	}
	if (state->result != ncclSuccess) {
		delete c;
		return rcclFail(state->result, "ncclCommInitRank");
	}
	c->nccl = state->comm;
	if (api->CommCount) {  // the size as RCCL itself reports it
		int count = 0;
		if (api->CommCount(c->nccl, &count) != ncclSuccess || count != world) {
			const int st = rcclFail(ncclInternalError, "ncclCommCount (communicator size differs from the requested world)");
			commAbort(c);
			delete c;
			return st;
		}
	}
	return commFinish(c, out);
}

int smm_hip_comm_create_host(int rank, int world, smm_hip_host_allreduce_fn allreduce, smm_hip_host_sendrecv_fn sendrecv, void* user,
                             smm_hip_comm** out) {
	if (!out || !allreduce || !sendrecv || world < 1 || rank < 0 || rank >= world) {
		setError("comm_create_host: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	*out = nullptr;
	SMM_TRY(ensureInit());
	auto* c = new smm_hip_comm();
	c->rank = rank;
	c->world = world;
	c->kind = SMM_COMM_HOST;
	c->hostAllreduce = allreduce;
	c->hostSendrecv = sendrecv;
	c->user = user;
	return commFinish(c, out);
}

int smm_hip_comm_destroy(smm_hip_comm* c) {
	if (!c) return SMM_HIP_OK;
	if (c->stream) {
		hipStreamSynchronize(c->stream);
		forgetStream(c->stream);
		hipStreamDestroy(c->stream);
	}
	for (hipEvent_t e : c->events) {
		if (e) hipEventDestroy(e);
	}
	if (c->nccl && rccl()) rccl()->CommDestroy(c->nccl);
	if (c->pinned) hipHostFree(c->pinned);
	if (c->d_i64) hipFree(c->d_i64);
	delete c;
	return SMM_HIP_OK;
}

int smm_hip_comm_info(const smm_hip_comm* c, int* rank, int* world, int* kind) {
	if (!c) {
		setError("comm_info: null communicator");
		return SMM_HIP_ERR_INVALID;
	}
	if (rank) *rank = c->rank;
	if (world) *world = c->world;
	if (kind) *kind = c->kind;
	return SMM_HIP_OK;
}

int smm_hip_comm_rccl_ranks(const smm_hip_comm* c, int* count) {
	if (!c || !count) {
		setError("comm_rccl_ranks: null argument");
		return SMM_HIP_ERR_INVALID;
	}
	*count = 0;
	if (c->kind != SMM_COMM_RCCL) return SMM_HIP_OK;  // not an RCCL communicator: 0
	SMM_TRY(commUsable(c));
	RcclApi* api = rccl();
	if (!api || !api->CommCount) {
		setError("comm_rccl_ranks: this librccl has no ncclCommCount");
		return SMM_HIP_ERR_COMM;
	}
	SMM_RCCL_TRY(api->CommCount(c->nccl, count));
	return SMM_HIP_OK;
}

// exercises every collective the loops use on this communicator: an all-reduce of (rank + 1) and a ring exchange (with one rank:
// a send to / receive from itself).  Returns SMM_HIP_ERR_COMM when a result is wrong.
static int commSelftest(smm_hip_comm* c);
int smm_hip_comm_selftest(smm_hip_comm* c) {
	if (!c) {
		setError("comm_selftest: null communicator");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	SMM_TRY(commUsable(c));
	return guardComm(c, commSelftest(c));
}

static int commSelftest(smm_hip_comm* c) {
	hipStream_t s = c->stream;
	DevBuf<double> buf;
	SMM_TRY(buf.alloc(4));
	double h[4] = {static_cast<double>(c->rank + 1), 2.0 * (c->rank + 1), static_cast<double>(c->rank), -1.0};
	SMM_HIP_TRY(hipMemcpyAsync(buf, h, sizeof(h), hipMemcpyHostToDevice, s));
	if (c->kind != SMM_COMM_SELF) SMM_TRY(commAllreduce<double>(c, buf, 2, s));
	if (c->kind != SMM_COMM_SELF) {
		const int next = (c->rank + 1) % c->world, prev = (c->rank + c->world - 1) % c->world;
		std::vector<Seg> sends{{next, 2, 1}}, recvs{{prev, 3, 1}};
		SMM_TRY(commExchange<double>(c, buf.p, sends, recvs, s));
	}
	double g[4];
	SMM_HIP_TRY(hipMemcpyAsync(g, buf, sizeof(g), hipMemcpyDeviceToHost, s));
	SMM_TRY(boundedSync(c, s));
	if (c->kind != SMM_COMM_SELF) {
		const double want = 0.5 * c->world * (c->world + 1);
		const int prev = (c->rank + c->world - 1) % c->world;
		if (g[0] != want || g[1] != 2 * want || g[3] != static_cast<double>(prev)) {
			setError("comm_selftest: got %g %g %g, expected %g %g %d", g[0], g[1], g[3], want, 2 * want, prev);
			return SMM_HIP_ERR_COMM;
		}
	}
	return SMM_HIP_OK;
}

int smm_hip_partition_rows_by_nnz(const int* start, int rows, int world, int* bounds) {
	if (!start || !bounds || rows < 0 || world < 1) {
		setError("partition_rows_by_nnz: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	const long long total = start[rows];
	bounds[0] = 0;
	for (int g = 1; g < world; ++g) {
		const long long target = total * g / world;
		const int* it = std::lower_bound(start + bounds[g - 1], start + rows, target, [](int v, long long t) { return static_cast<long long>(v) < t; });
		bounds[g] = static_cast<int>(it - start);
	}
	bounds[world] = rows;
	return SMM_HIP_OK;
}

int smm_hip_dist_csr_create_dev_f32(smm_hip_comm* comm, int n_global, const int* bounds, const int* d_start, const int* d_positions,
                                    const float* d_values, smm_hip_dist_csr** out) {
	SMM_TRY(commUsable(comm));
	return guardComm(comm, distCreate<float>(comm, n_global, bounds, d_start, d_positions, d_values, out));
}
int smm_hip_dist_csr_create_dev_f64(smm_hip_comm* comm, int n_global, const int* bounds, const int* d_start, const int* d_positions,
                                    const double* d_values, smm_hip_dist_csr** out) {
	SMM_TRY(commUsable(comm));
	return guardComm(comm, distCreate<double>(comm, n_global, bounds, d_start, d_positions, d_values, out));
}

int smm_hip_dist_csr_destroy(smm_hip_dist_csr* D) {
	if (!D) return SMM_HIP_OK;
	p2pTeardown(D);
	smm_hip_csr_destroy(D->aLoc);
	smm_hip_csr_destroy(D->aRem);
	for (smm_hip_csr* piece : D->aRemK) smm_hip_csr_destroy(piece);
	for (void* p : D->chunkArrays) devFree(p);
	for (void* p : D->arrays) devFree(p);
	for (void* p : {D->r, D->r0, D->ap, D->as, D->scratch, D->pExt, D->sExt, D->xExt, D->partsA, D->partsB, D->partsC, D->sc}) devFree(p);
	devFree(D->splitSync);
	devFree(D->thinRows);
	devFree(D->partsThin);
	devFree(D->rExt);
	for (void* p : D->lazyExt) devFree(p);
	delete D;
	return SMM_HIP_OK;
}

int smm_hip_dist_csr_info(const smm_hip_dist_csr* D, int* n_local, int* ext_len, int* own_offset, int* halo_elements, long long* nnz_loc,
                          long long* nnz_rem) {
	if (!D) {
		setError("dist_csr_info: null handle");
		return SMM_HIP_ERR_INVALID;
	}
	if (n_local) *n_local = D->nLocal;
	if (ext_len) *ext_len = D->extLen;
	if (own_offset) *own_offset = D->ownOffset;
	if (halo_elements) *halo_elements = D->haloElements;
	if (nnz_loc) *nnz_loc = D->nnzLoc;
	if (nnz_rem) *nnz_rem = D->nnzRem;
	return SMM_HIP_OK;
}

int smm_hip_dist_csr_halo_chunks(const smm_hip_dist_csr* D, int* chunks) {
	if (!D || !chunks) {
		setError("dist_csr_halo_chunks: null argument");
		return SMM_HIP_ERR_INVALID;
	}
	*chunks = D->chunks;
	return SMM_HIP_OK;
}

// The peer-to-peer plan of one rank as plain numbers (host arithmetic only: no device, no communicator) -- what the set-up derives from
// the globally known column ranges and row bounds.  The CPU tests replay it for every rank of a world of 8 and check that every halo
// element arrives exactly once.  Records of 8 values:
//   {0, dst, relay (-1: direct), first global column, position (dst's landing area / the relay's staging area), count, path, relay job (-1)}   push
//   {1, src, dst, position in my staging area, position in dst's landing area, count, path, relay job}                                           forward
//   {2, src, offset in my halo-extended vector, position in my landing area, count, paths, 0, 0}                                                 land
int smm_hip_dist_p2p_plan(int world, int rank, const long long* needs, const int* bounds, int relays, double direct_share, long long* out, int out_capacity,
                          int* out_count) {
	if (world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world || !needs || !bounds || !out_count || relays < 0 || relays > P2P_MAX_PATHS - 1) {
		setError("dist_p2p_plan: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	relays = std::min(relays, std::max(0, world - 2));
	if (relays == 0) direct_share = 1.0;
	bool tooMany = false;
	const P2PPlan plan = p2pMakePlan(world, needs, bounds, relays, direct_share, &tooMany);
	if (tooMany) {
		setError("dist_p2p_plan: more relay jobs per rank than the header holds");
		return SMM_HIP_ERR_INVALID;
	}
	std::vector<long long> rec;
	for (const P2PRoute& rt : plan.routes) {
		const long long n = rt.b - rt.a;
		if (n <= 0) continue;
		const int q = rt.seg.dst;
		if (rt.seg.src == rank) {
			const long long col = needs[2 * static_cast<size_t>(q)] + rt.seg.extOff + rt.a;
			rec.insert(rec.end(), {0LL, q, rt.relay, col, rt.relay < 0 ? rt.seg.landOff + rt.a : rt.stagePos, n, rt.part, rt.relay < 0 ? -1LL : rt.job});
		}
		if (rt.relay == rank) rec.insert(rec.end(), {1LL, rt.seg.src, q, rt.stagePos, rt.seg.landOff + rt.a, n, rt.part, rt.job});
	}
	for (const PlanSeg& g : planRecvs(world, needs, bounds, rank)) {
		long long paths = 0;
		for (const P2PRoute& rt : plan.routes) {
			if (rt.seg.dst == rank && rt.seg.src == g.src && rt.b > rt.a) paths = std::max<long long>(paths, rt.part + 1);
		}
		rec.insert(rec.end(), {2LL, g.src, g.extOff, g.landOff, g.count, paths, 0LL, 0LL});
	}
	*out_count = static_cast<int>(rec.size() / 8);
	if (out && static_cast<size_t>(out_capacity) >= rec.size()) memcpy(out, rec.data(), rec.size() * sizeof(long long));
	return SMM_HIP_OK;
}

int smm_hip_dist_csr_options(const smm_hip_dist_csr* D, int* p2p, int* relays, int* halo_first, double* direct_share) {
	if (!D) {
		setError("dist_csr_options: null handle");
		return SMM_HIP_ERR_INVALID;
	}
	if (p2p) *p2p = D->p2p ? (D->p2p->haloOn ? 1 : 2) : 0;
	if (relays) *relays = D->p2p ? D->p2p->relays : 0;
	if (halo_first) *halo_first = D->haloFirst ? 1 : 0;
	if (direct_share) *direct_share = D->p2p ? D->p2p->directShare : 1.0;
	return SMM_HIP_OK;
}

int smm_hip_dist_csr_matvec_forms(const smm_hip_dist_csr* D, long long* one_launch, long long* two_launches) {
	if (!D) {
		setError("dist_csr_matvec_forms: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	if (one_launch) *one_launch = D->matvecsSplit;
	if (two_launches) *two_launches = D->matvecsTwo;
	return SMM_HIP_OK;
}

int smm_hip_dist_csr_thin_remote(const smm_hip_dist_csr* D, int* rows, long long* matvecs) {
	if (!D) {
		setError("dist_csr_thin_remote: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	if (rows) *rows = D->nThin;
	if (matvecs) *matvecs = D->matvecsThin;
	return SMM_HIP_OK;
}

int smm_hip_dist_csr_cg_fused(const smm_hip_dist_csr* D, long long* matvecs) {
	if (!D || !matvecs) {
		setError("dist_csr_cg_fused: null argument");
		return SMM_HIP_ERR_INVALID;
	}
	*matvecs = D->cgFused;
	return SMM_HIP_OK;
}

int smm_hip_dist_csr_split_wait(smm_hip_dist_csr* D, double* waited_ms, int reset) {
	if (!D || !waited_ms) {
		setError("dist_csr_split_wait: null argument");
		return SMM_HIP_ERR_INVALID;
	}
	*waited_ms = 0.0;
	if (!D->splitSync) return SMM_HIP_OK;
	unsigned long long ticks = 0;
	SMM_HIP_TRY(hipDeviceSynchronize());
	SMM_HIP_TRY(hipMemcpy(&ticks, D->splitSync + P2P_KINDS + 1, sizeof(ticks), hipMemcpyDeviceToHost));
	if (reset) SMM_HIP_TRY(hipMemset(D->splitSync + P2P_KINDS + 1, 0, sizeof(ticks)));
	*waited_ms = static_cast<double>(ticks) * 1.0e-5;  // (wall_clock64 counts at 100 MHz on gfx9)
	return SMM_HIP_OK;
}

int smm_hip_dist_csr_local_block(const smm_hip_dist_csr* D, smm_hip_csr** a_loc, smm_hip_csr** a_rem) {
	if (!D) {
		setError("dist_csr_local_block: null handle");
		return SMM_HIP_ERR_INVALID;
	}
	if (a_loc) *a_loc = D->aLoc;
	if (a_rem) *a_rem = D->aRem;
	return SMM_HIP_OK;
}

int smm_hip_dist_spmv_dev_f32(smm_hip_dist_csr* D, int op, const float* d_lhs, const float* d_x, float* d_out, smm_hip_stream stream) {
	SMM_TRY(commUsable(D ? D->comm : nullptr));
	return guardComm(D ? D->comm : nullptr, distSpmv<float>(D, op, d_lhs, d_x, d_out, pickStream(stream)));
}
int smm_hip_dist_spmv_dev_f64(smm_hip_dist_csr* D, int op, const double* d_lhs, const double* d_x, double* d_out, smm_hip_stream stream) {
	SMM_TRY(commUsable(D ? D->comm : nullptr));
	return guardComm(D ? D->comm : nullptr, distSpmv<double>(D, op, d_lhs, d_x, d_out, pickStream(stream)));
}

int smm_hip_dist_bicgstab_dev_f32(smm_hip_dist_csr* D, const float* d_b, float* d_x, int maxIterations, float eps, const smm_hip_precond* M_loc,
                                  smm_hip_stream stream, int* solver_status, int* iterations, float* resnorm) {
	SMM_TRY(commUsable(D ? D->comm : nullptr));
	return guardComm(D ? D->comm : nullptr, distBicgstab<float>(D, d_b, d_x, maxIterations, eps, M_loc, pickStream(stream), solver_status, iterations, resnorm));
}
int smm_hip_dist_bicgstab_dev_f64(smm_hip_dist_csr* D, const double* d_b, double* d_x, int maxIterations, double eps, const smm_hip_precond* M_loc,
                                  smm_hip_stream stream, int* solver_status, int* iterations, double* resnorm) {
	SMM_TRY(commUsable(D ? D->comm : nullptr));
	return guardComm(D ? D->comm : nullptr, distBicgstab<double>(D, d_b, d_x, maxIterations, eps, M_loc, pickStream(stream), solver_status, iterations, resnorm));
}

int smm_hip_dist_cg_dev_f32(smm_hip_dist_csr* D, const float* d_b, const float* d_x0, float* d_x, int maxIterations, float eps,
                            smm_hip_stream stream, int* solver_status, int* iterations, float* resnorm2) {
	SMM_TRY(commUsable(D ? D->comm : nullptr));
	return guardComm(D ? D->comm : nullptr, distCg<float>(D, d_b, d_x0, d_x, maxIterations, eps, pickStream(stream), solver_status, iterations, resnorm2));
}
int smm_hip_dist_cg_dev_f64(smm_hip_dist_csr* D, const double* d_b, const double* d_x0, double* d_x, int maxIterations, double eps,
                            smm_hip_stream stream, int* solver_status, int* iterations, double* resnorm2) {
	SMM_TRY(commUsable(D ? D->comm : nullptr));
	return guardComm(D ? D->comm : nullptr, distCg<double>(D, d_b, d_x0, d_x, maxIterations, eps, pickStream(stream), solver_status, iterations, resnorm2));
}

}  // extern "C"
