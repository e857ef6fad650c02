// smm_internal.h -- shared declarations of libsmm_hip.so (gfx950 only; no CPU fallback anywhere).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/smm_hip.h"

namespace smm {

// ---------------------------------------------------------------------------------------------------------
// error plumbing: every C entry point returns an int; HIP failures land in a thread-local message
// ---------------------------------------------------------------------------------------------------------
void setError(const char* fmt, ...);
int hipFail(hipError_t e, const char* what, const char* file, int line);

#define SMM_HIP_TRY(expr)                                             \
	do {                                                              \
		hipError_t _e = (expr);                                       \
		if (_e != hipSuccess) {                                       \
			return ::smm::hipFail(_e, #expr, __FILE__, __LINE__);     \
		}                                                             \
	} while (0)

#define SMM_TRY(expr)              \
	do {                           \
		int _s = (expr);           \
		if (_s != SMM_HIP_OK) {    \
			return _s;             \
		}                          \
	} while (0)

// SMM_HIP_TRACE_SETUP=1: the host-side stages of the one-off set-up work (PATTERN analysis, tile tables) print their wall time on stderr
struct SetupTrace {
	const char* what;
	std::chrono::steady_clock::time_point t0;
	static bool on() {
		static const bool v = [] {
			const char* env = getenv("SMM_HIP_TRACE_SETUP");
			return env && atoi(env) != 0;
		}();
		return v;
	}
	explicit SetupTrace(const char* w) : what(w) {
		if (on()) t0 = std::chrono::steady_clock::now();
	}
	~SetupTrace() {
		if (on()) {
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			std::fprintf(stderr, "[smm-hip setup] %-40s %8.3f ms\n", what, ms);
		}
	}
};

// make sure a device is selected; returns SMM_HIP_ERR_NO_DEVICE when there is none
int ensureInit();
hipStream_t libStream();
// `_dev` entry points follow plain HIP semantics: the handle is a hipStream_t and NULL is the null (default) stream --
// which is also what torch.cuda.current_stream().cuda_stream is for PyTorch's default stream
// (the allocator is told: a block freed later becomes reusable only after an event recorded on this stream has completed)
void noteStream(hipStream_t s);
void forgetStream(hipStream_t s);  // before the library destroys a stream of its own
inline hipStream_t pickStream(smm_hip_stream s) {
	noteStream(static_cast<hipStream_t>(s));
	return static_cast<hipStream_t>(s);
}
int numCUs();
// one kernel of each hot-path translation unit is touched at init (see smm_runtime.hip, preloadUnits)
void preloadSpmvUnit();
void preloadPatternUnit();
void preloadMarchUnit();
void preloadBlas1Unit();
void preloadSolversUnit();
void preloadResidentUnits();  // the two single-launch solvers (smm_resident.hip, smm_resident_bicg.hip)

// caching device allocator (solver temporaries are allocated per call like the reference's SMM::Vector,
// ref:2336-2339, but hipMalloc is far too slow to sit in that path)
int devAlloc(void** p, size_t bytes);
void devFree(void* p);
void devTrim();

// Host <-> device copies of the host-pointer entry points (the reference's calling convention: caller-owned, PAGEABLE arrays), staged
// through pinned chunks of the library's own instead of letting the HIP runtime pin the caller's pages in place.  r05 measurement
// (tools/lab/pageable_copy_probe.hip, profiles/r05/pageable_copy_probe.txt): with in-place pinning the SECOND pair of 10 MB copies of a
// process takes 21-28 ms instead of 0.4 (and 40 MB vectors 14-16 ms instead of 3.5 for four calls in a row) -- hipHostRegister around the
// call behaves the same --, staged copies take 0.8 / 6 ms every time.  hostToDev returns when `h_src` has been read (the device copy may
// still be in flight on `s`); devToHost synchronises `s`.
int hostToDev(void* d_dst, const void* h_src, size_t bytes, hipStream_t s);
int devToHost(void* h_dst, const void* d_src, size_t bytes, hipStream_t s);

template <typename T>
struct DevBuf {
	T* p = nullptr;
	size_t n = 0;
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	~DevBuf() { release(); }
	int alloc(size_t count) {
		release();
		n = count;
		return devAlloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T));
	}
	void release() {
		if (p) devFree(p);
		p = nullptr;
		n = 0;
	}
	T* detach() {  // the caller takes over the allocation
		T* q = p;
		p = nullptr;
		n = 0;
		return q;
	}
	operator T*() const { return p; }
};

template <typename T>
constexpr int dtypeOf() { return std::is_same<T, float>::value ? SMM_DTYPE_F32 : SMM_DTYPE_F64; }

}  // namespace smm

// ---------------------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------------------
struct smm_hip_csr {
	int rows = 0, cols = 0, nnz = 0, dtype = 0;
	int firstActiveStart = 0;
	// nnz / firstActiveStart / the SpMV configuration of a matrix wrapped around caller-owned DEVICE arrays are read from the device
	// the first time they are needed, on the stream of that first use (ensureCsrReady): creating the handle enqueues nothing and
	// waits for nothing, so it is naturally ordered behind whatever the caller still has in flight on that stream
	bool ready = false;
	bool kernelForced = false;  // smm_hip_csr_set_kernel chose the family / lanes: the heuristic must not overwrite them
	std::mutex readyMutex;
	int* d_start = nullptr;
	int* d_positions = nullptr;
	void* d_values = nullptr;
	bool owns = false;
	// resolved SpMV configuration: family (low byte) and lanes per row, published TOGETHER as one word -- concurrent solves on one const
	// matrix are allowed (ref:2316-2324 take `const CSRMatrix<T>&`), and the automatic switch to the PATTERN family
	// happens inside such a solve: a launch reads the word once and never sees one family with the other's lanes
	std::atomic<int> kernelWord{SMM_SPMV_VECTOR | (4 << 8)};
	int family() const { return kernelWord.load(std::memory_order_acquire) & 0xFF; }
	int lanes() const { return kernelWord.load(std::memory_order_acquire) >> 8; }
	void setKernel(int family, int lanes) { kernelWord.store((family & 0xFF) | (lanes << 8), std::memory_order_release); }
	std::mutex adoptMutex;  // the automatic STREAM -> PATTERN switch (analysis + publication of the word) is done by one thread at a time
	// STREAM family: row blocks (first row of every block; n_rowblocks+1 entries) staged through LDS
	int* d_rowblocks = nullptr;
	int n_rowblocks = 0;
	int stream_nnz_cap = 0;  // nonzeros / rows per tile the row blocks were cut for
	int stream_max_rows = 0;
	int stream_chunk_tiles = 0;  // tiles dealt to an XCD group at a time (0: one contiguous eighth per group)
	int stream_mid_len = 0;      // nonzeros of the middle row (a typical row: sizes the gather batches of the TILE kernel)
	int max_row_len = -1;        // longest row, found on first demand (smm_resident.hip)
	std::mutex tileMutex;  // the tile table is built lazily by the first SpMV; concurrent solves on one matrix are allowed
	// PATTERN family (opt-in, smm_spmv_pattern.hip): shared column offsets + one 64-bit mask per row, its own tile table
	// 0 not analysed, 1 usable, -1 the matrix has no such pattern, -2 no masks and the dictionary not tried, -3 an automatic attempt failed
	// for lack of resources (an explicit request tries again; automatic ones do not) -- ensurePattern.  Every pat_* field below is written
	// under tileMutex BEFORE the state turns 1 (release) and read only after it was seen as 1 (acquire)
	std::atomic<int> pat_state{0};
	int pat_encoding = 0;  // 0: one 64-bit mask per row (<= 64 offsets); 1: one 16-bit code per entry (<= 65536 offsets)
	unsigned short* d_pat_codes = nullptr;
	bool pat_const = false;  // masks + every diagonal holds one value (d_pat_cval[j], raw bits): the CONST kernel needs no values[]
	unsigned long long* d_pat_cval = nullptr;
	std::vector<unsigned long long> pat_cval_host;  // the same raw bits on the host (the single-launch BiCGStab passes them as kernel arguments)
	void* d_res_ell = nullptr;  // MASKS with varying diagonals: values by offset slot, [pat_k][rows], built on first use by smm_resident_bicg.hip (under tileMutex)
	bool pat_const_off = false;  // smm_hip_csr_pattern_allow_const(m, 0): keep reading values[] (A/B measurements)
	int pat_max_off = 0;  // largest |column - row| of the offset list
	// CONST on grid-shaped matrices: the plan of the 2.5-D kernel (smm_spmv_march.hip), made once at the end of the CONST analysis
	bool march_ok = false;
	int march_P = 0, march_H = 0, march_lo = 0, march_hi = 0;  // rows per plane, halo (elements), whether -P / +P are offsets
	// far offsets in CLUSTERS around -P / +P (19- / 27-point stencils): march_lo / march_hi then count the offsets of the two clusters and the
	// three-window kernel (spmvPatternConstMarch3Kernel) serves the matrix; constant diagonals only
	bool march_clusters = false;
	unsigned* d_pat_masks32 = nullptr;  // the low halves of d_pat_masks, what the 2.5-D kernel streams (4 bytes per row)
	unsigned char* d_pat_masks8 = nullptr;  // the low bytes (matrices of at most 8 offsets with constant diagonals: 1 byte per row)
	std::vector<int> pat_offs_host;  // MASKS: the sorted offsets (host copy: the brick partition of the block preconditioners reads the grid from them)
	int pat_k = 0;
	int* d_pat_off = nullptr;
	unsigned long long* d_pat_masks = nullptr;
	int* d_pat_rowblocks = nullptr;
	int pat_n_rowblocks = 0;
	int pat_nnz_cap = 0;
	int pat_max_rows = 0;
	int pat_chunk_tiles = 0;
	// one-launch form of the row-partitioned SpMV (smm_spmv_split.hip): the most entries any run of 256 / 128 / 64 consecutive rows (cut at
	// multiples of that) holds -- its fixed row tiles are staged whole, so their LDS is sized from this; -1: not counted yet (under tileMutex)
	int split_tile_max[3] = {-1, -1, -1};
	bool split_uneven = false;  // one contiguous eighth of the rows holds more than 1.5 x its share of the entries (counted with the above)
};

struct smm_hip_precond {
	int kind = SMM_PRECOND_NONE;
	int dtype = 0;
	const smm_hip_csr* a = nullptr;
	void* d_values = nullptr;  // JACOBI: diag[rows]; ILU0 / IC0: factor values on A's pattern [nnz]; SGS: null (uses A)
	size_t n_values = 0;
	// level schedules of the lower / upper triangular sweeps: rows sorted by level
	int* d_order_lo = nullptr;
	int* d_order_up = nullptr;
	std::vector<int> lvl_ptr_lo, lvl_ptr_up;  // host: level l covers order[lvl_ptr[l] .. lvl_ptr[l+1])
	struct smm_precond_plan* plan = nullptr;  // launch groups of the two sweeps (smm_precond.hip)
	struct smm_precond_block* blk = nullptr;  // BLOCK_ILU0 / BLOCK_SGS: row blocks + packed sweep records (smm_precond_block.hip)
};

namespace smm {

// ---------------------------------------------------------------------------------------------------------
// kernel launchers (device pointers, asynchronous on `s`)
// ---------------------------------------------------------------------------------------------------------
// Partial-sum slots every fused reduction writes: NPART partials per reduced quantity
constexpr int NPART = 2048;  // 256 CUs x 8 resident workgroups of 256 lanes

// dotMode of the SpMV epilogue: which per-row products are block-reduced into `partials`
//   0 none; 1: out[i]*w1[i] -> partials[0..NPART); 2: out[i]*out[i] -> partials[0..), out[i]*w1[i] -> partials[NPART..)
// extraFlags: SPMV_FINISH -- the partial-sum buffer is a "finishing" buffer of PARTS_LEN elements: the workgroup that ends last adds
// the partials (lastBlockSums, smm_device.h) and leaves the totals at partials[PARTS_TOTALS + k]; the ticket counter sits behind them.
template <typename T>
int launchSpmv(const smm_hip_csr* m, int op, const T* lhs, const T* x, T* out, int dotMode, const T* w1, T* partials,
               const int* doneFlag, hipStream_t s, int extraFlags = 0, const T* divisor = nullptr);

// live event timing of SpMV launches (smm_hip_profile_*): begin returns a slot or -1 when profiling is off
int profBegin(hipStream_t s);
void profEnd(int slot, hipStream_t s);
// "the awaited stream's work ends here" (returns a slot, -1 when profiling is off) / "the waiting stream's own work ends here":
// smm_hip_profile_read_waits sums max(0, awaited - waiting) over the pairs
int profWaitAwaited(hipStream_t awaited);
void profWaitWaiting(int slot, hipStream_t waiting);

// flag ORed into the `op` argument of the STREAM / PATTERN kernels: write out[] with non-temporal stores.  Set for outputs too
// large to still be in cache when the next kernel reads them (measured on the 512^3 fp64 Laplacian: 3.15 -> 2.97 ms; written
// bytes cost 2-4x read bytes on this memory system, tools/membw.hip)
constexpr int SPMV_NT_OUT = 0x100;
// flag ORed into `op`: finish the fused reduction in the launch itself (see launchSpmv)
constexpr int SPMV_FINISH = 0x200;
// launch-side only (never reaches a kernel): size the persistent grid for all but 8 CUs, so that a kernel of another stream (an RCCL
// exchange running beside the local block of a distributed SpMV) finds free workgroup slots
constexpr int SPMV_LEAVE_ROOM = 0x400;
// launch-side: out[i] = (A x)[i] / lhs[i] (op must be SMM_OP_ASSIGN).  Reaches the kernels as the internal operation SPMV_OP_DIV.
constexpr int SPMV_DIV_LHS = 0x800;
constexpr int SPMV_OP_DIV = 3;
// launch-side: out[i] = (lhs[i] + (A x)[i]) / divisor[i] (op must be SMM_OP_ADD; `divisor` argument of launchSpmv): the remote block of a
// row-partitioned SpMV with the Jacobi division folded in ("add, then divide").  Internal operation SPMV_OP_ADD_DIV.
constexpr int SPMV_ADD_DIV = 0x1000;
constexpr int SPMV_OP_ADD_DIV = 4;
// launch-side: ConjugateGradient's SpMV (+ p.Ap) launches.  On the 2.5-D constant-diagonal kernel with outputs beyond the caches they
// take HALF the tile height (fp32 4 rows per lane instead of 8, fp64 2 instead of 4): the launch that forms the next direction inside the SpMV (MarchFuse) holds two
// streams' request sets and fits three workgroups per CU only with half tiles (512^3 fp32: 0.95 -> 0.83 ms per iteration, fp64 1.71 -> 1.67); the partial sums of
// p.Ap follow the tiles, so EVERY loop form of CG uses the same tiles and they stay bit for bit equal.  Other kernels ignore the flag.
constexpr int SPMV_HALF_TILES = 0x2000;
// kernel-side (spmvPatternTileKernel): no fast path for wavefronts whose 64 rows all hold the same offsets (SMM_HIP_FULL_ROWS=0, measurements)
constexpr int SPMV_NO_FULL_ROWS = 0x4000;
constexpr int PARTS_TOTALS = 2 * NPART;    // index of the two totals inside a finishing buffer
constexpr int PARTS_TICKETS = 16;             // sub-counters of the "last workgroup" ticket (lastBlockSums, smm_device.h)
constexpr int PARTS_LEN = 2 * NPART + 2 + 2 * (PARTS_TICKETS + 1);  // elements of a finishing buffer: 2 x NPART partials, 2 totals, the ticket words
template <typename T>
__host__ __device__ inline unsigned* partsTicket(T* partials) { return reinterpret_cast<unsigned*>(partials + PARTS_TOTALS + 2); }
inline int spmvOutFlags(const smm_hip_csr* m, size_t elemBytes) {
	static const int forced = [] {  // SMM_HIP_NT_OUT=0 / 1: A/B measurements of the store policy
		const char* env = getenv("SMM_HIP_NT_OUT");
		return env ? atoi(env) : -1;
	}();
	if (forced >= 0) return forced ? SPMV_NT_OUT : 0;
	// (from 64 MiB per vector, inclusive: config 4's fp32 slab at 8 GPUs is exactly that -- CG there 139 -> 124 us per iteration with the non-temporal
	// policy and the fused direction it brings, profiles/r06/dist_cg_timing_nt.txt)
	return static_cast<double>(m->rows) * static_cast<double>(elemBytes) >= 64.0 * 1024 * 1024 ? SPMV_NT_OUT : 0;
}

int buildRowBlocks(smm_hip_csr* m, int capNnz, int maxRows, hipStream_t s);
// the cut behind buildRowBlocks without touching the handle (the PATTERN family keeps a table of its own): *blocks is allocated with
// devAlloc and owned by the caller; *chunkTiles = how the tiles are dealt to the XCDs (0: one contiguous eighth each)
int buildTileTable(const smm_hip_csr* m, int capNnz, int maxRows, hipStream_t s, int** blocks, int* nBlocks, int* chunkTiles);
// greedy cut of the rows into runs of <= capNnz stored entries and <= maxRows rows, on the device: tiles[0 .. nTiles] = {first row,
// start[first row]}, closed by {rows, nnz}; allocated with devAlloc, owned by the caller; synchronises `s`
int cutRows(const int* d_start, int rows, long long nnz, int capNnz, int maxRows, hipStream_t s, int2** tiles, int* nTiles, int tilesPerChunk = 0);
// streamKnown: `s` is the stream the caller orders its work on (the `_dev` entry points); otherwise (host-side queries and set-up
// calls that have no stream) the whole device is drained first and the library's own stream is used
int ensureCsrReady(const smm_hip_csr* m, hipStream_t s, bool streamKnown);
// PATTERN family: analyse + verify the matrix (idempotent), and the launch behind launchSpmv
int ensurePattern(smm_hip_csr* m, hipStream_t s = nullptr, bool streamKnown = false, bool quiet = false, bool masksOnly = false);
// before a solver's loop: lets a mid-size matrix (>= 2^20 entries) take the PATTERN family when it fits (smm_spmv_pattern.hip)
int adoptPatternForSolver(const smm_hip_csr* m, int plannedIterations, hipStream_t s);
// the automatic attempt of a single SpMV (launchSpmv): never fails -- a matrix the analysis refuses, or cannot find memory for, stays with STREAM
void adoptPatternQuietly(const smm_hip_csr* m, hipStream_t s);
int patternLanesFor(const smm_hip_csr* m);
// the PATTERN kernel launchPat picks for `lanes` and the bytes one launch moves (smm_hip_csr_kernel_desc)
const char* patternKernelDesc(const smm_hip_csr* m, int lanes, long long* bytes);
void planMarch(smm_hip_csr* m);
int marchBuildMasks32(smm_hip_csr* m, hipStream_t s);
// Launch plumbing of the persistent SpMV kernels (per template instantiation: `slot` / `granted` are statics of the launcher).
// occupancyCached: workgroups per CU of `kernel` at `lds` bytes of dynamic LDS, asked from the runtime once per LDS size -- the query costs
// 10+ us of host time and sat on the solvers' hot path beside kernels of 10-20 us.  ensureDynamicLds: the kernel's
// hipFuncAttributeMaxDynamicSharedMemorySize raised whenever a launch needs more than the largest size granted so far (r04 set it ONCE,
// to the first qualifying matrix's size: a later matrix with longer rows got a failed launch); false: the runtime refused.
template <typename K>
inline int occupancyCached(std::atomic<long long>& slot, K kernel, int tpb, size_t lds, int fallback) {
	const long long seen = slot.load(std::memory_order_acquire);
	if (seen != 0 && static_cast<size_t>(seen >> 8) == lds) return static_cast<int>(seen & 0xFF);
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, tpb, lds) != hipSuccess || n < 1) {
		(void)hipGetLastError();
		n = fallback;
	}
	n = n > 255 ? 255 : n;
	slot.store((static_cast<long long>(lds) << 8) | n, std::memory_order_release);
	return n;
}
template <typename K>
inline bool ensureDynamicLds(std::atomic<int>& granted, K kernel, size_t lds) {
	if (lds <= 64 * 1024) return true;  // (what every kernel may use without asking)
	int have = granted.load(std::memory_order_acquire);
	while (static_cast<size_t>(have) < lds) {
		if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
			(void)hipGetLastError();
			return false;
		}
		if (granted.compare_exchange_weak(have, static_cast<int>(lds), std::memory_order_acq_rel)) break;  // (a concurrent, larger grant stays)
	}
	return true;
}
// SMM_HIP_STREAM_WGS_PER_CU (measurements): workgroups per CU of the persistent STREAM / PATTERN grids; 0: ask the runtime.  Read once.
inline int forcedWgsPerCU() {
	static const int forced = [] {
		const char* env = getenv("SMM_HIP_STREAM_WGS_PER_CU");
		return env ? (atoi(env) > 1 ? atoi(env) : 1) : 0;
	}();
	return forced;
}

bool cgHalfTiles(const smm_hip_csr* m, size_t elemBytes);  // (smm_spmv_march.hip)
// bytes per vector from which ConjugateGradient defers its x update (smm_solvers.hip; the row-partitioned loop asks too)
long long cgLazyMinBytes();
// non-temporal loads / stores for an update kernel over `vectors` vectors of n elements (they cannot stay in the Infinity Cache anyway)
bool updateNT(long long n, size_t elemBytes, int vectors);
bool masksMarchApplies(const smm_hip_csr* m);  // the masks kernels' march form serves this (analysed) matrix
bool constMarchApplies(const smm_hip_csr* m);  // ... the constant-diagonal march
template <typename T>
bool launchPatMasksMarch(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s);
template <typename T>
bool launchPatConstMarch(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s);
template <typename T>
int launchSpmvPattern(const smm_hip_csr* m, int lanes, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                      const int* doneFlag, hipStream_t s);
void chooseSpmvConfig(smm_hip_csr* m);
// The row-partitioned SpMV in ONE launch (smm_spmv_split.hip): out = op(lhs, A_loc own) (+|-) A_rem ext [/ divisor], the local half while the
// halo is in flight, the remote half once `landed` (a device word, raised on the exchange's stream by launchSplitSignal) has reached `seq`
// (null: the halo is already in place); bounded by `ticks` (100 MHz), an expired wait leaves a code in *err.  op / extraFlags as the two-launch
// form passes them to its FIRST launch plus SPMV_ADD_DIV when the Jacobi division rides in the epilogue (`divisor`).  Returns SMM_HIP_OK
// (launched), 1 (this form does not serve the pair of blocks: nothing was enqueued, run the two launches) or an error.
template <typename T>
int launchSpmvSplit(const smm_hip_csr* aLoc, const smm_hip_csr* aRem, int op, const T* lhs, const T* divisor, const T* own, const T* ext, T* out, int dotMode,
                    const T* w1, T* partials, const int* doneFlag, int extraFlags, const unsigned long long* landed, unsigned long long seq, unsigned long long* err,
                    long long ticks, hipStream_t s, const struct P2PSlotArgs* slots = nullptr, int sumsLdsMax = 16384, unsigned long long* waited = nullptr);  // slots (smm_p2p.h): the launch runs the reduction point of its dot products itself; sumsLdsMax: bytes of LDS the local half's row sums may take (more: they travel through out[])
void launchSplitSignal(unsigned long long* landed, unsigned long long seq, hipStream_t s);
void preloadSplitUnit();
// ConjugateGradient's next direction formed in the SpMV's load phase (smm_spmv_march.hip, MarchFuse): Ap = A (beta pOld + r), the new
// direction written to pNew, p.Ap into `partials`; `sc` is the solver's Scal<T>.  constMarchFusable: the matrix is served by the 2.5-D
// constant-diagonal kernel in the form this exists for; the launch returns false when it could not be made (the caller must not have
// relied on it: ask constMarchFusable first).
// the words of the solver's scalar block the fused launch does its bookkeeping in (Scal<T> on one GPU, DistScal<T> in smm_dist.hip)
template <typename T>
struct CgFuseBook {
	const int* pad = nullptr;  // the done flag as the update before found it
	T* rrPing = nullptr;
	T* res = nullptr;
	int *iters = nullptr, *done = nullptr, *status = nullptr, *flushIter = nullptr;
};
template <typename T>
struct CgFuseArgs {
	const T* r;
	T* pNew;
	CgFuseBook<T> bk;
	const T* partsC;   // ||r||^2: NPART partial sums (summed by every workgroup in the fixed order) ...
	T eps;
	int par, iter;
	const T* totalsC = nullptr;  // ... or, when set, the finished (all-reduced) total
	int extraFlags = 0;          // SPMV_FINISH / SPMV_LEAVE_ROOM of the launch (the row-partitioned loop)
};
bool constMarchFusable(const smm_hip_csr* m, size_t elemBytes);
template <typename T>
bool launchConstMarchFusedP(const smm_hip_csr* m, const T* pOld, T* Ap, T* partials, const int* doneFlag, const CgFuseArgs<T>& f, hipStream_t s);

// register-resident ConjugateGradient (smm_resident.hip): *handled = false when the matrix does not fit the register file
template <typename T>
int cgResidentTry(const smm_hip_csr* a, const T* b, const T* x0, T* x, int maxIterations, T eps, hipStream_t s, int* status, int* iterations,
                  T* resnorm2, bool* handled);

// partials[0..NPART) = per-block sums of a[i]*b[i]
template <typename T>
int launchDotPartials(int n, const T* a, const T* b, T* partials, const int* doneFlag, hipStream_t s);
// result[0] = sum of partials[0..NPART) in fixed order
template <typename T>
int launchSumPartials(const T* partials, T* result, hipStream_t s);
template <typename T>
int launchAxpy(int n, T a, const T* x, const T* y, T* out, hipStream_t s);
template <typename T>
int launchCopy2(int n, const T* src, T* dst1, T* dst2, hipStream_t s);

// Polls the device `done` flag without stalling the queue: every `interval` iterations the flag is copied into a
// pinned mailbox behind an event; the host reads mailboxes whose event has completed and waits only when more
// than two are outstanding.
struct DonePoller {
	static constexpr int SLOTS = 4;
	int* mailbox = nullptr;
	hipEvent_t ev[SLOTS] = {};
	bool pending[SLOTS] = {};
	int head = 0, count = 0;
	hipStream_t s = nullptr;
	// one poller per host thread, reused by every solve of that thread (pinned memory and events are expensive to create)
	int init(hipStream_t stream) {
		s = stream;
		head = count = 0;
		if (!mailbox) {
			SMM_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&mailbox), SLOTS * sizeof(int), hipHostMallocDefault));
			for (int i = 0; i < SLOTS; ++i) SMM_HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
		}
		for (int i = 0; i < SLOTS; ++i) {
			mailbox[i] = 0;
			pending[i] = false;
		}
		return SMM_HIP_OK;
	}
	~DonePoller() {
		for (int i = 0; i < SLOTS; ++i) {
			if (ev[i]) hipEventDestroy(ev[i]);
		}
		if (mailbox) hipHostFree(mailbox);
	}
	// returns 1 when a completed check saw done, 0 otherwise, <0 on error
	int post(const int* d_done) {
		int seen = 0;
		if (count == SLOTS - 1) seen = drain(true);
		if (seen) return seen;
		const int slot = (head + count) % SLOTS;
		SMM_HIP_TRY(hipMemcpyAsync(&mailbox[slot], d_done, sizeof(int), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipEventRecord(ev[slot], s));
		pending[slot] = true;
		++count;
		return drain(false);
	}
	int drain(bool block) {
		while (count > 0) {
			if (block && count >= 2) {
				SMM_HIP_TRY(hipEventSynchronize(ev[head]));
			} else {
				const hipError_t q = hipEventQuery(ev[head]);
				if (q == hipErrorNotReady) return 0;
				if (q != hipSuccess) return hipFail(q, "hipEventQuery", __FILE__, __LINE__);
			}
			const int v = mailbox[head];
			pending[head] = false;
			head = (head + 1) % SLOTS;
			--count;
			if (v) return 1;
		}
		return 0;
	}
};

// solver drivers (smm_solvers.hip)
template <typename T>
int cgDev(const smm_hip_csr* a, const T* b, const T* x0, T* x, int maxIterations, T eps, const smm_hip_precond* M, hipStream_t s,
          int* status, int* iterations, T* resnorm2);
template <typename T>
int bicgstabDev(const smm_hip_csr* a, const T* b, T* x, int maxIterations, T eps, const smm_hip_precond* M, hipStream_t s,
                int* status, int* iterations, T* resnorm);
template <typename T>
int bicgsymmetricDev(const smm_hip_csr* a, const T* b, T* x, int maxIterations, T eps, hipStream_t s, int* status, int* iterations);

// single-launch BiCGStab (smm_resident_bicg.hip): *handled = false when the matrix does not qualify or a barrier timed out (x untouched)
template <typename T>
int bicgstabResidentTry(const smm_hip_csr* a, const T* b, T* x, int maxIterations, T eps, const T* jacobiDiag, hipStream_t s, int* status,
                        int* iterations, T* resnorm, bool* handled);

// preconditioner apply (smm_precond.hip); doneFlag may be null
template <typename T>
int precondApplyDev(const smm_hip_precond* M, const T* rhs, T* x, const int* doneFlag, hipStream_t s);
// sticky error of the synchronisation-free sweeps applied on `s`; synchronises `s`.  M may be null.
int precondTakeError(const smm_hip_precond* M, hipStream_t s);

// block preconditioners (smm_precond_block.hip)
inline bool isBlockKind(int kind) { return kind == SMM_PRECOND_BLOCK_ILU0 || kind == SMM_PRECOND_BLOCK_SGS; }
template <typename T>
int blockCreateTyped(const smm_hip_csr* a, int kind, int blockRows, int levelCap, int partition, smm_hip_precond* M);
// x = M^-1 rhs; dotMode / w1 / partials: dot products of x fused into the epilogue, as in launchSpmv (partials: 2 * NPART elements)
template <typename T>
int blockApplyDev(const smm_hip_precond* M, const T* rhs, T* x, int dotMode, const T* w1, T* partials, const int* doneFlag, hipStream_t s);
template <typename T>
int blockApplySpmvDev(const smm_hip_precond* M, const T* v, T* x, int dotMode, const T* w1, T* partials, const int* doneFlag, hipStream_t s);
bool blockFuseSpmv(const smm_hip_precond* M, bool asked);
void blockDestroy(struct smm_precond_block* B);
void blockLevels(const struct smm_precond_block* B, int* lo, int* up);
int blockDefaultRows();
int blockDefaultLevelCap();

}  // namespace smm
