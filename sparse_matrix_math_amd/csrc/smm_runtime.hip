// smm_runtime.hip -- device selection, error text, library stream, caching allocator, CSR handles.
#include <atomic>
#include <cstdarg>
#include <chrono>
#include <map>
#include <thread>

#include "smm_internal.h"

namespace smm {

static thread_local std::string g_lastError;
static std::mutex g_mutex;
static bool g_inited = false;
static int g_device = -1;
static int g_cus = 0;
static hipStream_t g_stream = nullptr;

void setError(const char* fmt, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_lastError = buf;
}

int hipFail(hipError_t e, const char* what, const char* file, int line) {
	setError("HIP error %d (%s) at %s:%d in `%s`", static_cast<int>(e), hipGetErrorString(e), file, line, what);
	return e == hipErrorOutOfMemory ? SMM_HIP_ERR_NOMEM : SMM_HIP_ERR_HIP;
}

__global__ void countLeadingEmpty(int rows, const int* __restrict__ start, int* __restrict__ out);

static int initLocked(int device) {
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		setError("no HIP device available (hipGetDeviceCount -> %d, count %d); libsmm_hip has no CPU fallback", static_cast<int>(e), count);
		return SMM_HIP_ERR_NO_DEVICE;
	}
	if (device < 0 || device >= count) {
		setError("device %d out of range (0..%d)", device, count - 1);
		return SMM_HIP_ERR_INVALID;
	}
	if (g_inited && device == g_device) {
		return SMM_HIP_OK;
	}
	SMM_HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	SMM_HIP_TRY(hipGetDeviceProperties(&prop, device));
	g_cus = prop.multiProcessorCount;
	if (g_stream) {
		forgetStream(g_stream);
		hipStreamDestroy(g_stream);
		g_stream = nullptr;
	}
	SMM_HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
	g_device = device;
	g_inited = true;
	static const bool preload = [] {
		const char* env = getenv("SMM_HIP_PRELOAD");
		return env ? atoi(env) != 0 : true;
	}();
	if (preload) {
		SetupTrace trace("init: code objects of the hot path");
		hipFuncAttributes attr;
		(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(countLeadingEmpty));
		(void)hipGetLastError();
		preloadSpmvUnit();
		preloadPatternUnit();
		preloadMarchUnit();
		preloadBlas1Unit();
		preloadSolversUnit();
		{
			// (r05: the first single-launch solve of a process paid 5-7 ms for its code object -- config 5's stand-in 20.4 ms instead of 13.6)
			SetupTrace traceRes("init:   of which the single-launch solvers");
			preloadResidentUnits();
		}
	}
	return SMM_HIP_OK;
}

int ensureInit() {
	std::lock_guard<std::mutex> lock(g_mutex);
	if (g_inited) {
		// another thread / torch may have switched the current device
		hipError_t e = hipSetDevice(g_device);
		if (e != hipSuccess) return hipFail(e, "hipSetDevice", __FILE__, __LINE__);
		return SMM_HIP_OK;
	}
	int dev = 0;
	if (const char* env = getenv("SMM_HIP_DEVICE")) dev = atoi(env);
	return initLocked(dev);
}

hipStream_t libStream() { return g_stream; }
int numCUs() { return g_cus > 0 ? g_cus : 256; }

// ---- caching allocator: free blocks keyed by size, reused exactly --------------------------------------
// Every `_dev` entry point is asynchronous on the caller's stream, and handles are destroyed (Python __del__, error returns)
// while kernels that read their buffers may still be queued.  A freed block therefore first goes to a QUARANTINE.  The quarantine
// is kept in EPOCHS: when an allocation wants a quarantined size, the blocks freed so far are closed into an epoch and one event is
// recorded on EVERY stream the library has ever been given work on (noteStream; a handful -- and never only the "recent" ones: a
// `_dev` call notes its stream when it starts and frees its temporaries when it returns, with its kernels still queued); an epoch's
// blocks become reusable once all its events have completed -- work enqueued BEFORE the free on any of those streams is then done.
// r06: NO HOST THREAD EVER WAITS for an epoch.  An allocation whose size sits in an epoch that has not completed takes fresh memory
// (hipMalloc) instead.  r03-r05 polled the epoch's events here, and those events sit on every noted stream of the PROCESS: with ranks
// that are threads of one process and the peer-to-peer transport (smm_p2p.h), a rank entering a solve waited in devAlloc for an event
// recorded behind a PEER's land kernel, which itself waited (on the device) for the push kernel this rank was about to enqueue -- a
// cycle only the transport's bounded waits broke (the time-outs of gpurun_out/r05/p2p_thread.txt: the Jacobi solve's scratch vector,
// allocated between the stand-alone SpMV and the solver's first exchange).  In the steady state of solve after
// solve a solve ends with a synchronise of its stream, the epochs are complete when the next allocation looks, and nothing changes.
static std::mutex g_allocMutex;
static std::multimap<size_t, void*> g_free;  // safe to hand out
static std::map<void*, size_t> g_live;
struct QuarantineEpoch {
	unsigned long long id = 0;
	std::multimap<size_t, void*> blocks;
	std::vector<hipEvent_t> events;
	bool needIdle = false;  // an event could not be recorded (a stream its owner destroyed meanwhile): complete only once every noted stream is idle
};
static std::multimap<size_t, void*> g_open;       // freed since the last close
static std::vector<QuarantineEpoch> g_epochs;     // closed, oldest first
static std::vector<hipStream_t> g_recentStreams;  // every stream the library has been given work on (small: linear search)
static std::vector<hipEvent_t> g_eventPool;       // (under g_allocMutex)
static std::mutex g_streamMutex;

// The list is bounded: the most recently used MAX_NOTED_STREAMS caller streams, most recent last.  A stream that falls
// off the end may still have work queued that reads a block freed later, and no epoch will record on it any more: an event is
// recorded on it AS IT LEAVES the list and joins the next epoch that is closed (r03-r05 drained the device there instead).
constexpr size_t MAX_NOTED_STREAMS = 32;
static std::vector<hipEvent_t> g_strayEvents;  // of evicted streams, waiting for the next epoch (under g_streamMutex)
static bool g_strayUnrecorded = false;         // an evicted stream could not be given an event: the next epoch needs every stream idle

void noteStream(hipStream_t s) {
	std::lock_guard<std::mutex> lock(g_streamMutex);
	for (size_t i = 0; i < g_recentStreams.size(); ++i) {
		if (g_recentStreams[i] == s) {
			if (i + 1 != g_recentStreams.size()) {  // move to the back: most recently used
				g_recentStreams.erase(g_recentStreams.begin() + static_cast<long>(i));
				g_recentStreams.push_back(s);
			}
			return;
		}
	}
	if (g_recentStreams.size() >= MAX_NOTED_STREAMS) {
		hipStream_t old = g_recentStreams.front();
		g_recentStreams.erase(g_recentStreams.begin());
		hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
		const bool inCapture = hipStreamIsCapturing(old, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone;
		(void)hipGetLastError();
		if (!inCapture && hipStreamQuery(old) != hipSuccess) {
			(void)hipGetLastError();
			hipEvent_t ev = nullptr;
			if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, old) == hipSuccess) {
				g_strayEvents.push_back(ev);
			} else {
				(void)hipGetLastError();
				if (ev) (void)hipEventDestroy(ev);
				g_strayUnrecorded = true;
			}
		}
	}
	g_recentStreams.push_back(s);
}

// a stream of the library's own that is about to be destroyed (already synchronised by its owner)
void forgetStream(hipStream_t s) {
	std::lock_guard<std::mutex> lock(g_streamMutex);
	for (size_t i = 0; i < g_recentStreams.size(); ++i) {
		if (g_recentStreams[i] == s) {
			g_recentStreams.erase(g_recentStreams.begin() + static_cast<long>(i));
			return;
		}
	}
}

static void closeEpochLocked() {
	if (g_open.empty()) return;
	static unsigned long long nextId = 1;
	QuarantineEpoch ep;
	ep.id = nextId++;
	ep.blocks.swap(g_open);
	std::vector<hipStream_t> streams;
	{
		std::lock_guard<std::mutex> lock(g_streamMutex);
		streams = g_recentStreams;
		ep.events.swap(g_strayEvents);
		ep.needIdle = g_strayUnrecorded;
		g_strayUnrecorded = false;
	}
	bool haveLib = false;
	for (hipStream_t k : streams) haveLib = haveLib || k == g_stream;
	if (!haveLib && g_stream) streams.push_back(g_stream);  // the host-pointer entry points run here
	for (hipStream_t k : streams) {
		// a stream the caller is capturing into a hipGraph executes nothing now: recording on it would put a node into the caller's graph
		// (or invalidate the capture), and there is nothing queued on it to wait for
		hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
		if (hipStreamIsCapturing(k, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) continue;
		(void)hipGetLastError();
		// A stream with nothing pending needs no event: whatever was queued on it before the frees is done.  This is the steady state of
		// solve after solve (a solve ends with a synchronise of its stream): no record, no polls.  (r05 first suspected this wait of the 18-22 ms
		// the SECOND host-pointer solve of a process lost; with the events gone the same delay showed up in the next copy from the caller's
		// pageable memory -- the HIP runtime's in-place pinning, see hostToDev below.)
		if (hipStreamQuery(k) == hipSuccess) continue;
		(void)hipGetLastError();
		hipEvent_t ev = nullptr;
		if (!g_eventPool.empty()) {
			ev = g_eventPool.back();
			g_eventPool.pop_back();
		} else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
			(void)hipGetLastError();
			ep.needIdle = true;
			continue;
		}
		if (hipEventRecord(ev, k) != hipSuccess) {  // a stream its owner has destroyed meanwhile: forget it
			(void)hipGetLastError();
			g_eventPool.push_back(ev);
			ep.needIdle = true;
			forgetStream(k);
			continue;
		}
		ep.events.push_back(ev);
	}
	g_epochs.push_back(std::move(ep));
}

static bool everyNotedStreamIdle() {
	std::vector<hipStream_t> streams;
	{
		std::lock_guard<std::mutex> lock(g_streamMutex);
		streams = g_recentStreams;
	}
	if (g_stream) streams.push_back(g_stream);
	for (hipStream_t k : streams) {
		if (hipStreamQuery(k) != hipSuccess) {
			(void)hipGetLastError();
			return false;
		}
	}
	return true;
}

// moves the blocks of every COMPLETED epoch to the free list.  Never waits (see the note at the top): events are queried, not
// synchronised; every epoch is looked at on its own (its events alone say that its blocks are safe), so one that stays open -- a
// peer rank's kernel that spins until this very thread has enqueued more work -- holds back nothing but its own blocks.
static void reapEpochsLocked() {
	for (size_t i = 0; i < g_epochs.size();) {
		QuarantineEpoch& ep = g_epochs[i];
		bool done = true;
		for (hipEvent_t ev : ep.events) {
			if (!done) break;
			done = hipEventQuery(ev) != hipErrorNotReady;  // (an error state does not come back: treat the event as over)
		}
		(void)hipGetLastError();
		if (done && ep.needIdle) done = everyNotedStreamIdle();
		if (!done) {
			++i;
			continue;
		}
		g_free.insert(ep.blocks.begin(), ep.blocks.end());
		g_eventPool.insert(g_eventPool.end(), ep.events.begin(), ep.events.end());
		g_epochs.erase(g_epochs.begin() + static_cast<long>(i));
	}
}

static bool quarantineHoldsLocked(size_t bytes) {
	if (g_open.count(bytes)) return true;
	for (const auto& e : g_epochs) {
		if (e.blocks.count(bytes)) return true;
	}
	return false;
}

// test hook (smm_hip_debug_fail_next_alloc): the next device allocation of at least this many bytes fails once, as if the device were full
static std::atomic<size_t> g_failNextAllocAtLeast{0};

int devAlloc(void** p, size_t bytes) {
	bytes = (bytes + 255) & ~static_cast<size_t>(255);
	if (size_t want = g_failNextAllocAtLeast.load(std::memory_order_relaxed); want != 0 && bytes >= want) {
		if (g_failNextAllocAtLeast.compare_exchange_strong(want, 0)) {
			*p = nullptr;
			setError("device allocation of %zu bytes refused (injected by smm_hip_debug_fail_next_alloc)", bytes);
			return SMM_HIP_ERR_NOMEM;
		}
	}
	{
		std::unique_lock<std::mutex> lock(g_allocMutex);
		auto it = g_free.find(bytes);
		if (it == g_free.end() && quarantineHoldsLocked(bytes)) {
			SetupTrace trace("allocator: close the epoch, take what has completed");
			closeEpochLocked();
			reapEpochsLocked();
			it = g_free.find(bytes);  // (still quarantined behind running work: fresh memory below -- never a wait)
		}
		if (it != g_free.end()) {
			*p = it->second;
			g_free.erase(it);
			g_live[*p] = bytes;
			return SMM_HIP_OK;
		}
	}
	SetupTrace traceMalloc("allocator: hipMalloc");
	hipError_t e = hipMalloc(p, bytes);
	if (e == hipErrorOutOfMemory) {
		devTrim();
		e = hipMalloc(p, bytes);
	}
	if (e != hipSuccess) {
		*p = nullptr;
		return hipFail(e, "hipMalloc", __FILE__, __LINE__);
	}
	std::lock_guard<std::mutex> lock(g_allocMutex);
	g_live[*p] = bytes;
	return SMM_HIP_OK;
}

void devFree(void* p) {
	if (!p) return;
	std::lock_guard<std::mutex> lock(g_allocMutex);
	auto it = g_live.find(p);
	if (it == g_live.end()) {
		hipFree(p);  // synchronises implicitly
		return;
	}
	g_open.emplace(it->second, p);
	g_live.erase(it);
}

void devTrim() {
	(void)hipDeviceSynchronize();  // everything queued anywhere is done: every quarantined block is free
	std::unique_lock<std::mutex> lock(g_allocMutex);
	g_free.insert(g_open.begin(), g_open.end());
	g_open.clear();
	for (auto& ep : g_epochs) {
		g_free.insert(ep.blocks.begin(), ep.blocks.end());
		g_eventPool.insert(g_eventPool.end(), ep.events.begin(), ep.events.end());
	}
	g_epochs.clear();
	for (auto& kv : g_free) hipFree(kv.second);
	g_free.clear();
}

// ---- staged host <-> device copies (declared in smm_internal.h) ---------------------------------------------------------------
namespace {
struct CopyStage {
	static constexpr int SLOTS = 4;
	static constexpr size_t CHUNK = 8u << 20;
	char* buf[SLOTS] = {};
	hipEvent_t ev[SLOTS] = {};
	bool busy[SLOTS] = {};
	int next = 0;
	bool ready = false, refused = false;
	// true: every slot has its pinned chunk and its event.  A failure releases what was set up and is remembered (the thread's copies then
	// take the caller's pointers, as with SMM_HIP_STAGED_COPIES=0): ADVICE r05 -- `buf[0] != nullptr` used to stand for "all four slots".
	bool init() {
		if (ready || refused) return ready;
		for (int i = 0; i < SLOTS; ++i) {
			if (hipHostMalloc(reinterpret_cast<void**>(&buf[i]), CHUNK, hipHostMallocDefault) != hipSuccess) buf[i] = nullptr;
			if (!buf[i] || hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
				(void)hipGetLastError();
				release();
				refused = true;
				return false;
			}
		}
		ready = true;
		return true;
	}
	void release() {
		for (int i = 0; i < SLOTS; ++i) {
			if (ev[i]) {
				if (busy[i]) (void)hipEventSynchronize(ev[i]);
				(void)hipEventDestroy(ev[i]);
			}
			if (buf[i]) (void)hipHostFree(buf[i]);
			ev[i] = nullptr;
			buf[i] = nullptr;
			busy[i] = false;
		}
		ready = false;
	}
	int wait(int i) {
		if (busy[i]) SMM_HIP_TRY(hipEventSynchronize(ev[i]));
		busy[i] = false;
		return SMM_HIP_OK;
	}
	~CopyStage() {
		release();
		(void)hipGetLastError();  // (a thread that outlives the HIP runtime: nothing left to release)
	}
};
}  // namespace

static CopyStage& copyStage() {
	static thread_local CopyStage st;  // per host thread: concurrent solves of different threads never share a chunk
	return st;
}

static bool stagedCopies() {
	static const bool on = [] {
		const char* env = getenv("SMM_HIP_STAGED_COPIES");  // 0: hand the caller's pointers to hipMemcpyAsync as before (measurements)
		return env ? atoi(env) != 0 : true;
	}();
	return on;
}

int hostToDev(void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
	if (!bytes) return SMM_HIP_OK;
	if (!stagedCopies()) {
		SMM_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
		return SMM_HIP_OK;
	}
	CopyStage& st = copyStage();
	if (!st.init()) {
		SMM_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
		return SMM_HIP_OK;
	}
	for (size_t at = 0; at < bytes; at += CopyStage::CHUNK) {
		const size_t len = std::min(CopyStage::CHUNK, bytes - at);
		const int i = st.next;
		st.next = (st.next + 1) % CopyStage::SLOTS;
		SMM_TRY(st.wait(i));  // (the chunk's previous device copy has ended)
		memcpy(st.buf[i], static_cast<const char*>(h_src) + at, len);
		SMM_HIP_TRY(hipMemcpyAsync(static_cast<char*>(d_dst) + at, st.buf[i], len, hipMemcpyHostToDevice, s));
		SMM_HIP_TRY(hipEventRecord(st.ev[i], s));
		st.busy[i] = true;
	}
	return SMM_HIP_OK;
}

int devToHost(void* h_dst, const void* d_src, size_t bytes, hipStream_t s) {
	if (!stagedCopies() || !bytes) {
		if (bytes) SMM_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
		return SMM_HIP_OK;
	}
	CopyStage& st = copyStage();
	if (!st.init()) {
		SMM_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
		return SMM_HIP_OK;
	}
	const size_t nChunks = (bytes + CopyStage::CHUNK - 1) / CopyStage::CHUNK;
	int slotOf[CopyStage::SLOTS];
	size_t issued = 0, drained = 0;
	while (drained < nChunks) {
		// keep up to SLOTS device copies in flight, hand finished chunks to the caller's array in order
		while (issued < nChunks && issued - drained < static_cast<size_t>(CopyStage::SLOTS)) {
			const int i = st.next;
			st.next = (st.next + 1) % CopyStage::SLOTS;
			SMM_TRY(st.wait(i));
			const size_t at = issued * CopyStage::CHUNK, len = std::min(CopyStage::CHUNK, bytes - at);
			SMM_HIP_TRY(hipMemcpyAsync(st.buf[i], static_cast<const char*>(d_src) + at, len, hipMemcpyDeviceToHost, s));
			SMM_HIP_TRY(hipEventRecord(st.ev[i], s));
			st.busy[i] = true;
			slotOf[issued % CopyStage::SLOTS] = i;
			++issued;
		}
		const int i = slotOf[drained % CopyStage::SLOTS];
		SMM_TRY(st.wait(i));
		const size_t at = drained * CopyStage::CHUNK, len = std::min(CopyStage::CHUNK, bytes - at);
		memcpy(static_cast<char*>(h_dst) + at, st.buf[i], len);
		++drained;
	}
	SMM_HIP_TRY(hipStreamSynchronize(s));  // (everything else queued on `s` too: the callers' contract)
	return SMM_HIP_OK;
}

// ---- live SpMV timing -------------------------------------------------------------------------------------
static std::mutex g_profMutex;
static bool g_profOn = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_profEvents;  // pool, reused after reset
static size_t g_profUsed = 0;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_profWaitEvents;  // (end of own work on the waiting stream, end of the awaited work)
static size_t g_profWaitUsed = 0;
static std::vector<bool> g_profWaitHalf;  // slot: the waiting side has been recorded too

int profBegin(hipStream_t s) {
	std::lock_guard<std::mutex> lock(g_profMutex);
	if (!g_profOn) return -1;
	if (g_profUsed == g_profEvents.size()) {
		hipEvent_t a, b;
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
		g_profEvents.emplace_back(a, b);
	}
	const int slot = static_cast<int>(g_profUsed++);
	(void)hipEventRecord(g_profEvents[slot].first, s);
	return slot;
}

// second channel: how long a stream had to WAIT for another one (the row-partitioned SpMV: A_rem on the caller's stream waits for the
// halo exchange on the communicator's).  profWaitAwaited records "the awaited work ends here" on the other stream and returns a slot
// (-1: profiling off); profWaitWaiting records "the waiting stream's own work ends here".  The exposed wait of the pair is
// max(0, awaited - waiting).
int profWaitAwaited(hipStream_t awaited) {
	std::lock_guard<std::mutex> lock(g_profMutex);
	if (!g_profOn) return -1;
	if (g_profWaitUsed == g_profWaitEvents.size()) {
		hipEvent_t a, b;
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
		g_profWaitEvents.emplace_back(a, b);
	}
	const int slot = static_cast<int>(g_profWaitUsed++);
	(void)hipEventRecord(g_profWaitEvents[static_cast<size_t>(slot)].second, awaited);
	g_profWaitHalf.resize(g_profWaitEvents.size(), false);
	g_profWaitHalf[static_cast<size_t>(slot)] = false;
	return slot;
}

void profWaitWaiting(int slot, hipStream_t waiting) {
	if (slot < 0) return;
	std::lock_guard<std::mutex> lock(g_profMutex);
	if (static_cast<size_t>(slot) >= g_profWaitUsed) return;  // (read and reset in between)
	(void)hipEventRecord(g_profWaitEvents[static_cast<size_t>(slot)].first, waiting);
	g_profWaitHalf[static_cast<size_t>(slot)] = true;
}

void profEnd(int slot, hipStream_t s) {
	if (slot < 0) return;
	std::lock_guard<std::mutex> lock(g_profMutex);
	(void)hipEventRecord(g_profEvents[slot].second, s);
}

// number of leading rows without entries = firstActiveStart (ref:1619-1628): start[] is non-decreasing, so
// it is the count of i in [0,rows) with start[i+1] == 0
__global__ void countLeadingEmpty(int rows, const int* __restrict__ start, int* __restrict__ out) {
	int local = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x) {
		local += start[i + 1] == 0 ? 1 : 0;
	}
	if (local) atomicAdd(out, local);
}

int ensureCsrReady(const smm_hip_csr* cm, hipStream_t s, bool streamKnown) {
	auto* m = const_cast<smm_hip_csr*>(cm);
	std::lock_guard<std::mutex> lock(m->readyMutex);
	if (m->ready) return SMM_HIP_OK;
	SetupTrace trace("csr: nnz / first active row / typical row");
	if (!streamKnown) {
		// a host-side query (csr_info, set_kernel, precond_create ...) has no stream to be ordered behind: the arrays may still be
		// being written on any of the caller's streams
		SMM_HIP_TRY(hipDeviceSynchronize());
		s = libStream();
	}
	DevBuf<int> cnt;  // (released on every return path)
	SMM_TRY(cnt.alloc(1));
	int* d_cnt = cnt.p;
	SMM_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int), s));
	int nnz = 0;
	SMM_HIP_TRY(hipMemcpyAsync(&nnz, m->d_start + m->rows, sizeof(int), hipMemcpyDeviceToHost, s));
	if (m->rows > 0) {
		const int grid = std::min(1024, (m->rows + 255) / 256);
		countLeadingEmpty<<<grid, 256, 0, s>>>(m->rows, m->d_start, d_cnt);
	}
	int first = 0, mid[2] = {0, 0};
	SMM_HIP_TRY(hipMemcpyAsync(&first, d_cnt, sizeof(int), hipMemcpyDeviceToHost, s));
	if (m->rows > 0) SMM_HIP_TRY(hipMemcpyAsync(mid, m->d_start + m->rows / 2, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));  // the caller's stream only (16 bytes come back)
	m->stream_mid_len = mid[1] - mid[0];
	m->nnz = nnz;
	m->firstActiveStart = first;
	if (!m->kernelForced) chooseSpmvConfig(m);
	m->ready = true;
	return SMM_HIP_OK;
}

template <typename T>
static int csrCreate(int rows, int cols, const int* start, const int* positions, const T* values, bool onDevice, smm_hip_csr** out) {
	if (!out) {
		setError("csr_create: out is null");
		return SMM_HIP_ERR_INVALID;
	}
	*out = nullptr;
	if (rows < 0 || cols < 0 || !start) {
		setError("csr_create: bad shape or null start");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	auto* m = new smm_hip_csr();
	m->rows = rows;
	m->cols = cols;
	m->dtype = dtypeOf<T>();
	if (onDevice) {
		m->d_start = const_cast<int*>(start);
		m->d_positions = const_cast<int*>(positions);
		m->d_values = const_cast<T*>(values);
		m->owns = false;
	} else {
		const int nnz = start[rows];
		if (nnz < 0 || (nnz > 0 && (!positions || !values))) {
			delete m;
			setError("csr_create: start[rows] negative or null positions/values");
			return SMM_HIP_ERR_INVALID;
		}
		m->owns = true;
		int st = devAlloc(reinterpret_cast<void**>(&m->d_start), (static_cast<size_t>(rows) + 1) * sizeof(int));
		if (st == SMM_HIP_OK) st = devAlloc(reinterpret_cast<void**>(&m->d_positions), static_cast<size_t>(nnz ? nnz : 1) * sizeof(int));
		if (st == SMM_HIP_OK) st = devAlloc(&m->d_values, static_cast<size_t>(nnz ? nnz : 1) * sizeof(T));
		if (st != SMM_HIP_OK) {
			smm_hip_csr_destroy(m);
			return st;
		}
		hipStream_t s = libStream();
		hipError_t e = hipMemcpyAsync(m->d_start, start, (static_cast<size_t>(rows) + 1) * sizeof(int), hipMemcpyHostToDevice, s);
		if (e == hipSuccess && nnz) e = hipMemcpyAsync(m->d_positions, positions, static_cast<size_t>(nnz) * sizeof(int), hipMemcpyHostToDevice, s);
		if (e == hipSuccess && nnz) e = hipMemcpyAsync(m->d_values, values, static_cast<size_t>(nnz) * sizeof(T), hipMemcpyHostToDevice, s);
		if (e == hipSuccess) e = hipStreamSynchronize(s);
		if (e != hipSuccess) {
			smm_hip_csr_destroy(m);
			return hipFail(e, "csr upload", __FILE__, __LINE__);
		}
	}
	if (!onDevice) {
		// host arrays: everything is known here
		m->nnz = start[rows];
		m->firstActiveStart = rows;
		for (int i = 0; i < rows; ++i) {  // ref:1619-1628
			if (start[i + 1] != 0) {
				m->firstActiveStart = i;
				break;
			}
		}
		if (rows > 0) m->stream_mid_len = start[rows / 2 + 1] - start[rows / 2];
		chooseSpmvConfig(m);
		m->ready = true;
	}
	*out = m;
	return SMM_HIP_OK;
}

}  // namespace smm

using namespace smm;

extern "C" {

int smm_hip_init(int device) {
	std::lock_guard<std::mutex> lock(g_mutex);
	return initLocked(device);
}

int smm_hip_shutdown(void) {
	std::lock_guard<std::mutex> lock(g_mutex);
	if (!g_inited) return SMM_HIP_OK;
	hipDeviceSynchronize();
	devTrim();
	if (g_stream) {
		forgetStream(g_stream);
		hipStreamDestroy(g_stream);
	}
	g_stream = nullptr;
	g_inited = false;
	g_device = -1;
	return SMM_HIP_OK;
}

const char* smm_hip_last_error(void) { return g_lastError.c_str(); }

int smm_hip_uses_std_fma(void) {
#ifdef SMM_WITH_STD_FMA
	return 1;
#else
	return 0;
#endif
}

int smm_hip_device_info(char* name, size_t name_cap, int* cus, size_t* hbm_bytes) {
	SMM_TRY(ensureInit());
	hipDeviceProp_t prop;
	SMM_HIP_TRY(hipGetDeviceProperties(&prop, g_device));
	if (name && name_cap) {
		snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
	}
	if (cus) *cus = prop.multiProcessorCount;
	if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
	return SMM_HIP_OK;
}

int smm_hip_stream_synchronize(smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	SMM_HIP_TRY(hipStreamSynchronize(pickStream(stream)));
	return SMM_HIP_OK;
}

int smm_hip_debug_fail_next_alloc(size_t min_bytes) {
	g_failNextAllocAtLeast.store(min_bytes, std::memory_order_relaxed);
	return SMM_HIP_OK;
}

int smm_hip_profile_enable(int on) {
	std::lock_guard<std::mutex> lock(g_profMutex);
	g_profOn = on != 0;
	return SMM_HIP_OK;
}

int smm_hip_profile_read(double* spmv_ms, long long* spmv_launches, int reset) {
	SMM_TRY(ensureInit());
	std::lock_guard<std::mutex> lock(g_profMutex);
	double total = 0.0;
	for (size_t i = 0; i < g_profUsed; ++i) {
		SMM_HIP_TRY(hipEventSynchronize(g_profEvents[i].second));
		float ms = 0.f;
		SMM_HIP_TRY(hipEventElapsedTime(&ms, g_profEvents[i].first, g_profEvents[i].second));
		total += ms;
	}
	if (spmv_ms) *spmv_ms = total;
	if (spmv_launches) *spmv_launches = static_cast<long long>(g_profUsed);
	if (reset) g_profUsed = 0;
	return SMM_HIP_OK;
}

int smm_hip_profile_read_waits(double* exposed_ms, long long* pairs, int reset) {
	SMM_TRY(ensureInit());
	std::lock_guard<std::mutex> lock(g_profMutex);
	double total = 0.0;
	for (size_t i = 0; i < g_profWaitUsed; ++i) {
		if (i >= g_profWaitHalf.size() || !g_profWaitHalf[i]) continue;
		SMM_HIP_TRY(hipEventSynchronize(g_profWaitEvents[i].first));
		SMM_HIP_TRY(hipEventSynchronize(g_profWaitEvents[i].second));
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, g_profWaitEvents[i].first, g_profWaitEvents[i].second) != hipSuccess) {
			(void)hipGetLastError();
			ms = 0.f;
		}
		if (ms > 0.f) total += ms;
	}
	if (exposed_ms) *exposed_ms = total;
	if (pairs) *pairs = static_cast<long long>(g_profWaitUsed);
	if (reset) g_profWaitUsed = 0;
	return SMM_HIP_OK;
}

int smm_hip_csr_create_f32(int rows, int cols, const int* start, const int* positions, const float* values, smm_hip_csr** out) {
	return csrCreate<float>(rows, cols, start, positions, values, false, out);
}
int smm_hip_csr_create_f64(int rows, int cols, const int* start, const int* positions, const double* values, smm_hip_csr** out) {
	return csrCreate<double>(rows, cols, start, positions, values, false, out);
}
int smm_hip_csr_create_dev_f32(int rows, int cols, const int* d_start, const int* d_positions, const float* d_values, smm_hip_csr** out) {
	return csrCreate<float>(rows, cols, d_start, d_positions, d_values, true, out);
}
int smm_hip_csr_create_dev_f64(int rows, int cols, const int* d_start, const int* d_positions, const double* d_values, smm_hip_csr** out) {
	return csrCreate<double>(rows, cols, d_start, d_positions, d_values, true, out);
}

int smm_hip_csr_destroy(smm_hip_csr* m) {
	if (!m) return SMM_HIP_OK;
	if (m->owns) {
		devFree(m->d_start);
		devFree(m->d_positions);
		devFree(m->d_values);
	}
	devFree(m->d_rowblocks);
	devFree(m->d_pat_off);
	devFree(m->d_pat_masks);
	devFree(m->d_pat_codes);
	devFree(m->d_pat_cval);
	devFree(m->d_res_ell);
	devFree(m->d_pat_masks32);
	devFree(m->d_pat_masks8);
	devFree(m->d_pat_rowblocks);
	delete m;
	return SMM_HIP_OK;
}

int smm_hip_csr_info(const smm_hip_csr* m, int* rows, int* cols, int* nnz, int* dtype, int* first_active_start) {
	if (!m) {
		setError("csr_info: null handle");
		return SMM_HIP_ERR_INVALID;
	}
	if (nnz || first_active_start) SMM_TRY(ensureCsrReady(m, nullptr, false));
	if (rows) *rows = m->rows;
	if (cols) *cols = m->cols;
	if (nnz) *nnz = m->nnz;
	if (dtype) *dtype = m->dtype;
	if (first_active_start) *first_active_start = m->firstActiveStart;
	return SMM_HIP_OK;
}

}  // extern "C"
