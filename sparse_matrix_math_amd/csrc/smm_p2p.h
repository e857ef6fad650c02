// smm_p2p.h -- the row-partitioned solvers' data movement WITHOUT collectives launched per exchange: peers write straight into each other's
// memory over xGMI and signal with sequence numbers (round 5).
//
// Why: the per-iteration budget at 8 GPUs is ~190 us (one-GPU iteration 1.16-1.21 ms / 6), of which ~179 us are the rank's own kernels.  An
// RCCL exchange costs ~15 us of launch and completion on top of 4.2 MB over ONE link per neighbour (~70 us) and an RCCL all-reduce of 8-16
// bytes ~10-12 us plus two cross-stream event hops -- the r04 estimate was 4-4.6 x.  Here
//   * every rank owns a SYMMETRIC BLOCK of fine-grained device memory (coherent without kernel boundaries: remote writes are seen by
//     system-scope loads), exported once with hipIpcGetMemHandle and mapped by every peer (ranks must be separate PROCESSES: between threads
//     of one process the set-up votes for the collectives -- HIP may put two streams of a process on one hardware queue, and a kernel that
//     waits for a peer's kernel would then sit in front of it; smm_dist.hip, p2pSetup): flags, reduction slots, a LANDING area per
//     halo-extended vector and a STAGING area for data this rank relays;
//   * halo: a push kernel on the communicator's side stream copies the boundary slices of the freshly updated vector into the
//     neighbours' landing areas, one share DIRECTLY and the other shares through RELAY ranks (rank g -> r -> g + 1 uses the links
//     g -> r and r -> g + 1, which a nearest-neighbour exchange leaves idle: 5 of 7 per GPU); a relay's forward kernel, enqueued ahead on
//     ITS side stream, waits for the staged share's flag and passes it on.  Every part ends with a release fence and a sequence number in
//     the destination's flag word.  The destination's land kernel (on the solver's stream, in front of the remote block's SpMV) waits for
//     all parts of all its segments and copies the landing area into the halo of the ordinary, cached vector the SpMV reads.  Pure data
//     movement: the bits of the one-path exchange;
//   * scalars: one single-workgroup kernel per reduction point adds the rank's own partials (what the producing kernel left: nobody else
//     finishes them in this transport), writes the 1-2 totals into its slot in EVERY rank's block,
//     waits for all slots of that sequence number and adds them in rank order -- the same order, hence the same bits, on every rank -- into
//     the place the RCCL all-reduce would have left them: the update kernels are unchanged;
//   * every wait is bounded (SMM_HIP_P2P_TIMEOUT_S, default 20 s): an expired wait raises the block's error word, every later wait of the
//     rank returns at once, and the host -- which reads the word wherever it reads `done` -- fails the call with SMM_HIP_ERR_COMM.
// r06: the DEFAULT between processes (SMM_HIP_P2P=0 on any rank turns it off): taken when every rank succeeds in mapping every peer and in
// the create-time self-test -- every reduction point through the slots, five halo exchanges through every path --; the halo part failing
// leaves the scalars in the slots and the halo with the communicator (the hybrid), the scalar part failing leaves everything with the
// communicator's collectives (smm_dist.hip, p2pSetup).
#pragma once
#include <hip/hip_runtime.h>

#include "smm_device.h"

namespace smm {

constexpr int P2P_KINDS = 3;       // which halo-extended vector of the solver: p, s, x
constexpr int P2P_MAX_WORLD = 16;
constexpr int P2P_MAX_PATHS = 8;   // the direct path + up to 7 relays of one segment
constexpr int P2P_MAX_JOBS = 64;   // relay jobs one rank can serve
constexpr int P2P_RED_POINTS = 4;  // reduction points of a loop (BiCGStab: 3, CG: 2) + the set-up one
constexpr int P2P_BLOCKS_PER_JOB = 8;
constexpr int P2P_TPB = 256;

struct P2PSlot {
	unsigned long long bits[2];
	unsigned long long seq;
	unsigned long long pad;
};

// offset 0 of every rank's symmetric block; zeroed at creation.  Every word another rank polls or writes sits in a 32-byte slot of its own.
struct P2PHeader {
	unsigned long long err;  // != 0: a bounded wait of THIS rank expired (written locally, read by the host)
	unsigned long long pad[7];
	unsigned long long haloFlag[P2P_KINDS][P2P_MAX_WORLD][P2P_MAX_PATHS];  // [kind][source rank][path]: sequence number of the last part that landed
	unsigned long long stageFlag[P2P_KINDS][P2P_MAX_JOBS];                 // [kind][relay job]: the staged share of that sequence number is complete
	unsigned long long ackFlag[P2P_KINDS][P2P_MAX_WORLD];                  // [kind][destination rank]: that rank has copied my segment of this sequence number out of its landing area
	P2PSlot slot[P2P_RED_POINTS][2][P2P_MAX_WORLD];                        // [reduction point][parity][source rank]
};

// one contiguous piece this rank moves: a push job reads the rank's own halo-extended vector, a forward job its staging area
struct P2PJob {
	long long srcOff;                    // element offset: in the halo-extended vector (push) / in the staging area of the kind (forward)
	char* dst[P2P_KINDS];                // where the piece goes (a peer's landing or staging area, as mapped here)
	unsigned long long* flag[P2P_KINDS]; // the word to set there once the piece has landed
	int count;                           // elements
	int waitJob;                         // forward: the stageFlag index to wait for; push: -1
	int ackFrom;                         // push: the rank the piece is for (its landing area is free once it acknowledged the previous exchange)
};

// one segment the land kernel copies from the landing area into the halo of the cached vector
struct P2PLandSeg {
	long long landOff;  // element offset in the landing area of the kind
	long long extOff;   // element offset in the halo-extended vector
	int count;
	int src;            // source rank
	int paths;          // parts (direct + relays) that must have landed
	unsigned long long* ack[P2P_KINDS];  // the source's ackFlag[kind][me], as mapped here
};

__device__ __forceinline__ unsigned long long p2pLoad(const unsigned long long* p) {
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void p2pStore(unsigned long long* p, unsigned long long v) {
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one lane waits until *word >= seq; false when the bound expired or the rank has already given up (hdr->err)
// what: which wait (1 a push for the destination's acknowledgement, 2 a forward for its staged share, 3 the land kernel for a part, 4 a
// reduction for a rank's slot) and whom / what it waited for -- left in the error word for the host's message
__device__ __forceinline__ bool p2pWaitGe(const unsigned long long* word, unsigned long long seq, P2PHeader* hdr, long long ticks, unsigned what) {
	const long long t0 = wall_clock64();
	for (unsigned spins = 0;; ++spins) {
		if (p2pLoad(word) >= seq) return true;
		__builtin_amdgcn_s_sleep(2);
		if ((spins & 63u) == 63u) {
			if (p2pLoad(&hdr->err) != 0) return false;
			if (wall_clock64() - t0 > ticks) {
				// first writer wins: the word names the wait that expired FIRST -- the later ones are its consequences
				unsigned long long none = 0ull;
				(void)__hip_atomic_compare_exchange_strong(&hdr->err, &none, (static_cast<unsigned long long>(what) << 32) | (seq & 0xFFFFFFFFull), __ATOMIC_RELAXED,
				                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				return false;
			}
		}
	}
}

// ---- copies: grid (P2P_BLOCKS_PER_JOB, jobs); the last workgroup of a job to finish publishes the job's flag -----------------------------
// (16-byte accesses at element alignment on both sides: gfx950 global accesses may be unaligned; the pieces are a few MB)
// WT (the land kernel): the destination -- the halo of the cached vector -- is stored WRITE-THROUGH (sc1: the bytes leave this XCD's L2 at once),
// because the land kernel itself raises the word the one-launch SpMV polls, in the same launch: that kernel's workgroups, on every XCD, read the
// halo with sc1 loads as soon as they see the word (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores, every storing wave drained,
// then the flag).  16-byte accesses at element alignment either way.
template <typename T, bool WT = false>
__device__ __forceinline__ void p2pCopyPiece(const T* __restrict__ src, T* __restrict__ dst, int count, int part, int parts, bool coherentLoads) {
	constexpr int VEC = 16 / sizeof(T);
	typedef T V __attribute__((ext_vector_type(VEC)));
	typedef V U __attribute__((aligned(sizeof(T))));
	const long long lo = static_cast<long long>(count) * part / parts, hi = static_cast<long long>(count) * (part + 1) / parts;
	const long long nvec = (hi - lo) / VEC;
	for (long long i = threadIdx.x; i < nvec; i += blockDim.x) {
		const U* sp = reinterpret_cast<const U*>(src + lo + i * VEC);
		U v;
		// (staging / landing areas are fine-grained memory written by another device and read once: streamed past the caches)
		if (coherentLoads) v = __builtin_nontemporal_load(sp);
		else v = *sp;
		if constexpr (WT) {
			const V w = v;
			asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst + lo + i * VEC), "v"(w) : "memory");
		} else {
			*reinterpret_cast<U*>(dst + lo + i * VEC) = v;
		}
	}
	for (long long i = lo + nvec * VEC + threadIdx.x; i < hi; i += blockDim.x) {
		const T v = coherentLoads ? __builtin_nontemporal_load(src + i) : src[i];
		if constexpr (WT) __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else dst[i] = v;
	}
}

// kind: which vector; seq: this exchange's sequence number.  jobs[blockIdx.y]; counters[blockIdx.y] counts finished workgroups (left at 0).
template <typename T, bool FORWARD>
__global__ __launch_bounds__(P2P_TPB) void p2pCopyKernel(const P2PJob* __restrict__ jobs, unsigned* counters, const T* srcBase, int kind, unsigned long long seq,
                                                        P2PHeader* hdr, long long ticks) {
	__shared__ int sGo;
	const P2PJob job = jobs[blockIdx.y];
	// a forward waits for the staged share; a push waits until the destination has emptied its landing area of the PREVIOUS exchange of this
	// kind (inside a solver loop that is long past: a reduction lies between two exchanges of a vector; back-to-back distributed SpMVs need it)
	if (threadIdx.x == 0) {
		sGo = FORWARD ? (p2pWaitGe(&hdr->stageFlag[kind][job.waitJob], seq, hdr, ticks, 0x2000u | (kind << 8) | job.waitJob) ? 1 : 0)
		              : (p2pWaitGe(&hdr->ackFlag[kind][job.ackFrom], seq - 1, hdr, ticks, 0x1000u | (kind << 8) | job.ackFrom) ? 1 : 0);
	}
	__syncthreads();
	if (!sGo) return;  // (the error word is up: the destination's own wait expires or sees its rank's word; nothing is signalled)
	if (FORWARD) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
	p2pCopyPiece<T>(srcBase + job.srcOff, reinterpret_cast<T*>(job.dst[kind]), job.count, blockIdx.x, gridDim.x, FORWARD);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: this wave's stores are visible to the destination before the flag is
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned before = __hip_atomic_fetch_add(&counters[blockIdx.y], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		if (before + 1 == gridDim.x) {
			__hip_atomic_store(&counters[blockIdx.y], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
			p2pStore(job.flag[kind], seq);
		}
	}
}

// the destination: wait for every part of every segment, then landing area -> halo of the cached vector.  grid (blocks, segments)
// word != nullptr (the one-launch SpMV is waiting, smm_spmv_split.hip): the workgroup that ends the LAST segment raises *word to wordSeq --
// `total` counts finished segments (left at 0) -- so the halo path has no further launch behind this one.
template <typename T>
__global__ __launch_bounds__(P2P_TPB) void p2pLandKernel(const P2PLandSeg* __restrict__ segs, unsigned* counters, const T* landing, T* ext, int kind,
                                                        unsigned long long seq, P2PHeader* hdr, long long ticks, unsigned long long* word, unsigned long long wordSeq,
                                                        unsigned* total) {
	__shared__ int sGo;
	const P2PLandSeg g = segs[blockIdx.y];
	if (threadIdx.x == 0) sGo = 1;
	__syncthreads();
	if (threadIdx.x < g.paths) {
		if (!p2pWaitGe(&hdr->haloFlag[kind][g.src][threadIdx.x], seq, hdr, ticks, 0x3000u | (kind << 8) | (g.src << 4) | threadIdx.x)) sGo = 0;
	}
	__syncthreads();
	if (!sGo) return;
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
	p2pCopyPiece<T, true>(landing + g.landOff, ext + g.extOff, g.count, blockIdx.x, gridDim.x, true);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every storing wave drains its write-through stores before anything is signalled)
	__syncthreads();  // (every lane's loads of the landing area have returned: the values were stored)
	if (threadIdx.x == 0) {
		const unsigned before = __hip_atomic_fetch_add(&counters[blockIdx.y], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		if (before + 1 == gridDim.x) {
			__hip_atomic_store(&counters[blockIdx.y], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			p2pStore(g.ack[kind], seq);  // the source may overwrite this segment of the landing area
			if (word) {
				if (__hip_atomic_fetch_add(total, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1 == gridDim.y) {
					__hip_atomic_store(total, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(word, wordSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
		}
	}
}

// sum over the ranks of `count` (1 or 2) values at totals[0 .. count), in place, the same bits on every rank: one workgroup.
// slots[q] = rank q's header as mapped here (slots[me] = mine).
struct P2PPeers {
	P2PHeader* hdr[P2P_MAX_WORLD];
};

// what a kernel needs to run a reduction point itself (world == 0: it does not)
struct P2PSlotArgs {
	P2PPeers peers;
	int world, me, point, count;
	unsigned long long seq;
	long long ticks;
};

// The exchange itself, by ALL P2P_TPB threads of ONE workgroup: mine[0 .. count) (shared or global memory, complete and visible to the
// workgroup) goes into this rank's slot in EVERY rank's block with the sequence number; every rank's slot of that number is awaited
// (bounded) and the values are added in rank order -- the same order, hence the same bits, on every rank -- into totals[0 .. count).
template <typename T>
__device__ __forceinline__ void p2pSlotExchange(const P2PPeers& peers, int world, int me, int point, unsigned long long seq, int count, const T* mine, T* totals,
                                                long long ticks) {
	__shared__ unsigned long long sBits[P2P_MAX_WORLD][2];
	__shared__ int sOk;
	const int t = threadIdx.x;
	const int par = static_cast<int>(seq & 1ull);
	if (t == 0) sOk = 1;
	__syncthreads();
	if (t < world) {
		unsigned long long b[2] = {0ull, 0ull};
		for (int k = 0; k < count; ++k) {
			T v = mine[k];
			__builtin_memcpy(&b[k], &v, sizeof(T));
		}
		P2PSlot* slot = &peers.hdr[t]->slot[point][par][me];
		p2pStore(&slot->bits[0], b[0]);
		p2pStore(&slot->bits[1], b[1]);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
		p2pStore(&slot->seq, seq);
	}
	if (t < world) {
		P2PHeader* own = peers.hdr[me];
		P2PSlot* slot = &own->slot[point][par][t];
		if (!p2pWaitGe(&slot->seq, seq, own, ticks, 0x4000u | (point << 8) | t)) sOk = 0;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
		sBits[t][0] = p2pLoad(&slot->bits[0]);
		sBits[t][1] = p2pLoad(&slot->bits[1]);
	}
	__syncthreads();
	if (t < count && sOk) {
		T acc = T(0);
		for (int q = 0; q < world; ++q) {  // rank order: identical on every rank
			T v;
			__builtin_memcpy(&v, &sBits[q][t], sizeof(T));
			acc = q == 0 ? v : acc + v;
		}
		totals[t] = acc;
	}
}

// parts != nullptr: the rank's own totals are first formed HERE from the npart partials per quantity the producing kernel left behind --
// in the order of lastBlockSums / distFinishSums (i = t, t + 256, ...; then blockSum256): the same bits -- so that in the peer-to-peer
// transport neither the update kernels finish their sums nor a finishing launch runs (r05: ~12 us per BiCGStab iteration).  (r06: the
// one-launch SpMV finishes its sums AND runs this exchange in its last workgroup -- P2PSlotArgs --, so two of BiCGStab's three points are no
// launch at all.)
template <typename T>
__global__ __launch_bounds__(P2P_TPB) void p2pAllreduceKernel(P2PPeers peers, int world, int me, int point, unsigned long long seq, int count, T* totals,
                                                             const T* __restrict__ parts, int npart, long long ticks, const int* __restrict__ doneFlag) {
	__shared__ T sRed[4];
	__shared__ T sMine[2];
	// (no early return on `done`: every rank must publish for every reduction the others may still be waiting in; the totals of a finished
	// solve are simply not used)
	(void)doneFlag;
	const int t = threadIdx.x;
	for (int k = 0; k < count; ++k) {
		if (parts) {
			T acc = T(0);
			for (int i = t; i < npart; i += P2P_TPB) acc += parts[k * npart + i];
			const T sum = blockSum256(acc, sRed);
			if (t == 0) sMine[k] = sum;
		} else if (t == 0) {
			sMine[k] = totals[k];
		}
	}
	__syncthreads();
	p2pSlotExchange<T>(peers, world, me, point, seq, count, sMine, totals, ticks);
}

}  // namespace smm
