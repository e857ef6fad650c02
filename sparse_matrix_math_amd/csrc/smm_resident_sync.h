// smm_resident_sync.h -- what the single-launch ("resident") solvers share: the XCD-hierarchical grid barrier with bounded waits, the
// write-through publishing of rows other workgroups gather, and workgroup / grid sums that give every workgroup the same bits.
// Used by smm_resident.hip (ConjugateGradient, matrix in registers) and smm_resident_bicg.hip (BiCGStab, vectors in registers).
#pragma once
#include <mutex>

#include "smm_device.h"
#include "smm_internal.h"

namespace smm {

constexpr int RTPB = 512;      // lanes per workgroup: one workgroup per CU, 2 waves per SIMD, up to 256 VGPRs per lane
constexpr int RWAVES = RTPB / WAVE;
constexpr int MAX_XCD = 8;

struct ResidentSync {  // zeroed before every launch; every word that is polled sits on a 128-byte line of its own
	unsigned census[32];
	unsigned top[32];
	unsigned timeout[32];
	unsigned pop[MAX_XCD][32];
	unsigned arrive[MAX_XCD][32];
	unsigned gen[MAX_XCD][32];
};

template <typename T>
struct ResidentOut {
	T res;
	int iters;
	int status;
	int timedOut;
	int pad;
};

__device__ __forceinline__ int residentXcc() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0x7; }

// one lane polls `word` until it reaches `target`; false when the bound expired or another workgroup already gave up
__device__ __forceinline__ bool waitAtLeast(unsigned* word, unsigned target, unsigned* timeoutWord, long long waitTicks) {
	for (long long spins = 0;; ++spins) {
		if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
		__builtin_amdgcn_s_sleep(1);
		if ((spins & 255) == 255) {
			if (__hip_atomic_load(timeoutWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
			if (spins > waitTicks) {
				__hip_atomic_store(timeoutWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return false;
			}
		}
	}
}

struct BarrierState {
	unsigned epoch;  // barriers passed so far + 1
	unsigned pop;    // workgroups on this XCD
	unsigned nx;     // XCDs that hold at least one workgroup
	int xcc;
};

// Grid-wide barrier.  Called by all RTPB lanes of every workgroup.  Everything a workgroup stored before the call is visible to plain
// loads of every workgroup after it.  Returns false (in all lanes) when a wait timed out.
// RELEASE: the workgroups stored with plain stores (the XCD's last arriver writes the L2 back); false when everything that has to be
// seen was stored write-through (agent-scope atomic stores), which needs no write-back.
template <bool RELEASE>
__device__ __forceinline__ bool gridBarrier(ResidentSync* sy, BarrierState& st, int* sOk, long long waitTicks) {
#ifdef SMM_RESIDENT_LAB
	if (waitTicks < 0) {  // lab: no grid-wide wait at all (wrong results; what do the barriers cost?)
		__syncthreads();
		return true;
	}
#endif
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2
	__syncthreads();
	if (threadIdx.x == 0) {
		bool ok;
		const unsigned e = st.epoch;
		const unsigned before = __hip_atomic_fetch_add(&sy->arrive[st.xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (before + 1 == st.pop * e) {
			// last workgroup of this XCD: one write-back of the XCD's L2 publishes the stores of all its workgroups
			if (RELEASE) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
			__hip_atomic_fetch_add(&sy->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			ok = waitAtLeast(&sy->top[0], st.nx * e, &sy->timeout[0], waitTicks);
			__hip_atomic_store(&sy->gen[st.xcc][0], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			ok = waitAtLeast(&sy->gen[st.xcc][0], e, &sy->timeout[0], waitTicks);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		*sOk = ok ? 1 : 0;
	}
	__syncthreads();
	st.epoch += 1;
	return *sOk != 0;
}

// How the rows other workgroups gather are stored inside the loop.  Write-through (agent-scope atomic store = global_store ... sc1): the
// bytes leave for the memory side at once and the barrier needs no L2 write-back; plain: they stay dirty in the L2 until the XCD's last
// arriver writes the L2 back.  Measured on config 2 (profiles/r02/resident_lab.txt): SMM_RESIDENT_PLAIN_PUBLISH builds are the plain form.
#ifdef SMM_RESIDENT_PLAIN_PUBLISH
constexpr bool PLAIN_PUBLISH = true;
#else
constexpr bool PLAIN_PUBLISH = false;
#endif
template <typename T>
__device__ __forceinline__ void publish(T* p, T v) {
	if (PLAIN_PUBLISH) *p = v;
	else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum over the workgroup, the same value in every lane.  Order: butterfly inside each wave, then the RWAVES wave sums left to right.
template <typename T>
__device__ __forceinline__ T blockSumAll(T v, T* lds /* RWAVES + 1 */) {
	v = groupSum<WAVE>(v);
	if ((threadIdx.x & (WAVE - 1)) == 0) lds[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) {
		T s = lds[0];
#pragma unroll
		for (int w = 1; w < RWAVES; ++w) s += lds[w];
		lds[RWAVES] = s;
	}
	__syncthreads();
	const T r = lds[RWAVES];
	__syncthreads();
	return r;
}

// total of the per-workgroup partial sums (slot t by lane t), the same bits in every workgroup
template <typename T>
__device__ __forceinline__ T sumSlots(const T* slots, T* lds) {
	T v = T(0);
	if (threadIdx.x < gridDim.x) v = __hip_atomic_load(slots + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return blockSumAll(v, lds);
}

// two grid-barrier kernels must never share the chip: each would wait for CUs the other holds (one mutex for every resident solver)
std::mutex& residentMutex();

}  // namespace smm
