// smm_device.h -- device-side helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>

namespace smm {

constexpr int WAVE = 64;  // CDNA wavefront

// _smm_fma (ref:28-36): a*x+b with two roundings by default, fma(a,x,b) under SMM_WITH_STD_FMA.
// The library is compiled with -ffp-contract=off so the first form is never contracted behind our back.
template <typename T>
__device__ __forceinline__ T smmFma(T a, T x, T b) {
#ifdef SMM_WITH_STD_FMA
	return __builtin_fma(a, x, b);
#else
	return a * x + b;
#endif
}
__device__ __forceinline__ float smmFma(float a, float x, float b) {
#ifdef SMM_WITH_STD_FMA
	return __builtin_fmaf(a, x, b);
#else
	return a * x + b;
#endif
}

// butterfly sum over groups of L adjacent lanes (L power of two <= 64); every lane of the group ends with
// the same value, and the order of additions is fixed by the lane numbers (deterministic)
template <int L, typename T>
__device__ __forceinline__ T groupSum(T v) {
#pragma unroll
	for (int o = L / 2; o > 0; o >>= 1) {
		v += __shfl_xor(v, o, WAVE);
	}
	return v;
}

// Sum over a 256-thread workgroup.  Result valid in thread 0.  Order: butterfly inside each wave, then the
// 4 wave sums left to right -- fixed, so every launch with the same data gives the same bits.
template <typename T>
__device__ __forceinline__ T blockSum256(T v, T* lds4) {
	v = groupSum<WAVE>(v);
	const int lane = threadIdx.x & (WAVE - 1);
	const int wave = threadIdx.x >> 6;
	if (lane == 0) lds4[wave] = v;
	__syncthreads();
	T r = T(0);
	if (threadIdx.x == 0) {
		r = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
	}
	__syncthreads();
	return r;
}

// ---------------------------------------------------------------------------------------------------------
// "Last workgroup done" completion of a fused reduction.  Every workgroup of a launch has stored its partial sum(s) to
// partials[k * npart + blockIdx.x] (k < nsets).  Instead of a separate one-workgroup kernel that adds them, the workgroup that
// finishes LAST adds the npart partials of each quantity -- in the same fixed order (i = t, t + 256, ...; then blockSum256), so the
// bits are those of the separate kernel -- and stores the totals to totals[0 .. nsets).  One launch (and one dependent kernel
// boundary, ~1.5-2 us) fewer per reduction: the row-partitioned solvers (smm_dist.hip) all-reduce `totals` right after.
//
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility).  r01-r04 released at agent scope in every workgroup: on gfx950 that is a
// WRITE-BACK OF THE XCD'S L2 per workgroup, and behind a kernel that has just stored 100+ MB of vector with ordinary stores it made the
// row-partitioned loops' update kernels 2.3 x slower than their single-GPU twins (r05: distCgR 148 us against cgFusedR's 64 us for the
// same three passes; profiles/r05/dist_cg_kernel_stats_before.csv).  Now the few words that must cross workgroups are published the way
// the single-launch solvers publish rows (smm_resident_sync.h): each workgroup re-stores its own slot(s) -- and its share of the slots no
// workgroup owns, which are zero -- WRITE-THROUGH (agent-scope atomic stores), drains them (s_waitcnt vmcnt(0)), meets at a barrier and
// lane 0 takes a ticket with a relaxed agent-scope atomic add (two levels, below); no cache is written back.  The workgroup that ends last
// acquires at agent scope and reads the partials with agent-scope (cache-bypassing) loads.  The last workgroup resets the ticket
// counter, so the buffer is ready for the next launch on the stream.  (The caller's own plain stores of the same values stay: a launch
// that is not asked to finish its sums leaves them to the next kernel as before.)
// Lane 0 of every workgroup must be the one that stored partials[k * npart + blockIdx.x]; workgroups have 256 threads; all of them must
// call this (it contains barriers).
// ---------------------------------------------------------------------------------------------------------
// Returns (to every thread of the workgroup alike) whether THIS workgroup was the last one and formed the totals.
template <typename T>
__device__ __forceinline__ bool lastBlockSums(T* partials, int npart, int nsets, T* totals, unsigned* ticket) {
	__shared__ int sIsLast;
	__shared__ T sRed[4];
	if (threadIdx.x == 0) {
		for (int k = 0; k < nsets; ++k) {
			T* slot = partials + k * npart + blockIdx.x;
			__hip_atomic_store(slot, *slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	for (int i = gridDim.x + blockIdx.x * 256 + threadIdx.x; i < npart; i += gridDim.x * 256) {
		for (int k = 0; k < nsets; ++k) __hip_atomic_store(partials + k * npart + i, T(0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) {
		// two levels: the workgroups of a launch are dealt to SUBS sub-counters (ticket[1 + ...]); the last one of each takes a ticket of
		// the top counter (ticket[0]).  With ONE counter the 2048 read-modify-writes of a launch queue up on one address: ~25 us at the end of
		// every kernel that finishes its sums (r05: distBicgR 33 us for 20 MB of vectors).  Counters are left at 0 by whoever ends them.
		constexpr unsigned SUBS = 16;  // = PARTS_TICKETS (smm_internal.h)
		const unsigned nsub = gridDim.x < SUBS ? gridDim.x : SUBS;
		const unsigned sub = blockIdx.x % nsub;
		const unsigned members = (gridDim.x - sub + nsub - 1) / nsub;
		int last = 0;
		if (__hip_atomic_fetch_add(ticket + 1 + sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
			__hip_atomic_store(ticket + 1 + sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1 ? 1 : 0;
		}
		if (last) {
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		sIsLast = last;
	}
	__syncthreads();
	if (!sIsLast) return false;
	for (int k = 0; k < nsets; ++k) {
		T acc = T(0);
		for (int i = threadIdx.x; i < npart; i += 256) {
			acc += __hip_atomic_load(partials + k * npart + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		const T s = blockSum256(acc, sRed);
		if (threadIdx.x == 0) totals[k] = s;
	}
	if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return true;
}

// out[] of an SpMV row with a RUN-TIME choice of the cache policy.  `if (nt) __builtin_nontemporal_store(...) else out[row] = ...` does
// not survive the compiler -- the hint is metadata, the two arms are merged into ONE plain store (r04 finding: every kernel that had the
// flag stored plainly whatever the flag said) -- and a template argument would double the instantiations of nine kernels.  The non-temporal
// arm is therefore spelled out: global_store ... nt (the data depends on everything the lane loaded for the row, including lhs[row] when
// out aliases lhs, so program order needs no memory clobber -- which would stop the compiler moving the next tile's loads across the store).
__device__ __forceinline__ void storeOut(float* p, float v, bool nt) {
	if (nt) asm volatile("global_store_dword %0, %1, off nt" : : "v"(p), "v"(v));
	else *p = v;
}
__device__ __forceinline__ void storeOut(double* p, double v, bool nt) {
	if (nt) asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v));
	else *p = v;
}

// Workgroup barrier for kernels whose waves meet ONLY in LDS.  __syncthreads() is a workgroup-scope fence + s_barrier, and the
// fence (which cannot know the address space) waits for every outstanding global access of the wave as well: vmcnt(0).  In a
// software-pipelined tile loop that drains the prefetch of the next tile and the acknowledgement of the out[] store at every
// barrier (measured on the 512^3 fp64 Laplacian: the 7 % of bytes that are stores cost 23 % of the time).  Here only LDS traffic
// (lgkmcnt) is awaited; the compiler still inserts its own waits before a loaded register is used.
__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------
// Streaming element-wise map over vectors: out[k][i] = f(in[0][i], in[1][i], ...), optionally with per-lane accumulators
// captured by f.  Measured on MI355X (tools/membw.hip, the CG x/r update on 2^27 doubles): one element per lane per trip
// 4.3-4.7 TB/s, 16-byte accesses with 4 trips' loads issued before the first store 5.7 TB/s -- the loads of a lane must be in
// flight together, and an in-place update (x and xcur may be the same vector) stops the compiler from hoisting them itself.
//   * every lane loads all its inputs for U packs of 16 bytes, then computes, then stores: element-wise aliasing between
//     inputs and outputs is therefore allowed;
//   * f(const T (&in)[NIN], T (&out)[NOUT]) is called once per element, in ascending index order within a lane;
//   * NT selects the non-temporal policy for vectors that do not fit the caches anyway;
//   * pointers that are not 16-byte aligned (a slice of a larger vector) and the last n % (16 / sizeof(T)) elements take the
//     one-element-per-lane path.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct Pack16;
template <>
struct Pack16<float> {
	typedef float V __attribute__((ext_vector_type(4)));
	static constexpr int N = 4;
};
template <>
struct Pack16<double> {
	typedef double V __attribute__((ext_vector_type(2)));
	static constexpr int N = 2;
};

constexpr int STREAM_TPB = 256;
constexpr int STREAM_U = 4;

template <typename T, bool NT, int NIN, int NOUT, typename F>
__device__ __forceinline__ void streamMap(long long n, const T* const* in, T* const* out, F&& f) {  // out may be null when NOUT == 0
	using V = typename Pack16<T>::V;
	constexpr int N = Pack16<T>::N;
	unsigned long long bits = 0;
#pragma unroll
	for (int k = 0; k < NIN; ++k) bits |= reinterpret_cast<unsigned long long>(in[k]);
#pragma unroll
	for (int k = 0; k < NOUT; ++k) bits |= reinterpret_cast<unsigned long long>(out[k]);
	long long done = 0;
	if ((bits & 15ull) == 0) {
		const long long nvec = n / N;
		const long long chunk = static_cast<long long>(STREAM_U) * STREAM_TPB;
		for (long long base = static_cast<long long>(blockIdx.x) * chunk; base < nvec; base += static_cast<long long>(gridDim.x) * chunk) {
			V a[NIN][STREAM_U];
#pragma unroll
			for (int u = 0; u < STREAM_U; ++u) {
				const long long i = base + u * STREAM_TPB + threadIdx.x;
				if (i < nvec) {
#pragma unroll
					for (int k = 0; k < NIN; ++k) {
						const V* p = reinterpret_cast<const V*>(in[k]) + i;
						a[k][u] = NT ? __builtin_nontemporal_load(p) : *p;
					}
				}
			}
#pragma unroll
			for (int u = 0; u < STREAM_U; ++u) {
				const long long i = base + u * STREAM_TPB + threadIdx.x;
				if (i < nvec) {
					V o[NOUT > 0 ? NOUT : 1];
#pragma unroll
					for (int e = 0; e < N; ++e) {
						T iv[NIN];
						T ov[NOUT > 0 ? NOUT : 1];
#pragma unroll
						for (int k = 0; k < NIN; ++k) iv[k] = a[k][u][e];
						f(iv, ov);
#pragma unroll
						for (int k = 0; k < NOUT; ++k) o[k][e] = ov[k];
					}
#pragma unroll
					for (int k = 0; k < NOUT; ++k) {
						V* p = reinterpret_cast<V*>(out[k]) + i;
						if (NT) __builtin_nontemporal_store(o[k], p);
						else *p = o[k];
					}
				}
			}
		}
		done = nvec * N;
	}
	for (long long i = done + static_cast<long long>(blockIdx.x) * STREAM_TPB + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * STREAM_TPB) {
		T iv[NIN];
		T ov[NOUT > 0 ? NOUT : 1];
#pragma unroll
		for (int k = 0; k < NIN; ++k) iv[k] = in[k][i];
		f(iv, ov);
#pragma unroll
		for (int k = 0; k < NOUT; ++k) out[k][i] = ov[k];
	}
}

}  // namespace smm
