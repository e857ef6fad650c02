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

}  // namespace smm
