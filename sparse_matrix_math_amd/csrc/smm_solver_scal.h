// smm_solver_scal.h -- the recurrence scalars of the device-resident Krylov loops and the fixed-order sums of their partials: shared by the
// update kernels (smm_solvers.hip) and the SpMV kernel that forms CG's next direction in its own load phase (smm_spmv_march.hip).
#pragma once
#include "smm_device.h"
#include "smm_internal.h"

namespace smm {

constexpr int SCAL_TPB = 256;
// ConjugateGradient with the deferred x update (smm_solvers.hip cgLazyXP, smm_dist.hip distCgLazyP): directions kept before x is brought up to date
constexpr int LAZY_M = 8;

template <typename T>
struct Scal {
	T rr;       // CG: residualNormSquared / rz for PCG ; BiCGStab: rr0
	T denom;    // p.Ap / ap.r0
	T alpha;
	T beta;
	T omega;
	T res;      // CG: last ||r||^2 ; BiCGStab: last ||r||
	T rrPing[2];  // fused loops: rr (CG) / rr0 (BiCGStab) double-buffered by iteration parity
	T alphaRing[LAZY_M];  // CG with the deferred x update: alpha of the last LAZY_M iterations
	int done;
	int iters;
	int status;
	int pad;
	int flushIter;  // CG, deferred x update: the iteration whose SpMV launch found its predecessor converged (-1: none): that iteration's flush launch completes x
	int pad2;
};

template <typename T>
__device__ __forceinline__ T sumParts(const T* __restrict__ partials, T* red) {
	T acc = T(0);
	for (int i = threadIdx.x; i < NPART; i += SCAL_TPB) acc += partials[i];
	return blockSum256(acc, red);
}

template <typename T>
__device__ __forceinline__ T sumPartsAll(const T* __restrict__ partials, T* red5) {
	const T s = sumParts(partials, red5);
	if (threadIdx.x == 0) red5[4] = s;
	__syncthreads();
	const T v = red5[4];
	__syncthreads();
	return v;
}


}  // namespace smm
