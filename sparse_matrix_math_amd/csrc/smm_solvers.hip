// smm_solvers.hip -- device-resident Krylov loops: ConjugateGradient (ref:2316-2398, IC0 overload ref:2414-2505),
// BiCGStab (ref:2191-2303), BiCGSymmetric (ref:2021-2102).
//
// Design (MI355X-first, not a translation of the CPU loop):
//   * every vector and every scalar (alpha, beta, omega, ||r||^2, the convergence flag, the iteration count)
//     lives in HBM; the host only enqueues kernels.  No per-iteration host round trip: the host peeks at the
//     `done` flag through a pinned mailbox every few iterations, and every kernel of the loop starts by testing
//     the flag, so iterations enqueued past convergence are no-ops and x / iterations / status are exactly
//     those of the iteration at which the reference would have returned.
//   * the dot products that follow an SpMV are fused into the SpMV epilogue (p.Ap; ap.r0; as.as + as.s), the
//     ||r||^2 and r.r0 sums are fused into the x/r update, exactly the fusions the reference's serial loops
//     do (ref:2371-2375, 2263-2267) plus the SpMV ones.  Partial sums are combined in a fixed order.
//   * update loops keep the reference's expression shapes (_smm_fma nesting), so given the same scalars they
//     are bit-identical to the CPU loops.
#include <algorithm>
#include <atomic>
#include <cmath>

#include "smm_device.h"
#include "smm_internal.h"
#include "smm_solver_scal.h"

namespace smm {

constexpr int TPB = 256;

// ---- scalar stages (one workgroup each) ----------------------------------------------------------------
// CG start: rr = r.r ; if eps^2 > rr -> done, SUCCESS, x untouched (ref:2341-2344)
template <typename T>
__global__ __launch_bounds__(TPB) void cgInitScal(const T* __restrict__ partials, Scal<T>* sc, T eps, int pcg, const T* __restrict__ partials2) {
	__shared__ T red[4];
	const T rr = sumParts(partials, red);
	T rz = T(0);
	if (pcg) rz = sumParts(partials2, red);
	if (threadIdx.x == 0) {
		sc->rr = pcg ? rz : rr;
		sc->rrPing[0] = sc->rr;
		sc->res = rr;
		sc->iters = 0;
		sc->status = SMM_SOLVER_MAX_ITERATIONS_REACHED;
		sc->done = 0;
		sc->pad = 0;
		sc->flushIter = -1;
		if (eps * eps > rr) {
			sc->done = 1;
			sc->status = SMM_SOLVER_SUCCESS;
		}
	}
}

// alpha = rr / (Ap.p)  (ref:2354-2358)
template <typename T>
__global__ __launch_bounds__(TPB) void cgAlphaScal(const T* __restrict__ partials, Scal<T>* sc) {
	__shared__ T red[4];
	if (sc->done) return;
	const T pAp = sumParts(partials, red);
	if (threadIdx.x == 0) {
		sc->denom = pAp;
		sc->alpha = sc->rr / pAp;
	}
}

// after the x/r update: convergence test, beta (ref:2377-2382); for PCG partials hold r.r and partials2 r.z (ref:2483-2488)
template <typename T>
__global__ __launch_bounds__(TPB) void cgBetaScal(const T* __restrict__ partials, const T* __restrict__ partials2, Scal<T>* sc, T eps, int pcg) {
	__shared__ T red[4];
	if (sc->done) return;
	const T rrNew = sumParts(partials, red);
	T rzNew = T(0);
	if (pcg) rzNew = sumParts(partials2, red);
	if (threadIdx.x == 0) {
		sc->iters += 1;
		sc->res = rrNew;
		if (eps * eps > rrNew) {
			sc->done = 1;
			sc->status = SMM_SOLVER_SUCCESS;
		} else if (pcg) {
			sc->beta = rzNew / sc->rr;
			sc->rr = rzNew;
		} else {
			sc->beta = rrNew / sc->rr;
			sc->rr = rrNew;
		}
	}
}

// ---- vector stages -------------------------------------------------------------------------------------
// x = alpha p + xcur ; r = -alpha Ap + r ; partial ||r||^2   (ref:2371-2375)
template <typename T>
__global__ __launch_bounds__(TPB) void cgUpdateXR(int n, const Scal<T>* __restrict__ sc, const T* p, const T* Ap, const T* xcur, T* x, T* r,
                                                  T* __restrict__ partials, int wantNorm) {
	__shared__ T red[4];
	if (sc->done) return;
	const T alpha = sc->alpha;
	T acc = T(0);
	const T* const in[4] = {p, xcur, Ap, r};
	T* const out[2] = {x, r};
	streamMap<T, false, 4, 2>(n, in, out, [&](const T(&v)[4], T(&o)[2]) {
		o[0] = smmFma(alpha, v[0], v[1]);
		const T ri = smmFma(-alpha, v[2], v[3]);
		o[1] = ri;
		acc += ri * ri;
	});
	if (wantNorm) {
		const T s = blockSum256(acc, red);
		if (threadIdx.x == 0) partials[blockIdx.x] = s;
	}
}

// p = beta p + z   (ref:2391-2393 with z = r; ref:2497-2499 with z = M^-1 r)
template <typename T>
__global__ __launch_bounds__(TPB) void cgUpdateP(int n, const Scal<T>* __restrict__ sc, T* p, const T* z) {
	if (sc->done) return;
	const T beta = sc->beta;
	const T* const in[2] = {p, z};
	T* const out[1] = {p};
	streamMap<T, false, 2, 1>(n, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(beta, v[0], v[1]); });
}

// ---- fused forms: the scalar stage is folded into the vector kernel that consumes it --------------------------------------
// Every workgroup re-adds the NPART partial sums itself (same order => same bits in every workgroup), so alpha / beta / omega
// need no kernel of their own: two launches per CG iteration and three per BiCGStab iteration disappear.  State that must
// survive a kernel (rr, alpha, omega, done, iterations) is written by workgroup 0 only; rr is double-buffered by iteration
// parity because workgroup 0 writes the next value while the others still read the current one.
// The two update kernels of a CG iteration (ref:2354-2394), split so that every vector is passed as few times as the data flow
// allows: the first needs only Ap and r (r = -alpha Ap + r and ||r||^2 -- the next reduction point), the second reads p once for
// BOTH of its uses (x = alpha p + xcur, the reference's :2362-2366, and p = beta p + r): 8 vector passes per iteration instead of
// the 9 of "x and r, then p" (r03; config 4's iteration is two thirds vector updates since its SpMV reads no values[]).  Same
// expressions on the same operands: same bits.
// alpha = rr / (Ap.p) ; r = -alpha Ap + r ; partial ||r||^2.  sc->pad = the done flag as this iteration found it: the second kernel
// decides by it (its own workgroup 0 raises `done` while other workgroups of the same launch may not have started yet).
// (the vector loops of the fused kernels go through streamMap, smm_device.h: 16-byte accesses, 4 packs per lane in flight)
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void cgFusedR(int n, Scal<T>* sc, int par, const T* __restrict__ partsA, const T* Ap, T* r,
                                                T* __restrict__ partsC, int alphaSlot) {
	__shared__ T red[5];
	const int done = sc->done;
	if (blockIdx.x == 0 && threadIdx.x == 0) sc->pad = done;
	if (done) return;
	const T alpha = sc->rrPing[par] / sumPartsAll(partsA, red);
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
	const T s = blockSum256(acc, red);
	if (threadIdx.x == 0) partsC[blockIdx.x] = s;
}

// x = alpha p + xcur ; convergence test ; beta ; p = beta p + r   (ref:2362-2394: x is updated before the test, p only when the loop goes on)
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void cgFusedXP(int n, Scal<T>* sc, int par, const T* __restrict__ partsC, T eps, T* p, const T* r, const T* xcur,
                                                 T* x) {
	__shared__ T red[5];
	if (sc->pad) return;
	const T rrNew = sumPartsAll(partsC, red);
	const T rrOld = sc->rrPing[par];
	const T alpha = sc->alpha;
	const bool converged = eps * eps > rrNew;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		sc->iters += 1;
		sc->res = rrNew;
		if (converged) {
			sc->done = 1;
			sc->status = SMM_SOLVER_SUCCESS;
		} else {
			sc->rrPing[par ^ 1] = rrNew;
		}
	}
	if (converged) {
		const T* const in[2] = {p, xcur};
		T* const out[1] = {x};
		streamMap<T, NT, 2, 1>(n, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(alpha, v[0], v[1]); });
		return;
	}
	const T beta = rrNew / rrOld;
	const T* const in[3] = {p, xcur, r};
	T* const out[2] = {x, p};
	streamMap<T, NT, 3, 2>(n, in, out, [&](const T(&v)[3], T(&o)[2]) {
		o[0] = smmFma(alpha, v[0], v[1]);
		o[1] = smmFma(beta, v[0], v[2]);
	});
}

// ---- the x update deferred (r05) ------------------------------------------------------------------------------------------------
// For vectors that do not fit the caches an iteration of CG is 10 vector passes, 5 of them in cgFusedXP (read p, x, r; write x, p).  x is
// only an accumulator: x_{k+1} = alpha_k p_k + x_k (ref:2362-2366).  Keeping the last LAZY_M directions in a ring, x is brought up to date
// every LAZY_M-th iteration -- x = alpha_k p_k + (... + (alpha_{k-M+1} p_{k-M+1} + x)): the reference's roundings in the reference's order,
// bit for bit -- in the launch that forms the next p anyway: (3 (M - 1) + M + 4) / M = 4 + 1 / M passes per iteration instead of 5 (M = 8:
// 4.125; with p formed inside the SpMV the flush is a launch of its own, (M + 2) / M = 1.25), at the price of M more vectors of device memory.  The launch that finds the iteration converged -- or is told it is the last -- flushes
// whatever is pending, so x is complete whenever the loop ends.
template <typename T>
struct LazyRing {
	T* p[LAZY_M + 1];
};

// PENDING directions (this iteration's included) are applied to x when FLUSH; p_next = beta p_cur + r unless the iteration converged
template <typename T, bool NT, int PENDING, bool PUPD>
__device__ __forceinline__ void cgLazyFlush(int n, const LazyRing<T>& ring, int cur, const T (&alpha)[LAZY_M], T beta, const T* r, const T* xcur, T* x) {
	// inputs: x, p_cur-PENDING+1 .. p_cur (oldest first) [, r]; outputs: x [, p_next]
	const T* in[PENDING + 2];
	in[0] = xcur;
#pragma unroll
	for (int k = 0; k < PENDING; ++k) in[1 + k] = ring.p[(cur + (LAZY_M + 1) - (PENDING - 1 - k)) % (LAZY_M + 1)];
	in[PENDING + 1] = PUPD ? r : xcur;  // (without the p update r is not read: any valid vector)
	T* out[2] = {x, ring.p[(cur + 1) % (LAZY_M + 1)]};
	streamMap<T, NT, PENDING + 2, PUPD ? 2 : 1>(n, in, out, [&](const T(&v)[PENDING + 2], T(&o)[PUPD ? 2 : 1]) {
		T xv = v[0];
#pragma unroll
		for (int k = 0; k < PENDING; ++k) xv = smmFma(alpha[LAZY_M - PENDING + k], v[1 + k], xv);  // oldest direction first
		o[0] = xv;
		if (PUPD) o[1] = smmFma(beta, v[PENDING], v[PENDING + 1]);
	});
}

// convergence test ; beta ; p_next = beta p_cur + r ; x brought up to date when `flush` (host: every LAZY_M-th iteration and the last one)
// or when the iteration converged.  pending = directions not yet applied to x, this iteration's included (1 .. LAZY_M).
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void cgLazyXP(int n, Scal<T>* sc, int par, const T* __restrict__ partsC, T eps, LazyRing<T> ring, int cur, int pending, int flush,
                                                int alphaSlot, const T* r, const T* xcur, T* x) {
	__shared__ T red[5];
	if (sc->pad) return;
	const T rrNew = sumPartsAll(partsC, red);
	const T rrOld = sc->rrPing[par];
	const bool converged = eps * eps > rrNew;
	T alpha[LAZY_M];  // alpha[LAZY_M - 1] = this iteration's, alpha[LAZY_M - 2] the one before, ...
#pragma unroll
	for (int k = 0; k < LAZY_M; ++k) alpha[LAZY_M - 1 - k] = sc->alphaRing[(alphaSlot + LAZY_M - k) % LAZY_M];
	if (blockIdx.x == 0 && threadIdx.x == 0) {
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
		streamMap<T, NT, 2, 1>(n, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(beta, v[0], v[1]); });
		return;
	}
#define SMM_LAZY_CASE(P)                                                                         \
	case P:                                                                                      \
		if (converged) cgLazyFlush<T, NT, P, false>(n, ring, cur, alpha, beta, r, xcur, x);      \
		else cgLazyFlush<T, NT, P, true>(n, ring, cur, alpha, beta, r, xcur, x);                 \
		break;
	switch (pending) {
		SMM_LAZY_CASE(1)
		SMM_LAZY_CASE(2)
		SMM_LAZY_CASE(3)
		SMM_LAZY_CASE(4)
		SMM_LAZY_CASE(5)
		SMM_LAZY_CASE(6)
		SMM_LAZY_CASE(7)
		SMM_LAZY_CASE(8)
	default: break;
	}
#undef SMM_LAZY_CASE
}

// With the next direction formed inside the SpMV (launchConstMarchFusedP) nothing is left of cgLazyXP but x: this launch sits behind
// every such SpMV and completes x when it is scheduled (every LAZY_M-th iteration) or when that SpMV found its predecessor converged
// (flushIter == iter: one shot -- the launches enqueued behind a finished solve find another number there).  pending directions end at
// ring slot `cur`; alphaSlot: the ring position of the newest one's alpha.
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void cgLazyFlushOnly(int n, const Scal<T>* __restrict__ sc, LazyRing<T> ring, int cur, int pending, int scheduled, int iter,
                                                       int alphaSlot, const T* xcur, T* x) {
	const bool converged = sc->flushIter == iter;
	if (!converged && (!scheduled || sc->done)) return;
	T alpha[LAZY_M];
#pragma unroll
	for (int k = 0; k < LAZY_M; ++k) alpha[LAZY_M - 1 - k] = sc->alphaRing[(alphaSlot + LAZY_M - k) % LAZY_M];
	switch (pending) {
	case 1: cgLazyFlush<T, NT, 1, false>(n, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 2: cgLazyFlush<T, NT, 2, false>(n, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 3: cgLazyFlush<T, NT, 3, false>(n, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 4: cgLazyFlush<T, NT, 4, false>(n, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 5: cgLazyFlush<T, NT, 5, false>(n, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 6: cgLazyFlush<T, NT, 6, false>(n, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 7: cgLazyFlush<T, NT, 7, false>(n, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	case 8: cgLazyFlush<T, NT, 8, false>(n, ring, cur, alpha, T(0), nullptr, xcur, x); break;
	default: break;
	}
}

// alpha = rr0 / (ap.r0) ; s = -alpha ap + r   (ref:2243-2247)
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void bicgFusedS(int n, Scal<T>* sc, int par, const T* __restrict__ partsA, const T* ap, const T* r, T* sv) {
	__shared__ T red[5];
	if (sc->done) return;
	const T alpha = sc->rrPing[par] / sumPartsAll(partsA, red);
	if (blockIdx.x == 0 && threadIdx.x == 0) sc->alpha = alpha;
	const T* const in[2] = {ap, r};
	T* const out[1] = {sv};
	streamMap<T, NT, 2, 1>(n, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(-alpha, v[0], v[1]); });
}

// omega = (as.s)/(as.as) ; x, r update ; partial ||r||^2 and r.r0   (ref:2259-2269); partsB = [as.as | as.s], partsC = [r.r | r.r0]
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void bicgFusedXR(int n, Scal<T>* sc, const T* __restrict__ partsB, const T* p, const T* sv, const T* as,
                                                   const T* r0, T* x, T* r, T* __restrict__ partsC) {
	__shared__ T red[5];
	if (sc->done) return;
	const T asas = sumPartsAll(partsB, red);
	const T ass = sumPartsAll(partsB + NPART, red);
	const T omega = ass / asas;
	const T alpha = sc->alpha;
	if (blockIdx.x == 0 && threadIdx.x == 0) sc->omega = omega;
	T acc0 = T(0), acc1 = T(0);
	const T* const in[5] = {sv, x, p, as, r0};
	T* const out[2] = {x, r};
	streamMap<T, NT, 5, 2>(n, in, out, [&](const T(&v)[5], T(&o)[2]) {
		const T si = v[0];
		o[0] = smmFma(alpha, v[2], smmFma(omega, si, v[1]));
		const T ri = smmFma(-omega, v[3], si);
		o[1] = ri;
		acc0 += ri * ri;
		acc1 += ri * v[4];
	});
	const T s0 = blockSum256(acc0, red);
	const T s1 = blockSum256(acc1, red);
	if (threadIdx.x == 0) {
		partsC[blockIdx.x] = s0;
		partsC[NPART + blockIdx.x] = s1;
	}
}

// resL2Norm, loop test, beta, p = beta (-omega ap + p) + r   (ref:2268-2277)
template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void bicgFusedP(int n, Scal<T>* sc, int par, const T* __restrict__ partsC, T eps, const T* ap, const T* r, T* p) {
	__shared__ T red[5];
	if (sc->done) return;
	const T rr = sumPartsAll(partsC, red);
	const T newRR0 = sumPartsAll(partsC + NPART, red);
	const T res = sizeof(T) == 4 ? static_cast<T>(__fsqrt_rn(static_cast<float>(rr))) : static_cast<T>(__dsqrt_rn(static_cast<double>(rr)));
	const T alpha = sc->alpha;
	const T omega = sc->omega;
	const T rr0 = sc->rrPing[par];
	const bool leave = !(res > eps);  // while (resL2Norm > eps ...): NaN leaves the loop too
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		sc->res = res;
		sc->rrPing[par ^ 1] = newRR0;
		sc->iters += 1;
		if (leave) sc->done = 1;
	}
	if (leave) return;
	const T beta = (newRR0 * alpha) / (rr0 * omega);  // ref:2271
	const T* const in[3] = {ap, p, r};
	T* const out[1] = {p};
	streamMap<T, NT, 3, 1>(n, in, out, [&](const T(&v)[3], T(&o)[1]) { o[0] = smmFma(beta, smmFma(-omega, v[0], v[1]), v[2]); });
}

// two dot products that share an operand: partials = a.a, partials2 = a.b
template <typename T>
__global__ __launch_bounds__(TPB) void dot2Partials(int n, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ partials,
                                                    T* __restrict__ partials2, const int* __restrict__ doneFlag) {
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	T acc0 = T(0), acc1 = T(0);
	const T* const in[2] = {a, b};
	streamMap<T, false, 2, 0>(n, in, nullptr, [&](const T(&v)[2], T(&)[1]) {
		acc0 += v[0] * v[0];
		acc1 += v[0] * v[1];
	});
	const T s0 = blockSum256(acc0, red);
	const T s1 = blockSum256(acc1, red);
	if (threadIdx.x == 0) {
		partials[blockIdx.x] = s0;
		partials2[blockIdx.x] = s1;
	}
}

// ---- BiCGStab stages -----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(TPB) void bicgInitScal(const T* __restrict__ partials, Scal<T>* sc) {
	__shared__ T red[4];
	const T rr0 = sumParts(partials, red);  // ref:2231
	if (threadIdx.x == 0) {
		sc->rr = rr0;
		sc->rrPing[0] = rr0;
		sc->res = T(0);
		sc->iters = 0;
		sc->done = 0;
		sc->status = SMM_SOLVER_SUCCESS;
	}
}

// ---- BiCGSymmetric stages (ref:2021-2102) ----------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(TPB) void bsymInitScal(const T* __restrict__ partials, Scal<T>* sc) {
	__shared__ T red[4];
	const T rr = sumParts(partials, red);  // ref:2043
	if (threadIdx.x == 0) {
		sc->rr = rr;
		sc->iters = 0;
		sc->done = 0;
		sc->status = SMM_SOLVER_SUCCESS;
	}
}

template <typename T>
__global__ __launch_bounds__(TPB) void bsymAlphaScal(const T* __restrict__ partials, Scal<T>* sc, T eps) {
	__shared__ T red[4];
	if (sc->done) return;
	const T denom = sumParts(partials, red);
	if (threadIdx.x == 0) {
		if (eps > (denom < T(0) ? -denom : denom) && sc->rr > T(1)) {  // ref:2056-2058
			sc->done = 1;
			sc->status = SMM_SOLVER_DIVERGED;
		} else {
			sc->alpha = sc->rr / denom;
		}
	}
}

// x += alpha p ; r -= alpha ap  (ref:2068-2071: plain += / -= forms, not _smm_fma)
template <typename T>
__global__ __launch_bounds__(TPB) void bsymUpdateXR(int n, const Scal<T>* __restrict__ sc, const T* p, const T* ap, T* x, T* r,
                                                    T* __restrict__ partials) {
	__shared__ T red[4];
	if (sc->done) return;
	const T alpha = sc->alpha;
	T acc = T(0);
	const T* const in[4] = {p, x, ap, r};
	T* const out[2] = {x, r};
	streamMap<T, false, 4, 2>(n, in, out, [&](const T(&v)[4], T(&o)[2]) {
		const T pa = alpha * v[0];
		o[0] = v[1] + pa;
		const T qa = alpha * v[2];
		const T ri = v[3] - qa;
		o[1] = ri;
		acc += ri * ri;
	});
	const T s = blockSum256(acc, red);
	if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

template <typename T>
__global__ __launch_bounds__(TPB) void bsymBetaScal(const T* __restrict__ partials, Scal<T>* sc, T eps) {
	__shared__ T red[4];
	if (sc->done) return;
	const T newRR = sumParts(partials, red);  // ref:2075
	if (threadIdx.x == 0) {
		if (newRR > T(1) && sc->rr < eps) {  // ref:2079-2081
			sc->done = 1;
			sc->status = SMM_SOLVER_DIVERGED;
		} else {
			sc->beta = newRR / sc->rr;
			sc->rr = newRR;
			sc->iters += 1;
			if (!(newRR > eps * eps)) sc->done = 1;  // ref:2096
		}
	}
}

// p = r + beta p  (ref:2090-2092)
template <typename T>
__global__ __launch_bounds__(TPB) void bsymUpdateP(int n, const Scal<T>* __restrict__ sc, const T* r, T* p) {
	if (sc->done) return;
	const T beta = sc->beta;
	const T* const in[2] = {p, r};
	T* const out[1] = {p};
	streamMap<T, false, 2, 1>(n, in, out, [&](const T(&v)[2], T(&o)[1]) {
		const T bp = beta * v[0];
		o[0] = v[1] + bp;
	});
}

// ---------------------------------------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------------------------------------
static int gridFor(long long n) { return static_cast<int>(std::max<long long>(1, std::min<long long>((n + TPB - 1) / TPB, NPART))); }

// Non-temporal loads / stores for an update kernel whose vectors cannot stay in the 256 MB Infinity Cache until the next kernel
// reads them anyway; cache-resident problems keep the default policy.  SMM_HIP_UPDATE_NT=0/1 overrides (measurements).
bool updateNT(long long n, size_t elemBytes, int vectors) {
	static const int forced = [] {
		const char* env = getenv("SMM_HIP_UPDATE_NT");
		return env ? atoi(env) : -1;
	}();
	if (forced >= 0) return forced != 0;
	return static_cast<double>(n) * static_cast<double>(elemBytes) * vectors > 192.0 * 1024 * 1024;
}

#define SMM_LAUNCH_UPDATE(KERNEL, NTFLAG, GRID, STREAM, ...)                    \
	do {                                                                       \
		if (NTFLAG) KERNEL<T, true><<<(GRID), TPB, 0, (STREAM)>>>(__VA_ARGS__); \
		else KERNEL<T, false><<<(GRID), TPB, 0, (STREAM)>>>(__VA_ARGS__);       \
	} while (0)

static int checkInterval(int it) { return std::max(4, std::min(64, it / 4)); }

// from how many bytes per vector CG defers its x update (cgLazyXP): where five vectors no longer fit the 256 MB Infinity Cache the passes
// are what an iteration costs; below, the extra ring of directions buys nothing.  smm_hip_set_cg_lazy_x_min_bytes (tests) / SMM_HIP_CG_LAZY_X=0
static std::atomic<long long> g_lazyMinBytes{-1};
long long cgLazyMinBytes() {
	const long long forced = g_lazyMinBytes.load(std::memory_order_relaxed);
	if (forced >= 0) return forced;
	static const long long env = [] {
		const char* e = getenv("SMM_HIP_CG_LAZY_X");
		return e && atoi(e) == 0 ? (1LL << 62) : (64LL << 20);
	}();
	return env;
}

template <typename T>
static int readScal(const Scal<T>* d_sc, Scal<T>* h, hipStream_t s) {
	SMM_HIP_TRY(hipMemcpyAsync(h, d_sc, sizeof(Scal<T>), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

template <typename T>
int cgDev(const smm_hip_csr* a, const T* b, const T* x0, T* x, int maxIterations, T eps, const smm_hip_precond* M, hipStream_t s, int* status,
          int* iterations, T* resnorm2) {
	if (!a || a->dtype != dtypeOf<T>()) {
		setError("cg: null matrix or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	if (a->rows != a->cols) {
		setError("cg: matrix must be square");
		return SMM_HIP_ERR_INVALID;
	}
	const int pcg = M != nullptr;
	if (pcg && (M->kind != SMM_PRECOND_IC0 || M->a != a)) {
		// the reference only has the IC0 overload (ref:2414-2422)
		setError("cg: preconditioner must be an IC0 preconditioner created for this matrix");
		return SMM_HIP_ERR_INVALID;
	}
	const int n = a->rows;
	if (n > 0 && (!b || !x0 || !x)) {
		setError("cg: null vector");
		return SMM_HIP_ERR_INVALID;
	}
	if (!pcg) {
		// a matrix that fits the chip's register file is solved in one launch (smm_resident.hip)
		bool handled = false;
		SMM_TRY(ensureCsrReady(a, s, true));
		SMM_TRY(cgResidentTry<T>(a, b, x0, x, maxIterations, eps, s, status, iterations, resnorm2, &handled));
		if (handled) return SMM_HIP_OK;
	}
	SMM_TRY(ensureCsrReady(a, s, true));
	SMM_TRY(adoptPatternForSolver(a, maxIterations, s));  // many SpMVs ahead: a mid-size banded / stencil matrix takes the index-free family
	DevBuf<T> r, p, Ap, z, parts, parts2;
	DevBuf<T> ringBuf[LAZY_M];
	DevBuf<Scal<T>> sc;
	// the deferred x update (cgLazyXP): unpreconditioned CG on vectors too large for the caches; LAZY_M more vectors for the ring of directions
	bool lazy = !pcg && static_cast<long long>(n) * static_cast<long long>(sizeof(T)) >= cgLazyMinBytes();
	LazyRing<T> ring{};
	SMM_TRY(r.alloc(n));
	SMM_TRY(p.alloc(n));
	if (lazy) {
		ring.p[0] = p;
		for (int k = 0; k < LAZY_M && lazy; ++k) {
			if (ringBuf[k].alloc(n) != SMM_HIP_OK) lazy = false;  // no room for the ring: the eager loop needs none
			ring.p[k + 1] = ringBuf[k];
		}
		if (!lazy) {
			for (int k = 0; k < LAZY_M; ++k) ringBuf[k].release();
			(void)hipGetLastError();
		}
	}
	const bool fuseP = lazy && constMarchFusable(a, sizeof(T));  // (decided once per solve: the two loop forms do their bookkeeping in different launches)
	SMM_TRY(Ap.alloc(n));
	if (pcg) SMM_TRY(z.alloc(n));
	SMM_TRY(parts.alloc(2 * NPART));
	SMM_TRY(parts2.alloc(NPART));
	SMM_TRY(sc.alloc(1));
	const int g = gridFor(n);

	SMM_TRY(launchSpmv<T>(a, SMM_OP_SUB, b, x0, r, 0, nullptr, nullptr, nullptr, s));  // r = b - A x0, ref:2337
	if (pcg) {
		SMM_TRY(precondApplyDev<T>(M, r, z, nullptr, s));           // z = M^-1 r, ref:2441
		SMM_TRY(launchCopy2<T>(n, z, p, nullptr, s));               // p = z, ref:2447
		SMM_TRY(launchDotPartials<T>(n, r, r, parts, nullptr, s));  // ||r||^2
		SMM_TRY(launchDotPartials<T>(n, r, z, parts2, nullptr, s)); // r.z
	} else {
		SMM_TRY(launchCopy2<T>(n, r, p, nullptr, s));               // p = r, ref:2340
		SMM_TRY(launchDotPartials<T>(n, r, r, parts, nullptr, s));  // ref:2341
	}
	cgInitScal<T><<<1, TPB, 0, s>>>(parts, sc, eps, pcg, parts2);
	if (maxIterations == -1) maxIterations = n;  // ref:2345-2347 (no clamp otherwise)

	static thread_local DonePoller poller;
	SMM_TRY(poller.init(s));
	int seenDone = 0;
	int nextCheck = 0;
	const int* doneFlag = &sc.p->done;
	for (int i = 0; i < maxIterations && !seenDone; ++i) {
		if (i == nextCheck) {
			seenDone = poller.post(doneFlag);
			if (seenDone < 0) return seenDone;
			if (seenDone) break;
			nextCheck = i + checkInterval(i);
		}
		if (lazy && fuseP) {
			// the direction is formed INSIDE the SpMV (2.5-D constant-diagonal kernel): SpMV' (bookkeeping of iteration i - 1, p_i, A p_i, p.Ap),
			// the flush of x behind it (scheduled every LAZY_M-th iteration; or because SpMV' found iteration i - 1 converged), the r update
			const int cur = i % (LAZY_M + 1);
			if (i == 0) {
				SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, ring.p[0], Ap, 1, ring.p[0], parts, doneFlag, s, SPMV_HALF_TILES));
			} else {
				const int prev = (i - 1) % (LAZY_M + 1);
				const CgFuseBook<T> bk{&sc.p->pad, sc.p->rrPing, &sc.p->res, &sc.p->iters, &sc.p->done, &sc.p->status, &sc.p->flushIter};
				const CgFuseArgs<T> f{r, ring.p[cur], bk, parts2, eps, (i - 1) & 1, i};
				if (!launchConstMarchFusedP<T>(a, ring.p[prev], Ap, parts, doneFlag, f, s)) {
					setError("cg: the fused SpMV could not be launched");
					return SMM_HIP_ERR_HIP;
				}
				SMM_LAUNCH_UPDATE(cgLazyFlushOnly, updateNT(n, sizeof(T), 5), g, s, n, sc, ring, prev, (i - 1) % LAZY_M + 1, i % LAZY_M == 0 ? 1 : 0, i, (i - 1) % LAZY_M,
				                  i <= LAZY_M ? x0 : x, x);
			}
			SMM_LAUNCH_UPDATE(cgFusedR, updateNT(n, sizeof(T), 3), NPART, s, n, sc, i & 1, parts, Ap, r, parts2, i % LAZY_M);
			if (i == maxIterations - 1) {  // the last planned iteration has no SpMV' behind it: its bookkeeping and the rest of x
				SMM_LAUNCH_UPDATE(cgLazyXP, updateNT(n, sizeof(T), 5), g, s, n, sc, i & 1, parts2, eps, ring, cur, i % LAZY_M + 1, 1, i % LAZY_M, r, i < LAZY_M ? x0 : x, x);
			}
			continue;
		}
		if (lazy) {
			// the direction of iteration i lives in ring slot i % (LAZY_M + 1); x is brought up to date every LAZY_M-th iteration, in the last
			// planned one, and by whichever launch finds the iteration converged
			const int cur = i % (LAZY_M + 1);
			const T* pc = ring.p[cur];
			SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, pc, Ap, 1, pc, parts, doneFlag, s, SPMV_HALF_TILES));
			SMM_LAUNCH_UPDATE(cgFusedR, updateNT(n, sizeof(T), 3), NPART, s, n, sc, i & 1, parts, Ap, r, parts2, i % LAZY_M);
			const int pending = i % LAZY_M + 1;
			const int flush = (pending == LAZY_M || i == maxIterations - 1) ? 1 : 0;
			const T* xc = i < LAZY_M ? x0 : x;  // (until the first scheduled flush x has not been written: ref:2351, 2395)
			SMM_LAUNCH_UPDATE(cgLazyXP, updateNT(n, sizeof(T), 5), g, s, n, sc, i & 1, parts2, eps, ring, cur, pending, flush, i % LAZY_M, r, xc, x);
			continue;
		}
		// Ap = A p with the p.Ap partial sums fused into the epilogue (ref:2353-2354)
		SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, p, Ap, 1, p, parts, doneFlag, s, pcg ? 0 : SPMV_HALF_TILES));
		const T* xcur = i == 0 ? x0 : x;  // ref:2351, 2395
		if (pcg) {
			cgAlphaScal<T><<<1, TPB, 0, s>>>(parts, sc);
			// NPART workgroups: every partial slot is (re)written each iteration, idle workgroups write 0
			cgUpdateXR<T><<<NPART, TPB, 0, s>>>(n, sc, p, Ap, xcur, x, r, parts, 0);
			SMM_TRY(precondApplyDev<T>(M, r, z, doneFlag, s));         // ref:2482
			dot2Partials<T><<<NPART, TPB, 0, s>>>(n, r, z, parts, parts2, doneFlag);  // r.r and r.z, ref:2483-2484
			cgBetaScal<T><<<1, TPB, 0, s>>>(parts, parts2, sc, eps, 1);
			cgUpdateP<T><<<g, TPB, 0, s>>>(n, sc, p, z);
		} else {
			// alpha and beta are formed inside the two update kernels (no scalar launches)
			SMM_LAUNCH_UPDATE(cgFusedR, updateNT(n, sizeof(T), 3), NPART, s, n, sc, i & 1, parts, Ap, r, parts2, 0);
			SMM_LAUNCH_UPDATE(cgFusedXP, updateNT(n, sizeof(T), 5), g, s, n, sc, i & 1, parts2, eps, p, r, xcur, x);
		}
	}
	SMM_HIP_TRY(hipGetLastError());
	Scal<T> h;
	SMM_TRY(readScal<T>(sc, &h, s));
	if (status) *status = h.status;
	if (iterations) *iterations = h.iters;
	if (resnorm2) *resnorm2 = h.res;
	return pcg ? precondTakeError(M, s) : SMM_HIP_OK;
}

// M^-1 of the preconditioned loop: the library's device-resident preconditioners, or a HOST functor of the caller (the reference's
// BiCGStab template takes any type with `int apply(const T* rhs, T* x) const`, ref:2191-2199, 2218, 2235, 2251)
template <typename T>
struct DevApplier {
	const smm_hip_precond* M;
	static constexpr bool hostSide = false;
	int operator()(const T* in, T* out, const int* doneFlag, hipStream_t s) const { return precondApplyDev<T>(M, in, out, doneFlag, s); }
};

// The vector goes to the host, the caller's functor runs there, the result comes back: two PCIe copies and two stream drains per
// apply.  Everything else of the iteration stays on the device.  A non-zero return of the functor ends the solve with
// SMM_HIP_ERR_PRECOND (the reference ignores apply()'s return value inside the loop; a failing apply leaves x undefined there).
template <typename T>
struct HostApplier {
	int (*fn)(void* user, const T* rhs, T* x);
	void* user;
	int n;
	T* hostIn;   // pinned
	T* hostOut;  // pinned
	static constexpr bool hostSide = true;
	int operator()(const T* in, T* out, const int* /*doneFlag*/, hipStream_t s) const {
		if (n == 0) return SMM_HIP_OK;
		SMM_HIP_TRY(hipMemcpyAsync(hostIn, in, sizeof(T) * n, hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
		if (fn(user, hostIn, hostOut) != 0) {
			setError("bicgstab: the caller's preconditioner apply() returned non-zero");
			return SMM_HIP_ERR_PRECOND;
		}
		SMM_HIP_TRY(hipMemcpyAsync(out, hostOut, sizeof(T) * n, hipMemcpyHostToDevice, s));
		return SMM_HIP_OK;
	}
};

template <typename T, typename Applier>
static int bicgstabLoop(const smm_hip_csr* a, const T* b, T* x, int maxIterations, T eps, const bool precondition, const Applier& apply, hipStream_t s,
                        int* status, int* iterations, T* resnorm, const T* jacobiDiag = nullptr, const smm_hip_precond* blockM = nullptr) {
	// jacobiDiag: M is the library's Jacobi preconditioner and its apply (x = rhs / diag, smm_precond.hip) is folded into the rows of the
	// SpMV that precedes it -- the same division on the same operands, so the same bits as SpMV + apply -- which lets the preconditioned
	// loop use the fused dots of the unpreconditioned one: no apply launches, no dot launches, no extra vector passes.
	// blockM: M is a block preconditioner (smm_precond_block.hip): its one-launch apply forms the dot products of its result in its
	// own epilogue, so the loop needs no dot launches either.
	const int n = a->rows;
	maxIterations = std::min(maxIterations, n);  // ref:2200
	if (maxIterations == -1) maxIterations = n;  // ref:2201-2203
	DevBuf<T> r, r0, p, ap, sv, as, scratch, parts, parts2;
	DevBuf<Scal<T>> sc;
	{
		SetupTrace trace("bicgstab loop:   allocate the temporaries");
		SMM_TRY(r.alloc(n));
		SMM_TRY(r0.alloc(n));
		SMM_TRY(p.alloc(n));
		SMM_TRY(ap.alloc(n));
		SMM_TRY(sv.alloc(n));
		SMM_TRY(as.alloc(n));
		if (precondition) SMM_TRY(scratch.alloc(n));
		SMM_TRY(parts.alloc(2 * NPART));   // [ap.r0] then [as.as | as.s]
		SMM_TRY(parts2.alloc(2 * NPART));  // [r.r | r.r0]
		SMM_TRY(sc.alloc(1));
	}
	SetupTrace traceLoop("bicgstab loop:   enqueue + run + read back");
	const int g = NPART;  // update kernels that write partials use the full partial grid
	// block preconditioner of THIS matrix: A p and A s are formed inside the apply's launch, row by row in the order of the stored entries
	const bool fuseBlk = blockM && blockM->a == a && blockFuseSpmv(blockM, false);

	if (precondition) {
		SMM_TRY(launchSpmv<T>(a, SMM_OP_SUB, b, x, scratch, 0, nullptr, nullptr, nullptr, s));  // ref:2215
		SMM_TRY(apply(scratch, r, nullptr, s));                                                  // ref:2217-2224
	} else {
		SMM_TRY(launchSpmv<T>(a, SMM_OP_SUB, b, x, r, 0, nullptr, nullptr, nullptr, s));
	}
	SMM_TRY(launchCopy2<T>(n, r, r0, p, s));                     // ref:2225-2226
	SMM_TRY(launchDotPartials<T>(n, r, r0, parts, nullptr, s));  // ref:2231
	bicgInitScal<T><<<1, TPB, 0, s>>>(parts, sc);

	static thread_local DonePoller poller;
	SMM_TRY(poller.init(s));
	const int* doneFlag = &sc.p->done;
	const int planned = std::max(1, maxIterations);  // do { } while: the body always runs once (ref:2232, 2277)
	int nextCheck = 1;
	for (int i = 0; i < planned; ++i) {
		if (Applier::hostSide && i > 0) {
			// a host functor drains the stream at every apply anyway: test the flag directly so that it is never called for an
			// iteration the loop has already left
			int done = 0;
			SMM_HIP_TRY(hipMemcpyAsync(&done, doneFlag, sizeof(int), hipMemcpyDeviceToHost, s));
			SMM_HIP_TRY(hipStreamSynchronize(s));
			if (done) break;
		} else if (i == nextCheck) {
			const int seen = poller.post(doneFlag);
			if (seen < 0) return seen;
			if (seen) break;
			nextCheck = i + checkInterval(i);
		}
		if (jacobiDiag) {
			SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, jacobiDiag, p, ap, 1, r0, parts, doneFlag, s, SPMV_DIV_LHS));  // ref:2234-2235 + 2243 fused
		} else if (blockM) {
			if (fuseBlk) {
				SMM_TRY(blockApplySpmvDev<T>(blockM, p, ap, 1, r0, parts, doneFlag, s));  // ref:2234 + 2235 + 2243 in one launch
			} else {
				SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, p, scratch, 0, nullptr, nullptr, doneFlag, s));  // ref:2234
				SMM_TRY(blockApplyDev<T>(blockM, scratch, ap, 1, r0, parts, doneFlag, s));                         // ref:2235 + 2243 fused
			}
		} else if (precondition) {
			SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, p, scratch, 0, nullptr, nullptr, doneFlag, s));  // ref:2234
			SMM_TRY(apply(scratch, ap, doneFlag, s));                                                          // ref:2235
			SMM_TRY(launchDotPartials<T>(n, ap, r0, parts, doneFlag, s));                                      // ref:2243
		} else {
			SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, p, ap, 1, r0, parts, doneFlag, s));  // ref:2240 + 2243 fused
		}
		// alpha, omega and beta are formed inside the three update kernels that consume them (no scalar launches)
		SMM_LAUNCH_UPDATE(bicgFusedS, updateNT(n, sizeof(T), 3), gridFor(n), s, n, sc, i & 1, parts, ap, r, sv);
		if (jacobiDiag) {
			SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, jacobiDiag, sv, as, 2, sv, parts, doneFlag, s, SPMV_DIV_LHS));  // ref:2250-2251 + 2256-2261 fused
		} else if (blockM) {
			if (fuseBlk) {
				SMM_TRY(blockApplySpmvDev<T>(blockM, sv, as, 2, sv, parts, doneFlag, s));  // ref:2250 + 2251 + 2259, 2261 in one launch
			} else {
				SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, sv, scratch, 0, nullptr, nullptr, doneFlag, s));  // ref:2250
				SMM_TRY(blockApplyDev<T>(blockM, scratch, as, 2, sv, parts, doneFlag, s));                          // ref:2251 + 2259, 2261 fused
			}
		} else if (precondition) {
			SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, sv, scratch, 0, nullptr, nullptr, doneFlag, s));  // ref:2250
			SMM_TRY(apply(scratch, as, doneFlag, s));                                                           // ref:2251
			dot2Partials<T><<<NPART, TPB, 0, s>>>(n, as, sv, parts, parts.p + NPART, doneFlag);                 // ref:2259, 2261
		} else {
			// as = A s with as.as -> parts[0..NPART) and as.s -> parts[NPART..2 NPART) fused (ref:2256-2261)
			SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, sv, as, 2, sv, parts, doneFlag, s));
		}
		SMM_LAUNCH_UPDATE(bicgFusedXR, updateNT(n, sizeof(T), 7), g, s, n, sc, parts, p, sv, as, r0, x, r, parts2);
		SMM_LAUNCH_UPDATE(bicgFusedP, updateNT(n, sizeof(T), 4), gridFor(n), s, n, sc, i & 1, parts2, eps, ap, r, p);
	}
	SMM_HIP_TRY(hipGetLastError());
	Scal<T> h;
	SMM_TRY(readScal<T>(sc, &h, s));
	if (status) *status = h.iters > maxIterations ? SMM_SOLVER_MAX_ITERATIONS_REACHED : SMM_SOLVER_SUCCESS;  // ref:2279-2282
	if (iterations) *iterations = h.iters;
	if (resnorm) *resnorm = h.res;
	return SMM_HIP_OK;
}

template <typename T>
static int bicgstabCheck(const smm_hip_csr* a, const T* b, T* x) {
	if (!a || a->dtype != dtypeOf<T>()) {
		setError("bicgstab: null matrix or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	if (a->rows != a->cols) {
		setError("bicgstab: matrix must be square");
		return SMM_HIP_ERR_INVALID;
	}
	if (a->rows > 0 && (!b || !x)) {
		setError("bicgstab: null vector");
		return SMM_HIP_ERR_INVALID;
	}
	return SMM_HIP_OK;
}

template <typename T>
int bicgstabDev(const smm_hip_csr* a, const T* b, T* x, int maxIterations, T eps, const smm_hip_precond* M, hipStream_t s, int* status,
                int* iterations, T* resnorm) {
	SMM_TRY(bicgstabCheck<T>(a, b, x));
	const bool precondition = M != nullptr && M->kind != SMM_PRECOND_NONE;  // ref:2209
	if (precondition && (M->a != a || M->kind == SMM_PRECOND_IC0)) {
		setError("bicgstab: preconditioner must be JACOBI / ILU0 / SGS / BLOCK_ILU0 / BLOCK_SGS created for this matrix");
		return SMM_HIP_ERR_INVALID;
	}
	const DevApplier<T> apply{M};
	SMM_TRY(ensureCsrReady(a, s, true));
	SMM_TRY(adoptPatternForSolver(a, maxIterations, s));
	const T* jacobiDiag = precondition && M->kind == SMM_PRECOND_JACOBI ? static_cast<const T*>(M->d_values) : nullptr;
	const smm_hip_precond* blockM = precondition && isBlockKind(M->kind) ? M : nullptr;
	if (!precondition || jacobiDiag) {
		// a matrix in the row-mask encoding whose vectors fit the register file is solved in ONE launch (smm_resident_bicg.hip)
		bool handled = false;
		SMM_TRY(bicgstabResidentTry<T>(a, b, x, maxIterations, eps, jacobiDiag, s, status, iterations, resnorm, &handled));
		if (handled) return SMM_HIP_OK;
	}
	SMM_TRY((bicgstabLoop<T, DevApplier<T>>(a, b, x, maxIterations, eps, precondition, apply, s, status, iterations, resnorm, jacobiDiag, blockM)));
	return precondition ? precondTakeError(M, s) : SMM_HIP_OK;
}

// host vectors, host functor: the generic form of the reference's template (ref:2191-2199)
template <typename T>
static int bicgstabFunctorHost(const smm_hip_csr* a, T* b, T* x, int maxIterations, T eps, int (*fn)(void*, const T*, T*), void* user, int* status,
                               int* iterations, T* resnorm) {
	if (!fn) {
		setError("bicgstab_functor: null apply function");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	SMM_TRY(bicgstabCheck<T>(a, b, x));
	const int n = a->rows;
	hipStream_t s = libStream();
	DevBuf<T> db, dx;
	SMM_TRY(db.alloc(n));
	SMM_TRY(dx.alloc(n));
	T* pinned = nullptr;
	SMM_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pinned), sizeof(T) * 2 * static_cast<size_t>(std::max(1, n)), hipHostMallocDefault));
	struct Unpin {
		T* p;
		~Unpin() { hipHostFree(p); }
	} unpin{pinned};
	if (n) {
		SMM_TRY(hostToDev(db, b, sizeof(T) * n, s));
		SMM_TRY(hostToDev(dx, x, sizeof(T) * n, s));
	}
	const HostApplier<T> apply{fn, user, n, pinned, pinned + std::max(1, n)};
	const int rc = bicgstabLoop<T, HostApplier<T>>(a, db, dx, maxIterations, eps, true, apply, s, status, iterations, resnorm);
	if (rc != SMM_HIP_OK) {
		hipStreamSynchronize(s);  // kernels of the abandoned loop may still be queued on buffers that are about to be released
		return rc;
	}
	if (n) {
		SMM_TRY(devToHost(x, dx, sizeof(T) * n, s));
	}
	return SMM_HIP_OK;
}

template <typename T>
int bicgsymmetricDev(const smm_hip_csr* a, const T* b, T* x, int maxIterations, T eps, hipStream_t s, int* status, int* iterations) {
	if (!a || a->dtype != dtypeOf<T>()) {
		setError("bicgsymmetric: null matrix or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	if (a->rows != a->cols) {
		setError("bicgsymmetric: matrix must be square");
		return SMM_HIP_ERR_INVALID;
	}
	const int n = a->rows;
	maxIterations = std::min(maxIterations, n);  // ref:2030-2033
	if (maxIterations == -1) maxIterations = n;
	SMM_TRY(ensureCsrReady(a, s, true));
	SMM_TRY(adoptPatternForSolver(a, maxIterations, s));
	DevBuf<T> r, p, ap, parts;
	DevBuf<Scal<T>> sc;
	SMM_TRY(r.alloc(n));
	SMM_TRY(p.alloc(n));
	SMM_TRY(ap.alloc(n));
	SMM_TRY(parts.alloc(2 * NPART));
	SMM_TRY(sc.alloc(1));
	SMM_TRY(launchSpmv<T>(a, SMM_OP_SUB, b, x, r, 0, nullptr, nullptr, nullptr, s));  // ref:2036
	SMM_TRY(launchCopy2<T>(n, r, p, nullptr, s));
	SMM_TRY(launchDotPartials<T>(n, r, r, parts, nullptr, s));
	bsymInitScal<T><<<1, TPB, 0, s>>>(parts, sc);
	static thread_local DonePoller poller;
	SMM_TRY(poller.init(s));
	const int* doneFlag = &sc.p->done;
	const int planned = std::max(1, maxIterations);
	int nextCheck = 1;
	for (int i = 0; i < planned; ++i) {
		if (i == nextCheck) {
			const int seen = poller.post(doneFlag);
			if (seen < 0) return seen;
			if (seen) break;
			nextCheck = i + checkInterval(i);
		}
		SMM_TRY(launchSpmv<T>(a, SMM_OP_ASSIGN, nullptr, p, ap, 1, p, parts, doneFlag, s));  // ref:2048-2049
		bsymAlphaScal<T><<<1, TPB, 0, s>>>(parts, sc, eps);
		bsymUpdateXR<T><<<NPART, TPB, 0, s>>>(n, sc, p, ap, x, r, parts);
		bsymBetaScal<T><<<1, TPB, 0, s>>>(parts, sc, eps);
		bsymUpdateP<T><<<gridFor(n), TPB, 0, s>>>(n, sc, r, p);
	}
	SMM_HIP_TRY(hipGetLastError());
	Scal<T> h;
	SMM_TRY(readScal<T>(sc, &h, s));
	int st = h.status;
	if (st == SMM_SOLVER_SUCCESS && h.iters > maxIterations) st = SMM_SOLVER_MAX_ITERATIONS_REACHED;  // ref:2098-2100
	if (status) *status = st;
	if (iterations) *iterations = h.iters;
	return SMM_HIP_OK;
}

// ---- host-pointer wrappers: the reference's calling convention -----------------------------------------
template <typename T>
static int cgHost(const smm_hip_csr* a, const T* b, const T* x0, T* x, int maxIterations, T eps, const smm_hip_precond* M, int* status,
                  int* iterations, T* resnorm2) {
	if (!a) {
		setError("cg: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	const int n = a->rows;
	if (n > 0 && (!b || !x0 || !x)) {
		setError("cg: null vector");
		return SMM_HIP_ERR_INVALID;
	}
	hipStream_t s = libStream();
	DevBuf<T> db, dx0, dx;
	SMM_TRY(db.alloc(n));
	SMM_TRY(dx0.alloc(n));
	SMM_TRY(dx.alloc(n));
	if (n) {
		SMM_TRY(hostToDev(db, b, sizeof(T) * n, s));
		SMM_TRY(hostToDev(dx0, x0, sizeof(T) * n, s));
		// x is only written once the loop runs (ref:2342-2344): start the device copy from the caller's x
		SMM_TRY(hostToDev(dx, x, sizeof(T) * n, s));
	}
	int it = 0;
	SMM_TRY(cgDev<T>(a, db, dx0, dx, maxIterations, eps, M, s, status, &it, resnorm2));
	if (iterations) *iterations = it;
	if (n && it > 0) {
		SMM_TRY(devToHost(x, dx, sizeof(T) * n, s));
	}
	return SMM_HIP_OK;
}

template <typename T>
static int bicgstabHost(const smm_hip_csr* a, T* b, T* x, int maxIterations, T eps, const smm_hip_precond* M, int* status, int* iterations,
                        T* resnorm) {
	if (!a) {
		setError("bicgstab: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	const int n = a->rows;
	if (n > 0 && (!b || !x)) {
		setError("bicgstab: null vector");
		return SMM_HIP_ERR_INVALID;
	}
	hipStream_t s = libStream();
	SetupTrace traceAll("bicgstab (host pointers): whole call");
	DevBuf<T> db, dx;
	{
		SetupTrace trace("bicgstab (host pointers):   allocate b, x");
		SMM_TRY(db.alloc(n));
		SMM_TRY(dx.alloc(n));
	}
	if (n) {
		SetupTrace trace("bicgstab (host pointers):   copy b, x in");
		SMM_TRY(hostToDev(db, b, sizeof(T) * n, s));
		SMM_TRY(hostToDev(dx, x, sizeof(T) * n, s));
		if (SetupTrace::on()) SMM_HIP_TRY(hipStreamSynchronize(s));
	}
	{
		SetupTrace trace("bicgstab (host pointers):   device loop");
		SMM_TRY(bicgstabDev<T>(a, db, dx, maxIterations, eps, M, s, status, iterations, resnorm));
	}
	if (n) {
		SetupTrace trace("bicgstab (host pointers):   copy x out");
		SMM_TRY(devToHost(x, dx, sizeof(T) * n, s));
	}
	return SMM_HIP_OK;
}

template <typename T>
static int bicgsymmetricHost(const smm_hip_csr* a, T* b, T* x, int maxIterations, T eps, int* status, int* iterations) {
	if (!a) {
		setError("bicgsymmetric: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	const int n = a->rows;
	if (n > 0 && (!b || !x)) {
		setError("bicgsymmetric: null vector");
		return SMM_HIP_ERR_INVALID;
	}
	hipStream_t s = libStream();
	DevBuf<T> db, dx;
	SMM_TRY(db.alloc(n));
	SMM_TRY(dx.alloc(n));
	if (n) {
		SMM_TRY(hostToDev(db, b, sizeof(T) * n, s));
		SMM_TRY(hostToDev(dx, x, sizeof(T) * n, s));
	}
	SMM_TRY(bicgsymmetricDev<T>(a, db, dx, maxIterations, eps, s, status, iterations));
	if (n) {
		SMM_TRY(devToHost(x, dx, sizeof(T) * n, s));
	}
	return SMM_HIP_OK;
}

// every translation unit of the library is a code object of its own, built for the device at the FIRST launch of any of its kernels
// (5-9 ms each, measured: profiles/r04/first_spmv_setup_trace.txt); smm_hip_init touches one kernel of each hot-path unit so that
// the first SpMV of a process does not pay for it (SMM_HIP_PRELOAD=0: load lazily as before)
void preloadSolversUnit() {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(cgAlphaScal<float>));
	(void)hipGetLastError();
}

}  // namespace smm

using namespace smm;

extern "C" {

int smm_hip_cg_f32(const smm_hip_csr* a, const float* b, const float* x0, float* x, int maxIterations, float eps, const smm_hip_precond* M,
                   int* solver_status, int* iterations, float* resnorm2) {
	return cgHost<float>(a, b, x0, x, maxIterations, eps, M, solver_status, iterations, resnorm2);
}
int smm_hip_cg_f64(const smm_hip_csr* a, const double* b, const double* x0, double* x, int maxIterations, double eps, const smm_hip_precond* M,
                   int* solver_status, int* iterations, double* resnorm2) {
	return cgHost<double>(a, b, x0, x, maxIterations, eps, M, solver_status, iterations, resnorm2);
}
int smm_hip_cg_dev_f32(const smm_hip_csr* a, const float* d_b, const float* d_x0, float* d_x, int maxIterations, float eps,
                       const smm_hip_precond* M, smm_hip_stream stream, int* solver_status, int* iterations, float* resnorm2) {
	SMM_TRY(ensureInit());
	return cgDev<float>(a, d_b, d_x0, d_x, maxIterations, eps, M, pickStream(stream), solver_status, iterations, resnorm2);
}
int smm_hip_cg_dev_f64(const smm_hip_csr* a, const double* d_b, const double* d_x0, double* d_x, int maxIterations, double eps,
                       const smm_hip_precond* M, smm_hip_stream stream, int* solver_status, int* iterations, double* resnorm2) {
	SMM_TRY(ensureInit());
	return cgDev<double>(a, d_b, d_x0, d_x, maxIterations, eps, M, pickStream(stream), solver_status, iterations, resnorm2);
}

int smm_hip_bicgstab_f32(const smm_hip_csr* a, float* b, float* x, int maxIterations, float eps, const smm_hip_precond* M, int* solver_status,
                         int* iterations, float* resnorm) {
	return bicgstabHost<float>(a, b, x, maxIterations, eps, M, solver_status, iterations, resnorm);
}
int smm_hip_bicgstab_f64(const smm_hip_csr* a, double* b, double* x, int maxIterations, double eps, const smm_hip_precond* M, int* solver_status,
                         int* iterations, double* resnorm) {
	return bicgstabHost<double>(a, b, x, maxIterations, eps, M, solver_status, iterations, resnorm);
}
int smm_hip_bicgstab_dev_f32(const smm_hip_csr* a, const float* d_b, float* d_x, int maxIterations, float eps, const smm_hip_precond* M,
                             smm_hip_stream stream, int* solver_status, int* iterations, float* resnorm) {
	SMM_TRY(ensureInit());
	return bicgstabDev<float>(a, d_b, d_x, maxIterations, eps, M, pickStream(stream), solver_status, iterations, resnorm);
}
int smm_hip_bicgstab_dev_f64(const smm_hip_csr* a, const double* d_b, double* d_x, int maxIterations, double eps, const smm_hip_precond* M,
                             smm_hip_stream stream, int* solver_status, int* iterations, double* resnorm) {
	SMM_TRY(ensureInit());
	return bicgstabDev<double>(a, d_b, d_x, maxIterations, eps, M, pickStream(stream), solver_status, iterations, resnorm);
}

int smm_hip_bicgstab_functor_f32(const smm_hip_csr* a, float* b, float* x, int maxIterations, float eps, smm_hip_apply_fn_f32 apply, void* user,
                                 int* solver_status, int* iterations, float* resnorm) {
	return bicgstabFunctorHost<float>(a, b, x, maxIterations, eps, apply, user, solver_status, iterations, resnorm);
}
int smm_hip_bicgstab_functor_f64(const smm_hip_csr* a, double* b, double* x, int maxIterations, double eps, smm_hip_apply_fn_f64 apply, void* user,
                                 int* solver_status, int* iterations, double* resnorm) {
	return bicgstabFunctorHost<double>(a, b, x, maxIterations, eps, apply, user, solver_status, iterations, resnorm);
}

int smm_hip_set_cg_lazy_x_min_bytes(long long bytes) {
	g_lazyMinBytes.store(bytes, std::memory_order_relaxed);
	return SMM_HIP_OK;
}

int smm_hip_bicgsymmetric_f32(const smm_hip_csr* a, float* b, float* x, int maxIterations, float eps, int* solver_status, int* iterations) {
	return bicgsymmetricHost<float>(a, b, x, maxIterations, eps, solver_status, iterations);
}
int smm_hip_bicgsymmetric_f64(const smm_hip_csr* a, double* b, double* x, int maxIterations, double eps, int* solver_status, int* iterations) {
	return bicgsymmetricHost<double>(a, b, x, maxIterations, eps, solver_status, iterations);
}

}  // extern "C"
