// smm_resident_bicg.hip -- BiCGStab (ref:2191-2283, incl. the preconditioned template with the library's Jacobi) as ONE launch per solve
// for matrices whose vectors fit the chip's register file.
//
// Why (BASELINE config 5 -- a 1.26 M-row nonsymmetric stencil matrix, fp64): the loop of smm_solvers.hip runs 7 dependent launches per
// iteration, each moving 20-50 MB that sit in the Infinity Cache anyway: 73 us per iteration of which more than half are the gaps
// between the launches (profiles/r04: 321 iterations in 23.5 ms).  Here one persistent kernel keeps the vectors of the recurrence in
// registers and LDS for the whole solve:
//
//   * a lane owns R rows (row = chunk0 + j * 512 + lane: coalesced whenever a vector does touch memory) and holds r, p, s, A p and A s
//     of its rows in registers, x and r0 -- which only their owner ever reads -- in LDS; one workgroup of 512 lanes per CU;
//   * the matrix must be in the PATTERN family's row-mask encoding (smm_spmv_pattern.hip: every row's columns are row + off[j] for the
//     set bits j of its mask, at most 16 offsets) -- what the solvers adopt for banded / stencil matrices from 2^20 entries.  With
//     constant diagonals (CONST) the values are 16 kernel arguments and an SpMV reads nothing but the gathered vector; otherwise the
//     values are read from a slot-major copy (`ell[j * n + row]`: coalesced, built once per matrix, cached in the handle);
//   * only what ANOTHER lane gathers crosses workgroups: p and s, published write-through (agent-scope stores) right after they are
//     formed and gathered from the L2 after a grid barrier.  A workgroup's chunk of rows is dealt to the XCDs in contiguous eighths, so
//     most gathers of a stencil hit the L2 of the XCD that wrote them;
//   * five grid barriers per iteration: the three reduction points of BiCGStab (ap.r0 ; as.as, as.s ; r.r, r.r0) and the two
//     publications (s, p).  Barriers, sums and publishing are those of the resident ConjugateGradient (smm_resident_sync.h): XCD-
//     hierarchical, every wait bounded -- a solve that times out reports it, nothing has been written to x, and the caller runs the loop;
//   * a row's dot product is formed left to right over its set bits with _smm_fma, like the one-lane-per-row SpMV kernels: A p has the
//     bits of the library's SpMV.  The update expressions are those of bicgFusedS / bicgFusedXR / bicgFusedP (smm_solvers.hip), i.e. the
//     reference's; the global sums add the rows in another (fixed) partition than the loop's, so alpha / omega / beta differ from the
//     loop's in the last bits (tests/test_gpu_resident.py).
//   * Jacobi (M = diag(A), ref:2217-2224, 2235, 2251): the division rides in the row, as SPMV_DIV_LHS does in the loop.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "smm_resident_sync.h"

namespace smm {

constexpr int BR_MAXK = 16;

template <typename T>
struct BicgResArgs {
	int n;
	int planned;  // passes of the loop body: max(1, maxIterations) -- do { } while (ref:2232, 2277)
	int k;        // offsets in use
	int chunksPerXcd;
	T eps;
	int off[BR_MAXK];
	T cval[BR_MAXK];  // CONST: the value of diagonal j
	const unsigned long long* masks;
	const T* ell;   // values by offset slot, [k][n]; null for CONST
	const T* diag;  // Jacobi: the diagonal; null without a preconditioner
	const T* b;
	const T* x;  // the caller's x (read only: the result goes to xOut and is copied after a clean run)
	T* xOut;
	T* pg;     // published p
	T* sg;     // published s
	T* parts;  // [5][gridDim.x]: ap.r0 | as.as | as.s | r.r | r.r0
	ResidentSync* sync;
	ResidentOut<T>* out;
	long long waitTicks;
};

// dot product of one row with a vector in global memory: the set bits of the mask in ascending order = ascending columns (ref:1484-1489)
template <typename T, int K, bool CONSTV>
__device__ __forceinline__ T rowDot(const BicgResArgs<T>& a, int row, unsigned mask, const T* __restrict__ vec) {
	T xv[K], vv[K];
#pragma unroll
	for (int j = 0; j < K; ++j) {
		xv[j] = T(0);
		vv[j] = T(0);
		if (j < a.k) {
			const bool on = ((mask >> j) & 1u) != 0;
			const int c = on ? row + a.off[j] : row;  // (a lane without the entry reads its own row and drops the product)
			xv[j] = vec[c];
			if (!CONSTV) vv[j] = a.ell[static_cast<size_t>(j) * a.n + row];
		}
	}
	T dot = T(0);
#pragma unroll
	for (int j = 0; j < K; ++j) {
		if (j < a.k) {
			const T next = smmFma(CONSTV ? a.cval[j] : vv[j], xv[j], dot);
			dot = ((mask >> j) & 1u) != 0 ? next : dot;
		}
	}
	return dot;
}

template <typename T>
__device__ __forceinline__ T sqrtRn(T v) {
	if constexpr (sizeof(T) == 4) return __fsqrt_rn(v);
	else return __dsqrt_rn(v);
}

template <typename T, int R, int K, bool CONSTV>
__global__ __launch_bounds__(RTPB) void bicgResidentKernel(BicgResArgs<T> a) {
	__shared__ T sRed[RWAVES + 1];
	__shared__ int sOk;
	__shared__ unsigned sCensus[2];
	extern __shared__ __attribute__((aligned(16))) unsigned char sDynBicg[];
	T* const sX = reinterpret_cast<T*>(sDynBicg);  // [R][RTPB]
	T* const sR0 = sX + R * RTPB;                  // [R][RTPB]
	ResidentSync* const sy = a.sync;
	const int n = a.n;
	const int tid = threadIdx.x;
	// workgroup b runs on XCD b % 8 (round-robin dispatch): the chunks of an XCD are contiguous, so the gathers of a stencil stay in its L2
	const int chunk = (blockIdx.x & 7) * a.chunksPerXcd + (blockIdx.x >> 3);
	const int chunk0 = chunk * (R * RTPB);  // (may lie past the last row: such a workgroup owns nothing and only keeps the barriers whole)
	const bool jacobi = a.diag != nullptr;

	BarrierState st;
	st.xcc = residentXcc();
	st.epoch = 1;
	if (tid == 0) {
		__hip_atomic_fetch_add(&sy->pop[st.xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_fetch_add(&sy->census[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}

	// ---- r = [M^-1] (b - A x), r0 = p = r (ref:2215-2226); rr0 = r.r0 (ref:2231) -------------------------------------------------------------
	unsigned mask[R];
	T rr[R], pr[R], sr[R], apr[R], asr[R];
	T acc0 = T(0), acc1 = T(0);
	// (the row index is laundered through an empty asm in every phase and every row ends with a compiler barrier: otherwise the 64-bit
	// addresses of all R rows in every vector stay alive across the whole loop and the R x K gathers of a phase are hoisted together --
	// kilobytes of scratch per lane; one row's gathers in flight per lane, eight waves per CU, is what the registers allow)
	int base = chunk0 + tid;
	asm volatile("" : "+v"(base));
#pragma unroll
	for (int j = 0; j < R; ++j) {
		const int row = base + j * RTPB;
		const bool live = row < n;
		const int rowc = live ? row : n - 1;
		mask[j] = live ? static_cast<unsigned>(a.masks[rowc]) : 0u;
		const T dot = rowDot<T, K, CONSTV>(a, rowc, mask[j], a.x);
		T v = a.b[rowc] - dot;
		if (jacobi) v = v / a.diag[rowc];
		v = live ? v : T(0);
		rr[j] = v;
		pr[j] = v;
		sr[j] = T(0);
		apr[j] = T(0);
		asr[j] = T(0);
		sX[j * RTPB + tid] = live ? a.x[rowc] : T(0);
		sR0[j * RTPB + tid] = v;
		if (live) publish(a.pg + row, v);
		acc0 += v * v;
		asm volatile("" ::: "memory");
	}
	const int G = gridDim.x;
	T* const partsA = a.parts;
	T* const partsB0 = a.parts + G;
	T* const partsB1 = a.parts + 2 * G;
	T* const partsC0 = a.parts + 3 * G;
	T* const partsC1 = a.parts + 4 * G;
	T s0 = blockSumAll(acc0, sRed);
	if (tid == 0) __hip_atomic_store(partsC1 + blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

	// census wait (flat), then the first real barrier
	if (tid == 0) {
		const bool ok = waitAtLeast(&sy->census[0], gridDim.x, &sy->timeout[0], a.waitTicks);
		unsigned nx = 0, mine = 0;
		for (int x = 0; x < MAX_XCD; ++x) {
			const unsigned p = __hip_atomic_load(&sy->pop[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			nx += p != 0;
			if (x == st.xcc) mine = p;
		}
		sCensus[0] = ok ? mine : 0;
		sCensus[1] = nx;
	}
	__syncthreads();
	st.pop = sCensus[0];
	st.nx = sCensus[1];
	bool alive = st.pop != 0;
	if (alive) alive = gridBarrier<false>(sy, st, &sOk, a.waitTicks);

	int iters = 0;
	T res = T(0);
	if (alive) {
		T rr0 = sumSlots(partsC1, sRed);
		for (int it = 0; it < a.planned; ++it) {
			// ap = [M^-1] A p ; ap.r0 (ref:2233-2243)
			acc0 = T(0);
			base = chunk0 + tid;
			asm volatile("" : "+v"(base));
#pragma unroll
			for (int j = 0; j < R; ++j) {
				const int row = base + j * RTPB;
				const bool live = row < n;
				const int rowc = live ? row : n - 1;
				T v = rowDot<T, K, CONSTV>(a, rowc, mask[j], a.pg);
				if (jacobi) v = v / a.diag[rowc];
				v = live ? v : T(0);
				apr[j] = v;
				acc0 += v * sR0[j * RTPB + tid];
				asm volatile("" ::: "memory");
			}
			s0 = blockSumAll(acc0, sRed);
			if (tid == 0) __hip_atomic_store(partsA + blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (!gridBarrier<false>(sy, st, &sOk, a.waitTicks)) {
				alive = false;
				break;
			}
			// alpha = rr0 / (ap.r0) ; s = -alpha ap + r (ref:2243-2247)
			const T alpha = rr0 / sumSlots(partsA, sRed);
			base = chunk0 + tid;
			asm volatile("" : "+v"(base));
#pragma unroll
			for (int j = 0; j < R; ++j) {
				const int row = base + j * RTPB;
				sr[j] = smmFma(-alpha, apr[j], rr[j]);
				if (row < n) publish(a.sg + row, sr[j]);
			}
			if (!gridBarrier<false>(sy, st, &sOk, a.waitTicks)) {
				alive = false;
				break;
			}
			// as = [M^-1] A s ; as.as, as.s (ref:2249-2261)
			acc0 = T(0);
			acc1 = T(0);
			base = chunk0 + tid;
			asm volatile("" : "+v"(base));
#pragma unroll
			for (int j = 0; j < R; ++j) {
				const int row = base + j * RTPB;
				const bool live = row < n;
				const int rowc = live ? row : n - 1;
				T v = rowDot<T, K, CONSTV>(a, rowc, mask[j], a.sg);
				if (jacobi) v = v / a.diag[rowc];
				v = live ? v : T(0);
				asr[j] = v;
				acc0 += v * v;
				acc1 += v * sr[j];
				asm volatile("" ::: "memory");
			}
			s0 = blockSumAll(acc0, sRed);
			T s1 = blockSumAll(acc1, sRed);
			if (tid == 0) {
				__hip_atomic_store(partsB0 + blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(partsB1 + blockIdx.x, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (!gridBarrier<false>(sy, st, &sOk, a.waitTicks)) {
				alive = false;
				break;
			}
			// omega = (as.s) / (as.as) ; x = alpha p + (omega s + x) ; r = -omega as + s ; r.r, r.r0 (ref:2259-2269)
			const T asas = sumSlots(partsB0, sRed);
			const T ass = sumSlots(partsB1, sRed);
			const T omega = ass / asas;
			acc0 = T(0);
			acc1 = T(0);
#pragma unroll
			for (int j = 0; j < R; ++j) {
				const T si = sr[j];
				sX[j * RTPB + tid] = smmFma(alpha, pr[j], smmFma(omega, si, sX[j * RTPB + tid]));
				const T ri = smmFma(-omega, asr[j], si);
				rr[j] = ri;
				acc0 += ri * ri;
				acc1 += ri * sR0[j * RTPB + tid];
			}
			s0 = blockSumAll(acc0, sRed);
			s1 = blockSumAll(acc1, sRed);
			if (tid == 0) {
				__hip_atomic_store(partsC0 + blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(partsC1 + blockIdx.x, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (!gridBarrier<false>(sy, st, &sOk, a.waitTicks)) {
				alive = false;
				break;
			}
			// resL2Norm, loop test, beta, p = beta (-omega ap + p) + r (ref:2268-2277)
			const T rrNew = sumSlots(partsC0, sRed);
			const T newRR0 = sumSlots(partsC1, sRed);
			res = sqrtRn<T>(rrNew);
			iters = it + 1;
			if (!(res > a.eps)) break;  // while (resL2Norm > eps ...): NaN leaves the loop too
			if (it + 1 == a.planned) break;
			const T beta = (newRR0 * alpha) / (rr0 * omega);  // ref:2271
			rr0 = newRR0;
			base = chunk0 + tid;
			asm volatile("" : "+v"(base));
#pragma unroll
			for (int j = 0; j < R; ++j) {
				const int row = base + j * RTPB;
				pr[j] = smmFma(beta, smmFma(-omega, apr[j], pr[j]), rr[j]);
				if (row < n) publish(a.pg + row, pr[j]);
			}
			if (!gridBarrier<false>(sy, st, &sOk, a.waitTicks)) {
				alive = false;
				break;
			}
		}
	}
	if (alive) {
		base = chunk0 + tid;
		asm volatile("" : "+v"(base));
#pragma unroll
		for (int j = 0; j < R; ++j) {
			const int row = base + j * RTPB;
			if (row < n) a.xOut[row] = sX[j * RTPB + tid];
		}
	}
	if (blockIdx.x == 0 && tid == 0) {
		a.out->res = res;
		a.out->iters = iters;
		a.out->status = SMM_SOLVER_SUCCESS;
		a.out->timedOut = alive ? 0 : 1;
	}
}

// ---- slot-major values for the matrices whose diagonals vary: ell[j * n + row] = value of the row's entry at offset j ----------------------
template <typename T>
__global__ __launch_bounds__(256) void ellFromMasksKernel(int n, int k, const int* __restrict__ start, const unsigned long long* __restrict__ masks,
                                                          const T* __restrict__ values, T* __restrict__ ell) {
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
		const unsigned long long m = masks[row];
		int at = start[row];
		for (int j = 0; j < k; ++j) {
			T v = T(0);
			if ((m >> j) & 1ull) v = values[at++];
			ell[static_cast<size_t>(j) * n + row] = v;
		}
	}
}

// SMM_CG_RESIDENT_OFF / AUTO / REQUIRE (shared constants); the environment variable SMM_HIP_BICGSTAB_RESIDENT sets the initial value
static std::atomic<int>& bicgResidentModeRef() {
	static std::atomic<int> mode{[] {
		const char* e = getenv("SMM_HIP_BICGSTAB_RESIDENT");
		const int v = e ? atoi(e) : SMM_CG_RESIDENT_AUTO;
		return v < SMM_CG_RESIDENT_OFF || v > SMM_CG_RESIDENT_REQUIRE ? SMM_CG_RESIDENT_AUTO : v;
	}()};
	return mode;
}

template <typename T, int R>
static size_t bicgLds() { return 2 * static_cast<size_t>(R) * RTPB * sizeof(T); }

template <typename T, int R, int K, bool CONSTV>
static int launchBicgResident(const BicgResArgs<T>& args, int grid, hipStream_t s) {
	static const int fits = [] {
		int perCU = 0;
		if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bicgResidentKernel<T, R, K, CONSTV>), hipFuncAttributeMaxDynamicSharedMemorySize,
		                        static_cast<int>(bicgLds<T, R>())) != hipSuccess)
			return 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, bicgResidentKernel<T, R, K, CONSTV>, RTPB, bicgLds<T, R>()) != hipSuccess) return 0;
		return perCU;
	}();
	if (fits < 1) {
		(void)hipGetLastError();
		return SMM_HIP_ERR_INVALID;
	}
	bicgResidentKernel<T, R, K, CONSTV><<<grid, RTPB, bicgLds<T, R>(), s>>>(args);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T, int R>
static int dispatchBicgRK(int K, bool constv, const BicgResArgs<T>& args, int grid, hipStream_t s) {
	if (K <= 8) return constv ? launchBicgResident<T, R, 8, true>(args, grid, s) : launchBicgResident<T, R, 8, false>(args, grid, s);
	return constv ? launchBicgResident<T, R, 16, true>(args, grid, s) : launchBicgResident<T, R, 16, false>(args, grid, s);
}

// rows per lane that are compiled: registers 5 R values + R masks + the gathers of one row in flight
template <typename T>
static const int* bicgRowChoices(int* count) {
	static const int f64[] = {2, 4, 8, 10, 12};
	static const int f32[] = {2, 4, 8, 12, 16, 24};
	*count = sizeof(T) == 8 ? 5 : 6;
	return sizeof(T) == 8 ? f64 : f32;
}

template <typename T>
static int dispatchBicg(int R, int K, bool constv, const BicgResArgs<T>& args, int grid, hipStream_t s) {
	switch (R) {
	case 2: return dispatchBicgRK<T, 2>(K, constv, args, grid, s);
	case 4: return dispatchBicgRK<T, 4>(K, constv, args, grid, s);
	case 8: return dispatchBicgRK<T, 8>(K, constv, args, grid, s);
	case 12: return dispatchBicgRK<T, 12>(K, constv, args, grid, s);
	default: break;
	}
	if constexpr (sizeof(T) == 8) {
		if (R == 10) return dispatchBicgRK<T, 10>(K, constv, args, grid, s);
	} else {
		if (R == 16) return dispatchBicgRK<T, 16>(K, constv, args, grid, s);
		if (R == 24) return dispatchBicgRK<T, 24>(K, constv, args, grid, s);
	}
	return SMM_HIP_ERR_INVALID;
}

// Tries the single-launch solve.  *handled = false (and nothing written to x) when the matrix does not qualify, the mode is off or the
// launch gave up at a barrier; the caller then runs the loop of smm_solvers.hip.  jacobiDiag: M is the library's Jacobi preconditioner.
template <typename T>
int bicgstabResidentTry(const smm_hip_csr* ca, const T* b, T* x, int maxIterations, T eps, const T* jacobiDiag, hipStream_t s, int* status,
                        int* iterations, T* resnorm, bool* handled) {
	*handled = false;
	const int mode = bicgResidentModeRef().load();
	if (mode == SMM_CG_RESIDENT_OFF) return SMM_HIP_OK;
	auto notApplicable = [mode](const char* why) {
		if (mode != SMM_CG_RESIDENT_REQUIRE) return static_cast<int>(SMM_HIP_OK);
		setError("bicgstab: the single-launch solve was required but %s", why);
		return static_cast<int>(SMM_HIP_ERR_INVALID);
	};
	auto* a = const_cast<smm_hip_csr*>(ca);
	const int n = a->rows;
	if (n <= 0) return SMM_HIP_OK;
	if (a->family() != SMM_SPMV_PATTERN || a->pat_state.load(std::memory_order_acquire) <= 0 || a->pat_encoding != 0)
		return notApplicable("the matrix is not in the PATTERN family's row-mask encoding");
	const int k = a->pat_k;
	if (k < 1 || k > BR_MAXK || static_cast<int>(a->pat_offs_host.size()) != k) return notApplicable("its rows use more than 16 column offsets");
	// below ~2^17 rows the loop's kernels are a few microseconds each and the matrix never adopts PATTERN by itself; above, the rows must fit
	const int cus = numCUs();
	int nChoices = 0;
	const int* choices = bicgRowChoices<T>(&nChoices);
	int R = 0;
	for (int i = 0; i < nChoices && !R; ++i) {
		if (static_cast<long long>(n) <= static_cast<long long>(cus / 8 * 8) * choices[i] * RTPB) R = choices[i];
	}
	if (!R) return notApplicable("the vectors do not fit the register file");
	const bool constv = a->pat_const && !a->pat_const_off;
	if (constv && static_cast<int>(a->pat_cval_host.size()) < k) return notApplicable("the constant diagonals' values are not known to the host");
	int planned = std::min(maxIterations, n);  // ref:2200
	if (maxIterations == -1) planned = n;      // ref:2201-2203
	const int maxItClamped = planned;
	planned = std::max(1, planned);
	const int nChunks = (n + R * RTPB - 1) / (R * RTPB);
	const int chunksPerXcd = (nChunks + 7) / 8;
	const int grid = 8 * chunksPerXcd;
	if (grid > cus) return notApplicable("the vectors do not fit the register file");

	// slot-major values, once per matrix
	if (!constv) {
		std::lock_guard<std::mutex> lock(a->tileMutex);
		if (!a->d_res_ell) {
			void* p = nullptr;
			SMM_TRY(devAlloc(&p, static_cast<size_t>(k) * n * sizeof(T)));
			ellFromMasksKernel<T><<<std::min(4096, (n + 255) / 256), 256, 0, s>>>(n, k, a->d_start, a->d_pat_masks, static_cast<const T*>(a->d_values), static_cast<T*>(p));
			hipError_t e = hipGetLastError();
			if (e == hipSuccess) e = hipStreamSynchronize(s);  // (published in the handle: a solve on another stream may use it at once)
			if (e != hipSuccess) {
				devFree(p);
				return hipFail(e, "ellFromMasksKernel", __FILE__, __LINE__);
			}
			a->d_res_ell = p;
		}
	}

	DevBuf<T> pg, sg, xOut, parts;
	DevBuf<ResidentSync> sync;
	DevBuf<ResidentOut<T>> out;
	SMM_TRY(pg.alloc(n));
	SMM_TRY(sg.alloc(n));
	SMM_TRY(xOut.alloc(n));
	SMM_TRY(parts.alloc(5 * static_cast<size_t>(grid)));
	SMM_TRY(sync.alloc(1));
	SMM_TRY(out.alloc(1));
	BicgResArgs<T> args{};
	args.n = n;
	args.planned = planned;
	args.k = k;
	args.chunksPerXcd = chunksPerXcd;
	args.eps = eps;
	for (int j = 0; j < k; ++j) {
		args.off[j] = a->pat_offs_host[static_cast<size_t>(j)];
		if (constv) {
			const unsigned long long bits = a->pat_cval_host[static_cast<size_t>(j)];
			if constexpr (sizeof(T) == 8) {
				memcpy(&args.cval[j], &bits, 8);
			} else {
				const unsigned lo = static_cast<unsigned>(bits);
				memcpy(&args.cval[j], &lo, 4);
			}
		}
	}
	args.masks = a->d_pat_masks;
	args.ell = constv ? nullptr : static_cast<const T*>(a->d_res_ell);
	args.diag = jacobiDiag;
	args.b = b;
	args.x = x;
	args.xOut = xOut;
	args.pg = pg;
	args.sg = sg;
	args.parts = parts;
	args.sync = sync;
	args.out = out;
	args.waitTicks = mode == SMM_CG_RESIDENT_REQUIRE ? 1LL << 22 : 1LL << 19;
	ResidentOut<T> h;
	unsigned gaveUp = 0;
	{
		std::lock_guard<std::mutex> lock(residentMutex());
		SMM_HIP_TRY(hipMemsetAsync(sync, 0, sizeof(ResidentSync), s));
		const int st = dispatchBicg<T>(R, k, constv, args, grid, s);
		if (st != SMM_HIP_OK) return notApplicable("the kernel does not fit a CU of this device");
		SMM_HIP_TRY(hipMemcpyAsync(&h, out, sizeof(h), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipMemcpyAsync(&gaveUp, &sync.p->timeout[0], sizeof(unsigned), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
	}
	if (h.timedOut || gaveUp) {
		if (mode == SMM_CG_RESIDENT_AUTO) {
			int expected = SMM_CG_RESIDENT_AUTO;
			if (bicgResidentModeRef().compare_exchange_strong(expected, SMM_CG_RESIDENT_OFF)) {
				fprintf(stderr, "libsmm_hip: the single-launch BiCGStab gave up at a grid barrier (CUs held by another stream or process?); switched off "
				                "for this process (smm_hip_bicgstab_resident / SMM_HIP_BICGSTAB_RESIDENT turn it back on)\n");
			}
		}
		return notApplicable("a grid barrier timed out (is another persistent kernel holding CUs?)");
	}
	SMM_HIP_TRY(hipMemcpyAsync(x, xOut.p, static_cast<size_t>(n) * sizeof(T), hipMemcpyDeviceToDevice, s));
	if (status) *status = h.iters > maxItClamped ? SMM_SOLVER_MAX_ITERATIONS_REACHED : SMM_SOLVER_SUCCESS;  // ref:2279-2282
	if (iterations) *iterations = h.iters;
	if (resnorm) *resnorm = h.res;
	*handled = true;
	return SMM_HIP_OK;
}

void preloadBicgResidentUnit() {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(ellFromMasksKernel<double>));
	(void)hipGetLastError();
}

template int bicgstabResidentTry<float>(const smm_hip_csr*, const float*, float*, int, float, const float*, hipStream_t, int*, int*, float*, bool*);
template int bicgstabResidentTry<double>(const smm_hip_csr*, const double*, double*, int, double, const double*, hipStream_t, int*, int*, double*, bool*);

}  // namespace smm

extern "C" int smm_hip_bicgstab_resident(int mode) {
	const int before = smm::bicgResidentModeRef().load();
	if (mode >= SMM_CG_RESIDENT_OFF && mode <= SMM_CG_RESIDENT_REQUIRE) smm::bicgResidentModeRef().store(mode);
	return before;
}
