// smm_resident.hip -- ConjugateGradient (ref:2316-2398) for matrices that fit the chip's REGISTER FILE: one launch per solve.
//
// Why (BASELINE config 2, the 1000 x 1000 Poisson matrix, fp64): the whole problem -- 60 MB of matrix, four 8 MB vectors -- is
// smaller than the 128 MB of vector registers of an MI355X (256 CUs x 512 KB), yet the three-launch iteration of cgDev
// (smm_solvers.hip) streams the matrix from the Infinity Cache every iteration: 41 us per iteration, of which ~5 us are kernel
// boundaries and the rest is moving 150 MB that never change.  Here every lane keeps R whole rows of the matrix (values, padded to K
// entries) and its rows of r and p in registers for the whole solve; one workgroup of 512 lanes per CU (8 waves x 256 VGPRs = the
// CU's register file); the positions, x and the workgroup's own slice of p sit in the CU's 160 KB of LDS.  Per iteration only the
// part of p that ANOTHER workgroup gathers has to cross workgroups, and two scalars (p.Ap, r.r) have to be summed over the chip:
// two grid-wide barriers per iteration, nothing else.
//
//   * p is never exchanged as such: a third barrier would be needed between "beta known" and "p gathered".  Instead an owner
//     publishes r (before the r.r barrier) and its previous p, and a consumer forms p_new[c] = beta p_old[c] + r[c] itself for
//     every foreign column it gathers -- the same expression on the same operands as the owner's own update, hence the same bits.
//     Columns of the workgroup's own chunk are read from LDS; a wave none of whose lanes needs a foreign column for an entry
//     skips that entry's two loads (on a banded matrix nine waves in ten do); only rows that some other workgroup gathers
//     (found once, at set-up: `needed`) are published at all.
//   * barriers are XCD-hierarchical (MI355X_MICROARCH.md, barrier-xcd): workgroups arrive on a counter of their XCD, the last
//     arriver of an XCD writes that L2's dirty lines back (one agent-scope release per XCD), arrives on the top counter, waits
//     for the 8 leaders and opens a generation word for its XCD; every workgroup then acquires at agent scope.  Every wait is
//     bounded (poll count): a timed-out solve reports it and the caller falls back to the three-launch loop; the kernel
//     writes x into a scratch vector that is copied out only after a clean run.
//   * the partial sums of a reduction are one slot per workgroup, added by every workgroup in the same fixed order (lane t takes
//     slot t, wave butterflies, 8 wave sums left to right): all workgroups get the same bits and leave the loop together.
//   * a row's dot product is formed left to right with _smm_fma exactly like the one-lane-per-row SpMV (smm_spmv.hip), so A p is
//     bit-identical to the library's SpMV; the global sums use a different partition of the rows than cgDev's, so alpha / beta
//     differ from cgDev's in the last bits (tests/test_gpu_resident.py: <= 1e-5 relative in fp32, <= 2e-14 in fp64, the same
//     iteration counts).
//
// Measured on config 2 (tools/cg_c2.py, profiles/r02/cg_config2.txt): see DESIGN.md section 3.4.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "smm_resident_sync.h"

namespace smm {

template <typename T>
struct ResidentArgs {
	int n;
	int maxIterations;
	int nnzAny;  // the matrix has at least one entry
	T eps;
	const int* start;
	const int* positions;
	const T* values;
	const T* b;
	const T* x0;
	T* x;
	T* pb[2];   // p of the previous iteration, double-buffered by iteration parity
	T* rg;      // r of this iteration
	T* parts;   // [2][gridDim.x] partial sums: p.Ap, r.r
	int* needed;  // [n], zeroed: needed[c] != 0 when a workgroup other than c's owner gathers column c
	ResidentSync* sync;
	ResidentOut<T>* out;
#ifdef SMM_RESIDENT_LAB
	int lab;  // measurement builds only (tools/run_resident_lab.sh): 1 no gathers, 2 no barrier waits, 4 no published stores, 8 no global sums
#endif
	long long waitTicks;  // bound of one wait, in polls (a poll is a load + s_sleep, ~0.5-1 us; polls do not advance while the queue is switched out)
};


#ifdef SMM_RESIDENT_LAB  // measurement builds (tools/run_resident_lab.sh): parts of an iteration can be switched off
#define PUBLISH(j) (((pubMask >> (j)) & 1u) != 0 && !(a.lab & 4))
#define GATHERS (!(a.lab & 1))
#else
#define PUBLISH(j) (((pubMask >> (j)) & 1u) != 0)
#define GATHERS true
#endif

template <typename T, int R, int K>
__global__ __launch_bounds__(RTPB) void cgResidentKernel(ResidentArgs<T> a) {
	__shared__ T sRed[RWAVES + 1];
	__shared__ int sOk;
	__shared__ unsigned sCensus[2];
	// Positions, row lengths, x and p of the workgroup's rows live in LDS, lane-interleaved (entry e of lane t at [e * RTPB + t]:
	// conflict-free); together with the values they would not fit the registers of a lane.  A position is stored ENCODED: a column
	// inside this workgroup's chunk as its index in the chunk (>= 0: its p is read from sP), any other column c as -(c + 1) (its p is
	// formed from the published vectors).
	extern __shared__ int sDyn[];
	int* const sCol = sDyn;                                                     // [R * K][RTPB]
	T* const sX = reinterpret_cast<T*>(sDyn + R * K * RTPB);                    // [R][RTPB]: x is only ever updated in place
	T* const sP = sX + R * RTPB;                                                // [R][RTPB] = p of rows chunk0 .. chunk0 + R * RTPB
	unsigned char* const sLen = reinterpret_cast<unsigned char*>(sP + R * RTPB);  // [R][RTPB]
	ResidentSync* const sy = a.sync;
	const int n = a.n;
	const int tid = threadIdx.x;
	const int chunk0 = blockIdx.x * (R * RTPB);

	// ---- census: which XCD am I on, how many workgroups live there, how many XCDs are populated --------------------------
	BarrierState st;
	st.xcc = residentXcc();
	st.epoch = 1;
	if (tid == 0) {
		__hip_atomic_fetch_add(&sy->pop[st.xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_fetch_add(&sy->census[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}

	// ---- the matrix rows of this lane, into registers -----------------------------------------------------------------------
	T val[R][K];
	T rr[R], pr[R];
#pragma unroll
	for (int j = 0; j < R; ++j) {
		const int row = chunk0 + j * RTPB + tid;
		int b0 = 0, l = 0;
		if (row < n) {
			b0 = a.start[row];
			l = a.start[row + 1] - b0;
		}
		sLen[j * RTPB + tid] = static_cast<unsigned char>(l);
		// r = b - A x0 (ref:2337), x = x0, p starts at 0 (beta = 0 makes the first "p = beta p + r" a copy, ref:2340)
		T dot = T(0);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const bool on = k < l;
			const int e = on ? b0 + k : 0;  // a padded entry reads entry 0 of the arrays (l > 0 somewhere, or nothing is read at all)
			const T v = a.nnzAny ? a.values[e] : T(0);
			const int c = a.nnzAny ? a.positions[e] : 0;
			val[j][k] = on ? v : T(0);
			const unsigned local = static_cast<unsigned>(c - chunk0);
			const bool mine = !on || local < static_cast<unsigned>(R * RTPB);
			sCol[(j * K + k) * RTPB + tid] = on ? (mine ? static_cast<int>(local) : -(c + 1)) : 0;
			if (!mine) a.needed[c] = 1;  // that row's owner must publish it (several lanes may store the same 1)
			const T next = smmFma(v, a.x0[on ? c : 0], dot);
			dot = on ? next : dot;
		}
		sX[j * RTPB + tid] = row < n ? a.x0[row] : T(0);
		rr[j] = row < n ? a.b[row] - dot : T(0);
		pr[j] = T(0);
		if (row < n) {
			a.pb[0][row] = T(0);
			a.rg[row] = rr[j];
		}
		asm volatile("" ::: "memory");
	}
	T acc = T(0);
#pragma unroll
	for (int j = 0; j < R; ++j) acc += rr[j] * rr[j];
	T s = blockSumAll(acc, sRed);
	T* const partsA = a.parts;
	T* const partsC = a.parts + gridDim.x;
	if (tid == 0) __hip_atomic_store(partsC + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

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
	if (alive) alive = gridBarrier<true>(sy, st, &sOk, a.waitTicks);
	// which of my rows does another workgroup gather?  Only those are published from now on.
	unsigned pubMask = 0;
#pragma unroll
	for (int j = 0; j < R; ++j) {
		const int row = chunk0 + j * RTPB + tid;
		if (row < n && a.needed[row] != 0) pubMask |= 1u << j;
	}
	const int safeRow = min(chunk0 + tid, n - 1);  // where a lane that needs no published value points its (discarded) load

	int iters = 0;
	int status = SMM_SOLVER_MAX_ITERATIONS_REACHED;
	T res = T(0);
	if (alive) {
		T rrOld = sumSlots(partsC, sRed);
		res = rrOld;
		const T eps2 = a.eps * a.eps;
		if (eps2 > rrOld) {
			status = SMM_SOLVER_SUCCESS;  // ref:2341-2344: x untouched
		} else {
			T beta = T(0);
			for (int it = 0; it < a.maxIterations; ++it) {
				const T* const pOld = a.pb[it & 1];
				T* const pNew = a.pb[(it & 1) ^ 1];
				const T* const rg = a.rg;
				// p = beta p + r (ref:2391-2393): kept in registers, shared with the workgroup through LDS, and -- the rows another
				// workgroup gathers -- published for the gathers of the NEXT iteration
				// (the row index is laundered through an empty asm in every phase: otherwise the compiler keeps the 64-bit addresses of
				// all R rows in all three published vectors alive across the whole loop, ~6 registers per row that the values need)
				int base = chunk0 + tid;
				asm volatile("" : "+v"(base));
#pragma unroll
				for (int j = 0; j < R; ++j) {
					const int row = base + j * RTPB;
					pr[j] = smmFma(beta, pr[j], rr[j]);
					sP[j * RTPB + tid] = pr[j];
					if (row < n && PUBLISH(j)) publish(pNew + row, pr[j]);
				}
				__syncthreads();
				// Ap = A p (ref:2353); p.Ap (ref:2354).  p of a column of this chunk comes from LDS; of any other column it is formed on
				// the fly as beta pOld[c] + r[c] -- the owner's expression on the owner's operands.  A wave in which no lane needs a
				// published value for entry k skips those two loads altogether (most waves of a banded matrix do).
				T ap[R];
				acc = T(0);
#pragma unroll
				for (int j = 0; j < R; ++j) {
					T dot = T(0);
					const int l = sLen[j * RTPB + tid];
					// G entries of the row at a time: their (at most 2 G) gathers are in flight together
					constexpr int G = sizeof(T) == 8 ? 3 : 9;  // (8 rows x 5 doubles per lane leave room for 3)
#pragma unroll
					for (int k0 = 0; k0 < K; k0 += G) {
						int enc[G];
						T po[G], rv[G];
#pragma unroll
						for (int g = 0; g < G; ++g) {
							if (k0 + g < K) {
								enc[g] = sCol[(j * K + k0 + g) * RTPB + tid];
								const bool need = enc[g] < 0 && GATHERS;
								po[g] = T(0);
								rv[g] = T(0);
								if (__builtin_amdgcn_ballot_w64(need) != 0) {
									// unconditional inside (a lane that does not need the value reads its own row and drops it)
									const int cg = need ? -enc[g] - 1 : safeRow;
									po[g] = pOld[cg];
									rv[g] = rg[cg];
								}
							}
						}
#pragma unroll
						for (int g = 0; g < G; ++g) {
							if (k0 + g < K) {
								const bool inside = enc[g] >= 0;
								const T mine = sP[inside ? enc[g] : 0];
								const T pc = inside ? mine : smmFma(beta, po[g], rv[g]);
								const T next = smmFma(val[j][k0 + g], pc, dot);
								dot = k0 + g < l ? next : dot;  // padded entries are dropped by a select, not by a multiplication with 0
							}
						}
					}
					ap[j] = dot;
					acc += pr[j] * dot;
					asm volatile("" ::: "memory");  // one row's gathers at a time: the registers are spoken for
				}
				s = blockSumAll(acc, sRed);
				if (tid == 0) __hip_atomic_store(partsA + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (!gridBarrier<PLAIN_PUBLISH>(sy, st, &sOk, a.waitTicks)) {
					alive = false;
					break;
				}
#ifdef SMM_RESIDENT_LAB
				const T alpha = rrOld / ((a.lab & 8) ? T(1) : sumSlots(partsA, sRed));
#else
				const T alpha = rrOld / sumSlots(partsA, sRed);  // ref:2358
#endif
				// x = alpha p + x ; r = -alpha Ap + r ; r.r (ref:2371-2375)
				acc = T(0);
				base = chunk0 + tid;
				asm volatile("" : "+v"(base));
#pragma unroll
				for (int j = 0; j < R; ++j) {
					const int row = base + j * RTPB;
					sX[j * RTPB + tid] = smmFma(alpha, pr[j], sX[j * RTPB + tid]);
					rr[j] = smmFma(-alpha, ap[j], rr[j]);
					if (row < n && PUBLISH(j)) publish(a.rg + row, rr[j]);
					acc += rr[j] * rr[j];
				}
				s = blockSumAll(acc, sRed);
				if (tid == 0) __hip_atomic_store(partsC + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (!gridBarrier<PLAIN_PUBLISH>(sy, st, &sOk, a.waitTicks)) {
					alive = false;
					break;
				}
#ifdef SMM_RESIDENT_LAB
				const T rrNew = (a.lab & 8) ? T(1) : sumSlots(partsC, sRed);
#else
				const T rrNew = sumSlots(partsC, sRed);
#endif
				iters = it + 1;
				res = rrNew;
				if (eps2 > rrNew) {  // ref:2377-2380
					status = SMM_SOLVER_SUCCESS;
					break;
				}
				beta = rrNew / rrOld;  // ref:2381
				rrOld = rrNew;
			}
		}
	}
	if (alive && iters > 0) {
#pragma unroll
		for (int j = 0; j < R; ++j) {
			const int row = chunk0 + j * RTPB + tid;
			if (row < n) a.x[row] = sX[j * RTPB + tid];
		}
	}
	if (blockIdx.x == 0 && tid == 0) {
		a.out->res = res;
		a.out->iters = iters;
		a.out->status = status;
		a.out->timedOut = alive ? 0 : 1;
	}
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxRowLenKernel(int rows, const int* __restrict__ start, int* __restrict__ out) {
	int m = 0;
	for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < rows; i += static_cast<long long>(gridDim.x) * blockDim.x) {
		m = max(m, start[i + 1] - start[i]);
	}
#pragma unroll
	for (int o = WAVE / 2; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, WAVE));
	if ((threadIdx.x & (WAVE - 1)) == 0 && m > 0) atomicMax(out, m);
}

static int maxRowLen(const smm_hip_csr* cm, hipStream_t s, int* out) {
	auto* m = const_cast<smm_hip_csr*>(cm);
	std::lock_guard<std::mutex> lock(m->readyMutex);
	if (m->max_row_len < 0) {
		DevBuf<int> d;
		SMM_TRY(d.alloc(1));
		SMM_HIP_TRY(hipMemsetAsync(d, 0, sizeof(int), s));
		if (m->rows > 0) maxRowLenKernel<<<std::min(2048, (m->rows + 255) / 256), 256, 0, s>>>(m->rows, m->d_start, d);
		int h = 0;
		SMM_HIP_TRY(hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
		m->max_row_len = h;
	}
	*out = m->max_row_len;
	return SMM_HIP_OK;
}

// SMM_CG_RESIDENT_OFF / AUTO / REQUIRE; the environment variable SMM_HIP_CG_RESIDENT sets the initial value
static std::atomic<int>& residentModeRef() {
	static std::atomic<int> mode{[] {
		const char* e = getenv("SMM_HIP_CG_RESIDENT");
		const int v = e ? atoi(e) : SMM_CG_RESIDENT_AUTO;
		return v < SMM_CG_RESIDENT_OFF || v > SMM_CG_RESIDENT_REQUIRE ? SMM_CG_RESIDENT_AUTO : v;
	}()};
	return mode;
}
static int residentMode() { return residentModeRef().load(); }

void preloadBicgResidentUnit();
void preloadResidentUnits() {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(maxRowLenKernel));
	(void)hipGetLastError();
	preloadBicgResidentUnit();
}

static std::mutex g_residentMutex;  // two grid-barrier kernels must never share the chip: each would wait for CUs the other holds
std::mutex& residentMutex() { return g_residentMutex; }

template <typename T>
static size_t residentLds(int R, int K) { return static_cast<size_t>(R) * RTPB * (K * sizeof(int) + 2 * sizeof(T) + 1); }

template <typename T, int R, int K>
static int launchResident(const ResidentArgs<T>& args, int grid, hipStream_t s) {
	static const int fits = [] {
		int perCU = 0;
		if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cgResidentKernel<T, R, K>), hipFuncAttributeMaxDynamicSharedMemorySize,
		                        static_cast<int>(residentLds<T>(R, K))) != hipSuccess)
			return 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, cgResidentKernel<T, R, K>, RTPB, residentLds<T>(R, K)) != hipSuccess) return 0;
		return perCU;
	}();
	if (fits < 1) return SMM_HIP_ERR_INVALID;
	cgResidentKernel<T, R, K><<<grid, RTPB, residentLds<T>(R, K), s>>>(args);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

// the (rows per lane, entries per row) shapes that are compiled; registers: R * (K * (sizeof(T) / 4 + 1) + 5 * sizeof(T) / 4) + ~25
template <typename T>
static int dispatchResident(int R, int K, const ResidentArgs<T>& args, int grid, hipStream_t s) {
#define SMM_RES_CASE(RR, KK) \
	if (R == RR && K == KK) return launchResident<T, RR, KK>(args, grid, s);
	SMM_RES_CASE(1, 5)
	SMM_RES_CASE(2, 5)
	SMM_RES_CASE(4, 5)
	SMM_RES_CASE(8, 5)
	SMM_RES_CASE(1, 9)
	SMM_RES_CASE(2, 9)
	SMM_RES_CASE(4, 9)
	SMM_RES_CASE(1, 16)
	SMM_RES_CASE(2, 16)
	SMM_RES_CASE(1, 27)
	if constexpr (sizeof(T) == 4) {
		SMM_RES_CASE(2, 27)
	}
#undef SMM_RES_CASE
	return SMM_HIP_ERR_INVALID;
}

// Tries the register-resident solve.  *handled = false (and nothing written) when the matrix does not fit, the mode is off, or the
// launch gave up at a barrier; the caller then runs the three-launch loop.
template <typename T>
int cgResidentTry(const smm_hip_csr* a, const T* b, const T* x0, T* x, int maxIterations, T eps, hipStream_t s, int* status, int* iterations,
                  T* resnorm2, bool* handled) {
	*handled = false;
	const int mode = residentMode();
	if (mode == SMM_CG_RESIDENT_OFF) return SMM_HIP_OK;
	// under SMM_CG_RESIDENT_REQUIRE (tests, measurements) "does not apply" is an error instead of a silent fall-back
	auto notApplicable = [mode](const char* why) {
		if (mode != SMM_CG_RESIDENT_REQUIRE) return static_cast<int>(SMM_HIP_OK);
		setError("cg: the register-resident solve was required but %s", why);
		return static_cast<int>(SMM_HIP_ERR_INVALID);
	};
	const int n = a->rows;
	if (n <= 0 || maxIterations == 0) return SMM_HIP_OK;  // nothing to iterate: the general path handles the corner cases
	const int cus = numCUs();
	if (static_cast<long long>(n) > static_cast<long long>(cus) * 8 * RTPB) return notApplicable("the matrix has too many rows");
	int longest = 0;
	SMM_TRY(maxRowLen(a, s, &longest));
	static const int KS[] = {5, 9, 16, 27};
	const int RMAX[] = {8, 4, 2, sizeof(T) == 4 ? 2 : 1};
	int K = 0, R = 0;
	for (int i = 0; i < 4 && !K; ++i) {
		if (longest > KS[i]) continue;
		// fewest rows per lane that still fit the chip: more workgroups take part
		for (int r = 1; r <= RMAX[i]; r *= 2) {
			if (static_cast<long long>(n) <= static_cast<long long>(cus) * r * RTPB) {
				K = KS[i];
				R = r;
				break;
			}
		}
		if (!K) return notApplicable("rows of this length do not fit the register file");  // wider shapes hold fewer rows still
	}
	if (!K) return notApplicable("its longest row has more than 27 entries");
	const int grid = (n + R * RTPB - 1) / (R * RTPB);

	DevBuf<T> pb0, pb1, rg, parts, xOut;
	DevBuf<ResidentSync> sync;
	DevBuf<int> needed;
	SMM_TRY(needed.alloc(n));
	DevBuf<ResidentOut<T>> out;
	SMM_TRY(pb0.alloc(n));
	SMM_TRY(pb1.alloc(n));
	SMM_TRY(rg.alloc(n));
	SMM_TRY(xOut.alloc(n));
	SMM_TRY(parts.alloc(2 * static_cast<size_t>(grid)));
	SMM_TRY(sync.alloc(1));
	SMM_TRY(out.alloc(1));
	ResidentArgs<T> args;
	args.n = n;
	args.maxIterations = maxIterations == -1 ? n : maxIterations;  // ref:2345-2347
	args.eps = eps;
	args.nnzAny = a->nnz > 0 ? 1 : 0;
	args.start = a->d_start;
	args.positions = a->d_positions;
	args.values = static_cast<const T*>(a->d_values);
	args.b = b;
	args.x0 = x0;
	args.x = xOut;
	args.pb[0] = pb0;
	args.pb[1] = pb1;
	args.rg = rg;
	args.parts = parts;
	args.sync = sync;
	args.needed = needed;
	args.out = out;
	// Every barrier wait is bounded.  REQUIRE (tests, measurements) waits a few seconds; AUTO gives up after a fraction of that: a
	// plain launch cannot reserve the CUs (a cooperative launch only checks the grid against the occupancy query, it does not wait for
	// CUs another stream or process holds -- MI355X_MICROARCH.md, coop-launch), so when workgroups are not co-resident the honest move
	// is to fall back quickly, and -- below -- to stop trying for the rest of the process.
	args.waitTicks = mode == SMM_CG_RESIDENT_REQUIRE ? 1LL << 22 : 1LL << 19;
#ifdef SMM_RESIDENT_LAB
	args.lab = getenv("SMM_RESIDENT_LAB") ? atoi(getenv("SMM_RESIDENT_LAB")) : 0;
	if (args.lab & 2) args.waitTicks = -1;
#endif
	ResidentOut<T> h;
	unsigned gaveUp = 0;
	{
		std::lock_guard<std::mutex> lock(g_residentMutex);
		SMM_HIP_TRY(hipMemsetAsync(sync, 0, sizeof(ResidentSync), s));
		SMM_HIP_TRY(hipMemsetAsync(needed, 0, static_cast<size_t>(n) * sizeof(int), s));
		const int st = dispatchResident<T>(R, K, args, grid, s);
		if (st != SMM_HIP_OK) return notApplicable("the kernel does not fit a CU of this device");
		SMM_HIP_TRY(hipMemcpyAsync(&h, out, sizeof(h), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipMemcpyAsync(&gaveUp, &sync.p->timeout[0], sizeof(unsigned), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
	}
	if (h.timedOut || gaveUp) {
		if (mode == SMM_CG_RESIDENT_AUTO) {
			// remember it: the CUs this solve needs all at once are shared with somebody; every later AUTO solve would pay the same stall
			int expected = SMM_CG_RESIDENT_AUTO;
			if (residentModeRef().compare_exchange_strong(expected, SMM_CG_RESIDENT_OFF)) {
				fprintf(stderr, "libsmm_hip: the register-resident ConjugateGradient gave up at a grid barrier (CUs held by another stream or process?); "
				                "switched off for this process (smm_hip_cg_resident / SMM_HIP_CG_RESIDENT turn it back on)\n");
			}
		}
		return notApplicable("a grid barrier timed out (is another persistent kernel holding CUs?)");
	}
	if (h.iters > 0) SMM_HIP_TRY(hipMemcpyAsync(x, xOut.p, static_cast<size_t>(n) * sizeof(T), hipMemcpyDeviceToDevice, s));
	if (status) *status = h.status;
	if (iterations) *iterations = h.iters;
	if (resnorm2) *resnorm2 = h.res;
	*handled = true;
	return SMM_HIP_OK;
}

}  // namespace smm

extern "C" int smm_hip_cg_resident(int mode) {
	const int before = smm::residentModeRef().load();
	if (mode >= SMM_CG_RESIDENT_OFF && mode <= SMM_CG_RESIDENT_REQUIRE) smm::residentModeRef().store(mode);
	return before;
}

namespace smm {

template int cgResidentTry<float>(const smm_hip_csr*, const float*, const float*, float*, int, float, hipStream_t, int*, int*, float*, bool*);
template int cgResidentTry<double>(const smm_hip_csr*, const double*, const double*, double*, int, double, hipStream_t, int*, int*, double*, bool*);

}  // namespace smm
