// smm_precond.hip -- preconditioner apply on gfx950: Jacobi, Symmetric Gauss-Seidel (ref:1658-1713), ILU(0)
// (declared ref:1189-1212, unusable there), IC(0) (ref:1803-1837).
//
// SGS / ILU0 / IC0 apply are two sparse triangular solves on A's own pattern.  The reference walks the rows
// one after another; here the dependency DAG of each sweep is cut into levels once, at create time (rows of one
// level depend only on rows of earlier levels), rows are sorted by level, and a sweep runs level by level with
// one lane per row.  Inside a row the entries are visited in exactly the reference's order, and a row only
// starts when all rows it reads are final, so the result is bit-identical to the sequential sweep.
// Runs of small levels are executed by a single 1024-lane workgroup with a workgroup barrier between levels
// (no launch per level); a large level gets a multi-workgroup launch of its own.
//
// Set-up (structural checks, the level sets, the ILU0 / IC0 factorisation) runs on the device as well, at create time: the level sets
// by a synchronisation-free propagation kernel, the factorisation level by level with one wavefront per row.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>

#include <rocprim/device/device_radix_sort.hpp>

#include "smm_device.h"
#include "smm_internal.h"

namespace smm {

constexpr int TPB = 256;
constexpr int SMALL_LEVEL = 1024;  // levels up to this many rows are chained inside one workgroup

enum SweepMode { SGS_LO = 0, SGS_UP = 1, ILU_LO = 2, ILU_UP = 3, IC_LO = 4, IC_UP = 5 };

template <typename T, int MODE>
__device__ __forceinline__ void solveRow(int row, const int* __restrict__ start, const int* __restrict__ positions, const T* __restrict__ vals,
                                         const T* __restrict__ rhs, T* x) {
	const int b = start[row];
	const int e = start[row + 1];
	if (MODE == SGS_LO) {  // ref:1673-1695
		int k = b;
		T lhs = rhs[row];
		int col = positions[k];
		T value = vals[k];
		while (col < row) {
			lhs = smmFma(-value, x[col], lhs);
			++k;
			col = positions[k];
			value = vals[k];
		}
		x[row] = lhs / value;
	} else if (MODE == SGS_UP) {  // ref:1698-1711
		int k = e - 1;
		int col = positions[k];
		T value = vals[k];
		T lhs = T(0);
		while (col > row) {
			lhs = smmFma(value, x[col], lhs);
			--k;
			col = positions[k];
			value = vals[k];
		}
		x[row] = x[row] - lhs / value;
	} else if (MODE == ILU_LO) {  // L y = rhs, unit diagonal
		T sum = rhs[row];
		for (int k = b; k < e && positions[k] < row; ++k) {
			sum = smmFma(-vals[k], x[positions[k]], sum);
		}
		x[row] = sum;
	} else if (MODE == ILU_UP) {  // U x = y
		T sum = x[row];
		int k = e - 1;
		for (; k >= b && positions[k] > row; --k) {
			sum = smmFma(-vals[k], x[positions[k]], sum);
		}
		x[row] = sum / vals[k];
	} else if (MODE == IC_LO) {  // ref:1806-1819: sum -= l*x (plain multiply and subtract)
		T sum = rhs[row];
		int k = b;
		int col = positions[k];
		while (col < row && k < e) {
			const T prod = vals[k] * x[col];
			sum = sum - prod;
			++k;
			col = positions[k];
		}
		x[row] = sum / vals[k];
	} else {  // IC_UP, ref:1822-1835
		T sum = x[row];
		int k = e - 1;
		int col = positions[k];
		while (col > row && k >= b) {
			const T prod = vals[k] * x[col];
			sum = sum - prod;
			--k;
			col = positions[k];
		}
		x[row] = sum / vals[k];
	}
}

// one large level: rows order[begin .. begin+count)
template <typename T, int MODE>
__global__ __launch_bounds__(TPB) void sweepLevelKernel(const int* __restrict__ order, int begin, int count, const int* __restrict__ start,
                                                        const int* __restrict__ positions, const T* __restrict__ vals, const T* __restrict__ rhs,
                                                        T* x, const int* __restrict__ doneFlag) {
	if (doneFlag && *doneFlag) return;
	const int i = blockIdx.x * TPB + threadIdx.x;
	if (i < count) {
		solveRow<T, MODE>(order[begin + i], start, positions, vals, rhs, x);
	}
}

// a run of small levels [l0, l1) inside one workgroup; lvlPtr is the device copy of the level pointers
template <typename T, int MODE>
__global__ __launch_bounds__(SMALL_LEVEL) void sweepChainKernel(const int* __restrict__ order, const int* __restrict__ lvlPtr, int l0, int l1,
                                                                const int* __restrict__ start, const int* __restrict__ positions,
                                                                const T* __restrict__ vals, const T* __restrict__ rhs, T* x,
                                                                const int* __restrict__ doneFlag) {
	if (doneFlag && *doneFlag) return;
	for (int l = l0; l < l1; ++l) {
		const int begin = lvlPtr[l];
		const int count = lvlPtr[l + 1] - begin;
		if (static_cast<int>(threadIdx.x) < count) {
			solveRow<T, MODE>(order[begin + threadIdx.x], start, positions, vals, rhs, x);
		}
		// rows of the next level read x[] written above by other wavefronts of this workgroup
		__threadfence_block();
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------------------------------------
// Synchronisation-free sweeps: ONE launch per sweep instead of one per level.
//
// The output vector is pre-filled with a sentinel bit pattern (a NaN no arithmetic produces).  A wavefront draws the next
// 64 rows of the level-sorted order from a ticket counter; each lane walks its row's entries in the reference's order and,
// for an entry whose x[col] still holds the sentinel, stops and polls again on the next pass of the wave-wide loop; when its
// row is finished it publishes x[row] with ONE store that is value and ready flag at once.  Polls and publishes are relaxed
// agent-scope atomics (global_load / global_store with sc1): coherent across the 8 XCDs' L2s without any cache writeback or
// invalidate, the pattern of a decoupled look-back scan.  Every row reads only rows that come earlier in the sorted order,
// and tickets are handed out in that order to wavefronts that are running, so the wavefront holding the earliest unfinished
// rows can always finish them: no deadlock whatever the residency.  Lanes of one wavefront that depend on each other resolve
// over successive passes of the loop (nobody waits inside a pass).  A pass counter bounds the loop regardless: on overrun the
// row publishes NaN and raises the error word the host checks at the end of the solve.
// Arithmetic per row is exactly solveRow's, so results are bit-identical to the level-scheduled and the sequential sweeps.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct SweepBits;
template <>
struct SweepBits<float> {
	using U = unsigned int;
	static constexpr U SENT = 0x7FD5EAD5u;
	static constexpr U QNAN = 0x7FC00000u;
};
template <>
struct SweepBits<double> {
	using U = unsigned long long;
	static constexpr U SENT = 0x7FFD5EAD5EAD5EADull;
	static constexpr U QNAN = 0x7FF8000000000000ull;
};

// Agent scope (sc1) is the narrowest scope that is coherent between CUs: workgroup-scope (sc0) loads may hit the CU's own
// vector cache and never see another CU's store (tried: the sweep runs into its pass limit).  Confining a sweep to the
// wavefronts of one XCD does not make sc1 traffic cheaper either (tried: 108^3 stencil 2.4 ms instead of 1.2 ms per apply).
template <typename T>
__device__ __forceinline__ typename SweepBits<T>::U pollBits(const T* x, int col) {
	using U = typename SweepBits<T>::U;
	return __hip_atomic_load(reinterpret_cast<const U*>(x) + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ONE_XCD: every wavefront of the sweep runs on the same XCD, whose L2 is then the meeting point: the publishing store is a plain
// (workgroup-scope) store that KEEPS the line in that L2 instead of the sc1 store that pushes it out to the memory side, and the
// agent-scope poll (sc1: bypasses the CU's own L1 only) finds it there -- a hop costs an L2 round trip instead of two fabric trips.
template <typename T, bool ONE_XCD = false>
__device__ __forceinline__ void publishX(T* x, int row, T v) {
	using U = typename SweepBits<T>::U;
	U b;
	__builtin_memcpy(&b, &v, sizeof(T));
	if (b == SweepBits<T>::SENT) b = SweepBits<T>::QNAN;  // a NaN either way; never leave a finished row looking unfinished
	if (ONE_XCD) __hip_atomic_store(reinterpret_cast<U*>(x) + row, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	else __hip_atomic_store(reinterpret_cast<U*>(x) + row, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// HW_REG_XCC_ID (id 20), bits [3:0]: the XCD this wavefront runs on
__device__ __forceinline__ int xccId() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xF; }


template <typename T>
__global__ __launch_bounds__(TPB) void sweepPrefillKernel(int n, T* __restrict__ y, T* __restrict__ x, int* __restrict__ words,
                                                          const int* __restrict__ doneFlag) {
	if (doneFlag && *doneFlag) return;
	using U = typename SweepBits<T>::U;
	if (blockIdx.x == 0 && threadIdx.x < 2) words[threadIdx.x] = 0;  // the two ticket counters; words[2] (error) is sticky
	if (blockIdx.x == 0 && threadIdx.x >= 3 && threadIdx.x < 5) words[threadIdx.x] = -1;  // XCD elected by each sweep (ONE_XCD)
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * TPB) {
		reinterpret_cast<U*>(y)[i] = SweepBits<T>::SENT;
		reinterpret_cast<U*>(x)[i] = SweepBits<T>::SENT;
	}
}

// lower sweeps: in = rhs, out = y.  upper sweeps: in = y (the lower sweep's result), out = x.
//
// What is on the critical path of a sweep is the time from "the last row I read has been published" to "my row is
// published"; every dependent memory round trip in between (~1 us each) is paid once per LEVEL.  So a lane keeps a window of
// the next SWEEP_WINDOW entries of its row (column and value) in registers, polls ALL entries of the window that are still
// missing with independent loads in one go, and then consumes them strictly in the row's order as far as they are final:
// one round trip (the poll) + arithmetic + the publishing store per level.
constexpr int SWEEP_WINDOW = 8;

template <typename T, int MODE, bool ONE_XCD>
__global__ __launch_bounds__(TPB) void sweepFreeKernel(int n, const int* __restrict__ order, const int* __restrict__ start,
                                                       const int* __restrict__ positions, const T* __restrict__ vals, const T* __restrict__ in,
                                                       T* out, int* ticket, int* err, const int* __restrict__ doneFlag, unsigned passLimit,
                                                       int* elected) {
	if (doneFlag && *doneFlag) return;
	if (ONE_XCD) {
		// the first wavefront to arrive elects its XCD; wavefronts elsewhere leave at once.  Any number (>= 1) of participants
		// finishes the sweep: rows are drawn from the ticket counter in dependency order by whoever is running.
		int mine = xccId(), chosen = 0;
		if ((threadIdx.x & (WAVE - 1)) == 0) {
			const int prev = atomicCAS(elected, -1, mine);
			chosen = prev == -1 ? mine : prev;
		}
		chosen = __builtin_amdgcn_readfirstlane(chosen);
		if (chosen != mine) return;
	}
	constexpr bool LOWER = MODE == SGS_LO || MODE == ILU_LO || MODE == IC_LO;
	constexpr int W = SWEEP_WINDOW;
	constexpr int DIR = LOWER ? 1 : -1;
	using U = typename SweepBits<T>::U;
	const int lane = threadIdx.x & (WAVE - 1);
	for (;;) {
		int chunk = 0;
		if (lane == 0) chunk = atomicAdd(ticket, 1);
		chunk = __builtin_amdgcn_readfirstlane(chunk);
		const long long base = static_cast<long long>(chunk) * WAVE;
		if (base >= n) return;
		bool pending = base + lane < n;
		int row = 0, k = 0, stop = 0;
		T acc = T(0), own = T(0);
		if (pending) {
			row = order[base + lane];
			const int b = start[row];
			const int e = start[row + 1];
			own = in[row];
			if (LOWER) {
				k = b;
				stop = e;
				acc = own;
			} else {
				k = e - 1;
				stop = b - 1;
				acc = MODE == SGS_UP ? T(0) : own;
			}
		}
		int wc[W];     // columns of the window; `row` marks the end of the row's triangular part
		T wv[W];       // values of the window
		int used = W;  // entries of the window already consumed (W: load the next window)
		unsigned passes = 0;
		while (__ballot(pending) != 0ull) {
			bool moved = false;
			if (pending) {
				if (used == W) {
#pragma unroll
					for (int u = 0; u < W; ++u) {
						const int idx = k + DIR * u;
						const bool inside = LOWER ? idx < stop : idx > stop;
						wc[u] = inside ? positions[idx] : row;
						wv[u] = inside ? vals[idx] : T(1);
					}
					used = 0;
				}
				// poll every entry of the window that is still needed: independent loads, one round trip for all of them
				U xb[W];
				bool ended = false;
#pragma unroll
				for (int u = 0; u < W; ++u) {
					ended = ended || (LOWER ? wc[u] >= row : wc[u] <= row);
					xb[u] = SweepBits<T>::SENT;
					if (u >= used && !ended) {
						xb[u] = pollBits<T>(out, wc[u]);
					}
				}
				// consume in the row's order as far as the values are final
				bool stopped = false;
#pragma unroll
				for (int u = 0; u < W; ++u) {
					if (u >= used && !stopped && pending) {
						const int col = wc[u];
						const T value = wv[u];
						if (LOWER ? col >= row : col <= row) {
							// the diagonal (or, for ILU's unit-lower part only, the end of the row)
							T result;
							if (MODE == SGS_LO) result = acc / value;             // ref:1695
							else if (MODE == SGS_UP) result = own - acc / value;  // ref:1710
							else if (MODE == ILU_LO) result = acc;
							else result = acc / value;                            // ILU_UP, IC_LO (ref:1818), IC_UP (ref:1834)
							publishX<T, ONE_XCD>(out, row, result);
							pending = false;
							moved = true;
						} else if (xb[u] != SweepBits<T>::SENT) {
							T xv;
							__builtin_memcpy(&xv, &xb[u], sizeof(T));
							if (MODE == SGS_LO) acc = smmFma(-value, xv, acc);
							else if (MODE == SGS_UP) acc = smmFma(value, xv, acc);
							else if (MODE == ILU_LO || MODE == ILU_UP) acc = smmFma(-value, xv, acc);
							else {
								const T prod = value * xv;  // ref:1812-1813, 1828-1829: multiply, then subtract
								acc = acc - prod;
							}
							used = u + 1;
							moved = true;
						} else {
							stopped = true;  // not final yet: poll again on the next pass
						}
					}
				}
				if (pending && used == W) k += DIR * W;
			}
			if (__ballot(moved) != 0ull) passes = 0;  // the bound counts passes without progress of any lane (pure waiting), like levelFreeKernel's
			if (++passes > passLimit && pending) {  // cannot happen; guarantees that the grid drains
				atomicOr(err, 1);
				publishX<T, ONE_XCD>(out, row, __builtin_nanf(""));
				pending = false;
			}
		}
	}
}

template <typename T>
__global__ __launch_bounds__(TPB) void jacobiApplyKernel(int n, const T* __restrict__ diag, const T* __restrict__ rhs, T* __restrict__ x,
                                                         const int* __restrict__ doneFlag) {
	if (doneFlag && *doneFlag) return;
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * TPB) {
		x[i] = rhs[i] / diag[i];
	}
}

// diag[i] = A_ii; err[0] is set when a diagonal is missing or |d| < 1e-5 (the SGS validity rule, ref:1691-1693)
template <typename T>
__global__ __launch_bounds__(TPB) void extractDiagKernel(int rows, const int* __restrict__ start, const int* __restrict__ positions,
                                                         const T* __restrict__ vals, T* __restrict__ diag, int* __restrict__ err) {
	for (long long row = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; row < rows; row += static_cast<long long>(gridDim.x) * TPB) {
		bool found = false;
		T d = T(0);
		for (int k = start[row]; k < start[row + 1]; ++k) {
			if (positions[k] == row) {
				d = vals[k];
				found = true;
				break;
			}
		}
		diag[row] = d;
		if (!found || (d < T(0) ? -d : d) < T(1e-5)) atomicOr(err, 1);
	}
}

// ---------------------------------------------------------------------------------------------------------
// Set-up on the device: structural checks, level sets of both sweeps, ILU(0) / IC(0) factorisation.  Nothing of the matrix is
// copied to the host; what comes back are a few words (error flags, the number of levels, the longest row) and the level pointers.
// ---------------------------------------------------------------------------------------------------------
// every row non-empty with its diagonal stored; with checkMagnitude also |d| >= 1e-5 (ref:1666-1693).  info[0] |= 1 on a violation,
// info[1] = longest row
template <typename T>
__global__ __launch_bounds__(TPB) void rowCheckKernel(int n, const int* __restrict__ start, const int* __restrict__ positions, const T* __restrict__ vals,
                                                      int checkMagnitude, int* info) {
	int longest = 0;
	bool bad = false;
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * TPB) {
		const int b = start[i], e = start[i + 1];
		longest = max(longest, e - b);
		bool found = false;
		for (int k = b; k < e; ++k) {
			if (positions[k] == i) {
				const T v = vals[k];
				found = !checkMagnitude || (v < T(0) ? -v : v) >= T(1e-5);
				break;
			}
		}
		bad = bad || !found;
	}
#pragma unroll
	for (int o = WAVE / 2; o > 0; o >>= 1) longest = max(longest, __shfl_xor(longest, o, WAVE));
	if ((threadIdx.x & (WAVE - 1)) == 0 && longest > 0) atomicMax(info + 1, longest);
	if (bad) atomicOr(info, 1);
}

// Level sets of the strictly-lower (LOWER) or strictly-upper part of the pattern: level[i] = 1 + the largest level among the rows that
// row i reads (0 if it reads none) -- the longest path to i in the dependency DAG.  One launch: rows are taken in the natural order
// (descending for the upper part) from a ticket counter, a lane walks its row's triangular entries and, where the level of a column is
// not known yet (-1), stops and polls again on the next pass of the wave-wide loop; a finished row publishes its level with one store
// that is value and ready flag at once -- the scheme of the synchronisation-free sweeps below, with the same freedom from deadlock
// (a row reads only rows that drew earlier tickets; nobody waits inside a pass).  words: [0] ticket, [2] error.
template <bool LOWER>
__global__ __launch_bounds__(TPB) void levelFreeKernel(int n, const int* __restrict__ start, const int* __restrict__ positions, int* level, int* words,
                                                       unsigned passLimit) {
	const int lane = threadIdx.x & (WAVE - 1);
	constexpr int DIR = LOWER ? 1 : -1;
	for (;;) {
		int chunk = 0;
		if (lane == 0) chunk = atomicAdd(words, 1);
		chunk = __builtin_amdgcn_readfirstlane(chunk);
		const long long base = static_cast<long long>(chunk) * WAVE;
		if (base >= n) return;
		bool pending = base + lane < n;
		int row = 0, k = 0, stop = 0, lv = 0;
		if (pending) {
			row = LOWER ? static_cast<int>(base + lane) : n - 1 - static_cast<int>(base + lane);
			k = LOWER ? start[row] : start[row + 1] - 1;
			stop = LOWER ? start[row + 1] : start[row] - 1;
		}
		unsigned passes = 0;  // passes since a lane of this wavefront last moved on
		while (__ballot(pending) != 0ull) {
			bool moved = false;
			if (pending) {
				// one entry per pass (the shape of the sweeps' loop: nobody waits, or loops, inside a pass)
				const bool inside = LOWER ? k < stop : k > stop;
				const int col = inside ? positions[k] : row;
				if (!(inside && (LOWER ? col < row : col > row))) {  // end of the triangular part (ref:1673-1694 stop at the diagonal too)
					__hip_atomic_store(level + row, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					pending = false;
					moved = true;
				} else {
					const int d = __hip_atomic_load(level + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (d >= 0) {  // known: take it and move on; otherwise poll again on the next pass
						lv = max(lv, d + 1);
						k += DIR;
						moved = true;
					}
				}
			}
			// The escape bound counts passes WITHOUT progress: a wavefront that consumes an entry or finishes a row starts afresh, so what
			// is bounded is the pure wait for the rows in front of it -- at most the chunks of the other resident wavefronts, each a
			// bounded number of passes (buildLevelsDevice sizes passLimit for exactly that) -- not the depth of the matrix's DAG.
			if (__ballot(moved) != 0ull) passes = 0;
			if (++passes > passLimit && pending) {  // cannot happen; guarantees that the grid drains
				atomicOr(words + 2, 1);
				__hip_atomic_store(level + row, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				pending = false;
			}
		}
		// (the largest level is found by a kernel of its own, maxLevelKernel: a wave reduction + "if (lane == 0) atomicMax" at this
		// place made hipcc 7.2 emit a loop in which lanes 1 .. 63 re-enter the chunk forever)
	}
}

__global__ __launch_bounds__(TPB) void maxLevelKernel(int n, const int* __restrict__ level, int* out) {
	int top = 0;
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * TPB) top = max(top, level[i]);
#pragma unroll
	for (int o = WAVE / 2; o > 0; o >>= 1) top = max(top, __shfl_xor(top, o, WAVE));
	if ((threadIdx.x & (WAVE - 1)) == 0) atomicMax(out, top);
}

__global__ __launch_bounds__(TPB) void iotaKernel(int n, int* out) {
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * TPB) out[i] = static_cast<int>(i);
}

// sortedLevel[] ascending: lvlPtr[l] = first index of level l (every level 0 .. nLevels-1 holds at least one row), lvlPtr[nLevels] = n
__global__ __launch_bounds__(TPB) void levelPtrKernel(int n, int nLevels, const int* __restrict__ sortedLevel, int* __restrict__ lvlPtr) {
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * TPB) {
		if (i == 0 || sortedLevel[i] != sortedLevel[i - 1]) lvlPtr[sortedLevel[i]] = static_cast<int>(i);
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) lvlPtr[nLevels] = n;
}

// position of column c among positions[b .. e) (ascending), or -1
__device__ __forceinline__ int findColumn(const int* __restrict__ positions, int b, int e, int c) {
	int lo = b, hi = e;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (positions[mid] < c) lo = mid + 1;
		else hi = mid;
	}
	return lo < e && positions[lo] == c ? lo : -1;
}

// ILU(0), IKJ ordering on A's pattern (Saad, Iterative Methods for Sparse Linear Systems, Alg. 10.4), one row by one WAVEFRONT: unit-lower
// L and U share one value array.  l_ik = a_ik * (1/u_kk) as ref:1767 intends.  The elimination steps of a row (its lower entries k,
// ascending) are sequential; inside a step the lanes share the entries of row k's U part -- every entry of the row is touched at most
// once per step, so the result does not depend on how the step is shared out: the same bits as the sequential loop.  The rows k are final
// (earlier levels).  A missing / tiny pivot raises *err (the create call then fails: reordering would be needed, ref:1741-1746).
template <typename T>
__device__ __forceinline__ void iluRowWave(int row, const int* __restrict__ start, const int* __restrict__ positions, T* lu, T* pivotInv, int* err) {
	const int lane = threadIdx.x & (WAVE - 1);
	const int rb = start[row], re = start[row + 1];
	int q = rb;
	for (; q < re; ++q) {
		const int k = positions[q];
		if (k >= row) break;
		const T lik = lu[q] * pivotInv[k];
		const int ke = start[k + 1];
		const int dk = findColumn(positions, start[k], ke, k);  // the diagonal of row k (rowCheckKernel made sure of it)
		for (int u = dk + 1 + lane; u < ke; u += WAVE) {
			const int target = findColumn(positions, rb, re, positions[u]);
			if (target != -1) lu[target] = lu[target] - lik * lu[u];
		}
		if (lane == 0) lu[q] = lik;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's updates are in place before the next step reads lu[q + 1]
	}
	const T piv = q < re ? lu[q] : T(0);
	const bool ok = q < re && positions[q] == row && (piv < T(0) ? -piv : piv) >= T(1e-6);
	if (lane == 0) {
		pivotInv[row] = ok ? T(1.0) / piv : T(0);
		if (!ok) atomicOr(err, 1);
	}
}

// IC(0): A ~ L L^T with L's values stored at A's lower positions and mirrored into the upper ones -- ref:1839-1928 computes it column
// by column; this is the same arithmetic row by row (up-looking), one row by one wavefront: l_ji = (a_ji - sum_{k<i} l_ik l_jk) / d_i
// with the sum in ascending k over row j's entries, d_j = sqrt(a_jj - sum_k l_jk^2) in ascending k.  The lanes share the look-ups of a
// sum (64 entries of row j at a time) and the products are then added one after the other in entry order: the operands and the order of
// every sum are the reference's, so are the bits (tests: assert_array_equal against the reference's factor).  Row i (< j) is final when
// row j starts (earlier level).  *err: bit 0 not positive definite on the pattern (ref:1871-1878) or diagonal missing, bit 1 pattern not
// symmetric.
template <typename T>
__device__ __forceinline__ void ic0RowWave(int j, const int* __restrict__ start, const int* __restrict__ positions, const T* __restrict__ a, T* ic, T* dinv,
                                           int* err) {
	const int lane = threadIdx.x & (WAVE - 1);
	const int jb = start[j], je = start[j + 1];
	T acc = T(0);
	int k = jb;
	for (; k < je; ++k) {
		const int i = positions[k];
		if (i >= j) break;
		const int ib = start[i], ie = start[i + 1];
		const int mirror = findColumn(positions, ib, ie, j);
		if (mirror == -1) {  // row i does not hold column j: not a symmetric pattern
			if (lane == 0) atomicOr(err, 2);
			continue;
		}
		T sum = T(0);
		for (int kk0 = jb; kk0 < k; kk0 += WAVE) {
			const int kk = kk0 + lane;
			const int mine = kk < k ? findColumn(positions, ib, ie, positions[kk]) : -1;
			const T prod = mine != -1 ? ic[mine] * ic[kk] : T(0);
			unsigned long long has = __ballot(mine != -1);
			while (has != 0ull) {  // in entry order
				const int t = __builtin_ctzll(has);
				sum += __shfl(prod, t, WAVE);
				has &= has - 1ull;
			}
		}
		const T lji = (a[k] - sum) * dinv[i];
		if (lane == 0) {
			ic[k] = lji;
			ic[mirror] = lji;
		}
		acc += lji * lji;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // l_ji is in place before the next entry's sum reads it
	}
	if (k >= je || positions[k] != j) {
		if (lane == 0) {
			atomicOr(err, 1);
			dinv[j] = T(0);
		}
		return;
	}
	const T under = a[k] - acc;
	const T d = sqrt(under);
	if (lane == 0) {
		if (!(under > T(0))) atomicOr(err, 1);
		ic[k] = d;
		dinv[j] = T(1) / d;
	}
}

// KIND: SMM_PRECOND_ILU0 or SMM_PRECOND_IC0.  One large level / a run of small levels inside one workgroup, like the sweeps; one
// wavefront per row (a row of a dense block costs O(length^2) look-ups: one lane per row took minutes on tests/test_gpu_misc.py's
// 700-row dense block).
template <typename T, int KIND>
__device__ __forceinline__ void factorRowWave(int row, const int* __restrict__ start, const int* __restrict__ positions, const T* __restrict__ a, T* f, T* piv,
                                              int* err) {
	if (KIND == SMM_PRECOND_ILU0) iluRowWave<T>(row, start, positions, f, piv, err);
	else ic0RowWave<T>(row, start, positions, a, f, piv, err);
}

template <typename T, int KIND>
__global__ __launch_bounds__(TPB) void factorLevelKernel(const int* __restrict__ order, int begin, int count, const int* __restrict__ start,
                                                         const int* __restrict__ positions, const T* __restrict__ a, T* f, T* piv, int* err) {
	const int r = blockIdx.x * (TPB / WAVE) + (threadIdx.x >> 6);
	if (r < count) factorRowWave<T, KIND>(order[begin + r], start, positions, a, f, piv, err);
}

template <typename T, int KIND>
__global__ __launch_bounds__(SMALL_LEVEL) void factorChainKernel(const int* __restrict__ order, const int* __restrict__ lvlPtr, int l0, int l1,
                                                                 const int* __restrict__ start, const int* __restrict__ positions, const T* __restrict__ a,
                                                                 T* f, T* piv, int* err) {
	for (int l = l0; l < l1; ++l) {
		const int begin = lvlPtr[l];
		const int count = lvlPtr[l + 1] - begin;
		for (int r = threadIdx.x >> 6; r < count; r += SMALL_LEVEL / WAVE) factorRowWave<T, KIND>(order[begin + r], start, positions, a, f, piv, err);
		__threadfence_block();
		__syncthreads();
	}
}

struct SweepPlan {
	// launch groups: chains of small levels, or single large levels
	struct Group {
		int l0, l1;
		bool chain;
	};
	std::vector<Group> groups;
	int* d_lvlPtr = nullptr;
};

static void planGroups(const std::vector<int>& lvlPtr, std::vector<SweepPlan::Group>& groups) {
	const int nl = static_cast<int>(lvlPtr.size()) - 1;
	int l = 0;
	while (l < nl) {
		const int sz = lvlPtr[l + 1] - lvlPtr[l];
		if (sz > SMALL_LEVEL) {
			groups.push_back({l, l + 1, false});
			++l;
		} else {
			int e = l + 1;
			while (e < nl && lvlPtr[e + 1] - lvlPtr[e] <= SMALL_LEVEL) ++e;
			groups.push_back({l, e, true});
			l = e;
		}
	}
}

}  // namespace smm

// the level pointers and launch groups ride along behind the public handle
struct smm_precond_plan {
	smm::SweepPlan lo, up;
	int sweep = SMM_SWEEP_AUTO;
	// Escape bound of the synchronisation-free sweeps, in polling passes of a waiting wavefront.  A waiter polls about as fast as a
	// worker consumes one window of SWEEP_WINDOW entries, so a row that must wait for a row of L entries needs ~L / SWEEP_WINDOW
	// passes per dependency: the bound grows with the longest row of the matrix and can only trip on a genuine hang.
	unsigned passLimit = 1u << 21;
	// scratch of the synchronisation-free sweeps, one set per stream the preconditioner is applied on: y[n] (result of the
	// lower sweep) and three words {ticket of the lower sweep, ticket of the upper sweep, sticky error}
	struct Scratch {
		void* y = nullptr;
		int* words = nullptr;
	};
	std::mutex scratchMutex;
	std::map<hipStream_t, Scratch> scratch;
	// one apply = prefill + lower sweep + upper sweep on the same scratch: the three launches of an apply are enqueued under this
	// lock, so applies issued by several host threads on one stream stay whole (the stream then runs them one after the other)
	std::mutex enqueueMutex;
};

namespace smm {

static smm_precond_plan* planOf(const smm_hip_precond* M) { return M->plan; }

template <typename T, int MODE>
static int runSweep(const smm_hip_precond* M, const SweepPlan& plan, const int* d_order, const std::vector<int>& lvlPtr, const T* vals, const T* rhs,
                    T* x, const int* doneFlag, hipStream_t s) {
	const smm_hip_csr* a = M->a;
	for (const auto& g : plan.groups) {
		if (g.chain) {
			sweepChainKernel<T, MODE><<<1, SMALL_LEVEL, 0, s>>>(d_order, plan.d_lvlPtr, g.l0, g.l1, a->d_start, a->d_positions, vals, rhs, x, doneFlag);
		} else {
			const int begin = lvlPtr[g.l0];
			const int count = lvlPtr[g.l1] - begin;
			sweepLevelKernel<T, MODE><<<(count + TPB - 1) / TPB, TPB, 0, s>>>(d_order, begin, count, a->d_start, a->d_positions, vals, rhs, x, doneFlag);
		}
	}
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

static int scratchFor(const smm_hip_precond* M, size_t elemBytes, hipStream_t s, smm_precond_plan::Scratch* out) {
	smm_precond_plan* plan = M->plan;
	std::lock_guard<std::mutex> lock(plan->scratchMutex);
	auto it = plan->scratch.find(s);
	if (it == plan->scratch.end()) {
		if (plan->scratch.size() >= 8) {
			// a caller that keeps creating streams would otherwise pin one n-sized buffer per stream for ever
			SMM_HIP_TRY(hipDeviceSynchronize());
			for (auto& kv : plan->scratch) {
				devFree(kv.second.y);
				devFree(kv.second.words);
			}
			plan->scratch.clear();
		}
		smm_precond_plan::Scratch sc;
		SMM_TRY(devAlloc(&sc.y, static_cast<size_t>(std::max(1, M->a->rows)) * elemBytes));
		SMM_TRY(devAlloc(reinterpret_cast<void**>(&sc.words), 8 * sizeof(int)));
		SMM_HIP_TRY(hipMemsetAsync(sc.words, 0, 8 * sizeof(int), s));
		it = plan->scratch.emplace(s, sc).first;
	}
	*out = it->second;
	return SMM_HIP_OK;
}

// Wavefronts of a sweep launch.  Only the wavefronts holding rows of the level that is currently being finished (and the next
// one or two) can make progress; every further wavefront only polls, and polling traffic slows the ones on the critical path
// (measured: 108^3 stencil 1.8 / 2.7 / 4.7 ms per apply with 1024 / 2048 / 4096 wavefronts).  So the launch is sized to a few
// levels' worth of rows, in one-wavefront workgroups so that they spread over the CUs.
static int sweepWaves(int n, size_t levels, bool oneXcd = false) {
	double perLevel = oneXcd ? 3.0 : 4.0;
	if (const char* env = getenv("SMM_HIP_SWEEP_WAVES_PER_LEVEL")) perLevel = std::max(0.25, atof(env));
	const double rowsPerLevel = static_cast<double>(n) / static_cast<double>(std::max<size_t>(1, levels));
	long long waves = static_cast<long long>(perLevel * rowsPerLevel / WAVE) + 1;
	if (const char* env = getenv("SMM_HIP_SWEEP_WAVES")) waves = std::max(1, atoi(env));
	const long long all = (static_cast<long long>(n) + WAVE - 1) / WAVE;
	const long long cap = oneXcd ? static_cast<long long>(numCUs() / 8) * 16 : static_cast<long long>(numCUs()) * 8;  // resident wavefronts
	return static_cast<int>(std::max<long long>(1, std::min<long long>(std::min<long long>(waves, all), cap)));
}

template <typename T, int LO, int UP, bool ONE_XCD>
static int runSweepsFree(const smm_hip_precond* M, const T* vals, const T* rhs, T* x, const int* doneFlag, hipStream_t s) {
	const smm_hip_csr* a = M->a;
	const int n = a->rows;
	smm_precond_plan::Scratch sc;
	SMM_TRY(scratchFor(M, sizeof(T), s, &sc));
	T* y = static_cast<T*>(sc.y);
	const int fill = static_cast<int>(std::min<long long>((n + TPB - 1LL) / TPB, numCUs() * 8LL));
	// ONE_XCD: workgroups are dealt round-robin over the 8 XCDs, so 8 x as many are launched and the elected eighth stays
	const int spread = ONE_XCD ? 8 : 1;
	const int wLo = spread * sweepWaves(n, M->lvl_ptr_lo.size() - 1, ONE_XCD);
	const int wUp = spread * sweepWaves(n, M->lvl_ptr_up.size() - 1, ONE_XCD);
	const unsigned limit = M->plan->passLimit;
	std::lock_guard<std::mutex> whole(M->plan->enqueueMutex);
	sweepPrefillKernel<T><<<fill, TPB, 0, s>>>(n, y, x, sc.words, doneFlag);
	sweepFreeKernel<T, LO, ONE_XCD><<<wLo, WAVE, 0, s>>>(n, M->d_order_lo, a->d_start, a->d_positions, vals, rhs, y, sc.words, sc.words + 2, doneFlag, limit, sc.words + 3);
	sweepFreeKernel<T, UP, ONE_XCD><<<wUp, WAVE, 0, s>>>(n, M->d_order_up, a->d_start, a->d_positions, vals, y, x, sc.words + 1, sc.words + 2, doneFlag, limit, sc.words + 4);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

// the sticky error word of the synchronisation-free sweeps run on `s` (call after the stream has been synchronised)
int precondTakeError(const smm_hip_precond* M, hipStream_t s) {
	if (!M || !M->plan) return SMM_HIP_OK;
	smm_precond_plan* plan = M->plan;
	int* words = nullptr;
	{
		std::lock_guard<std::mutex> lock(plan->scratchMutex);
		auto it = plan->scratch.find(s);
		if (it == plan->scratch.end()) return SMM_HIP_OK;
		words = it->second.words;
	}
	int err = 0;
	SMM_HIP_TRY(hipMemcpyAsync(&err, words + 2, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	if (err) {
		SMM_HIP_TRY(hipMemsetAsync(words + 2, 0, sizeof(int), s));
		setError("preconditioner: a synchronisation-free triangular sweep ran into its pass limit");
		return SMM_HIP_ERR_HIP;
	}
	return SMM_HIP_OK;
}

// AUTO: the synchronisation-free sweeps; confined to one XCD when the levels are small (an average level of <= 2048 rows keeps the 32
// CUs of one XCD busy, and a hop through that XCD's L2 is cheaper than one through the fabric: 2-D Poisson 1000^2 4.99 -> 4.18 ms per
// apply), spread over the chip when they are large (3-D 108^3: 8.7 K-row levels need the bandwidth of all XCDs, 1.16 vs 1.86 ms)
static int sweepModeOf(const smm_hip_precond* M) {
	const smm_precond_plan* plan = M->plan;
	int mode = plan->sweep;
	if (mode == SMM_SWEEP_AUTO) {
		const size_t levels = std::max<size_t>(1, std::min(M->lvl_ptr_lo.size(), M->lvl_ptr_up.size()) - 1);
		mode = static_cast<size_t>(M->a->rows) / levels <= 2048 ? SMM_SWEEP_SYNCFREE_XCD : SMM_SWEEP_SYNCFREE;
		if (const char* env = getenv("SMM_HIP_SWEEP")) mode = atoi(env);
	}
	return mode;
}

template <typename T>
int precondApplyDev(const smm_hip_precond* M, const T* rhs, T* x, const int* doneFlag, hipStream_t s) {
	if (!M || M->dtype != dtypeOf<T>()) {
		setError("precond_apply: null handle or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	const int n = M->a->rows;
	if (n == 0) return SMM_HIP_OK;
	if (!rhs || !x || rhs == x) {  // assert(rhs != x), ref:1667
		setError("precond_apply: null vector or rhs aliases x");
		return SMM_HIP_ERR_INVALID;
	}
	if (isBlockKind(M->kind)) return blockApplyDev<T>(M, rhs, x, 0, nullptr, nullptr, doneFlag, s);
	if (M->kind == SMM_PRECOND_JACOBI) {
		const int grid = static_cast<int>(std::min<long long>((n + TPB - 1LL) / TPB, numCUs() * 8LL));
		jacobiApplyKernel<T><<<grid, TPB, 0, s>>>(n, static_cast<const T*>(M->d_values), rhs, x, doneFlag);
		SMM_HIP_TRY(hipGetLastError());
		return SMM_HIP_OK;
	}
	const smm_precond_plan* plan = planOf(M);
	if (!plan) {
		setError("precond_apply: preconditioner has no sweep plan");
		return SMM_HIP_ERR_INVALID;
	}
	const T* aVals = static_cast<const T*>(M->a->d_values);
	const T* fVals = static_cast<const T*>(M->d_values);
	const int mode = sweepModeOf(M);
	if (mode == SMM_SWEEP_SYNCFREE_XCD) {
		switch (M->kind) {
		case SMM_PRECOND_SGS: return runSweepsFree<T, SGS_LO, SGS_UP, true>(M, aVals, rhs, x, doneFlag, s);
		case SMM_PRECOND_ILU0: return runSweepsFree<T, ILU_LO, ILU_UP, true>(M, fVals, rhs, x, doneFlag, s);
		case SMM_PRECOND_IC0: return runSweepsFree<T, IC_LO, IC_UP, true>(M, fVals, rhs, x, doneFlag, s);
		default: break;
		}
	} else if (mode != SMM_SWEEP_LEVELS) {
		switch (M->kind) {
		case SMM_PRECOND_SGS: return runSweepsFree<T, SGS_LO, SGS_UP, false>(M, aVals, rhs, x, doneFlag, s);
		case SMM_PRECOND_ILU0: return runSweepsFree<T, ILU_LO, ILU_UP, false>(M, fVals, rhs, x, doneFlag, s);
		case SMM_PRECOND_IC0: return runSweepsFree<T, IC_LO, IC_UP, false>(M, fVals, rhs, x, doneFlag, s);
		default: break;
		}
	}
	switch (M->kind) {
	case SMM_PRECOND_SGS:
		SMM_TRY((runSweep<T, SGS_LO>(M, plan->lo, M->d_order_lo, M->lvl_ptr_lo, aVals, rhs, x, doneFlag, s)));
		SMM_TRY((runSweep<T, SGS_UP>(M, plan->up, M->d_order_up, M->lvl_ptr_up, aVals, rhs, x, doneFlag, s)));
		return SMM_HIP_OK;
	case SMM_PRECOND_ILU0:
		SMM_TRY((runSweep<T, ILU_LO>(M, plan->lo, M->d_order_lo, M->lvl_ptr_lo, fVals, rhs, x, doneFlag, s)));
		SMM_TRY((runSweep<T, ILU_UP>(M, plan->up, M->d_order_up, M->lvl_ptr_up, fVals, rhs, x, doneFlag, s)));
		return SMM_HIP_OK;
	case SMM_PRECOND_IC0:
		SMM_TRY((runSweep<T, IC_LO>(M, plan->lo, M->d_order_lo, M->lvl_ptr_lo, fVals, rhs, x, doneFlag, s)));
		SMM_TRY((runSweep<T, IC_UP>(M, plan->up, M->d_order_up, M->lvl_ptr_up, fVals, rhs, x, doneFlag, s)));
		return SMM_HIP_OK;
	default:
		setError("precond_apply: kind %d has no apply", M->kind);
		return SMM_HIP_ERR_INVALID;
	}
}

template int precondApplyDev<float>(const smm_hip_precond*, const float*, float*, const int*, hipStream_t);
template int precondApplyDev<double>(const smm_hip_precond*, const double*, double*, const int*, hipStream_t);

// level sets of one sweep, on the device: d_order (rows sorted by level, ascending row inside a level), the level pointers on the host
// (the launch plan is made from them) and on the device (the chained kernels read them)
template <bool LOWER>
static int buildLevelsDevice(const smm_hip_csr* a, unsigned passLimit, int maxRowLen, hipStream_t s, int** d_order, std::vector<int>& lvlPtr, int** d_lvlPtr) {
	const int n = a->rows;
	*d_order = nullptr;
	*d_lvlPtr = nullptr;
	lvlPtr.assign(1, 0);
	SMM_TRY(devAlloc(reinterpret_cast<void**>(d_order), static_cast<size_t>(std::max(1, n)) * sizeof(int)));
	if (n == 0) {
		SMM_TRY(devAlloc(reinterpret_cast<void**>(d_lvlPtr), sizeof(int)));
		SMM_HIP_TRY(hipMemsetAsync(*d_lvlPtr, 0, sizeof(int), s));
		return SMM_HIP_OK;
	}
	DevBuf<int> level, sorted, rowsIn, words;
	SMM_TRY(level.alloc(n));
	SMM_TRY(sorted.alloc(n));
	SMM_TRY(rowsIn.alloc(n));
	SMM_TRY(words.alloc(4));
	SMM_HIP_TRY(hipMemsetAsync(level, 0xFF, static_cast<size_t>(n) * sizeof(int), s));  // -1: not known yet
	SMM_HIP_TRY(hipMemsetAsync(words, 0, 4 * sizeof(int), s));
	// Resident wavefronts only (8 per CU), and the escape bound follows from their number: the wavefront that drew the last of the
	// first `waves` tickets may have to wait for every chunk in front of it, each of which needs at most 64 rows x (longest row + 2)
	// passes once ITS predecessors are done (a pure chain -- a tridiagonal matrix, a dense band -- is the worst case; the r02 bound,
	// 2^21 + 64 x longest row for up to 8192 wavefronts, left a factor 2 for a tridiagonal matrix and none for a 50-entry band).
	const int waves = std::min((n + WAVE - 1) / WAVE, numCUs() * 8);
	const unsigned long long perChunk = 64ull * (static_cast<unsigned long long>(maxRowLen) + 2ull);
	const unsigned levelLimit = static_cast<unsigned>(std::min<unsigned long long>(0x7fffffffull, (1ull << 21) + 8ull * static_cast<unsigned long long>(waves) * perChunk));
	(void)passLimit;
	levelFreeKernel<LOWER><<<(waves + TPB / WAVE - 1) / (TPB / WAVE), TPB, 0, s>>>(n, a->d_start, a->d_positions, level, words, levelLimit);
	const int grid = static_cast<int>(std::min<long long>((n + TPB - 1LL) / TPB, numCUs() * 8LL));
	maxLevelKernel<<<grid, TPB, 0, s>>>(n, level, words.p + 1);
	iotaKernel<<<grid, TPB, 0, s>>>(n, rowsIn);
	int h[4] = {0, 0, 0, 0};
	SMM_HIP_TRY(hipMemcpyAsync(h, words, sizeof(h), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	if (h[2]) {
		setError("preconditioner: the level analysis ran into its pass limit");
		return SMM_HIP_ERR_HIP;
	}
	const int nLevels = h[1] + 1;
	int bits = 1;
	while ((1ll << bits) <= h[1]) ++bits;
	size_t tempBytes = 0;
	SMM_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tempBytes, level.p, sorted.p, rowsIn.p, *d_order, static_cast<size_t>(n), 0, bits, s));
	DevBuf<char> temp;
	SMM_TRY(temp.alloc(tempBytes ? tempBytes : 1));
	SMM_HIP_TRY(rocprim::radix_sort_pairs(temp.p, tempBytes, level.p, sorted.p, rowsIn.p, *d_order, static_cast<size_t>(n), 0, bits, s));
	SMM_TRY(devAlloc(reinterpret_cast<void**>(d_lvlPtr), (static_cast<size_t>(nLevels) + 1) * sizeof(int)));
	levelPtrKernel<<<grid, TPB, 0, s>>>(n, nLevels, sorted, *d_lvlPtr);
	lvlPtr.resize(static_cast<size_t>(nLevels) + 1);
	SMM_HIP_TRY(hipMemcpyAsync(lvlPtr.data(), *d_lvlPtr, lvlPtr.size() * sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));  // the scratch buffers go back to the allocator when this scope ends
	return SMM_HIP_OK;
}

// ILU(0) / IC(0) on the device, level by level in the order of the lower sweep (a row is eliminated with rows of earlier levels only)
template <typename T, int KIND>
static int factorDevice(const smm_hip_csr* a, const SweepPlan& plan, const int* d_order, const std::vector<int>& lvlPtr, T* f, int* herr, hipStream_t s) {
	const int n = a->rows;
	DevBuf<T> piv;
	DevBuf<int> err;
	SMM_TRY(piv.alloc(std::max(1, n)));
	SMM_TRY(err.alloc(1));
	SMM_HIP_TRY(hipMemsetAsync(err, 0, sizeof(int), s));
	const T* aVals = static_cast<const T*>(a->d_values);
	for (const auto& g : plan.groups) {
		if (g.chain) {
			factorChainKernel<T, KIND><<<1, SMALL_LEVEL, 0, s>>>(d_order, plan.d_lvlPtr, g.l0, g.l1, a->d_start, a->d_positions, aVals, f, piv.p, err.p);
		} else {
			const int begin = lvlPtr[g.l0];
			const int count = lvlPtr[g.l0 + 1] - begin;
			factorLevelKernel<T, KIND><<<(count + TPB / WAVE - 1) / (TPB / WAVE), TPB, 0, s>>>(d_order, begin, count, a->d_start, a->d_positions, aVals, f, piv.p, err.p);
		}
	}
	SMM_HIP_TRY(hipGetLastError());
	SMM_HIP_TRY(hipMemcpyAsync(herr, err, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

template <typename T>
static int createTyped(const smm_hip_csr* a, int kind, smm_hip_precond* M) {
	hipStream_t s = libStream();
	SMM_HIP_TRY(hipDeviceSynchronize());  // the matrix may still be being written on a caller's stream
	const int n = a->rows;
	if (kind == SMM_PRECOND_JACOBI) {
		SMM_TRY(devAlloc(&M->d_values, static_cast<size_t>(std::max(1, n)) * sizeof(T)));
		M->n_values = static_cast<size_t>(n);
		DevBuf<int> err;
		SMM_TRY(err.alloc(1));
		SMM_HIP_TRY(hipMemsetAsync(err, 0, sizeof(int), s));
		if (n) {
			const int grid = static_cast<int>(std::min<long long>((n + TPB - 1LL) / TPB, numCUs() * 8LL));
			extractDiagKernel<T><<<grid, TPB, 0, s>>>(n, a->d_start, a->d_positions, static_cast<const T*>(a->d_values), static_cast<T*>(M->d_values), err);
		}
		int herr = 0;
		SMM_HIP_TRY(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
		if (herr) {
			setError("jacobi: missing or |d|<1e-5 diagonal entry");
			return SMM_HIP_ERR_PRECOND;
		}
		return SMM_HIP_OK;
	}
	// SGS / ILU0 / IC0: analysis (and factorisation) on the device
	if (a->rows != a->cols) {
		setError("preconditioner needs a square matrix");
		return SMM_HIP_ERR_INVALID;
	}
	if (a->firstActiveStart != 0 && n > 0) {  // ref:1666-1670, 1737-1739
		setError("preconditioner: matrix has leading empty rows (firstActiveStart != 0)");
		return SMM_HIP_ERR_PRECOND;
	}
	int info[2] = {0, 0};  // structural error, longest row
	{
		DevBuf<int> dinfo;
		SMM_TRY(dinfo.alloc(2));
		SMM_HIP_TRY(hipMemsetAsync(dinfo, 0, 2 * sizeof(int), s));
		if (n) {
			const int grid = static_cast<int>(std::min<long long>((n + TPB - 1LL) / TPB, numCUs() * 8LL));
			rowCheckKernel<T><<<grid, TPB, 0, s>>>(n, a->d_start, a->d_positions, static_cast<const T*>(a->d_values), kind == SMM_PRECOND_SGS ? 1 : 0, dinfo);
		}
		SMM_HIP_TRY(hipMemcpyAsync(info, dinfo, sizeof(info), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
	}
	if (info[0]) {
		setError("preconditioner: empty row, missing diagonal or |d|<1e-5");
		return SMM_HIP_ERR_PRECOND;
	}
	auto* plan = new smm_precond_plan();
	M->plan = plan;  // (smm_hip_precond_destroy releases it together with whatever the steps below have allocated so far)
	plan->passLimit = static_cast<unsigned>(std::min<long long>(0x7fffffffLL, (1LL << 21) + 64LL * info[1]));
	SMM_TRY(buildLevelsDevice<true>(a, plan->passLimit, info[1], s, &M->d_order_lo, M->lvl_ptr_lo, &plan->lo.d_lvlPtr));
	SMM_TRY(buildLevelsDevice<false>(a, plan->passLimit, info[1], s, &M->d_order_up, M->lvl_ptr_up, &plan->up.d_lvlPtr));
	planGroups(M->lvl_ptr_lo, plan->lo.groups);
	planGroups(M->lvl_ptr_up, plan->up.groups);
	if (kind == SMM_PRECOND_ILU0 || kind == SMM_PRECOND_IC0) {
		const size_t nnz = static_cast<size_t>(a->nnz);
		SMM_TRY(devAlloc(&M->d_values, std::max<size_t>(1, nnz) * sizeof(T)));
		M->n_values = nnz;
		int herr = 0;
		if (kind == SMM_PRECOND_ILU0) {
			if (nnz) SMM_HIP_TRY(hipMemcpyAsync(M->d_values, a->d_values, nnz * sizeof(T), hipMemcpyDeviceToDevice, s));
			SMM_TRY((factorDevice<T, SMM_PRECOND_ILU0>(a, plan->lo, M->d_order_lo, M->lvl_ptr_lo, static_cast<T*>(M->d_values), &herr, s)));
			if (herr) {
				setError("ilu0: zero / missing pivot (reordering would be needed, ref:1741-1746)");
				return SMM_HIP_ERR_PRECOND;
			}
		} else {
			if (nnz) SMM_HIP_TRY(hipMemsetAsync(M->d_values, 0, nnz * sizeof(T), s));
			SMM_TRY((factorDevice<T, SMM_PRECOND_IC0>(a, plan->lo, M->d_order_lo, M->lvl_ptr_lo, static_cast<T*>(M->d_values), &herr, s)));
			if (herr) {
				setError(herr & 1 ? "ic0: matrix is not symmetric positive definite on its pattern (ref:1871-1878)"
				                  : "ic0: the pattern of the matrix is not symmetric");
				return SMM_HIP_ERR_PRECOND;
			}
		}
	}
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

template <typename T>
static int applyHost(const smm_hip_precond* M, const T* rhs, T* x) {
	if (!M) {
		setError("precond_apply: null handle");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	const int n = M->a->rows;
	if (n > 0 && (!rhs || !x || rhs == x)) {
		setError("precond_apply: null vector or rhs aliases x");
		return SMM_HIP_ERR_INVALID;
	}
	hipStream_t s = libStream();
	DevBuf<T> dr, dx;
	SMM_TRY(dr.alloc(n));
	SMM_TRY(dx.alloc(n));
	SMM_TRY(hostToDev(dr, rhs, sizeof(T) * n, s));
	SMM_TRY(precondApplyDev<T>(M, dr, dx, nullptr, s));
	SMM_TRY(devToHost(x, dx, sizeof(T) * n, s));
	return precondTakeError(M, s);
}

// x = M^-1 (A v), A = the matrix M was created for: what a preconditioned Krylov loop asks for twice per pass (ref:2234-2235,
// 2250-2251).  The block kinds form A v inside the apply's launch (smm_precond_block.hip); the others run the SpMV and then the apply.
template <typename T>
static int applySpmvDev(const smm_hip_precond* M, const T* v, T* x, hipStream_t s) {
	if (!M || M->dtype != dtypeOf<T>()) {
		setError("precond_apply_spmv: null handle / dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	const int n = M->a->rows;
	if (n > 0 && (!v || !x || v == x)) {
		setError("precond_apply_spmv: null vector or v aliases x");
		return SMM_HIP_ERR_INVALID;
	}
	if (n == 0 || M->kind == SMM_PRECOND_NONE) return launchSpmv<T>(M->a, SMM_OP_ASSIGN, nullptr, v, x, 0, nullptr, nullptr, nullptr, s);
	SMM_TRY(ensureCsrReady(M->a, s, true));
	if (isBlockKind(M->kind) && M->a->nnz > 0 && blockFuseSpmv(M, true)) return blockApplySpmvDev<T>(M, v, x, 0, nullptr, nullptr, nullptr, s);
	DevBuf<T> t;
	SMM_TRY(t.alloc(n));
	SMM_TRY(launchSpmv<T>(M->a, SMM_OP_ASSIGN, nullptr, v, t, 0, nullptr, nullptr, nullptr, s));
	return precondApplyDev<T>(M, t, x, nullptr, s);
}

template <typename T>
static int applySpmvHost(const smm_hip_precond* M, const T* v, T* x) {
	if (!M) {
		setError("precond_apply_spmv: null handle");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	const int n = M->a->rows;
	if (n > 0 && (!v || !x || v == x)) {
		setError("precond_apply_spmv: null vector or v aliases x");
		return SMM_HIP_ERR_INVALID;
	}
	hipStream_t s = libStream();
	DevBuf<T> dv, dx;
	SMM_TRY(dv.alloc(n));
	SMM_TRY(dx.alloc(n));
	SMM_TRY(hostToDev(dv, v, sizeof(T) * n, s));
	SMM_TRY(applySpmvDev<T>(M, dv, dx, s));
	SMM_TRY(devToHost(x, dx, sizeof(T) * n, s));
	return precondTakeError(M, s);
}

template <typename T>
static int valuesHost(const smm_hip_precond* M, T* out, size_t count) {
	if (!M || M->dtype != dtypeOf<T>() || !out) {
		setError("precond_values: null handle / dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	if (!M->d_values || count > M->n_values) {
		setError("precond_values: this preconditioner holds %zu values", M->n_values);
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	hipStream_t s = libStream();
	if (count) SMM_HIP_TRY(hipMemcpyAsync(out, M->d_values, count * sizeof(T), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

}  // namespace smm

using namespace smm;

extern "C" {

static int precondCreate(const smm_hip_csr* a, int kind, int blockRows, int levelCap, int partition, smm_hip_precond** out) {
	if (!a || !out) {
		setError("precond_create: null argument");
		return SMM_HIP_ERR_INVALID;
	}
	*out = nullptr;
	if (kind < SMM_PRECOND_NONE || kind > SMM_PRECOND_BLOCK_SGS) {
		setError("precond_create: unknown kind %d", kind);
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	SMM_TRY(ensureCsrReady(a, nullptr, false));  // a set-up call without a stream: drains the device once
	auto* M = new smm_hip_precond();
	M->kind = kind;
	M->dtype = a->dtype;
	M->a = a;
	int st = SMM_HIP_OK;
	if (isBlockKind(kind)) {
		st = a->dtype == SMM_DTYPE_F32 ? blockCreateTyped<float>(a, kind, blockRows, levelCap, partition, M) : blockCreateTyped<double>(a, kind, blockRows, levelCap, partition, M);
	} else if (kind != SMM_PRECOND_NONE) {
		st = a->dtype == SMM_DTYPE_F32 ? createTyped<float>(a, kind, M) : createTyped<double>(a, kind, M);
	}
	if (st != SMM_HIP_OK) {
		smm_hip_precond_destroy(M);
		return st;
	}
	*out = M;
	return SMM_HIP_OK;
}

int smm_hip_precond_create(const smm_hip_csr* a, int kind, smm_hip_precond** out) { return precondCreate(a, kind, blockDefaultRows(), blockDefaultLevelCap(), SMM_BLOCKS_AUTO, out); }

int smm_hip_precond_create_block(const smm_hip_csr* a, int kind, int block_rows, smm_hip_precond** out) {
	if (!isBlockKind(kind)) {
		setError("precond_create_block: kind %d is not a block preconditioner", kind);
		return SMM_HIP_ERR_INVALID;
	}
	if (block_rows != 0 && (block_rows < 64 || block_rows > 2048)) {
		setError("precond_create_block: block_rows must be 0 (default) or 64 .. 2048");
		return SMM_HIP_ERR_INVALID;
	}
	return precondCreate(a, kind, block_rows ? block_rows : blockDefaultRows(), blockDefaultLevelCap(), SMM_BLOCKS_AUTO, out);
}

static int createBlockChecked(const char* who, const smm_hip_csr* a, int kind, int block_rows, int level_cap, int partition, smm_hip_precond** out) {
	if (!isBlockKind(kind)) {
		setError("%s: kind %d is not a block preconditioner", who, kind);
		return SMM_HIP_ERR_INVALID;
	}
	if (block_rows != 0 && (block_rows < 64 || block_rows > 2048)) {
		setError("%s: block_rows must be 0 (default) or 64 .. 2048", who);
		return SMM_HIP_ERR_INVALID;
	}
	if (level_cap < -1 || level_cap == 1 || level_cap > 4095) {
		setError("%s: level_cap must be -1 (default), 0 (no cut) or 2 .. 4095", who);
		return SMM_HIP_ERR_INVALID;
	}
	if (partition < SMM_BLOCKS_AUTO || partition > SMM_BLOCKS_BRICKS) {
		setError("%s: partition must be SMM_BLOCKS_AUTO, _CONTIGUOUS or _BRICKS", who);
		return SMM_HIP_ERR_INVALID;
	}
	return precondCreate(a, kind, block_rows ? block_rows : blockDefaultRows(), level_cap < 0 ? blockDefaultLevelCap() : level_cap, partition, out);
}

int smm_hip_precond_create_block_capped(const smm_hip_csr* a, int kind, int block_rows, int level_cap, smm_hip_precond** out) {
	return createBlockChecked("precond_create_block_capped", a, kind, block_rows, level_cap, SMM_BLOCKS_AUTO, out);
}

int smm_hip_precond_create_block_ex(const smm_hip_csr* a, int kind, int block_rows, int level_cap, int partition, smm_hip_precond** out) {
	return createBlockChecked("precond_create_block_ex", a, kind, block_rows, level_cap, partition, out);
}

int smm_hip_precond_destroy(smm_hip_precond* M) {
	if (!M) return SMM_HIP_OK;
	blockDestroy(M->blk);
	smm_precond_plan* plan = M->plan;
	if (plan) {
		devFree(plan->lo.d_lvlPtr);
		devFree(plan->up.d_lvlPtr);
		for (auto& kv : plan->scratch) {
			devFree(kv.second.y);
			devFree(kv.second.words);
		}
		delete plan;
	}
	devFree(M->d_values);
	devFree(M->d_order_lo);
	devFree(M->d_order_up);
	delete M;
	return SMM_HIP_OK;
}

int smm_hip_precond_info(const smm_hip_precond* M, int* kind, int* levels_lower, int* levels_upper) {
	if (!M) {
		setError("precond_info: null handle");
		return SMM_HIP_ERR_INVALID;
	}
	if (kind) *kind = M->kind;
	if (M->blk) {  // block kinds: the deepest block
		blockLevels(M->blk, levels_lower, levels_upper);
		return SMM_HIP_OK;
	}
	if (levels_lower) *levels_lower = M->lvl_ptr_lo.empty() ? 0 : static_cast<int>(M->lvl_ptr_lo.size()) - 1;
	if (levels_upper) *levels_upper = M->lvl_ptr_up.empty() ? 0 : static_cast<int>(M->lvl_ptr_up.size()) - 1;
	return SMM_HIP_OK;
}

int smm_hip_precond_set_sweep(smm_hip_precond* M, int mode) {
	if (!M || mode < SMM_SWEEP_AUTO || mode > SMM_SWEEP_SYNCFREE_XCD) {
		setError("precond_set_sweep: null handle or unknown mode");
		return SMM_HIP_ERR_INVALID;
	}
	if (M->plan) M->plan->sweep = mode;  // Jacobi / NONE have no sweeps: accepted, no effect
	return SMM_HIP_OK;
}

int smm_hip_precond_take_error(const smm_hip_precond* M, smm_hip_stream stream) {
	if (!M) {
		setError("precond_take_error: null handle");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return precondTakeError(M, pickStream(stream));
}

int smm_hip_precond_apply_f32(const smm_hip_precond* M, const float* rhs, float* x) { return applyHost<float>(M, rhs, x); }
int smm_hip_precond_apply_f64(const smm_hip_precond* M, const double* rhs, double* x) { return applyHost<double>(M, rhs, x); }
int smm_hip_precond_apply_dev_f32(const smm_hip_precond* M, const float* d_rhs, float* d_x, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	return precondApplyDev<float>(M, d_rhs, d_x, nullptr, pickStream(stream));
}
int smm_hip_precond_apply_dev_f64(const smm_hip_precond* M, const double* d_rhs, double* d_x, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	return precondApplyDev<double>(M, d_rhs, d_x, nullptr, pickStream(stream));
}
int smm_hip_precond_apply_spmv_f32(const smm_hip_precond* M, const float* v, float* x) { return applySpmvHost<float>(M, v, x); }
int smm_hip_precond_apply_spmv_f64(const smm_hip_precond* M, const double* v, double* x) { return applySpmvHost<double>(M, v, x); }
int smm_hip_precond_apply_spmv_dev_f32(const smm_hip_precond* M, const float* d_v, float* d_x, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	return applySpmvDev<float>(M, d_v, d_x, pickStream(stream));
}
int smm_hip_precond_apply_spmv_dev_f64(const smm_hip_precond* M, const double* d_v, double* d_x, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	return applySpmvDev<double>(M, d_v, d_x, pickStream(stream));
}
int smm_hip_precond_values_f32(const smm_hip_precond* M, float* out, size_t count) { return valuesHost<float>(M, out, count); }
int smm_hip_precond_values_f64(const smm_hip_precond* M, double* out, size_t count) { return valuesHost<double>(M, out, count); }

}  // extern "C"
