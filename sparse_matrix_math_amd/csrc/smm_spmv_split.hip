// smm_spmv_split.hip -- the row-partitioned SpMV in ONE launch (round 6; the loop it serves: ref:2232-2277).
//
// A rank's rows are split into A_loc (owned columns) and A_rem (halo columns), smm_dist.hip.  r01-r05 ran them as two launches: A_loc while
// the halo is in flight, then -- behind an event of the communicator's stream -- A_rem, which re-reads and re-writes out[].  The second
// launch costs a second ramp and tail of a persistent grid, a pass over out[] and a dependent kernel boundary behind the exchange: ~10 us
// per SpMV of a rank's 1.25 M-row share, two SpMVs per BiCGStab iteration, where the whole iteration has ~190 us at 8 GPUs.
//
// Here ONE persistent grid does both halves.  The rows are cut into SUPER TILES of 256 consecutive rows (no table: row r belongs to super
// tile r / 256), dealt to the workgroups once -- XCD group g owns a contiguous eighth, its workgroups interleave inside it, as in the tile
// kernels (r06, second form: in chunks of 64 super tiles dealt round-robin to the groups -- SplitMap).  A workgroup
//   phase 1: walks the A_loc part of each of its K super tiles and keeps the row sums in LDS (K x 256 values: a few KB);
//   waits  : one lane polls the word the communicator's stream raises once the halo of THIS exchange has landed in `ext`
//            (splitSignalKernel behind the land kernel / the grouped receive; bounded: an expired wait leaves a code in the error word);
//   phase 2: walks the A_rem part of the same super tiles -- every gather a cache-bypassing (sc1) load: the halo was written by another
//            kernel while this one ran -- and writes out[row] = op(lhs, loc) (+|-) rem [/ diag] ONCE, with the solver's dot products of
//            the finished vector in the epilogue.
// No workgroup ever reads what another one wrote: no grid barrier, no seam, no second pass over out[].
//
// Arithmetic.  Each half is walked exactly as spmvPatternTileKernel walks a matrix at the half's own lanes per row (1, 2 or 4 pieces per
// row, added left to right), and the two halves meet as the second launch met the first: out = op(lhs, loc); out = out +|- rem
// [then / diag].  Bit for bit the two-launch form (tests/test_gpu_dist_native.py).
//
// Applies when both blocks are in the PATTERN family's row-mask encoding with values read (what the solvers adopt from 2^20 entries) at
// 1, 2 or 4 lanes per row and the K x 256 row sums fit the LDS budget; otherwise the caller runs the two launches.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

#include "smm_device.h"
#include "smm_internal.h"
#include "smm_p2p.h"
#include "smm_pattern_dev.h"

namespace smm {

extern __shared__ __attribute__((aligned(16))) unsigned char smmSplitLds[];

constexpr int SUPER = TPB;  // rows per super tile

// Which super tiles a workgroup walks.  Workgroups b, b + 8, ... share an XCD (group g = b % 8).  Two deals, chosen per pair of blocks:
//   chunk == 0: group g owns the contiguous eighth [g * perGroup, (g + 1) * perGroup) -- neighbouring tiles share x lines, one L2 serves them:
//               the deal of the tile kernels, best whenever the work is spread evenly over the rows (a rank of eight: 198 against 222 us per iteration);
//   chunk > 0 : the tiles are dealt in chunks of `chunk` consecutive ones, chunk c to group c % 8 -- a rank of two or four has its remote
//               entries in the rows near one end, and contiguous eighths leave that half of the work to three of the eight groups (795 against 758 us).
// Inside a group the workgroups interleave.
constexpr int SPLIT_CHUNK = 64;
struct SplitMap {
	int nSupers, nGroups, group, slot, groupSlots, chunk, perGroup;
	__device__ __forceinline__ int superOf(int k) const {  // the k-th super tile of this workgroup; >= nSupers: none left
		const int i = slot + k * groupSlots;  // index in the group's list of tiles
		if (chunk == 0) return i < perGroup ? group * perGroup + i : nSupers;
		const int c = i / chunk;
		return (c * nGroups + group) * chunk + (i - c * chunk);
	}
};

template <typename T>
struct SplitSide {
	const int* offs;
	const int* start;
	const unsigned long long* masks;
	const int* positions;
	const T* values;
	const T* x;  // the vector this half gathers from: the owned part for A_loc (columns numbered from the first owned one), the halo-extended one for A_rem
	int cols, nOff, cap, batch, stageLimit;
};

template <typename T>
struct SplitArgs {
	SplitSide<T> a, b;
	int rows, nSupers, chunk, perGroup, K;
	int sumsInLds;  // 1: the local half's row sums stay in LDS (K x 256 values); 0: K is too large for that (few ranks, many rows) and they travel
	                // through out[] -- written as the two-launch form's first launch wrote it, read back by the same workgroup
	int opFlags, dotMode;
	const T* lhs;
	const T* divisor;
	T* out;
	const T* w1;
	T* partials;
	const int* doneFlag;
	const unsigned long long* landed;  // raised to >= seq once the halo of this exchange is in x (null: it already is)
	unsigned long long seq;
	unsigned long long* err;
	long long ticks;
	unsigned long long* waited;  // workgroup 0 adds the ticks (100 MHz) it spent waiting for the word: what the exchange cost beyond the local half
	P2PSlotArgs slots;  // world != 0 (peer-to-peer scalars, dotMode != 0, SPMV_FINISH): the last workgroup runs the reduction point itself
};

template <typename T>
__device__ __forceinline__ T splitGather(const T* __restrict__ x, unsigned byteOffset, bool bypass) {
	T* p = reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<T*>(x)) + byteOffset);
	// (relaxed, agent scope = global_load ... sc1: served by the L2, never by this CU's L1, which may hold a line of the halo from before it landed)
	return bypass ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

// One half of a workgroup's K super tiles: its K x L sub tiles of 256 / L rows in sequence, each staged through LDS and walked like a tile
// of spmvPatternTileKernel<T, L, G> -- and SOFTWARE-PIPELINED: sub tile j + 1's slice of values[], its masks and row starts are requested
// into registers as soon as sub tile j's have been stored to LDS, so they travel while j is walked.  Measured on one rank's share of the
// benchmark matrix (profiles/r06/split_spmv_ablation.txt): without it either half took ~47 us whatever its bytes (152 MB or 96 MB): a
// sub tile is a chain of 4-5 dependent trips to memory (row bounds, values, two gather batches) and a CU holds five workgroups.
// done(row, index in the workgroup's row sums, dot) is called by the lane that holds a row's sum.
template <typename T, int L, int G, typename Done>
__device__ __forceinline__ void splitPhase(const SplitSide<T>& M, int rows, const SplitMap& map, int K, bool REMOTE, T* sVal, unsigned long long* sMask,
                                           int* sStart, const int* sOff, T* sPart, Done&& done) {
	const T* __restrict__ x = M.x;
	using Cfg = PatCfg<T>;
	constexpr int NVP = sizeof(T) == 4 ? 6 : 4;  // passes (one 16-byte load per lane, fp64: two) requested ahead: 24 / 32 VGPRs
	constexpr int GROUPS = TPB / WAVE / L;
	constexpr int RT = 64 * GROUPS;
	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int wave = t >> 6;
	const int piece = wave % L;
	const int rl = (wave / L) * 64 + lane;
	const int nv = (M.cap + Cfg::PIECE - 1) / Cfg::PIECE;
	// the request set of ONE sub tile (registers are re-issued for j + 1 right after j's have been stored)
	typename Pack16<T>::V rv[NVP * (sizeof(T) == 4 ? 1 : 2)];
	int ps = 0;
	unsigned long long pm = 0ULL;
	int r0 = 0, nrows = 0, n0 = 0, n1 = 0, a0 = 0;
	bool direct = false, valid = false;
	auto subTile = [&](int j, int& rr0) {  // first row of sub tile j of this workgroup, false past its last one
		const int k = j / L, q = j - k * L;
		const int sp = map.superOf(k);
		rr0 = sp * SUPER + q * RT;
		return k < K && sp < map.nSupers && rr0 < rows;
	};
	auto request = [&](int j) {
		valid = subTile(j, r0);
		if (!valid) return;
		const int r1 = min(rows, r0 + RT);
		nrows = r1 - r0;
		n0 = M.start[r0];
		n1 = M.start[r1];
		a0 = n0 & ~3;
		direct = n1 - n0 > M.cap - 3 || a0 > M.stageLimit;
		if (direct || n1 == n0) return;  // (nothing is staged for a sub tile without entries: the interior rows of a rank have no remote part)
		ps = 0;
		pm = 0ULL;
		if (t < nrows) {
			ps = M.start[r0 + t];
			pm = M.masks[r0 + t];
		}
#pragma unroll
		for (int v = 0; v < NVP; ++v) {
			const int i = a0 + 4 * (t + v * TPB);
			if (v < nv && i < n1) {
				if constexpr (sizeof(T) == 4) {
					rv[v] = __builtin_nontemporal_load(reinterpret_cast<const pf32x4*>(M.values + i));
				} else {
					rv[2 * v] = __builtin_nontemporal_load(reinterpret_cast<const pf64x2*>(M.values + i));
					rv[2 * v + 1] = __builtin_nontemporal_load(reinterpret_cast<const pf64x2*>(M.values + i + 2));
				}
			}
		}
	};
	request(0);
#pragma unroll 1
	for (int j = 0; valid; ++j) {  // (`valid` is what request() found for sub tile j)
		// this sub tile's geometry (the request set is about to be re-issued)
		const int cr0 = r0, cnrows = nrows, cn1 = n1, ca0 = a0;
		const bool cdirect = direct;
		const bool cempty = n1 == n0;
		const int base = (j / L) * SUPER + (j % L) * RT;  // where this sub tile's rows sit among the workgroup's row sums
		if (cempty) {
			// no entry in these rows of this half: every row's sum is +0 (what its empty pieces add up to) -- no staging, no barrier
			request(j + 1);
			for (int rr = t; rr < cnrows; rr += TPB) done(cr0 + rr, base + rr, T(0));
			continue;
		}
		if (!cdirect) {
#pragma unroll
			for (int v = 0; v < NVP; ++v) {
				const int li = 4 * (t + v * TPB);
				if (v < nv && ca0 + li < cn1) {
					if constexpr (sizeof(T) == 4) {
						*reinterpret_cast<pf32x4*>(sVal + li) = rv[v];
					} else {
						*reinterpret_cast<pf64x2*>(sVal + li) = rv[2 * v];
						*reinterpret_cast<pf64x2*>(sVal + li + 2) = rv[2 * v + 1];
					}
				}
			}
			// tiles of more passes than are requested ahead (long rows at one lane per row): the rest now
			for (int v = NVP; v < nv; ++v) {
				const int li = 4 * (t + v * TPB);
				if (ca0 + li < cn1) {
					if constexpr (sizeof(T) == 4) {
						*reinterpret_cast<pf32x4*>(sVal + li) = __builtin_nontemporal_load(reinterpret_cast<const pf32x4*>(M.values + ca0 + li));
					} else {
						*reinterpret_cast<pf64x2*>(sVal + li) = __builtin_nontemporal_load(reinterpret_cast<const pf64x2*>(M.values + ca0 + li));
						*reinterpret_cast<pf64x2*>(sVal + li + 2) = __builtin_nontemporal_load(reinterpret_cast<const pf64x2*>(M.values + ca0 + li + 2));
					}
				}
			}
			if (t < cnrows) {
				sStart[t] = ps - ca0;
				sMask[t] = pm;
			}
			if (t == 0) sStart[cnrows] = cn1 - ca0;
		}
		ldsBarrier();
		request(j + 1);  // (travels while this sub tile is walked)
		if (cdirect) {
			// rows longer than a tile and the last tiles of the matrix: one lane per row straight from HBM, in the staged path's pieces
			for (int rr = t; rr < cnrows; rr += TPB) {
				const int row = cr0 + rr;
				const int b = M.start[row], e = M.start[row + 1];
				const int piecelen = (e - b + L - 1) / L;
				T tot = T(0);
#pragma unroll
				for (int pq = 0; pq < L; ++pq) {
					const int kb = b + pq * piecelen, ke = min(e, kb + piecelen);
					T dot = T(0);
					for (int k = kb; k < ke; ++k) {
						dot = smmFma(M.values[k], splitGather<T>(x, static_cast<unsigned>(M.positions[k]) * static_cast<unsigned>(sizeof(T)), REMOTE), dot);
					}
					tot = pq == 0 ? dot : tot + dot;
				}
				done(row, base + rr, tot);
			}
		} else {
			T dot = T(0);
			const int row = cr0 + rl;
			int kb = 0, ke = 0;
			unsigned long long mm = 0ULL;
			if (rl < cnrows) {
				const int b = sStart[rl];
				const int e = sStart[rl + 1];
				const int piecelen = (e - b + L - 1) / L;
				kb = b + piece * piecelen;
				ke = min(e, kb + piecelen);
				mm = sMask[rl];
			}
			const unsigned long long rowMask = mm;
			// wavefronts whose 64 rows all hold the SAME offsets (everywhere but where a diagonal enters or leaves the rank's column range):
			// spmvPatternTileKernel's fast path
			const unsigned long long lead = patUniform64(rowMask);
			if (__builtin_amdgcn_ballot_w64(!(rl < cnrows && rowMask == lead)) == 0ULL) {
				const int len = __builtin_popcountll(lead);
				const int pl = (len + L - 1) / L;
				const int pu = __builtin_amdgcn_readfirstlane(piece);
				const int e0 = pu * pl, cnt = max(0, min(len, e0 + pl) - e0);
				// lane u of the wavefront looks up the offset of the piece's u-th entry ONCE per tile (a piece has at most 64 entries); the loop below
				// then reads it with v_readlane at a wave-uniform index: no per-entry bit scan at all
				const int offPiece = sOff[lane < cnt ? selectBit(lead, e0 + lane) : 0];
				const unsigned rowBytes = static_cast<unsigned>(row) * static_cast<unsigned>(sizeof(T));
				for (int e = 0; e < cnt; e += G) {
					T xv[G], vv[G];
					const int nvalid = cnt - e;  // (wave-uniform)
#pragma unroll
					for (int u = 0; u < G; ++u) {
						// entries past the end of the piece repeat its last (valid) column; their products are discarded
						xv[u] = splitGather<T>(x + __builtin_amdgcn_readlane(offPiece, min(e + u, cnt - 1)), rowBytes, REMOTE);
						vv[u] = sVal[kb + e + u];
					}
#pragma unroll
					for (int u = 0; u < G; ++u) {
						if (u < nvalid) dot = smmFma(vv[u], xv[u], dot);
					}
				}
			} else {
				// this piece starts at the (kb - b)-th entry of the row = the (kb - b)-th set bit of the mask
				if (rl < cnrows && piece > 0 && kb < ke) mm &= ~0ULL << selectBit(mm, kb - sStart[rl]);
				for (int k = kb; k < ke; k += G) {
					unsigned off[G];
					T xv[G], vv[G];
#pragma unroll
					for (int u = 0; u < G; ++u) {
						const int jj = mm ? __builtin_ctzll(mm) : 0;
						mm &= mm - 1;
						// entries past the end of the piece get a clamped, valid column; their products are discarded
						const int col = min(max(row + sOff[jj], 0), M.cols - 1);
						off[u] = static_cast<unsigned>(col) * static_cast<unsigned>(sizeof(T));
						vv[u] = sVal[k + u];
					}
#pragma unroll
					for (int u = 0; u < G; ++u) xv[u] = splitGather<T>(x, off[u], REMOTE);
					const int nvalid = ke - k;
#pragma unroll
					for (int u = 0; u < G; ++u) {
						const T next = smmFma(vv[u], xv[u], dot);
						dot = u < nvalid ? next : dot;
					}
				}
			}
			if constexpr (L > 1) {
				// pieces of a row meet in LDS and are added left to right: ((p0 + p1) + p2) + p3
				if (piece > 0 && rl < cnrows) sPart[(piece - 1) * RT + rl] = dot;
				ldsBarrier();
				if (piece == 0 && rl < cnrows) {
#pragma unroll
					for (int pq = 1; pq < L; ++pq) dot += sPart[(pq - 1) * RT + rl];
					done(row, base + rl, dot);
				}
			} else {
				if (rl < cnrows) done(row, base + rl, dot);
			}
		}
		ldsBarrier();
	}
}

// the gather batch is a run-time property of a block (fitted to its rows, patBatch): three compiled sizes per half
template <typename T, int L, typename Done>
__device__ __forceinline__ void splitPhaseG(const SplitSide<T>& M, int rows, const SplitMap& map, int K, bool REMOTE, T* sVal, unsigned long long* sMask,
                                            int* sStart, const int* sOff, T* sPart, Done&& done) {
	if (M.batch <= 8) splitPhase<T, L, 8>(M, rows, map, K, REMOTE, sVal, sMask, sStart, sOff, sPart, done);
	else if (M.batch <= 13) splitPhase<T, L, 13>(M, rows, map, K, REMOTE, sVal, sMask, sStart, sOff, sPart, done);
	else splitPhase<T, L, 16>(M, rows, map, K, REMOTE, sVal, sMask, sStart, sOff, sPart, done);
}

// what the second launch of the two-launch form did with the first one's out[row]
template <typename T>
__device__ __forceinline__ T splitCombine(int op, const T* __restrict__ lhs, const T* __restrict__ divisor, int row, T loc, T rem) {
	if (op == SPMV_OP_ADD_DIV) return (loc + rem) / divisor[row];   // Jacobi folded in: out = A_loc x; out = (out + A_rem x) / diag
	if (op == SMM_OP_ASSIGN) return loc + rem;                      // out = A_loc x; out = out + A_rem x
	const T l = lhs[row];
	return op == SMM_OP_ADD ? (l + loc) + rem : (l - loc) - rem;    // out = lhs +|- A_loc x; out = out +|- A_rem x
}

// the same in two steps, for row sums that travel through out[]: out = op(lhs, loc) as the first launch left it, then out (+|-) rem [/ diag]
template <typename T>
__device__ __forceinline__ T splitFirst(int op, const T* __restrict__ lhs, int row, T loc) {
	if (op == SPMV_OP_ADD_DIV || op == SMM_OP_ASSIGN) return loc;
	const T l = lhs[row];
	return op == SMM_OP_ADD ? l + loc : l - loc;
}
template <typename T>
__device__ __forceinline__ T splitSecond(int op, const T* __restrict__ divisor, int row, T first, T rem) {
	if (op == SPMV_OP_ADD_DIV) return (first + rem) / divisor[row];
	return op == SMM_OP_SUB ? first - rem : first + rem;
}

template <typename T, int LA, int LB>
__global__ __launch_bounds__(TPB) void spmvPatternSplitKernel(const SplitArgs<T> A) {
	using Cfg = PatCfg<T>;
	const int op = A.opFlags & 0xFF;
	const bool ntOut = (A.opFlags & SPMV_NT_OUT) != 0;
	const int capMax = max(A.a.cap, A.b.cap);
	constexpr int LMAX = LA > LB ? LA : LB;
	constexpr int PARTS = LMAX > 1 ? (LMAX - 1) * (SUPER / LMAX) : 0;
	// LDS: sVal[capMax + PAD] | sMask[SUPER] | sStart[SUPER + 4] | sOffA[MAXOFF] | sOffB[MAXOFF] | sPart[PARTS] | sLoc[K * SUPER] | red[4] | sGo
	T* sVal = reinterpret_cast<T*>(smmSplitLds);
	unsigned long long* sMask = reinterpret_cast<unsigned long long*>(sVal + ((capMax + Cfg::PAD + 1) & ~1));
	int* sStart = reinterpret_cast<int*>(sMask + SUPER);
	int* sOffA = sStart + SUPER + 4;
	int* sOffB = sOffA + MAXOFF;
	T* sPart = reinterpret_cast<T*>(sOffB + MAXOFF);
	T* sLoc = sPart + ((PARTS + 1) & ~1);
	T* red = sLoc + (A.sumsInLds ? static_cast<size_t>(A.K) * SUPER : 0);
	int* sGo = reinterpret_cast<int*>(red + 4);
	if (A.doneFlag && *A.doneFlag) {
		// (a finished solve's launches are no-ops -- except the reduction point this launch carries: every rank publishes for every point, the slots
		// are double-buffered by the parity of a sequence number that must not skip; the values are never used)
		if (A.slots.world && blockIdx.x == 0) {
			p2pSlotExchange<T>(A.slots.peers, A.slots.world, A.slots.me, A.slots.point, A.slots.seq, A.slots.count, A.partials + PARTS_TOTALS, A.partials + PARTS_TOTALS,
			                   A.slots.ticks);
		}
		return;
	}

	const int t = threadIdx.x;
	for (int i = t; i < capMax + Cfg::PAD; i += TPB) sVal[i] = T(0);
	if (t < MAXOFF) {
		sOffA[t] = t < A.a.nOff ? A.a.offs[t] : 0;
		sOffB[t] = t < A.b.nOff ? A.b.offs[t] : 0;
	}
	// workgroups b, b + 8, ... share an XCD: which super tiles this one walks -- SplitMap
	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int group = blockIdx.x % nGroups;
	const int slot = blockIdx.x / nGroups;
	const int groupSlots = (static_cast<int>(gridDim.x) - group + nGroups - 1) / nGroups;
	const SplitMap map{A.nSupers, nGroups, group, slot, groupSlots, A.chunk, A.perGroup};
	__syncthreads();

	// ---- phase 1: the local block; row sums stay in LDS
	splitPhaseG<T, LA>(A.a, A.rows, map, A.K, false, sVal, sMask, sStart, sOffA, sPart, [&](int row, int at, T dot) {
		if (A.sumsInLds) sLoc[at] = dot;
		else A.out[row] = splitFirst<T>(op, A.lhs, row, dot);  // (what the first of two launches left in out[]; out may alias lhs: element by element)
	});
	if (!A.sumsInLds) __syncthreads();  // (every lane's stores of out[] have completed -- the barrier's fence waits for them -- before a lane of phase 2 reads one back)
	// ---- the halo of this exchange must have landed (the word is raised on the communicator's stream, behind the land kernel / the receive)
	if (A.landed) {
		if (t == 0) {
			int go = 1;
			const long long t0 = wall_clock64();
			for (unsigned spins = 0; __hip_atomic_load(A.landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < A.seq; ++spins) {
				__builtin_amdgcn_s_sleep(8);
				if ((spins & 63u) == 63u) {
					if (__hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
						go = 0;
						break;
					}
					if (wall_clock64() - t0 > A.ticks) {
						unsigned long long none = 0ull;
						(void)__hip_atomic_compare_exchange_strong(A.err, &none, (0x5000ull << 32) | (A.seq & 0xFFFFFFFFull), __ATOMIC_RELAXED, __ATOMIC_RELAXED,
						                                           __HIP_MEMORY_SCOPE_AGENT);
						go = 0;
						break;
					}
				}
			}
			*sGo = go;
			if (blockIdx.x == 0 && A.waited) atomicAdd(A.waited, static_cast<unsigned long long>(wall_clock64() - t0));
		}
		__syncthreads();
		if (!*sGo) {  // (the host finds the error word where it reads `done`: the solve fails with SMM_HIP_ERR_COMM; the peers' waits are released all the same)
			if (A.slots.world && blockIdx.x == 0) {
				p2pSlotExchange<T>(A.slots.peers, A.slots.world, A.slots.me, A.slots.point, A.slots.seq, A.slots.count, A.partials + PARTS_TOTALS,
				                   A.partials + PARTS_TOTALS, A.slots.ticks);
			}
			return;
		}
	}
	// ---- phase 2: the remote block; out[] is written once, the solver's dot products ride along
	T acc0 = T(0), acc1 = T(0);
	splitPhaseG<T, LB>(A.b, A.rows, map, A.K, true, sVal, sMask, sStart, sOffB, sPart, [&](int row, int at, T dot) {
		// (sums through out[]: written by a lane of this workgroup before the barriers of phase 1's end; read past this CU's L1)
		const T o = A.sumsInLds ? splitCombine<T>(op, A.lhs, A.divisor, row, sLoc[at], dot)
		                        : splitSecond<T>(op, A.divisor, row, __hip_atomic_load(A.out + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), dot);
		storeOut(A.out + row, o, ntOut);
		if (A.dotMode == 2) acc0 += o * o;
		if (A.dotMode) acc1 += o * A.w1[row];
	});
	if (A.dotMode) {
		if (A.dotMode == 2) {
			const T s0 = blockSum256(acc0, red);
			if (t == 0) A.partials[blockIdx.x] = s0;
		}
		const T s1 = blockSum256(acc1, red);
		if (t == 0) A.partials[(A.dotMode == 2 ? NPART : 0) + blockIdx.x] = s1;
		for (int i = gridDim.x + blockIdx.x * TPB + t; i < NPART; i += gridDim.x * TPB) {
			A.partials[i] = T(0);
			if (A.dotMode == 2) A.partials[NPART + i] = T(0);
		}
		if (A.opFlags & SPMV_FINISH) {
			const bool last = lastBlockSums<T>(A.partials, NPART, A.dotMode == 2 ? 2 : 1, A.partials + PARTS_TOTALS, partsTicket(A.partials));
			// the peer-to-peer transport's reduction point, by the workgroup that has just formed this rank's totals: no launch of its own
			if (last && A.slots.world) {
				__syncthreads();
				p2pSlotExchange<T>(A.slots.peers, A.slots.world, A.slots.me, A.slots.point, A.slots.seq, A.slots.count, A.partials + PARTS_TOTALS,
				                   A.partials + PARTS_TOTALS, A.slots.ticks);
			}
		}
	}
}

// the word an exchange raises when its halo is in place (one lane; on the stream the exchange ran on, behind it)
__global__ void splitSignalKernel(unsigned long long* landed, unsigned long long seq) {
	__hip_atomic_store(landed, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

void launchSplitSignal(unsigned long long* landed, unsigned long long seq, hipStream_t s) { splitSignalKernel<<<1, 1, 0, s>>>(landed, seq); }

// lanes per row this form walks a block with: what the two-launch form would use (the block's own word), 0 when the form cannot serve it
static int splitLanes(const smm_hip_csr* m) {
	if (!m || m->rows <= 0 || m->family() != SMM_SPMV_PATTERN) return 0;
	if (m->pat_state.load(std::memory_order_acquire) <= 0 || m->pat_encoding != 0) return 0;
	if (m->pat_const && !m->pat_const_off) return 0;  // (constant diagonals: the gather / march kernels read no values[]; the two launches stay)
	const int L = std::min(m->lanes(), WAVE);
	return L == 1 || L == 2 || L == 4 ? L : 0;
}

// the most entries a run of `rt` consecutive rows (cut at multiples of rt) holds; out[1 .. 9] = start[] at the borders of the eight contiguous
// eighths of the rows (how evenly the block's entries are spread over them decides the deal of the tiles: SplitMap)
__global__ void splitTileMaxKernel(int rows, const int* __restrict__ start, int rt, int* out) {
	if (blockIdx.x == 0 && threadIdx.x < 9) out[1 + threadIdx.x] = start[static_cast<long long>(rows) * threadIdx.x / 8];
	int mx = 0;
	const int nTiles = (rows + rt - 1) / rt;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nTiles; i += gridDim.x * blockDim.x) {
		const int r0 = i * rt, r1 = min(rows, r0 + rt);
		mx = max(mx, start[r1] - start[r0]);
	}
	mx = max(mx, __shfl_xor(mx, 32, WAVE));
	mx = max(mx, __shfl_xor(mx, 16, WAVE));
	mx = max(mx, __shfl_xor(mx, 8, WAVE));
	mx = max(mx, __shfl_xor(mx, 4, WAVE));
	mx = max(mx, __shfl_xor(mx, 2, WAVE));
	mx = max(mx, __shfl_xor(mx, 1, WAVE));
	if ((threadIdx.x & (WAVE - 1)) == 0) atomicMax(out, mx);
}

// LDS values a tile of this half is staged through: the fixed row tiles are staged whole, so the capacity follows the FULLEST one (counted
// once per block and lanes, on the caller's stream: 4 bytes come back) -- an average with a margin sent every tile of a matrix whose
// rows are all equally long down the unstaged path (the first version: 441 us instead of 90).  0: this form does not serve the block.
template <typename T>
static int splitCap(const smm_hip_csr* cm, int L, hipStream_t s) {
	auto* m = const_cast<smm_hip_csr*>(cm);
	const int slot = L == 1 ? 0 : L == 2 ? 1 : 2;
	int mx;
	{
		std::lock_guard<std::mutex> lock(m->tileMutex);
		if (m->split_tile_max[slot] < 0) {
			DevBuf<int> d;
			if (d.alloc(10) != SMM_HIP_OK) return 0;
			int h[10] = {};
			if (hipMemsetAsync(d.p, 0, sizeof(h), s) != hipSuccess) return 0;
			const int rt = SUPER / L;
			const int nTiles = (m->rows + rt - 1) / rt;
			splitTileMaxKernel<<<std::max(1, std::min(1024, (nTiles + 255) / 256)), 256, 0, s>>>(m->rows, m->d_start, rt, d.p);
			if (hipMemcpyAsync(h, d.p, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
				(void)hipGetLastError();
				return 0;
			}
			m->split_tile_max[slot] = h[0];
			long long most = 0;
			for (int k = 0; k < 8; ++k) most = std::max<long long>(most, h[2 + k] - h[1 + k]);
			m->split_uneven = static_cast<long long>(h[9] - h[1]) > 0 && most * 8 > (static_cast<long long>(h[9] - h[1]) * 3) / 2;  // an eighth holds > 1.5 x its share
		}
		mx = m->split_tile_max[slot];
	}
	const int limit = static_cast<int>(64 * 1024 / sizeof(T));  // (64 KB of staged values per workgroup at most)
	const int cap = (mx + 3 + 63) & ~63;  // (the LDS a workgroup takes decides how many share a CU: no rounding up to whole passes)
	if (cap <= limit) return std::max(cap, 256);
	// a few fuller tiles may take the unstaged path; a matrix whose typical tile does not fit is left to the two launches
	const double avgTile = m->rows > 0 ? static_cast<double>(m->nnz) / m->rows * (SUPER / L) : 0.0;
	return avgTile * 1.25 + 3 <= limit ? limit : 0;
}

static int splitBatch(const smm_hip_csr* m, int L) {  // (patBatch of smm_spmv_pattern.hip: a piece of p entries in ceil(p / 16) equal batches)
	const double len = m->stream_mid_len > 0 ? m->stream_mid_len : (m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 1.0);
	const int p = std::max(1, static_cast<int>(std::ceil(len / L)));
	const int nb = (p + 15) / 16;
	const int g = (p + nb - 1) / nb;
	return g <= 8 ? 8 : g <= 13 ? 13 : 16;
}

// grid, super tiles per workgroup and LDS of a launch: the grid decides K, K the LDS, the LDS how many workgroups a CU holds -- a fixed point
// found once per (instantiation, matrix shape) and kept (the occupancy query costs 10+ us of host time; the solvers launch this every ~90 us)
struct SplitPlan {
	int nSupers = 0, capMax = 0, cus = 0, sumsLdsMax = 0, chunked = 0;
	int grid = 0, K = 0, chunk = 0, perGroup = 0, sumsInLds = 1;
	size_t lds = 0;
	bool ok = false;
};

template <typename T, int LA, int LB>
static int launchSplitL(const SplitArgs<T>& base, int cus, int sumsLdsMax, int chunked, hipStream_t s) {
	SplitArgs<T> a = base;
	const int capMax = std::max(a.a.cap, a.b.cap);
	constexpr int LMAX = LA > LB ? LA : LB;
	constexpr int PARTS = LMAX > 1 ? (LMAX - 1) * (SUPER / LMAX) : 0;
	static std::mutex planMutex;
	static std::vector<SplitPlan> plans;
	static std::atomic<int> granted{0};
	SplitPlan plan;
	{
		std::lock_guard<std::mutex> lock(planMutex);
		for (const SplitPlan& p : plans) {
			if (p.nSupers == a.nSupers && p.capMax == capMax && p.cus == cus && p.sumsLdsMax == sumsLdsMax && p.chunked == chunked) plan = p;
		}
		if (!plan.nSupers) {
			const size_t fixed = static_cast<size_t>((capMax + PatCfg<T>::PAD + 1) & ~1) * sizeof(T) + SUPER * 8 + (SUPER + 4) * 4 + 2 * MAXOFF * 4 +
			                     static_cast<size_t>((PARTS + 1) & ~1) * sizeof(T) + 4 * sizeof(T) + 16;
			plan.nSupers = a.nSupers;
			plan.capMax = capMax;
			plan.cus = cus;
			plan.sumsLdsMax = sumsLdsMax;
			plan.chunked = chunked;
			int perCU = forcedWgsPerCU() > 0 ? forcedWgsPerCU() : 8;
			for (int pass = 0; pass < 8 && !plan.ok; ++pass) {
				plan.grid = std::max(1, std::min(std::min(a.nSupers, cus * perCU), NPART));
				const int nGroups = std::min(8, plan.grid);
				int perGroup = (a.nSupers + nGroups - 1) / nGroups;  // contiguous eighths ...
				plan.chunk = 0;
				if (chunked) {  // ... or chunks dealt round-robin (at least eight per group): the most tiles a group's list can hold
					plan.chunk = std::max(1, std::min(SPLIT_CHUNK, a.nSupers / 64));
					const int nChunks = (a.nSupers + plan.chunk - 1) / plan.chunk;
					perGroup = ((nChunks + nGroups - 1) / nGroups) * plan.chunk;
				}
				plan.perGroup = perGroup;
				const int slotsMin = std::max(1, plan.grid / nGroups);  // (the groups with the fewest workgroups)
				plan.K = (perGroup + slotsMin - 1) / slotsMin;
				// the local half's row sums: in LDS while that costs no workgroup per CU worth having (<= 16 KB); beyond -- few ranks, many rows per
				// workgroup -- through out[] (two small passes more, the same arithmetic)
				plan.sumsInLds = static_cast<size_t>(plan.K) * SUPER * sizeof(T) <= static_cast<size_t>(sumsLdsMax) ? 1 : 0;
				plan.lds = fixed + (plan.sumsInLds ? static_cast<size_t>(plan.K) * SUPER * sizeof(T) : 0);
				if (plan.lds > 150 * 1024) break;
				if (!ensureDynamicLds(granted, spmvPatternSplitKernel<T, LA, LB>, plan.lds)) break;
				int fit = 0;
				if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, spmvPatternSplitKernel<T, LA, LB>, TPB, plan.lds) != hipSuccess || fit < 1) {
					(void)hipGetLastError();
					break;
				}
				if (fit >= perCU || forcedWgsPerCU() > 0) plan.ok = true;
				else perCU = fit;
			}
			if (plans.size() >= 16) plans.erase(plans.begin());
			plans.push_back(plan);
		}
	}
	if (!plan.ok) return 1;
	a.K = plan.K;
	a.sumsInLds = plan.sumsInLds;
	a.chunk = plan.chunk;
	a.perGroup = plan.perGroup;
	const int profSlot = profBegin(s);  // (smm_hip_profile_*: one SpMV launch)
	spmvPatternSplitKernel<T, LA, LB><<<plan.grid, TPB, plan.lds, s>>>(a);
	profEnd(profSlot, s);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

// SMM_HIP_OK: launched; 1: this form does not serve the pair of blocks (nothing was enqueued: run the two launches); < 0: a failure
template <typename T>
int launchSpmvSplit(const smm_hip_csr* aLoc, const smm_hip_csr* aRem, int op, const T* lhs, const T* divisor, const T* own, const T* ext, T* out, int dotMode,
                    const T* w1, T* partials, const int* doneFlag, int extraFlags, const unsigned long long* landed, unsigned long long seq, unsigned long long* err,
                    long long ticks, hipStream_t s, const P2PSlotArgs* slots, int sumsLdsMax, unsigned long long* waited) {
	const int LA = splitLanes(aLoc), LB = splitLanes(aRem);
	if (!LA || !LB || aLoc->rows != aRem->rows) return 1;
	if (aLoc->dtype != dtypeOf<T>() || aRem->dtype != dtypeOf<T>()) return 1;
	SplitArgs<T> a{};
	const int capA = splitCap<T>(aLoc, LA, s), capB = splitCap<T>(aRem, LB, s);
	if (!capA || !capB) return 1;
	auto side = [&](const smm_hip_csr* m, int L, const T* x) {
		SplitSide<T> sd{};
		sd.x = x;
		sd.cols = m->cols;
		sd.offs = m->d_pat_off;
		sd.start = m->d_start;
		sd.masks = m->d_pat_masks;
		sd.positions = m->d_positions;
		sd.values = static_cast<const T*>(m->d_values);
		sd.nOff = m->pat_k;
		sd.cap = m == aLoc ? capA : capB;
		sd.batch = splitBatch(m, L);
		sd.stageLimit = (m->nnz & ~3) - (sd.cap + 4);
		return sd;
	};
	a.a = side(aLoc, LA, own);
	a.b = side(aRem, LB, ext);
	a.rows = aLoc->rows;
	a.nSupers = (a.rows + SUPER - 1) / SUPER;
	int kop = op;
	if (extraFlags & SPMV_ADD_DIV) kop = SPMV_OP_ADD_DIV;
	a.opFlags = kop | (extraFlags & SPMV_FINISH) | spmvOutFlags(aLoc, sizeof(T));
	a.dotMode = dotMode;
	a.lhs = lhs;
	a.divisor = divisor;
	a.out = out;
	a.w1 = w1;
	a.partials = partials;
	a.doneFlag = doneFlag;
	a.landed = landed;
	a.seq = seq;
	a.err = err;
	a.ticks = ticks;
	a.waited = waited;
	if (slots && dotMode && (extraFlags & SPMV_FINISH)) a.slots = *slots;
	// the exchange is itself a few workgroups (the land kernel, the peers' pushes, an RCCL kernel) that must find room beside this grid while it
	// waits for them
	// (8 CUs' worth of slots -- one per XCD -- stay free: ~40 of the 256 CUs then hold four workgroups instead of five and have registers and wave
	// slots to spare; the push, forward and land kernels of an exchange run one after the other on the communicator's stream, so one of them at a
	// time needs room.  16 CUs' worth cost the SpMV 2-3 us and bought nothing measurable)
	const int cus = landed ? std::max(8, numCUs() - 8) : numCUs();
	const int chunked = (aLoc->split_uneven || aRem->split_uneven) ? 1 : 0;
	switch (LA * 8 + LB) {
	case 1 * 8 + 1: return launchSplitL<T, 1, 1>(a, cus, sumsLdsMax, chunked, s);
	case 1 * 8 + 2: return launchSplitL<T, 1, 2>(a, cus, sumsLdsMax, chunked, s);
	case 1 * 8 + 4: return launchSplitL<T, 1, 4>(a, cus, sumsLdsMax, chunked, s);
	case 2 * 8 + 1: return launchSplitL<T, 2, 1>(a, cus, sumsLdsMax, chunked, s);
	case 2 * 8 + 2: return launchSplitL<T, 2, 2>(a, cus, sumsLdsMax, chunked, s);
	case 2 * 8 + 4: return launchSplitL<T, 2, 4>(a, cus, sumsLdsMax, chunked, s);
	case 4 * 8 + 1: return launchSplitL<T, 4, 1>(a, cus, sumsLdsMax, chunked, s);
	case 4 * 8 + 2: return launchSplitL<T, 4, 2>(a, cus, sumsLdsMax, chunked, s);
	default: return launchSplitL<T, 4, 4>(a, cus, sumsLdsMax, chunked, s);
	}
}

template int launchSpmvSplit<float>(const smm_hip_csr*, const smm_hip_csr*, int, const float*, const float*, const float*, const float*, float*, int, const float*, float*,
                                    const int*, int, const unsigned long long*, unsigned long long, unsigned long long*, long long, hipStream_t, const P2PSlotArgs*, int, unsigned long long*);
template int launchSpmvSplit<double>(const smm_hip_csr*, const smm_hip_csr*, int, const double*, const double*, const double*, const double*, double*, int, const double*,
                                     double*, const int*, int, const unsigned long long*, unsigned long long, unsigned long long*, long long, hipStream_t, const P2PSlotArgs*, int, unsigned long long*);

void preloadSplitUnit() {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(splitSignalKernel));
	(void)hipGetLastError();
}

}  // namespace smm
