// smm_precond_block.hip -- BLOCK_ILU0 / BLOCK_SGS: the ILU(0) / Symmetric Gauss-Seidel preconditioner of the block-diagonal part of A,
// one WAVEFRONT per block, the block's vector in LDS, one launch per apply.
//
// Why (profiles/r02/sweep_timing.txt): the exact global sweeps of smm_precond.hip pay one fabric round trip (1.2-1.8 us) per
// dependency level -- 644 of them per apply on the 108^3 convection-diffusion problem -- and lose to the unpreconditioned solve.  Here
// the rows are cut into contiguous blocks (<= 1024 rows, <= 8192 stored entries), couplings between blocks are dropped from M only, and
// every dependency of a sweep stays inside one wavefront: a level costs one LDS round trip (~60 ns), all blocks run at the same time.
//
// Per block, built once at create time (all on the device):
//   * the dependency levels of the lower and of the upper sweep, rows sorted by level, cut into CHUNKS of 64 sorted rows;
//   * for ILU0 the factorisation of the block (IKJ on the block's pattern, one lane per row, level by level, out of LDS);
//   * one fixed-size RECORD per (chunk, lane) and sweep: {row, level, count | columns (16-bit, local) | diagonal | values} of that
//     row's first KREG in-block entries in the order the sequential sweep visits them; rows with more entries continue in an
//     overflow list.  Records of a chunk are adjacent, so a wavefront streams them with 8-byte loads, D = 2 chunks ahead of their use.
// Apply (blkApplyKernel): a wavefront loads its block's slice of rhs into LDS, walks the lower chunks and then the upper ones, and
// inside a chunk the levels one after the other: the lanes whose row is in that level read the x[] they need from LDS, run the
// row's multiply-adds in the sequential order and store the row's result -- the LDS pipeline of a wavefront is in order, so a later
// level sees it without any barrier.  Same operands, same order, same roundings as the sequential sweeps over the block-diagonal
// matrix (the tests compare bit for bit with the CPU restatement of exactly that), and BLOCK_SGS thereby the reference's SGSPreconditioner::apply
// (ref:1658-1713) on that matrix.  The dot products BiCGStab needs of the result ride in the epilogue.
#include <algorithm>
#include <cmath>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "smm_device.h"
#include "smm_internal.h"

struct smm_precond_block {
	int nBlocks = 0;
	int blockRows = 0;      // the cut: at most this many rows ...
	int levelCap = 0;       // the level cut: no sweep of a block deeper than this many levels (0 = no cut)
	int capNnz = 0;         // ... and this many stored entries per block
	int kreg = 2;           // in-block entries per row and sweep that live in the records
	bool overflow = false;  // some row has more than kreg: the rest is in the overflow lists
	int levelsLo = 0, levelsUp = 0;  // deepest block
	int maxInBlock = 0;              // most in-block entries of a block
	long long nChunks = 0;
	int2* d_bounds = nullptr;  // [nBlocks + 1] {first row (contiguous) / first position in d_rowOrder (bricks), unused}
	int* d_rowOrder = nullptr; // bricks: the rows block by block, ascending inside a block; nullptr = contiguous blocks
	int* d_invOrder = nullptr; // bricks: row -> position in d_rowOrder
	int brick[3] = {0, 0, 0};  // bricks: rows of a brick along each grid axis
	int* d_chunk0 = nullptr;   // [nBlocks + 1] chunks in front of block b
	unsigned* d_recLo = nullptr;
	unsigned* d_recUp = nullptr;
	int* d_ovPtrLo = nullptr;  // [nChunks * 64 + 1] when overflow
	int* d_ovPtrUp = nullptr;
	unsigned short* d_ovColLo = nullptr;
	unsigned short* d_ovColUp = nullptr;
	void* d_ovValLo = nullptr;
	void* d_ovValUp = nullptr;
	std::vector<int> hostBounds;  // filled on first query
	std::mutex boundsMutex;
};

namespace smm {

constexpr int BLK_TPB = 256;            // set-up kernels: one workgroup per block
constexpr int BLK_DEFAULT_ROWS = 1024;
constexpr int BLK_MIN_ROWS = 64;
constexpr int BLK_MAX_ROWS = 2048;
constexpr int BLK_DEFAULT_LEVEL_CAP = 16;  // deepest sweep of a block, in dependent levels (the level cut below)
constexpr int BLK_CAP_NNZ = 8192;
constexpr unsigned BLK_NOROW = 0xFFFu;  // row field of a padding lane
constexpr unsigned short BLK_UNKNOWN = 0xFFFFu;

enum BlkMode { B_ILU_LO = 0, B_ILU_UP = 1, B_SGS_LO = 2, B_SGS_UP = 3 };

// record of one (chunk, lane): dwords {meta | ceil(KREG / 2) column pairs | [diagonal] | KREG values}, padded to an even count
template <typename T, bool HASD, int KREG>
struct RecLayout {
	static constexpr int VW = sizeof(T) / 4;
	static constexpr int COLW = (KREG + 1) / 2;
	static constexpr int DIAG_AT = 1 + COLW;
	static constexpr int VAL_AT = DIAG_AT + (HASD ? VW : 0);
	static constexpr int RAW = VAL_AT + VW * KREG;
	static constexpr int DW = (RAW + 1) & ~1;
};

template <typename T>
__device__ __forceinline__ T fromWords(const unsigned* w) {
	T v;
	if (sizeof(T) == 4) {
		__builtin_memcpy(&v, w, 4);
	} else {
		const unsigned long long b = static_cast<unsigned long long>(w[0]) | (static_cast<unsigned long long>(w[1]) << 32);
		__builtin_memcpy(&v, &b, 8);
	}
	return v;
}
template <typename T>
__device__ __forceinline__ void toWords(T v, unsigned* w) {
	if (sizeof(T) == 4) {
		__builtin_memcpy(w, &v, 4);
	} else {
		unsigned long long b;
		__builtin_memcpy(&b, &v, 8);
		w[0] = static_cast<unsigned>(b);
		w[1] = static_cast<unsigned>(b >> 32);
	}
}

// ---------------------------------------------------------------------------------------------------------
// apply
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct BlkApplyArgs {
	int nBlocks;
	const int2* bounds;
	const int* rowOrder;  // nullptr: block b = rows [bounds[b], bounds[b+1]); else rows rowOrder[bounds[b] .. bounds[b+1])
	const int* chunk0;
	const unsigned* recLo;
	const unsigned* recUp;
	const int* ovPtrLo;
	const int* ovPtrUp;
	const unsigned short* ovColLo;
	const unsigned short* ovColUp;
	const T* ovValLo;
	const T* ovValUp;
	const T* rhs;
	T* x;
	// SPMV form (blkApplyKernel<..., true>): rhs = A v is formed row by row on the way into LDS -- the three CSR arrays and v
	const int* aStart;
	const int* aPos;
	const T* aVal;
	const T* spmvX;
	// ... or, for a matrix in the row-mask encoding (smm_spmv_pattern.hip: verified against every entry at analysis time), its row masks and
	// the <= 64 offsets -- positions[] is not read -- and, with constant diagonals (patCval != nullptr), the diagonals' values: values[] is
	// not read either
	const unsigned long long* patMasks;
	const int* patOff;
	const unsigned long long* patCval;
	int patK;
	int cols;
	int ldsRows;  // elements of the block vector in LDS (the tables of the constant-diagonal form sit behind it)
	int dotMode;  // 0 none; 1: x.w1 -> partials[0..NPART); 2: x.x -> partials[0..NPART), x.w1 -> partials[NPART..2 NPART)
	const T* w1;
	T* partials;
	const int* doneFlag;
};

template <int DW>
__device__ __forceinline__ void loadRec(unsigned (&w)[DW], const uint2* p) {
#pragma unroll
	for (int i = 0; i < DW / 2; ++i) {
		const uint2 v = p[i];
		w[2 * i] = v.x;
		w[2 * i + 1] = v.y;
	}
}

// What bounds a sweep (measured on the 108^3 problem, r03, tools/block_apply_only.py): NOT the record stream -- 1, 2, 4 or 8 chunks
// requested ahead give 47.6 / 48.5 / 51.1 / 51.2 us per apply (SMM_HIP_BLOCK_DEPTH; the deeper rings only add code) --, not the number of
// instructions of a level (26 -> 17 by dropping the per-entry predicates: 48 -> 48 us), but the dependent chain of a level itself:
// ds_write of the previous level -> ds_read behind it in the wavefront's in-order LDS queue -> the row's fp64 multiply-adds -> ds_write,
// ~120 ns, plus ~400 ns per chunk of unpacking: 264 levels = 47 us.  So the ring is shallow (D = 2), and the first chunks of BOTH
// sweeps are requested before anything is waited for.
//
// Register sets: chunk c lives in set c % (2 D) and is loaded D chunks before its use into the set chunk c - D has just left -- so a
// set is never reloaded while it is read, no register is copied at the loop's back edge, and the wait in front of a chunk's first use
// counts only the younger loads (the D chunks behind it stay in flight).  Every load is issued unconditionally (past the end: the last
// chunk again), so that the number of loads in flight is the same on every path and the compiler can wait with a count instead of
// draining the queue.
template <typename T, int MODE, int KREG, int D>
struct SweepRing {
	static constexpr int DW = RecLayout<T, MODE != B_ILU_LO, KREG>::DW;
	static constexpr int U = 2 * D;
	unsigned r[U][DW];
	const uint2* base;
	__device__ __forceinline__ void prologue(const unsigned* __restrict__ rec, int nc) {
		base = reinterpret_cast<const uint2*>(rec) + static_cast<size_t>(threadIdx.x) * (DW / 2);
#pragma unroll
		for (int d = 0; d < D; ++d) loadRec<DW>(r[d], base + static_cast<size_t>(min(d, nc - 1)) * (static_cast<size_t>(WAVE) * (DW / 2)));
	}
};

// one sweep of one block: nc chunks of records (the ring's prologue has been issued); xs = the block's vector in LDS
template <typename T, int MODE, int KREG, bool OV, int D>
__device__ __forceinline__ void blkSweep(SweepRing<T, MODE, KREG, D>& sr, long long recIndex0, int nc, T* xs, const int* __restrict__ ovPtr,
                                         const unsigned short* __restrict__ ovCol, const T* __restrict__ ovVal) {
	constexpr bool LOWER = MODE == B_ILU_LO || MODE == B_SGS_LO;
	constexpr bool HASD = MODE != B_ILU_LO;
	using L = RecLayout<T, HASD, KREG>;
	constexpr int DW = L::DW;
	const int lane = threadIdx.x;
	const uint2* base = sr.base;
	constexpr size_t CHUNK_STRIDE = static_cast<size_t>(WAVE) * (DW / 2);  // in uint2
	constexpr int U = 2 * D;
	unsigned (&ring)[U][DW] = sr.r;
	for (int c0 = 0; c0 < nc; c0 += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int c = c0 + u;
			loadRec<DW>(ring[(u + D) % U], base + static_cast<size_t>(min(c + D, nc - 1)) * CHUNK_STRIDE);
			if (c < nc) {
				const unsigned(&w)[DW] = ring[u];
				const unsigned meta = w[0];
				const unsigned row = meta & 0xFFFu;
				const int lvl = static_cast<int>((meta >> 12) & 0xFFFu);
				const int n = static_cast<int>(meta >> 24);  // in-block entries of this sweep, capped at 255
				const bool valid = row != BLK_NOROW;
				const int lFirst = __builtin_amdgcn_readfirstlane(lvl);
				const int lLast = __builtin_amdgcn_readlane(lvl, WAVE - 1);
				const int myLvl = valid ? lvl : -1;  // padding lanes never match a level
				const T own = valid ? xs[row] : T(0);  // rhs (lower sweep) or y (upper sweep) of the lane's row: no other row writes it
				// the step is the dependency chain of the whole sweep (one LDS round trip + the row's arithmetic per level): when every
				// row of the chunk has exactly KREG entries (the interior of a stencil) it runs without any per-entry predicate
				const bool allFull = __ballot(valid && n != KREG) == 0ull;
				unsigned cols[KREG];
				T vals[KREG];
#pragma unroll
				for (int k = 0; k < KREG; ++k) {
					cols[k] = (w[1 + k / 2] >> (16 * (k & 1))) & 0xFFFFu;
					vals[k] = fromWords<T>(&w[L::VAL_AT + L::VW * k]);
				}
				T diag = T(1);
				if (HASD) diag = fromWords<T>(&w[L::DIAG_AT]);
				if (allFull) {
					for (int lv = lFirst; lv <= lLast; ++lv) {
						if (myLvl == lv) {
							T xv[KREG];
#pragma unroll
							for (int k = 0; k < KREG; ++k) xv[k] = xs[cols[k]];
							T acc = MODE == B_SGS_UP ? T(0) : own;
#pragma unroll
							for (int k = 0; k < KREG; ++k) acc = MODE == B_SGS_UP ? smmFma(vals[k], xv[k], acc) : smmFma(-vals[k], xv[k], acc);  // ref:1706 / ref:1686
							T result;
							if (MODE == B_ILU_LO) result = acc;
							else if (MODE == B_SGS_UP) result = own - acc / diag;  // ref:1710
							else result = acc / diag;                               // ref:1694; ILU: U x = y
							xs[row] = result;
						}
					}
				} else {
					for (int lv = lFirst; lv <= lLast; ++lv) {
						if (myLvl == lv) {
							T xv[KREG];
#pragma unroll
							for (int k = 0; k < KREG; ++k) {
								if (k < n) xv[k] = xs[cols[k]];
							}
							T acc = MODE == B_SGS_UP ? T(0) : own;
#pragma unroll
							for (int k = 0; k < KREG; ++k) {
								if (k < n) acc = MODE == B_SGS_UP ? smmFma(vals[k], xv[k], acc) : smmFma(-vals[k], xv[k], acc);
							}
							if (OV && n > KREG) {  // the tail of a long row, straight from memory (slow path)
								const long long g = recIndex0 + static_cast<long long>(c) * WAVE + lane;
								for (int e = ovPtr[g]; e < ovPtr[g + 1]; ++e) {
									const T val = ovVal[e];
									const T xo = xs[ovCol[e]];
									acc = MODE == B_SGS_UP ? smmFma(val, xo, acc) : smmFma(-val, xo, acc);
								}
							}
							T result;
							if (MODE == B_ILU_LO) result = acc;
							else if (MODE == B_SGS_UP) result = own - acc / diag;
							else result = acc / diag;
							xs[row] = result;
						}
					}
				}
			}
		}
	}
	(void)LOWER;
}

// SPMV form: the rows of A v that make up a block's right-hand side, summed in the order of the stored entries from 0 (ref:1484-1489:
// the sum the SpMV kernels form at one lane per row), by BLK_SPMV_TPB lanes -- four wavefronts instead of the one that sweeps, since
// a row costs three dependent trips to memory (row -> start[] -> positions[] / values[] -> v[]) and only independent rows hide them.
// RB rows per lane and pass, the first KU entries of each requested together (every load unconditional, clamped to the row's last
// entry; the multiply-add is what is predicated), longer rows finish in a loop.
constexpr int BLK_SPMV_TPB = 256;
template <typename T>
__device__ __forceinline__ void blkSpmvRows(const BlkApplyArgs<T>& a, int r0, int nb, T* xs) {
	constexpr int RB = 2, KU = 8;  // (registers: the launch keeps as many of these workgroups on a CU as it has blocks there)
	const int* __restrict__ start = a.aStart;
	const int* __restrict__ pos = a.aPos;
	const T* __restrict__ val = a.aVal;
	const T* __restrict__ v = a.spmvX;
	for (int i0 = threadIdx.x; i0 < nb; i0 += BLK_SPMV_TPB * RB) {
		int g[RB], kb[RB], kn[RB];
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			const int i = i0 + j * BLK_SPMV_TPB;
			g[j] = -1;
			if (i < nb) g[j] = a.rowOrder ? a.rowOrder[r0 + i] : r0 + i;
		}
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			kb[j] = g[j] >= 0 ? start[g[j]] : 0;
			kn[j] = g[j] >= 0 ? start[g[j] + 1] - kb[j] : 0;
		}
		int c[RB][KU];
		T av[RB][KU], xv[RB][KU];
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			const int last = max(kb[j] + kn[j] - 1, 0);
#pragma unroll
			for (int u = 0; u < KU; ++u) {
				const int k = min(kb[j] + u, last);
				c[j][u] = pos[k];
				av[j][u] = val[k];
			}
		}
#pragma unroll
		for (int j = 0; j < RB; ++j) {
#pragma unroll
			for (int u = 0; u < KU; ++u) xv[j][u] = v[c[j][u]];
		}
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			T dot = T(0);
#pragma unroll
			for (int u = 0; u < KU; ++u) {
				const T next = smmFma(av[j][u], xv[j][u], dot);
				dot = u < kn[j] ? next : dot;
			}
			for (int k = kb[j] + KU; k < kb[j] + kn[j]; ++k) dot = smmFma(val[k], v[pos[k]], dot);
			const int i = i0 + j * BLK_SPMV_TPB;
			if (i < nb) xs[i] = dot;
		}
	}
}

// the same rows from the constant-diagonal encoding: the row's mask instead of start[] / positions[] / values[] (two trips to memory, 8
// bytes per row); sOff / sC: the offsets and the diagonals' values in LDS.  Products and order are spmvPatternConstKernel's: the mask's
// bits ascending = the row's columns ascending (ref:1484-1499), dead slots gather a clamped column and are not added.
constexpr int BLK_MAXOFF = 64;
// VALS: the row-mask encoding of a matrix whose diagonals are NOT constant -- the columns still come from the mask, the values from
// values[] (the k-th set bit of a row's mask is its k-th stored entry): start[] and the mask are read together, values[] and the gathers
// together -- a trip to memory and 4 bytes per entry less than the three arrays.
template <typename T, bool VALS>
__device__ __forceinline__ void blkSpmvRowsConst(const BlkApplyArgs<T>& a, int r0, int nb, T* xs, const int* sOff, const T* sC) {
	constexpr int RB = 2, KU = 8;
	const unsigned long long* __restrict__ masks = a.patMasks;
	const T* __restrict__ v = a.spmvX;
	const T* __restrict__ val = a.aVal;
	for (int i0 = threadIdx.x; i0 < nb; i0 += BLK_SPMV_TPB * RB) {
		int g[RB];
		unsigned long long mm[RB];
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			const int i = i0 + j * BLK_SPMV_TPB;
			g[j] = -1;
			if (i < nb) g[j] = a.rowOrder ? a.rowOrder[r0 + i] : r0 + i;
		}
#pragma unroll
		for (int j = 0; j < RB; ++j) mm[j] = g[j] >= 0 ? masks[g[j]] : 0ULL;
		int kb[RB], klast[RB];  // VALS: the row's next and last stored entry
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			kb[j] = VALS && g[j] >= 0 ? a.aStart[g[j]] : 0;
			klast[j] = VALS && g[j] >= 0 ? max(a.aStart[g[j] + 1] - 1, 0) : 0;
		}
		T dot[RB];
#pragma unroll
		for (int j = 0; j < RB; ++j) dot[j] = T(0);
		bool more;
		do {
			int c[RB][KU];
			T cv[RB][KU], xv[RB][KU];
			int cnt[RB];  // entries of the row that are still to come: slot u of this pass is live when u < cnt
#pragma unroll
			for (int j = 0; j < RB; ++j) {
				cnt[j] = __popcll(mm[j]);
#pragma unroll
				for (int u = 0; u < KU; ++u) {
					const int jj = mm[j] ? __builtin_ctzll(mm[j]) : 0;
					mm[j] &= mm[j] - 1;
					c[j][u] = min(max(max(g[j], 0) + sOff[jj], 0), a.cols - 1);
					if (VALS) cv[j][u] = val[min(kb[j] + u, klast[j])];  // (dead slots: the row's last entry, not added)
					else cv[j][u] = sC[jj];
				}
				if (VALS) kb[j] += KU;
			}
#pragma unroll
			for (int j = 0; j < RB; ++j) {
#pragma unroll
				for (int u = 0; u < KU; ++u) xv[j][u] = v[c[j][u]];
			}
			more = false;
#pragma unroll
			for (int j = 0; j < RB; ++j) {
#pragma unroll
				for (int u = 0; u < KU; ++u) {
					const T next = smmFma(cv[j][u], xv[j][u], dot[j]);
					dot[j] = u < cnt[j] ? next : dot[j];
				}
				more = more || mm[j] != 0ULL;
			}
		} while (more);
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			const int i = i0 + j * BLK_SPMV_TPB;
			if (i < nb) xs[i] = dot[j];
		}
	}
}

template <typename T, int KIND, int KREG, bool OV, int D, bool SPMV = false>
__global__ __launch_bounds__(SPMV ? BLK_SPMV_TPB : WAVE) void blkApplyKernel(const BlkApplyArgs<T> a) {
	extern __shared__ __align__(16) unsigned char blkLds[];
	T* xs = reinterpret_cast<T*>(blkLds);
	if (a.doneFlag && *a.doneFlag) return;
	// SPMV form: wavefront 0 sweeps (the sweeps lean on ONE wavefront's in-order LDS queue), all four form the right-hand side
	const bool sweeper = !SPMV || threadIdx.x < WAVE;
	constexpr int LO = KIND == SMM_PRECOND_BLOCK_ILU0 ? B_ILU_LO : B_SGS_LO;
	constexpr int UP = KIND == SMM_PRECOND_BLOCK_ILU0 ? B_ILU_UP : B_SGS_UP;
	using LL = RecLayout<T, LO != B_ILU_LO, KREG>;
	using LU = RecLayout<T, true, KREG>;
	const int lane = threadIdx.x;
	T acc0 = T(0), acc1 = T(0);
	for (int b = blockIdx.x; b < a.nBlocks; b += gridDim.x) {
		const int r0 = a.bounds[b].x;
		const int nb = a.bounds[b + 1].x - r0;
		const int nc = (nb + WAVE - 1) / WAVE;
		const long long rec0 = static_cast<long long>(a.chunk0[b]) * WAVE;
		SweepRing<T, LO, KREG, D> ringLo;
		SweepRing<T, UP, KREG, D> ringUp;
		if (!SPMV) {
			ringLo.prologue(a.recLo + rec0 * LL::DW, nc);
			ringUp.prologue(a.recUp + rec0 * LU::DW, nc);  // (in flight across the whole lower sweep)
		}
		if (SPMV) {
			if (a.patMasks) {
				// (behind the block's vector: the launch sized the LDS for it)
				T* sC = xs + a.ldsRows;
				int* sOff = reinterpret_cast<int*>(sC + BLK_MAXOFF);
				if (threadIdx.x < BLK_MAXOFF) {
					const int t = threadIdx.x;
					sOff[t] = t < a.patK ? a.patOff[t] : 0;
					T c = T(0);
					if (t < a.patK && a.patCval) {
						const unsigned long long bits = a.patCval[t];
						if (sizeof(T) == 4) {
							const unsigned lo = static_cast<unsigned>(bits);
							__builtin_memcpy(&c, &lo, 4);
						} else {
							__builtin_memcpy(&c, &bits, sizeof(T));
						}
					}
					sC[t] = c;
				}
				__syncthreads();
				if (a.patCval) blkSpmvRowsConst<T, false>(a, r0, nb, xs, sOff, sC);
				else blkSpmvRowsConst<T, true>(a, r0, nb, xs, sOff, sC);
			} else {
				blkSpmvRows<T>(a, r0, nb, xs);
			}
			__syncthreads();
			// the launch has one workgroup per block (launchBlkApply): the three helper wavefronts are done and give their registers back --
			// a sweep is a chain of LDS round trips, and what a sweep costs is how many blocks of a CU sweep at the same time
			if (!sweeper) return;
			// (the first records are requested only now: in flight across the SpMV they would cost every wavefront of the launch their
			// registers, and the registers decide how many workgroups start at once)
			ringLo.prologue(a.recLo + rec0 * LL::DW, nc);
			ringUp.prologue(a.recUp + rec0 * LU::DW, nc);
		} else {
			for (int i = lane; i < nb; i += WAVE) xs[i] = a.rhs[a.rowOrder ? a.rowOrder[r0 + i] : r0 + i];
		}
		blkSweep<T, LO, KREG, OV, D>(ringLo, rec0, nc, xs, a.ovPtrLo, a.ovColLo, a.ovValLo);
		blkSweep<T, UP, KREG, OV, D>(ringUp, rec0, nc, xs, a.ovPtrUp, a.ovColUp, a.ovValUp);
		if (a.dotMode == 0) {
			for (int i = lane; i < nb; i += WAVE) a.x[a.rowOrder ? a.rowOrder[r0 + i] : r0 + i] = xs[i];
		} else {
			for (int i = lane; i < nb; i += WAVE) {
				const T v = xs[i];
				const int g = a.rowOrder ? a.rowOrder[r0 + i] : r0 + i;
				a.x[g] = v;
				const T w = a.w1[g];
				if (a.dotMode == 2) {
					acc0 += v * v;
					acc1 += v * w;
				} else {
					acc0 += v * w;
				}
			}
		}
	}
	if (a.dotMode != 0) {
		acc0 = groupSum<WAVE>(acc0);
		if (a.dotMode == 2) acc1 = groupSum<WAVE>(acc1);
		if (lane == 0) {
			a.partials[blockIdx.x] = acc0;
			if (a.dotMode == 2) a.partials[NPART + blockIdx.x] = acc1;
		}
		// every slot is (re)written by every launch: idle slots hold 0
		for (int i = gridDim.x + blockIdx.x * WAVE + lane; i < NPART; i += gridDim.x * WAVE) {
			a.partials[i] = T(0);
			if (a.dotMode == 2) a.partials[NPART + i] = T(0);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// set-up
// ---------------------------------------------------------------------------------------------------------
__global__ void blkChunkCountKernel(int nBlocks, const int2* __restrict__ bounds, int* __restrict__ counts) {
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b < nBlocks) counts[b] = (bounds[b + 1].x - bounds[b].x + WAVE - 1) / WAVE;
	if (b == nBlocks) counts[b] = 0;
}

// exclusive scan of v over the 256 threads of a workgroup; *total = the sum.  scratch: 8 ints of LDS.
__device__ __forceinline__ int blockExclusiveScan256(int v, int* scratch, int* total) {
	const int lane = threadIdx.x & (WAVE - 1);
	const int wave = threadIdx.x >> 6;
	int incl = v;
#pragma unroll
	for (int o = 1; o < WAVE; o <<= 1) {
		const int t = __shfl_up(incl, o, WAVE);
		if (lane >= o) incl += t;
	}
	if (lane == WAVE - 1) scratch[wave] = incl;
	__syncthreads();
	int offset = 0;
	for (int k = 0; k < wave; ++k) offset += scratch[k];
	const int all = scratch[0] + scratch[1] + scratch[2] + scratch[3];
	__syncthreads();
	*total = all;
	return offset + incl - v;
}

// first index in positions[lo .. hi) with positions[.] >= key
__device__ __forceinline__ int lowerBoundCol(const int* __restrict__ positions, int lo, int hi, int key) {
	while (lo < hi) {
		const int mid = lo + (hi - lo) / 2;
		if (positions[mid] < key) lo = mid + 1;
		else hi = mid;
	}
	return lo;
}

// The block's own CSR (entries whose column lies inside the block, columns local) in LDS.  ROWS_PER_THREAD * 256 >= blockRows.
struct BlkStage {
	int* lptr;             // [maxRows + 1]
	int* ibG;              // [maxRows]   global index of the first in-block entry of the row
	unsigned short* nlow;  // [maxRows]   in-block entries left of the diagonal
	unsigned short* lcol;  // [cap]
	unsigned short* nall = nullptr;  // [maxRows] (optional) stored in-block entries of the row, the ones the level cut drops included
};

// The LEVEL CUT (r03): a block's forward sweep gives every row a level (0: no kept entry left of the diagonal; else 1 + the deepest
// level among the rows its kept entries point to), the backward sweep likewise.  With a cap C, entries that point to a row of level
// C - 1 (`top`) are dropped from M -- exactly like the entries that couple two blocks -- so neither sweep of any block is deeper than
// C levels, whatever the matrix: a 1024-row block of a 108^3 grid has 116 dependent levels, 16 after the cut, for 2 % more BiCGStab
// iterations (profiles/r03/level_cap.txt).  The rule is a recurrence in the sweep's own row order; the tests' CPU checker
// states it sequentially (its block_level_cut).  top = C - 1, 0 = no cut.
__device__ __forceinline__ bool blkKeeps(int i, int c, const unsigned short* lvlLo, const unsigned short* lvlUp, int top) {
	if (top <= 0 || c == i) return true;
	return static_cast<int>(c < i ? lvlLo[c] : lvlUp[c]) < top;
}

// Which rows form a block.  CONTIGUOUS (rowOrder == nullptr): block b = rows [p0, p0 + nb).  PERMUTED (bricks of a grid, r03): block b =
// rows rowOrder[p0 .. p0 + nb) -- ascending inside a block, so a block's rows keep their natural order -- and invOrder[] is the inverse
// map (row -> position).  Local index of row / column g: its position - p0; it belongs to the block when that lies in [0, nb).
struct BlkRows {
	const int* rowOrder;
	const int* invOrder;
	__device__ __forceinline__ int rowAt(int p) const { return rowOrder ? rowOrder[p] : p; }
	__device__ __forceinline__ int posOf(int g) const { return invOrder ? invOrder[g] : g; }
};

// returns bit 0: some row of this thread has no diagonal entry (or is empty), bit 1 (checkMagnitude): |d| < 1e-5; fills st.lptr / ibG / nlow / lcol (and lval when LVAL is not null).
// Staged: the entries of the block's rows whose column belongs to the block, columns local, in the row's own (ascending) order; top > 0:
// minus the entries the level cut drops (levels of both sweeps in lvlLo / lvlUp).  st.ibG[i] = first stored entry of local row i,
// st.nall[i] = its stored entries (all of them: the factor write-back walks the row again with the same tests).
template <typename T>
__device__ int stageBlock(const BlkStage& st, T* lval, const BlkRows& br, int p0, int nb, const int* __restrict__ start, const int* __restrict__ positions,
                          const T* __restrict__ vals, int* scanScratch, int checkMagnitude = 0, const unsigned short* lvlLo = nullptr,
                          const unsigned short* lvlUp = nullptr, int top = 0) {
	constexpr int RPT = BLK_MAX_ROWS / BLK_TPB;  // 8 consecutive rows per thread
	const int t = threadIdx.x;
	int cnt[RPT], all[RPT], ib[RPT];
	int mine = 0;
#pragma unroll
	for (int j = 0; j < RPT; ++j) {
		const int i = t * RPT + j;
		cnt[j] = 0;
		all[j] = 0;
		ib[j] = 0;
		if (i < nb) {
			const int g = br.rowAt(p0 + i);
			const int b = start[g], e = start[g + 1];
			ib[j] = b;
			all[j] = e - b;
			int kept = 0;
			for (int k = b; k < e; ++k) {
				const int c = br.posOf(positions[k]) - p0;
				kept += (static_cast<unsigned>(c) < static_cast<unsigned>(nb) && blkKeeps(i, c, lvlLo, lvlUp, top)) ? 1 : 0;
			}
			cnt[j] = kept;
		}
		mine += cnt[j];
	}
	int total = 0;
	int at = blockExclusiveScan256(mine, scanScratch, &total);
	int bad = 0;
#pragma unroll
	for (int j = 0; j < RPT; ++j) {
		const int i = t * RPT + j;
		if (i < nb) {
			st.lptr[i] = at;
			st.ibG[i] = ib[j];
			if (st.nall) st.nall[i] = static_cast<unsigned short>(min(all[j], 0xFFFF));
			int low = 0, q = 0;
			bool diag = false;
			for (int e = 0; e < all[j]; ++e) {
				const int c = br.posOf(positions[ib[j] + e]) - p0;
				if (static_cast<unsigned>(c) >= static_cast<unsigned>(nb) || !blkKeeps(i, c, lvlLo, lvlUp, top)) continue;
				st.lcol[at + q] = static_cast<unsigned short>(c);
				if (lval) lval[at + q] = vals[ib[j] + e];
				++q;
				low += c < i ? 1 : 0;
				if (c == i) {
					diag = true;
					if (checkMagnitude) {  // |d| >= 1e-5, ref:1691-1693
						const T d = vals[ib[j] + e];
						if ((d < T(0) ? -d : d) < T(1e-5)) bad |= 2;
					}
				}
			}
			st.nlow[i] = static_cast<unsigned short>(low);
			if (!diag) bad |= 1;
			at += cnt[j];
		}
	}
	if (t == 0) st.lptr[nb] = total;
	__syncthreads();
	return bad;
}

// Levels of one sweep by ONE wavefront (call with the first 64 threads): rows in the sweep's natural order, 64 at a time; a lane
// consumes the entries of its row whose level is known, and publishes its own level once all are -- a later pass of the same
// wavefront sees it (LDS is in order for a wavefront).  The first unfinished lane always finishes within a pass: <= 64 passes per
// chunk.  Returns the largest level (>= 2^20: the pass bound tripped).
template <bool LOWER>
__device__ int blkLevels(const BlkStage& st, int nb, unsigned short* lvl, int top) {
	const int lane = threadIdx.x & (WAVE - 1);
	int deepest = 0;
	for (int c = 0; c * WAVE < nb; ++c) {
		const int q = c * WAVE + lane;
		bool pending = q < nb;
		const int i = LOWER ? q : nb - 1 - q;
		int k = 0, kEnd = 0, lv = 0;
		if (pending) {
			if (LOWER) {
				k = st.lptr[i];
				kEnd = k + st.nlow[i];
			} else {
				k = st.lptr[i + 1] - 1;
				kEnd = st.lptr[i] + st.nlow[i];  // the diagonal: entries (kEnd, lptr[i+1]) are the upper part
			}
		}
		for (int pass = 0; pass <= WAVE && __ballot(pending) != 0ull; ++pass) {
			if (pending) {
				while (LOWER ? k < kEnd : k > kEnd) {
					const unsigned short d = lvl[st.lcol[k]];
					if (d == BLK_UNKNOWN) break;
					if (top <= 0 || static_cast<int>(d) < top) lv = max(lv, static_cast<int>(d) + 1);  // (else: the level cut drops this entry)
					k += LOWER ? 1 : -1;
				}
				if (LOWER ? k >= kEnd : k <= kEnd) {
					lvl[i] = static_cast<unsigned short>(lv);
					pending = false;
				}
			}
		}
		if (pending) {  // cannot happen (see above); never leave a row without a level
			lvl[i] = static_cast<unsigned short>(lv);
			deepest = 1 << 20;
		}
		deepest = max(deepest, lv);
	}
#pragma unroll
	for (int o = WAVE / 2; o > 0; o >>= 1) deepest = max(deepest, __shfl_xor(deepest, o, WAVE));
	return deepest;
}

// rows sorted by level (all 256 threads): hist[] is scratch of maxRows + 1 ints; ord[q] = row at sorted position q
__device__ void blkSortByLevel(int nb, const unsigned short* lvl, int* hist, unsigned short* ord, int* scanScratch) {
	constexpr int RPT = BLK_MAX_ROWS / BLK_TPB;
	const int t = threadIdx.x;
	for (int i = t; i <= nb; i += BLK_TPB) hist[i] = 0;
	__syncthreads();
	for (int i = t; i < nb; i += BLK_TPB) atomicAdd(&hist[lvl[i]], 1);
	__syncthreads();
	int h[RPT];
	int mine = 0;
#pragma unroll
	for (int j = 0; j < RPT; ++j) {
		const int i = t * RPT + j;
		h[j] = i < nb ? hist[i] : 0;
		mine += h[j];
	}
	int total = 0;
	int at = blockExclusiveScan256(mine, scanScratch, &total);
#pragma unroll
	for (int j = 0; j < RPT; ++j) {
		const int i = t * RPT + j;
		if (i < nb) hist[i] = at;
		at += h[j];
	}
	__syncthreads();
	for (int i = t; i < nb; i += BLK_TPB) {
		const int q = atomicAdd(&hist[lvl[i]], 1);
		ord[q] = static_cast<unsigned short>(i);
	}
	__syncthreads();
}

// LDS carve-up shared by the two set-up kernels
__device__ __forceinline__ unsigned char* carve(unsigned char*& p, size_t bytes) {
	unsigned char* q = p;
	p += (bytes + 15) & ~static_cast<size_t>(15);
	return q;
}
static size_t carveSize(size_t bytes) { return (bytes + 15) & ~static_cast<size_t>(15); }

// info: [0] most lower entries of a row, [1] most upper entries, [2] error bits (1: empty row / missing diagonal, 2: |d| < 1e-5 for
// SGS), [3] deepest lower sweep, [4] deepest upper sweep (levels), [5] most in-block entries of a block
template <typename T>
__global__ __launch_bounds__(BLK_TPB) void blkAnalyzeKernel(int maxRows, int cap, int top, const int2* __restrict__ bounds, BlkRows br, const int* __restrict__ chunk0,
                                                           const int* __restrict__ start, const int* __restrict__ positions, const T* __restrict__ vals,
                                                           int checkMagnitude, unsigned* __restrict__ metaLo, unsigned* __restrict__ metaUp,
                                                           unsigned short* __restrict__ nEntLo, unsigned short* __restrict__ nEntUp, int* info) {
	extern __shared__ __align__(16) unsigned char blkLds[];
	unsigned char* p = blkLds;
	BlkStage st;
	st.lptr = reinterpret_cast<int*>(carve(p, (maxRows + 1) * sizeof(int)));
	st.ibG = reinterpret_cast<int*>(carve(p, maxRows * sizeof(int)));
	st.nlow = reinterpret_cast<unsigned short*>(carve(p, maxRows * sizeof(unsigned short)));
	st.lcol = reinterpret_cast<unsigned short*>(carve(p, cap * sizeof(unsigned short)));
	unsigned short* lvlLo = reinterpret_cast<unsigned short*>(carve(p, maxRows * sizeof(unsigned short)));
	unsigned short* lvlUp = reinterpret_cast<unsigned short*>(carve(p, maxRows * sizeof(unsigned short)));
	unsigned short* ord = reinterpret_cast<unsigned short*>(carve(p, maxRows * sizeof(unsigned short)));
	int* hist = reinterpret_cast<int*>(carve(p, (maxRows + 1) * sizeof(int)));
	int* scanScratch = reinterpret_cast<int*>(carve(p, 8 * sizeof(int)));
	int* tops = reinterpret_cast<int*>(carve(p, 4 * sizeof(int)));

	const int b = blockIdx.x;
	const int r0 = bounds[b].x;
	const int nb = bounds[b + 1].x - r0;
	const int nc = (nb + WAVE - 1) / WAVE;
	const long long rec0 = static_cast<long long>(chunk0[b]) * WAVE;
	const int t = threadIdx.x;

	const int bad = stageBlock<T>(st, nullptr, br, r0, nb, start, positions, vals, scanScratch, checkMagnitude);
	if (bad) atomicOr(info + 2, bad);
	if (t == 0) atomicMax(info + 5, st.lptr[nb]);  // the most in-block entries of any block: sizes the LDS of the factor / pack kernel
	for (int i = t; i < nb; i += BLK_TPB) {
		lvlLo[i] = BLK_UNKNOWN;
		lvlUp[i] = BLK_UNKNOWN;
	}
	// a missing diagonal makes nlow / the upper part meaningless: the create call fails anyway, skip the analysis
	if (__syncthreads_or(bad & 1) != 0) return;
	if (t < WAVE) {  // two wavefronts, one sweep each, at the same time
		const int deepest = blkLevels<true>(st, nb, lvlLo, top);
		if (t == 0) tops[0] = deepest;
	} else if (t < 2 * WAVE) {
		const int deepest = blkLevels<false>(st, nb, lvlUp, top);
		if (t == WAVE) tops[1] = deepest;
	}
	__syncthreads();
	if (t == 0) {
		if (tops[0] >= (1 << 20) || tops[1] >= (1 << 20)) atomicOr(info + 2, 8);
		atomicMax(info + 3, (tops[0] & 0xFFFFF) + 1);
		atomicMax(info + 4, (tops[1] & 0xFFFFF) + 1);
	}
	int mostLo = 0, mostUp = 0;
	// lower sweep: sorted order, meta, entry counts
	blkSortByLevel(nb, lvlLo, hist, ord, scanScratch);
	for (int q = t; q < nc * WAVE; q += BLK_TPB) {
		unsigned meta;
		int n = 0;
		if (q < nb) {
			const int row = ord[q];
			n = st.nlow[row];
			if (top > 0) {  // the entries the level cut keeps
				n = 0;
				for (int k = st.lptr[row]; k < st.lptr[row] + st.nlow[row]; ++k) n += static_cast<int>(lvlLo[st.lcol[k]]) < top ? 1 : 0;
			}
			meta = static_cast<unsigned>(row) | (static_cast<unsigned>(lvlLo[row]) << 12) | (static_cast<unsigned>(min(n, 255)) << 24);
		} else {
			meta = BLK_NOROW | (static_cast<unsigned>(lvlLo[ord[nb - 1]]) << 12);
		}
		metaLo[rec0 + q] = meta;
		nEntLo[rec0 + q] = static_cast<unsigned short>(n);
		mostLo = max(mostLo, n);
	}
	__syncthreads();
	blkSortByLevel(nb, lvlUp, hist, ord, scanScratch);
	for (int q = t; q < nc * WAVE; q += BLK_TPB) {
		unsigned meta;
		int n = 0;
		if (q < nb) {
			const int row = ord[q];
			n = st.lptr[row + 1] - st.lptr[row] - st.nlow[row] - 1;
			if (top > 0) {
				n = 0;
				for (int k = st.lptr[row] + st.nlow[row] + 1; k < st.lptr[row + 1]; ++k) n += static_cast<int>(lvlUp[st.lcol[k]]) < top ? 1 : 0;
			}
			meta = static_cast<unsigned>(row) | (static_cast<unsigned>(lvlUp[row]) << 12) | (static_cast<unsigned>(min(n, 255)) << 24);
		} else {
			meta = BLK_NOROW | (static_cast<unsigned>(lvlUp[ord[nb - 1]]) << 12);
		}
		metaUp[rec0 + q] = meta;
		nEntUp[rec0 + q] = static_cast<unsigned short>(n);
		mostUp = max(mostUp, n);
	}
#pragma unroll
	for (int o = WAVE / 2; o > 0; o >>= 1) {
		mostLo = max(mostLo, __shfl_xor(mostLo, o, WAVE));
		mostUp = max(mostUp, __shfl_xor(mostUp, o, WAVE));
	}
	if ((t & (WAVE - 1)) == 0) {
		if (mostLo) atomicMax(info + 0, mostLo);
		if (mostUp) atomicMax(info + 1, mostUp);
	}
}

// entries beyond the KREG a record holds
__global__ void blkOverflowCountKernel(long long n, int kreg, const unsigned short* __restrict__ nEnt, int* __restrict__ counts) {
	const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < n) counts[i] = max(0, static_cast<int>(nEnt[i]) - kreg);
	if (i == n) counts[i] = 0;
}

// ILU(0) of one row of the block, by one lane, out of LDS: IKJ on the block's pattern (Saad, Alg. 10.4), the arithmetic of
// the sequential textbook loop: l_ik = a_ik * (1 / u_kk); a_ij -= l_ik * u_kj for the j both rows hold.  Row k (an earlier level)
// is final.  Both rows are ascending, so the search for column j in row i is a merge walk.
template <typename T>
__device__ __forceinline__ bool blkIluRow(const BlkStage& st, T* lval, T* pinv, int i) {
	const int rb = st.lptr[i], re = st.lptr[i + 1];
	const int kd = rb + st.nlow[i];
	for (int q = rb; q < kd; ++q) {
		const int k = st.lcol[q];
		const T lik = lval[q] * pinv[k];
		lval[q] = lik;
		int tgt = q + 1;
		for (int u = st.lptr[k] + st.nlow[k] + 1; u < st.lptr[k + 1]; ++u) {
			const int c = st.lcol[u];
			while (tgt < re && st.lcol[tgt] < c) ++tgt;
			if (tgt < re && st.lcol[tgt] == c) lval[tgt] = lval[tgt] - lik * lval[u];
		}
	}
	const T piv = lval[kd];
	const bool ok = (piv < T(0) ? -piv : piv) >= T(1e-6);
	pinv[i] = ok ? T(1.0) / piv : T(0);
	return ok;
}

// factorise (ILU0) and write the records of both sweeps.  err: bit 2 = zero / tiny pivot.
template <typename T, int KIND, int KREG, bool OV>
__global__ __launch_bounds__(BLK_TPB) void blkPackKernel(int maxRows, int cap, int top, const int2* __restrict__ bounds, BlkRows br, const int* __restrict__ chunk0,
                                                        const int* __restrict__ start, const int* __restrict__ positions, const T* __restrict__ vals,
                                                        const unsigned* __restrict__ metaLo, const unsigned* __restrict__ metaUp, T* __restrict__ lu,
                                                        unsigned* __restrict__ recLo, unsigned* __restrict__ recUp, const int* __restrict__ ovPtrLo,
                                                        const int* __restrict__ ovPtrUp, unsigned short* __restrict__ ovColLo,
                                                        unsigned short* __restrict__ ovColUp, T* __restrict__ ovValLo, T* __restrict__ ovValUp, int* info) {
	extern __shared__ __align__(16) unsigned char blkLds[];
	unsigned char* p = blkLds;
	BlkStage st;
	st.lptr = reinterpret_cast<int*>(carve(p, (maxRows + 1) * sizeof(int)));
	st.ibG = reinterpret_cast<int*>(carve(p, maxRows * sizeof(int)));
	st.nlow = reinterpret_cast<unsigned short*>(carve(p, maxRows * sizeof(unsigned short)));
	st.lcol = reinterpret_cast<unsigned short*>(carve(p, cap * sizeof(unsigned short)));
	T* lval = reinterpret_cast<T*>(carve(p, cap * sizeof(T)));
	T* pinv = reinterpret_cast<T*>(carve(p, maxRows * sizeof(T)));
	int* scanScratch = reinterpret_cast<int*>(carve(p, 8 * sizeof(int)));
	st.nall = reinterpret_cast<unsigned short*>(carve(p, maxRows * sizeof(unsigned short)));
	unsigned short* lvlLo = reinterpret_cast<unsigned short*>(carve(p, maxRows * sizeof(unsigned short)));
	unsigned short* lvlUp = reinterpret_cast<unsigned short*>(carve(p, maxRows * sizeof(unsigned short)));

	constexpr bool ILU = KIND == SMM_PRECOND_BLOCK_ILU0;
	const int b = blockIdx.x;
	const int r0 = bounds[b].x;
	const int nb = bounds[b + 1].x - r0;
	const int nc = (nb + WAVE - 1) / WAVE;
	const long long rec0 = static_cast<long long>(chunk0[b]) * WAVE;
	const int t = threadIdx.x;

	if (top > 0) {  // the levels the analysis kernel found, back by row: the level cut needs them to know what to stage
		for (int q = t; q < nc * WAVE; q += BLK_TPB) {
			const unsigned mlo = metaLo[rec0 + q], mup = metaUp[rec0 + q];
			if ((mlo & 0xFFFu) != BLK_NOROW) lvlLo[mlo & 0xFFFu] = static_cast<unsigned short>((mlo >> 12) & 0xFFFu);
			if ((mup & 0xFFFu) != BLK_NOROW) lvlUp[mup & 0xFFFu] = static_cast<unsigned short>((mup >> 12) & 0xFFFu);
		}
		__syncthreads();
	}
	stageBlock<T>(st, lval, br, r0, nb, start, positions, vals, scanScratch, 0, lvlLo, lvlUp, top);  // (the analysis kernel has already vetted the structure)
	if (ILU) {
		if (t < WAVE) {
			bool allOk = true;
			for (int c = 0; c < nc; ++c) {
				const unsigned meta = metaLo[rec0 + static_cast<long long>(c) * WAVE + t];
				const unsigned row = meta & 0xFFFu;
				const int lvl = static_cast<int>((meta >> 12) & 0xFFFu);
				const bool valid = row != BLK_NOROW;
				const int lFirst = __builtin_amdgcn_readfirstlane(lvl);
				const int lLast = __builtin_amdgcn_readlane(lvl, WAVE - 1);
				for (int lv = lFirst; lv <= lLast; ++lv) {
					if (valid && lvl == lv) allOk = blkIluRow<T>(st, lval, pinv, static_cast<int>(row)) && allOk;
				}
			}
			if (!allOk) atomicOr(info + 2, 4);
		}
		__syncthreads();
		// the factor on A's pattern (smm_hip_precond_values): in-block entries only, the rest keeps A's value
		// (entries of other blocks and entries the level cut dropped keep A's value)
		for (int i = t; i < nb; i += BLK_TPB) {
			const int g = st.ibG[i], l = st.lptr[i], all = st.nall[i];
			int q = 0;
			for (int e = 0; e < all; ++e) {
				const int c = br.posOf(positions[g + e]) - r0;
				if (static_cast<unsigned>(c) < static_cast<unsigned>(nb) && blkKeeps(i, c, lvlLo, lvlUp, top)) lu[g + e] = lval[l + q++];
			}
		}
	}
	using LL = RecLayout<T, !ILU, KREG>;
	using LU = RecLayout<T, true, KREG>;
	for (int q = t; q < nc * WAVE; q += BLK_TPB) {
		const long long g = rec0 + q;
		{  // lower sweep: entries left of the diagonal, ascending
			unsigned w[LL::DW];
#pragma unroll
			for (int i = 0; i < LL::DW; ++i) w[i] = 0u;
			const unsigned meta = metaLo[g];
			w[0] = meta;
			const unsigned row = meta & 0xFFFu;
			if (row != BLK_NOROW) {
				const int rb = st.lptr[row];
				const int n = st.nlow[row];
#pragma unroll
				for (int k = 0; k < KREG; ++k) {
					if (k < n) {
						w[1 + k / 2] |= static_cast<unsigned>(st.lcol[rb + k]) << (16 * (k & 1));
						toWords<T>(lval[rb + k], &w[LL::VAL_AT + LL::VW * k]);
					}
				}
				if (!ILU) toWords<T>(lval[rb + n], &w[LL::DIAG_AT]);
				if (OV && n > KREG) {
					int at = ovPtrLo[g];
					for (int k = KREG; k < n; ++k, ++at) {
						ovColLo[at] = st.lcol[rb + k];
						ovValLo[at] = lval[rb + k];
					}
				}
			}
			unsigned* out = recLo + g * LL::DW;
#pragma unroll
			for (int i = 0; i < LL::DW; ++i) out[i] = w[i];
		}
		{  // upper sweep: entries right of the diagonal, descending
			unsigned w[LU::DW];
#pragma unroll
			for (int i = 0; i < LU::DW; ++i) w[i] = 0u;
			const unsigned meta = metaUp[g];
			w[0] = meta;
			const unsigned row = meta & 0xFFFu;
			if (row != BLK_NOROW) {
				const int re = st.lptr[row + 1];
				const int kd = st.lptr[row] + st.nlow[row];
				const int n = re - kd - 1;
#pragma unroll
				for (int k = 0; k < KREG; ++k) {
					if (k < n) {
						w[1 + k / 2] |= static_cast<unsigned>(st.lcol[re - 1 - k]) << (16 * (k & 1));
						toWords<T>(lval[re - 1 - k], &w[LU::VAL_AT + LU::VW * k]);
					}
				}
				toWords<T>(lval[kd], &w[LU::DIAG_AT]);
				if (OV && n > KREG) {
					int at = ovPtrUp[g];
					for (int k = KREG; k < n; ++k, ++at) {
						ovColUp[at] = st.lcol[re - 1 - k];
						ovValUp[at] = lval[re - 1 - k];
					}
				}
			}
			unsigned* out = recUp + g * LU::DW;
#pragma unroll
			for (int i = 0; i < LU::DW; ++i) out[i] = w[i];
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static size_t analyzeLds(int maxRows, int cap) {
	return carveSize((maxRows + 1) * sizeof(int)) + carveSize(maxRows * sizeof(int)) + carveSize(maxRows * 2) + carveSize(static_cast<size_t>(cap) * 2) +
	       3 * carveSize(maxRows * 2) + carveSize((maxRows + 1) * sizeof(int)) + carveSize(8 * sizeof(int)) + carveSize(4 * sizeof(int));
}
template <typename T>
static size_t packLds(int maxRows, int cap) {
	return carveSize((maxRows + 1) * sizeof(int)) + carveSize(maxRows * sizeof(int)) + carveSize(maxRows * 2) + carveSize(static_cast<size_t>(cap) * 2) +
	       carveSize(static_cast<size_t>(cap) * sizeof(T)) + carveSize(maxRows * sizeof(T)) + carveSize(8 * sizeof(int)) + 3 * carveSize(maxRows * 2);
}

template <typename T, int KIND, int KREG, bool OV>
static int packTyped(const smm_hip_csr* a, smm_hip_precond* M, const unsigned* metaLo, const unsigned* metaUp, int* d_info, hipStream_t s) {
	smm_precond_block* B = M->blk;
	// LDS for what the blocks really hold (entries that couple two blocks are not staged): a 7-point stencil keeps 5 of its 7 entries per
	// row, 70 KB instead of 100 KB per block in fp64 -- two blocks per CU factorise at the same time instead of one
	const int cap = std::max(64, std::min(B->capNnz, (B->maxInBlock + 63) & ~63));
	const size_t lds = packLds<T>(B->blockRows, cap);
	auto kernel = blkPackKernel<T, KIND, KREG, OV>;
	SMM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
	kernel<<<B->nBlocks, BLK_TPB, lds, s>>>(B->blockRows, cap, B->levelCap > 0 ? B->levelCap - 1 : 0, B->d_bounds, BlkRows{B->d_rowOrder, B->d_invOrder}, B->d_chunk0, a->d_start, a->d_positions, static_cast<const T*>(a->d_values),
	                                       metaLo, metaUp, static_cast<T*>(M->d_values), B->d_recLo, B->d_recUp, B->d_ovPtrLo, B->d_ovPtrUp, B->d_ovColLo,
	                                       B->d_ovColUp, static_cast<T*>(B->d_ovValLo), static_cast<T*>(B->d_ovValUp), d_info);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T, int KIND>
static int packKind(const smm_hip_csr* a, smm_hip_precond* M, const unsigned* metaLo, const unsigned* metaUp, int* d_info, hipStream_t s) {
	const smm_precond_block* B = M->blk;
	if (B->overflow) return packTyped<T, KIND, 8, true>(a, M, metaLo, metaUp, d_info, s);
	switch (B->kreg) {
	case 2: return packTyped<T, KIND, 2, false>(a, M, metaLo, metaUp, d_info, s);
	case 3: return packTyped<T, KIND, 3, false>(a, M, metaLo, metaUp, d_info, s);
	case 4: return packTyped<T, KIND, 4, false>(a, M, metaLo, metaUp, d_info, s);
	default: return packTyped<T, KIND, 8, false>(a, M, metaLo, metaUp, d_info, s);
	}
}

static int exclusiveScanInPlace(int* d, size_t n, hipStream_t s) {
	size_t tempBytes = 0;
	SMM_HIP_TRY(rocprim::exclusive_scan(nullptr, tempBytes, d, d, 0, n, rocprim::plus<int>(), s));
	DevBuf<char> temp;
	SMM_TRY(temp.alloc(tempBytes ? tempBytes : 1));
	SMM_HIP_TRY(rocprim::exclusive_scan(temp.p, tempBytes, d, d, 0, n, rocprim::plus<int>(), s));
	SMM_HIP_TRY(hipStreamSynchronize(s));  // temp goes back to the allocator
	return SMM_HIP_OK;
}

// ---- the BRICK partition (r03) ----------------------------------------------------------------------------------------------
// Contiguous row blocks of a 3-D grid in natural order are a few grid LINES of one plane: every coupling to the planes above and below
// is dropped from M and BiCGStab needs 150 iterations on the 108^3 problem where the exact ILU0 needs 67.  When the matrix is a grid
// stencil -- its PATTERN analysis (smm_spmv_pattern.hip) found the offsets {0, +-1, +-nx[, +-nx ny]} -- the blocks are BRICKS of the grid
// instead (16 x 8 x 8 points by default): M keeps 87-92 % of A's entries instead of 65-68 %, the sweeps of a brick are 30 levels deep
// before the level cut, and the same solve takes 95-105 iterations (profiles/r03/brick_blocks.txt).  A brick's rows are not contiguous:
// the handle keeps the row order block by block (ascending inside a block, so the sweeps inside a block follow the natural order) and
// its inverse; the kernels address rows through them.  Matrices of any other shape keep the contiguous cut.
__global__ void brickKeyKernel(int n, int nx, int ny, int bx, int by, int bz, int nbx, int nby, unsigned* __restrict__ keys, int* __restrict__ rows) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int ix = i % nx, iy = (i / nx) % ny, iz = i / (nx * ny);
	keys[i] = static_cast<unsigned>(((iz / bz) * nby + iy / by) * nbx + ix / bx);
	rows[i] = i;
}
__global__ void brickHeadKernel(int n, const unsigned* __restrict__ keys, const int* __restrict__ order, int* __restrict__ inv, int* __restrict__ head) {
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p > n) return;
	if (p == n) {
		head[p] = 0;
		return;
	}
	inv[order[p]] = p;
	head[p] = (p == 0 || keys[p] != keys[p - 1]) ? 1 : 0;
}
__global__ void brickBoundsKernel(int n, const int* __restrict__ headScan, int2* __restrict__ bounds) {
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p > n) return;
	if (p == n) bounds[headScan[n]] = make_int2(n, 0);  // closing entry
	else if (headScan[p + 1] != headScan[p]) bounds[headScan[p]] = make_int2(p, 0);  // p starts block headScan[p]
}

// the grid a stencil matrix lives on, from the offsets of its PATTERN analysis: {0, +-1, +-nx} or {0, +-1, +-nx, +-nx ny}
static bool brickGrid(const smm_hip_csr* a, int* nx, int* ny) {
	if (a->pat_state <= 0 || a->pat_encoding != 0 || a->pat_offs_host.empty()) return false;
	std::vector<int> u;
	for (int o : a->pat_offs_host) {
		const int v = std::abs(o);
		if (v != 0 && std::find(u.begin(), u.end(), v) == u.end()) u.push_back(v);
	}
	std::sort(u.begin(), u.end());
	if (u.size() < 2 || u.size() > 3 || u[0] != 1 || u[1] < 4) return false;
	*nx = u[1];
	*ny = 0;  // 2-D
	if (u.size() == 3) {
		if (u[2] % u[1] != 0 || u[2] / u[1] < 2) return false;
		*ny = u[2] / u[1];
	}
	return true;
}

// SMM_HIP_BLOCK_BRICKS=0: always the contiguous cut
static bool bricksAllowed() {
	static const bool on = [] {
		const char* env = getenv("SMM_HIP_BLOCK_BRICKS");
		return env ? atoi(env) != 0 : true;
	}();
	return on;
}

// fills B->d_rowOrder / d_invOrder / d_bounds / nBlocks / brick[]; returns SMM_HIP_OK with B->d_rowOrder == nullptr when the matrix is
// not a grid stencil (the caller then cuts contiguous blocks)
static int brickPartition(const smm_hip_csr* a, smm_precond_block* B, hipStream_t s) {
	int nx = 0, ny = 0;
	if (!brickGrid(a, &nx, &ny)) return SMM_HIP_OK;
	const int n = a->rows;
	const int rowsMax = std::min(B->blockRows, B->capNnz / std::max(1, a->pat_k));  // a row holds at most pat_k entries (verified)
	int bx, by, bz;
	if (ny == 0) {  // 2-D: squares
		by = bx = std::max(2, static_cast<int>(std::sqrt(static_cast<double>(rowsMax))));
		bx = std::min(bx, nx);
		bz = 1;
		ny = (n + nx - 1) / nx;  // (one "plane": iz = 0 for every row)
		by = std::min(by, ny);
	} else {  // 3-D: rowsMax / 64 x 8 x 8 (16 x 8 x 8 for 1024 rows)
		by = std::min(8, ny);
		bz = 8;
		bx = std::min(nx, std::max(4, rowsMax / (by * bz)));
		while (bx * by * bz > rowsMax && bz > 1) bz /= 2;
		while (bx * by * bz > rowsMax && bx > 1) bx /= 2;
	}
	if (bx * by * bz < 64) return SMM_HIP_OK;  // (degenerate grids: not worth a permutation)
	const int nbx = (nx + bx - 1) / bx, nby = (ny + by - 1) / by;
	DevBuf<unsigned> keys, keysSorted;
	DevBuf<int> rows, head;
	DevBuf<char> temp;
	SMM_TRY(keys.alloc(static_cast<size_t>(n)));
	SMM_TRY(keysSorted.alloc(static_cast<size_t>(n)));
	SMM_TRY(rows.alloc(static_cast<size_t>(n)));
	SMM_TRY(head.alloc(static_cast<size_t>(n) + 1));
	SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_rowOrder), static_cast<size_t>(n) * sizeof(int)));
	SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_invOrder), static_cast<size_t>(n) * sizeof(int)));
	const int grid = (n + 1 + 255) / 256;
	brickKeyKernel<<<grid, 256, 0, s>>>(n, nx, ny, bx, by, bz, nbx, nby, keys, rows);
	size_t tempBytes = 0;  // (LSD radix sort: stable, so the rows of a block stay in ascending order)
	SMM_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tempBytes, keys.p, keysSorted.p, rows.p, B->d_rowOrder, static_cast<size_t>(n), 0, 32, s));
	SMM_TRY(temp.alloc(tempBytes ? tempBytes : 1));
	SMM_HIP_TRY(rocprim::radix_sort_pairs(temp.p, tempBytes, keys.p, keysSorted.p, rows.p, B->d_rowOrder, static_cast<size_t>(n), 0, 32, s));
	brickHeadKernel<<<grid, 256, 0, s>>>(n, keysSorted, B->d_rowOrder, B->d_invOrder, head);
	SMM_TRY(exclusiveScanInPlace(head, static_cast<size_t>(n) + 1, s));  // head[p] = blocks that start before position p; drains s
	int nBlocks = 0;
	SMM_HIP_TRY(hipMemcpyAsync(&nBlocks, head.p + n, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_bounds), (static_cast<size_t>(nBlocks) + 1) * sizeof(int2)));
	brickBoundsKernel<<<grid, 256, 0, s>>>(n, head, B->d_bounds);
	SMM_HIP_TRY(hipGetLastError());
	SMM_HIP_TRY(hipStreamSynchronize(s));  // the scratch buffers go back to the allocator when this scope ends
	B->nBlocks = nBlocks;
	B->brick[0] = bx;
	B->brick[1] = by;
	B->brick[2] = bz;
	return SMM_HIP_OK;
}

template <typename T>
int blockCreateTyped(const smm_hip_csr* a, int kind, int blockRows, int levelCap, int partition, smm_hip_precond* M) {
	hipStream_t s = libStream();
	SMM_HIP_TRY(hipDeviceSynchronize());  // the matrix may still be being written on a caller's stream
	const int n = a->rows;
	if (a->rows != a->cols) {
		setError("preconditioner needs a square matrix");
		return SMM_HIP_ERR_INVALID;
	}
	if (a->firstActiveStart != 0 && n > 0) {  // ref:1666-1670
		setError("preconditioner: matrix has leading empty rows (firstActiveStart != 0)");
		return SMM_HIP_ERR_PRECOND;
	}
	auto* B = new smm_precond_block();
	M->blk = B;  // (smm_hip_precond_destroy releases whatever the steps below have allocated so far)
	B->blockRows = blockRows;
	B->levelCap = levelCap < 2 ? 0 : std::min(levelCap, 4095);  // (a cap of 1 would drop every coupling: 0 / 1 mean no cut)
	B->capNnz = BLK_CAP_NNZ;
	if (n == 0) return SMM_HIP_OK;
	if (partition != SMM_BLOCKS_CONTIGUOUS && (partition == SMM_BLOCKS_BRICKS || bricksAllowed())) {
		// the grid is read from the matrix's PATTERN analysis (run here, quietly, when no SpMV has asked for it yet)
		const int st = ensurePattern(const_cast<smm_hip_csr*>(a), s, true, true, true);
		if (st != SMM_HIP_OK && st != SMM_HIP_ERR_INVALID) return st;
		SMM_TRY(brickPartition(a, B, s));
		// a preconditioner is built for a solver: let the matrix take the compressed SpMV family now (the analysis above has already
		// run; the solvers' own request would find nothing left to analyse)
		SMM_TRY(adoptPatternForSolver(a, -1, s));
	}
	if (!B->d_rowOrder) {
		if (partition == SMM_BLOCKS_BRICKS) {
			setError("block preconditioner: bricks were asked for, but the matrix is not a 2-D / 3-D grid stencil with offsets {0, +-1, +-nx[, +-nx ny]}");
			return SMM_HIP_ERR_INVALID;
		}
		// (seams every 16 blocks instead of every 64: the greedy cut of a super-chunk is a sequential chain of binary searches by one thread)
		SMM_TRY(cutRows(a->d_start, n, a->nnz, B->capNnz, B->blockRows, s, &B->d_bounds, &B->nBlocks, 16));
	}
	const int nBlocks = B->nBlocks;
	SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_chunk0), (static_cast<size_t>(nBlocks) + 1) * sizeof(int)));
	blkChunkCountKernel<<<(nBlocks + 1 + 255) / 256, 256, 0, s>>>(nBlocks, B->d_bounds, B->d_chunk0);
	SMM_TRY(exclusiveScanInPlace(B->d_chunk0, static_cast<size_t>(nBlocks) + 1, s));
	int nChunks = 0;
	SMM_HIP_TRY(hipMemcpyAsync(&nChunks, B->d_chunk0 + nBlocks, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	B->nChunks = nChunks;
	const size_t nRec = static_cast<size_t>(nChunks) * WAVE;

	DevBuf<unsigned> metaLo, metaUp;
	DevBuf<unsigned short> nEntLo, nEntUp;
	DevBuf<int> info;
	SMM_TRY(metaLo.alloc(nRec));
	SMM_TRY(metaUp.alloc(nRec));
	SMM_TRY(nEntLo.alloc(nRec));
	SMM_TRY(nEntUp.alloc(nRec));
	SMM_TRY(info.alloc(8));
	SMM_HIP_TRY(hipMemsetAsync(info, 0, 8 * sizeof(int), s));
	{
		const size_t lds = analyzeLds(B->blockRows, B->capNnz);
		auto kernel = blkAnalyzeKernel<T>;
		SMM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
		kernel<<<nBlocks, BLK_TPB, lds, s>>>(B->blockRows, B->capNnz, B->levelCap > 0 ? B->levelCap - 1 : 0, B->d_bounds, BlkRows{B->d_rowOrder, B->d_invOrder}, B->d_chunk0, a->d_start, a->d_positions, static_cast<const T*>(a->d_values),
		                                    kind == SMM_PRECOND_BLOCK_SGS ? 1 : 0, metaLo, metaUp, nEntLo, nEntUp, info);
		SMM_HIP_TRY(hipGetLastError());
	}
	int h[8] = {0};
	SMM_HIP_TRY(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	if (h[2] & 8) {
		setError("block preconditioner: the level analysis of a block did not finish");
		return SMM_HIP_ERR_HIP;
	}
	if (h[2]) {
		setError("block preconditioner: empty row, missing diagonal or |d|<1e-5");
		return SMM_HIP_ERR_PRECOND;
	}
	B->levelsLo = h[3];
	B->levelsUp = h[4];
	B->maxInBlock = h[5];
	const int most = std::max(h[0], h[1]);
	B->kreg = most <= 2 ? 2 : most <= 3 ? 3 : most <= 4 ? 4 : 8;  // (3: the rows of a 7-point stencil inside a brick)
	B->overflow = most > 8;
	if (const char* env = getenv("SMM_HIP_BLOCK_KREG")) {  // testing: force the overflow path on matrices with short rows
		const int k = atoi(env);
		if (k == 2 || k == 3 || k == 4 || k == 8) {
			B->kreg = std::max(B->kreg, k);
		} else if (k == -8) {
			B->kreg = 8;
			B->overflow = true;
		}
	}
	if (B->overflow) {
		SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_ovPtrLo), (nRec + 1) * sizeof(int)));
		SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_ovPtrUp), (nRec + 1) * sizeof(int)));
		const int grid = static_cast<int>((nRec + 1 + 255) / 256);
		blkOverflowCountKernel<<<grid, 256, 0, s>>>(static_cast<long long>(nRec), B->kreg, nEntLo, B->d_ovPtrLo);
		blkOverflowCountKernel<<<grid, 256, 0, s>>>(static_cast<long long>(nRec), B->kreg, nEntUp, B->d_ovPtrUp);
		SMM_TRY(exclusiveScanInPlace(B->d_ovPtrLo, nRec + 1, s));
		SMM_TRY(exclusiveScanInPlace(B->d_ovPtrUp, nRec + 1, s));
		int tot[2] = {0, 0};
		SMM_HIP_TRY(hipMemcpyAsync(&tot[0], B->d_ovPtrLo + nRec, sizeof(int), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipMemcpyAsync(&tot[1], B->d_ovPtrUp + nRec, sizeof(int), hipMemcpyDeviceToHost, s));
		SMM_HIP_TRY(hipStreamSynchronize(s));
		SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_ovColLo), std::max<size_t>(1, tot[0]) * sizeof(unsigned short)));
		SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_ovColUp), std::max<size_t>(1, tot[1]) * sizeof(unsigned short)));
		SMM_TRY(devAlloc(&B->d_ovValLo, std::max<size_t>(1, tot[0]) * sizeof(T)));
		SMM_TRY(devAlloc(&B->d_ovValUp, std::max<size_t>(1, tot[1]) * sizeof(T)));
	}
	const bool ilu = kind == SMM_PRECOND_BLOCK_ILU0;
	size_t dwLo = 0, dwUp = 0;
	switch (B->kreg) {
	case 2:
		dwLo = ilu ? RecLayout<T, false, 2>::DW : RecLayout<T, true, 2>::DW;
		dwUp = RecLayout<T, true, 2>::DW;
		break;
	case 3:
		dwLo = ilu ? RecLayout<T, false, 3>::DW : RecLayout<T, true, 3>::DW;
		dwUp = RecLayout<T, true, 3>::DW;
		break;
	case 4:
		dwLo = ilu ? RecLayout<T, false, 4>::DW : RecLayout<T, true, 4>::DW;
		dwUp = RecLayout<T, true, 4>::DW;
		break;
	default:
		dwLo = ilu ? RecLayout<T, false, 8>::DW : RecLayout<T, true, 8>::DW;
		dwUp = RecLayout<T, true, 8>::DW;
		break;
	}
	SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_recLo), std::max<size_t>(1, nRec * dwLo) * sizeof(unsigned)));
	SMM_TRY(devAlloc(reinterpret_cast<void**>(&B->d_recUp), std::max<size_t>(1, nRec * dwUp) * sizeof(unsigned)));
	if (ilu) {
		const size_t nnz = static_cast<size_t>(a->nnz);
		SMM_TRY(devAlloc(&M->d_values, std::max<size_t>(1, nnz) * sizeof(T)));
		M->n_values = nnz;
		if (nnz) SMM_HIP_TRY(hipMemcpyAsync(M->d_values, a->d_values, nnz * sizeof(T), hipMemcpyDeviceToDevice, s));
		SMM_TRY((packKind<T, SMM_PRECOND_BLOCK_ILU0>(a, M, metaLo, metaUp, info, s)));
	} else {
		SMM_TRY((packKind<T, SMM_PRECOND_BLOCK_SGS>(a, M, metaLo, metaUp, info, s)));
	}
	SMM_HIP_TRY(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));  // also: the scratch buffers go back to the allocator when this scope ends
	if (h[2] & 4) {
		setError("block ilu0: zero / tiny pivot (reordering would be needed, ref:1741-1746)");
		return SMM_HIP_ERR_PRECOND;
	}
	return SMM_HIP_OK;
}

template int blockCreateTyped<float>(const smm_hip_csr*, int, int, int, int, smm_hip_precond*);
template int blockCreateTyped<double>(const smm_hip_csr*, int, int, int, int, smm_hip_precond*);

template <typename T, int KIND, int KREG, bool OV>
static int launchBlkApply(const smm_hip_precond* M, const BlkApplyArgs<T>& args, hipStream_t s) {
	const smm_precond_block* B = M->blk;
	// chunks in flight per sweep (SMM_HIP_BLOCK_DEPTH = 1 / 2 / 4 / 8 overrides where compiled: measurements)
	static const int depth = [] {
		const char* env = getenv("SMM_HIP_BLOCK_DEPTH");
		return env ? atoi(env) : 0;
	}();
	const int grid = std::min(B->nBlocks, NPART);
	const size_t lds = static_cast<size_t>(B->blockRows) * sizeof(T);
	if (args.spmvX) {
		if (B->nBlocks > NPART) {
			setError("precond_apply: the fused SpMV serves at most %d blocks", NPART);
			return SMM_HIP_ERR_INVALID;
		}
		const size_t ldsSpmv = lds + (args.patMasks ? BLK_MAXOFF * (sizeof(T) + sizeof(int)) : 0);
		blkApplyKernel<T, KIND, KREG, OV, 2, true><<<grid, BLK_SPMV_TPB, ldsSpmv, s>>>(args);
		SMM_HIP_TRY(hipGetLastError());
		return SMM_HIP_OK;
	}
	if constexpr (KREG <= 2) {
		switch (depth) {
		case 1: blkApplyKernel<T, KIND, KREG, OV, 1><<<grid, WAVE, lds, s>>>(args); break;
		case 2: blkApplyKernel<T, KIND, KREG, OV, 2><<<grid, WAVE, lds, s>>>(args); break;
		case 8: blkApplyKernel<T, KIND, KREG, OV, 8><<<grid, WAVE, lds, s>>>(args); break;
		case 4: blkApplyKernel<T, KIND, KREG, OV, 4><<<grid, WAVE, lds, s>>>(args); break;
		default: blkApplyKernel<T, KIND, KREG, OV, 2><<<grid, WAVE, lds, s>>>(args); break;
		}
	} else {
		blkApplyKernel<T, KIND, KREG, OV, 2><<<grid, WAVE, lds, s>>>(args);
	}
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T, int KIND>
static int launchBlkApplyKind(const smm_hip_precond* M, const BlkApplyArgs<T>& args, hipStream_t s) {
	const smm_precond_block* B = M->blk;
	if (B->overflow) return launchBlkApply<T, KIND, 8, true>(M, args, s);
	switch (B->kreg) {
	case 2: return launchBlkApply<T, KIND, 2, false>(M, args, s);
	case 3: return launchBlkApply<T, KIND, 3, false>(M, args, s);
	case 4: return launchBlkApply<T, KIND, 4, false>(M, args, s);
	default: return launchBlkApply<T, KIND, 8, false>(M, args, s);
	}
}

// the matrix is in the row-mask encoding (every pat_* field is final once the state reads 1: smm_internal.h) ...
static bool blkMaskForm(const smm_hip_csr* A) {
	return A && A->pat_state.load(std::memory_order_acquire) == 1 && A->pat_encoding == 0 && A->d_pat_masks && A->d_pat_off && A->pat_k > 0 &&
	       A->pat_k <= BLK_MAXOFF;
}
// ... with constant diagonals on top
static bool blkConstForm(const smm_hip_csr* A) { return blkMaskForm(A) && A->pat_const && !A->pat_const_off && A->d_pat_cval; }

// x = M^-1 rhs with the dot products of x fused into the epilogue (dotMode as in launchSpmv; partials: 2 * NPART elements).
// spmvOf != nullptr: rhs is not read -- the right-hand side is A spmvOf (A = the matrix M was created for), formed inside the launch
// (ref:2234-2235 and ref:2250-2251 in one launch each: no SpMV launch, no pass of A p / A s through memory)
template <typename T>
static int blockApplyAny(const smm_hip_precond* M, const T* rhs, const T* spmvOf, T* x, int dotMode, const T* w1, T* partials, const int* doneFlag, hipStream_t s) {
	const smm_precond_block* B = M->blk;
	if (!B) {
		setError("precond_apply: block preconditioner without its tables");
		return SMM_HIP_ERR_INVALID;
	}
	if (B->nBlocks == 0) return SMM_HIP_OK;
	BlkApplyArgs<T> args;
	args.nBlocks = B->nBlocks;
	args.bounds = B->d_bounds;
	args.rowOrder = B->d_rowOrder;
	args.chunk0 = B->d_chunk0;
	args.recLo = B->d_recLo;
	args.recUp = B->d_recUp;
	args.ovPtrLo = B->d_ovPtrLo;
	args.ovPtrUp = B->d_ovPtrUp;
	args.ovColLo = B->d_ovColLo;
	args.ovColUp = B->d_ovColUp;
	args.ovValLo = static_cast<const T*>(B->d_ovValLo);
	args.ovValUp = static_cast<const T*>(B->d_ovValUp);
	args.rhs = rhs;
	args.x = x;
	args.aStart = spmvOf ? M->a->d_start : nullptr;
	args.aPos = spmvOf ? M->a->d_positions : nullptr;
	args.aVal = spmvOf ? static_cast<const T*>(M->a->d_values) : nullptr;
	args.spmvX = spmvOf;
	// the constant-diagonal encoding, where the matrix is in it (every pat_* field is final once the state reads 1: smm_internal.h)
	const smm_hip_csr* A = M->a;
	const bool maskForm = spmvOf && blkMaskForm(A);
	args.patMasks = maskForm ? A->d_pat_masks : nullptr;
	args.patOff = maskForm ? A->d_pat_off : nullptr;
	args.patCval = maskForm && blkConstForm(A) ? A->d_pat_cval : nullptr;  // (nullptr with masks: the values are read)
	args.patK = maskForm ? A->pat_k : 0;
	args.cols = A->cols;
	args.ldsRows = B->blockRows;
	args.dotMode = dotMode;
	args.w1 = w1;
	args.partials = partials;
	args.doneFlag = doneFlag;
	if (M->kind == SMM_PRECOND_BLOCK_ILU0) return launchBlkApplyKind<T, SMM_PRECOND_BLOCK_ILU0>(M, args, s);
	return launchBlkApplyKind<T, SMM_PRECOND_BLOCK_SGS>(M, args, s);
}

template <typename T>
int blockApplyDev(const smm_hip_precond* M, const T* rhs, T* x, int dotMode, const T* w1, T* partials, const int* doneFlag, hipStream_t s) {
	return blockApplyAny<T>(M, rhs, nullptr, x, dotMode, w1, partials, doneFlag, s);
}

// x = M^-1 (A v) in one launch; v and x must not overlap (other blocks read v while this one writes x)
template <typename T>
int blockApplySpmvDev(const smm_hip_precond* M, const T* v, T* x, int dotMode, const T* w1, T* partials, const int* doneFlag, hipStream_t s) {
	if (!M->a || !M->a->d_start || !M->a->d_positions || !M->a->d_values || M->a->nnz <= 0 || !v) {
		setError("precond_apply: the fused SpMV needs the matrix the block preconditioner was created for");
		return SMM_HIP_ERR_INVALID;
	}
	return blockApplyAny<T>(M, nullptr, v, x, dotMode, w1, partials, doneFlag, s);
}

// Whether x = M^-1 (A v) runs as ONE launch.  asked = true: the caller asked for exactly that operator (smm_hip_precond_apply_spmv):
// yes wherever the launch exists.  asked = false: a solver loop choosing between one launch and SpMV + apply -- measured, fp64, per
// BiCGStab pass (profiles/r06/block_spmv_inside_apply*.txt; 108^3: 1372 blocks, 2-D Poisson 1000^2: 1024 blocks):
//   * the matrix in the row-mask encoding: always.  Constant diagonals (a row of A v costs its mask and the gathers): 153 -> 134 us and
//     129 -> 104; values read (mask and start[] together, then values[] and the gathers together): 180 -> 175 and 146 -> 118;
//   * the rows read from start[] / positions[] / values[] (a trip to memory more in front of every sweep, 12 bytes per entry): where
//     every block of a CU starts at once, i.e. at most 4 blocks per CU (2-D Poisson: 129 -> 122; 108^3: 153 -> 187, the workgroups of a
//     CU start in turns).
// SMM_HIP_BLOCK_FUSE_SPMV=0 / 1 (read per call) forces either.
bool blockFuseSpmv(const smm_hip_precond* M, bool asked) {
	if (!M || !M->blk || M->blk->nBlocks > NPART) return false;  // (one workgroup per block: the helper wavefronts leave early)
	if (M->blk->nBlocks == 0 || !M->a || M->a->nnz <= 0) return false;  // (nothing to multiply: the plain apply knows what to do)
	const char* env = getenv("SMM_HIP_BLOCK_FUSE_SPMV");
	if (env && env[0] == '0') return false;
	if (asked || (env && env[0] == '1')) return true;
	return blkMaskForm(M->a) || M->blk->nBlocks <= 4 * numCUs();
}

template int blockApplySpmvDev<float>(const smm_hip_precond*, const float*, float*, int, const float*, float*, const int*, hipStream_t);
template int blockApplySpmvDev<double>(const smm_hip_precond*, const double*, double*, int, const double*, double*, const int*, hipStream_t);
template int blockApplyDev<float>(const smm_hip_precond*, const float*, float*, int, const float*, float*, const int*, hipStream_t);
template int blockApplyDev<double>(const smm_hip_precond*, const double*, double*, int, const double*, double*, const int*, hipStream_t);

void blockDestroy(smm_precond_block* B) {
	if (!B) return;
	devFree(B->d_bounds);
	devFree(B->d_rowOrder);
	devFree(B->d_invOrder);
	devFree(B->d_chunk0);
	devFree(B->d_recLo);
	devFree(B->d_recUp);
	devFree(B->d_ovPtrLo);
	devFree(B->d_ovPtrUp);
	devFree(B->d_ovColLo);
	devFree(B->d_ovColUp);
	devFree(B->d_ovValLo);
	devFree(B->d_ovValUp);
	delete B;
}

void blockLevels(const smm_precond_block* B, int* lo, int* up) {
	if (lo) *lo = B ? B->levelsLo : 0;
	if (up) *up = B ? B->levelsUp : 0;
}

// SMM_HIP_BLOCK_LEVEL_CAP: the default level cut of a block preconditioner (0 = none)
int blockDefaultLevelCap() {
	int cap = BLK_DEFAULT_LEVEL_CAP;
	if (const char* env = getenv("SMM_HIP_BLOCK_LEVEL_CAP")) cap = atoi(env);
	return cap < 2 ? 0 : std::min(cap, 4095);
}

int blockDefaultRows() {
	int rows = BLK_DEFAULT_ROWS;
	if (const char* env = getenv("SMM_HIP_BLOCK_ROWS")) rows = atoi(env);
	return std::max(BLK_MIN_ROWS, std::min(BLK_MAX_ROWS, rows));
}

}  // namespace smm

using namespace smm;

extern "C" {

int smm_hip_precond_block_count(const smm_hip_precond* M, int* nblocks) {
	if (!M || !M->blk || !nblocks) {
		setError("precond_block_count: not a block preconditioner");
		return SMM_HIP_ERR_INVALID;
	}
	*nblocks = M->blk->nBlocks;
	return SMM_HIP_OK;
}

int smm_hip_precond_block_rows(const smm_hip_precond* M, int* order, size_t count, int* brick) {
	if (!M || !M->blk) {
		setError("precond_block_rows: not a block preconditioner");
		return SMM_HIP_ERR_INVALID;
	}
	const smm_precond_block* B = M->blk;
	const size_t n = M->a ? static_cast<size_t>(M->a->rows) : 0;
	if (brick) {
		brick[0] = B->brick[0];
		brick[1] = B->brick[1];
		brick[2] = B->brick[2];
	}
	if (!order) return SMM_HIP_OK;
	if (count != n) {
		setError("precond_block_rows: the matrix has %zu rows", n);
		return SMM_HIP_ERR_INVALID;
	}
	if (!B->d_rowOrder) {
		for (size_t i = 0; i < n; ++i) order[i] = static_cast<int>(i);
		return SMM_HIP_OK;
	}
	SMM_TRY(ensureInit());
	hipStream_t s = libStream();
	SMM_HIP_TRY(hipMemcpyAsync(order, B->d_rowOrder, n * sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

// bytes one apply moves per row, apart from the vectors: the record of the lower and of the upper sweep, the row-order entries (bricks)
int smm_hip_precond_block_record_bytes(const smm_hip_precond* M, int* lower, int* upper, int* order) {
	if (!M || !M->blk) {
		setError("precond_block_record_bytes: not a block preconditioner");
		return SMM_HIP_ERR_INVALID;
	}
	const smm_precond_block* B = M->blk;
	const bool f32 = M->dtype == SMM_DTYPE_F32;
	const bool ilu = M->kind == SMM_PRECOND_BLOCK_ILU0;
	auto dw = [&](bool hasDiag) {  // RecLayout<T, hasDiag, kreg>::DW
		const int vw = f32 ? 1 : 2;
		const int raw = 1 + (B->kreg + 1) / 2 + (hasDiag ? vw : 0) + vw * B->kreg;
		return (raw + 1) & ~1;
	};
	if (lower) *lower = 4 * dw(!ilu);
	if (upper) *upper = 4 * dw(true);
	if (order) *order = B->d_rowOrder ? 8 : 0;  // read once for rhs, once for x
	return SMM_HIP_OK;
}

int smm_hip_precond_block_level_cap(const smm_hip_precond* M, int* level_cap) {
	if (!M || !M->blk || !level_cap) {
		setError("precond_block_level_cap: not a block preconditioner");
		return SMM_HIP_ERR_INVALID;
	}
	*level_cap = M->blk->levelCap;
	return SMM_HIP_OK;
}

int smm_hip_precond_block_bounds(const smm_hip_precond* M, int* bounds, size_t count) {
	if (!M || !M->blk || !bounds) {
		setError("precond_block_bounds: not a block preconditioner");
		return SMM_HIP_ERR_INVALID;
	}
	smm_precond_block* B = M->blk;
	if (count != static_cast<size_t>(B->nBlocks) + 1) {
		setError("precond_block_bounds: %d blocks need %d entries", B->nBlocks, B->nBlocks + 1);
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	std::lock_guard<std::mutex> lock(B->boundsMutex);
	if (B->hostBounds.empty()) {
		if (B->nBlocks == 0) {
			B->hostBounds.assign(1, 0);
		} else {
			std::vector<int2> tmp(static_cast<size_t>(B->nBlocks) + 1);
			hipStream_t s = libStream();
			SMM_HIP_TRY(hipMemcpyAsync(tmp.data(), B->d_bounds, tmp.size() * sizeof(int2), hipMemcpyDeviceToHost, s));
			SMM_HIP_TRY(hipStreamSynchronize(s));
			B->hostBounds.resize(tmp.size());
			for (size_t i = 0; i < tmp.size(); ++i) B->hostBounds[i] = tmp[i].x;
		}
	}
	std::copy(B->hostBounds.begin(), B->hostBounds.end(), bounds);
	return SMM_HIP_OK;
}

}  // extern "C"
