// smm_spmv_pattern.hip -- SpMV for CSR matrices whose rows all draw their columns from ONE small set of offsets:
//     positions[k] - row  in  { off[0] < off[1] < ... < off[K-1] },  K <= 64
// i.e. every stencil / banded matrix (the Laplacians, convection-diffusion and banded-random matrices of BASELINE.json).
//
// For such a matrix positions[] is redundant: row i's columns are i + off[j] for the bits j set in a 64-bit row mask.  The
// masks (8 bytes per ROW instead of 4 bytes per NONZERO) are built once and verified against every entry of positions[] on
// the device; the kernel then streams only values[] -- for fp32 that halves the bytes of an SpMV.  The result is the same
// number as the generic kernels bit for bit (same products, same left-to-right order with one lane per row).
//
// Selected explicitly (smm_hip_csr_set_kernel(m, SMM_SPMV_PATTERN, lanes): fails with SMM_HIP_ERR_INVALID when the matrix has no such
// pattern) or by AUTO: the first SpMV of a LARGE matrix (>= 2^25 stored entries, rows of <= 64 entries; launchSpmv) runs the analysis
// below on the caller's stream -- offsets discovered from a sample of rows on the device, then EVERY entry verified -- and switches the
// matrix to this family when it passes; a matrix that does not fit (i.i.d. columns ...) stays with STREAM.  SMM_HIP_AUTO_PATTERN=0
// turns the automatic choice off.  bench.py prices every kernel with the bytes ITS layout moves (smm_hip_csr_kernel_desc): `roofline` is
// the timed region's kernel -- this family's for the benchmark matrix --, `roofline_csr` the reference's layout (values + positions +
// start) on the STREAM kernel, measured in a leg of its own.
//
// When every diagonal of such a matrix holds ONE value (constant-coefficient stencils: the Laplacians) values[] is redundant too: the
// CONST encoding further down reads the row's mask, x and <= 32 numbers.  Grid-shaped matrices of some millions of rows -- offsets = a few near
// ones plus the pair -P / +P -- run the 2.5-D kernels of smm_spmv_march.hip instead of the gather kernels of this file: x through a
// plane's LDS window and the lane's registers (r04).
// Matrices the masks cannot describe (more than 64 offsets, rows of more than 64 entries) but whose entries use <= 65 536 distinct
// offsets get the CODES encoding instead: a 16-bit dictionary index per entry (further down: "DICTIONARY encoding").
//
// Kernel structure = spmvStreamKernel (smm_spmv.hip): persistent workgroups walk row tiles, values[] is fetched one tile
// ahead with 16-byte non-temporal loads into registers, stored to LDS, and lane (row, piece) walks its piece of its row in
// batches of 8 independent x[] gathers; the column of an entry comes from the next set bit of the row mask.
#include <algorithm>
#include <cmath>

#include <rocprim/device/device_radix_sort.hpp>

#include "smm_device.h"
#include "smm_internal.h"
#include "smm_pattern_dev.h"

namespace smm {

// ---- analysis: row masks + verification of every entry ----------------------------------------------------------------------
// One WAVEFRONT per 64 consecutive rows: their entries are one contiguous range of positions[], read coalesced (r02 gave every lane a
// row of its own: each load instruction touched 64 cache lines -- 4.6 ms for the 1.9 GB of the benchmark matrix).  An entry finds its
// row by a binary search over the 65 row starts of the group (LDS), its offset index by a binary search over the offset list (LDS), and
// sets its bit in the row's mask with an LDS atomic.  Verified on the way: the offset is in the list; columns ascend strictly inside a
// row (ref:1247-1249: the kernels pair the n-th value of a row with the n-th set bit); no two entries of a row share a bit.
// Constant diagonals (values != nullptr and the sample has not already said no -- flags[1]): every entry's value is compared, bit for
// bit, with the one value its offset was given (cvalBits, from the sampled rows); any difference raises flags[1].
// `meta` = {number of offsets, refusal code} as patSortOffsets left them on the device: the whole analysis is ONE enqueue (r04) and the
// host learns k with the same read-back as the verdicts; a refused matrix (meta[1] != 0) costs these kernels one load.
__global__ __launch_bounds__(TPB) void patBuildMasks(int rows, const int* __restrict__ meta, const int* __restrict__ offs, const int* __restrict__ start,
                                                     const int* __restrict__ positions, unsigned long long* __restrict__ masks,
                                                     int* __restrict__ flags, const void* __restrict__ values, int elemBytes,
                                                     const unsigned long long* __restrict__ cvalBits) {
	if (meta[1] != 0) return;
	const int k = meta[0];
	if (values != nullptr && k > 32) values = nullptr;  // constant diagonals are looked for in matrices of stencil shape only (<= 32 offsets)
	int* mismatch = flags;
	__shared__ int sOff[MAXOFF];
	__shared__ unsigned long long sCval[MAXOFF];
	const bool checkConst = values != nullptr && flags[1] == 0;
	if (checkConst && threadIdx.x < k) sCval[threadIdx.x] = cvalBits[threadIdx.x];
	bool varies = false;
	__shared__ int sRow[TPB / WAVE][WAVE + 1];
	__shared__ unsigned sLo[TPB / WAVE][WAVE], sHi[TPB / WAVE][WAVE];
	if (threadIdx.x < k) sOff[threadIdx.x] = offs[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & (WAVE - 1);
	const int w = threadIdx.x >> 6;
	const long long groups = (static_cast<long long>(rows) + WAVE - 1) / WAVE;
	bool bad = false;
	for (long long g = static_cast<long long>(blockIdx.x) * (TPB / WAVE) + w; g < groups; g += static_cast<long long>(gridDim.x) * (TPB / WAVE)) {
		const int r0 = static_cast<int>(g * WAVE);
		const int nr = min(WAVE, rows - r0);
		sRow[w][lane] = start[r0 + min(lane, nr)];
		if (lane == 0) sRow[w][WAVE] = start[r0 + nr];
		sLo[w][lane] = 0u;
		sHi[w][lane] = 0u;
		const int eBegin = __builtin_amdgcn_readfirstlane(sRow[w][0]);
		const int eEnd = start[r0 + nr];
		int carryCol = -1, carryRow = -1;  // the last entry of the previous sweep of 64 (lane 0's predecessor)
		for (int e0 = eBegin; e0 < eEnd; e0 += WAVE) {
			const int e = e0 + lane;
			const bool ok = e < eEnd;
			int col = -1, i = -1;
			if (ok) {
				col = positions[e];
				int lo = 0, hi = nr;  // largest i in [0, nr) with sRow[i] <= e  (empty rows: the LAST such row owns the entry)
				while (hi - lo > 1) {
					const int mid = (lo + hi) >> 1;
					if (sRow[w][mid] <= e) lo = mid; else hi = mid;
				}
				i = lo;
				const int rel = col - (r0 + i);
				int a = 0, b = k;  // first index with sOff >= rel
				while (a < b) {
					const int mid = (a + b) >> 1;
					if (sOff[mid] < rel) a = mid + 1; else b = mid;
				}
				if (a >= k || sOff[a] != rel) {
					bad = true;
				} else {
					const unsigned bit = 1u << (a & 31);
					const unsigned before = a < 32 ? atomicOr(&sLo[w][i], bit) : atomicOr(&sHi[w][i], bit);
					if (before & bit) bad = true;  // two entries of a row on one diagonal
					if (checkConst) {
						const unsigned long long bits = elemBytes == 4 ? static_cast<unsigned long long>(static_cast<const unsigned*>(values)[e])
						                                               : static_cast<const unsigned long long*>(values)[e];
						if (bits != sCval[a]) varies = true;
					}
				}
			}
			int prevCol = __shfl_up(col, 1, WAVE), prevRow = __shfl_up(i, 1, WAVE);
			if (lane == 0) {
				prevCol = carryCol;
				prevRow = carryRow;
			}
			if (ok && prevRow == i && col <= prevCol) bad = true;  // columns must ascend strictly inside a row
			carryCol = __shfl(col, WAVE - 1, WAVE);
			carryRow = __shfl(i, WAVE - 1, WAVE);
		}
		if (lane < nr) {
			const unsigned long long m = static_cast<unsigned long long>(sLo[w][lane]) | (static_cast<unsigned long long>(sHi[w][lane]) << 32);
			masks[r0 + lane] = m;
			if (__popcll(m) != sRow[w][lane + 1] - sRow[w][lane]) bad = true;
		}
	}
	if (bad) atomicOr(mismatch, 1);
	if (varies) atomicOr(flags + 1, 1);
}

// One value per offset from the sampled rows (mode 0: plain stores -- any of them will do, the full check follows), then (mode 1) the same
// rows compared with it: a matrix whose diagonals vary inside the sample is told apart here, and patBuildMasks never reads values[].
__global__ __launch_bounds__(TPB) void patConstSample(int rows, int samples, const int* __restrict__ meta, const int* __restrict__ offs,
                                                      const int* __restrict__ start, const int* __restrict__ positions, const void* __restrict__ values,
                                                      int elemBytes, unsigned long long* cvalBits, int* flags, int mode) {
	if (meta[1] != 0 || meta[0] > 32) return;
	const int k = meta[0];
	__shared__ int sOff[MAXOFF];
	if (threadIdx.x < k) sOff[threadIdx.x] = offs[threadIdx.x];
	__syncthreads();
	bool varies = false;
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < samples; i += static_cast<long long>(gridDim.x) * TPB) {
		const int row = static_cast<int>(i * (rows - 1) / max(1, samples - 1));
		for (int e = start[row]; e < start[row + 1]; ++e) {
			const int rel = positions[e] - row;
			int a = 0, b = k;
			while (a < b) {
				const int mid = (a + b) >> 1;
				if (sOff[mid] < rel) a = mid + 1; else b = mid;
			}
			if (a >= k || sOff[a] != rel) continue;  // (patBuildMasks refuses such a matrix)
			const unsigned long long bits = elemBytes == 4 ? static_cast<unsigned long long>(static_cast<const unsigned*>(values)[e])
			                                               : static_cast<const unsigned long long*>(values)[e];
			if (mode == 0) cvalBits[a] = bits;
			else if (cvalBits[a] != bits) varies = true;
		}
	}
	if (varies) atomicOr(flags + 1, 1);
}

extern __shared__ __attribute__((aligned(16))) unsigned char smmPatLds[];

template <typename T>
struct PatStaged;
template <>
struct PatStaged<float> {
	pf32x4 v[PatCfg<float>::NVMAX];
};
template <>
struct PatStaged<double> {
	pf64x2 v[2 * PatCfg<double>::NVMAX];
};

template <typename T>
__device__ __forceinline__ void patStageLoad(PatStaged<T>& r, int t, int nv, int a0, int n1, const T* __restrict__ values) {
#pragma unroll
	for (int v = 0; v < PatCfg<T>::NVMAX; ++v) {
		const int i = a0 + 4 * (t + v * TPB);
		if (v < nv && i < n1) {
			if constexpr (sizeof(T) == 4) {
				r.v[v] = __builtin_nontemporal_load(reinterpret_cast<const pf32x4*>(values + i));
			} else {
				r.v[2 * v] = __builtin_nontemporal_load(reinterpret_cast<const pf64x2*>(values + i));
				r.v[2 * v + 1] = __builtin_nontemporal_load(reinterpret_cast<const pf64x2*>(values + i + 2));
			}
		}
	}
}

template <typename T>
__device__ __forceinline__ void patStageStore(const PatStaged<T>& r, int t, int nv, int a0, int n1, T* sVal) {
#pragma unroll
	for (int v = 0; v < PatCfg<T>::NVMAX; ++v) {
		const int li = 4 * (t + v * TPB);
		if (v < nv && a0 + li < n1) {
			if constexpr (sizeof(T) == 4) {
				*reinterpret_cast<pf32x4*>(sVal + li) = r.v[v];
			} else {
				*reinterpret_cast<pf64x2*>(sVal + li) = r.v[2 * v];
				*reinterpret_cast<pf64x2*>(sVal + li + 2) = r.v[2 * v + 1];
			}
		}
	}
}

template <typename T, int L>
__global__ __launch_bounds__(TPB) void spmvPatternKernel(int nTiles, int cap, int cols, int nOff, const int* __restrict__ offs,
                                                         const int2* __restrict__ rowBlocks, const int* __restrict__ start,
                                                         const unsigned long long* __restrict__ masks, const int* __restrict__ positions,
                                                         const T* __restrict__ values, int opFlags, const T* lhs, const T* __restrict__ divisor, const T* __restrict__ x,
                                                         T* out, int dotMode, const T* __restrict__ w1, T* __restrict__ partials,
                                                         const int* __restrict__ doneFlag) {
	using Cfg = PatCfg<T>;
	constexpr int GATHER = 8;
	const int op = opFlags & 0xFF;
	const bool ntOut = (opFlags & SPMV_NT_OUT) != 0;
	constexpr int LW = L > WAVE ? WAVE : L;
	constexpr int RW = WAVE / LW;
	constexpr int RT = RW * (TPB / WAVE);
	// LDS: sVal[cap + PAD] | sMask[RT] (8-byte aligned) | sStart[RT + 1] | sOff[MAXOFF] | red[4]
	T* sVal = reinterpret_cast<T*>(smmPatLds);
	unsigned long long* sMask = reinterpret_cast<unsigned long long*>(sVal + cap + Cfg::PAD);
	int* sStart = reinterpret_cast<int*>(sMask + RT);
	int* sOff = sStart + RT + 4;
	T* red = reinterpret_cast<T*>(sOff + MAXOFF);
	if (doneFlag && *doneFlag) return;

	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int rowInWave = lane % RW;
	const int piece = lane / RW;
	const int rl = (t >> 6) * RW + rowInWave;
	const int nv = cap / Cfg::PIECE;
	T acc0 = T(0), acc1 = T(0);
	for (int i = t; i < cap + Cfg::PAD; i += TPB) sVal[i] = T(0);
	if (t < MAXOFF) sOff[t] = t < nOff ? offs[t] : 0;

	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int xcdGroup = blockIdx.x % nGroups;
	const int groupSlots = (static_cast<int>(gridDim.x) - xcdGroup + nGroups - 1) / nGroups;
	const int perGroup = (nTiles + nGroups - 1) / nGroups;
	const int tileEnd = min(nTiles, (xcdGroup + 1) * perGroup);
	const int stageLimit = (rowBlocks[nTiles].y & ~3) - cap;
	int tile = xcdGroup * perGroup + blockIdx.x / nGroups;

	PatStaged<T> regs;
	int ps = 0;
	unsigned long long pm = 0ULL;
	int2 m0 = make_int2(0, 0), m1 = make_int2(0, 0), nm0 = make_int2(0, 0), nm1 = make_int2(0, 0);
	if (tile < tileEnd) {
		m0 = rowBlocks[tile];
		m1 = rowBlocks[tile + 1];
		if (tile + groupSlots < tileEnd) {
			nm0 = rowBlocks[tile + groupSlots];
			nm1 = rowBlocks[tile + groupSlots + 1];
		}
		if (m1.y - m0.y <= cap - 3 && (m0.y & ~3) <= stageLimit) {
			patStageLoad<T>(regs, t, nv, m0.y & ~3, m1.y, values);
			if (t < m1.x - m0.x) {
				ps = start[m0.x + t];
				pm = masks[m0.x + t];
			}
		}
	}
	__syncthreads();
	while (tile < tileEnd) {
		const int r0 = m0.x, n0 = m0.y, r1 = m1.x, n1 = m1.y;
		const int nrows = r1 - r0;
		const int a0 = n0 & ~3;
		const bool direct = n1 - n0 > cap - 3 || a0 > stageLimit;
		if (!direct) {
			patStageStore<T>(regs, t, nv, a0, n1, sVal);
			if (t < nrows) {
				sStart[t] = ps - a0;
				sMask[t] = pm;
			}
			if (t == 0) sStart[nrows] = n1 - a0;
		}
		ldsBarrier();
		const int ntile = tile + groupSlots;
		const int2 m0n = nm0, m1n = nm1;
		if (ntile < tileEnd) {
			if (m1n.y - m0n.y <= cap - 3 && (m0n.y & ~3) <= stageLimit) {
				patStageLoad<T>(regs, t, nv, m0n.y & ~3, m1n.y, values);
				if (t < m1n.x - m0n.x) {
					ps = start[m0n.x + t];
					pm = masks[m0n.x + t];
				}
			}
			if (ntile + groupSlots < tileEnd) {
				nm0 = rowBlocks[ntile + groupSlots];
				nm1 = rowBlocks[ntile + groupSlots + 1];
			}
		}
		if (direct) {
			// over-long rows and the last tiles of the matrix: one lane per row, left to right, straight from HBM (with positions[])
			for (int rr = t; rr < nrows; rr += TPB) {
				const int row = r0 + rr;
				const int e = start[row + 1];
				T dot = T(0);
				for (int k = start[row]; k < e; ++k) {
					dot = smmFma(values[k], x[positions[k]], dot);
				}
				const T o = patApplyOp(op, lhs, divisor, row, dot);
				out[row] = o;
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		} else {
			T dot = T(0);
			const int row = r0 + rl;
			if (rl < nrows) {
				const int b = sStart[rl];
				const int e = sStart[rl + 1];
				unsigned long long mm = sMask[rl];
				int kb = b, ke = e;
				if (LW > 1) {
					const int piecelen = (e - b + LW - 1) / LW;
					kb = b + piece * piecelen;
					ke = min(e, kb + piecelen);
					// this piece starts at the (kb - b)-th entry of the row = the (kb - b)-th set bit of the mask
					if (piece > 0 && kb < ke) mm &= ~0ULL << selectBit(mm, kb - b);
				}
				for (int k = kb; k < ke; k += GATHER) {
					const int nvalid = ke - k;
					unsigned off[GATHER];
					T xv[GATHER], vv[GATHER];
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						const int j = mm ? __builtin_ctzll(mm) : 0;
						mm &= mm - 1;
						// entries past the end of the piece get a clamped, valid column; their products are discarded
						const int col = min(max(row + sOff[j], 0), cols - 1);
						off[u] = static_cast<unsigned>(col) * static_cast<unsigned>(sizeof(T));
						vv[u] = sVal[k + u];
					}
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						xv[u] = patGather<T>(x, off[u]);
					}
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						const T next = smmFma(vv[u], xv[u], dot);
						dot = u < nvalid ? next : dot;
					}
				}
			}
			if (LW > 1) {
				T total = dot;
#pragma unroll
				for (int q = 1; q < LW; ++q) {
					total += __shfl(dot, rowInWave + q * RW, WAVE);
				}
				dot = total;
			}
			if (piece == 0 && rl < nrows) {
				const T o = patApplyOp(op, lhs, divisor, row, dot);
				storeOut(out + row, o, ntOut);
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		}
		ldsBarrier();
		tile = ntile;
		m0 = m0n;
		m1 = m1n;
	}
	if (dotMode) {
		if (dotMode == 2) {
			const T s0 = blockSum256(acc0, red);
			if (t == 0) partials[blockIdx.x] = s0;
		}
		const T s1 = blockSum256(acc1, red);
		if (t == 0) partials[(dotMode == 2 ? NPART : 0) + blockIdx.x] = s1;
		for (int i = gridDim.x + blockIdx.x * TPB + t; i < NPART; i += gridDim.x * TPB) {
			partials[i] = T(0);
			if (dotMode == 2) partials[NPART + i] = T(0);
		}
		if (opFlags & SPMV_FINISH) lastBlockSums<T>(partials, NPART, dotMode == 2 ? 2 : 1, partials + PARTS_TOTALS, partsTicket(partials));
	}
}

// ---------------------------------------------------------------------------------------------------------------------------------
// TILE form (rows of ~25-128 entries, 2 or 4 pieces per row: the benchmark matrix) -- the structure smm_spmv.hip's spmvTileKernel
// measured best for the CSR stream, with the column of an entry taken from the row's mask instead of a staged positions[] slice:
//   * the L pieces of a row live in DIFFERENT waves (wave w: piece w % L of the 64 rows of group w / L), so one gather instruction
//     reads ONE window of 64 adjacent columns instead of L windows of 64 / L;
//   * no software pipeline inside the workgroup: a tile's values[] slice is loaded, stored to LDS and summed; with no positions[] in
//     LDS a tile needs half the space, so five to six workgroups share a CU (three for the CSR kernel) and overlap each other;
//   * G gathers per batch, fitted to the piece length (26 entries: 2 x 13).
// Same products in the same order as spmvPatternKernel / the STREAM family at equal L: same bits.
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T, int L, int G>
__global__ __launch_bounds__(TPB) void spmvPatternTileKernel(int nTiles, int cap, int chunkTiles, int cols, int nOff, const int* __restrict__ offs,
                                                             const int2* __restrict__ rowBlocks, const int* __restrict__ start,
                                                             const unsigned long long* __restrict__ masks, const int* __restrict__ positions,
                                                             const T* __restrict__ values, int opFlags, const T* lhs, const T* __restrict__ divisor,
                                                             const T* __restrict__ x, T* out, int dotMode, const T* __restrict__ w1,
                                                             T* __restrict__ partials, const int* __restrict__ doneFlag) {
	using Cfg = PatCfg<T>;
	static_assert(L == 2 || L == 4, "pieces per row");
	static_assert(G <= Cfg::PAD, "a batch may read G - 1 slots past its piece");
	constexpr int NVMAX = Cfg::NVMAX;
	const int op = opFlags & 0xFF;
	const bool ntOut = (opFlags & SPMV_NT_OUT) != 0;
	constexpr int GROUPS = TPB / WAVE / L;
	constexpr int RT = 64 * GROUPS;
	// LDS: sVal[cap + PAD] | sMask[RT] (8-byte aligned) | sStart[RT + 4] | sOff[MAXOFF] | sPart[(L - 1) * RT] | red[4]
	T* sVal = reinterpret_cast<T*>(smmPatLds);
	unsigned long long* sMask = reinterpret_cast<unsigned long long*>(sVal + cap + Cfg::PAD);
	int* sStart = reinterpret_cast<int*>(sMask + RT);
	int* sOff = sStart + RT + 4;
	T* sPart = reinterpret_cast<T*>(sOff + MAXOFF);
	T* red = sPart + (L - 1) * RT;
	if (doneFlag && *doneFlag) return;

	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int wave = t >> 6;
	const int piece = wave % L;
	const int rl = (wave / L) * 64 + lane;
	const int nv = (cap + Cfg::PIECE - 1) / Cfg::PIECE;
	T acc0 = T(0), acc1 = T(0);
	for (int i = t; i < cap + Cfg::PAD; i += TPB) sVal[i] = T(0);
	if (t < MAXOFF) sOff[t] = t < nOff ? offs[t] : 0;
	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int xcdGroup = blockIdx.x % nGroups;
	const int groupSlots = (static_cast<int>(gridDim.x) - xcdGroup + nGroups - 1) / nGroups;
	auto tileOf = [&](int j) {
		const int c = j / chunkTiles;
		const long long tIdx = (static_cast<long long>(c) * nGroups + xcdGroup) * chunkTiles + (j - c * chunkTiles);
		return tIdx < nTiles ? static_cast<int>(tIdx) : nTiles;
	};
	const int stageLimit = (rowBlocks[nTiles].y & ~3) - (cap + 4);
	int j = blockIdx.x / nGroups;
	int tile = tileOf(j);
	int2 m0 = make_int2(0, 0), m1 = make_int2(0, 0);
	if (tile < nTiles) {
		m0 = rowBlocks[tile];
		m1 = rowBlocks[tile + 1];
	}
	__syncthreads();
	while (tile < nTiles) {
		const int r0 = m0.x, n0 = m0.y, r1 = m1.x, n1 = m1.y;
		const int nrows = r1 - r0;
		const int a0 = n0 & ~3;
		const bool direct = n1 - n0 > cap - 3 || a0 > stageLimit;
		if (!direct) {
			typename Pack16<T>::V rv[NVMAX * (sizeof(T) == 4 ? 1 : 2)];
#pragma unroll
			for (int v = 0; v < NVMAX; ++v) {
				const int i = a0 + 4 * (t + v * TPB);
				if (v < nv && i < n1) {
					if constexpr (sizeof(T) == 4) {
						rv[v] = __builtin_nontemporal_load(reinterpret_cast<const pf32x4*>(values + i));
					} else {
						rv[2 * v] = __builtin_nontemporal_load(reinterpret_cast<const pf64x2*>(values + i));
						rv[2 * v + 1] = __builtin_nontemporal_load(reinterpret_cast<const pf64x2*>(values + i + 2));
					}
				}
			}
			int ps = 0;
			unsigned long long pm = 0ULL;
			if (t < nrows) {
				ps = start[r0 + t];
				pm = masks[r0 + t];
			}
#pragma unroll
			for (int v = 0; v < NVMAX; ++v) {
				const int li = 4 * (t + v * TPB);
				if (v < nv && a0 + li < n1) {
					if constexpr (sizeof(T) == 4) {
						*reinterpret_cast<pf32x4*>(sVal + li) = rv[v];
					} else {
						*reinterpret_cast<pf64x2*>(sVal + li) = rv[2 * v];
						*reinterpret_cast<pf64x2*>(sVal + li + 2) = rv[2 * v + 1];
					}
				}
			}
			if (t < nrows) {
				sStart[t] = ps - a0;
				sMask[t] = pm;
			}
			if (t == 0) sStart[nrows] = n1 - a0;
		}
		ldsBarrier();
		j += groupSlots;
		const int ntile = tileOf(j);
		int2 m0n = make_int2(0, 0), m1n = make_int2(0, 0);
		if (ntile < nTiles) {
			m0n = rowBlocks[ntile];
			m1n = rowBlocks[ntile + 1];
		}
		if (direct) {
			// over-long rows and the last tiles of the matrix: one lane per row, left to right, straight from HBM (with positions[])
			for (int rr = t; rr < nrows; rr += TPB) {
				const int row = r0 + rr;
				const T dot = patRowDirect<T, L>(start[row], start[row + 1], values, positions, x);  // (the staged path's pieces: the same bits)
				const T o = patApplyOp(op, lhs, divisor, row, dot);
				out[row] = o;
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		} else {
			T dot = T(0);
			const int row = r0 + rl;
			int kb = 0, ke = 0;
			unsigned long long mm = 0ULL;
			if (rl < nrows) {
				const int b = sStart[rl];
				const int e = sStart[rl + 1];
				const int piecelen = (e - b + L - 1) / L;
				kb = b + piece * piecelen;
				ke = min(e, kb + piecelen);
				mm = sMask[rl];
			}
			// UNIFORM rows (r06): when the 64 rows of this wavefront all hold the SAME offsets -- everywhere in a band except where a diagonal
			// enters or leaves the matrix (or a rank's column range) -- entry e of a row is the same offset for every lane.  The column of an
			// entry then needs no per-lane bit scan, clamp or address arithmetic: the piece's offsets are looked up ONCE per tile (lane u: the u-th),
			// an entry's offset comes out of that vector register with v_readlane, the gather is a load at (x + offset) + 4 row [one 64-bit add].
			// The general path below spends ~15 vector instructions per entry on exactly that.  Same products in the same order: the same bits.
			const unsigned long long lead = patUniform64(mm);
			const bool uniform = !(opFlags & SPMV_NO_FULL_ROWS) && __builtin_amdgcn_ballot_w64(!(rl < nrows && mm == lead)) == 0ULL;
			if (uniform) {
				const int len = __builtin_popcountll(lead);
				const int pl = (len + L - 1) / L;
				const int pu = __builtin_amdgcn_readfirstlane(piece);
				const int e0 = pu * pl, cnt = max(0, min(len, e0 + pl) - e0);
				// lane u of the wavefront looks up the offset of the piece's u-th entry ONCE per tile (a piece has at most 64 entries); the loop below
				// then reads it with v_readlane at a wave-uniform index: no per-entry bit scan at all
				const int offPiece = sOff[lane < cnt ? selectBit(lead, e0 + lane) : 0];
				const T* const xr = x + row;
				for (int e = 0; e < cnt; e += G) {
					T xv[G], vv[G];
					const int nvalid = cnt - e;  // (wave-uniform)
#pragma unroll
					for (int u = 0; u < G; ++u) {
						// entries past the end of the piece repeat its last (valid) column; their products are discarded
						xv[u] = xr[__builtin_amdgcn_readlane(offPiece, min(e + u, cnt - 1))];
						vv[u] = sVal[kb + e + u];
					}
#pragma unroll
					for (int u = 0; u < G; ++u) {
						if (u < nvalid) dot = smmFma(vv[u], xv[u], dot);
					}
				}
			} else {
				// this piece starts at the (kb - b)-th entry of the row = the (kb - b)-th set bit of the mask
				if (rl < nrows && piece > 0 && kb < ke) mm &= ~0ULL << selectBit(mm, kb - sStart[rl]);
				for (int k = kb; k < ke; k += G) {
					unsigned off[G];
					T xv[G], vv[G];
#pragma unroll
					for (int u = 0; u < G; ++u) {
						const int jj = mm ? __builtin_ctzll(mm) : 0;
						mm &= mm - 1;
						// entries past the end of the piece get a clamped, valid column; their products are discarded
						const int col = min(max(row + sOff[jj], 0), cols - 1);
						off[u] = static_cast<unsigned>(col) * static_cast<unsigned>(sizeof(T));
						vv[u] = sVal[k + u];
					}
#pragma unroll
					for (int u = 0; u < G; ++u) xv[u] = patGather<T>(x, off[u]);
					const int nvalid = ke - k;
#pragma unroll
					for (int u = 0; u < G; ++u) {
						const T next = smmFma(vv[u], xv[u], dot);
						dot = u < nvalid ? next : dot;
					}
				}
			}
			// pieces of a row meet in LDS and are added left to right: ((p0 + p1) + p2) + p3
			if (piece > 0 && rl < nrows) sPart[(piece - 1) * RT + rl] = dot;
			ldsBarrier();
			if (piece == 0 && rl < nrows) {
#pragma unroll
				for (int q = 1; q < L; ++q) dot += sPart[(q - 1) * RT + rl];
				const T o = patApplyOp(op, lhs, divisor, row, dot);
				storeOut(out + row, o, ntOut);
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		}
		ldsBarrier();
		tile = ntile;
		m0 = m0n;
		m1 = m1n;
	}
	if (dotMode) {
		if (dotMode == 2) {
			const T s0 = blockSum256(acc0, red);
			if (t == 0) partials[blockIdx.x] = s0;
		}
		const T s1 = blockSum256(acc1, red);
		if (t == 0) partials[(dotMode == 2 ? NPART : 0) + blockIdx.x] = s1;
		for (int i = gridDim.x + blockIdx.x * TPB + t; i < NPART; i += gridDim.x * TPB) {
			partials[i] = T(0);
			if (dotMode == 2) partials[NPART + i] = T(0);
		}
		if (opFlags & SPMV_FINISH) lastBlockSums<T>(partials, NPART, dotMode == 2 ? 2 : 1, partials + PARTS_TOTALS, partsTicket(partials));
	}
}

// ---- analysis ON THE DEVICE ---------------------------------------------------------------------------------------------------
// (1) the offset set from a sample of rows: every workgroup collects the distinct (column - row) of its rows in an LDS table (a value is
// looked up with plain LDS reads first; only a new one is inserted, with an LDS compare-and-swap) and then merges its table into the
// global one the same way.  More than MAXOFF distinct offsets: `state[1]` is raised.  (2) patSortOffsets sorts the <= 64 offsets on the
// device (r04; r03 sorted them on the host between two stream waits), (3) patBuildMasks forms the row masks and VERIFIES every entry of
// positions[] against the set: a matrix whose unsampled rows use other offsets is refused there.  One enqueue, one wait: k, the sorted
// offsets and the verdicts come back together.  Nothing of the matrix travels to the host (r02 copied start[] and 512 rows).
constexpr int PAT_EMPTY = static_cast<int>(0x80000000u);

__device__ __forceinline__ bool patInsert(int* table, int rel) {  // table[MAXOFF], PAT_EMPTY = free; false: the table is full
	for (int slot = 0; slot < MAXOFF; ++slot) {
		int cur = __hip_atomic_load(table + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (cur == rel) return true;
		if (cur == PAT_EMPTY) {
			cur = atomicCAS(table + slot, PAT_EMPTY, rel);
			if (cur == PAT_EMPTY || cur == rel) return true;
		}
	}
	return false;
}

// state: [0 .. MAXOFF) the global table, [MAXOFF] overflow flag, [MAXOFF + 1] longest sampled row
__global__ __launch_bounds__(TPB) void patSampleOffsets(int rows, int samples, const int* __restrict__ start, const int* __restrict__ positions, int* state) {
	__shared__ int sTab[MAXOFF];
	__shared__ int sOver;
	if (threadIdx.x < MAXOFF) sTab[threadIdx.x] = PAT_EMPTY;
	if (threadIdx.x == 0) sOver = 0;
	__syncthreads();
	int longest = 0;
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < samples; i += static_cast<long long>(gridDim.x) * TPB) {
		const int row = static_cast<int>(i * (rows - 1) / max(1, samples - 1));
		const int b = start[row], e = start[row + 1];
		longest = max(longest, e - b);
		for (int k = b; k < e && k - b <= MAXOFF; ++k) {
			if (!patInsert(sTab, positions[k] - row)) sOver = 1;
		}
	}
	if (longest) atomicMax(state + MAXOFF + 1, longest);
	__syncthreads();
	if (threadIdx.x < MAXOFF && sTab[threadIdx.x] != PAT_EMPTY) {
		if (!patInsert(state, sTab[threadIdx.x])) sOver = 1;
	}
	__syncthreads();
	if (threadIdx.x == 0 && sOver) atomicOr(state + MAXOFF, 1);
}

// (1b) the sampled offsets sorted ON THE DEVICE (r04: the host sorted them between two stream waits): one wavefront, a bitonic sort of
// the 64 table slots (free slots sort last), the sorted list padded with 0 to offs[MAXOFF], meta = {k, refusal}: 1 a sampled row holds more
// than 64 entries, 2 the sampled rows use more than 64 offsets, 3 no entries at all.
__global__ __launch_bounds__(WAVE) void patSortOffsets(const int* __restrict__ state, int* __restrict__ offs, int* __restrict__ meta) {
	const int lane = threadIdx.x;
	const int raw = state[lane];
	long long key = raw == PAT_EMPTY ? (1LL << 40) : static_cast<long long>(raw);
#pragma unroll
	for (int size = 2; size <= WAVE; size <<= 1) {
#pragma unroll
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			const long long other = __shfl_xor(key, stride, WAVE);
			const bool up = (lane & size) == 0;        // ascending block
			const bool lower = (lane & stride) == 0;   // this lane keeps the smaller of the pair in an ascending block
			const bool takeMin = up == lower;
			key = takeMin ? (other < key ? other : key) : (other > key ? other : key);
		}
	}
	const bool valid = key < (1LL << 40);
	const int k = __popcll(__ballot(valid));
	offs[lane] = valid ? static_cast<int>(key) : 0;
	if (lane == 0) {
		meta[0] = k;
		meta[1] = state[MAXOFF + 1] > MAXOFF ? 1 : state[MAXOFF] ? 2 : k == 0 ? 3 : 0;
	}
}

// ---------------------------------------------------------------------------------------------------------------------------------
// CONSTANT DIAGONALS (r03): when, on top of the row masks, every entry of a diagonal holds the SAME value -- the Laplacians of BASELINE
// configs 1, 2 and 4, every constant-coefficient stencil -- values[] is redundant as well: an SpMV needs the row's mask, x and one value
// per offset (<= 64 numbers, LDS).  For the 512^3 fp64 Laplacian that is 24 bytes per row instead of 84 (PATTERN) or 104 (CSR).  The
// property is verified bit for bit against EVERY entry on the device (patConstSample, patBuildMasks) before the encoding is used, so the
// products are the reference's products: c_j == values[k] as bit patterns, multiplied with the same x[] in the same left-to-right order
// (the mask's bit order is the ascending column order of the row, ref:1484-1499).  One lane per row only (the stencils' choice anyway);
// a request for more lanes per row is served by the mask kernels above, so "same bits as STREAM at equal lanes" holds.
// Rows are dealt in tiles of TPB rows; chunkTiles as in the tile kernels (0: one contiguous eighth per XCD).
template <typename T>
__global__ __launch_bounds__(TPB) void spmvPatternConstKernel(int rows, int cols, int nOff, const int* __restrict__ offs,
                                                              const unsigned long long* __restrict__ cvalBits,
                                                              const unsigned long long* __restrict__ masks, int chunkTiles, int opFlags,
                                                              const T* lhs, const T* __restrict__ divisor, const T* __restrict__ x, T* out, int dotMode,
                                                              const T* __restrict__ w1, T* __restrict__ partials, const int* __restrict__ doneFlag) {
	constexpr int GATHER = 8;
	__shared__ int sOff[MAXOFF];
	__shared__ T sC[MAXOFF];
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	const int op = opFlags & 0xFF;
	const bool ntOut = (opFlags & SPMV_NT_OUT) != 0;
	const int t = threadIdx.x;
	if (t < MAXOFF) {
		sOff[t] = t < nOff ? offs[t] : 0;
		T c = T(0);
		if (t < nOff) {
			const unsigned long long bits = cvalBits[t];
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
	const int nTiles = (rows + TPB - 1) / TPB;
	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int xcdGroup = blockIdx.x % nGroups;
	const int groupSlots = (static_cast<int>(gridDim.x) - xcdGroup + nGroups - 1) / nGroups;
	const int perGroup = (nTiles + nGroups - 1) / nGroups;
	auto tileOf = [&](int j) {  // the j-th tile of this XCD group
		if (chunkTiles <= 0) {
			const int tl = xcdGroup * perGroup + j;
			return j < perGroup && tl < nTiles ? tl : nTiles;
		}
		const int c = j / chunkTiles;
		const long long tIdx = (static_cast<long long>(c) * nGroups + xcdGroup) * chunkTiles + (j - c * chunkTiles);
		return tIdx < nTiles ? static_cast<int>(tIdx) : nTiles;
	};
#if defined(SMM_EXP_CONST) && (SMM_EXP_CONST & 4)  // ablation builds only: no mask stream (every row takes the first row's mask)
#define SMM_CONST_MASK_AT(i) masks[((i) & 0) + static_cast<size_t>(rows) / 2 + rows / 1024 + 37]  // (an interior row of a cubic grid)
#else
#define SMM_CONST_MASK_AT(i) masks[i]
#endif
	T acc0 = T(0), acc1 = T(0);
	int j = blockIdx.x / nGroups;
	int tile = tileOf(j);
	unsigned long long nextMask = 0ULL;
	if (tile < nTiles && tile * TPB + t < rows) nextMask = SMM_CONST_MASK_AT(static_cast<size_t>(tile) * TPB + t);
	while (tile < nTiles) {
		const int row = tile * TPB + t;
		unsigned long long mm = nextMask;
		j += groupSlots;
		const int ntile = tileOf(j);
		nextMask = 0ULL;
		if (ntile < nTiles && ntile * TPB + t < rows) nextMask = SMM_CONST_MASK_AT(static_cast<size_t>(ntile) * TPB + t);
		if (row < rows) {
			T dot = T(0);
			do {  // (an empty row runs one batch of discarded products)
				unsigned off[GATHER];
				T cv[GATHER], xv[GATHER];
				bool live[GATHER];
#pragma unroll
				for (int u = 0; u < GATHER; ++u) {
					live[u] = mm != 0ULL;
					const int jj = mm ? __builtin_ctzll(mm) : 0;
					mm &= mm - 1;
	#if defined(SMM_EXP_CONST) && (SMM_EXP_CONST & 2)  // ablation builds only (tools/run_const_ablate.sh): every gather reads x[row]
				const int col = row;
#elif defined(SMM_EXP_CONST) && (SMM_EXP_CONST & 8)  // ... only the far (plane) gathers are redirected to x[row]
				const int col = (sOff[jj] > 4096 || sOff[jj] < -4096) ? row : min(max(row + sOff[jj], 0), cols - 1);
#else
				const int col = min(max(row + sOff[jj], 0), cols - 1);  // (dead slots: a clamped, valid column)
#endif
					off[u] = static_cast<unsigned>(col) * static_cast<unsigned>(sizeof(T));
					cv[u] = sC[jj];
				}
#pragma unroll
				for (int u = 0; u < GATHER; ++u) xv[u] = patGather<T>(x, off[u]);
#pragma unroll
				for (int u = 0; u < GATHER; ++u) {
					const T next = smmFma(cv[u], xv[u], dot);
					dot = live[u] ? next : dot;
				}
			} while (mm != 0ULL);
			const T o = patApplyOp(op, lhs, divisor, row, dot);
#if defined(SMM_EXP_CONST) && (SMM_EXP_CONST & 1)  // ablation builds only: the kernel without its out[] stream
			if (o == T(-1.2345e30)) out[row] = o;
#else
			storeOut(out + row, o, ntOut);
#endif
			if (dotMode == 2) acc0 += o * o;
			if (dotMode) acc1 += o * w1[row];
		}
		tile = ntile;
	}
	if (dotMode) {
		if (dotMode == 2) {
			const T s0 = blockSum256(acc0, red);
			if (t == 0) partials[blockIdx.x] = s0;
		}
		const T s1 = blockSum256(acc1, red);
		if (t == 0) partials[(dotMode == 2 ? NPART : 0) + blockIdx.x] = s1;
		for (int i = gridDim.x + blockIdx.x * TPB + t; i < NPART; i += gridDim.x * TPB) {
			partials[i] = T(0);
			if (dotMode == 2) partials[NPART + i] = T(0);
		}
		if (opFlags & SPMV_FINISH) lastBlockSums<T>(partials, NPART, dotMode == 2 ? 2 : 1, partials + PARTS_TOTALS, partsTicket(partials));
	}
}

// ---------------------------------------------------------------------------------------------------------------------------------
// MASKS, short rows (<= KMAX entries: the 5- and 7-point stencils with varying coefficients), one lane per row, WITHOUT workgroup tiles
// (r03): every WAVEFRONT owns 64 consecutive rows; their values[] are one contiguous run, fetched with coalesced loads one group
// ahead into registers, passed through a wave-private LDS slice (a wavefront's LDS operations are in order: no barrier anywhere) and
// read back by the row's lane.  Same products in the same order as spmvPatternKernel<T, 1> / the reference; the tile kernel's two
// workgroup barriers per 256 rows and its tile table are what this form drops.
template <typename T, int KMAX>
__global__ __launch_bounds__(TPB) void spmvPatternWaveKernel(int rows, int cols, int nOff, const int* __restrict__ offs, const int* __restrict__ start,
                                                             const T* __restrict__ values, const unsigned long long* __restrict__ masks, int chunkTiles,
                                                             int opFlags, const T* lhs, const T* __restrict__ divisor, const T* __restrict__ x, T* out,
                                                             int dotMode, const T* __restrict__ w1, T* __restrict__ partials,
                                                             const int* __restrict__ doneFlag) {
	constexpr int GATHER = 8;
	__shared__ int sOff[MAXOFF];
	__shared__ T sVal[TPB / WAVE][WAVE * KMAX + GATHER];
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	const int op = opFlags & 0xFF;
	const bool ntOut = (opFlags & SPMV_NT_OUT) != 0;
	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int wv = t >> 6;
	if (t < MAXOFF) sOff[t] = t < nOff ? offs[t] : 0;
	for (int i = lane; i < WAVE * KMAX + GATHER; i += WAVE) sVal[wv][i] = T(0);
	__syncthreads();
	const int nTiles = (rows + TPB - 1) / TPB;
	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int xcdGroup = blockIdx.x % nGroups;
	const int groupSlots = (static_cast<int>(gridDim.x) - xcdGroup + nGroups - 1) / nGroups;
	const int perGroup = (nTiles + nGroups - 1) / nGroups;
	auto tileOf = [&](int j) {  // the j-th tile of this XCD group
		if (chunkTiles <= 0) {
			const int tl = xcdGroup * perGroup + j;
			return j < perGroup && tl < nTiles ? tl : nTiles;
		}
		const int c = j / chunkTiles;
		const long long tIdx = (static_cast<long long>(c) * nGroups + xcdGroup) * chunkTiles + (j - c * chunkTiles);
		return tIdx < nTiles ? static_cast<int>(tIdx) : nTiles;
	};
	// what a wavefront fetches for its 64 rows of a tile: mask and row start per lane, and the rows' values[] run in KMAX coalesced loads
	struct Fetched {
		unsigned long long mask;
		int b, eBeg, eEnd;
		T v[KMAX];
	};
	auto fetch = [&](int tile, Fetched& f) {
		f.mask = 0ULL;
		f.b = 0;
		f.eBeg = 0;
		f.eEnd = 0;
		if (tile >= nTiles) return;
		const int r0 = tile * TPB + wv * WAVE;
		if (r0 >= rows) return;
		const int nr = min(WAVE, rows - r0);
		const int row = r0 + min(lane, nr - 1);
		f.b = start[row];
		if (lane < nr) f.mask = masks[row];
		f.eBeg = __builtin_amdgcn_readfirstlane(f.b);
		f.eEnd = start[r0 + nr];
#pragma unroll
		for (int k = 0; k < KMAX; ++k) {
			const int idx = f.eBeg + k * WAVE + lane;
			f.v[k] = idx < f.eEnd ? __builtin_nontemporal_load(values + idx) : T(0);
		}
	};
	T acc0 = T(0), acc1 = T(0);
	int j = blockIdx.x / nGroups;
	int tile = tileOf(j);
	Fetched cur, nxt;
	fetch(tile, cur);
	while (tile < nTiles) {
		const int row = tile * TPB + t;
		j += groupSlots;
		const int ntile = tileOf(j);
		fetch(ntile, nxt);  // in flight while this group is summed
#pragma unroll
		for (int k = 0; k < KMAX; ++k) sVal[wv][k * WAVE + lane] = cur.v[k];
		if (row < rows) {
			unsigned long long mm = cur.mask;
			int at = cur.b - cur.eBeg;  // the row's next value in the wavefront's slice
			T dot = T(0);
			do {  // (an empty row runs one batch of discarded products)
				unsigned off[GATHER];
				T cv[GATHER], xv[GATHER];
				bool live[GATHER];
#pragma unroll
				for (int u = 0; u < GATHER; ++u) {
					live[u] = mm != 0ULL;
					const int jj = mm ? __builtin_ctzll(mm) : 0;
					mm &= mm - 1;
					const int col = min(max(row + sOff[jj], 0), cols - 1);  // (dead slots: a clamped, valid column)
					off[u] = static_cast<unsigned>(col) * static_cast<unsigned>(sizeof(T));
					cv[u] = sVal[wv][at + u];  // (dead slots read past the row: the slice is padded by GATHER)
				}
				at += GATHER;
#pragma unroll
				for (int u = 0; u < GATHER; ++u) xv[u] = patGather<T>(x, off[u]);
#pragma unroll
				for (int u = 0; u < GATHER; ++u) {
					const T next = smmFma(cv[u], xv[u], dot);
					dot = live[u] ? next : dot;
				}
			} while (mm != 0ULL);
			const T o = patApplyOp(op, lhs, divisor, row, dot);
			storeOut(out + row, o, ntOut);
			if (dotMode == 2) acc0 += o * o;
			if (dotMode) acc1 += o * w1[row];
		}
		tile = ntile;
		cur = nxt;
	}
	if (dotMode) {
		if (dotMode == 2) {
			const T s0 = blockSum256(acc0, red);
			if (t == 0) partials[blockIdx.x] = s0;
		}
		const T s1 = blockSum256(acc1, red);
		if (t == 0) partials[(dotMode == 2 ? NPART : 0) + blockIdx.x] = s1;
		for (int i = gridDim.x + blockIdx.x * TPB + t; i < NPART; i += gridDim.x * TPB) {
			partials[i] = T(0);
			if (dotMode == 2) partials[NPART + i] = T(0);
		}
		if (opFlags & SPMV_FINISH) lastBlockSums<T>(partials, NPART, dotMode == 2 ? 2 : 1, partials + PARTS_TOTALS, partsTicket(partials));
	}
}

// ---------------------------------------------------------------------------------------------------------------------------------
// DICTIONARY encoding of the same family (r03): matrices whose entries use MORE than 64 distinct offsets
// column - row, or hold rows of more than 64 entries, but no more than 65 536 distinct offsets in all -- banded matrices with hundreds
// of diagonals, meshes numbered along a band.  positions[] (4 bytes per entry) is replaced by a 16-bit CODE per entry, the index of
// the entry's offset in the matrix's sorted dictionary: 6 instead of 8 bytes per fp32 entry.  (For <= 64 offsets the masks above cost
// 8 bytes per ROW and win.)  Built on the device from ALL entries -- there is nothing to sample and nothing to verify afterwards: an
// entry's code is found by searching the dictionary for its own offset.  The kernel is spmvPatternKernel with a staged slice of codes
// in place of the row masks: same lanes, same products, same order, same bits as the STREAM family at equal lanes.
constexpr int DICT_MAX = 65536;
constexpr int DICT_HCAP = 1 << 18;    // open-addressing table of the offsets seen (<= 25 % full)
constexpr int DICT_LDS_MAX = 2048;    // dictionaries up to this size are copied to LDS by every workgroup (the kernel stays below 64 KB of LDS), larger ones stay in L2
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct DictCfg {
	static constexpr int NCMAX = (PatCfg<T>::NVMAX * PatCfg<T>::PIECE + 8 + 8 * TPB - 1) / (8 * TPB);  // 16-byte loads of 8 codes per lane
};

// Walks the entries of 64 consecutive rows with one wavefront, coalesced (the scheme of patBuildMasks): f(entry index, row, column).
template <typename F>
__device__ __forceinline__ void forEntriesOf64Rows(int rows, long long group, const int* __restrict__ start, const int* __restrict__ positions,
                                                   int* sRow, F f) {
	const int lane = threadIdx.x & (WAVE - 1);
	const int r0 = static_cast<int>(group * WAVE);
	const int nr = min(WAVE, rows - r0);
	sRow[lane] = start[r0 + min(lane, nr)];
	if (lane == 0) sRow[WAVE] = start[r0 + nr];
	const int eBegin = __builtin_amdgcn_readfirstlane(sRow[0]);
	const int eEnd = start[r0 + nr];
	for (int e0 = eBegin; e0 < eEnd; e0 += WAVE) {
		const int e = e0 + lane;
		if (e < eEnd) {
			int lo = 0, hi = nr;  // the last row i in [0, nr) with sRow[i] <= e
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (sRow[mid] <= e) lo = mid; else hi = mid;
			}
			f(e, r0 + lo, positions[e]);
		}
	}
}

__global__ void fillIntKernel(int* p, int n, int v) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}

__device__ __forceinline__ unsigned dictHash(int off) { return (static_cast<unsigned>(off) * 0x9E3779B1u) >> (32 - 18); }

// state: [0] distinct offsets so far, [1] raised when there are more than DICT_MAX (every wavefront then stops at its next group)
__global__ __launch_bounds__(TPB) void dictCollectKernel(int rows, const int* __restrict__ start, const int* __restrict__ positions, int* table, int* state) {
	__shared__ int sRow[TPB / WAVE][WAVE + 1];
	__shared__ int sSeen[1024];  // offsets this workgroup has already handed to the global table (direct-mapped; a lost race only repeats a lookup)
	for (int i = threadIdx.x; i < 1024; i += TPB) sSeen[i] = PAT_EMPTY;
	__syncthreads();
	const int w = threadIdx.x >> 6;
	const long long groups = (static_cast<long long>(rows) + WAVE - 1) / WAVE;
	for (long long g = static_cast<long long>(blockIdx.x) * (TPB / WAVE) + w; g < groups; g += static_cast<long long>(gridDim.x) * (TPB / WAVE)) {
		if (__hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
		forEntriesOf64Rows(rows, g, start, positions, sRow[w], [&](int, int row, int col) {
			const int off = col - row;
			const unsigned h = dictHash(off);
			if (sSeen[h & 1023u] == off) return;
			sSeen[h & 1023u] = off;
			for (unsigned probe = 0; probe < static_cast<unsigned>(DICT_HCAP); ++probe) {
				int* slot = table + ((h + probe) & (DICT_HCAP - 1));
				int cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (cur == off) return;
				if (cur == PAT_EMPTY) {
					cur = atomicCAS(slot, PAT_EMPTY, off);
					if (cur == PAT_EMPTY) {
						if (atomicAdd(state, 1) + 1 > DICT_MAX) atomicOr(state + 1, 1);
						return;
					}
					if (cur == off) return;
				}
				if (__hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // (the table fills up only past the limit)
			}
		});
	}
}

// codes[e] = index of (column - row) of entry e in the sorted dictionary; an offset that is not there (impossible: the dictionary was
// collected from these very entries) raises the flag
__global__ __launch_bounds__(TPB) void dictEncodeKernel(int rows, int k, const int* __restrict__ dict, const int* __restrict__ start,
                                                        const int* __restrict__ positions, unsigned short* __restrict__ codes, int* mismatch) {
	__shared__ int sRow[TPB / WAVE][WAVE + 1];
	const int w = threadIdx.x >> 6;
	const long long groups = (static_cast<long long>(rows) + WAVE - 1) / WAVE;
	bool bad = false;
	for (long long g = static_cast<long long>(blockIdx.x) * (TPB / WAVE) + w; g < groups; g += static_cast<long long>(gridDim.x) * (TPB / WAVE)) {
		forEntriesOf64Rows(rows, g, start, positions, sRow[w], [&](int e, int row, int col) {
			const int off = col - row;
			int a = 0, b = k;
			while (a < b) {
				const int mid = (a + b) >> 1;
				if (dict[mid] < off) a = mid + 1; else b = mid;
			}
			if (a >= k || dict[a] != off) {
				bad = true;
				a = 0;
			}
			codes[e] = static_cast<unsigned short>(a);
		});
	}
	if (bad) atomicOr(mismatch, 1);
}

template <typename T, int L, bool DLDS>
__global__ __launch_bounds__(TPB) void spmvDictKernel(int nTiles, int cap, int cols, int nOff, const int* __restrict__ dict,
                                                      const int2* __restrict__ rowBlocks, const int* __restrict__ start,
                                                      const unsigned short* __restrict__ codes, const int* __restrict__ positions,
                                                      const T* __restrict__ values, int opFlags, const T* lhs, const T* __restrict__ divisor,
                                                      const T* __restrict__ x, T* out, int dotMode, const T* __restrict__ w1,
                                                      T* __restrict__ partials, const int* __restrict__ doneFlag) {
	using Cfg = PatCfg<T>;
	constexpr int GATHER = 8;
	constexpr int NCMAX = DictCfg<T>::NCMAX;
	const int op = opFlags & 0xFF;
	const bool ntOut = (opFlags & SPMV_NT_OUT) != 0;
	constexpr int LW = L > WAVE ? WAVE : L;
	constexpr int RW = WAVE / LW;
	constexpr int RT = RW * (TPB / WAVE);
	// LDS: sVal[cap + PAD] | sStart[RT + 4] | sCode[cap + PAD + 8, 16-bit] | sDict[nOff, when it fits] | red[4]
	T* sVal = reinterpret_cast<T*>(smmPatLds);
	int* sStart = reinterpret_cast<int*>(sVal + cap + Cfg::PAD);
	unsigned short* sCode = reinterpret_cast<unsigned short*>(sStart + RT + 4);
	const int codeSlots = (cap + Cfg::PAD + 8 + 7) & ~7;
	int* sDict = reinterpret_cast<int*>(sCode + codeSlots);
	T* red = reinterpret_cast<T*>(sDict + (DLDS ? ((nOff + 1) & ~1) : 0));
	if (doneFlag && *doneFlag) return;

	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int rowInWave = lane % RW;
	const int piece = lane / RW;
	const int rl = (t >> 6) * RW + rowInWave;
	const int nv = cap / Cfg::PIECE;
	T acc0 = T(0), acc1 = T(0);
	for (int i = t; i < cap + Cfg::PAD; i += TPB) sVal[i] = T(0);
	for (int i = t; i < codeSlots; i += TPB) sCode[i] = 0;  // whatever a lane reads past its piece is a valid code (the products are discarded)
	if (DLDS) {
		for (int i = t; i < nOff; i += TPB) sDict[i] = dict[i];
	}

	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int xcdGroup = blockIdx.x % nGroups;
	const int groupSlots = (static_cast<int>(gridDim.x) - xcdGroup + nGroups - 1) / nGroups;
	const int perGroup = (nTiles + nGroups - 1) / nGroups;
	const int tileEnd = min(nTiles, (xcdGroup + 1) * perGroup);
	const int stageLimit = (rowBlocks[nTiles].y & ~3) - cap;
	int tile = xcdGroup * perGroup + blockIdx.x / nGroups;

	PatStaged<T> regs;
	pu32x4 rc[NCMAX];
	int ps = 0;
	// codes of the entries [c0, n1), c0 = the tile's first entry rounded down to 8: 16-byte loads; the array is padded, so a load that starts
	// inside it may run past the last entry
	auto loadCodes = [&](int c0, int n1) {
#pragma unroll
		for (int v = 0; v < NCMAX; ++v) {
			const int i = c0 + 8 * (t + v * TPB);
			if (i < n1) rc[v] = __builtin_nontemporal_load(reinterpret_cast<const pu32x4*>(codes + i));
		}
	};
	auto storeCodes = [&](int c0, int n1) {
#pragma unroll
		for (int v = 0; v < NCMAX; ++v) {
			const int li = 8 * (t + v * TPB);
			if (c0 + li < n1) *reinterpret_cast<pu32x4*>(sCode + li) = rc[v];
		}
	};
	int2 m0 = make_int2(0, 0), m1 = make_int2(0, 0), nm0 = make_int2(0, 0), nm1 = make_int2(0, 0);
	if (tile < tileEnd) {
		m0 = rowBlocks[tile];
		m1 = rowBlocks[tile + 1];
		if (tile + groupSlots < tileEnd) {
			nm0 = rowBlocks[tile + groupSlots];
			nm1 = rowBlocks[tile + groupSlots + 1];
		}
		if (m1.y - m0.y <= cap - 3 && (m0.y & ~3) <= stageLimit) {
			patStageLoad<T>(regs, t, nv, m0.y & ~3, m1.y, values);
			loadCodes(m0.y & ~7, m1.y);
			if (t < m1.x - m0.x) ps = start[m0.x + t];
		}
	}
	__syncthreads();
	while (tile < tileEnd) {
		const int r0 = m0.x, n0 = m0.y, r1 = m1.x, n1 = m1.y;
		const int nrows = r1 - r0;
		const int a0 = n0 & ~3;
		const int delta = a0 - (n0 & ~7);  // sCode[i] is entry (n0 & ~7) + i, sVal[i] is entry a0 + i
		const bool direct = n1 - n0 > cap - 3 || a0 > stageLimit;
		if (!direct) {
			patStageStore<T>(regs, t, nv, a0, n1, sVal);
			storeCodes(n0 & ~7, n1);
			if (t < nrows) sStart[t] = ps - a0;
			if (t == 0) sStart[nrows] = n1 - a0;
		}
		ldsBarrier();
		const int ntile = tile + groupSlots;
		const int2 m0n = nm0, m1n = nm1;
		if (ntile < tileEnd) {
			if (m1n.y - m0n.y <= cap - 3 && (m0n.y & ~3) <= stageLimit) {
				patStageLoad<T>(regs, t, nv, m0n.y & ~3, m1n.y, values);
				loadCodes(m0n.y & ~7, m1n.y);
				if (t < m1n.x - m0n.x) ps = start[m0n.x + t];
			}
			if (ntile + groupSlots < tileEnd) {
				nm0 = rowBlocks[ntile + groupSlots];
				nm1 = rowBlocks[ntile + groupSlots + 1];
			}
		}
		if (direct) {
			// over-long rows and the last tiles of the matrix: one lane per row, left to right, straight from HBM (with positions[])
			for (int rr = t; rr < nrows; rr += TPB) {
				const int row = r0 + rr;
				const int e = start[row + 1];
				T dot = T(0);
				for (int k = start[row]; k < e; ++k) dot = smmFma(values[k], x[positions[k]], dot);
				const T o = patApplyOp(op, lhs, divisor, row, dot);
				out[row] = o;
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		} else {
			T dot = T(0);
			const int row = r0 + rl;
			if (rl < nrows) {
				const int b = sStart[rl];
				const int e = sStart[rl + 1];
				int kb = b, ke = e;
				if (LW > 1) {
					const int piecelen = (e - b + LW - 1) / LW;
					kb = b + piece * piecelen;
					ke = min(e, kb + piecelen);
				}
				for (int k = kb; k < ke; k += GATHER) {
					const int nvalid = ke - k;
					unsigned off[GATHER];
					T xv[GATHER], vv[GATHER];
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						const int code = sCode[k + u + delta];
						const int rel = DLDS ? sDict[code] : dict[code];
						// entries past the end of the piece get a clamped, valid column; their products are discarded
						const int col = min(max(row + rel, 0), cols - 1);
						off[u] = static_cast<unsigned>(col) * static_cast<unsigned>(sizeof(T));
						vv[u] = sVal[k + u];
					}
#pragma unroll
					for (int u = 0; u < GATHER; ++u) xv[u] = patGather<T>(x, off[u]);
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						const T next = smmFma(vv[u], xv[u], dot);
						dot = u < nvalid ? next : dot;
					}
				}
			}
			if (LW > 1) {
				T total = dot;
#pragma unroll
				for (int q = 1; q < LW; ++q) total += __shfl(dot, rowInWave + q * RW, WAVE);
				dot = total;
			}
			if (piece == 0 && rl < nrows) {
				const T o = patApplyOp(op, lhs, divisor, row, dot);
				storeOut(out + row, o, ntOut);
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		}
		ldsBarrier();
		tile = ntile;
		m0 = m0n;
		m1 = m1n;
	}
	if (dotMode) {
		if (dotMode == 2) {
			const T s0 = blockSum256(acc0, red);
			if (t == 0) partials[blockIdx.x] = s0;
		}
		const T s1 = blockSum256(acc1, red);
		if (t == 0) partials[(dotMode == 2 ? NPART : 0) + blockIdx.x] = s1;
		for (int i = gridDim.x + blockIdx.x * TPB + t; i < NPART; i += gridDim.x * TPB) {
			partials[i] = T(0);
			if (dotMode == 2) partials[NPART + i] = T(0);
		}
		if (opFlags & SPMV_FINISH) lastBlockSums<T>(partials, NPART, dotMode == 2 ? 2 : 1, partials + PARTS_TOTALS, partsTicket(partials));
	}
}

// every translation unit of the library is a code object of its own, built for the device at the FIRST launch of any of its kernels
// (5-9 ms each, measured: profiles/r04/first_spmv_setup_trace.txt); smm_hip_init touches one kernel of each hot-path unit so that
// the first SpMV of a process does not pay for it (SMM_HIP_PRELOAD=0: load lazily as before)
void preloadPatternUnit() {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(patSampleOffsets));
	(void)hipGetLastError();
}

// SMM_HIP_AUTO_DICT=0: the automatic attempt (first SpMV of a large matrix) stops at the masks; the dictionary encoding then needs an
// explicit smm_hip_csr_set_kernel(m, SMM_SPMV_PATTERN, lanes)
static bool autoDictAllowed() {
	static const bool on = [] {
		const char* env = getenv("SMM_HIP_AUTO_DICT");
		return env ? atoi(env) != 0 : true;
	}();
	return on;
}

// the mask encoding: SMM_HIP_OK, SMM_HIP_ERR_INVALID (no such pattern: *why says which test failed) or a HIP failure
static int tryMasks(smm_hip_csr* m, hipStream_t s, const char** why) {
	SetupTrace traceAll("pattern: masks (sample + sort + build + verify)");
	// ONE enqueue, ONE wait (r04; r03: sample -> wait -> host sort -> build -> wait): the offsets of the sampled rows are sorted on the
	// device, the kernels behind read k from device memory and return at once for a refused matrix, and everything the host has to
	// know -- the refusal, k, the sorted offsets, the two verdicts -- comes back in one copy.
	DevBuf<int> d_state, d_off, d_flag, d_meta;  // released on every early return
	DevBuf<unsigned long long> d_masks, d_cval;
	SMM_TRY(d_state.alloc(MAXOFF + 2));
	SMM_TRY(d_off.alloc(MAXOFF));
	SMM_TRY(d_flag.alloc(2));  // [0] an entry off the offset set / out of order, [1] some diagonal holds more than one value
	SMM_TRY(d_meta.alloc(2));
	SMM_TRY(d_cval.alloc(MAXOFF));
	{
		SetupTrace trace("pattern:   allocate the masks");
		SMM_TRY(d_masks.alloc(static_cast<size_t>(m->rows)));
	}
	std::vector<int> init(MAXOFF + 2, PAT_EMPTY);
	init[MAXOFF] = 0;
	init[MAXOFF + 1] = 0;
	SMM_HIP_TRY(hipMemcpyAsync(d_state, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice, s));
	SMM_HIP_TRY(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), s));
	SMM_HIP_TRY(hipMemsetAsync(d_cval, 0, MAXOFF * sizeof(unsigned long long), s));
	const int samples = std::min(m->rows, 16384);
	patSampleOffsets<<<(samples + TPB - 1) / TPB, TPB, 0, s>>>(m->rows, samples, m->d_start, m->d_positions, d_state);
	patSortOffsets<<<1, WAVE, 0, s>>>(d_state, d_off, d_meta);
	const int elemBytes = m->dtype == SMM_DTYPE_F32 ? 4 : 8;
	// constant diagonals are looked for in matrices of stencil shape only (<= 32 offsets: the kernels check); SMM_HIP_PATTERN_CONST=0 turns the encoding off
	static const bool constAllowed = [] {
		const char* env = getenv("SMM_HIP_PATTERN_CONST");
		return env ? atoi(env) != 0 : true;
	}();
	if (constAllowed) {
		const int sgrid = (samples + TPB - 1) / TPB;
		patConstSample<<<sgrid, TPB, 0, s>>>(m->rows, samples, d_meta, d_off, m->d_start, m->d_positions, m->d_values, elemBytes, d_cval, d_flag, 0);
		patConstSample<<<sgrid, TPB, 0, s>>>(m->rows, samples, d_meta, d_off, m->d_start, m->d_positions, m->d_values, elemBytes, d_cval, d_flag, 1);
	}
	const int grid = static_cast<int>(std::min<long long>((m->rows + TPB - 1LL) / TPB, numCUs() * 8LL));
	patBuildMasks<<<grid, TPB, 0, s>>>(m->rows, d_meta, d_off, m->d_start, m->d_positions, d_masks, d_flag, constAllowed ? m->d_values : nullptr, elemBytes, d_cval);
	struct Back {
		int meta[2], flags[2], offs[MAXOFF];
		unsigned long long cval[MAXOFF];
	} back{};
	SMM_HIP_TRY(hipMemcpyAsync(back.meta, d_meta, sizeof(back.meta), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipMemcpyAsync(back.flags, d_flag, sizeof(back.flags), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipMemcpyAsync(back.offs, d_off, sizeof(back.offs), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipMemcpyAsync(back.cval, d_cval, sizeof(back.cval), hipMemcpyDeviceToHost, s));
	{
		SetupTrace trace("pattern:   wait for the whole analysis");
		SMM_HIP_TRY(hipStreamSynchronize(s));
	}
	auto no = [why](const char* text) {
		*why = text;
		return static_cast<int>(SMM_HIP_ERR_INVALID);
	};
	if (back.meta[1] == 1) return no("a row holds more than 64 entries");
	if (back.meta[1] == 2) return no("the rows do not share a set of <= 64 column offsets");
	if (back.meta[1] == 3) return no("no entries in the sampled rows");
	if (back.flags[0]) return no("some entry's column offset is outside the offset set of the sampled rows");
	const int k = back.meta[0];
	std::vector<int> offs(back.offs, back.offs + k);
	m->pat_k = k;
	m->pat_offs_host = offs;
	m->pat_encoding = 0;
	m->pat_max_off = std::max(std::abs(offs.front()), std::abs(offs.back()));
	m->d_pat_off = d_off.detach();
	m->d_pat_masks = d_masks.detach();
	m->pat_const = constAllowed && k <= 32 && back.flags[1] == 0;
	if (m->pat_const) {
		m->d_pat_cval = d_cval.detach();
		m->pat_cval_host.assign(back.cval, back.cval + k);
	}
	planMarch(m);
	SMM_TRY(marchBuildMasks32(m, s));
	return SMM_HIP_OK;
}

// the dictionary encoding: every entry's offset into a device hash set, the set sorted (rocPRIM radix sort), every entry encoded
static int tryDict(smm_hip_csr* m, hipStream_t s, const char** why) {
	auto no = [why](const char* text) {
		*why = text;
		return static_cast<int>(SMM_HIP_ERR_INVALID);
	};
	DevBuf<int> d_table, d_sorted, d_state, d_dict;
	DevBuf<unsigned short> d_codes;
	DevBuf<char> temp;
	SMM_TRY(d_table.alloc(DICT_HCAP));
	SMM_TRY(d_sorted.alloc(DICT_HCAP));
	SMM_TRY(d_state.alloc(4));
	SMM_HIP_TRY(hipMemsetAsync(d_state, 0, 4 * sizeof(int), s));
	fillIntKernel<<<DICT_HCAP / TPB, TPB, 0, s>>>(d_table, DICT_HCAP, PAT_EMPTY);
	const long long groups = (static_cast<long long>(m->rows) + WAVE - 1) / WAVE;
	const int grid = static_cast<int>(std::max<long long>(1, std::min<long long>((groups + TPB / WAVE - 1) / (TPB / WAVE), numCUs() * 8LL)));
	dictCollectKernel<<<grid, TPB, 0, s>>>(m->rows, m->d_start, m->d_positions, d_table, d_state);
	int st[4] = {0, 0, 0, 0};
	SMM_HIP_TRY(hipMemcpyAsync(st, d_state, sizeof(st), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	if (st[1] || st[0] > DICT_MAX) return no("the entries use more than 65536 distinct column offsets");
	const int k = st[0];
	if (k <= 0) return no("no entries");
	// PAT_EMPTY is INT_MIN: after an ascending sort of the whole table the dictionary is its last k entries
	size_t tempBytes = 0;
	SMM_HIP_TRY(rocprim::radix_sort_keys(nullptr, tempBytes, d_table.p, d_sorted.p, static_cast<size_t>(DICT_HCAP), 0, 32, s));
	SMM_TRY(temp.alloc(tempBytes ? tempBytes : 1));
	SMM_HIP_TRY(rocprim::radix_sort_keys(temp.p, tempBytes, d_table.p, d_sorted.p, static_cast<size_t>(DICT_HCAP), 0, 32, s));
	SMM_TRY(d_dict.alloc(static_cast<size_t>(k)));
	SMM_HIP_TRY(hipMemcpyAsync(d_dict, d_sorted.p + (DICT_HCAP - k), static_cast<size_t>(k) * sizeof(int), hipMemcpyDeviceToDevice, s));
	// 16 codes of padding: the kernel's 16-byte loads may start at the last entry
	const size_t nCodes = static_cast<size_t>(m->nnz) + 16;
	SMM_TRY(d_codes.alloc(nCodes));
	SMM_HIP_TRY(hipMemsetAsync(d_codes.p + m->nnz, 0, 16 * sizeof(unsigned short), s));
	dictEncodeKernel<<<grid, TPB, 0, s>>>(m->rows, k, d_dict, m->d_start, m->d_positions, d_codes, d_state.p + 2);
	SMM_HIP_TRY(hipMemcpyAsync(st, d_state, sizeof(st), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));  // also: the scratch buffers go back to the allocator when this scope ends
	if (st[2]) return no("an entry's offset is missing from the dictionary built from the entries");
	m->pat_k = k;
	m->pat_encoding = 1;
	m->d_pat_off = d_dict.detach();
	m->d_pat_codes = d_codes.detach();
	return SMM_HIP_OK;
}

// streamKnown: `s` is the stream the caller orders its work on (the first SpMV of a matrix): everything is enqueued there and only
// a few hundred bytes come back; otherwise (smm_hip_csr_set_kernel: no stream) the device is drained once and the library's stream used.
// quiet: the automatic attempt -- a matrix without a pattern is not an error then (no error text, SMM_HIP_ERR_INVALID still returned).
// pat_state: 0 not analysed, 1 usable, -1 refused by both encodings, -2 refused by the masks with the dictionary not tried yet (an
// automatic attempt with SMM_HIP_AUTO_DICT=0): an explicit request tries it then.
int ensurePattern(smm_hip_csr* m, hipStream_t s, bool streamKnown, bool quiet, bool masksOnly) {
	SMM_TRY(ensureCsrReady(m, s, streamKnown));
	std::lock_guard<std::mutex> lock(m->tileMutex);
	// masksOnly: a caller that can only use the row masks (the brick partition of the block preconditioners reads the grid from their
	// offsets) does not pay for the dictionary; an explicit request later still tries it (state -2)
	const bool dictWanted = !masksOnly && (!quiet || autoDictAllowed());
	const int state = m->pat_state.load(std::memory_order_acquire);
	if (state > 0) return SMM_HIP_OK;
	if (state == -1 || (state == -2 && !dictWanted)) {
		if (!quiet) setError("pattern SpMV: the entries of this matrix use more than %d distinct column offsets", DICT_MAX);
		return SMM_HIP_ERR_INVALID;
	}
	if (state == -3 && quiet) return SMM_HIP_ERR_INVALID;  // an automatic attempt ran out of resources before: only an explicit request tries again
	const bool masksTried = state == -2;
	// (the state turns -1 only when the matrix was really refused -- not before the allocations, whose failure says nothing about it)
	auto refuse = [quiet, m](const char* why, int newState) {
		m->pat_state.store(newState, std::memory_order_release);
		if (!quiet) setError("pattern SpMV: %s", why);
		return static_cast<int>(SMM_HIP_ERR_INVALID);
	};
	if (m->rows == 0 || m->nnz == 0) return refuse("empty matrix", -1);
	if (!streamKnown) {
		SMM_HIP_TRY(hipDeviceSynchronize());
		s = libStream();
	}
	const char* why = "";
	int st = masksTried ? static_cast<int>(SMM_HIP_ERR_INVALID) : tryMasks(m, s, &why);
	if (st == SMM_HIP_ERR_INVALID) {
		if (!dictWanted) return refuse(why, -2);
		st = tryDict(m, s, &why);
	}
	if (st == SMM_HIP_ERR_INVALID) return refuse(why, -1);
	if (st != SMM_HIP_OK) {  // a HIP failure (no memory for the masks / codes / sort scratch ...): nothing was learnt about the matrix
		if (quiet) m->pat_state.store(-3, std::memory_order_release);
		return st;
	}
	m->pat_state.store(1, std::memory_order_release);
	return SMM_HIP_OK;
}

// lanes per row the PATTERN family runs this (analysed) matrix with: the rule of smm_spmv.hip's lanesForAvg; constant diagonals
// (<= 32 entries per row): the kernel without values[] is the one-lane one
int patternLanesFor(const smm_hip_csr* m) {
	const double avg = m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 0.0;
	if (m->pat_state.load(std::memory_order_acquire) > 0 && m->pat_encoding == 0 && m->pat_const) return 1;
	return avg <= 24 ? 1 : avg <= 64 ? 2 : avg <= 128 ? 4 : 8;
}

// The automatic STREAM -> PATTERN switch, shared by launchSpmv's first-SpMV rule and the solvers' rule below.  Concurrent solves on one
// const matrix may arrive here together: one thread analyses (adoptMutex), the others wait and find the word already published.  The
// attempt is an optimisation: whatever goes wrong inside it (the matrix has no pattern; no memory for 8 bytes per row or 2 per entry or
// the sort's scratch) leaves the matrix on STREAM, clears HIP's sticky "last error" and is NOT reported -- the caller's SpMV or solve
// would have run fine without it.
void adoptPatternQuietly(const smm_hip_csr* cm, hipStream_t s) {
	auto* m = const_cast<smm_hip_csr*>(cm);
	std::lock_guard<std::mutex> lock(m->adoptMutex);
	if (m->kernelForced || m->family() != SMM_SPMV_STREAM) return;  // another thread switched it while this one waited
	SetupTrace trace("auto: PATTERN analysis, whole");
	int state = m->pat_state.load(std::memory_order_acquire);
	if (state < 0 && state != -2) return;
	// (state > 0: already analysed -- e.g. by a block preconditioner that read the grid from the offsets -- and only not adopted yet)
	const int st = state > 0 ? static_cast<int>(SMM_HIP_OK) : ensurePattern(m, s, true, true);
	if (st == SMM_HIP_OK) {
		m->setKernel(SMM_SPMV_PATTERN, patternLanesFor(m));
	} else if (st != SMM_HIP_ERR_INVALID) {
		(void)hipGetLastError();
		setError("");
	}
}

// A SOLVER is about to run many SpMVs with this matrix: matrices far below AUTO's single-SpMV threshold are worth the one-off analysis
// then (convection-diffusion 108^3, 8.7 M entries: the analysis costs a few hundred microseconds, BiCGStab's 660 SpMVs gain 3 ms of
// 37.7; profiles/r03/solver_pattern_small.txt).  Same rules as the AUTO step of launchSpmv otherwise: never against a forced kernel,
// never twice, SMM_HIP_AUTO_PATTERN=0 turns it off; SMM_HIP_SOLVER_PATTERN_MIN_NNZ moves the threshold (default 2^20).
int adoptPatternForSolver(const smm_hip_csr* m, int plannedIterations, hipStream_t s) {
	static const int allowed = [] {
		const char* env = getenv("SMM_HIP_AUTO_PATTERN");
		return env ? atoi(env) : 1;
	}();
	static const long long minNnz = [] {
		const char* env = getenv("SMM_HIP_SOLVER_PATTERN_MIN_NNZ");
		return env ? atoll(env) : (1LL << 20);
	}();
	if (!allowed || !m || m->rows <= 0 || m->kernelForced || m->family() != SMM_SPMV_STREAM) return SMM_HIP_OK;
	const int state = m->pat_state.load(std::memory_order_acquire);
	if (state == -1 || state == -3) return SMM_HIP_OK;
	if (plannedIterations >= 0 && plannedIterations < 16) return SMM_HIP_OK;  // (a few passes do not pay for a pass over positions[])
	const double avg = static_cast<double>(m->nnz) / m->rows;
	if (m->nnz < minNnz || avg > 64.0) return SMM_HIP_OK;
	adoptPatternQuietly(m, s);
	return SMM_HIP_OK;  // "no pattern" / "no memory for the analysis" are not failures of the solve
}

template <typename T>
static int patCap(const smm_hip_csr* m, int lanes) {
	const double avg = m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 1.0;
	const int rowsPerTile = TPB / std::min(lanes, WAVE);
	const double want = avg * rowsPerTile * 1.04 + 3;
	int nv = static_cast<int>((want + PatCfg<T>::PIECE - 1) / PatCfg<T>::PIECE);
	nv = std::max(1, std::min(nv, PatCfg<T>::NVMAX));
	return nv * PatCfg<T>::PIECE;
}

// SMM_HIP_PATTERN_VARIANT=0 keeps the pipelined row-per-lane kernel for every L (A/B measurements)
static bool patUseTile(int lanes) {
	static const int forced = [] {
		const char* env = getenv("SMM_HIP_PATTERN_VARIANT");
		return env ? atoi(env) : -1;
	}();
	if (forced == 0) return false;
	return lanes == 2 || lanes == 4;
}

// gathers per batch: a piece of p entries is walked in ceil(p / 16) batches of equal size (26 -> 2 x 13), in the three compiled sizes
static int patBatch(const smm_hip_csr* m, int lanes) {
	const double len = m->stream_mid_len > 0 ? m->stream_mid_len : (m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 1.0);
	const int p = std::max(1, static_cast<int>(std::ceil(len / lanes)));
	const int nb = (p + 15) / 16;
	int g = (p + nb - 1) / nb;
	if (const char* env = getenv("SMM_HIP_TILE_BATCH")) g = atoi(env);
	return g <= 8 ? 8 : g <= 13 ? 13 : 16;
}

// SMM_HIP_FULL_ROWS=0: the tile kernel's general path for every row (A/B measurements of the fast path for wavefronts of uniform rows).  Read once.
static int noFullRowsFlag() {
	static const int flag = [] {
		const char* env = getenv("SMM_HIP_FULL_ROWS");
		return env && atoi(env) == 0 ? SPMV_NO_FULL_ROWS : 0;
	}();
	return flag;
}

template <typename T, int L, int G>
static void launchPatTileG(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                           const int* doneFlag, hipStream_t s) {
	constexpr int RT = 64 * (TPB / WAVE / L);
	const int cap = m->pat_nnz_cap + 3;
	const size_t lds = static_cast<size_t>(cap + PatCfg<T>::PAD) * sizeof(T) + RT * 8 + (RT + 4) * 4 + MAXOFF * 4 + static_cast<size_t>(L - 1) * RT * sizeof(T) +
	                   4 * sizeof(T) + 32;
	static std::atomic<long long> occ{0};  // (the query is cached per LDS size, the override read once: smm_internal.h)
	int perCU = occupancyCached(occ, spmvPatternTileKernel<T, L, G>, TPB, lds, 4);
	if (forcedWgsPerCU() > 0) perCU = forcedWgsPerCU();
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();
	const int grid = std::max(1, std::min(std::min(m->pat_n_rowblocks, cus * perCU), NPART));
	const int nGroups = std::min(8, grid);
	const int chunkTiles = m->pat_chunk_tiles > 0 ? m->pat_chunk_tiles : (m->pat_n_rowblocks + nGroups - 1) / nGroups;
	spmvPatternTileKernel<T, L, G><<<grid, TPB, lds, s>>>(m->pat_n_rowblocks, cap, chunkTiles, m->cols, m->pat_k, m->d_pat_off,
	                                                     reinterpret_cast<const int2*>(m->d_pat_rowblocks), m->d_start, m->d_pat_masks, m->d_positions,
	                                                     static_cast<const T*>(m->d_values), (op & ~SPMV_LEAVE_ROOM) | spmvOutFlags(m, sizeof(T)) | noFullRowsFlag(), lhs,
	                                                     divisor, x, out, dotMode, w1, partials, doneFlag);
}

template <typename T, int L>
static void launchPatTile(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                          const int* doneFlag, hipStream_t s) {
	switch (patBatch(m, L)) {
	case 8: launchPatTileG<T, L, 8>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;
	case 13: launchPatTileG<T, L, 13>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;
	default: launchPatTileG<T, L, 16>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;
	}
}

template <typename T>
static void launchPatConst(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                           const int* doneFlag, hipStream_t s) {
	if (launchPatConstMarch<T>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)) return;  // grid-shaped: the 2.5-D form
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();  // room for the RCCL kernel beside A_loc (smm_dist.hip)
	op &= ~SPMV_LEAVE_ROOM;
	const int nTiles = (m->rows + TPB - 1) / TPB;
	static const int perCU = [] {
		const char* env = getenv("SMM_HIP_CONST_WGS_PER_CU");
		return env ? std::max(1, atoi(env)) : 8;
	}();
	const int grid = std::max(1, std::min(std::min(nTiles, cus * perCU), NPART));
	// the tiles' deal to the XCDs: one span of the farthest diagonal (a grid plane) per XCD in turn when that is many tiles but a small
	// part of the matrix -- the rule of buildRowBlocks (smm_spmv.hip) -- else one contiguous eighth each
	int chunkTiles = 0;
	const long long farTiles = m->pat_max_off / TPB;
	if (farTiles >= 256 && farTiles * 32 <= nTiles) chunkTiles = static_cast<int>(farTiles);
	if (const char* env = getenv("SMM_HIP_XCD_CHUNK_TILES")) chunkTiles = std::max(0, atoi(env));
	spmvPatternConstKernel<T><<<grid, TPB, 0, s>>>(m->rows, m->cols, m->pat_k, m->d_pat_off, m->d_pat_cval, m->d_pat_masks, chunkTiles, op | spmvOutFlags(m, sizeof(T)),
	                                              lhs, divisor, x, out, dotMode, w1, partials, doneFlag);
}

template <typename T, int L>
static void launchDict(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials, const int* doneFlag,
                       hipStream_t s) {
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();
	op &= ~SPMV_LEAVE_ROOM;
	constexpr int LW = L > WAVE ? WAVE : L;
	constexpr int RT = (WAVE / LW) * (TPB / WAVE);
	const int cap = m->pat_nnz_cap + 3;
	const bool dlds = m->pat_k <= DICT_LDS_MAX;
	const size_t lds = static_cast<size_t>(cap + PatCfg<T>::PAD) * sizeof(T) + (RT + 4) * 4 + static_cast<size_t>((cap + PatCfg<T>::PAD + 8 + 7) & ~7) * 2 +
	                   (dlds ? static_cast<size_t>((m->pat_k + 1) & ~1) * 4 : 0) + 4 * sizeof(T) + 32;
	static std::atomic<long long> occLds{0}, occMem{0};
	const int perCU = dlds ? occupancyCached(occLds, spmvDictKernel<T, L, true>, TPB, lds, 4) : occupancyCached(occMem, spmvDictKernel<T, L, false>, TPB, lds, 4);
	const int grid = std::max(1, std::min(std::min(m->pat_n_rowblocks, cus * perCU), NPART));
	const int flags = op | spmvOutFlags(m, sizeof(T));
	const int2* tiles = reinterpret_cast<const int2*>(m->d_pat_rowblocks);
	if (dlds) {
		spmvDictKernel<T, L, true><<<grid, TPB, lds, s>>>(m->pat_n_rowblocks, cap, m->cols, m->pat_k, m->d_pat_off, tiles, m->d_start, m->d_pat_codes, m->d_positions,
		                                                 static_cast<const T*>(m->d_values), flags, lhs, divisor, x, out, dotMode, w1, partials, doneFlag);
	} else {
		spmvDictKernel<T, L, false><<<grid, TPB, lds, s>>>(m->pat_n_rowblocks, cap, m->cols, m->pat_k, m->d_pat_off, tiles, m->d_start, m->d_pat_codes, m->d_positions,
		                                                  static_cast<const T*>(m->d_values), flags, lhs, divisor, x, out, dotMode, w1, partials, doneFlag);
	}
}

template <typename T, int L>
static void launchPat(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials, const int* doneFlag,
                      hipStream_t s) {
	if constexpr (L == 1) {
		if (m->pat_encoding == 0 && m->pat_const && !m->pat_const_off) {  // constant diagonals: no values[] either (one lane per row only)
			launchPatConst<T>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s);
			return;
		}
	}
	if constexpr (L == 1) {
		// workgroups per CU of the wave-private form (0 = off: the tile kernel below); measured best on the 512^3 Laplacian: 4 for fp64
		// (2.29 ms against 2.38 at 8 and 2.99 for the tile kernel), 8 for fp32 (1.40 against 1.55 at 4 and 1.50)
		static const int waveEnv = [] {
			const char* env = getenv("SMM_HIP_PATTERN_WAVE");
			return env ? atoi(env) : -1;
		}();
		const int waveForm = waveEnv >= 0 ? waveEnv : (sizeof(T) == 8 ? 4 : 8);
		// big grid-shaped matrices (masksMarchApplies): the march form (x through LDS windows and registers, smm_spmv_march.hip)
		if (m->pat_encoding == 0 && launchPatMasksMarch<T>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)) return;
		if (waveForm && m->pat_encoding == 0 && m->pat_k <= 16) {
			const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();
			const int nTiles = (m->rows + TPB - 1) / TPB;
			const int grid = std::max(1, std::min(std::min(nTiles, cus * waveForm), NPART));
			int chunkTiles = 0;
			const long long farTiles = m->pat_max_off / TPB;
			if (farTiles >= 256 && farTiles * 32 <= nTiles) chunkTiles = static_cast<int>(farTiles);
			if (const char* env = getenv("SMM_HIP_XCD_CHUNK_TILES")) chunkTiles = std::max(0, atoi(env));
			const int flags = (op & ~SPMV_LEAVE_ROOM) | spmvOutFlags(m, sizeof(T));
			if (m->pat_k <= 8) {
				spmvPatternWaveKernel<T, 8><<<grid, TPB, 0, s>>>(m->rows, m->cols, m->pat_k, m->d_pat_off, m->d_start, static_cast<const T*>(m->d_values), m->d_pat_masks, chunkTiles,
				                                                flags, lhs, divisor, x, out, dotMode, w1, partials, doneFlag);
			} else {  // 9 .. 16 entries per row: the same kernel with a slice of 16 values per row
				spmvPatternWaveKernel<T, 16><<<grid, TPB, 0, s>>>(m->rows, m->cols, m->pat_k, m->d_pat_off, m->d_start, static_cast<const T*>(m->d_values), m->d_pat_masks, chunkTiles,
				                                                 flags, lhs, divisor, x, out, dotMode, w1, partials, doneFlag);
			}
			return;
		}
	}
	if (m->pat_encoding == 1) {  // the dictionary encoding: one kernel form for every L
		launchDict<T, L>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s);
		return;
	}
	if constexpr (L == 2 || L == 4) {
		if (patUseTile(L)) {
			launchPatTile<T, L>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s);
			return;
		}
	}
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();
	op &= ~SPMV_LEAVE_ROOM;
	constexpr int LW = L > WAVE ? WAVE : L;
	constexpr int RT = (WAVE / LW) * (TPB / WAVE);
	const int cap = m->pat_nnz_cap + 3;
	const size_t lds = static_cast<size_t>(cap + PatCfg<T>::PAD) * sizeof(T) + RT * 8 + (RT + 4) * 4 + MAXOFF * 4 + 4 * sizeof(T) + 32;
	static std::atomic<long long> occ{0};
	const int perCU = occupancyCached(occ, spmvPatternKernel<T, L>, TPB, lds, 4);
	const int grid = std::max(1, std::min(std::min(m->pat_n_rowblocks, cus * perCU), NPART));
	spmvPatternKernel<T, L><<<grid, TPB, lds, s>>>(m->pat_n_rowblocks, cap, m->cols, m->pat_k, m->d_pat_off, reinterpret_cast<const int2*>(m->d_pat_rowblocks),
	                                             m->d_start, m->d_pat_masks, m->d_positions, static_cast<const T*>(m->d_values), op | spmvOutFlags(m, sizeof(T)), lhs, divisor, x, out, dotMode,
	                                             w1, partials, doneFlag);
}

// mirrors launchPat's choice (keep the two together)
const char* patternKernelDesc(const smm_hip_csr* m, int lanes, long long* bytes) {
	const long long s = m->dtype == SMM_DTYPE_F32 ? 4 : 8;
	const long long rows = m->rows, cols = m->cols, nnz = m->nnz;
	const long long vectors = cols * s + rows * s, startBytes = (rows + 1) * 4;
	const int L = std::min(lanes, WAVE);
	if (m->pat_encoding == 1) {
		*bytes = nnz * (s + 2) + startBytes + vectors;
		return "spmvDictKernel";
	}
	if (L == 1 && m->pat_const && !m->pat_const_off) {
		static const bool marchOn = [] {
			const char* env = getenv("SMM_HIP_CONST_MARCH");
			return env ? atoi(env) != 0 : true;
		}();
		const bool march = marchOn && constMarchApplies(m);
		// the row's mask (32 bits in the 2.5-D forms, 8 where the two-window kernel knows its five near offsets at compile time), x, out:
		// neither values[] nor start[]
		const bool bytesMasks = march && !m->march_clusters && m->d_pat_masks8 && m->pat_k - m->march_lo - m->march_hi == 5;
		*bytes = rows * (bytesMasks ? 1 : march ? 4 : 8) + vectors;
		return march ? (m->march_clusters ? "spmvPatternConstMarch3Kernel" : "spmvPatternConstMarchKernel") : "spmvPatternConstKernel";
	}
	*bytes = nnz * s + rows * 8 + startBytes + vectors;
	if (L == 1 && masksMarchApplies(m)) {
		static const bool masksMarchOn = [] {
			const char* env = getenv("SMM_HIP_MASKS_MARCH");
			return env ? atoi(env) != 0 : true;
		}();
		if (masksMarchOn) {
			*bytes = nnz * s + rows + (rows / 64 + 1) * 4 + vectors;  // values, one byte of mask per row, one start[] per 64 rows, x, out
			return "spmvPatternMasksMarchKernel";
		}
	}
	if (L == 1) {
		static const int waveEnv = [] {
			const char* env = getenv("SMM_HIP_PATTERN_WAVE");
			return env ? atoi(env) : -1;
		}();
		if (waveEnv != 0 && m->pat_k <= 16) return "spmvPatternWaveKernel";
	}
	if ((L == 2 || L == 4) && patUseTile(L)) return "spmvPatternTileKernel";
	return "spmvPatternKernel";
}

// mirrors launchPat: which launches read the tile table at all
static bool patNeedsTiles(const smm_hip_csr* m, int L) {
	if (L != 1 || m->pat_encoding != 0) return true;
	if (m->pat_const && !m->pat_const_off) return false;  // the constant-diagonal kernels
	static const int waveEnv = [] {
		const char* env = getenv("SMM_HIP_PATTERN_WAVE");
		return env ? atoi(env) : -1;
	}();
	return !(waveEnv != 0 && m->pat_k <= 16);  // the wave kernel and the masks march walk rows
}

// tiles for this family are cut for its own LDS capacity (values only): kept beside the STREAM family's table
static int buildPatternTiles(smm_hip_csr* m, int capNnz, int maxRows, hipStream_t s) {
	int* blocks = nullptr;
	int n = 0, chunk = 0;
	SMM_TRY(buildTileTable(m, capNnz, maxRows, s, &blocks, &n, &chunk));  // (never through the STREAM family's fields: a launch of another thread may be reading them)
	devFree(m->d_pat_rowblocks);
	m->d_pat_rowblocks = blocks;
	m->pat_n_rowblocks = n;
	m->pat_nnz_cap = capNnz;
	m->pat_max_rows = maxRows;
	m->pat_chunk_tiles = chunk;
	return SMM_HIP_OK;
}

template <typename T>
int launchSpmvPattern(const smm_hip_csr* m, int lanes, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                      const int* doneFlag, hipStream_t s) {
	auto* mm = const_cast<smm_hip_csr*>(m);
	SMM_TRY(ensurePattern(mm, s, true));
	const int L = std::min(lanes, WAVE);
	const int capNnz = patCap<T>(m, L) - 3;
	const int maxRows = TPB / L;
	if (patNeedsTiles(m, L)) {  // (the one-lane kernels of stencil shape -- gather, wave, march -- walk rows, not tiles)
		std::lock_guard<std::mutex> lock(mm->tileMutex);
		if (!m->d_pat_rowblocks || m->pat_nnz_cap != capNnz || m->pat_max_rows != maxRows) {
			SetupTrace trace("pattern: tile table");
			SMM_TRY(buildPatternTiles(mm, capNnz, maxRows, s));
		}
	}
	switch (L) {
	case 1: launchPat<T, 1>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;
	case 2: launchPat<T, 2>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;
	case 4: launchPat<T, 4>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;
	default: launchPat<T, 8>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;
	}
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template int launchSpmvPattern<float>(const smm_hip_csr*, int, int, const float*, const float*, const float*, float*, int, const float*, float*, const int*, hipStream_t);
template int launchSpmvPattern<double>(const smm_hip_csr*, int, int, const double*, const double*, const double*, double*, int, const double*, double*, const int*, hipStream_t);

}  // namespace smm

extern "C" int smm_hip_csr_pattern_allow_const(smm_hip_csr* m, int allow) {
	if (!m) {
		smm::setError("csr_pattern_allow_const: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	m->pat_const_off = allow == 0;
	return SMM_HIP_OK;
}

extern "C" int smm_hip_csr_pattern_info(const smm_hip_csr* m, int* encoding, int* offsets) {
	if (!m) {
		smm::setError("csr_pattern_info: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	if (encoding) *encoding = m->pat_state > 0 ? (m->pat_encoding == 1 ? SMM_PATTERN_CODES : m->pat_const && !m->pat_const_off ? SMM_PATTERN_CONST : SMM_PATTERN_MASKS) : SMM_PATTERN_NONE;
	if (offsets) *offsets = m->pat_state > 0 ? m->pat_k : 0;
	return SMM_HIP_OK;
}
