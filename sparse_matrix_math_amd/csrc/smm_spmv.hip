// smm_spmv.hip -- CSR SpMV for gfx950:  out[i] = op(lhs[i], sum_k values[k] * x[positions[k]])
//
// Replaces CSRMatrix<T>::rMultOp / rMult / rMultAdd / rMultSub (ref:1458-1515).  Two kernel families:
//
//  VECTOR  L adjacent lanes of a wavefront own one row (L = 1..64); the lanes stride over the row so that
//          positions[]/values[] are read coalesced, partial sums meet in a wave butterfly (__shfl_xor).
//          Rows are grid-strided.  Best for long rows and for tiny matrices.
//
//  STREAM  The matrix is cut once (at the first SpMV) into row tiles of <= 256/L whole rows and <= cap-3 nonzeros
//          (cap = 1024..4096 from the mean row length).  A persistent workgroup walks its tiles: the tile's
//          positions[] / values[] slices -- >95 % of the bytes of an SpMV -- are fetched ONE TILE AHEAD with
//          16-byte coalesced non-temporal loads into registers and stored to LDS (positions as byte offsets into
//          x); then lane (row, piece) walks its piece of its row out of LDS in batches of 8 independent x[]
//          gathers.  Rows run fastest across the lanes of a wavefront, so for banded / stencil matrices a gather
//          instruction reads adjacent columns (coalesced) although x itself is never staged.  The L pieces of a
//          row meet through wave shuffles, left to right.
//          With L == 1 the sum is formed in exactly the reference's order (ref:1484-1489): bit-identical
//          results.  With L > 1 the L piece sums are added left to right (deterministic).
//
// Both families can fuse one or two dot products of the freshly computed out[] into the epilogue (the p.Ap of
// CG ref:2354, ap.r0 / as.as / as.s of BiCGStab ref:2243, 2259-2261) so those vectors are not re-read: each
// workgroup writes its partial sums to partials[blockIdx.x]; consumers add NPART slots in a fixed order.
//
// No MFMA: 2 flops per 8-12 bytes, the path is HBM-bound.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "smm_device.h"
#include "smm_internal.h"

namespace smm {

constexpr int TPB = 256;  // threads per workgroup = 4 wavefronts

// values[] / positions[] are read exactly once per SpMV: stream them with the non-temporal policy so they do not push
// x[] (re-read ~nnz/row times) out of L2 / Infinity Cache.  SMM_SPMV_NO_NT builds use plain loads (A/B measurements).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
#ifdef SMM_SPMV_NO_NT
#define NT_LOAD(p) (*(p))
#else
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#endif

template <typename T>
__device__ __forceinline__ T applyOp(int op, const T* __restrict__ lhs, const T* __restrict__ divisor, int row, T dot) {
	if (op == SMM_OP_ASSIGN) return dot;
	// internal (SPMV_DIV_LHS / SPMV_ADD_DIV): the Jacobi apply x = rhs / diag (smm_precond.hip) folded into the row -- the same division
	// on the same operands as SpMV + apply, so the same bits
	if (op == SPMV_OP_DIV) return dot / divisor[row];
	const T l = lhs[row];
	if (op == SPMV_OP_ADD_DIV) return (l + dot) / divisor[row];  // the remote block of a row-partitioned SpMV: add to A_loc x, then divide
	return op == SMM_OP_ADD ? l + dot : l - dot;
}

// ---------------------------------------------------------------------------------------------------------
// VECTOR family
// ---------------------------------------------------------------------------------------------------------
template <typename T, int L>
__global__ __launch_bounds__(TPB) void spmvVectorKernel(int rows, const int* __restrict__ start, const int* __restrict__ positions,
                                                        const T* __restrict__ values, int op, const T* lhs, const T* __restrict__ divisor, const T* __restrict__ x, T* out,
                                                        int dotMode, const T* __restrict__ w1, T* __restrict__ partials,
                                                        const int* __restrict__ doneFlag) {
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	const int opFlags = op;
	op &= 0xFF;
	const int lane = threadIdx.x % L;
	const int rowsPerBlock = TPB / L;
	const int rowInBlock = threadIdx.x / L;
	T acc0 = T(0), acc1 = T(0);
	// all lanes of a group run the same trip count, so the butterfly below never sees an exited lane
	for (long long base = static_cast<long long>(blockIdx.x) * rowsPerBlock; base < rows; base += static_cast<long long>(gridDim.x) * rowsPerBlock) {
		const long long row = base + rowInBlock;
		T dot = T(0);
		if (row < rows) {
			const int b = start[row];
			const int e = start[row + 1];
			for (int k = b + lane; k < e; k += L) {
				dot = smmFma(values[k], x[positions[k]], dot);
			}
		}
		dot = groupSum<L>(dot);
		if (row < rows && lane == 0) {
			const T o = applyOp(op, lhs, divisor, static_cast<int>(row), dot);
			out[row] = o;
			if (dotMode == 1) {
				acc1 += o * w1[row];
			} else if (dotMode == 2) {
				acc0 += o * o;
				acc1 += o * w1[row];
			}
		}
	}
	if (dotMode) {
		if (dotMode == 2) {
			const T s0 = blockSum256(acc0, red);
			if (threadIdx.x == 0) partials[blockIdx.x] = s0;
		}
		const T s1 = blockSum256(acc1, red);
		if (threadIdx.x == 0) partials[(dotMode == 2 ? NPART : 0) + blockIdx.x] = s1;
		if (opFlags & SPMV_FINISH) lastBlockSums<T>(partials, NPART, dotMode == 2 ? 2 : 1, partials + PARTS_TOTALS, partsTicket(partials));
	}
}

// ---------------------------------------------------------------------------------------------------------
// STREAM family
// ---------------------------------------------------------------------------------------------------------
// A "row tile" is a run of at most TPB / L whole rows holding at most cap - 3 nonzeros; it is the unit of work of one
// workgroup.  (A row longer than that is a tile of its own and is summed straight from HBM.)  cap = 1024 * nv is chosen
// per matrix from the mean row length so that a tile of TPB / L average rows fits; LDS is sized at launch accordingly.
template <typename T>
struct StreamCfg {
	static constexpr int PIECE = 4 * TPB;                   // nonzeros staged per pass: one 16-byte load per lane and array
	static constexpr int NVMAX = sizeof(T) == 4 ? 4 : 2;    // passes a lane can hold in registers one tile ahead
	static constexpr int PAD = 16;                          // slack a gather batch may read past its piece
};

// One lane's slice of a staged tile, held in registers between the load (issued one tile ahead) and the LDS store
template <typename T>
struct Staged;
template <>
struct Staged<float> {
	i32x4 p[StreamCfg<float>::NVMAX];
	f32x4 v[StreamCfg<float>::NVMAX];
};
template <>
struct Staged<double> {
	i32x4 p[StreamCfg<double>::NVMAX];
	f64x2 v[2 * StreamCfg<double>::NVMAX];
};

// issue the loads of positions[a0 .. n1) / values[a0 .. n1) (16-byte aligned start a0) -- no wait here.  The caller
// guarantees a0 + cap <= nnzTotal rounded down to 4, i.e. the 16-byte pieces never run past the arrays.
template <typename T>
__device__ __forceinline__ void stageLoad(Staged<T>& r, int t, int nv, int a0, int n1, const int* __restrict__ positions,
                                          const T* __restrict__ values) {
#pragma unroll
	for (int v = 0; v < StreamCfg<T>::NVMAX; ++v) {
		const int i = a0 + 4 * (t + v * TPB);
		if (v < nv && i < n1) {
			r.p[v] = NT_LOAD(reinterpret_cast<const i32x4*>(positions + i));
			if constexpr (sizeof(T) == 4) {
				r.v[v] = NT_LOAD(reinterpret_cast<const f32x4*>(values + i));
			} else {
				r.v[2 * v] = NT_LOAD(reinterpret_cast<const f64x2*>(values + i));
				r.v[2 * v + 1] = NT_LOAD(reinterpret_cast<const f64x2*>(values + i + 2));
			}
		}
	}
}

// registers -> LDS.  positions are stored as BYTE offsets into x (col * sizeof(T)) so that a gather is one
// scalar-base + 32-bit-offset load with no address arithmetic.
template <typename T>
__device__ __forceinline__ void stageStore(const Staged<T>& r, int t, int nv, int a0, int n1, unsigned* sOff, T* sVal) {
#pragma unroll
	for (int v = 0; v < StreamCfg<T>::NVMAX; ++v) {
		const int li = 4 * (t + v * TPB);
		if (v < nv && a0 + li < n1) {
			*reinterpret_cast<i32x4*>(sOff + li) = r.p[v] * static_cast<int>(sizeof(T));
			if constexpr (sizeof(T) == 4) {
				*reinterpret_cast<f32x4*>(sVal + li) = r.v[v];
			} else {
				*reinterpret_cast<f64x2*>(sVal + li) = r.v[2 * v];
				*reinterpret_cast<f64x2*>(sVal + li + 2) = r.v[2 * v + 1];
			}
		}
	}
}

// value of lane `j` (wave-uniform j) broadcast to every lane
__device__ __forceinline__ float readLane(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }
__device__ __forceinline__ double readLane(double v, int j) {
	const long long bits = __double_as_longlong(v);
	const int lo = __builtin_amdgcn_readlane(static_cast<int>(bits), j);
	const int hi = __builtin_amdgcn_readlane(static_cast<int>(bits >> 32), j);
	return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
}

template <typename T>
__device__ __forceinline__ T gatherX(const T* __restrict__ x, unsigned byteOffset) {
#ifdef SMM_EXP_NOGATHER  // ablation builds only: what the kernel costs without its x[] loads
	return T(1) + T(byteOffset & 1u);
#elif defined(SMM_EXP_NTGATHER)  // ablation builds only: x[] gathers with the non-temporal policy
	return __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(x) + byteOffset));
#else
	return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(x) + byteOffset);
#endif
}

extern __shared__ __attribute__((aligned(16))) unsigned char smmDynLds[];

// rowBlocks holds {first row, start[first row]} per tile (nTiles + 1 entries, the last one {rows, nnz}).
//
// Per tile: (1) the tile's positions[] / values[] slice, fetched one tile ahead with 16-byte coalesced loads into
// registers, is stored to LDS (positions as byte offsets into x); (2) lane (row, piece) walks its piece of its row out of
// LDS in batches of 8 independent x[] gathers, rows run fastest across the lanes of a wavefront so the gathers of banded
// matrices are coalesced; (3) the L pieces of a row meet through wave shuffles, left to right.  Two workgroup barriers per
// tile (LDS ready / LDS free); the HBM stream of tile i+1 is in flight during (2) of tile i.
template <typename T, int L>
__global__ __launch_bounds__(TPB) void spmvStreamKernel(int nTiles, int cap, int chunkTiles, const int2* __restrict__ rowBlocks, const int* __restrict__ start,
                                                        const int* __restrict__ positions, const T* __restrict__ values, int opFlags, const T* lhs,
                                                        const T* __restrict__ divisor, const T* __restrict__ x, T* out, int dotMode, const T* __restrict__ w1,
                                                        T* __restrict__ partials, const int* __restrict__ doneFlag) {
	using Cfg = StreamCfg<T>;
	constexpr int GATHER = 8;
	const int op = opFlags & 0xFF;
	const bool ntOut = (opFlags & SPMV_NT_OUT) != 0;  // out[] of a large matrix is streamed, not cached (host decides)
	constexpr int LW = L > WAVE ? WAVE : L;  // lanes per row
	constexpr int RW = WAVE / LW;            // rows per wavefront
	constexpr int RT = RW * (TPB / WAVE);    // rows per tile
	// LDS carve-up (sized by the host): sVal[cap + PAD] | sOff[cap + PAD] | sStart[RT + 1] | red[4]
	T* sVal = reinterpret_cast<T*>(smmDynLds);
	unsigned* sOff = reinterpret_cast<unsigned*>(sVal + cap + Cfg::PAD);
	int* sStart = reinterpret_cast<int*>(sOff + cap + Cfg::PAD);
	T* red = reinterpret_cast<T*>(sStart + RT + 4);
	if (doneFlag && *doneFlag) return;

	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int rowInWave = lane % RW;
	const int piece = lane / RW;
	const int rl = (t >> 6) * RW + rowInWave;  // this lane's row within the tile
	const int nv = cap / Cfg::PIECE;
	T acc0 = T(0), acc1 = T(0);

	// every slot must always hold a valid byte offset: a batch of GATHER entries may run past the end of its piece and
	// the x[] loads of those extra entries are issued (their products are discarded)
	for (int i = t; i < cap + Cfg::PAD; i += TPB) {
		sOff[i] = 0u;
		sVal[i] = T(0);
	}

	// XCD-aware work split: workgroups b, b+8, b+16, ... share an XCD (and its L2), so each of the 8 groups owns one
	// contiguous eighth of the tiles and walks it with consecutive tiles resident together -- the x[] cache lines that
	// neighbouring tiles share are then served by that XCD's L2 instead of being fetched once per XCD.
	// (Placement only affects speed: any blockIdx -> XCD mapping gives the same result.)
	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int xcdGroup = blockIdx.x % nGroups;
	const int groupSlots = (static_cast<int>(gridDim.x) - xcdGroup + nGroups - 1) / nGroups;  // workgroups in this group
	// Tiles are dealt to the groups in chunks of `chunkTiles` consecutive tiles, round-robin: the j-th tile of group g is
	// tile ((j / chunkTiles) * nGroups + g) * chunkTiles + j % chunkTiles.  chunkTiles = ceil(nTiles / 8) gives each group one
	// contiguous eighth; for a large 3-D stencil the host passes one grid plane per chunk, so the 8 XCDs sweep 8 adjacent planes
	// at a time (buildRowBlocks explains the choice).
	const int tileEnd = nTiles;
	auto tileOf = [&](int j) {
		const int c = j / chunkTiles;
		const long long tIdx = (static_cast<long long>(c) * nGroups + xcdGroup) * chunkTiles + (j - c * chunkTiles);
		return tIdx < nTiles ? static_cast<int>(tIdx) : nTiles;
	};
	// tiles that end within `cap` entries of the end of the arrays are not staged (their 16-byte pieces could run past the
	// arrays); like over-long rows they are summed straight from HBM
	const int stageLimit = (rowBlocks[nTiles].y & ~3) - cap;
	int j = blockIdx.x / nGroups;
	int tile = tileOf(j);

	// software pipeline: the 16-byte loads of tile i+1 are issued (into registers) before tile i is summed out of LDS;
	// the tile descriptors are fetched one step further ahead
	Staged<T> regs;
	int ps = 0;
	int2 m0 = make_int2(0, 0), m1 = make_int2(0, 0), nm0 = make_int2(0, 0), nm1 = make_int2(0, 0);
	if (tile < tileEnd) {
		m0 = rowBlocks[tile];
		m1 = rowBlocks[tile + 1];
		const int t1 = tileOf(j + groupSlots);
		if (t1 < tileEnd) {
			nm0 = rowBlocks[t1];
			nm1 = rowBlocks[t1 + 1];
		}
		if (m1.y - m0.y <= cap - 3 && (m0.y & ~3) <= stageLimit) {
			stageLoad<T>(regs, t, nv, m0.y & ~3, m1.y, positions, values);
			if (t < m1.x - m0.x) ps = start[m0.x + t];
		}
	}
	__syncthreads();  // LDS initialised
	while (tile < tileEnd) {
		const int r0 = m0.x, n0 = m0.y, r1 = m1.x, n1 = m1.y;
		const int nrows = r1 - r0;
		const int a0 = n0 & ~3;  // 16-byte aligned element index for both arrays (fp64: 32-byte aligned)
		const bool direct = n1 - n0 > cap - 3 || a0 > stageLimit;
		if (!direct) {
			stageStore<T>(regs, t, nv, a0, n1, sOff, sVal);
			if (t < nrows) sStart[t] = ps - a0;
			if (t == 0) sStart[nrows] = n1 - a0;
		}
		ldsBarrier();  // the tile is in LDS (LDS-only barrier: global loads and stores stay in flight across it)
		// ---- prefetch this workgroup's next tile ----
		j += groupSlots;
		const int ntile = tileOf(j);
		const int2 m0n = nm0, m1n = nm1;
		if (ntile < tileEnd) {
			if (m1n.y - m0n.y <= cap - 3 && (m0n.y & ~3) <= stageLimit) {
				stageLoad<T>(regs, t, nv, m0n.y & ~3, m1n.y, positions, values);
				if (t < m1n.x - m0n.x) ps = start[m0n.x + t];
			}
			const int t2 = tileOf(j + groupSlots);
			if (t2 < tileEnd) {
				nm0 = rowBlocks[t2];
				nm1 = rowBlocks[t2 + 1];
			}
		}
		if (direct && nrows == 1) {
			// an over-long row: wavefront 0 streams it straight from HBM
			if (t < WAVE) {
				T dot = T(0);
				if constexpr (LW == 1) {
					// one-lane-per-row configurations promise the reference's left-to-right order (ref:1484-1489): the
					// wavefront fetches 64 entries at a time and folds them in lane order (every lane forms the same sum)
					for (int k0 = n0; k0 < n1; k0 += WAVE) {
						const int k = k0 + lane;
						const bool ok = k < n1;
						const T v = ok ? values[k] : T(0);
						const T xx = ok ? x[positions[k]] : T(0);
						const int cnt = min(WAVE, n1 - k0);
						for (int j = 0; j < cnt; ++j) {
							dot = smmFma(readLane(v, j), readLane(xx, j), dot);
						}
					}
				} else {
					for (int k = n0 + lane; k < n1; k += WAVE) {
						dot = smmFma(values[k], x[positions[k]], dot);
					}
					dot = groupSum<WAVE>(dot);
				}
				if (lane == 0) {
					const T o = applyOp(op, lhs, divisor, r0, dot);
					out[r0] = o;
					if (dotMode == 2) acc0 += o * o;
					if (dotMode) acc1 += o * w1[r0];
				}
			}
		} else if (direct) {
			// one of the last tiles of the matrix: one lane per row, left to right, straight from HBM
			if (t < nrows) {
				const int row = r0 + t;
				const int e = start[row + 1];
				T dot = T(0);
				for (int k = start[row]; k < e; ++k) {
					dot = smmFma(values[k], x[positions[k]], dot);
				}
				const T o = applyOp(op, lhs, divisor, row, dot);
				out[row] = o;
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		} else {
			// ---- row sums from LDS: lane -> (row, piece), rows fastest across lanes so that the x gather is coalesced ----
			T dot = T(0);
			if (rl < nrows) {
				const int b = sStart[rl];
				const int e = sStart[rl + 1];
				int kb = b, ke = e;
				if (LW > 1) {
					const int piecelen = (e - b + LW - 1) / LW;
					kb = b + piece * piecelen;
					ke = min(e, kb + piecelen);
				}
				// batches of GATHER entries: the x[] gathers of a batch are all issued before the first multiply-add, so a lane
				// keeps GATHER independent loads in flight; the sum itself stays strictly left to right.  Entries past the end
				// of the piece are loaded too (valid offsets, see above) but never added.
				int k = kb;
				for (; k + GATHER <= ke; k += GATHER) {  // full batches: no predication at all
					unsigned off[GATHER];
					T xv[GATHER], vv[GATHER];
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						off[u] = sOff[k + u];
						vv[u] = sVal[k + u];
					}
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						xv[u] = gatherX<T>(x, off[u]);
					}
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						dot = smmFma(vv[u], xv[u], dot);
					}
				}
				if (k < ke) {  // last, partial batch
					const int nvalid = ke - k;
					unsigned off[GATHER];
					T xv[GATHER], vv[GATHER];
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						off[u] = sOff[k + u];
						vv[u] = sVal[k + u];
					}
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						xv[u] = gatherX<T>(x, off[u]);
					}
#pragma unroll
					for (int u = 0; u < GATHER; ++u) {
						const T next = smmFma(vv[u], xv[u], dot);
						dot = u < nvalid ? next : dot;
					}
				}
			}
			if (LW > 1) {
				// pieces of one row sit RW lanes apart in the same wavefront; add them left to right: ((p0 + p1) + p2) + ...
				T total = dot;
#pragma unroll
				for (int q = 1; q < LW; ++q) {
					total += __shfl(dot, rowInWave + q * RW, WAVE);
				}
				dot = total;
			}
			if (piece == 0 && rl < nrows) {
				const int row = r0 + rl;
				const T o = applyOp(op, lhs, divisor, row, dot);
#ifdef SMM_EXP_NOOUT  // ablation builds only: the kernel without its out[] stream
				if (o == T(123.456)) out[row] = o;
#else
				storeOut(out + row, o, ntOut);
#endif
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		}
		ldsBarrier();  // every lane is done with the LDS copy of this tile
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
		// consumers always add NPART slots per quantity: clear the ones no workgroup of this (smaller) grid owns
		for (int i = gridDim.x + blockIdx.x * TPB + t; i < NPART; i += gridDim.x * TPB) {
			partials[i] = T(0);
			if (dotMode == 2) partials[NPART + i] = T(0);
		}
		if (opFlags & SPMV_FINISH) lastBlockSums<T>(partials, NPART, dotMode == 2 ? 2 : 1, partials + PARTS_TOTALS, partsTicket(partials));
	}
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// TILE kernel: the STREAM family's kernel for rows of up to ~128 nonzeros (L = 1, 2 or 4 pieces per row).
//
// Same data path as spmvStreamKernel (tile of whole rows -> 16-byte coalesced non-temporal loads -> LDS -> lane-per-row gathers),
// two things changed, both measured on the benchmark matrix with tools/spmv_lab.hip (profiles/r02/spmv_variants.txt):
//   * the L pieces of a row live in DIFFERENT waves: wave w gathers piece w % L of the 64 consecutive rows of row group w / L, so a
//     gather instruction reads ONE window of 64 adjacent columns (256 B: 2-3 cache lines) instead of L windows of 64 / L columns --
//     11 % fewer L1->L2 requests (53.2 M vs 59.8 M per launch on the benchmark matrix, profiles/r02/pmc_spmv_c3_summary.txt) and
//     64-lane-wide returns; the piece sums meet in LDS and are added left to right, exactly the order of the in-wave form: same bits;
//   * no software pipelining inside the workgroup: a tile is loaded, stored to LDS and summed, then the next one.  A wave's loads
//     return in order, so a gather issued behind the next tile's stream loads cannot return before them: the prefetch bought nothing
//     (mode0 vs mode2 of the lab) and cost 32 VGPRs that are now free for deeper gather batches; the overlap comes from the other
//     workgroups of the CU.
// G (gathers in flight per lane and batch) is chosen per matrix so that a piece is walked in the fewest, evenly filled batches (a piece
// of 26 entries: 2 x 13, not 3 x 8 + 2): the number of gather round trips per tile is what the tile time follows.
// With L == 1 the mapping is the row-per-lane mapping of spmvStreamKernel and the sum is the reference's left-to-right sum (ref:1484-1489).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct TileCfg {
	static constexpr int PIECE = 4 * TPB;
	static constexpr int NVMAX = sizeof(T) == 4 ? 7 : 5;  // staging passes of 1024 entries (the last one may be partial)
	static constexpr int PAD = 16;
	// three workgroups per CU: 160 KiB / 3, less the small arrays behind the tile
	static constexpr int LDS_BUDGET = (160 * 1024) / 3 - 2048;
	static constexpr int CAP_MAX = LDS_BUDGET / static_cast<int>(sizeof(T) + 4) - PAD;  // 6554 (fp32) / 4364 (fp64) nonzeros
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T, int L, int G>
__global__ __launch_bounds__(TPB) void spmvTileKernel(int nTiles, int cap, int chunkTiles, const int2* __restrict__ rowBlocks, const int* __restrict__ start,
                                                      const int* __restrict__ positions, const T* __restrict__ values, int opFlags, const T* lhs,
                                                      const T* __restrict__ divisor, const T* __restrict__ x, T* out, int dotMode, const T* __restrict__ w1,
                                                      T* __restrict__ partials, const int* __restrict__ doneFlag) {
	using Cfg = TileCfg<T>;
	static_assert(L == 1 || L == 2 || L == 4, "pieces per row");
	static_assert(G <= Cfg::PAD, "a batch may read G - 1 slots past its piece");
	constexpr int NVMAX = Cfg::NVMAX;
	// ({offset, value} pairs in ONE LDS array, read with a single ds_read_b64, were tried for fp32: 2-3 % slower -- tile_kernel_tuning.txt)
	constexpr bool PAIRS = false;
	const int op = opFlags & 0xFF;
	const bool ntOut = (opFlags & SPMV_NT_OUT) != 0;
	constexpr int GROUPS = TPB / WAVE / L;  // row groups of 64 rows
	constexpr int RT = 64 * GROUPS;         // rows per tile
	// LDS (sized by the host): tile [fp32: pairs[cap + PAD] | fp64: sVal[cap + PAD], sOff[cap + PAD]] | sStart[RT + 4] | sPart[(L - 1) * RT] | red[4]
	uint2* sPair = reinterpret_cast<uint2*>(smmDynLds);
	T* sVal = reinterpret_cast<T*>(smmDynLds);
	unsigned* sOff = reinterpret_cast<unsigned*>(sVal + cap + Cfg::PAD);
	int* sStart = PAIRS ? reinterpret_cast<int*>(sPair + cap + Cfg::PAD) : reinterpret_cast<int*>(sOff + cap + Cfg::PAD);
	T* sPart = reinterpret_cast<T*>(sStart + RT + 4);
	T* red = sPart + (L > 1 ? (L - 1) * RT : 0);
	if (doneFlag && *doneFlag) return;

	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int wave = t >> 6;
	const int piece = wave % L;
	const int rl = (wave / L) * 64 + lane;  // this lane's row within the tile
	const int nv = (cap + Cfg::PIECE - 1) / Cfg::PIECE;
	T acc0 = T(0), acc1 = T(0);
	// every slot must always hold a valid byte offset: a batch may run past the end of its piece (those products are discarded)
	for (int i = t; i < cap + Cfg::PAD; i += TPB) {
		if constexpr (PAIRS) {
			sPair[i] = make_uint2(0u, 0u);
		} else {
			sOff[i] = 0u;
			sVal[i] = T(0);
		}
	}
	// XCD-aware work split, as in spmvStreamKernel
	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int xcdGroup = blockIdx.x % nGroups;
	const int groupSlots = (static_cast<int>(gridDim.x) - xcdGroup + nGroups - 1) / nGroups;
	auto tileOf = [&](int j) {
		const int c = j / chunkTiles;
		const long long tIdx = (static_cast<long long>(c) * nGroups + xcdGroup) * chunkTiles + (j - c * chunkTiles);
		return tIdx < nTiles ? static_cast<int>(tIdx) : nTiles;
	};
	// a tile is staged in whole 16-byte pieces from its aligned start: those may run (cap + 3 at most) past the tile, never past the arrays
	const int stageLimit = (rowBlocks[nTiles].y & ~3) - (cap + 4);
	int j = blockIdx.x / nGroups;
	int tile = tileOf(j);
	int2 m0 = make_int2(0, 0), m1 = make_int2(0, 0);
	if (tile < nTiles) {
		m0 = rowBlocks[tile];
		m1 = rowBlocks[tile + 1];
	}
	__syncthreads();  // LDS initialised
	while (tile < nTiles) {
		const int r0 = m0.x, n0 = m0.y, r1 = m1.x, n1 = m1.y;
		const int nrows = r1 - r0;
		const int a0 = n0 & ~3;
		const bool direct = n1 - n0 > cap - 3 || a0 > stageLimit;
		if (!direct) {
			// the tile's slices of positions[] / values[]: 16-byte coalesced non-temporal loads, straight on to LDS
			i32x4 rp[NVMAX];
			typename Pack16<T>::V rv[NVMAX * (sizeof(T) == 4 ? 1 : 2)];
#pragma unroll
			for (int v = 0; v < NVMAX; ++v) {
				const int i = a0 + 4 * (t + v * TPB);
				if (v < nv && i < n1) {
					rp[v] = NT_LOAD(reinterpret_cast<const i32x4*>(positions + i));
					if constexpr (sizeof(T) == 4) {
						rv[v] = NT_LOAD(reinterpret_cast<const f32x4*>(values + i));
					} else {
						rv[2 * v] = NT_LOAD(reinterpret_cast<const f64x2*>(values + i));
						rv[2 * v + 1] = NT_LOAD(reinterpret_cast<const f64x2*>(values + i + 2));
					}
				}
			}
			const int ps = t < nrows ? start[r0 + t] : 0;
#pragma unroll
			for (int v = 0; v < NVMAX; ++v) {
				const int li = 4 * (t + v * TPB);
				if (v < nv && a0 + li < n1) {
					const i32x4 bo = rp[v] * static_cast<int>(sizeof(T));  // byte offsets into x: a gather is base + 32-bit offset
					if constexpr (PAIRS) {
						u32x4 lo, hi;
						lo.x = static_cast<unsigned>(bo.x);
						lo.y = __float_as_uint(rv[v].x);
						lo.z = static_cast<unsigned>(bo.y);
						lo.w = __float_as_uint(rv[v].y);
						hi.x = static_cast<unsigned>(bo.z);
						hi.y = __float_as_uint(rv[v].z);
						hi.z = static_cast<unsigned>(bo.w);
						hi.w = __float_as_uint(rv[v].w);
						*reinterpret_cast<u32x4*>(sPair + li) = lo;
						*reinterpret_cast<u32x4*>(sPair + li + 2) = hi;
					} else {
						*reinterpret_cast<i32x4*>(sOff + li) = bo;
						if constexpr (sizeof(T) == 4) {
							*reinterpret_cast<f32x4*>(sVal + li) = rv[v];
						} else {
							*reinterpret_cast<f64x2*>(sVal + li) = rv[2 * v];
							*reinterpret_cast<f64x2*>(sVal + li + 2) = rv[2 * v + 1];
						}
					}
				}
			}
			if (t < nrows) sStart[t] = ps - a0;
			if (t == 0) sStart[nrows] = n1 - a0;
		}
		ldsBarrier();  // the tile is in LDS
		// descriptors of this workgroup's next tile (two short loads, in flight during the gathers)
		j += groupSlots;
		const int ntile = tileOf(j);
		int2 m0n = make_int2(0, 0), m1n = make_int2(0, 0);
		if (ntile < nTiles) {
			m0n = rowBlocks[ntile];
			m1n = rowBlocks[ntile + 1];
		}
		if (direct && nrows == 1) {
			// an over-long row: wavefront 0 streams it straight from HBM
			if (t < WAVE) {
				T dot = T(0);
				if constexpr (L == 1) {
					for (int k0 = n0; k0 < n1; k0 += WAVE) {  // the reference's left-to-right order (ref:1484-1489)
						const int k = k0 + lane;
						const bool ok = k < n1;
						const T v = ok ? values[k] : T(0);
						const T xx = ok ? x[positions[k]] : T(0);
						const int cnt = min(WAVE, n1 - k0);
						for (int q = 0; q < cnt; ++q) dot = smmFma(readLane(v, q), readLane(xx, q), dot);
					}
				} else {
					for (int k = n0 + lane; k < n1; k += WAVE) dot = smmFma(values[k], x[positions[k]], dot);
					dot = groupSum<WAVE>(dot);
				}
				if (lane == 0) {
					const T o = applyOp(op, lhs, divisor, r0, dot);
					out[r0] = o;
					if (dotMode == 2) acc0 += o * o;
					if (dotMode) acc1 += o * w1[r0];
				}
			}
		} else if (direct) {
			// one of the last tiles of the matrix: one lane per row, left to right, straight from HBM
			if (t < nrows) {
				const int row = r0 + t;
				// (in the PIECE structure of the staged path -- L chains added left to right: the same bits whatever the tile cut; r02-r05 walked
				// these rows as one chain)
				const int b = start[row], e = start[row + 1];
				const int piecelen = (e - b + L - 1) / L;
				T dot = T(0);
#pragma unroll
				for (int q = 0; q < L; ++q) {
					const int kb = b + q * piecelen, ke = min(e, kb + piecelen);
					T pd = T(0);
					for (int k = kb; k < ke; ++k) pd = smmFma(values[k], x[positions[k]], pd);
					dot = q == 0 ? pd : dot + pd;
				}
				const T o = applyOp(op, lhs, divisor, row, dot);
				out[row] = o;
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		} else {
			// ---- piece sums out of LDS: lane = row of the group, wave = (group, piece) ----
			T dot = T(0);
			int kb = 0, ke = 0;
			if (rl < nrows) {
				const int b = sStart[rl];
				const int e = sStart[rl + 1];
				kb = b;
				ke = e;
				if (L > 1) {
					const int piecelen = (e - b + L - 1) / L;
					kb = b + piece * piecelen;
					ke = min(e, kb + piecelen);
				}
			}
			for (int k = kb; k < ke; k += G) {
				unsigned off[G];
				T xv[G], vv[G];
#pragma unroll
				for (int u = 0; u < G; ++u) {
					if constexpr (PAIRS) {
						const uint2 e2 = sPair[k + u];
						off[u] = e2.x;
						vv[u] = __uint_as_float(e2.y);
					} else {
						off[u] = sOff[k + u];
						vv[u] = sVal[k + u];
					}
				}
#pragma unroll
				for (int u = 0; u < G; ++u) xv[u] = gatherX<T>(x, off[u]);
				const int nvalid = ke - k;
#pragma unroll
				for (int u = 0; u < G; ++u) {
					const T next = smmFma(vv[u], xv[u], dot);
					dot = u < nvalid ? next : dot;
				}
			}
			if (L > 1) {
				// pieces of a row meet in LDS and are added left to right: ((p0 + p1) + p2) + p3
				if (piece > 0 && rl < nrows) sPart[(piece - 1) * RT + rl] = dot;
				ldsBarrier();
				if (piece == 0 && rl < nrows) {
#pragma unroll
					for (int q = 1; q < L; ++q) dot += sPart[(q - 1) * RT + rl];
				}
			}
			if (piece == 0 && rl < nrows) {
				const int row = r0 + rl;
				const T o = applyOp(op, lhs, divisor, row, dot);
				storeOut(out + row, o, ntOut);
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		}
		ldsBarrier();  // every lane is done with the LDS copy of this tile
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

// ---- tile table, built ON THE DEVICE -----------------------------------------------------------------------------------------
// Rows are cut into tiles of <= capNnz nonzeros and <= maxRows whole rows (a row longer than capNnz is a tile of its own).  A
// greedy cut from row 0 is a sequential chain over the whole matrix, so the row range is first divided at fixed seams -- the rows
// whose start[] falls into [c G, (c+1) G), G = 64 tiles' worth of nonzeros, found by binary search -- and each of these
// super-chunks is cut greedily by one thread (every step is a binary search over at most maxRows + 1 entries of start[], not a
// walk over rows).  A seam costs at most one under-filled tile per 64.  Two passes (count, exclusive scan, write) give a dense
// table.  start[] never travels to the host (537 MB and a 134 M-iteration host loop for the 512^3 Laplacian before); everything is
// enqueued on the CALLER's stream and only the tile count (8 bytes) comes back.
constexpr int TILES_PER_CHUNK = 64;

__device__ __forceinline__ int firstRowAtOrAfter(const int* __restrict__ start, int rows, long long target) {
	int lo = 0, hi = rows;  // smallest r in [0, rows] with start[r] >= target (start[rows] = nnz is the sentinel)
	while (lo < hi) {
		const int mid = lo + (hi - lo) / 2;
		if (start[mid] >= target) hi = mid;
		else lo = mid + 1;
	}
	return lo;
}

// offsets == nullptr: counts[c] = tiles of super-chunk c.  Otherwise: tiles of chunk c are written from tiles[offsets[c]].
__global__ void tileCutKernel(int rows, const int* __restrict__ start, int capNnz, int maxRows, long long chunkNnz, int nChunks, int* __restrict__ counts,
                              const int* __restrict__ offsets, int2* __restrict__ tiles) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= nChunks) return;
	int r = firstRowAtOrAfter(start, rows, static_cast<long long>(c) * chunkNnz);
	const int rEnd = c == nChunks - 1 ? rows : firstRowAtOrAfter(start, rows, static_cast<long long>(c + 1) * chunkNnz);
	int n = 0;
	int at = offsets ? offsets[c] : 0;
	while (r < rEnd) {
		const int base = start[r];
		if (offsets) tiles[at + n] = make_int2(r, base);
		++n;
		// largest e in (r, min(rEnd, r + maxRows)] with start[e] - base <= capNnz; at least r + 1 (an over-long row stands alone)
		int lo = r + 1, hi = min(rEnd, r + maxRows);
		while (lo < hi) {
			const int mid = lo + (hi - lo + 1) / 2;
			if (start[mid] - base <= capNnz) lo = mid;
			else hi = mid - 1;
		}
		r = lo;
	}
	if (!offsets) counts[c] = n;
	if (offsets && c == nChunks - 1) tiles[offsets[nChunks]] = make_int2(rows, start[rows]);  // closing sentinel {rows, nnz}
}

// info[1] = the farthest column a middle row touches, as a distance from the row
__global__ void farColumnKernel(int rows, const int* __restrict__ start, const int* __restrict__ positions, int* __restrict__ info) {
	const int mid = rows / 2;
	int far = 0;
	for (int k = start[mid] + threadIdx.x; k < start[mid + 1]; k += blockDim.x) far = max(far, abs(positions[k] - mid));
	if (far) atomicMax(info + 1, far);
	if (threadIdx.x == 0) info[0] = start[mid + 1] - start[mid];  // length of a typical (interior) row
}

// The cut itself: tiles[0 .. *nTiles] = {first row, start[first row]} with the closing sentinel {rows, nnz}; the table is allocated here
// (devAlloc) and owned by the caller.  Also used for the row blocks of the block preconditioners (smm_precond_block.hip).
int cutRows(const int* d_start, int rows, long long nnz, int capNnz, int maxRows, hipStream_t s, int2** tiles, int* nTiles, int tilesPerChunk) {
	*tiles = nullptr;
	*nTiles = 0;
	const long long chunkNnz = static_cast<long long>(tilesPerChunk > 0 ? tilesPerChunk : TILES_PER_CHUNK) * capNnz;
	const int nChunks = static_cast<int>(nnz / chunkNnz) + 1;
	DevBuf<int> counts;
	SMM_TRY(counts.alloc(static_cast<size_t>(nChunks) + 1));
	SMM_HIP_TRY(hipMemsetAsync(counts, 0, (static_cast<size_t>(nChunks) + 1) * sizeof(int), s));
	const int grid = (nChunks + 63) / 64;
	tileCutKernel<<<grid, 64, 0, s>>>(rows, d_start, capNnz, maxRows, chunkNnz, nChunks, counts, nullptr, nullptr);
	size_t tempBytes = 0;
	SMM_HIP_TRY(rocprim::exclusive_scan(nullptr, tempBytes, counts.p, counts.p, 0, static_cast<size_t>(nChunks) + 1, rocprim::plus<int>(), s));
	DevBuf<char> temp;
	SMM_TRY(temp.alloc(tempBytes ? tempBytes : 1));
	SMM_HIP_TRY(rocprim::exclusive_scan(temp.p, tempBytes, counts.p, counts.p, 0, static_cast<size_t>(nChunks) + 1, rocprim::plus<int>(), s));
	int n = 0;
	SMM_HIP_TRY(hipMemcpyAsync(&n, counts.p + nChunks, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));  // the caller's stream: the table's size decides the allocation and the grid
	SMM_TRY(devAlloc(reinterpret_cast<void**>(tiles), (static_cast<size_t>(n) + 1) * sizeof(int2)));
	tileCutKernel<<<grid, 64, 0, s>>>(rows, d_start, capNnz, maxRows, chunkNnz, nChunks, nullptr, counts, *tiles);
	SMM_HIP_TRY(hipGetLastError());
	// One more wait, once per table (r05): the callers publish the table in the handle as soon as this returns, and a
	// concurrent solve of the same const matrix on ANOTHER stream would otherwise launch a kernel that reads a table still being written
	// on `s` (the allocator's quarantine only protects the freed scratch buffers, not readers of the new table).
	SMM_HIP_TRY(hipStreamSynchronize(s));
	*nTiles = n;
	return SMM_HIP_OK;
}

int buildTileTable(const smm_hip_csr* m, int capNnz, int maxRows, hipStream_t s, int** blocks, int* nBlocks, int* chunkTiles) {
	const int rows = m->rows;
	*blocks = nullptr;
	*nBlocks = 0;
	*chunkTiles = 0;
	DevBuf<int> info;
	SMM_TRY(info.alloc(2));
	SMM_HIP_TRY(hipMemsetAsync(info, 0, 2 * sizeof(int), s));
	if (rows > 0) farColumnKernel<<<1, 64, 0, s>>>(rows, m->d_start, m->d_positions, info);
	int far = 0;
	SMM_HIP_TRY(hipMemcpyAsync(&far, info.p + 1, sizeof(int), hipMemcpyDeviceToHost, s));
	int2* tiles = nullptr;
	int nTiles = 0;
	SMM_TRY(cutRows(m->d_start, rows, m->nnz, capNnz, maxRows, s, &tiles, &nTiles));  // synchronises s: `far` has arrived
	*blocks = reinterpret_cast<int*>(tiles);
	*nBlocks = nTiles;
	// How the tiles are dealt to the 8 XCDs.  The farthest column a middle row touches tells how far apart (in rows) two uses of
	// the same x[] line are.  When that distance is many tiles but a small fraction of the matrix (3-D stencils: one grid plane),
	// the tiles are dealt one such span per XCD, round-robin, so the 8 XCDs sweep 8 adjacent planes together instead of 8 regions
	// a gigabyte apart: measured -8 % on the 512^3 Laplacian (3.55 -> 3.28 ms), neutral on smaller grids; with far offsets that are
	// a large fraction of the matrix (the banded-random benchmark matrix) contiguous eighths are best (tools/sweep_chunk.sh).
	if (rows > 0 && nTiles > 0) {
		const double rowsPerTile = static_cast<double>(rows) / nTiles;
		const long long farTiles = static_cast<long long>(far / rowsPerTile);
		if (farTiles >= 256 && farTiles * 32 <= nTiles) *chunkTiles = static_cast<int>(farTiles);
	}
	if (const char* env = getenv("SMM_HIP_XCD_CHUNK_TILES")) *chunkTiles = std::max(0, atoi(env));  // tuning override
	return SMM_HIP_OK;
}

// the STREAM family's table (caller holds tileMutex).  The old table goes back to the allocator's quarantine: launches already
// enqueued with it stay valid
int buildRowBlocks(smm_hip_csr* m, int capNnz, int maxRows, hipStream_t s) {
	int* blocks = nullptr;
	int n = 0, chunk = 0;
	SMM_TRY(buildTileTable(m, capNnz, maxRows, s, &blocks, &n, &chunk));
	devFree(m->d_rowblocks);
	m->d_rowblocks = blocks;
	m->n_rowblocks = n;
	m->stream_nnz_cap = capNnz;
	m->stream_max_rows = maxRows;
	m->stream_chunk_tiles = chunk;
	return SMM_HIP_OK;
}

// Which kernel of the STREAM family: the TILE kernel for 2 or 4 pieces per row (benchmark matrix 0.825 -> 0.796 ms on one box,
// profiles/r02/stream_kernels_ab.txt); with one lane per row (every stencil matrix) the two kernels map rows to lanes identically and
// the pipelined one is 1-2 % faster (512^3 fp64 3.28 vs 3.33 ms), so it stays.  SMM_HIP_STREAM_VARIANT = 0 / 1 forces the pipelined /
// the TILE kernel wherever it exists (A/B measurements).
static bool useTileKernel(int lanes) {
	static const int forced = [] {
		const char* env = getenv("SMM_HIP_STREAM_VARIANT");
		return env ? atoi(env) : -1;
	}();
	if (forced == 0) return false;
	if (forced == 1) return lanes == 1 || lanes == 2 || lanes == 4;
	return lanes == 2 || lanes == 4;
}
static int tileRows(int lanes) { return 64 * (TPB / WAVE / lanes); }

// gathers per batch: a piece of p entries is walked in ceil(p / 16) batches of equal size (26 -> 2 x 13; 13 -> 1 x 13; 7 -> 1 x 7)
static int tileBatch(const smm_hip_csr* m, int lanes) {
	// a typical row: the middle row of the matrix (read when the tile table is built), else the mean
	const double len = m->stream_mid_len > 0 ? m->stream_mid_len : (m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 1.0);
	const int p = std::max(1, static_cast<int>(std::ceil(len / lanes)));
	const int nb = (p + 15) / 16;
	int g = (p + nb - 1) / nb;
	if (const char* env = getenv("SMM_HIP_TILE_BATCH")) g = atoi(env);  // tuning override
	return std::max(4, std::min(16, g));
}

// LDS capacity (nonzeros) of a tile of TPB / lanes average rows, in staging passes of 1024
template <typename T>
static int streamCap(const smm_hip_csr* m, int lanes) {
	if (useTileKernel(lanes)) {
		// exactly what tileRows() typical rows need (every lane of every wave then has a row), within the LDS budget of three
		// workgroups per CU; a multiple of 4 so that the staged 16-byte pieces stay whole
		const double avg = m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 1.0;
		const double len = std::max(avg, static_cast<double>(m->stream_mid_len));
		int cap = static_cast<int>(len * tileRows(lanes)) + 8;
		if (const char* env = getenv("SMM_HIP_STREAM_NV")) cap = atoi(env) * TileCfg<T>::PIECE;
		cap = std::max(256, std::min(cap, TileCfg<T>::CAP_MAX));
		return (cap + 3) & ~3;
	}
	const double avg = m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 1.0;
	const int rowsPerTile = TPB / std::min(lanes, WAVE);
	const double want = avg * rowsPerTile * 1.04 + 3;
	int nv = static_cast<int>((want + StreamCfg<T>::PIECE - 1) / StreamCfg<T>::PIECE);
	if (const char* env = getenv("SMM_HIP_STREAM_NV")) nv = atoi(env);  // tuning override (tools/spmv_sweep.py)
	nv = std::max(1, std::min(nv, StreamCfg<T>::NVMAX));
	return nv * StreamCfg<T>::PIECE;
}

static int lanesForAvg(double avg, int family) {
	if (family == SMM_SPMV_PATTERN) return avg <= 24 ? 1 : avg <= 64 ? 2 : avg <= 128 ? 4 : 8;
	if (family == SMM_SPMV_STREAM) {
		// pieces of ~12-16 entries per lane; L == 1 keeps the reference's summation order bit for bit
		// measured on MI355X (tools/spmv_sweep.py): ~25 entries per lane is the sweet spot
		if (avg <= 24) return 1;
		if (avg <= 64) return 2;
		if (avg <= 128) return 4;
		if (avg <= 256) return 8;
		if (avg <= 512) return 16;
		return 32;
	}
	int l = 1;
	while (l < 64 && l * 2 <= avg) l *= 2;
	return l;
}

// AUTO considers the PATTERN family for matrices of at least 2^25 stored entries (below that an SpMV is a few tens of microseconds and
// the one-off analysis -- a pass over positions[] -- would not pay for itself within a short solve) whose rows could fit 64 offsets
static bool autoPatternWanted(const smm_hip_csr* m) {
	static const int allowed = [] {
		const char* env = getenv("SMM_HIP_AUTO_PATTERN");
		return env ? atoi(env) : 1;
	}();
	static const long long minNnz = [] {
		const char* env = getenv("SMM_HIP_AUTO_PATTERN_MIN_NNZ");
		return env ? atoll(env) : (1LL << 25);
	}();
	if (!allowed || m->rows <= 0) return false;
	return m->nnz >= minNnz && static_cast<double>(m->nnz) / m->rows <= 64.0;
}

void chooseSpmvConfig(smm_hip_csr* m) {
	const double avg = m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 0.0;
	int family = SMM_SPMV_STREAM;
	if (const char* env = getenv("SMM_HIP_SPMV_FAMILY")) {
		const int f = atoi(env);
		if (f == SMM_SPMV_VECTOR || f == SMM_SPMV_STREAM) family = f;
	}
	if (m->rows == 0 || m->nnz == 0) family = SMM_SPMV_VECTOR;
	int lanes = lanesForAvg(avg, family);
	if (const char* env = getenv("SMM_HIP_SPMV_LANES")) {
		const int l = atoi(env);
		if (l >= 1 && l <= 64 && (l & (l - 1)) == 0) lanes = l;
	}
	m->setKernel(family, lanes);
}

// every translation unit of the library is a code object of its own, built for the device at the FIRST launch of any of its kernels
// (5-9 ms each, measured: profiles/r04/first_spmv_setup_trace.txt); smm_hip_init touches one kernel of each hot-path unit so that
// the first SpMV of a process does not pay for it (SMM_HIP_PRELOAD=0: load lazily as before)
void preloadSpmvUnit() {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(tileCutKernel));
	(void)hipGetLastError();
}

template <typename T, int L>
static void launchVector(const smm_hip_csr* m, int grid, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s) {
	spmvVectorKernel<T, L><<<grid, TPB, 0, s>>>(m->rows, m->d_start, m->d_positions, static_cast<const T*>(m->d_values), op, lhs, divisor, x, out,
	                                            dotMode, w1, partials, doneFlag);
}

template <typename T, int L, int G>
static void launchTileG(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials, const int* doneFlag,
                        hipStream_t s) {
	const int cap = m->stream_nnz_cap + 3;
	const int rt = tileRows(L);
	const size_t lds = static_cast<size_t>(cap + TileCfg<T>::PAD) * (sizeof(T) + 4) + (rt + 4) * sizeof(int) + static_cast<size_t>(L - 1) * rt * sizeof(T) +
	                   4 * sizeof(T) + 32;
	static std::atomic<int> granted{0};      // per instantiation: more than 64 KB of dynamic LDS where a tile needs it, raised on demand
	static std::atomic<long long> occ{0};
	(void)ensureDynamicLds(granted, spmvTileKernel<T, L, G>, lds);  // (refused: the launch below fails and launchSpmv reports it)
	int perCU = occupancyCached(occ, spmvTileKernel<T, L, G>, TPB, lds, 3);
	if (forcedWgsPerCU() > 0) perCU = forcedWgsPerCU();
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();
	const int grid = std::max(1, std::min(std::min(m->n_rowblocks, cus * perCU), NPART));
	const int nGroups = std::min(8, grid);
	const int chunkTiles = m->stream_chunk_tiles > 0 ? m->stream_chunk_tiles : (m->n_rowblocks + nGroups - 1) / nGroups;
	spmvTileKernel<T, L, G><<<grid, TPB, lds, s>>>(m->n_rowblocks, cap, chunkTiles, reinterpret_cast<const int2*>(m->d_rowblocks), m->d_start, m->d_positions,
	                                             static_cast<const T*>(m->d_values), (op & ~SPMV_LEAVE_ROOM) | spmvOutFlags(m, sizeof(T)), lhs, divisor, x, out, dotMode, w1,
	                                             partials, doneFlag);
}

template <typename T, int L>
static void launchTile(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials, const int* doneFlag,
                       hipStream_t s) {
#define SMM_TILE_G(GV) \
	case GV: launchTileG<T, L, GV>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;
	switch (tileBatch(m, L)) {
		SMM_TILE_G(4) SMM_TILE_G(5) SMM_TILE_G(6) SMM_TILE_G(7) SMM_TILE_G(8) SMM_TILE_G(9) SMM_TILE_G(10) SMM_TILE_G(11) SMM_TILE_G(12)
		SMM_TILE_G(13) SMM_TILE_G(14) SMM_TILE_G(15)
	default: launchTileG<T, L, 16>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;
	}
#undef SMM_TILE_G
}

template <typename T, int L>
static void launchStream(const smm_hip_csr* m, int grid, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s) {
	if constexpr (L == 1 || L == 2 || L == 4) {
		if (useTileKernel(L)) {
			launchTile<T, L>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s);
			return;
		}
	}
	const int cap = m->stream_nnz_cap + 3;
	const size_t lds = static_cast<size_t>(cap + StreamCfg<T>::PAD) * (sizeof(T) + 4) + (TPB + 8) * sizeof(int) + 4 * sizeof(T) + 16;
	// persistent grid = exactly the workgroups that are resident together (a larger grid would run in two uneven rounds)
	static std::atomic<long long> occ{0};
	int perCU = occupancyCached(occ, spmvStreamKernel<T, L>, TPB, lds, 4);
	if (forcedWgsPerCU() > 0) perCU = forcedWgsPerCU();
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();
	grid = std::max(1, std::min(std::min(m->n_rowblocks, cus * perCU), NPART));
	const int nGroups = std::min(8, grid);
	const int chunkTiles = m->stream_chunk_tiles > 0 ? m->stream_chunk_tiles : (m->n_rowblocks + nGroups - 1) / nGroups;
	spmvStreamKernel<T, L><<<grid, TPB, lds, s>>>(m->n_rowblocks, cap, chunkTiles, reinterpret_cast<const int2*>(m->d_rowblocks), m->d_start, m->d_positions, static_cast<const T*>(m->d_values),
	                                            (op & ~SPMV_LEAVE_ROOM) | spmvOutFlags(m, sizeof(T)), lhs, divisor, x, out, dotMode, w1, partials, doneFlag);
}

template <typename T>
int launchSpmv(const smm_hip_csr* m, int op, const T* lhs, const T* x, T* out, int dotMode, const T* w1, T* partials, const int* doneFlag,
               hipStream_t s, int extraFlags, const T* divisor) {
	if (m->dtype != dtypeOf<T>()) {
		setError("spmv: matrix dtype does not match the _f32/_f64 entry point");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureCsrReady(m, s, true));
	if (op < SMM_OP_ASSIGN || op > SMM_OP_SUB) {
		setError("spmv: bad op %d", op);
		return SMM_HIP_ERR_INVALID;
	}
	if (m->rows > 0 && (!x || !out || (op != SMM_OP_ASSIGN && !lhs))) {
		setError("spmv: null vector");
		return SMM_HIP_ERR_INVALID;
	}
	if (x == out && m->rows > 0) {  // assert(mult != res), ref:1503
		setError("spmv: x must not alias out");
		return SMM_HIP_ERR_INVALID;
	}
	if (dotMode && (!w1 || !partials)) {
		setError("spmv: fused dot needs w1 and partials");
		return SMM_HIP_ERR_INVALID;
	}
	if ((extraFlags & ~(SPMV_FINISH | SPMV_LEAVE_ROOM | SPMV_DIV_LHS | SPMV_ADD_DIV | SPMV_HALF_TILES)) || ((extraFlags & SPMV_FINISH) && !dotMode) ||
	    ((extraFlags & SPMV_DIV_LHS) && (extraFlags & SPMV_ADD_DIV))) {
		setError("spmv: bad extra flags");
		return SMM_HIP_ERR_INVALID;
	}
	if (extraFlags & SPMV_DIV_LHS) {  // out = (A x) / lhs, row by row
		if (op != SMM_OP_ASSIGN || (m->rows > 0 && !lhs)) {
			setError("spmv: the divide-by-lhs form needs SMM_OP_ASSIGN and a divisor vector");
			return SMM_HIP_ERR_INVALID;
		}
		op = SPMV_OP_DIV;
		divisor = lhs;
		extraFlags &= ~SPMV_DIV_LHS;
	} else if (extraFlags & SPMV_ADD_DIV) {  // out = (lhs + A x) / divisor, row by row
		if (op != SMM_OP_ADD || (m->rows > 0 && (!lhs || !divisor))) {
			setError("spmv: the add-then-divide form needs SMM_OP_ADD, lhs and a divisor vector");
			return SMM_HIP_ERR_INVALID;
		}
		op = SPMV_OP_ADD_DIV;
		extraFlags &= ~SPMV_ADD_DIV;
	} else {
		divisor = nullptr;
	}
	if (m->rows == 0 && !dotMode) return SMM_HIP_OK;
	op |= extraFlags;  // the kernels split `op` into the operation (low byte) and flags
	// AUTO, large matrices: the first SpMV tries the index-free PATTERN family (smm_spmv_pattern.hip) -- analysis and verification of
	// every entry on the caller's stream, once; a matrix that passes is served by it from here on (same bits as STREAM at equal lanes,
	// half the bytes for fp32), one that does not -- or whose analysis could not get its memory -- stays where it is: the attempt is an
	// optimisation and never fails the caller's SpMV.  One thread at a time (adoptMutex); the others wait and then read the word.
	if (!m->kernelForced && m->family() == SMM_SPMV_STREAM && m->pat_state.load(std::memory_order_acquire) == 0 && autoPatternWanted(m)) {
		adoptPatternQuietly(m, s);
	}
	const int kernelWord = m->kernelWord.load(std::memory_order_acquire);  // family and lanes of THIS launch, read once
	const int family = kernelWord & 0xFF;
	const int L = kernelWord >> 8;
	if (family == SMM_SPMV_PATTERN) {
		const int profSlot = profBegin(s);
		const int st = launchSpmvPattern<T>(m, L, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s);  // `op` carries the flags
		profEnd(profSlot, s);
		return st;
	}
	if (family == SMM_SPMV_STREAM) {
		const int capNnz = streamCap<T>(m, L) - 3;
		const int maxRows = useTileKernel(L) ? tileRows(L) : TPB / std::min(L, WAVE);
		std::lock_guard<std::mutex> lock(const_cast<smm_hip_csr*>(m)->tileMutex);
		if (!m->d_rowblocks || m->stream_nnz_cap != capNnz || m->stream_max_rows != maxRows) {
			SMM_TRY(buildRowBlocks(const_cast<smm_hip_csr*>(m), capNnz, maxRows, s));
		}
	}
	int grid;
	if (dotMode) {
		grid = NPART;  // fixed number of partial sums
	} else if (family == SMM_SPMV_STREAM) {
		grid = std::max(1, std::min(m->n_rowblocks, numCUs() * 8));
	} else {
		const long long rowsPerBlock = TPB / L;
		grid = static_cast<int>(std::max<long long>(1, std::min<long long>((m->rows + rowsPerBlock - 1) / rowsPerBlock, numCUs() * 8LL)));
	}
#define SMM_DISPATCH_L(FN)                                                           \
	switch (L) {                                                                     \
	case 1: FN<T, 1>(m, grid, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;   \
	case 2: FN<T, 2>(m, grid, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;   \
	case 4: FN<T, 4>(m, grid, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;   \
	case 8: FN<T, 8>(m, grid, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break;   \
	case 16: FN<T, 16>(m, grid, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break; \
	case 32: FN<T, 32>(m, grid, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break; \
	default: FN<T, 64>(m, grid, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s); break; \
	}
	const int profSlot = profBegin(s);
	if (family == SMM_SPMV_STREAM) {
		SMM_DISPATCH_L(launchStream)
	} else {
		SMM_DISPATCH_L(launchVector)
	}
	profEnd(profSlot, s);
#undef SMM_DISPATCH_L
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template int launchSpmv<float>(const smm_hip_csr*, int, const float*, const float*, float*, int, const float*, float*, const int*, hipStream_t, int, const float*);
template int launchSpmv<double>(const smm_hip_csr*, int, const double*, const double*, double*, int, const double*, double*, const int*, hipStream_t, int, const double*);

// host-pointer entry: copy in, run, copy out (the reference's calling convention, ref:1110-1126)
template <typename T>
static int spmvHost(const smm_hip_csr* m, int op, const T* lhs, const T* x, T* out) {
	if (!m) {
		setError("spmv: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	if (m->rows > 0 && x == out) {
		setError("spmv: x must not alias out");
		return SMM_HIP_ERR_INVALID;
	}
	hipStream_t s = libStream();
	DevBuf<T> dx, dl, dout;
	SMM_TRY(dx.alloc(m->cols));
	SMM_TRY(dout.alloc(m->rows));
	SMM_TRY(hostToDev(dx, x, sizeof(T) * m->cols, s));
	const T* dlhs = nullptr;
	if (op != SMM_OP_ASSIGN) {
		if (!lhs && m->rows) {
			setError("spmv: null lhs");
			return SMM_HIP_ERR_INVALID;
		}
		SMM_TRY(dl.alloc(m->rows));
		SMM_TRY(hostToDev(dl, lhs, sizeof(T) * m->rows, s));
		dlhs = dl;
	}
	SMM_TRY(launchSpmv<T>(m, op, dlhs, dx, dout, 0, nullptr, nullptr, nullptr, s));
	SMM_TRY(devToHost(out, dout, sizeof(T) * m->rows, s));
	return SMM_HIP_OK;
}

}  // namespace smm

using namespace smm;

extern "C" {

int smm_hip_spmv_f32(const smm_hip_csr* m, int op, const float* lhs, const float* x, float* out) { return spmvHost<float>(m, op, lhs, x, out); }
int smm_hip_spmv_f64(const smm_hip_csr* m, int op, const double* lhs, const double* x, double* out) { return spmvHost<double>(m, op, lhs, x, out); }

int smm_hip_spmv_dev_f32(const smm_hip_csr* m, int op, const float* d_lhs, const float* d_x, float* d_out, smm_hip_stream stream) {
	if (!m) {
		setError("spmv: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return launchSpmv<float>(m, op, d_lhs, d_x, d_out, 0, nullptr, nullptr, nullptr, pickStream(stream));
}
int smm_hip_spmv_dev_f64(const smm_hip_csr* m, int op, const double* d_lhs, const double* d_x, double* d_out, smm_hip_stream stream) {
	if (!m) {
		setError("spmv: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return launchSpmv<double>(m, op, d_lhs, d_x, d_out, 0, nullptr, nullptr, nullptr, pickStream(stream));
}

int smm_hip_csr_set_kernel(smm_hip_csr* m, int family, int lanes_per_row) {
	if (!m) {
		setError("csr_set_kernel: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	if (family != SMM_SPMV_AUTO && family != SMM_SPMV_VECTOR && family != SMM_SPMV_STREAM && family != SMM_SPMV_PATTERN) {
		setError("csr_set_kernel: unknown family %d", family);
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	SMM_TRY(ensureCsrReady(m, nullptr, false));
	if (family == SMM_SPMV_PATTERN) {
		// opt-in: analyse + verify every entry now, so a matrix without a shared offset pattern is refused here and not mid-solve
		SMM_TRY(ensurePattern(m));
		if (lanes_per_row > 8) {
			setError("csr_set_kernel: the PATTERN family takes 1, 2, 4 or 8 lanes per row");
			return SMM_HIP_ERR_INVALID;
		}
	}
	if (lanes_per_row != 0 && (lanes_per_row < 1 || lanes_per_row > 64 || (lanes_per_row & (lanes_per_row - 1)))) {
		setError("csr_set_kernel: lanes_per_row must be 0 or a power of two in 1..64");
		return SMM_HIP_ERR_INVALID;
	}
	if (family == SMM_SPMV_AUTO) {
		chooseSpmvConfig(m);
		if (lanes_per_row == 0 && m->pat_state > 0 && autoPatternWanted(m)) {  // already analysed and verified: AUTO's choice stands
			m->setKernel(SMM_SPMV_PATTERN, patternLanesFor(m));
		}
	} else {
		const double avg = m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 0.0;
		m->setKernel(family, lanesForAvg(avg, family));
	}
	if (lanes_per_row) m->setKernel(m->family(), lanes_per_row);
	m->kernelForced = family != SMM_SPMV_AUTO || lanes_per_row != 0;
	return SMM_HIP_OK;
}

int smm_hip_csr_get_kernel(const smm_hip_csr* m, int* family, int* lanes_per_row) {
	if (!m) {
		setError("csr_get_kernel: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	SMM_TRY(ensureCsrReady(m, nullptr, false));
	const int word = m->kernelWord.load(std::memory_order_acquire);
	if (family) *family = word & 0xFF;
	if (lanes_per_row) *lanes_per_row = word >> 8;
	return SMM_HIP_OK;
}

int smm_hip_csr_tile_info(const smm_hip_csr* m, int* tiles, int* tile_nnz_cap, int* tile_max_rows, int* tile_kernel) {
	if (!m) {
		setError("csr_tile_info: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	SMM_TRY(ensureCsrReady(m, nullptr, false));
	std::lock_guard<std::mutex> lock(const_cast<smm_hip_csr*>(m)->tileMutex);
	if (tiles) *tiles = m->d_rowblocks ? m->n_rowblocks : 0;
	if (tile_nnz_cap) *tile_nnz_cap = m->d_rowblocks ? m->stream_nnz_cap : 0;
	if (tile_max_rows) *tile_max_rows = m->d_rowblocks ? m->stream_max_rows : 0;
	if (tile_kernel) *tile_kernel = m->family() == SMM_SPMV_STREAM && useTileKernel(m->lanes()) ? 1 : 0;
	return SMM_HIP_OK;
}

int smm_hip_csr_kernel_desc(const smm_hip_csr* m, char* name, int name_cap, long long* bytes_per_launch) {
	if (!m) {
		setError("csr_kernel_desc: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	SMM_TRY(ensureCsrReady(m, nullptr, false));
	const int word = m->kernelWord.load(std::memory_order_acquire);
	const int family = word & 0xFF, lanes = word >> 8;
	const long long s = m->dtype == SMM_DTYPE_F32 ? 4 : 8;
	long long bytes = static_cast<long long>(m->nnz) * (s + 4) + (static_cast<long long>(m->rows) + 1) * 4 + static_cast<long long>(m->cols) * s +
	                  static_cast<long long>(m->rows) * s;  // B_spmv of the CSR layout
	const char* kernel = "spmvVectorKernel";
	if (family == SMM_SPMV_PATTERN && m->pat_state.load(std::memory_order_acquire) > 0) {
		kernel = patternKernelDesc(m, lanes, &bytes);
	} else if (family == SMM_SPMV_STREAM || family == SMM_SPMV_PATTERN) {
		kernel = (lanes == 1 || lanes == 2 || lanes == 4) && useTileKernel(lanes) ? "spmvTileKernel" : "spmvStreamKernel";
	}
	if (name && name_cap > 0) {
		std::strncpy(name, kernel, static_cast<size_t>(name_cap));
		name[name_cap - 1] = 0;
	}
	if (bytes_per_launch) *bytes_per_launch = bytes;
	return SMM_HIP_OK;
}

int smm_hip_csr_autotune(smm_hip_csr* m) {
	if (!m) {
		setError("csr_autotune: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	SMM_TRY(ensureCsrReady(m, nullptr, false));
	if (m->rows == 0 || m->nnz == 0) return SMM_HIP_OK;
	hipStream_t s = libStream();
	SMM_HIP_TRY(hipDeviceSynchronize());  // the matrix may still be being written on a caller's stream (set-up call, no stream)
	const size_t esz = m->dtype == SMM_DTYPE_F32 ? 4 : 8;
	void *dx = nullptr, *dy = nullptr;
	SMM_TRY(devAlloc(&dx, esz * static_cast<size_t>(m->cols ? m->cols : 1)));
	SMM_TRY(devAlloc(&dy, esz * static_cast<size_t>(m->rows)));
	SMM_HIP_TRY(hipMemsetAsync(dx, 0, esz * static_cast<size_t>(m->cols ? m->cols : 1), s));
	hipEvent_t e0, e1;
	SMM_HIP_TRY(hipEventCreate(&e0));
	SMM_HIP_TRY(hipEventCreate(&e1));
	const double avg = static_cast<double>(m->nnz) / m->rows;
	float best = 1e30f;
	int bestFamily = m->family(), bestLanes = m->lanes();
	int status = SMM_HIP_OK;
	for (int family = SMM_SPMV_VECTOR; family <= SMM_SPMV_STREAM && status == SMM_HIP_OK; ++family) {
		const int center = lanesForAvg(avg, family);
		for (int lanes = std::max(1, center / 2); lanes <= std::min(64, center * 2) && status == SMM_HIP_OK; lanes *= 2) {
			m->setKernel(family, lanes);
			for (int rep = 0; rep < 3 && status == SMM_HIP_OK; ++rep) {
				hipEventRecord(e0, s);
				if (m->dtype == SMM_DTYPE_F32) {
					status = launchSpmv<float>(m, SMM_OP_ASSIGN, nullptr, static_cast<float*>(dx), static_cast<float*>(dy), 0, nullptr, nullptr, nullptr, s);
				} else {
					status = launchSpmv<double>(m, SMM_OP_ASSIGN, nullptr, static_cast<double*>(dx), static_cast<double*>(dy), 0, nullptr, nullptr, nullptr, s);
				}
				hipEventRecord(e1, s);
				hipEventSynchronize(e1);
				float ms = 0;
				hipEventElapsedTime(&ms, e0, e1);
				if (rep > 0 && ms < best) {
					best = ms;
					bestFamily = family;
					bestLanes = lanes;
				}
			}
		}
	}
	m->setKernel(bestFamily, bestLanes);
	m->kernelForced = true;
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	devFree(dx);
	devFree(dy);
	return status;
}

}  // extern "C"
