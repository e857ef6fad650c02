// smm_spmv_march.hip -- the CONST encoding (row masks + one value per diagonal, smm_spmv_pattern.hip) for GRID-shaped matrices: the
// "2.5-D" form of rMult (ref:1458-1499) for stencils whose offsets are a few NEAR ones (|off| <= H: the neighbours inside a grid plane)
// plus at most the pair -P / +P (the neighbours in the planes below and above; P = rows per plane).
//
// Why (profiles/r04/const_kernel_ablation.txt): spmvPatternConstKernel issues one global gather per entry -- 7 per row of the 3-D
// Laplacian -- and is bound by exactly that: with EVERY memory stream but one read of x removed it still takes 0.59 of its 1.12 ms on
// the 512^3 grid, pointing the two far gathers at x[row] changes nothing, pointing all seven there saves a third.  What costs is the
// number of vector-memory instructions and first-level-cache transactions per row (9 per 64 rows), not the bytes behind them.
//
// Here every element of x crosses the memory path ONCE per (tile, plane) and is then served from LDS or a register:
//   * a workgroup owns a run of B = 2048 consecutive rows of ONE plane (a "tile") and marches along the far direction: rows
//     tile + z P for z = z0 .. z1.  Per plane it holds the WINDOW x[base - H, base + B + H) in LDS (base = z P + tile start): the centre
//     part comes from its own registers, the two halos of H elements are loaded beside it;
//   * near entries x[row + off] are LDS reads of the window; the far entries x[row -+ P] are the centre values of the planes before
//     and after, which the lane keeps in registers (loaded two planes ahead, with 16-byte loads, 8 rows per lane);
//   * masks, out[] (and lhs / w1 where an epilogue needs them) move as 16-byte packs as well.
// Vector-memory instructions per 64 rows: 1.75 (x 0.5 + halo 0.25 + masks 0.5 + out 0.5) instead of 9; x is fetched 1 + 2 H / B times
// instead of once per entry.
//
// Same products in the same order as every other kernel of the library with one lane per row: a row's entries are visited in ascending
// column order (-P, the near offsets ascending, +P -- the bit order of the row's mask), each product is c_j * x[col] with c_j the bit
// pattern verified against every values[k] of that diagonal, folded left to right with smmFma: the reference's bits (ref:1484-1489).
#include <algorithm>
#include <atomic>

#include "smm_device.h"
#include "smm_internal.h"
#include "smm_solver_scal.h"

namespace smm {

namespace {

constexpr int TPB = 256;
constexpr int MARCH_RMAX = 8;  // rows per lane and plane: 8 or 4 (template argument R; a tile is B = TPB * R rows)

template <typename T, int R>
struct MarchCfg {
	static constexpr int VEC = 16 / sizeof(T);  // rows per 16-byte pack
	static constexpr int PACKS = R / VEC;       // packs per lane and plane
	static constexpr int B = TPB * R;           // rows of a tile
};

// 16-byte packs of x / out at ELEMENT alignment (a plane need not start on a 16-byte boundary; gfx950 global accesses may be
// unaligned): vector types, so that the non-temporal builtins take them, with the alignment lowered through the typedef
template <typename T>
struct PackOf;
template <>
struct PackOf<float> {
	typedef float V __attribute__((ext_vector_type(4)));
	typedef V U __attribute__((aligned(4)));
	typedef unsigned M __attribute__((ext_vector_type(4)));  // the 32-bit masks of the pack's 4 rows
};
template <>
struct PackOf<double> {
	typedef double V __attribute__((ext_vector_type(2)));
	typedef V U __attribute__((aligned(8)));
	typedef unsigned M __attribute__((ext_vector_type(2)));
};
template <typename T>
using PackU = typename PackOf<T>::U;
template <typename T>
using PackV = typename PackOf<T>::V;  // the same pack where 16-byte alignment is known (LDS windows)
template <typename T>
using MaskP = typename PackOf<T>::M;
template <typename T>
struct MaskBytesOf;  // the integer that holds the byte masks of one pack's rows
template <>
struct MaskBytesOf<float> {
	using type = unsigned;
};
template <>
struct MaskBytesOf<double> {
	using type = unsigned short;
};

template <typename T>
__device__ __forceinline__ T marchApplyOp(int op, const T* __restrict__ lhs, const T* __restrict__ divisor, long long row, T dot) {
	if (op == SMM_OP_ASSIGN) return dot;
	if (op == SPMV_OP_DIV) return dot / divisor[row];
	const T l = lhs[row];
	if (op == SPMV_OP_ADD_DIV) return (l + dot) / divisor[row];
	return op == SMM_OP_ADD ? l + dot : l - dot;
}

template <typename T>
__device__ __forceinline__ T bitsToValue(unsigned long long bits) {
	T c;
	if (sizeof(T) == 4) {
		const unsigned lo = static_cast<unsigned>(bits);
		__builtin_memcpy(&c, &lo, 4);
	} else {
		__builtin_memcpy(&c, &bits, sizeof(T));
	}
	return c;
}

// what a lane requests for one plane of its tile: its 8 centre values of x, its share of the two halos, the masks of its 8 rows
template <typename T, int R, int HP, bool FUSE = false>
struct MarchSet {
	PackU<T> c[MarchCfg<T, R>::PACKS];
	PackU<T> h[HP];
	MaskP<T> m[MarchCfg<T, R>::PACKS];
	// FUSE (MarchFuse below): the same packs of the second stream, r -- until the set is first used, when c / h become beta * c + r
	PackU<T> c2[FUSE ? MarchCfg<T, R>::PACKS : 1];
	PackU<T> h2[FUSE ? HP : 1];
};

}  // namespace

// ConjugateGradient's next direction formed in the SpMV's own load phase (r05): the launch reads the PREVIOUS direction
// where it would read x, and r beside it, and every element it touches -- centre, halo, the planes above and below -- becomes
// p = beta p_old + r (ref:2391-2393) the moment its request set is first used: the owner's expression on the owner's operands, the same bits
// in every workgroup that needs the element.  The rows a unit computes are written to pNew (each exactly once); beta comes from the partials
// of ||r||^2 the update before left behind, added by every workgroup in the fixed order (sumPartsAll); workgroup 0 does the iteration's
// bookkeeping (ref:2377-2382) that cgFusedXP / cgLazyXP would have done, and a launch that finds the iteration converged only records
// that (flushIter: the flush launch behind it completes x) and returns.  p.Ap takes p from the LDS window instead of a second read.
template <typename T>
struct MarchFuse {
	const T* r = nullptr;
	T* pNew = nullptr;
	CgFuseBook<T> bk{};
	const T* partsC = nullptr;   // ||r||^2 as NPART partial sums (one GPU) ...
	const T* totalsC = nullptr;  // ... or as the all-reduced total (the row-partitioned loop, smm_dist.hip)
	T eps = T(0);
	int par = 0;
	int iter = 0;
};

extern __shared__ __attribute__((aligned(16))) unsigned char smmMarchLds[];

// the 32-bit copy of the row masks the kernel streams (CONST: at most 32 offsets, so the upper halves of the 64-bit masks are empty)
__global__ __launch_bounds__(256) void marchNarrowMasks(long long rows, const unsigned long long* __restrict__ masks, unsigned* __restrict__ masks32) {
	for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < rows; i += static_cast<long long>(gridDim.x) * blockDim.x) {
		masks32[i] = static_cast<unsigned>(masks[i]);
	}
}

// the 8-bit copy for matrices of at most 8 offsets (the 5- and 7-point stencils: the KN = 5 instances below): at 512^3 the 32-bit masks are
// 0.54 GB of the 2.68 GB an fp64 launch moves -- a fifth -- and a third of an fp32 launch's bytes; one byte per row makes them 0.13 GB
__global__ __launch_bounds__(256) void marchNarrowMasks8(long long rows, const unsigned long long* __restrict__ masks, unsigned char* __restrict__ masks8) {
	for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < rows; i += static_cast<long long>(gridDim.x) * blockDim.x) {
		masks8[i] = static_cast<unsigned char>(masks[i]);
	}
}

// KN > 0: the number of near offsets is known at compile time (5: the 5- and 7-point stencils); 0: run time.  NT: out[] with non-temporal
// stores.  HP: halo packs a lane can hold (2 H <= HP * TPB * VEC).
// grid <= NPART persistent workgroups; unit u = (tile, z-chunk); XCD group g = blockIdx % 8 owns the tiles [g nT / 8, (g + 1) nT / 8) when
// xcdTiles (neighbouring tiles, which share their halos, then run on one L2), else the units are dealt round-robin.
//
// Pipeline of a unit: TWO request sets alternate.  In step z (plane z's window is in LDS) the set that held plane z is re-issued for
// plane z + 2 -- centre values, halos, masks -- while the other one, requested a step earlier, supplies plane z + 1: its centre values
// are this step's +P operands and, with its halos, the next window.  No register that a load is still in flight for is ever copied
// (the z loop is unrolled by two instead), so nothing waits for a request issued in the same step.
// Registers: fp64 with 8 rows per lane needs ~185 VGPRs (two workgroups per CU; capped at 168 for three it spills and loses, 0.60 ->
// 0.65 ms on the 512^3 grid), fp32 fits three either way and schedules better when told so (0.354 -> 0.322 ms): the bound below.
#ifndef SMM_MARCH_MIN_WAVES
#define SMM_MARCH_MIN_WAVES(T, R) ((sizeof(T) == 4 || (R) < 8) ? 3 : 1)
#endif
template <typename T, int R, int KN, bool NT, int HP, bool FUSE = false>
// (FUSE: 197-241 VGPRs in fp64, two workgroups per CU; capped at 168 for three it spills 120-400 bytes per lane and loses: 1.82 -> 2.08 ms per
// CG iteration at 512^3 fp64, 0.97 -> 1.41 in fp32)
#ifndef SMM_MARCH_FUSE_MIN_WAVES
#define SMM_MARCH_FUSE_MIN_WAVES 2
#endif
__global__ __launch_bounds__(TPB, FUSE ? SMM_MARCH_FUSE_MIN_WAVES : SMM_MARCH_MIN_WAVES(T, R)) void spmvPatternConstMarchKernel(int rows, int cols, int P, int nPlanes, int nT, int zc, int nChunks, int xcdTiles, int H,
                                                                   int nOff, int hasLo, int hasHi, const int* __restrict__ offs,
                                                                   const unsigned long long* __restrict__ cvalBits, const void* __restrict__ masksN,
                                                                   int opFlags, const T* lhs, const T* __restrict__ divisor, const T* __restrict__ x, T* out,
                                                                   int dotMode, const T* __restrict__ w1, T* __restrict__ partials,
                                                                   const int* __restrict__ doneFlag, MarchFuse<T> fz = MarchFuse<T>()) {
	// the row masks: one BYTE per row when the offset count is a compile-time one (KN = 5: at most 7 offsets), 32 bits otherwise
	const unsigned* __restrict__ masks32 = static_cast<const unsigned*>(masksN);
	const unsigned char* __restrict__ masks8 = static_cast<const unsigned char*>(masksN);
	using Cfg = MarchCfg<T, R>;
	using Set = MarchSet<T, R, HP, FUSE>;
	constexpr int VEC = Cfg::VEC;
	constexpr int PACKS = Cfg::PACKS;
	constexpr int MARCH_B = Cfg::B;
	const int winLen = MARCH_B + 2 * H;  // elements of one window buffer (H is a multiple of VEC)
	T* sWin0 = reinterpret_cast<T*>(smmMarchLds);
	T* sWin1 = sWin0 + winLen;
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	const int op = opFlags & 0xFF;
	const int t = threadIdx.x;
	T beta = T(0);
	if constexpr (FUSE) {
		__shared__ T redF[5];
		if (*fz.bk.pad) return;  // (the done flag as the update before found it)
		const T rrNew = fz.totalsC ? fz.totalsC[0] : sumPartsAll(fz.partsC, redF);
		const T rrOld = fz.bk.rrPing[fz.par];
		const bool converged = fz.eps * fz.eps > rrNew;
		if (blockIdx.x == 0 && t == 0) {  // ref:2377-2382
			*fz.bk.iters += 1;
			*fz.bk.res = rrNew;
			if (converged) {
				*fz.bk.done = 1;
				*fz.bk.status = SMM_SOLVER_SUCCESS;
				*fz.bk.flushIter = fz.iter;
			} else {
				fz.bk.rrPing[fz.par ^ 1] = rrNew;
			}
		}
		if (converged) return;
		beta = rrNew / rrOld;
	}
	const int nNear = KN > 0 ? KN : nOff - hasLo - hasHi;
	const T cLo = hasLo ? bitsToValue<T>(cvalBits[0]) : T(0);
	const T cHi = hasHi ? bitsToValue<T>(cvalBits[nOff - 1]) : T(0);
	const int haloPacks = 2 * H / VEC;
	const unsigned fullMask = nOff >= 32 ? 0xFFFFFFFFu : ((1u << nOff) - 1u);

	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int units = nT * nChunks;
	int uFirst, uStride, uEnd, tLo = 0, tCount = nT;
	if (xcdTiles) {
		const int g = blockIdx.x % nGroups;
		tLo = static_cast<int>(static_cast<long long>(g) * nT / nGroups);
		tCount = static_cast<int>(static_cast<long long>(g + 1) * nT / nGroups) - tLo;
		uFirst = blockIdx.x / nGroups;
		uStride = (static_cast<int>(gridDim.x) - g + nGroups - 1) / nGroups;
		uEnd = tCount * nChunks;
	} else {
		uFirst = blockIdx.x;
		uStride = gridDim.x;
		uEnd = units;
	}
	T acc0 = T(0), acc1 = T(0);
	int loc[PACKS];  // first local row of the lane's pack p
#pragma unroll
	for (int p = 0; p < PACKS; ++p) loc[p] = (p * TPB + t) * VEC;

	for (int u = uFirst; u < uEnd; u += uStride) {
		// chunk-major inside the group: neighbouring tiles of one z-chunk are in flight together
		const int chunk = u / tCount;
		const int tile = tLo + (u - chunk * tCount);
		const int z0 = chunk * zc;
		const int z1 = min(nPlanes, z0 + zc);
		const int r0 = tile * MARCH_B;           // first row of the tile inside its plane
		const int bAct = min(MARCH_B, P - r0);   // rows of this tile (P and MARCH_B are multiples of VEC)
		bool act[PACKS];
#pragma unroll
		for (int p = 0; p < PACKS; ++p) act[p] = loc[p] < bAct;

		// halo pack i of a plane: window index i VEC for the left halo (i < H / VEC), H + bAct + (i - H / VEC) VEC for the right one
		auto haloWin = [&](int i) { return i < H / VEC ? i * VEC : H + bAct + (i - H / VEC) * VEC; };
		// requests for plane z: centre (when wantCentre), halos and masks (when wantWindow: the plane will be a window of this unit)
		auto issue = [&](Set& f, int z, bool wantCentre, bool wantWindow) {
			const bool inside = z >= 0 && z < nPlanes;
			const long long base = static_cast<long long>(z) * P + r0;
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				if (inside && wantCentre && act[p] && base + loc[p] < rows) {  // (the last plane may be a partial one)
					f.c[p] = *reinterpret_cast<const PackU<T>*>(x + base + loc[p]);  // (kept cacheable: a neighbouring tile reads these lines as its halo)
					if constexpr (FUSE) f.c2[p] = *reinterpret_cast<const PackU<T>*>(fz.r + base + loc[p]);
				} else {
#pragma unroll
					for (int e = 0; e < VEC; ++e) {
						f.c[p][e] = T(0);
						if constexpr (FUSE) f.c2[p][e] = T(0);
					}
				}
			}
			if (inside && wantWindow) {
#pragma unroll
				for (int k = 0; k < HP; ++k) {
					const int i = k * TPB + t;
					if (i < haloPacks) {
						long long gidx = base - H + haloWin(i);
						// a pack that sticks out of x (the first / last rows of the matrix) is never used by a live entry: any valid address will do
						gidx = gidx < 0 ? 0 : (gidx + VEC > cols ? cols - VEC : gidx);
						f.h[k] = *reinterpret_cast<const PackU<T>*>(x + gidx);
						if constexpr (FUSE) f.h2[k] = *reinterpret_cast<const PackU<T>*>(fz.r + gidx);
					}
				}
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
					if (act[p] && base + loc[p] < rows) {
						if constexpr (KN > 0) {
							// the pack's VEC bytes in one load (base + loc[p] is a multiple of VEC: P and the tile size are)
							using MB = typename MaskBytesOf<T>::type;
							// (kept as loaded -- element 0 -- until the set is used: unpacking here would wait for the load in the step that issues it)
							f.m[p][0] = __builtin_nontemporal_load(reinterpret_cast<const MB*>(masks8 + base + loc[p]));
						} else {
							f.m[p] = __builtin_nontemporal_load(reinterpret_cast<const MaskP<T>*>(masks32 + base + loc[p]));
						}
					} else {
#pragma unroll
						for (int e = 0; e < VEC; ++e) f.m[p][e] = 0u;
					}
				}
			}
		};
		auto storeWindow = [&](T* win, const Set& f) {
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				if (act[p]) *reinterpret_cast<PackV<T>*>(win + H + loc[p]) = f.c[p];
			}
#pragma unroll
			for (int k = 0; k < HP; ++k) {
				const int i = k * TPB + t;
				if (i < haloPacks) *reinterpret_cast<PackV<T>*>(win + haloWin(i)) = f.h[k];
			}
		};

		// FUSE: the set's packs become the new direction, beta * p_old + r (ref:2391-2393); the rows of plane z this unit computes are written out
		auto combine = [&](Set& f, int z, bool write) {
			if constexpr (FUSE) {
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
#pragma unroll
					for (int e = 0; e < VEC; ++e) f.c[p][e] = smmFma(beta, f.c[p][e], f.c2[p][e]);
				}
#pragma unroll
				for (int k = 0; k < HP; ++k) {
#pragma unroll
					for (int e = 0; e < VEC; ++e) f.h[k][e] = smmFma(beta, f.h[k][e], f.h2[k][e]);
				}
				if (write) {
					const long long base = static_cast<long long>(z) * P + r0;
#pragma unroll
					for (int p = 0; p < PACKS; ++p) {
						if (act[p] && base + loc[p] < rows) {
							if constexpr (NT) __builtin_nontemporal_store(f.c[p], reinterpret_cast<PackU<T>*>(fz.pNew + base + loc[p]));
							else *reinterpret_cast<PackU<T>*>(fz.pNew + base + loc[p]) = f.c[p];
						}
					}
				}
			}
		};

		PackU<T> xp[PACKS];      // centre of the plane below the window's (the -P operands)
		MaskP<T> mk[PACKS];      // masks of the window's plane
		Set fa, fb;
		auto masksOf = [&](const Set& f, int p) {
			if constexpr (KN > 0) {
				MaskP<T> m;
#pragma unroll
				for (int e = 0; e < VEC; ++e) m[e] = (f.m[p][0] >> (8 * e)) & 0xFFu;
				return m;
			} else {
				return f.m[p];
			}
		};

		// one plane: `use` holds plane z + 1 (requested a step ago), `re` is free and is re-issued for plane z + 2
		auto step = [&](int z, Set& use, Set& re) {
			T* win = ((z - z0) & 1) ? sWin1 : sWin0;
			T* winNext = ((z - z0) & 1) ? sWin0 : sWin1;
			const bool more = z + 1 < z1;
			if (more) issue(re, z + 2, z + 2 < z1 || hasHi, z + 2 < z1);
			const long long base = static_cast<long long>(z) * P + r0;
			T dot[PACKS][VEC];
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
#pragma unroll
				for (int e = 0; e < VEC; ++e) dot[p][e] = T(0);
			}
			// interior rows hold every offset: when that is true for all rows of the wavefront the per-entry selects are skipped
			bool full = true;
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
#pragma unroll
				for (int e = 0; e < VEC; ++e) full = full && (mk[p][e] == fullMask || !act[p]);
			}
			const bool masked = !__all(full);
			auto fold = [&](T c, T xv, T d, unsigned m, int b) {
				const T next = smmFma(c, xv, d);
				return (!masked || ((m >> b) & 1u) != 0u) ? next : d;
			};
			if (hasLo) {
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
#pragma unroll
					for (int e = 0; e < VEC; ++e) dot[p][e] = fold(cLo, xp[p][e], dot[p][e], mk[p][e], 0);
				}
			}
			auto nearStep = [&](int j) {
				const int off = offs[hasLo + j];
				const T c = bitsToValue<T>(cvalBits[hasLo + j]);
				const int b = hasLo + j;
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
					T xv[VEC];
#pragma unroll
					for (int e = 0; e < VEC; ++e) xv[e] = win[H + loc[p] + e + off];
#pragma unroll
					for (int e = 0; e < VEC; ++e) dot[p][e] = fold(c, xv[e], dot[p][e], mk[p][e], b);
				}
			};
			if constexpr (KN > 0) {
#pragma unroll
				for (int j = 0; j < KN; ++j) nearStep(j);
			} else {
				for (int j = 0; j < nNear; ++j) nearStep(j);
			}
			combine(use, z + 1, more);  // (FUSE: plane z + 1's set is first used here; its rows are this unit's when another step follows)
			if (hasHi) {
				const int b = hasLo + nNear;
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
#pragma unroll
					for (int e = 0; e < VEC; ++e) dot[p][e] = fold(cHi, use.c[p][e], dot[p][e], mk[p][e], b);
				}
			}
			// epilogue: op(lhs, dot), out[] as 16-byte packs, the fused dot products
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				if (act[p] && base + loc[p] < rows) {
					const long long row = base + loc[p];
					PackU<T> o;
#pragma unroll
					for (int e = 0; e < VEC; ++e) o[e] = marchApplyOp(op, lhs, divisor, row + e, dot[p][e]);
					// (a run-time `if (ntOut) nt-store else store` does not survive the compiler: the hint is metadata and the two arms are merged
					// into ONE plain store -- which is what happens in the older kernels; here the policy is a template argument)
					if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<PackU<T>*>(out + row));
					else *reinterpret_cast<PackU<T>*>(out + row) = o;
					if (dotMode) {
						PackU<T> w;
						if constexpr (FUSE) w = *reinterpret_cast<const PackV<T>*>(win + H + loc[p]);  // (p itself: the window's centre)
						else w = *reinterpret_cast<const PackU<T>*>(w1 + row);
#pragma unroll
						for (int e = 0; e < VEC; ++e) {
							if (dotMode == 2) acc0 += o[e] * o[e];
							acc1 += o[e] * w[e];
						}
					}
				}
			}
			if (more) {
				// the window's own centre becomes the plane below; the next window and its masks come out of `use`
#pragma unroll
				for (int p = 0; p < PACKS; ++p) xp[p] = *reinterpret_cast<const PackV<T>*>(win + H + loc[p]);
				storeWindow(winNext, use);
#pragma unroll
				for (int p = 0; p < PACKS; ++p) mk[p] = masksOf(use, p);
				ldsBarrier();  // winNext is complete; everyone is done reading `win` (it is overwritten in the step after next)
			}
		};

		// prologue: plane z0 - 1 (far below) and plane z0 itself, then the requests for plane z0 + 1
		issue(fa, z0 - 1, hasLo != 0, false);
		if constexpr (FUSE) {
#pragma unroll
			for (int k = 0; k < HP; ++k) {
#pragma unroll
				for (int e = 0; e < VEC; ++e) fa.h[k][e] = fa.h2[k][e] = T(0);  // (no halos were requested for the plane far below)
			}
		}
		combine(fa, z0 - 1, false);
#pragma unroll
		for (int p = 0; p < PACKS; ++p) xp[p] = fa.c[p];
		issue(fa, z0, true, true);
		issue(fb, z0 + 1, z0 + 1 < z1 || hasHi, z0 + 1 < z1);
		combine(fa, z0, true);
		__syncthreads();  // (the previous unit's last window reads are over)
		storeWindow(sWin0, fa);
#pragma unroll
		for (int p = 0; p < PACKS; ++p) mk[p] = masksOf(fa, p);
		__syncthreads();
		for (int z = z0; z < z1; z += 2) {
			step(z, fb, fa);
			if (z + 1 < z1) step(z + 1, fa, fb);
		}
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
// The march for stencils whose FAR offsets come in CLUSTERS (r05): 19- and 27-point stencils -- HPCG's matrix -- touch
// the planes below and above at -P + d and +P + d for several near d, not only at -P / +P, so the lane's own centre values of the
// neighbouring planes (registers, above) are not enough: here the planes z - 1, z and z + 1 are ALL windows in LDS.  Four window buffers
// rotate (plane z + 2 is stored while z - 1, z, z + 1 are read; the buffer it overwrites held z - 2, which nobody reads after the barrier
// that ended step z - 1): still ONE workgroup barrier per plane.  Request sets alternate as above, shifted by one plane (`use` holds plane
// z + 2).  Offsets are visited in ascending order -- the cluster around -P, the near ones, the cluster around +P: ascending columns, the
// row mask's bit order, the reference's products in the reference's order (ref:1484-1489).
// LDS: 4 (B + 2 H) elements per workgroup; a 27-point row costs 27 LDS reads instead of 27 global gathers (spmvPatternConstKernel).
template <typename T, int R, bool NT, int HP>
__global__ __launch_bounds__(TPB, SMM_MARCH_MIN_WAVES(T, R)) void spmvPatternConstMarch3Kernel(int rows, int cols, int P, int nPlanes, int nT, int zc, int nChunks, int xcdTiles,
                                                                    int H, int nOff, int nLo, int nHi, const int* __restrict__ offs,
                                                                    const unsigned long long* __restrict__ cvalBits, const unsigned* __restrict__ masks32,
                                                                    int opFlags, const T* lhs, const T* __restrict__ divisor, const T* __restrict__ x, T* out,
                                                                    int dotMode, const T* __restrict__ w1, T* __restrict__ partials,
                                                                    const int* __restrict__ doneFlag) {
	using Cfg = MarchCfg<T, R>;
	using Set = MarchSet<T, R, HP>;
	constexpr int VEC = Cfg::VEC;
	constexpr int PACKS = Cfg::PACKS;
	constexpr int MARCH_B = Cfg::B;
	const int winLen = MARCH_B + 2 * H;
	T* const sWin = reinterpret_cast<T*>(smmMarchLds);  // four windows of winLen elements
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	const int op = opFlags & 0xFF;
	const int t = threadIdx.x;
	const int nMid = nOff - nLo - nHi;
	const int haloPacks = 2 * H / VEC;
	const unsigned fullMask = nOff >= 32 ? 0xFFFFFFFFu : ((1u << nOff) - 1u);

	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int units = nT * nChunks;
	int uFirst, uStride, uEnd, tLo = 0, tCount = nT;
	if (xcdTiles) {
		const int g = blockIdx.x % nGroups;
		tLo = static_cast<int>(static_cast<long long>(g) * nT / nGroups);
		tCount = static_cast<int>(static_cast<long long>(g + 1) * nT / nGroups) - tLo;
		uFirst = blockIdx.x / nGroups;
		uStride = (static_cast<int>(gridDim.x) - g + nGroups - 1) / nGroups;
		uEnd = tCount * nChunks;
	} else {
		uFirst = blockIdx.x;
		uStride = gridDim.x;
		uEnd = units;
	}
	T acc0 = T(0), acc1 = T(0);
	int loc[PACKS];
#pragma unroll
	for (int p = 0; p < PACKS; ++p) loc[p] = (p * TPB + t) * VEC;

	for (int u = uFirst; u < uEnd; u += uStride) {
		const int chunk = u / tCount;
		const int tile = tLo + (u - chunk * tCount);
		const int z0 = chunk * zc;
		const int z1 = min(nPlanes, z0 + zc);
		const int r0 = tile * MARCH_B;
		const int bAct = min(MARCH_B, P - r0);
		bool act[PACKS];
#pragma unroll
		for (int p = 0; p < PACKS; ++p) act[p] = loc[p] < bAct;
		auto haloWin = [&](int i) { return i < H / VEC ? i * VEC : H + bAct + (i - H / VEC) * VEC; };
		auto winOf = [&](int z) { return sWin + static_cast<size_t>((z - z0 + 1) & 3) * winLen; };
		// the whole window of plane z (centre + halos); masks only for the planes this unit computes.  What is loaded is decided element by
		// element, not plane by plane: the window of the plane "below the first" still holds the first plane's leading elements in its right halo
		// (an entry at -P + d, d > 0, of a row near the end of plane 0 points there), and likewise above the last plane
		auto issue = [&](Set& f, int z, bool wantMasks) {
			const long long base = static_cast<long long>(z) * P + r0;
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				const long long gi = base + loc[p];
				if (act[p] && gi >= 0 && gi < rows) {  // (P, r0, rows are multiples of the pack: a pack is inside or outside as a whole)
					f.c[p] = *reinterpret_cast<const PackU<T>*>(x + gi);
				} else {
#pragma unroll
					for (int e = 0; e < VEC; ++e) f.c[p][e] = T(0);
				}
			}
#pragma unroll
			for (int k = 0; k < HP; ++k) {
				const int i = k * TPB + t;
				if (i < haloPacks) {
					const long long gidx = base - H + haloWin(i);
					if (gidx >= 0 && gidx + VEC <= cols) {
						f.h[k] = *reinterpret_cast<const PackU<T>*>(x + gidx);
					} else {
#pragma unroll
						for (int e = 0; e < VEC; ++e) f.h[k][e] = T(0);
					}
				}
			}
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				const long long gi = base + loc[p];
				if (wantMasks && act[p] && gi >= 0 && gi < rows) {
					f.m[p] = __builtin_nontemporal_load(reinterpret_cast<const MaskP<T>*>(masks32 + gi));
				} else {
#pragma unroll
					for (int e = 0; e < VEC; ++e) f.m[p][e] = 0u;
				}
			}
		};
		auto storeWindow = [&](T* win, const Set& f) {
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				if (act[p]) *reinterpret_cast<PackV<T>*>(win + H + loc[p]) = f.c[p];
			}
#pragma unroll
			for (int k = 0; k < HP; ++k) {
				const int i = k * TPB + t;
				if (i < haloPacks) *reinterpret_cast<PackV<T>*>(win + haloWin(i)) = f.h[k];
			}
		};

		MaskP<T> mk[PACKS], mk1[PACKS];  // masks of plane z and of plane z + 1
		Set fa, fb;

		// one plane: windows z - 1, z, z + 1 are in LDS; `use` holds plane z + 2 (requested a step ago), `re` is re-issued for plane z + 3
		auto step = [&](int z, Set& use, Set& re) {
			const bool more = z + 1 < z1;
			if (more) issue(re, z + 3, z + 3 < z1);
			const long long base = static_cast<long long>(z) * P + r0;
			T dot[PACKS][VEC];
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
#pragma unroll
				for (int e = 0; e < VEC; ++e) dot[p][e] = T(0);
			}
			bool full = true;
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
#pragma unroll
				for (int e = 0; e < VEC; ++e) full = full && (mk[p][e] == fullMask || !act[p]);
			}
			const bool masked = !__all(full);
			auto entry = [&](const T* win, int j, int centre) {
				const int off = offs[j] - centre;  // (wave-uniform: scalar loads; LDS copies of the offsets and values measured slower -- this kernel
				const T c = bitsToValue<T>(cvalBits[j]);  // lives on its LDS reads --, and so did cluster sizes as template arguments)
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
					T xv[VEC];
#pragma unroll
					for (int e = 0; e < VEC; ++e) xv[e] = win[H + loc[p] + e + off];
#pragma unroll
					for (int e = 0; e < VEC; ++e) {
						const T next = smmFma(c, xv[e], dot[p][e]);
						dot[p][e] = (!masked || ((mk[p][e] >> j) & 1u) != 0u) ? next : dot[p][e];
					}
				}
			};
			// three offsets in a row (dx = -1, 0, 1 of a stencil line) share their LDS reads: VEC + 2 elements serve the 3 x VEC products of a
			// pack (a third fewer reads in fp64, half in fp32: this kernel lives on them); any other offset goes alone.  Same products, same order.
			auto line3 = [&](const T* win, int j, int centre) {
				const int off = offs[j] - centre;
				const T c0 = bitsToValue<T>(cvalBits[j]), c1 = bitsToValue<T>(cvalBits[j + 1]), c2 = bitsToValue<T>(cvalBits[j + 2]);
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
					T xv[VEC + 2];
#pragma unroll
					for (int e = 0; e < VEC + 2; ++e) xv[e] = win[H + loc[p] + e + off];
#pragma unroll
					for (int e = 0; e < VEC; ++e) {
						const unsigned mm = mk[p][e] >> j;
						T d = dot[p][e];
						T next = smmFma(c0, xv[e], d);
						d = (!masked || (mm & 1u) != 0u) ? next : d;
						next = smmFma(c1, xv[e + 1], d);
						d = (!masked || (mm & 2u) != 0u) ? next : d;
						next = smmFma(c2, xv[e + 2], d);
						d = (!masked || (mm & 4u) != 0u) ? next : d;
						dot[p][e] = d;
					}
				}
			};
			auto clusterRun = [&](const T* win, int jFirst, int jCount, int centre) {
				int jj = 0;
				while (jj < jCount) {
					const int j = jFirst + jj;
					if (jj + 2 < jCount && offs[j + 1] == offs[j] + 1 && offs[j + 2] == offs[j] + 2) {
						line3(win, j, centre);
						jj += 3;
					} else {
						entry(win, j, centre);
						jj += 1;
					}
				}
			};
			if (nLo) clusterRun(winOf(z - 1), 0, nLo, -P);
			clusterRun(winOf(z), nLo, nMid, 0);
			if (nHi) clusterRun(winOf(z + 1), nLo + nMid, nHi, P);
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				if (act[p] && base + loc[p] < rows) {
					const long long row = base + loc[p];
					PackU<T> o;
#pragma unroll
					for (int e = 0; e < VEC; ++e) o[e] = marchApplyOp(op, lhs, divisor, row + e, dot[p][e]);
					if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<PackU<T>*>(out + row));
					else *reinterpret_cast<PackU<T>*>(out + row) = o;
					if (dotMode) {
						const PackU<T> w = *reinterpret_cast<const PackU<T>*>(w1 + row);
#pragma unroll
						for (int e = 0; e < VEC; ++e) {
							if (dotMode == 2) acc0 += o[e] * o[e];
							acc1 += o[e] * w[e];
						}
					}
				}
			}
			if (more) {
				storeWindow(winOf(z + 2), use);  // (the buffer held plane z - 2: unread since the barrier that ended step z - 1)
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
					mk[p] = mk1[p];
					mk1[p] = use.m[p];
				}
				ldsBarrier();
			}
		};

		// prologue: the windows of planes z0 - 1, z0, z0 + 1, then the requests for plane z0 + 2
		__syncthreads();  // (the previous unit's last window reads are over)
		issue(fa, z0 - 1, false);
		issue(fb, z0, true);
		storeWindow(winOf(z0 - 1), fa);
		issue(fa, z0 + 1, z0 + 1 < z1);
		storeWindow(winOf(z0), fb);
#pragma unroll
		for (int p = 0; p < PACKS; ++p) mk[p] = fb.m[p];
		issue(fb, z0 + 2, z0 + 2 < z1);
		storeWindow(winOf(z0 + 1), fa);
#pragma unroll
		for (int p = 0; p < PACKS; ++p) mk1[p] = fa.m[p];
		__syncthreads();
		for (int z = z0; z < z1; z += 2) {
			step(z, fb, fa);
			if (z + 1 < z1) step(z + 1, fa, fb);
		}
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
// The same march for matrices whose diagonals VARY (MASKS: values[] is read): stencils with varying coefficients on big grids.  What
// the wave-private mask kernel (spmvPatternWaveKernel, smm_spmv_pattern.hip) loses there is x: 13.9 GB across the fabric for the 11.3 GB
// of the 512^3 fp64 stencil, x fetched 3.6 x.  Here x goes through the plane's LDS window and the lane's registers exactly as above;
// values[] keeps the wave kernel's route: the 64 consecutive rows of a wavefront own ONE contiguous run of values[], fetched with
// coalesced loads a whole plane ahead (one register set per sub-step, re-issued as soon as its values sit in LDS), passed through a
// wave-private LDS slice and read back by the row's
// lane.  A tile is 4 (fp64) or 2 (fp32) sub-steps of 256 rows (one row per lane and sub-step: LDS reads at lane stride -- conflict-free).
// Row starts come from ONE start[] per 64 rows plus a prefix sum of the masks' popcounts across the wavefront: no start[] stream.
// Bytes per fp64 row of the 7-point stencil: 56 (values) + 1 (mask; 4 until r05) + 8 (x) + 8 (out) = 73 against 84 in the wave kernel
// (and the 104 that kernel really moves).  Same products in the same order: -P, the near offsets ascending, +P; value n of a row pairs with its
// n-th set bit (ref:1484-1489): the reference's bits with one lane per row.
// ---------------------------------------------------------------------------------------------------------------------------------
// Q: sub-steps per tile and plane (a tile is Q x 256 rows).  4: tiles of 1024 rows, 229 VGPRs in fp64, two workgroups per CU; 2: 151 VGPRs
// (fp32 117), three (four) per CU and twice the tiles per plane.  fp32 always takes 2 (512^3: 1.238 -> 1.120 ms, 384^3 0.548 -> 0.497); fp64
// takes 2 below 10^8 rows (256^3 0.248 -> 0.234 ms, 320^3 0.538 -> 0.479, 384^3 0.862 -> 0.834, 448^3 1.373 -> 1.357) and 4 from there
// (512^3: 1.872 against 2.023) (profiles/r05/masks_march_substeps.txt)
template <typename T, int HP, int Q>
struct MasksMarchSet {
	T c[Q];
	PackU<T> h[HP];
	unsigned m[Q];
	int s0[Q];  // start[] of the first row of the lane's wavefront in sub-step q
};

template <typename T, int KMAX>
struct MasksMarchVals {
	T v[KMAX];
	int at;  // the lane's row begins here in the wavefront's run
};

template <typename T, int KMAX, bool NT, int HP, int Q>
__global__ __launch_bounds__(TPB, (sizeof(T) == 4 && KMAX <= 8 && HP <= 2 ? 3 : 2)) void spmvPatternMasksMarchKernel(int rows, int cols, int P, int nPlanes, int nT, int zc, int nChunks, int xcdTiles, int H,
                                                                      int nOff, int hasLo, int hasHi, const int* __restrict__ offs, const int* __restrict__ start,
                                                                      const T* __restrict__ values, const unsigned char* __restrict__ masks8, int opFlags,
                                                                      const T* lhs, const T* __restrict__ divisor, const T* __restrict__ x, T* out, int dotMode,
                                                                      const T* __restrict__ w1, T* __restrict__ partials, const int* __restrict__ doneFlag) {
	using Set = MasksMarchSet<T, HP, Q>;
	using Vals = MasksMarchVals<T, KMAX>;
	constexpr int VEC = 16 / sizeof(T);
	const int winLen = (Q * TPB) + 2 * H;
	T* sWin0 = reinterpret_cast<T*>(smmMarchLds);
	T* sWin1 = sWin0 + winLen;
	__shared__ T sVal[TPB / WAVE][WAVE * KMAX + KMAX];
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	const int op = opFlags & 0xFF;
	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int wv = t >> 6;
	const int nNear = nOff - hasLo - hasHi;
	const int haloPacks = 2 * H / VEC;
	const unsigned fullMask = nOff >= 32 ? 0xFFFFFFFFu : ((1u << nOff) - 1u);
	for (int i = lane; i < WAVE * KMAX + KMAX; i += WAVE) sVal[wv][i] = T(0);

	const int nGroups = min(8, static_cast<int>(gridDim.x));
	const int units = nT * nChunks;
	int uFirst, uStride, uEnd, tLo = 0, tCount = nT;
	if (xcdTiles) {
		const int g = blockIdx.x % nGroups;
		tLo = static_cast<int>(static_cast<long long>(g) * nT / nGroups);
		tCount = static_cast<int>(static_cast<long long>(g + 1) * nT / nGroups) - tLo;
		uFirst = blockIdx.x / nGroups;
		uStride = (static_cast<int>(gridDim.x) - g + nGroups - 1) / nGroups;
		uEnd = tCount * nChunks;
	} else {
		uFirst = blockIdx.x;
		uStride = gridDim.x;
		uEnd = units;
	}
	T acc0 = T(0), acc1 = T(0);

	for (int u = uFirst; u < uEnd; u += uStride) {
		const int chunk = u / tCount;
		const int tile = tLo + (u - chunk * tCount);
		const int z0 = chunk * zc;
		const int z1 = min(nPlanes, z0 + zc);
		const int r0 = tile * (Q * TPB);
		const int bAct = min((Q * TPB), P - r0);
		auto haloWin = [&](int i) { return i < H / VEC ? i * VEC : H + bAct + (i - H / VEC) * VEC; };
		auto issue = [&](Set& f, int z, bool wantCentre, bool wantWindow) {
			const bool inside = z >= 0 && z < nPlanes;
			const long long base = static_cast<long long>(z) * P + r0;
#pragma unroll
			for (int q = 0; q < Q; ++q) {
				const int l = q * TPB + t;
				const bool live = inside && l < bAct && base + l < rows;
				f.c[q] = live && wantCentre ? x[base + l] : T(0);  // (cacheable: a neighbouring tile reads these lines as its halo)
				if (wantWindow) {
					f.m[q] = live ? static_cast<unsigned>(__builtin_nontemporal_load(masks8 + base + l)) : 0u;  // (at most 8 offsets: one byte per row since r05)
					const int lw = q * TPB + (t & ~(WAVE - 1));  // the wavefront's first row of this sub-step
					f.s0[q] = inside && lw < bAct && base + lw < rows ? start[base + lw] : 0;
				}
			}
			if (inside && wantWindow) {
#pragma unroll
				for (int k = 0; k < HP; ++k) {
					const int i = k * TPB + t;
					if (i < haloPacks) {
						long long gidx = base - H + haloWin(i);
						gidx = gidx < 0 ? 0 : (gidx + VEC > cols ? cols - VEC : gidx);  // (a pack that sticks out of x is never used by a live entry)
						f.h[k] = *reinterpret_cast<const PackU<T>*>(x + gidx);
					}
				}
			}
		};
		auto storeWindow = [&](T* win, const Set& f) {
#pragma unroll
			for (int q = 0; q < Q; ++q) {
				if (q * TPB + t < bAct) win[H + q * TPB + t] = f.c[q];
			}
#pragma unroll
			for (int k = 0; k < HP; ++k) {
				const int i = k * TPB + t;
				if (i < haloPacks) *reinterpret_cast<PackV<T>*>(win + haloWin(i)) = f.h[k];
			}
		};
		// the wavefront's run of values[] for one sub-step: the lanes' popcounts give every row's place in it
		auto fetchVals = [&](Vals& V, unsigned m, int s0) {
			const int pc = __popc(m);
			int incl = pc;
#pragma unroll
			for (int o = 1; o < WAVE; o <<= 1) {
				const int up = __shfl_up(incl, o, WAVE);
				if (lane >= o) incl += up;
			}
			const int total = __shfl(incl, WAVE - 1, WAVE);
			V.at = incl - pc;
#pragma unroll
			for (int k = 0; k < KMAX; ++k) {
				const int i = k * WAVE + lane;
				V.v[k] = i < total ? __builtin_nontemporal_load(values + s0 + i) : T(0);
			}
		};

		T xp[Q];
		unsigned mk[Q];
		int s0c[Q];
		Set fa, fb;
		Vals vals[Q];  // one request set per sub-step: re-issued for the NEXT plane as soon as its values sit in LDS (a whole plane ahead)

		auto subStep = [&](int z, int q, Vals& V, const Set& use, const T* win, bool more) {
#pragma unroll
			for (int k = 0; k < KMAX; ++k) sVal[wv][k * WAVE + lane] = V.v[k];
			const int at = V.at;
			if (more) fetchVals(V, use.m[q], use.s0[q]);
			const long long base = static_cast<long long>(z) * P + r0;
			const int l = q * TPB + t;
			const bool live = l < bAct && base + l < rows;
			const unsigned m = mk[q];
			const bool masked = !__all(m == fullMask || !live);
			T dot = T(0);
			auto fold = [&](int j, T xv) {
				const int idx = masked ? at + __popc(m & ((1u << j) - 1u)) : at + j;
				const T next = smmFma(sVal[wv][idx], xv, dot);
				dot = (!masked || ((m >> j) & 1u) != 0u) ? next : dot;
			};
			if (hasLo) fold(0, xp[q]);
			for (int j = 0; j < nNear; ++j) fold(hasLo + j, win[H + l + offs[hasLo + j]]);
			if (hasHi) fold(hasLo + nNear, use.c[q]);
			if (live) {
				const long long row = base + l;
				const T o = marchApplyOp(op, lhs, divisor, row, dot);
				if constexpr (NT) __builtin_nontemporal_store(o, out + row);
				else out[row] = o;
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		};
		// one plane: `use` holds plane z + 1, `re` is re-issued for plane z + 2; every sub-step's values were requested a whole plane earlier
		auto step = [&](int z, Set& use, Set& re) {
			T* win = ((z - z0) & 1) ? sWin1 : sWin0;
			T* winNext = ((z - z0) & 1) ? sWin0 : sWin1;
			const bool more = z + 1 < z1;
			if (more) issue(re, z + 2, z + 2 < z1 || hasHi, z + 2 < z1);
#pragma unroll
			for (int q = 0; q < Q; ++q) subStep(z, q, vals[q], use, win, more);
			if (more) {
#pragma unroll
				for (int q = 0; q < Q; ++q) xp[q] = win[H + q * TPB + t];
				storeWindow(winNext, use);
#pragma unroll
				for (int q = 0; q < Q; ++q) {
					mk[q] = use.m[q];
					s0c[q] = use.s0[q];
				}
				ldsBarrier();
			}
		};

		issue(fa, z0 - 1, hasLo != 0, false);
#pragma unroll
		for (int q = 0; q < Q; ++q) xp[q] = fa.c[q];
		issue(fa, z0, true, true);
		issue(fb, z0 + 1, z0 + 1 < z1 || hasHi, z0 + 1 < z1);
		__syncthreads();  // (the previous unit's last window reads are over)
		storeWindow(sWin0, fa);
#pragma unroll
		for (int q = 0; q < Q; ++q) {
			mk[q] = fa.m[q];
			s0c[q] = fa.s0[q];
		}
		__syncthreads();
#pragma unroll
		for (int q = 0; q < Q; ++q) fetchVals(vals[q], mk[q], s0c[q]);
		for (int z = z0; z < z1; z += 2) {
			step(z, fb, fa);
			if (z + 1 < z1) step(z + 1, fa, fb);
		}
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
void preloadMarchUnit() {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(marchNarrowMasks));
	(void)hipGetLastError();
}

// ---- the plan (once per matrix, at the end of the MASKS analysis; caller holds tileMutex) and the launch ----------------------------------
// From the sorted offset list: the far pair is -P / +P with P the largest |offset|, present on at least one side, when the rest of the
// offsets is near (|off| <= the halo a lane can hold), planes are at least four tiles large and the rows are a whole number of planes
// (grids in natural order: P = nx ny); otherwise, when EVERY offset is near (2-D grids, narrow bands), the matrix is one plane.  A matrix
// that is neither keeps the gather kernel (spmvPatternConstKernel).
// From how many rows the march kernels serve a matrix (profiles/r04/march_threshold.txt: cubic grids, march forced on / off, one box;
// gather / march for constant diagonals, wave / march for values read, microseconds):
//   fp64 constant: 96^3 8.0 / 8.0, 108^3 10.4 / 9.5, 128^3 16.2 / 12.4, 160^3 28.9 / 24.4, 256^3 143 / 70      -> from 2^21 rows (at 108^3 the
//                  9 % of the bare SpMV do not survive inside BiCGStab: 24.4 against 23.4 ms for config 5's stand-in, Jacobi fold 26.2 / 24.1)
//   fp32 constant: 108^3 8.5 / 9.0, 128^3 13.0 / 9.5, 144^3 17.1 / 17.6, 160^3 22.0 / 20.4, 256^3 86 / 55      -> from 2^21 rows
//   fp64 values:   160^3 63.5 / 65.2, 200^3 122 / 124, 256^3 299 / 234, 512^3 2470 / 1840                       -> r04: from 12 x 2^20 rows
//   fp32 values:   200^3 72 / 109, 256^3 156 / 171, 512^3 1417 / 1261                                          -> r04: from 2^26 rows
//   r05, two sub-steps per tile (profiles/r05/masks_march_thresholds.txt):
//   fp64 values:   160^3 67.1 / 67.3, 200^3 131 / 112, 232^3 200 / 171                                          -> from 6 x 2^20 rows
//   fp32 values:   200^3 68.5 / 71.7, 256^3 162 / 143, 320^3 342 / 282, 384^3 581 / 490                           -> from 2^24 rows
// (below, a unit's few planes do not fill the chip's workgroup slots).
// SMM_HIP_MARCH_MIN_ROWS (environment, all four) and smm_hip_set_march_min_rows (tests run the kernels on smaller grids) override them.
static std::atomic<long long> g_marchMinRowsConst{-1}, g_marchMinRowsMasks{-1};

static long long marchMinRows(bool masksKernel, int dtype) {
	const long long forced = (masksKernel ? g_marchMinRowsMasks : g_marchMinRowsConst).load(std::memory_order_relaxed);
	if (forced >= 0) return forced;
	static const long long env = [] {
		const char* e = getenv("SMM_HIP_MARCH_MIN_ROWS");
		return e ? atoll(e) : -1LL;
	}();
	if (env >= 0) return env;
	const bool f32 = dtype == SMM_DTYPE_F32;
	if (masksKernel) return f32 ? 1LL << 24 : 6LL << 20;  // (r05, two sub-steps per tile: profiles/r05/masks_march_thresholds.txt; r04: 2^26 / 12 x 2^20)
	return 1LL << 21;
}

void planMarch(smm_hip_csr* m) {
	m->march_ok = false;
	m->march_clusters = false;
	const long long minRows = std::min(marchMinRows(false, m->dtype), marchMinRows(true, m->dtype));
	const std::vector<int>& offs = m->pat_offs_host;
	const int k = static_cast<int>(offs.size());
	if (k < 1 || k > 32 || m->rows != m->cols || m->rows < minRows) return;  // (constant diagonals or not: the masks kernels march as well)
	const int vec = m->dtype == SMM_DTYPE_F32 ? 4 : 2;
	const int hCap = 4 * TPB * vec / 2;  // 2 H / VEC halo packs <= 4 per lane (the kernel is compiled for 2 and for 4)
	auto roundUp = [vec](int h) { return (h + vec - 1) / vec * vec; };
	const int far = std::max(std::abs(offs.front()), std::abs(offs.back()));
	// (a) the far pair
	if (far >= 4 * TPB * MARCH_RMAX && m->rows % vec == 0 && far % vec == 0 && m->rows / far >= 2) {  // (the last plane may be partial: a slab of a partitioned grid)
		const int lo = offs.front() == -far ? 1 : 0, hi = offs.back() == far ? 1 : 0;
		int h = 0;
		for (int j = lo; j < k - hi; ++j) h = std::max(h, std::abs(offs[j]));
		if (k - lo - hi >= 1 && roundUp(std::max(h, 1)) <= hCap && 2 * h < far) {
			m->march_ok = true;
			m->march_P = far;
			m->march_H = roundUp(std::max(h, 1));
			m->march_lo = lo;
			m->march_hi = hi;
			return;
		}
	}
	// (a') far offsets in clusters around -P and +P (19- / 27-point stencils): the offsets beyond the halo cap split into one group below and
	// one above, the plane size is the centre of a group (both centres must agree), every offset lies within H of -P, 0 or +P.  Constant
	// diagonals only (the three-window kernel has no values[] form)
	if (m->pat_const && far > hCap && m->rows % vec == 0) {
		int nLo = 0, nHi = 0;
		while (nLo < k && offs[nLo] < -hCap) ++nLo;
		while (nHi < k && offs[k - 1 - nHi] > hCap) ++nHi;
		long long P = 0;
		bool ok = nLo + nHi < k && (nLo > 1 || nHi > 1);
		if (ok && nHi > 0) {
			const long long sum = static_cast<long long>(offs[k - nHi]) + offs[k - 1];
			ok = sum % 2 == 0;
			P = sum / 2;
		}
		if (ok && nLo > 0) {
			const long long sum = -(static_cast<long long>(offs[0]) + offs[nLo - 1]);
			ok = sum % 2 == 0 && (nHi == 0 || sum / 2 == P);
			P = sum / 2;
		}
		if (ok) {
			int h = 1;
			for (int j = 0; j < k; ++j) {
				const long long centre = j < nLo ? -P : (j >= k - nHi ? P : 0);
				h = std::max<long long>(h, std::llabs(offs[j] - centre));
			}
			ok = P >= 4LL * TPB * MARCH_RMAX && P % vec == 0 && m->rows / P >= 2 && roundUp(h) <= hCap && 2LL * roundUp(h) < P && P < (1LL << 30);
			if (ok) {
				m->march_ok = true;
				m->march_clusters = true;
				m->march_P = static_cast<int>(P);
				m->march_H = roundUp(h);
				m->march_lo = nLo;
				m->march_hi = nHi;
				return;
			}
		}
	}
	// (b) one plane: every offset is near
	if (roundUp(std::max(far, 1)) <= hCap && m->rows % vec == 0) {
		m->march_ok = true;
		m->march_P = m->rows;
		m->march_H = roundUp(std::max(far, 1));
		m->march_lo = m->march_hi = 0;
	}
}

// the 32-bit copy of the masks (caller: the MASKS analysis, after the plan said yes; the buffer is the handle's).  Built only where a march
// kernel can ever serve the matrix: constant diagonals, or the structural conditions of the masks march (masksMarchApplies without the row
// threshold, which tests move).  Waits for the narrowing kernel (r05): ensurePattern publishes pat_state = 1 right after this
// returns, and a concurrent SpMV of the same matrix on ANOTHER stream would otherwise run a march kernel on masks that are not written yet.
int marchBuildMasks32(smm_hip_csr* m, hipStream_t s) {
	if (!m->march_ok || m->d_pat_masks32) return SMM_HIP_OK;
	const bool masksShape = m->pat_k <= 8 && m->march_P > 0 && (m->rows + m->march_P - 1) / m->march_P >= 8;
	if (!m->pat_const && !masksShape) return SMM_HIP_OK;
	// (the kernels that stream one byte of mask per row: the KN = 5 instances of the constant-diagonal two-window kernel -- five near
	// offsets, no clusters -- and the masks march, which serves matrices of at most 8 offsets only)
	const bool wantBytes = (m->pat_const && !m->march_clusters && m->pat_k - m->march_lo - m->march_hi == 5) || masksShape;
	void *p = nullptr, *p8 = nullptr;
	SMM_TRY(devAlloc(&p, static_cast<size_t>(m->rows) * sizeof(unsigned) + 16));
	if (wantBytes && devAlloc(&p8, static_cast<size_t>(m->rows) + 16) != SMM_HIP_OK) {
		(void)hipGetLastError();
		p8 = nullptr;  // (no room: the KN = 0 instances read the 32-bit masks)
	}
	const int grid = static_cast<int>(std::min<long long>((m->rows + 255LL) / 256, numCUs() * 16LL));
	marchNarrowMasks<<<grid, 256, 0, s>>>(m->rows, m->d_pat_masks, static_cast<unsigned*>(p));
	if (p8) marchNarrowMasks8<<<grid, 256, 0, s>>>(m->rows, m->d_pat_masks, static_cast<unsigned char*>(p8));
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(s);
	if (e != hipSuccess) {
		devFree(p);
		if (p8) devFree(p8);
		return hipFail(e, "marchNarrowMasks", __FILE__, __LINE__);
	}
	m->d_pat_masks8 = static_cast<unsigned char*>(p8);
	m->d_pat_masks32 = static_cast<unsigned*>(p);
	return SMM_HIP_OK;
}

// Launch plumbing shared by the march kernels, per template instantiation (`State` is a static of the launcher):
//   * hipFuncAttributeMaxDynamicSharedMemorySize is raised whenever a launch needs more dynamic LDS than the largest size granted so far
//     (r04 raised it once, to the FIRST qualifying matrix's size: a later matrix with a larger halo got a failed launch; the flag was a plain
//     bool written by concurrent solves).  A failed raise is reported to the caller, who falls back to the gather / wave kernel.
//   * the occupancy query is cached per LDS size (it sat on the solvers' hot path beside kernels of 10-20 us), the environment overrides are
//     read once.
struct MarchLaunchState {
	std::atomic<int> granted{0};       // largest dynamic LDS size the attribute was set to
	std::atomic<long long> occ{0};     // (lds << 8) | perCU of the last occupancy query
	std::mutex mu;
};
static int marchEnvInt(const char* name) {
	const char* env = getenv(name);
	return env ? atoi(env) : 0;
}
static int marchWgsPerCuOverride() {
	static const int v = std::max(0, marchEnvInt("SMM_HIP_MARCH_WGS_PER_CU"));
	return v;
}
static int marchZcOverride() {
	static const int v = std::max(0, marchEnvInt("SMM_HIP_MARCH_ZC"));
	return v;
}
// false: this launch cannot have its LDS (the caller keeps the kernel that needs none)
template <typename Kernel>
static bool marchPrepare(MarchLaunchState& st, Kernel kernel, size_t lds, size_t staticLds, int* perCU) {
	constexpr size_t DEFAULT_LIMIT = 64 * 1024, HW_LIMIT = 160 * 1024;
	if (lds + staticLds > HW_LIMIT) return false;
	if (lds + staticLds > DEFAULT_LIMIT && static_cast<int>(lds) > st.granted.load(std::memory_order_acquire)) {
		std::lock_guard<std::mutex> lock(st.mu);
		if (static_cast<int>(lds) > st.granted.load(std::memory_order_relaxed)) {
			if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
				(void)hipGetLastError();
				return false;
			}
			st.granted.store(static_cast<int>(lds), std::memory_order_release);
		}
	}
	const long long cached = st.occ.load(std::memory_order_acquire);
	if (cached != 0 && static_cast<size_t>(cached >> 8) == lds) {
		*perCU = static_cast<int>(cached & 0xFF);
	} else {
		int n = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, TPB, lds) != hipSuccess || n < 1) {
			(void)hipGetLastError();
			n = 2;
		}
		n = std::min(n, 255);
		st.occ.store((static_cast<long long>(lds) << 8) | n, std::memory_order_release);
		*perCU = n;
	}
	if (marchWgsPerCuOverride() > 0) *perCU = marchWgsPerCuOverride();
	return true;
}

template <typename T, int R, int KN, bool NT, int HP, bool FUSE = false>
static bool launchMarchKN(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s, const MarchFuse<T>& fz = MarchFuse<T>()) {
	const int P = m->march_P, H = m->march_H;
	const int nPlanes = (m->rows + P - 1) / P;
	constexpr int MARCH_B = TPB * R;
	const int nT = (P + MARCH_B - 1) / MARCH_B;
	const size_t lds = 2 * static_cast<size_t>(MARCH_B + 2 * H) * sizeof(T);
	static MarchLaunchState state;
	int perCU = 0;
	if (!marchPrepare(state, spmvPatternConstMarchKernel<T, R, KN, NT, HP, FUSE>, lds, 128 /* red[], redF[] */, &perCU)) return false;
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();  // room for the RCCL kernel beside A_loc (smm_dist.hip)
	op &= ~SPMV_LEAVE_ROOM;
	const int resident = cus * perCU;
	// planes per unit: enough units for ~1.5 x the chip's workgroup slots -- on mid-size grids THAT is what matters, not the two extra
	// planes a unit loads for its first and last plane's far entries (they are L2 hits): 128^3 with 8 / 4 / 2 planes per unit 19.6 / 13.9 /
	// 12.2 us (the gather kernel: 16.6), 160^3 24.2 / 21.6 / 24.0 (29.7), 512^3 flat from 16 to 128 (profiles/r04/march_threshold.txt); at
	// least 2 planes, so that something is requested ahead
	int zc = nPlanes;
	if (nPlanes > 1) {
		const int wantChunks = std::max(1, std::min(nPlanes, (3 * resident / 2 + nT - 1) / nT));
		zc = std::max(std::min(2, nPlanes), (nPlanes + wantChunks - 1) / wantChunks);
		if (marchZcOverride() > 0) zc = std::max(1, std::min(nPlanes, marchZcOverride()));
	}
	const int nChunks = (nPlanes + zc - 1) / zc;
	const long long units = static_cast<long long>(nT) * nChunks;
	int grid = static_cast<int>(std::max<long long>(1, std::min<long long>(std::min<long long>(units, resident), NPART)));
	const int xcdTiles = (nT % 8 == 0 || nT >= 64) && grid >= 8 ? 1 : 0;
	if (xcdTiles) grid -= grid % 8;  // the same number of workgroups in every XCD group
	spmvPatternConstMarchKernel<T, R, KN, NT, HP, FUSE><<<grid, TPB, lds, s>>>(m->rows, m->cols, P, nPlanes, nT, zc, nChunks, xcdTiles, H, m->pat_k, m->march_lo, m->march_hi,
	                                                                   m->d_pat_off, m->d_pat_cval, KN > 0 ? static_cast<const void*>(m->d_pat_masks8) : static_cast<const void*>(m->d_pat_masks32), op, lhs,
	                                                                   divisor, x, out, dotMode, w1, partials, doneFlag, fz);
	return true;
}

template <typename T, int R, bool NT, int HP>
static bool launchMarch3(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials, const int* doneFlag,
                         hipStream_t s) {
	const int P = m->march_P, H = m->march_H;
	const int nPlanes = (m->rows + P - 1) / P;
	constexpr int MARCH_B = TPB * R;
	const int nT = (P + MARCH_B - 1) / MARCH_B;
	const size_t lds = 4 * static_cast<size_t>(MARCH_B + 2 * H) * sizeof(T);  // the windows of planes z - 1, z, z + 1 and the one being filled
	static MarchLaunchState state;
	int perCU = 0;
	if (!marchPrepare(state, spmvPatternConstMarch3Kernel<T, R, NT, HP>, lds, 64 /* red[] */, &perCU)) return false;
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();
	op &= ~SPMV_LEAVE_ROOM;
	const int resident = cus * perCU;
	// (a unit loads TWO extra windows -- the planes below its first and above its last --, so longer z-chunks than the two-window kernel's)
	int zc = nPlanes;
	if (nPlanes > 1) {
		const int wantChunks = std::max(1, std::min(nPlanes, (3 * resident / 2 + nT - 1) / nT));
		zc = std::max(std::min(4, nPlanes), (nPlanes + wantChunks - 1) / wantChunks);
		if (marchZcOverride() > 0) zc = std::max(1, std::min(nPlanes, marchZcOverride()));
	}
	const int nChunks = (nPlanes + zc - 1) / zc;
	const long long units = static_cast<long long>(nT) * nChunks;
	int grid = static_cast<int>(std::max<long long>(1, std::min<long long>(std::min<long long>(units, resident), NPART)));
	const int xcdTiles = (nT % 8 == 0 || nT >= 64) && grid >= 8 ? 1 : 0;
	if (xcdTiles) grid -= grid % 8;
	spmvPatternConstMarch3Kernel<T, R, NT, HP><<<grid, TPB, lds, s>>>(m->rows, m->cols, P, nPlanes, nT, zc, nChunks, xcdTiles, H, m->pat_k, m->march_lo, m->march_hi,
	                                                               m->d_pat_off, m->d_pat_cval, m->d_pat_masks32, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag);
	return true;
}

// ConjugateGradient's launches on the two-window kernel take half tiles (fp32 4 rows per lane, fp64 2) when the outputs are beyond the caches
// (the size at which the direction is formed inside the SpMV: SPMV_HALF_TILES, smm_internal.h).  SMM_HIP_MARCH_FUSE_FULL_TILES=1: full tiles
// everywhere (A/B measurements)
bool cgHalfTiles(const smm_hip_csr* m, size_t elemBytes) {
	static const bool fullTiles = [] {
		const char* env = getenv("SMM_HIP_MARCH_FUSE_FULL_TILES");
		return env && atoi(env) != 0;
	}();
	static const bool rowsForced = getenv("SMM_HIP_MARCH_R") != nullptr;
	return !fullTiles && !rowsForced && !m->march_clusters && (spmvOutFlags(m, elemBytes) & SPMV_NT_OUT) != 0;
}

// true: the launch went to the march kernel.  SMM_HIP_CONST_MARCH=0 keeps the gather kernel (A/B measurements).
template <typename T>
bool launchPatConstMarch(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s) {
	static const bool enabled = [] {
		const char* env = getenv("SMM_HIP_CONST_MARCH");
		return env ? atoi(env) != 0 : true;
	}();
	if (!enabled || !constMarchApplies(m)) return false;
	const bool nt = (spmvOutFlags(m, sizeof(T)) & SPMV_NT_OUT) != 0;  // outputs too large to still be cached when the next kernel reads them
	const bool halfTiles = (op & SPMV_HALF_TILES) != 0 && cgHalfTiles(m, sizeof(T));  // (ConjugateGradient's launches: smm_internal.h)
	op &= ~SPMV_HALF_TILES;
	if (m->march_clusters) {
		// far offsets in clusters: the three-window kernel, 4 rows per lane in fp64 and 8 in fp32 as below
		const bool hp2c = 2 * m->march_H / (16 / static_cast<int>(sizeof(T))) <= 2 * TPB;
		constexpr int RC = sizeof(T) == 8 ? 4 : 8;
		return nt && hp2c ? launchMarch3<T, RC, true, 2>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)
		       : nt     ? launchMarch3<T, RC, true, 4>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)
		       : hp2c   ? launchMarch3<T, RC, false, 2>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)
		                : launchMarch3<T, RC, false, 4>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s);
	}
	const int nNear = m->pat_k - m->march_lo - m->march_hi;
	// rows per lane (profiles/r04/march_rows_per_lane.txt, 512^3): fp64 4 -- tiles of 1024 rows, 120 VGPRs, four workgroups per CU: 0.565 ms
	// against 0.604 with 8 (185 VGPRs, two per CU); fp32 8 -- 0.318 against 0.335.  SMM_HIP_MARCH_R=4 / 8 forces one (A/B measurements).
	static const int forcedRows = [] {
		const char* env = getenv("SMM_HIP_MARCH_R");
		return env ? atoi(env) : 0;
	}();
	const int rowsPerLane = forcedRows == 4 || forcedRows == 8 ? forcedRows : (sizeof(T) == 8 || halfTiles ? 4 : 8);
	const int vec = 16 / static_cast<int>(sizeof(T));
	const bool r4 = rowsPerLane == 4;
	const bool hp2 = 2 * m->march_H / vec <= 2 * TPB;  // the halo fits two packs per lane (fewer registers)
#define SMM_MARCH_GO2(RV, KNV)                                                                                                        \
	(nt && hp2 ? launchMarchKN<T, RV, KNV, true, 2>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)                  \
	 : nt      ? launchMarchKN<T, RV, KNV, true, 4>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)                  \
	 : hp2     ? launchMarchKN<T, RV, KNV, false, 2>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)                 \
	           : launchMarchKN<T, RV, KNV, false, 4>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s))
#define SMM_MARCH_GO(KNV) (r4 ? SMM_MARCH_GO2(4, KNV) : SMM_MARCH_GO2(8, KNV))
	bool launched;
	if constexpr (sizeof(T) == 8) {
		if (halfTiles && !forcedRows) {  // (fp64 half tiles: 2 rows per lane; only with non-temporal outputs, which cgHalfTiles implies)
			launched = nNear == 5 && m->d_pat_masks8 ? (hp2 ? launchMarchKN<T, 2, 5, true, 2>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)
			                                                : launchMarchKN<T, 2, 5, true, 4>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s))
			                                         : (hp2 ? launchMarchKN<T, 2, 0, true, 2>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)
			                                                : launchMarchKN<T, 2, 0, true, 4>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s));
			return launched;
		}
	}
	launched = nNear == 5 && m->d_pat_masks8 ? SMM_MARCH_GO(5) : SMM_MARCH_GO(0);  // (KN = 5 reads the byte masks)
#undef SMM_MARCH_GO2
#undef SMM_MARCH_GO
	return launched;  // false: the windows do not fit the LDS a launch can be granted -- the gather kernel serves the matrix
}

static std::atomic<int> g_cgFuseP{1};  // smm_hip_set_cg_fuse_p (tests: A/B against the launch that forms p in cgLazyXP)

static bool marchEnabled() {
	static const bool enabled = [] {
		const char* env = getenv("SMM_HIP_CONST_MARCH");
		return env ? atoi(env) != 0 : true;
	}();
	return enabled;
}

// the form the fused launch exists in: the two-window kernel, default rows per lane, non-temporal outputs (i.e. vectors beyond 64 MB -- the
// only size at which forming p in the SpMV pays), PATTERN family with constant diagonals at one lane per row
bool constMarchFusable(const smm_hip_csr* m, size_t elemBytes) {
	static const bool allowed = [] {
		const char* env = getenv("SMM_HIP_CG_FUSE_P");
		return env ? atoi(env) != 0 : true;
	}();
	static const bool rowsForced = getenv("SMM_HIP_MARCH_R") != nullptr;
	return allowed && g_cgFuseP.load(std::memory_order_relaxed) != 0 && !rowsForced && marchEnabled() && m->family() == SMM_SPMV_PATTERN && m->lanes() == 1 && m->pat_state.load(std::memory_order_acquire) > 0 &&
	       m->pat_encoding == 0 && m->pat_const && !m->pat_const_off && constMarchApplies(m) && !m->march_clusters && (spmvOutFlags(m, elemBytes) & SPMV_NT_OUT) != 0;
}

template <typename T>
bool launchConstMarchFusedP(const smm_hip_csr* m, const T* pOld, T* Ap, T* partials, const int* doneFlag, const CgFuseArgs<T>& f, hipStream_t s) {
	if (!constMarchFusable(m, sizeof(T))) return false;
	MarchFuse<T> fz;
	fz.r = f.r;
	fz.pNew = f.pNew;
	fz.bk = f.bk;
	fz.partsC = f.partsC;
	fz.totalsC = f.totalsC;
	fz.eps = f.eps;
	fz.par = f.par;
	fz.iter = f.iter;
	const int nNear = m->pat_k - m->march_lo - m->march_hi;
	// rows per lane: what ConjugateGradient's plain launches use for this matrix (cgHalfTiles: fp32 4 instead of 8, fp64 2 instead of 4 --
	// the fused launch holds two streams' request sets and fits three workgroups per CU at 143-154 VGPRs instead of two at 195-205: 512^3
	// fp32 0.95 -> 0.83 ms per CG iteration, fp64 1.71 -> 1.67, profiles/r05/cg_fuse_half_tiles.txt); the partial sums of p.Ap follow the
	// tiles, so all loop forms share them
	constexpr int RF = sizeof(T) == 8 ? 4 : 8;
	const bool half = cgHalfTiles(m, sizeof(T));
	const bool hp2 = 2 * m->march_H / (16 / static_cast<int>(sizeof(T))) <= 2 * TPB;
	const bool kn5 = nNear == 5 && m->d_pat_masks8;
#define SMM_FUSE_GO(RV, KNV, HPV) launchMarchKN<T, RV, KNV, true, HPV, true>(m, SMM_OP_ASSIGN | f.extraFlags, nullptr, nullptr, pOld, Ap, 1, nullptr, partials, doneFlag, s, fz)
	bool launched;
	if (half) {
		constexpr int RH = RF / 2;
		launched = kn5 ? (hp2 ? SMM_FUSE_GO(RH, 5, 2) : SMM_FUSE_GO(RH, 5, 4)) : (hp2 ? SMM_FUSE_GO(RH, 0, 2) : SMM_FUSE_GO(RH, 0, 4));
		return launched;
	}
	launched = kn5 ? (hp2 ? SMM_FUSE_GO(RF, 5, 2) : SMM_FUSE_GO(RF, 5, 4)) : (hp2 ? SMM_FUSE_GO(RF, 0, 2) : SMM_FUSE_GO(RF, 0, 4));
#undef SMM_FUSE_GO
	return launched;
}
template bool launchConstMarchFusedP<float>(const smm_hip_csr*, const float*, float*, float*, const int*, const CgFuseArgs<float>&, hipStream_t);
template bool launchConstMarchFusedP<double>(const smm_hip_csr*, const double*, double*, double*, const int*, const CgFuseArgs<double>&, hipStream_t);

template <typename T, int KMAX, bool NT, int HP, int Q>
static bool launchMasksMarchK(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                              const int* doneFlag, hipStream_t s) {
	const int P = m->march_P, H = m->march_H;
	const int nPlanes = (m->rows + P - 1) / P;
	const int nT = (P + (Q * TPB) - 1) / (Q * TPB);
	const size_t lds = 2 * static_cast<size_t>((Q * TPB) + 2 * H) * sizeof(T);
	static MarchLaunchState state;
	int perCU = 0;
	// (the kernel holds 8-17 KB of static LDS beside the windows: together they may pass the 64 KB default)
	if (!marchPrepare(state, spmvPatternMasksMarchKernel<T, KMAX, NT, HP, Q>, lds, (TPB / WAVE) * (WAVE * KMAX + KMAX) * sizeof(T) + 64 /* sVal[], red[] */, &perCU)) return false;
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();
	op &= ~SPMV_LEAVE_ROOM;
	const int resident = cus * perCU;
	int zc = nPlanes;
	if (nPlanes > 1) {
		const int wantChunks = std::max(1, std::min(nPlanes, (4 * resident + nT - 1) / nT));
		zc = std::max(std::min(8, nPlanes), (nPlanes + wantChunks - 1) / wantChunks);
		if (marchZcOverride() > 0) zc = std::max(1, std::min(nPlanes, marchZcOverride()));
	}
	const int nChunks = (nPlanes + zc - 1) / zc;
	const long long units = static_cast<long long>(nT) * nChunks;
	int grid = static_cast<int>(std::max<long long>(1, std::min<long long>(std::min<long long>(units, resident), NPART)));
	const int xcdTiles = (nT % 8 == 0 || nT >= 64) && grid >= 8 ? 1 : 0;
	if (xcdTiles) grid -= grid % 8;
	spmvPatternMasksMarchKernel<T, KMAX, NT, HP, Q><<<grid, TPB, lds, s>>>(m->rows, m->cols, P, nPlanes, nT, zc, nChunks, xcdTiles, H, m->pat_k, m->march_lo, m->march_hi,
	                                                                  m->d_pat_off, m->d_start, static_cast<const T*>(m->d_values), m->d_pat_masks8, op, lhs, divisor, x,
	                                                                  out, dotMode, w1, partials, doneFlag);
	return true;
}

bool masksMarchApplies(const smm_hip_csr* m) {
	return m->march_ok && !m->march_clusters && m->d_pat_masks8 && m->pat_k <= 8 && m->march_P > 0 && (m->rows + m->march_P - 1) / m->march_P >= 8 &&
	       m->rows >= marchMinRows(true, m->dtype);
}
bool constMarchApplies(const smm_hip_csr* m) {
	return m->march_ok && m->d_pat_masks32 && m->rows >= marchMinRows(false, m->dtype) && (!m->march_clusters || (m->pat_const && !m->pat_const_off));
}

// true: the launch went to the march form of the masks kernels (values[] read).  SMM_HIP_MASKS_MARCH=0 keeps the wave kernel.
template <typename T>
bool launchPatMasksMarch(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s) {
	static const bool enabled = [] {
		const char* env = getenv("SMM_HIP_MASKS_MARCH");
		return env ? atoi(env) != 0 : true;
	}();
	// rows of 9 .. 16 entries keep the wave kernel (four value sets of 16 would not fit the registers), and so do matrices of fewer than 8
	// planes -- in ONE-plane mode (every offset near) a unit is a single step with nothing requested ahead, and the wave kernel wins:
	// 7 random diagonals within +-1000, 8 M rows, fp64: 0.123 ms against 0.153 (profiles/r04/one_plane_mode.txt)
	if (!enabled || !masksMarchApplies(m)) return false;
	const bool nt = (spmvOutFlags(m, sizeof(T)) & SPMV_NT_OUT) != 0;
	const bool hp2 = 2 * m->march_H / (16 / static_cast<int>(sizeof(T))) <= 2 * TPB;
#define SMM_MM_GO(KV, QV)                                                                                                   \
	(nt && hp2 ? launchMasksMarchK<T, KV, true, 2, QV>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)      \
	 : nt      ? launchMasksMarchK<T, KV, true, 4, QV>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)      \
	 : hp2     ? launchMasksMarchK<T, KV, false, 2, QV>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s)     \
	           : launchMasksMarchK<T, KV, false, 4, QV>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s))
	static const int forcedQ = [] {
		const char* env = getenv("SMM_HIP_MASKS_MARCH_Q");  // 2 / 4: force the sub-steps per tile (A/B measurements)
		return env ? atoi(env) : 0;
	}();
	const bool q2 = forcedQ == 2 || (forcedQ != 4 && (sizeof(T) == 4 || m->rows < 100000000LL));
	const bool launched = q2 ? SMM_MM_GO(8, 2) : SMM_MM_GO(8, 4);
#undef SMM_MM_GO
	return launched;  // false: the wave kernel serves the matrix
}

template bool launchPatMasksMarch<float>(const smm_hip_csr*, int, const float*, const float*, const float*, float*, int, const float*, float*, const int*, hipStream_t);
template bool launchPatMasksMarch<double>(const smm_hip_csr*, int, const double*, const double*, const double*, double*, int, const double*, double*, const int*, hipStream_t);

template bool launchPatConstMarch<float>(const smm_hip_csr*, int, const float*, const float*, const float*, float*, int, const float*, float*, const int*, hipStream_t);
template bool launchPatConstMarch<double>(const smm_hip_csr*, int, const double*, const double*, const double*, double*, int, const double*, double*, const int*, hipStream_t);

}  // namespace smm

extern "C" int smm_hip_set_cg_fuse_p(int on) {
	smm::g_cgFuseP.store(on ? 1 : 0, std::memory_order_relaxed);
	return SMM_HIP_OK;
}

extern "C" int smm_hip_set_march_min_rows(long long const_diagonals_rows, long long values_read_rows) {
	smm::g_marchMinRowsConst.store(const_diagonals_rows, std::memory_order_relaxed);
	smm::g_marchMinRowsMasks.store(values_read_rows, std::memory_order_relaxed);
	return SMM_HIP_OK;
}
