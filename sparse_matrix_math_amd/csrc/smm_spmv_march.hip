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

#include "smm_device.h"
#include "smm_internal.h"

namespace smm {

namespace {

constexpr int TPB = 256;
constexpr int MARCH_R = 8;              // rows per lane and plane
constexpr int MARCH_B = TPB * MARCH_R;  // rows of a tile
constexpr int MARCH_HP = 4;             // halo packs a lane can hold: 2 H <= MARCH_HP * TPB * VEC

template <typename T>
struct MarchCfg {
	static constexpr int VEC = 16 / sizeof(T);      // rows per 16-byte pack
	static constexpr int PACKS = MARCH_R / VEC;     // packs per lane and plane
	static constexpr int MH = VEC / 2;              // 16-byte packs of two 64-bit masks per row pack
	static constexpr int MPACKS = PACKS * MH;       // (4 for fp32 and fp64 alike: 8 masks per lane and plane)
};

// 16-byte packs at ELEMENT alignment (a plane need not start on a 16-byte boundary; gfx950 global accesses may be unaligned): vector
// types, so that the non-temporal builtins take them, with the alignment lowered through the typedef
template <typename T>
struct PackOf;
template <>
struct PackOf<float> {
	typedef float V __attribute__((ext_vector_type(4)));
	typedef V U __attribute__((aligned(4)));
};
template <>
struct PackOf<double> {
	typedef double V __attribute__((ext_vector_type(2)));
	typedef V U __attribute__((aligned(8)));
};
template <typename T>
using PackU = typename PackOf<T>::U;
typedef unsigned long long MaskV __attribute__((ext_vector_type(2)));
typedef MaskV MaskU __attribute__((aligned(8)));

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

}  // namespace

extern __shared__ __attribute__((aligned(16))) unsigned char smmMarchLds[];

// KN > 0: the number of near offsets is known at compile time (5: the 5- and 7-point stencils; 3: 1-D chains inside a plane); 0: run time.
// grid <= NPART persistent workgroups; unit u = (tile, z-chunk); XCD group g = blockIdx % 8 owns the tiles [g nT / 8, (g + 1) nT / 8) when
// xcdTiles (nT a multiple of 8: neighbouring tiles, which share their halos, then run on one L2), else the units are dealt round-robin.
template <typename T, int KN, bool NT>
__global__ __launch_bounds__(TPB) void spmvPatternConstMarchKernel(int rows, int cols, int P, int nPlanes, int nT, int zc, int nChunks, int xcdTiles, int H,
                                                                   int nOff, int hasLo, int hasHi, const int* __restrict__ offs,
                                                                   const unsigned long long* __restrict__ cvalBits,
                                                                   const unsigned long long* __restrict__ masks, int opFlags, const T* lhs,
                                                                   const T* __restrict__ divisor, const T* __restrict__ x, T* out, int dotMode,
                                                                   const T* __restrict__ w1, T* __restrict__ partials, const int* __restrict__ doneFlag) {
	using Cfg = MarchCfg<T>;
	constexpr int VEC = Cfg::VEC;
	constexpr int PACKS = Cfg::PACKS;
	constexpr int MPACKS = Cfg::MPACKS;
	constexpr int MH = Cfg::MH;
	const int winLen = MARCH_B + 2 * H;  // elements of one window buffer (H is a multiple of VEC)
	T* sWin0 = reinterpret_cast<T*>(smmMarchLds);
	T* sWin1 = sWin0 + winLen;
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	const int op = opFlags & 0xFF;
	const int t = threadIdx.x;
	const int nNear = KN > 0 ? KN : nOff - hasLo - hasHi;
	const T cLo = hasLo ? bitsToValue<T>(cvalBits[0]) : T(0);
	const T cHi = hasHi ? bitsToValue<T>(cvalBits[nOff - 1]) : T(0);
	const int haloPacks = 2 * H / VEC;

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
		// chunk-major inside the group: neighbouring tiles of one z-chunk are in flight together
		const int chunk = u / tCount;
		const int tile = tLo + (u - chunk * tCount);
		const int z0 = chunk * zc;
		const int z1 = min(nPlanes, z0 + zc);
		const int r0 = tile * MARCH_B;           // first row of the tile inside its plane
		const int bAct = min(MARCH_B, P - r0);   // rows of this tile (P and MARCH_B are multiples of VEC)
		bool act[PACKS];
		int loc[PACKS];                          // first local row of the lane's pack p
#pragma unroll
		for (int p = 0; p < PACKS; ++p) {
			loc[p] = (p * TPB + t) * VEC;
			act[p] = loc[p] < bAct;
		}
		PackU<T> xp[PACKS], xc[PACKS], xn[PACKS], xnn[PACKS];
		PackU<T> hh[MARCH_HP];
		MaskU mk[MPACKS], mkNext[MPACKS];
		auto loadCentre = [&](PackU<T>(&dst)[PACKS], int z) {
			const bool ok = z >= 0 && z < nPlanes;
			const long long base = static_cast<long long>(z) * P + r0;
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				if (ok && act[p]) {
					dst[p] = *reinterpret_cast<const PackU<T>*>(x + base + loc[p]);  // (kept cacheable: a neighbouring tile reads these lines as its halo)
				} else {
#pragma unroll
					for (int e = 0; e < VEC; ++e) dst[p][e] = T(0);
				}
			}
		};
		// halo pack i of plane z: window index w = i VEC for the left halo (i < H / VEC), H + bAct + (i - H / VEC) VEC for the right one
		auto haloWin = [&](int i) { return i < H / VEC ? i * VEC : H + bAct + (i - H / VEC) * VEC; };
		auto loadHalo = [&](int z) {
			const long long base = static_cast<long long>(z) * P + r0 - H;
#pragma unroll
			for (int k = 0; k < MARCH_HP; ++k) {
				const int i = k * TPB + t;
				if (i < haloPacks) {
					long long gidx = base + haloWin(i);
					// a pack that sticks out of x (the first / last rows of the matrix) is never used by a live entry: any valid address will do
					gidx = gidx < 0 ? 0 : (gidx + VEC > cols ? cols - VEC : gidx);
					hh[k] = *reinterpret_cast<const PackU<T>*>(x + gidx);
				}
			}
		};
		// the masks of the lane's own rows: row pack p = mask packs p MH .. p MH + MH - 1 (two rows each)
		auto loadMasks = [&](MaskU(&dst)[MPACKS], int z) {
			const long long base = static_cast<long long>(z) * P + r0;
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
#pragma unroll
				for (int h = 0; h < MH; ++h) {
					if (act[p]) {
						dst[p * MH + h] = __builtin_nontemporal_load(reinterpret_cast<const MaskU*>(masks + base + loc[p] + 2 * h));
					} else {
						dst[p * MH + h] = MaskU{0ULL, 0ULL};
					}
				}
			}
		};
		auto storeWindow = [&](T* win, const PackU<T>(&centre)[PACKS]) {
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				if (act[p]) *reinterpret_cast<PackU<T>*>(win + H + loc[p]) = centre[p];
			}
#pragma unroll
			for (int k = 0; k < MARCH_HP; ++k) {
				const int i = k * TPB + t;
				if (i < haloPacks) *reinterpret_cast<PackU<T>*>(win + haloWin(i)) = hh[k];
			}
		};

		// prologue: planes z0 - 1 (far below), z0 (window), z0 + 1 in registers; z0 + 2 is requested inside the first step
		loadCentre(xp, hasLo ? z0 - 1 : -1);
		loadCentre(xc, z0);
		loadCentre(xn, z0 + 1);
		loadHalo(z0);
		loadMasks(mk, z0);
		__syncthreads();  // (the previous unit's last window reads are over)
		storeWindow(sWin0, xc);
		__syncthreads();

		for (int z = z0; z < z1; ++z) {
			T* win = ((z - z0) & 1) ? sWin1 : sWin0;
			T* winNext = ((z - z0) & 1) ? sWin0 : sWin1;
			const bool more = z + 1 < z1;
			// requests for the next steps, issued before this plane's arithmetic
			if (more) {
				loadCentre(xnn, hasHi || z + 2 < z1 ? z + 2 : -1);
				loadHalo(z + 1);
				loadMasks(mkNext, z + 1);
			}
			const long long base = static_cast<long long>(z) * P + r0;
			T dot[PACKS][VEC];
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
#pragma unroll
				for (int e = 0; e < VEC; ++e) dot[p][e] = T(0);
			}
			// the rows' masks as 32 bits (CONST: at most 32 offsets)
			auto bitOn = [&](unsigned m, int b) { return ((m >> b) & 1u) != 0u; };
			unsigned m32[PACKS][VEC];
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
#pragma unroll
				for (int e = 0; e < VEC; ++e) m32[p][e] = static_cast<unsigned>(mk[p * MH + e / 2][e & 1]);
			}
			int bit = 0;
			if (hasLo) {
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
#pragma unroll
					for (int e = 0; e < VEC; ++e) {
						const T next = smmFma(cLo, xp[p][e], dot[p][e]);
						dot[p][e] = bitOn(m32[p][e], 0) ? next : dot[p][e];
					}
				}
				bit = 1;
			}
			auto nearStep = [&](int j) {
				const int off = offs[hasLo + j];
				const T c = bitsToValue<T>(cvalBits[hasLo + j]);
				const int b = bit + j;
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
					T xv[VEC];
#pragma unroll
					for (int e = 0; e < VEC; ++e) xv[e] = win[H + loc[p] + e + off];
#pragma unroll
					for (int e = 0; e < VEC; ++e) {
						const T next = smmFma(c, xv[e], dot[p][e]);
						dot[p][e] = bitOn(m32[p][e], b) ? next : dot[p][e];
					}
				}
			};
			if constexpr (KN > 0) {
#pragma unroll
				for (int j = 0; j < KN; ++j) nearStep(j);
			} else {
				for (int j = 0; j < nNear; ++j) nearStep(j);
			}
			if (hasHi) {
				const int b = bit + nNear;
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
#pragma unroll
					for (int e = 0; e < VEC; ++e) {
						const T next = smmFma(cHi, xn[p][e], dot[p][e]);
						dot[p][e] = bitOn(m32[p][e], b) ? next : dot[p][e];
					}
				}
			}
			// epilogue: op(lhs, dot), out[] as 16-byte packs, the fused dot products
#pragma unroll
			for (int p = 0; p < PACKS; ++p) {
				if (act[p]) {
					const long long row = base + loc[p];
					PackU<T> o;
#pragma unroll
					for (int e = 0; e < VEC; ++e) o[e] = marchApplyOp(op, lhs, divisor, row + e, dot[p][e]);
					// (a run-time `if (ntOut) nt-store else store` does not survive the compiler: the hint is metadata and the two arms are merged
					// into ONE plain store -- which is what happens in the older kernels; here the policy is a template argument)
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
				storeWindow(winNext, xn);
#pragma unroll
				for (int p = 0; p < PACKS; ++p) {
					xp[p] = xc[p];
					xc[p] = xn[p];
					xn[p] = xnn[p];
				}
#pragma unroll
				for (int q = 0; q < MPACKS; ++q) mk[q] = mkNext[q];
				ldsBarrier();  // winNext is complete; everyone is done reading `win` (it is overwritten in the step after next)
			}
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

// ---- the plan (once per matrix, at the end of the CONST analysis; caller holds tileMutex) and the launch ----------------------------------
// From the sorted offset list: the far pair is -P / +P with P the largest |offset|, present on at least one side, when the rest of the
// offsets is near (|off| <= the halo a lane can hold), planes are at least four tiles large and the rows are a whole number of planes
// (grids in natural order: P = nx ny); otherwise, when EVERY offset is near (2-D grids, narrow bands), the matrix is one plane.  A matrix
// that is neither keeps the gather kernel (spmvPatternConstKernel).
void planConstMarch(smm_hip_csr* m) {
	m->march_ok = false;
	static const long long minRows = [] {
		const char* env = getenv("SMM_HIP_MARCH_MIN_ROWS");
		return env ? atoll(env) : (1LL << 21);  // below that an SpMV of this encoding is a handful of microseconds: launch-bound either way
	}();
	const std::vector<int>& offs = m->pat_offs_host;
	const int k = static_cast<int>(offs.size());
	if (!m->pat_const || k < 1 || k > 32 || m->rows != m->cols || m->rows < minRows) return;
	const int vec = m->dtype == SMM_DTYPE_F32 ? 4 : 2;
	const int hCap = MARCH_HP * TPB * vec / 2;  // 2 H / VEC halo packs <= MARCH_HP per lane
	auto roundUp = [vec](int h) { return (h + vec - 1) / vec * vec; };
	const int far = std::max(std::abs(offs.front()), std::abs(offs.back()));
	// (a) the far pair
	if (far >= 4 * MARCH_B && m->rows % far == 0 && far % vec == 0 && m->rows / far >= 2) {
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
	// (b) one plane: every offset is near
	if (roundUp(std::max(far, 1)) <= hCap && m->rows % vec == 0) {
		m->march_ok = true;
		m->march_P = m->rows;
		m->march_H = roundUp(std::max(far, 1));
		m->march_lo = m->march_hi = 0;
	}
}

template <typename T, int KN, bool NT>
static int launchMarchKN(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s) {
	const int P = m->march_P, H = m->march_H;
	const int nPlanes = m->rows / P;
	const int nT = (P + MARCH_B - 1) / MARCH_B;
	const size_t lds = 2 * static_cast<size_t>(MARCH_B + 2 * H) * sizeof(T);
	static bool raised = false;
	if (lds > 64 * 1024 && !raised) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(spmvPatternConstMarchKernel<T, KN, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
		raised = true;
	}
	int perCU = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, spmvPatternConstMarchKernel<T, KN, NT>, TPB, lds) != hipSuccess || perCU < 1) perCU = 2;
	if (const char* env = getenv("SMM_HIP_MARCH_WGS_PER_CU")) perCU = std::max(1, atoi(env));
	const int cus = (op & SPMV_LEAVE_ROOM) ? std::max(8, numCUs() - 8) : numCUs();  // room for the RCCL kernel beside A_loc (smm_dist.hip)
	op &= ~SPMV_LEAVE_ROOM;
	const int resident = cus * perCU;
	// planes per unit: ~4 units per resident workgroup (the tail of the launch stays short), at least 8 planes (the two extra planes a
	// unit loads for its first and last plane's far entries then cost <= 25 %)
	int zc = nPlanes;
	if (nPlanes > 1) {
		const int wantChunks = std::max(1, std::min(nPlanes, (4 * resident + nT - 1) / nT));
		zc = std::max(std::min(8, nPlanes), (nPlanes + wantChunks - 1) / wantChunks);
		if (const char* env = getenv("SMM_HIP_MARCH_ZC")) zc = std::max(1, std::min(nPlanes, atoi(env)));
	}
	const int nChunks = (nPlanes + zc - 1) / zc;
	const long long units = static_cast<long long>(nT) * nChunks;
	int grid = static_cast<int>(std::max<long long>(1, std::min<long long>(std::min<long long>(units, resident), NPART)));
	const int xcdTiles = (nT % 8 == 0 || nT >= 64) && grid >= 8 ? 1 : 0;
	if (xcdTiles) grid -= grid % 8;  // the same number of workgroups in every XCD group
	spmvPatternConstMarchKernel<T, KN, NT><<<grid, TPB, lds, s>>>(m->rows, m->cols, P, nPlanes, nT, zc, nChunks, xcdTiles, H, m->pat_k, m->march_lo, m->march_hi,
	                                                             m->d_pat_off, m->d_pat_cval, m->d_pat_masks, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag);
	return SMM_HIP_OK;
}

// true: the launch went to the march kernel.  SMM_HIP_CONST_MARCH=0 keeps the gather kernel (A/B measurements).
template <typename T>
bool launchPatConstMarch(const smm_hip_csr* m, int op, const T* lhs, const T* divisor, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s) {
	static const bool enabled = [] {
		const char* env = getenv("SMM_HIP_CONST_MARCH");
		return env ? atoi(env) != 0 : true;
	}();
	if (!enabled || !m->march_ok) return false;
	const int nNear = m->pat_k - m->march_lo - m->march_hi;
	const bool nt = (spmvOutFlags(m, sizeof(T)) & SPMV_NT_OUT) != 0;  // outputs too large to still be cached when the next kernel reads them
#define SMM_MARCH_GO(KNV)                                                                                        \
	do {                                                                                                         \
		if (nt) launchMarchKN<T, KNV, true>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s);    \
		else launchMarchKN<T, KNV, false>(m, op, lhs, divisor, x, out, dotMode, w1, partials, doneFlag, s);      \
	} while (0)
	if (nNear == 5) SMM_MARCH_GO(5);
	else if (nNear == 3) SMM_MARCH_GO(3);
	else SMM_MARCH_GO(0);
#undef SMM_MARCH_GO
	return true;
}

template bool launchPatConstMarch<float>(const smm_hip_csr*, int, const float*, const float*, const float*, float*, int, const float*, float*, const int*, hipStream_t);
template bool launchPatConstMarch<double>(const smm_hip_csr*, int, const double*, const double*, const double*, double*, int, const double*, double*, const int*, hipStream_t);

}  // namespace smm
