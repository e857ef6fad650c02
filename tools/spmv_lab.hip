// tools/spmv_lab.hip -- variant laboratory for the STREAM SpMV on the bench matrix (BASELINE config 3: banded-random 10M rows, fp32).
// Every variant computes the same y = A x from the reference's CSR arrays (values / positions / start, nothing re-encoded) and is
// checked against the library's kernel; the table it prints (time, algorithmic GB/s) is what profiles/r02/spmv_variants.txt holds.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -o tools/bin/spmv_lab tools/spmv_lab.hip \
//         -Lsparse_matrix_math_amd/lib -lsmm_hip -Wl,-rpath,'$ORIGIN/../../sparse_matrix_math_amd/lib'
//   tools/bin/spmv_lab [rows] [reps] [only=<substring>]
//
// Variants (template knobs):  TPB threads per workgroup, NV staging passes (tile capacity = NV * 4 * TPB nonzeros), L lanes per
// row, G gathers issued per batch, MODE:
//   0  the library's structure: next tile fetched to registers one tile ahead, loads issued BEFORE the gathers of this tile
//   1  same, but the first batch of gathers is issued before the next tile's loads (a wave's loads return in order: a gather
//      issued behind the stream loads cannot return before them)
//   2  not software-pipelined: load tile -> LDS -> gathers; overlap comes from the other workgroups of the CU only
//   3  as 2 with LDS-DMA (global_load_lds_dwordx4) instead of register staging
//   4  wave-specialised: NP producer waves stream tiles into an LDS ring by LDS-DMA, the other waves only gather and sum;
//      one s_barrier per tile
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "smm_hip.h"

#define CHECK(x)                                                                          \
	do {                                                                                  \
		hipError_t e_ = (x);                                                              \
		if (e_ != hipSuccess) {                                                           \
			std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			std::exit(1);                                                                 \
		}                                                                                 \
	} while (0)
#define SMMCHECK(x)                                                                  \
	do {                                                                             \
		int s_ = (x);                                                                \
		if (s_ != 0) {                                                               \
			std::printf("smm error %d (%s) at %s:%d\n", s_, smm_hip_last_error(), __FILE__, __LINE__); \
			std::exit(1);                                                            \
		}                                                                            \
	} while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVE = 64;

__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void compilerFence() { asm volatile("" ::: "memory"); }

extern __shared__ __attribute__((aligned(16))) unsigned char labLds[];

struct TileMap {  // XCD-aware split: workgroups b, b+8, ... share an XCD; each of the 8 groups owns one contiguous eighth of the tiles
	int nTiles, nGroups, xcdGroup, groupSlots, chunkTiles;
	__device__ TileMap(int nTiles_) : nTiles(nTiles_) {
		nGroups = min(8, static_cast<int>(gridDim.x));
		xcdGroup = blockIdx.x % nGroups;
		groupSlots = (static_cast<int>(gridDim.x) - xcdGroup + nGroups - 1) / nGroups;
		chunkTiles = (nTiles + nGroups - 1) / nGroups;
	}
	__device__ int tileOf(int j) const {  // j-th tile of this workgroup's XCD group, or nTiles past the end
		const int c = j / chunkTiles;
		if (c > 0) return nTiles;
		const long long t = static_cast<long long>(xcdGroup) * chunkTiles + j;
		return t < nTiles ? static_cast<int>(t) : nTiles;
	}
};

// ---------------------------------------------------------------------------------------------------------------------
// MODE 0..3: every wave loads and gathers
// ---------------------------------------------------------------------------------------------------------------------
template <int TPB, int NV, int L, int G, int MODE>
__global__ __launch_bounds__(TPB) void labKernel(int nTiles, const int2* __restrict__ rowBlocks, const int* __restrict__ start,
                                                 const int* __restrict__ positions, const float* __restrict__ values,
                                                 const float* __restrict__ x, float* __restrict__ out) {
	constexpr int PIECE = 4 * TPB;
	constexpr int CAP = NV * PIECE;
	constexpr int PAD = G > 16 ? G : 16;
	constexpr int RW = WAVE / L;
	constexpr int RT = MODE >= 5 ? 64 * (TPB / WAVE) / L : RW * (TPB / WAVE);
	constexpr bool RAWCOL = MODE == 3;  // LDS holds raw columns (LDS-DMA cannot scale them)
	float* sVal = reinterpret_cast<float*>(labLds);
	unsigned* sOff = reinterpret_cast<unsigned*>(sVal + CAP + PAD);
	int* sStart = reinterpret_cast<int*>(sOff + CAP + PAD);
	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int wave = t >> 6;
	// MODE 5/6: a wave gathers ONE piece of 64 consecutive rows (a gather instruction then reads one 256-byte window of x instead of
	// L windows of 256 / L bytes); the L pieces of a row sit in different waves and meet in LDS.  Rows per tile: 64 * (TPB / 64) / L.
	constexpr bool XW = MODE >= 5;
	const int rowInWave = XW ? lane : lane % RW;
	const int piece = XW ? wave % L : lane / RW;
	const int rl = XW ? (wave / L) * 64 + lane : wave * RW + rowInWave;
	float* sPart = reinterpret_cast<float*>(sStart + RT + 8);  // XW: partial sums of pieces 1 .. L-1
	for (int i = t; i < CAP + PAD; i += TPB) {
		sOff[i] = 0u;
		sVal[i] = 0.f;
	}
	const TileMap tm(nTiles);
	int j = blockIdx.x / tm.nGroups;
	int tile = tm.tileOf(j);
	i32x4 rp[NV];
	f32x4 rv[NV];
	int ps = 0;
	int2 m0 = make_int2(0, 0), m1 = make_int2(0, 0), nm0 = make_int2(0, 0), nm1 = make_int2(0, 0);
	auto stageLoad = [&](int a0, int n1) {
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			const int i = a0 + 4 * (t + v * TPB);
			if (i < n1) {
				rp[v] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(positions + i));
				rv[v] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(values + i));
			}
		}
	};
	auto stageStore = [&](int a0, int n1) {
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			const int li = 4 * (t + v * TPB);
			if (a0 + li < n1) {
				*reinterpret_cast<i32x4*>(sOff + li) = rp[v] * 4;
				*reinterpret_cast<f32x4*>(sVal + li) = rv[v];
			}
		}
	};
	if (tile < nTiles) {
		m0 = rowBlocks[tile];
		m1 = rowBlocks[tile + 1];
		const int t1 = tm.tileOf(j + tm.groupSlots);
		if (t1 < nTiles) {
			nm0 = rowBlocks[t1];
			nm1 = rowBlocks[t1 + 1];
		}
		if (MODE <= 1 || MODE == 6) {
			stageLoad(m0.y & ~3, m1.y);
			if (t < m1.x - m0.x) ps = start[m0.x + t];
		}
	}
	__syncthreads();
	while (tile < nTiles) {
		const int r0 = m0.x, n0 = m0.y, r1 = m1.x, n1 = m1.y;
		const int nrows = r1 - r0;
		const int a0 = n0 & ~3;
		if (MODE == 2 || MODE == 5) {
			stageLoad(a0, n1);
			if (t < nrows) ps = start[r0 + t];
		}
		if (MODE == 3) {
			// each wave moves 256-entry chunks (1 KiB) straight into LDS; chunk c of the tile -> sOff / sVal + 256 c
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				const int c = v * (TPB / WAVE) + wave;
				const int i = a0 + 256 * c;
				if (i < n1) {
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(positions + i + 4 * lane),
					                                 (__attribute__((address_space(3))) void*)(sOff + 256 * c), 16, 0, 2);
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(values + i + 4 * lane),
					                                 (__attribute__((address_space(3))) void*)(sVal + 256 * c), 16, 0, 2);
				}
			}
			if (t < nrows) ps = start[r0 + t];
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		} else {
			stageStore(a0, n1);
		}
		if (t < nrows) sStart[t] = ps - a0;
		if (t == 0) sStart[nrows] = n1 - a0;
		ldsBarrier();
		j += tm.groupSlots;
		const int ntile = tm.tileOf(j);
		const int2 m0n = nm0, m1n = nm1;
		auto prefetchNext = [&]() {
			if (ntile < nTiles) {
				if (MODE <= 1 || MODE == 6) {
					stageLoad(m0n.y & ~3, m1n.y);
					if (t < m1n.x - m0n.x) ps = start[m0n.x + t];
				}
				const int t2 = tm.tileOf(j + tm.groupSlots);
				if (t2 < nTiles) {
					nm0 = rowBlocks[t2];
					nm1 = rowBlocks[t2 + 1];
				}
			}
		};
		if (MODE != 1) prefetchNext();
		float dot = 0.f;
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
		int k = kb;
		if (MODE == 1) {
			// first batch: gathers go out before the next tile's stream loads
			unsigned off[G];
			float xv[G];
#pragma unroll
			for (int u = 0; u < G; ++u) off[u] = sOff[k + u];
#pragma unroll
			for (int u = 0; u < G; ++u) xv[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + off[u]);
			compilerFence();
			prefetchNext();
			compilerFence();
			const int nvalid = ke - k;
#pragma unroll
			for (int u = 0; u < G; ++u) {
				const float next = sVal[k + u] * xv[u] + dot;
				dot = u < nvalid ? next : dot;
			}
			k += G;
		}
		for (; k < ke; k += G) {
			unsigned off[G];
			float xv[G], vv[G];
#pragma unroll
			for (int u = 0; u < G; ++u) {
				off[u] = sOff[k + u];
				vv[u] = sVal[k + u];
			}
#pragma unroll
			for (int u = 0; u < G; ++u) {
				const unsigned bo = RAWCOL ? off[u] * 4u : off[u];
				xv[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + bo);
			}
			const int nvalid = ke - k;
#pragma unroll
			for (int u = 0; u < G; ++u) {
				const float next = vv[u] * xv[u] + dot;
				dot = u < nvalid ? next : dot;
			}
		}
		if (XW) {
			if (L > 1) {
				if (piece > 0) sPart[(piece - 1) * RT + rl] = dot;
				ldsBarrier();
				if (piece == 0) {
#pragma unroll
					for (int q = 1; q < L; ++q) dot += sPart[(q - 1) * RT + rl];
				}
			}
		} else if (L > 1) {
			float total = dot;
#pragma unroll
			for (int q = 1; q < L; ++q) total += __shfl(dot, rowInWave + q * RW, WAVE);
			dot = total;
		}
		if (piece == 0 && rl < nrows) __builtin_nontemporal_store(dot, out + r0 + rl);
		ldsBarrier();
		tile = ntile;
		m0 = m0n;
		m1 = m1n;
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// MODE 4: NP producer waves (LDS-DMA into a ring of NS tile slots) + NC consumer waves (gathers and row sums)
//
// On record: the FIRST run of this mode ended in "Memory access fault by GPU node-2 ... on address 0x791c6c566000" (r02,
// gpurun_out/lab1_ring.log, before the first configuration printed a line).  That source was never committed -- the lab entered git at
// 4da9802 with the guards below already in -- so the faulting line cannot be quoted from a diff; what the surviving artefacts say:
//   * the address is page-aligned and nothing was printed: a read that ran off the END of a global allocation into the next, unmapped
//     page, in the very first launch.  LDS accesses cannot raise this fault (ring slot indexing (kk % NS) * SLOT and the PAD over-read of
//     a gather batch stay inside the workgroup's LDS allocation or return zeros; they would have shown as wrong sums, and every later run
//     printed `ok`), and out[] is only written for rl < nrows.
//   * the only global accesses of this mode that can leave their buffer are the producer's LDS-DMA reads of positions[] / values[]:
//     a producer issues whole slots -- CAP entries from a0 = rowBlocks[tile].y & ~3, whatever the tile really holds -- and runs NS - 1
//     tiles AHEAD of the consumers, i.e. it issues slots for kk >= myTiles too.  For those TileMap::tileOf() returns nTiles, whose table
//     entry is the closing sentinel {rows, nnz}: a DMA of CAP (4096-8192) entries starting AT nnz, 16-32 KB past the arrays, where the
//     modes 0-3 over-read less than one 4 KB piece.  With the arrays allocated with the small slack those modes needed, that read ends in
//     an unmapped page.
// Both guards are in the committed source: issue() clamps to the workgroup's last real tile (`min(kk, myTiles - 1)`, a dummy reload
// whose slot no consumer reads) and skips everything when myTiles == 0, and main() allocates positions[] / values[] with 64 K entries
// of zero-filled slack (> CAP for every configuration: static_assert below), so even the last tile's own over-read stays mapped.
// ---------------------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void waitVm() {
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NP, int NC, int CAP, int NS, int L, int G>
__global__ __launch_bounds__((NP + NC) * WAVE) void labRingKernel(int nTiles, const int2* __restrict__ rowBlocks, const int* __restrict__ start,
                                                                  const int* __restrict__ positions, const float* __restrict__ values,
                                                                  const float* __restrict__ x, float* __restrict__ out) {
	constexpr int TPB = (NP + NC) * WAVE;
	constexpr int PAD = G > 16 ? G : 16;
	constexpr int RW = WAVE / L;
	constexpr int CHUNKS = CAP / 256;      // 1 KiB pieces per array and tile
	constexpr int IPT = 2 * CHUNKS / NP;   // LDS-DMA instructions per tile and producer wave
	static_assert(2 * CHUNKS % NP == 0, "chunks must divide over the producer waves");
	static_assert((NS - 1) * IPT <= 60, "vmcnt is a 6-bit counter");
	static_assert(CAP + 256 <= 64 * 1024, "a slot's DMA may start at the last entry: it must stay inside the slack main() allocates behind the arrays");
	// ring: slot s = sOff[CAP + PAD] then sVal[CAP]; a batch of G entries may read past the end of its piece: what it finds there must
	// be a valid column (the zeroed PAD), never value bits; one more PAD behind the last slot
	constexpr int SLOT = 2 * CAP + PAD;
	unsigned* ring = reinterpret_cast<unsigned*>(labLds);
	const int t = threadIdx.x;
	const int lane = t & (WAVE - 1);
	const int wave = t >> 6;
	for (int i = t; i < NS * SLOT + PAD; i += TPB) ring[i] = 0u;
	const TileMap tm(nTiles);
	const int j0 = blockIdx.x / tm.nGroups;
	// number of tiles of this workgroup
	int myTiles = 0;
	{
		const int first = tm.xcdGroup * tm.chunkTiles;
		const int last = min(nTiles, first + tm.chunkTiles);
		const int cnt = max(0, last - first);  // tiles of the group
		myTiles = cnt > j0 ? (cnt - j0 + tm.groupSlots - 1) / tm.groupSlots : 0;
	}
	__syncthreads();
	if (wave < NP) {
		// ---------------- producer ----------------
		auto issue = [&](int kk) {  // DMA of this workgroup's kk-th tile into slot kk % NS (a dummy reload of the last tile past the end)
			const int kc = min(kk, myTiles - 1);
			const int tile = tm.tileOf(j0 + kc * tm.groupSlots);
			const int a0 = rowBlocks[tile].y & ~3;
			unsigned* slot = ring + (kk % NS) * SLOT;
#pragma unroll
			for (int i = 0; i < IPT; ++i) {
				const int c = i * NP + wave;  // 0 .. 2 CHUNKS-1: positions chunks first, then values chunks
				const void* src = c < CHUNKS ? static_cast<const void*>(positions + a0 + 256 * c + 4 * lane)
				                             : static_cast<const void*>(values + a0 + 256 * (c - CHUNKS) + 4 * lane);
				unsigned* dst = c < CHUNKS ? slot + 256 * c : slot + CAP + PAD + 256 * (c - CHUNKS);
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
				                                 (__attribute__((address_space(3))) void*)(dst), 16, 0, 2);
			}
		};
		if (myTiles > 0) {
#pragma unroll
			for (int kk = 0; kk < NS - 1; ++kk) issue(kk);
			waitVm<(NS - 2) * IPT>();  // tile 0 has landed
		}
		ldsBarrier();
		for (int k = 0; k < myTiles; ++k) {
			issue(k + NS - 1);          // into the slot tile k-1 occupied (free since the last barrier)
			waitVm<(NS - 2) * IPT>();   // tile k+1 has landed
			ldsBarrier();
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	} else {
		// ---------------- consumers ----------------
		const int cw = wave - NP;
		const int rowInWave = lane % RW;
		const int piece = lane / RW;
		const int rl = cw * RW + rowInWave;
		int2 m0 = make_int2(0, 0), m1 = make_int2(0, 0);
		int ps = 0, pe = 0;
		if (myTiles > 0) {
			const int tile = tm.tileOf(j0);
			m0 = rowBlocks[tile];
			m1 = rowBlocks[tile + 1];
			if (rl < m1.x - m0.x) {
				ps = start[m0.x + rl];
				pe = start[m0.x + rl + 1];
			}
		}
		ldsBarrier();
		for (int k = 0; k < myTiles; ++k) {
			const int r0 = m0.x, nrows = m1.x - m0.x, a0 = m0.y & ~3;
			const unsigned* sOff = ring + (k % NS) * SLOT;
			const float* sVal = reinterpret_cast<const float*>(sOff + CAP + PAD);
			int kb = 0, ke = 0;
			if (rl < nrows) {
				const int b = ps - a0, e = pe - a0;
				kb = b;
				ke = e;
				if (L > 1) {
					const int piecelen = (e - b + L - 1) / L;
					kb = b + piece * piecelen;
					ke = min(e, kb + piecelen);
				}
			}
			// descriptors and row pointers of the next tile (short loads, issued before this tile's gathers)
			if (k + 1 < myTiles) {
				const int tile = tm.tileOf(j0 + (k + 1) * tm.groupSlots);
				m0 = rowBlocks[tile];
				m1 = rowBlocks[tile + 1];
				if (rl < m1.x - m0.x) {
					ps = start[m0.x + rl];
					pe = start[m0.x + rl + 1];
				}
			}
			float dot = 0.f;
			for (int kk = kb; kk < ke; kk += G) {
				unsigned col[G];
				float xv[G], vv[G];
#pragma unroll
				for (int u = 0; u < G; ++u) {
					col[u] = sOff[kk + u];
					vv[u] = sVal[kk + u];
				}
#pragma unroll
				for (int u = 0; u < G; ++u) xv[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + col[u] * 4u);
				const int nvalid = ke - kk;
#pragma unroll
				for (int u = 0; u < G; ++u) {
					const float next = vv[u] * xv[u] + dot;
					dot = u < nvalid ? next : dot;
				}
			}
			if (L > 1) {
				float total = dot;
#pragma unroll
				for (int q = 1; q < L; ++q) total += __shfl(dot, rowInWave + q * RW, WAVE);
				dot = total;
			}
			if (piece == 0 && rl < nrows) __builtin_nontemporal_store(dot, out + r0 + rl);
			ldsBarrier();
		}
	}
}


// ---------------------------------------------------------------------------------------------------------------------
// CEILING: the memory access stream of the SpMV with nothing else -- no LDS, no barriers, no row sums, no dependence of a gather on
// the positions it would come from.  Per tile the workgroup issues exactly the 16-byte stream loads of the STREAM kernel
// (positions[] / values[] slices, non-temporal) and exactly its x[] gather instructions (wave w, lane (row, piece): the windows
// x[row + off[k]] of its piece, one 4-byte load per lane and window -- the same cache lines in the same order from the same XCD),
// all independent, then adds everything up.  WHAT: 1 stream only, 2 gathers only, 3 both.  What this kernel reaches is what the
// memory system gives this address stream at full memory-level parallelism; the SpMV cannot be faster than it.
// ---------------------------------------------------------------------------------------------------------------------
template <int WHAT, int NOFFP>
__global__ __launch_bounds__(256) void ceilKernel(int nTiles, const int2* __restrict__ rowBlocks, const int* __restrict__ positions,
                                                  const float* __restrict__ values, const float* __restrict__ x, float* __restrict__ out,
                                                  const int* __restrict__ offs, int nOffs, int nCols) {
	const int t = threadIdx.x;
	const int lane = t & 63;
	const int wave = t >> 6;
	const int rowInWave = lane & 31;
	const int piece = lane >> 5;
	int myOff[NOFFP];
#pragma unroll
	for (int u = 0; u < NOFFP; ++u) {
		const int k = piece * NOFFP + u;
		myOff[u] = k < nOffs ? offs[k] : 0x40000000;  // a column no row has: the load is masked
	}
	const TileMap tm(nTiles);
	float acc = 0.f;
	int iacc = 0;
	for (int j = blockIdx.x / tm.nGroups;; j += tm.groupSlots) {
		const int tile = tm.tileOf(j);
		if (tile >= nTiles) break;
		const int2 m0 = rowBlocks[tile], m1 = rowBlocks[tile + 1];
		const int a0 = m0.y & ~3, n1 = m1.y, r0 = m0.x, nrows = m1.x - m0.x;
		i32x4 rp[4];
		f32x4 rv[4];
		float xv[NOFFP];
		if (WHAT & 1) {
#pragma unroll
			for (int v = 0; v < 4; ++v) {
				const int i = a0 + 4 * (t + v * 256);
				rp[v] = i32x4{0, 0, 0, 0};
				rv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
				if (i < n1) {
					rp[v] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(positions + i));
					rv[v] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(values + i));
				}
			}
		}
		if (WHAT & 2) {
			const int rl = wave * 32 + rowInWave;
			const int row = r0 + rl;
#pragma unroll
			for (int u = 0; u < NOFFP; ++u) {
				const int col = row + myOff[u];
				xv[u] = (rl < nrows && col >= 0 && col < nCols) ? x[col] : 0.f;
			}
		}
		if (WHAT & 1) {
#pragma unroll
			for (int v = 0; v < 4; ++v) {
				iacc += rp[v].x ^ rp[v].y ^ rp[v].z ^ rp[v].w;
				acc += rv[v].x + rv[v].y + rv[v].z + rv[v].w;
			}
		}
		if (WHAT & 2) {
#pragma unroll
			for (int u = 0; u < NOFFP; ++u) acc += xv[u];
		}
	}
	if (acc == 123.456f || iacc == 0x7fffffff) out[blockIdx.x] = acc;  // never true: keeps the loads alive
}

// CEILING 2: the same access stream with a per-window cache policy.  Wave w of the workgroup gathers ONE piece (w >> 1) of the rows of
// one half (w & 1) of the tile, so that a gather instruction serves a single window x[row + off[k]] and can carry that window's
// policy: bit k of ntMask set -> non-temporal load (the line is not kept in L2 for the next window that will want it).
template <int NOFFP>
__global__ __launch_bounds__(256) void ceilPolicyKernel(int nTiles, const int2* __restrict__ rowBlocks, const int* __restrict__ positions,
                                                        const float* __restrict__ values, const float* __restrict__ x, float* __restrict__ out,
                                                        const int* __restrict__ offs, int nOffs, int nCols, unsigned long long ntMask, int streamToo) {
	const int t = threadIdx.x;
	const int lane = t & 63;
	const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int half = wave & 1;
	const int piece = wave >> 1;
	const TileMap tm(nTiles);
	float acc = 0.f;
	int iacc = 0;
	for (int j = blockIdx.x / tm.nGroups;; j += tm.groupSlots) {
		const int tile = tm.tileOf(j);
		if (tile >= nTiles) break;
		const int2 m0 = rowBlocks[tile], m1 = rowBlocks[tile + 1];
		const int a0 = m0.y & ~3, n1 = m1.y, r0 = m0.x, nrows = m1.x - m0.x;
		i32x4 rp[4];
		f32x4 rv[4];
		float xv[NOFFP];
		if (streamToo) {
#pragma unroll
			for (int v = 0; v < 4; ++v) {
				const int i = a0 + 4 * (t + v * 256);
				rp[v] = i32x4{0, 0, 0, 0};
				rv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
				if (i < n1) {
					rp[v] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(positions + i));
					rv[v] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(values + i));
				}
			}
		}
		const int rl = 64 * half + lane;
		const int row = r0 + rl;
#pragma unroll
		for (int u = 0; u < NOFFP; ++u) {
			const int k = piece * NOFFP + u;  // wave-uniform
			xv[u] = 0.f;
			if (k < nOffs) {
				const int col = row + offs[k];
				const bool ok = rl < nrows && col >= 0 && col < nCols;
				if ((ntMask >> k) & 1ull) {
					if (ok) xv[u] = __builtin_nontemporal_load(x + col);
				} else {
					if (ok) xv[u] = x[col];
				}
			}
		}
		if (streamToo) {
#pragma unroll
			for (int v = 0; v < 4; ++v) {
				iacc += rp[v].x ^ rp[v].y ^ rp[v].z ^ rp[v].w;
				acc += rv[v].x + rv[v].y + rv[v].z + rv[v].w;
			}
		}
#pragma unroll
		for (int u = 0; u < NOFFP; ++u) acc += xv[u];
	}
	if (acc == 123.456f || iacc == 0x7fffffff) out[blockIdx.x] = acc;
}

// 16-byte stream load with an explicit cache policy (FLAVOR): 0 the compiler's non-temporal load, 1 plain, 2 `sc0 sc1`, 3 `sc1`,
// 4 `sc0 sc1 nt`.  The asm forms are invisible to the compiler's s_waitcnt bookkeeping: the caller waits with vmcnt(0) itself.
template <int FLAVOR, typename V>
__device__ __forceinline__ V streamLoad16(const V* p) {
	if (FLAVOR == 0) return __builtin_nontemporal_load(p);
	if (FLAVOR == 1) return *p;
	V v;
	if (FLAVOR == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
	if (FLAVOR == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
	if (FLAVOR == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
	return v;
}

// CEILING 3: stream + gathers (64-row windows, one piece per wave like ceilPolicyKernel) with the cache policy of the STREAM loads
// as the knob: does any flavour keep the once-read matrix lines from displacing x[] lines in L2?
template <int NOFFP, int FLAVOR>
__global__ __launch_bounds__(256) void ceilFlavorKernel(int nTiles, const int2* __restrict__ rowBlocks, const int* __restrict__ positions,
                                                        const float* __restrict__ values, const float* __restrict__ x, float* __restrict__ out,
                                                        const int* __restrict__ offs, int nOffs, int nCols) {
	const int t = threadIdx.x;
	const int lane = t & 63;
	const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int half = wave & 1;
	const int piece = wave >> 1;
	const TileMap tm(nTiles);
	float acc = 0.f;
	int iacc = 0;
	for (int j = blockIdx.x / tm.nGroups;; j += tm.groupSlots) {
		const int tile = tm.tileOf(j);
		if (tile >= nTiles) break;
		const int2 m0 = rowBlocks[tile], m1 = rowBlocks[tile + 1];
		const int a0 = m0.y & ~3, n1 = m1.y, r0 = m0.x, nrows = m1.x - m0.x;
		i32x4 rp[4];
		f32x4 rv[4];
		float xv[NOFFP];
#pragma unroll
		for (int v = 0; v < 4; ++v) {
			const int i = min(a0 + 4 * (t + v * 256), n1 & ~3);  // always a valid address: no divergence around the asm loads
			rp[v] = streamLoad16<FLAVOR>(reinterpret_cast<const i32x4*>(positions + i));
			rv[v] = streamLoad16<FLAVOR>(reinterpret_cast<const f32x4*>(values + i));
		}
		const int rl = 64 * half + lane;
		const int row = r0 + rl;
#pragma unroll
		for (int u = 0; u < NOFFP; ++u) {
			const int k = piece * NOFFP + u;
			xv[u] = 0.f;
			if (k < nOffs) {
				const int col = row + offs[k];
				if (rl < nrows && col >= 0 && col < nCols) xv[u] = x[col];
			}
		}
		if (FLAVOR >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
		for (int v = 0; v < 4; ++v) {
			iacc += rp[v].x ^ rp[v].y ^ rp[v].z ^ rp[v].w;
			acc += rv[v].x + rv[v].y + rv[v].z + rv[v].w;
		}
#pragma unroll
		for (int u = 0; u < NOFFP; ++u) acc += xv[u];
	}
	if (acc == 123.456f || iacc == 0x7fffffff) out[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------
struct Tiles {
	int2* d = nullptr;
	int n = 0;
};

static Tiles buildTiles(const std::vector<int>& hs, int rows, int capNnz, int maxRows) {
	std::vector<int> rb;
	int r = 0;
	while (r < rows) {
		rb.push_back(r);
		rb.push_back(hs[r]);
		const int base = hs[r];
		int e = r + 1;
		const int limitRow = std::min(rows, r + maxRows);
		while (e < limitRow && hs[e + 1] - base <= capNnz) ++e;
		if (hs[e] - base > capNnz) {
			std::printf("row %d longer than a tile (%d > %d): the lab does not handle that\n", r, hs[e] - base, capNnz);
			std::exit(1);
		}
		r = e;
	}
	rb.push_back(rows);
	rb.push_back(hs[rows]);
	Tiles t;
	t.n = static_cast<int>(rb.size() / 2) - 1;
	CHECK(hipMalloc(&t.d, rb.size() * sizeof(int)));
	CHECK(hipMemcpy(t.d, rb.data(), rb.size() * sizeof(int), hipMemcpyHostToDevice));
	return t;
}

struct Ctx {
	int rows = 0, reps = 20;
	long long nnz = 0;
	int *d_start = nullptr, *d_pos = nullptr;
	float *d_val = nullptr, *d_x = nullptr, *d_y = nullptr, *d_ref = nullptr;
	std::vector<int> hs;
	std::vector<float> href, hy;
	std::map<std::pair<int, int>, Tiles> tiles;
	double bytes = 0;
	const char* only = nullptr;
	int cus = 256;
	const Tiles& get(int capNnz, int maxRows) {
		auto key = std::make_pair(capNnz, maxRows);
		auto it = tiles.find(key);
		if (it == tiles.end()) it = tiles.emplace(key, buildTiles(hs, rows, capNnz, maxRows)).first;
		return it->second;
	}
};

template <typename K, typename... A>
static void runVariant(Ctx& c, const char* name, K kernel, int tpb, size_t lds, int wgsPerCU, const Tiles& tl, A... args) {
	if (c.only && !std::strstr(name, c.only)) return;
	hipFuncAttributes fa;
	CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)));
	if (lds > 64 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
	int occ = 0;
	CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, tpb, lds));
	const int per = wgsPerCU > 0 ? std::min(wgsPerCU, occ) : occ;
	if (per < 1) {
		std::printf("%-44s does not fit (lds %zu)\n", name, lds);
		return;
	}
	const int grid = std::min(tl.n, c.cus * per);
	CHECK(hipMemset(c.d_y, 0xFF, sizeof(float) * c.rows));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(tpb), lds, 0, tl.n, tl.d, args...);
	CHECK(hipGetLastError());
	CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(c.hy.data(), c.d_y, sizeof(float) * c.rows, hipMemcpyDeviceToHost));
	double maxrel = 0, scale = 0;
	for (int i = 0; i < c.rows; ++i) scale = std::max(scale, std::fabs(static_cast<double>(c.href[i])));
	for (int i = 0; i < c.rows; ++i) {
		const double d = std::fabs(static_cast<double>(c.hy[i]) - c.href[i]) / scale;
		if (!(d <= maxrel)) maxrel = d;  // NaN-propagating
	}
	float best = 1e30f, total = 0;
	for (int r = 0; r < c.reps; ++r) {
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(tpb), lds, 0, tl.n, tl.d, args...);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipEventSynchronize(e1));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		best = std::min(best, ms);
		total += ms;
	}
	const float avg = total / c.reps;
	std::printf("%-44s vgpr %3d lds %6zu wg/cu %d(occ %d) tiles %6d  avg %.4f ms  best %.4f ms  %7.1f GB/s  %5.1f %%  maxrel %.2e %s\n", name, fa.numRegs, lds, per, occ,
	            tl.n, avg, best, c.bytes / (avg * 1e-3) / 1e9, c.bytes / (avg * 1e-3) / 1e9 / 80.0, maxrel, maxrel < 2e-5 ? "ok" : "MISMATCH");
	std::fflush(stdout);
	CHECK(hipEventDestroy(e0));
	CHECK(hipEventDestroy(e1));
}

template <int TPB, int NV, int L, int G, int MODE>
static void runLab(Ctx& c, int wgsPerCU = 0) {
	char name[128];
	std::snprintf(name, sizeof(name), "mode%d tpb%d nv%d L%d G%d wg%d", MODE, TPB, NV, L, G, wgsPerCU);
	constexpr int CAP = NV * 4 * TPB;
	constexpr int PAD = G > 16 ? G : 16;
	constexpr int RT = MODE >= 5 ? 64 * (TPB / WAVE) / L : (WAVE / L) * (TPB / WAVE);
	const size_t lds = static_cast<size_t>(CAP + PAD) * 8 + (RT + 8) * sizeof(int) + (MODE >= 5 ? L * RT * sizeof(float) : 0) + 64;
	const Tiles& tl = c.get(CAP - 3, RT);
	runVariant(c, name, labKernel<TPB, NV, L, G, MODE>, TPB, lds, wgsPerCU, tl, c.d_start, c.d_pos, c.d_val, c.d_x, c.d_y);
}

template <int NP, int NC, int CAP, int NS, int L, int G>
static void runRing(Ctx& c, int wgsPerCU = 0) {
	char name[128];
	std::snprintf(name, sizeof(name), "ring np%d nc%d cap%d ns%d L%d G%d wg%d", NP, NC, CAP, NS, L, G, wgsPerCU);
	constexpr int PAD = G > 16 ? G : 16;
	constexpr int RT = (WAVE / L) * NC;
	const size_t lds = (static_cast<size_t>(NS) * (2 * CAP + PAD) + PAD) * 4 + 64;
	const Tiles& tl = c.get(CAP - 3, RT);
	runVariant(c, name, labRingKernel<NP, NC, CAP, NS, L, G>, (NP + NC) * WAVE, lds, wgsPerCU, tl, c.d_start, c.d_pos, c.d_val, c.d_x, c.d_y);
}


template <int WHAT>
static void runCeil(Ctx& c, int wgsPerCU = 0) {
	char name[128];
	std::snprintf(name, sizeof(name), "ceil what%d (1 stream, 2 gathers, 3 both) wg%d", WHAT, wgsPerCU);
	if (c.only && !std::strstr(name, c.only)) return;
	static int* d_offs = nullptr;
	static int nOffs = 0;
	if (!d_offs) {  // the shared offsets of the band: columns of a middle row minus the row
		const int mid = c.rows / 2;
		nOffs = c.hs[mid + 1] - c.hs[mid];
		std::vector<int> cols(nOffs);
		CHECK(hipMemcpy(cols.data(), c.d_pos + c.hs[mid], sizeof(int) * nOffs, hipMemcpyDeviceToHost));
		for (int& v : cols) v -= mid;
		CHECK(hipMalloc(&d_offs, sizeof(int) * nOffs));
		CHECK(hipMemcpy(d_offs, cols.data(), sizeof(int) * nOffs, hipMemcpyHostToDevice));
		std::printf("band offsets: %d per full row, span [%d, %d]\n", nOffs, cols.front(), cols.back());
	}
	constexpr int NOFFP = 26;
	if (nOffs > 2 * NOFFP) {
		std::printf("ceil: %d offsets do not fit 2 x %d\n", nOffs, NOFFP);
		return;
	}
	const Tiles& tl = c.get(4093, 128);  // the library's tiles for 2 lanes per row
	// the result check of runVariant does not apply (nothing is written): its MISMATCH column is meaningless for these rows
	runVariant(c, name, ceilKernel<WHAT, NOFFP>, 256, 0, wgsPerCU, tl, c.d_pos, c.d_val, c.d_x, c.d_y, d_offs, nOffs, c.rows);
}

// per-window policy sweep: windows whose line is next wanted more than `keepRows` rows later are loaded non-temporally
static void runCeilPolicy(Ctx& c, int keepRows, int wgsPerCU, int streamToo) {
	char name[128];
	std::snprintf(name, sizeof(name), "policy keep<=%d rows stream%d wg%d", keepRows, streamToo, wgsPerCU);
	if (c.only && !std::strstr(name, c.only)) return;
	static int* d_offs = nullptr;
	static std::vector<int> offs;
	if (!d_offs) {
		const int mid = c.rows / 2;
		const int nOffs = c.hs[mid + 1] - c.hs[mid];
		offs.resize(nOffs);
		CHECK(hipMemcpy(offs.data(), c.d_pos + c.hs[mid], sizeof(int) * nOffs, hipMemcpyDeviceToHost));
		for (int& v : offs) v -= mid;
		CHECK(hipMalloc(&d_offs, sizeof(int) * nOffs));
		CHECK(hipMemcpy(d_offs, offs.data(), sizeof(int) * nOffs, hipMemcpyHostToDevice));
	}
	const int nOffs = static_cast<int>(offs.size());
	constexpr int NOFFP = 26;
	if (nOffs > 2 * NOFFP) return;
	// rows are swept upwards, so x[j] is touched first through the LARGEST offset and then through each smaller one in turn: the
	// line loaded for window k is next wanted by window k-1, offs[k] - offs[k-1] rows later; window 0 is its last user
	unsigned long long mask = 0;
	int kept = 0;
	for (int k = 0; k < nOffs; ++k) {
		const long long gap = k == 0 ? (1ll << 40) : static_cast<long long>(offs[k]) - offs[k - 1];
		if (gap > keepRows) mask |= 1ull << k;
		else ++kept;
	}
	std::snprintf(name, sizeof(name), "policy keep<=%d rows (%d of %d kept) stream%d wg%d", keepRows, kept, nOffs, streamToo, wgsPerCU);
	const Tiles& tl = c.get(4093, 128);
	runVariant(c, name, ceilPolicyKernel<NOFFP>, 256, 0, wgsPerCU, tl, c.d_pos, c.d_val, c.d_x, c.d_y, d_offs, nOffs, c.rows, mask, streamToo);
}

static int* bandOffsets(Ctx& c, int* nOut) {
	static int* d_offs = nullptr;
	static int nOffs = 0;
	if (!d_offs) {
		const int mid = c.rows / 2;
		nOffs = c.hs[mid + 1] - c.hs[mid];
		std::vector<int> cols(nOffs);
		CHECK(hipMemcpy(cols.data(), c.d_pos + c.hs[mid], sizeof(int) * nOffs, hipMemcpyDeviceToHost));
		for (int& v : cols) v -= mid;
		CHECK(hipMalloc(&d_offs, sizeof(int) * nOffs));
		CHECK(hipMemcpy(d_offs, cols.data(), sizeof(int) * nOffs, hipMemcpyHostToDevice));
	}
	*nOut = nOffs;
	return d_offs;
}

template <int FLAVOR>
static void runCeilFlavor(Ctx& c, int wgsPerCU) {
	static const char* names[] = {"nt (builtin)", "plain", "sc0 sc1", "sc1", "sc0 sc1 nt"};
	char name[128];
	std::snprintf(name, sizeof(name), "flavor stream loads %s wg%d", names[FLAVOR], wgsPerCU);
	if (c.only && !std::strstr(name, c.only)) return;
	int nOffs = 0;
	int* d_offs = bandOffsets(c, &nOffs);
	if (nOffs > 52) return;
	const Tiles& tl = c.get(4093, 128);
	runVariant(c, name, ceilFlavorKernel<26, FLAVOR>, 256, 0, wgsPerCU, tl, c.d_pos, c.d_val, c.d_x, c.d_y, d_offs, nOffs, c.rows);
}

int main(int argc, char** argv) {
	Ctx c;
	c.rows = argc > 1 ? std::atoi(argv[1]) : 10000000;
	c.reps = argc > 2 ? std::atoi(argv[2]) : 20;
	for (int i = 3; i < argc; ++i) {
		if (!std::strncmp(argv[i], "only=", 5)) c.only = argv[i] + 5;
	}
	SMMCHECK(smm_hip_init(0));
	char devname[256];
	SMMCHECK(smm_hip_device_info(devname, sizeof(devname), &c.cus, nullptr));
	const int K = 25, MAXOFF = 1 << 20;
	const unsigned long long SEED = 0x5EED;
	c.nnz = smm_hip_gen_banded_nnz(c.rows, K, SEED, MAXOFF);
	const size_t slack = 64 * 1024;  // staging may run this far past the arrays (zero filled)
	CHECK(hipMalloc(&c.d_start, sizeof(int) * (c.rows + 1)));
	// LAB_MATRIX_ALLOC: memory type of the matrix stream (0 default coarse-grained, 1 fine-grained, 3 uncached): does a stream that
	// the L2 does not keep leave more room for x?
	const int matFlag = std::getenv("LAB_MATRIX_ALLOC") ? std::atoi(std::getenv("LAB_MATRIX_ALLOC")) : 0;
	if (matFlag) {
		CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&c.d_pos), sizeof(int) * (c.nnz + slack), matFlag));
		CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&c.d_val), sizeof(float) * (c.nnz + slack), matFlag));
		std::printf("matrix stream allocated with hipExtMallocWithFlags(%d)\n", matFlag);
	} else {
		CHECK(hipMalloc(&c.d_pos, sizeof(int) * (c.nnz + slack)));
		CHECK(hipMalloc(&c.d_val, sizeof(float) * (c.nnz + slack)));
	}
	CHECK(hipMemset(c.d_pos + c.nnz, 0, sizeof(int) * slack));
	CHECK(hipMemset(c.d_val + c.nnz, 0, sizeof(float) * slack));
	CHECK(hipMalloc(&c.d_x, sizeof(float) * c.rows));
	CHECK(hipMalloc(&c.d_y, sizeof(float) * c.rows));
	CHECK(hipMalloc(&c.d_ref, sizeof(float) * c.rows));
	SMMCHECK(smm_hip_gen_banded_dev_f32(c.rows, K, SEED, MAXOFF, 1.0f, c.d_start, c.d_pos, c.d_val, nullptr));
	CHECK(hipDeviceSynchronize());
	{
		std::vector<float> hx(c.rows);
		unsigned long long s = 12345;
		for (int i = 0; i < c.rows; ++i) {
			s = s * 6364136223846793005ull + 1442695040888963407ull;
			hx[i] = 0.5f + static_cast<float>((s >> 40) & 0xFFFF) / 65536.0f;
		}
		CHECK(hipMemcpy(c.d_x, hx.data(), sizeof(float) * c.rows, hipMemcpyHostToDevice));
	}
	c.hs.resize(c.rows + 1);
	CHECK(hipMemcpy(c.hs.data(), c.d_start, sizeof(int) * (c.rows + 1), hipMemcpyDeviceToHost));
	c.bytes = static_cast<double>(c.nnz) * 8 + (c.rows + 1) * 4.0 + c.rows * 4.0 + c.rows * 4.0;
	std::printf("%s: banded rows %d nnz %lld (%.1f/row) f32, B_spmv = %.3f GB, reps %d\n", devname, c.rows, c.nnz, static_cast<double>(c.nnz) / c.rows, c.bytes / 1e9, c.reps);
	// reference: the library's kernel
	smm_hip_csr* A = nullptr;
	SMMCHECK(smm_hip_csr_create_dev_f32(c.rows, c.rows, c.d_start, c.d_pos, c.d_val, &A));
	SMMCHECK(smm_hip_spmv_dev_f32(A, SMM_OP_ASSIGN, nullptr, c.d_x, c.d_ref, nullptr));
	CHECK(hipDeviceSynchronize());
	c.href.resize(c.rows);
	c.hy.resize(c.rows);
	CHECK(hipMemcpy(c.href.data(), c.d_ref, sizeof(float) * c.rows, hipMemcpyDeviceToHost));
	if (!c.only || std::strstr("library", c.only)) {
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0));
		CHECK(hipEventCreate(&e1));
		float total = 0;
		for (int r = 0; r < c.reps; ++r) {
			CHECK(hipEventRecord(e0, 0));
			SMMCHECK(smm_hip_spmv_dev_f32(A, SMM_OP_ASSIGN, nullptr, c.d_x, c.d_ref, nullptr));
			CHECK(hipEventRecord(e1, 0));
			CHECK(hipEventSynchronize(e1));
			float ms = 0;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			total += ms;
		}
		int fam = 0, lanes = 0;
		smm_hip_csr_get_kernel(A, &fam, &lanes);
		std::printf("%-44s avg %.4f ms  %7.1f GB/s  %5.1f %%\n", (std::string("library family ") + std::to_string(fam) + " lanes " + std::to_string(lanes)).c_str(), total / c.reps,
		            c.bytes / (total / c.reps * 1e-3) / 1e9, c.bytes / (total / c.reps * 1e-3) / 1e9 / 80.0);
	}

#ifndef LAB_SET
#define LAB_SET 1
#endif
#if LAB_SET == 1
	runLab<256, 4, 2, 8, 0>(c);
	runLab<256, 4, 2, 8, 2>(c);
	// pieces in different waves: 64-row gathers
	runLab<256, 6, 2, 13, 5>(c);
	runLab<256, 6, 2, 13, 5>(c, 2);
	runLab<256, 7, 2, 13, 5>(c);
	runLab<256, 5, 2, 13, 5>(c);
	runLab<256, 4, 2, 13, 5>(c);
	runLab<256, 6, 2, 7, 5>(c);
	runLab<256, 6, 2, 9, 5>(c);
	runLab<256, 6, 2, 10, 5>(c);
	runLab<256, 6, 2, 14, 5>(c);
	runLab<256, 6, 2, 16, 5>(c);
	runLab<384, 5, 3, 17, 5>(c);
	runLab<384, 5, 3, 9, 5>(c);
	runLab<512, 4, 4, 13, 5>(c);
	runLab<512, 4, 4, 7, 5>(c);
	runLab<512, 3, 4, 13, 5>(c);
	runLab<256, 4, 4, 13, 5>(c);
	runLab<256, 4, 4, 7, 5>(c);
#ifdef LAB_MORE
	runLab<256, 6, 2, 8, 5>(c);
	runLab<256, 6, 2, 26, 5>(c);
	runLab<256, 6, 2, 8, 6>(c);
	runLab<256, 6, 2, 13, 6>(c);
	runLab<128, 7, 2, 13, 5>(c);
	runLab<128, 7, 2, 26, 5>(c);
	runLab<512, 6, 2, 13, 5>(c);
	runLab<512, 6, 4, 13, 5>(c);
#endif
#ifdef LAB_FULL  // measured in profiles/r02/spmv_variants.txt; every one of them lands between 0.82 and 1.02 ms
	// --- the library's structure re-created, then one knob at a time ---
	runLab<256, 4, 2, 8, 0>(c);
	runLab<256, 4, 2, 16, 0>(c);
	runLab<256, 4, 2, 32, 0>(c);
	runLab<256, 4, 4, 16, 0>(c);
	runLab<256, 4, 2, 32, 1>(c);
	runLab<256, 4, 2, 16, 1>(c);
	runLab<256, 4, 4, 16, 1>(c);
	runLab<256, 3, 4, 16, 1>(c);
	runLab<256, 3, 4, 16, 0>(c);
	runLab<256, 2, 4, 16, 1>(c);
	runLab<256, 2, 8, 8, 1>(c);
	runLab<512, 4, 2, 16, 0>(c);
	runLab<512, 4, 2, 32, 1>(c);
	runLab<512, 2, 4, 16, 1>(c);
	runLab<512, 2, 2, 32, 1>(c);
	// --- not pipelined ---
	runLab<256, 4, 2, 8, 2>(c);
	runLab<256, 4, 2, 32, 2>(c);
	runLab<256, 2, 4, 16, 2>(c);
	runLab<256, 2, 2, 32, 2>(c);
	runLab<256, 4, 2, 32, 3>(c);
	runLab<256, 2, 4, 16, 3>(c);
	runLab<256, 2, 2, 32, 3>(c);
	runLab<256, 1, 4, 16, 3>(c);
	runLab<512, 2, 4, 16, 3>(c);
#endif
	// --- the access stream alone (ceiling) ---
	runCeil<1>(c);
	runCeil<2>(c);
	runCeil<3>(c);
	runCeil<3>(c, 4);
	runCeil<3>(c, 2);
	// --- cache policy of the matrix stream beside the gathers ---
	runCeilFlavor<0>(c, 4);
	runCeilFlavor<1>(c, 4);
	runCeilFlavor<2>(c, 4);
	runCeilFlavor<3>(c, 4);
	runCeilFlavor<4>(c, 4);
	// --- per-window cache policy on the access stream ---
	for (int wg : {0, 4, 2}) {
		for (int keep : {1 << 30, 0, 4000, 12000, 20000, 31000, 40000, 50000, 60000, 80000}) runCeilPolicy(c, keep, wg, 1);
	}
	for (int keep : {1 << 30, 0, 20000, 40000}) runCeilPolicy(c, keep, 0, 0);
#ifdef LAB_FULL
	// --- wave-specialised ring ---
	runRing<1, 7, 4096, 2, 4, 16>(c);
	runRing<1, 7, 4096, 2, 2, 32>(c);
	runRing<1, 3, 2048, 2, 4, 16>(c);
	runRing<1, 3, 2048, 3, 4, 16>(c);
	runRing<1, 3, 4096, 2, 2, 32>(c);
	runRing<2, 14, 4096, 4, 4, 16>(c);
	runRing<2, 14, 4096, 4, 8, 8>(c);
	runRing<2, 6, 4096, 3, 4, 16>(c);
	runRing<1, 7, 2048, 4, 4, 16>(c);
	runRing<2, 14, 8192, 2, 4, 16>(c);
#endif
#endif
	smm_hip_csr_destroy(A);
	return 0;
}
