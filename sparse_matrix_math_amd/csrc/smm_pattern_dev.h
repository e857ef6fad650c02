// smm_pattern_dev.h -- device-side pieces shared by the PATTERN family's kernels (smm_spmv_pattern.hip) and the one-launch form of the
// row-partitioned SpMV (smm_spmv_split.hip): the staging geometry, the row operation, the k-th set bit of a row mask.
#pragma once
#include <hip/hip_runtime.h>

#include "smm_device.h"
#include "smm_internal.h"

namespace smm {

constexpr int TPB = 256;
constexpr int MAXOFF = 64;

typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef double pf64x2 __attribute__((ext_vector_type(2)));

template <typename T>
struct PatCfg {
	static constexpr int PIECE = 4 * TPB;                  // values staged per pass: one 16-byte load per lane (fp32)
	static constexpr int NVMAX = sizeof(T) == 4 ? 8 : 4;   // passes held in registers one tile ahead (32 VGPRs)
	static constexpr int PAD = 16;
};

template <typename T>
__device__ __forceinline__ T patApplyOp(int op, const T* __restrict__ lhs, const T* __restrict__ divisor, int row, T dot) {
	if (op == SMM_OP_ASSIGN) return dot;
	if (op == SPMV_OP_DIV) return dot / divisor[row];  // the Jacobi apply folded into the row (smm_spmv.hip, applyOp)
	const T l = lhs[row];
	if (op == SPMV_OP_ADD_DIV) return (l + dot) / divisor[row];
	return op == SMM_OP_ADD ? l + dot : l - dot;
}

template <typename T>
__device__ __forceinline__ T patGather(const T* __restrict__ x, unsigned byteOffset) {
	return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(x) + byteOffset);
}

// index of the k-th (0-based) set bit of m; k < popcount(m)
__device__ __forceinline__ int selectBit(unsigned long long m, int k) {
	int pos = 0;
	unsigned v = static_cast<unsigned>(m);
	int c = __popc(v);
	if (k >= c) {
		k -= c;
		pos = 32;
		v = static_cast<unsigned>(m >> 32);
	}
	c = __popc(v & 0xFFFFu);
	if (k >= c) { k -= c; pos += 16; v >>= 16; }
	c = __popc(v & 0xFFu);
	if (k >= c) { k -= c; pos += 8; v >>= 8; }
	c = __popc(v & 0xFu);
	if (k >= c) { k -= c; pos += 4; v >>= 4; }
	c = __popc(v & 0x3u);
	if (k >= c) { k -= c; pos += 2; v >>= 2; }
	if (k >= static_cast<int>(v & 1u)) pos += 1;
	return pos;
}

// lane 0's value of a 64-bit quantity, as a wave-uniform (scalar) number
__device__ __forceinline__ unsigned long long patUniform64(unsigned long long v) {
	const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
	const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
	return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// A row straight from HBM (tiles that are not staged: the last ones of a matrix, rows longer than a tile), in the PIECE structure of the
// staged path: L chains over ceil(len / L) consecutive entries each, added left to right -- the bits of the tile kernels' lanes whatever
// the tile cut (r06: r02-r05 walked such rows as ONE chain, so the last ~cap entries of a matrix had the one-lane bits at every L, and two
// tile cuts of one matrix could differ there).
template <typename T, int L>
__device__ __forceinline__ T patRowDirect(int b, int e, const T* __restrict__ values, const int* __restrict__ positions, const T* __restrict__ x) {
	const int piecelen = (e - b + L - 1) / L;
	T tot = T(0);
#pragma unroll
	for (int q = 0; q < L; ++q) {
		const int kb = b + q * piecelen, ke = min(e, kb + piecelen);
		T dot = T(0);
		for (int k = kb; k < ke; ++k) dot = smmFma(values[k], x[positions[k]], dot);
		tot = q == 0 ? dot : tot + dot;
	}
	return tot;
}

}  // namespace smm
