// smm_spmv.hip -- CSR SpMV for gfx950:  out[i] = op(lhs[i], sum_k values[k] * x[positions[k]])
//
// Replaces CSRMatrix<T>::rMultOp / rMult / rMultAdd / rMultSub (ref:1458-1515).  Two kernel families:
//
//  VECTOR  L adjacent lanes of a wavefront own one row (L = 1..64); the lanes stride over the row so that
//          positions[]/values[] are read coalesced, partial sums meet in a wave butterfly (__shfl_xor).
//          Rows are grid-strided.  Best for long rows and for tiny matrices.
//
//  STREAM  The matrix is cut once (at csr_create) into row blocks of <= CAP nonzeros and <= 256 rows.  A
//          workgroup stages one block's positions[] and values[] into LDS with 16-byte fully coalesced loads
//          -- the two arrays that are >95 % of the bytes of an SpMV are streamed exactly once at full width
//          whatever the row lengths are -- then every row is summed from LDS by L lanes, each walking a
//          contiguous piece of the row left to right.  Lanes are mapped row-fastest (thread t -> row t % R,
//          piece t / R) so that at every step adjacent lanes read x[] at adjacent columns for banded /
//          stencil matrices: the x gather is coalesced even though x itself is never staged.
//          With L == 1 the sum is formed in exactly the reference's order (ref:1484-1489): bit-identical
//          results.  With L > 1 the L pieces are added left to right.
//
// Both families can fuse one or two dot products of the freshly computed out[] into the epilogue (the p.Ap of
// CG ref:2354, ap.r0 / as.as / as.s of BiCGStab ref:2243, 2259-2261) so those vectors are not re-read: each
// workgroup writes its partial sums to partials[blockIdx.x] (fixed grid of NPART workgroups, fixed order).
//
// No MFMA: 2 flops per 8-12 bytes, the path is HBM-bound.
#include <algorithm>

#include "smm_device.h"
#include "smm_internal.h"

namespace smm {

constexpr int TPB = 256;  // threads per workgroup = 4 wavefronts

template <typename T>
__device__ __forceinline__ T applyOp(int op, const T* __restrict__ lhs, int row, T dot) {
	if (op == SMM_OP_ASSIGN) return dot;
	const T l = lhs[row];
	return op == SMM_OP_ADD ? l + dot : l - dot;
}

// ---------------------------------------------------------------------------------------------------------
// VECTOR family
// ---------------------------------------------------------------------------------------------------------
template <typename T, int L>
__global__ __launch_bounds__(TPB) void spmvVectorKernel(int rows, const int* __restrict__ start, const int* __restrict__ positions,
                                                        const T* __restrict__ values, int op, const T* lhs, const T* __restrict__ x, T* out,
                                                        int dotMode, const T* __restrict__ w1, T* __restrict__ partials,
                                                        const int* __restrict__ doneFlag) {
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
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
			const T o = applyOp(op, lhs, static_cast<int>(row), dot);
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
	}
}

// ---------------------------------------------------------------------------------------------------------
// STREAM family
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct StreamCfg {
	// nonzeros staged per row block: 16 KiB of LDS for the two arrays (fp32) / 24 KiB (fp64) -> 8 / 6 workgroups
	// per CU, i.e. every CU keeps > 100 KiB of coalesced loads in flight while other workgroups compute
	static constexpr int CAP = 2048;
};

// element index i of the staged block -> LDS slot.  A skew of one slot per 32 keeps power-of-two row lengths
// off a single bank when adjacent lanes walk adjacent rows.
__device__ __forceinline__ int ldsSlot(int i) { return i + (i >> 5); }

template <typename T, int L>
__global__ __launch_bounds__(TPB) void spmvStreamKernel(int nBlocks, const int* __restrict__ rowBlocks, const int* __restrict__ start,
                                                        const int* __restrict__ positions, const T* __restrict__ values, int op, const T* lhs,
                                                        const T* __restrict__ x, T* out, int dotMode, const T* __restrict__ w1,
                                                        T* __restrict__ partials, const int* __restrict__ doneFlag) {
	constexpr int CAP = StreamCfg<T>::CAP;
	constexpr int SLOTS = CAP + 4 + ((CAP + 4) >> 5) + 1;
	__shared__ int sPos[SLOTS];
	__shared__ T sVal[SLOTS];
	__shared__ T sPart[TPB];
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;

	const int t = threadIdx.x;
	constexpr int R = TPB / L;  // rows summed per pass
	const int rowLocal = t % R;
	const int piece = t / R;
	T acc0 = T(0), acc1 = T(0);

	for (int blk = blockIdx.x; blk < nBlocks; blk += gridDim.x) {
		const int r0 = rowBlocks[blk];
		const int r1 = rowBlocks[blk + 1];
		const int n0 = start[r0];
		const int n1 = start[r1];
		if (n1 - n0 > CAP) {
			// a single row longer than the LDS block (r1 == r0 + 1 by construction): stream it straight from HBM
			T dot = T(0);
			for (int k = n0 + t; k < n1; k += TPB) {
				dot = smmFma(values[k], x[positions[k]], dot);
			}
			dot = blockSum256(dot, red);
			if (t == 0) {
				const T o = applyOp(op, lhs, r0, dot);
				out[r0] = o;
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[r0];
			}
			continue;
		}
		// ---- stage positions[n0,n1) and values[n0,n1) into LDS, 16 bytes per lane per load ----
		const int a0 = n0 & ~3;  // 16-byte aligned element index for both arrays (fp64: 32-byte aligned)
		for (int i = a0 + 4 * t; i < n1; i += 4 * TPB) {
			const int li = i - a0;
			if (i + 3 < n1) {
				const int4 p = *reinterpret_cast<const int4*>(positions + i);
				sPos[ldsSlot(li)] = p.x;
				sPos[ldsSlot(li + 1)] = p.y;
				sPos[ldsSlot(li + 2)] = p.z;
				sPos[ldsSlot(li + 3)] = p.w;
				if constexpr (sizeof(T) == 4) {
					const float4 v = *reinterpret_cast<const float4*>(values + i);
					sVal[ldsSlot(li)] = v.x;
					sVal[ldsSlot(li + 1)] = v.y;
					sVal[ldsSlot(li + 2)] = v.z;
					sVal[ldsSlot(li + 3)] = v.w;
				} else {
					const double2 v0 = *reinterpret_cast<const double2*>(values + i);
					const double2 v1 = *reinterpret_cast<const double2*>(values + i + 2);
					sVal[ldsSlot(li)] = v0.x;
					sVal[ldsSlot(li + 1)] = v0.y;
					sVal[ldsSlot(li + 2)] = v1.x;
					sVal[ldsSlot(li + 3)] = v1.y;
				}
			} else {
				for (int j = 0; j < 4 && i + j < n1; ++j) {
					sPos[ldsSlot(li + j)] = positions[i + j];
					sVal[ldsSlot(li + j)] = values[i + j];
				}
			}
		}
		__syncthreads();
		// ---- row sums from LDS: thread -> (row, piece), row-fastest so that the x gather is coalesced ----
		for (int rb = r0; rb < r1; rb += R) {
			const int row = rb + rowLocal;
			T dot = T(0);
			if (row < r1) {
				const int b = start[row] - a0;
				const int e = start[row + 1] - a0;
				int kb = b, ke = e;
				if (L > 1) {
					const int chunk = (e - b + L - 1) / L;
					kb = b + piece * chunk;
					ke = min(e, kb + chunk);
				}
				for (int k = kb; k < ke; ++k) {
					const int s = ldsSlot(k);
					dot = smmFma(sVal[s], x[sPos[s]], dot);
				}
			}
			if (L > 1) {
				sPart[t] = dot;
				__syncthreads();
				if (piece == 0 && row < r1) {
#pragma unroll
					for (int q = 1; q < L; ++q) {
						dot += sPart[q * R + rowLocal];
					}
				}
				__syncthreads();
			}
			if (piece == 0 && row < r1) {
				const T o = applyOp(op, lhs, row, dot);
				out[row] = o;
				if (dotMode == 2) acc0 += o * o;
				if (dotMode) acc1 += o * w1[row];
			}
		}
		__syncthreads();  // LDS is restaged by the next row block
	}
	if (dotMode) {
		if (dotMode == 2) {
			const T s0 = blockSum256(acc0, red);
			if (t == 0) partials[blockIdx.x] = s0;
		}
		const T s1 = blockSum256(acc1, red);
		if (t == 0) partials[(dotMode == 2 ? NPART : 0) + blockIdx.x] = s1;
	}
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// Cut rows into blocks of <= cap nonzeros and <= TPB rows; a row longer than cap gets a block of its own.
int buildRowBlocks(smm_hip_csr* m, int cap) {
	std::vector<int> hs(static_cast<size_t>(m->rows) + 1);
	hipStream_t s = libStream();
	SMM_HIP_TRY(hipMemcpyAsync(hs.data(), m->d_start, hs.size() * sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	std::vector<int> rb;
	rb.reserve(static_cast<size_t>(m->nnz / cap + m->rows / TPB + 2));
	int r = 0;
	const int rows = m->rows;
	while (r < rows) {
		rb.push_back(r);
		const int base = hs[r];
		int e = r + 1;  // always take at least one row
		const int limitRow = std::min(rows, r + TPB);
		while (e < limitRow && hs[e + 1] - base <= cap) ++e;
		if (hs[e] - base > cap) {
			// first row alone exceeds the cap: e == r + 1, long-row path
		}
		r = e;
	}
	rb.push_back(rows);
	devFree(m->d_rowblocks);
	m->d_rowblocks = nullptr;
	m->n_rowblocks = static_cast<int>(rb.size()) - 1;
	m->stream_nnz_cap = cap;
	SMM_TRY(devAlloc(reinterpret_cast<void**>(&m->d_rowblocks), rb.size() * sizeof(int)));
	SMM_HIP_TRY(hipMemcpyAsync(m->d_rowblocks, rb.data(), rb.size() * sizeof(int), hipMemcpyHostToDevice, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

static int lanesForAvg(double avg, int family) {
	if (family == SMM_SPMV_STREAM) {
		// pieces of ~12-16 entries per lane; L == 1 keeps the reference's summation order bit for bit
		if (avg <= 24) return 1;
		if (avg <= 40) return 2;
		if (avg <= 96) return 4;
		if (avg <= 192) return 8;
		if (avg <= 384) return 16;
		return 32;
	}
	int l = 1;
	while (l < 64 && l * 2 <= avg) l *= 2;
	return l;
}

void chooseSpmvConfig(smm_hip_csr* m) {
	const double avg = m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 0.0;
	int family = SMM_SPMV_STREAM;
	if (const char* env = getenv("SMM_HIP_SPMV_FAMILY")) {
		const int f = atoi(env);
		if (f == SMM_SPMV_VECTOR || f == SMM_SPMV_STREAM) family = f;
	}
	if (m->rows == 0 || m->nnz == 0) family = SMM_SPMV_VECTOR;
	m->family = family;
	m->lanes = lanesForAvg(avg, family);
	if (const char* env = getenv("SMM_HIP_SPMV_LANES")) {
		const int l = atoi(env);
		if (l >= 1 && l <= 64 && (l & (l - 1)) == 0) m->lanes = l;
	}
}

template <typename T, int L>
static void launchVector(const smm_hip_csr* m, int grid, int op, const T* lhs, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s) {
	spmvVectorKernel<T, L><<<grid, TPB, 0, s>>>(m->rows, m->d_start, m->d_positions, static_cast<const T*>(m->d_values), op, lhs, x, out,
	                                            dotMode, w1, partials, doneFlag);
}

template <typename T, int L>
static void launchStream(const smm_hip_csr* m, int grid, int op, const T* lhs, const T* x, T* out, int dotMode, const T* w1, T* partials,
                         const int* doneFlag, hipStream_t s) {
	spmvStreamKernel<T, L><<<grid, TPB, 0, s>>>(m->n_rowblocks, m->d_rowblocks, m->d_start, m->d_positions, static_cast<const T*>(m->d_values),
	                                            op, lhs, x, out, dotMode, w1, partials, doneFlag);
}

template <typename T>
int launchSpmv(const smm_hip_csr* m, int op, const T* lhs, const T* x, T* out, int dotMode, const T* w1, T* partials, const int* doneFlag,
               hipStream_t s) {
	if (m->dtype != dtypeOf<T>()) {
		setError("spmv: matrix dtype does not match the _f32/_f64 entry point");
		return SMM_HIP_ERR_INVALID;
	}
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
	if (m->rows == 0 && !dotMode) return SMM_HIP_OK;
	int family = m->family;
	if (family == SMM_SPMV_STREAM && !m->d_rowblocks) {
		SMM_TRY(buildRowBlocks(const_cast<smm_hip_csr*>(m), StreamCfg<T>::CAP));
	}
	const int L = m->lanes;
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
	case 1: FN<T, 1>(m, grid, op, lhs, x, out, dotMode, w1, partials, doneFlag, s); break;   \
	case 2: FN<T, 2>(m, grid, op, lhs, x, out, dotMode, w1, partials, doneFlag, s); break;   \
	case 4: FN<T, 4>(m, grid, op, lhs, x, out, dotMode, w1, partials, doneFlag, s); break;   \
	case 8: FN<T, 8>(m, grid, op, lhs, x, out, dotMode, w1, partials, doneFlag, s); break;   \
	case 16: FN<T, 16>(m, grid, op, lhs, x, out, dotMode, w1, partials, doneFlag, s); break; \
	case 32: FN<T, 32>(m, grid, op, lhs, x, out, dotMode, w1, partials, doneFlag, s); break; \
	default: FN<T, 64>(m, grid, op, lhs, x, out, dotMode, w1, partials, doneFlag, s); break; \
	}
	if (family == SMM_SPMV_STREAM) {
		SMM_DISPATCH_L(launchStream)
	} else {
		SMM_DISPATCH_L(launchVector)
	}
#undef SMM_DISPATCH_L
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template int launchSpmv<float>(const smm_hip_csr*, int, const float*, const float*, float*, int, const float*, float*, const int*, hipStream_t);
template int launchSpmv<double>(const smm_hip_csr*, int, const double*, const double*, double*, int, const double*, double*, const int*, hipStream_t);

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
	if (m->cols) SMM_HIP_TRY(hipMemcpyAsync(dx, x, sizeof(T) * m->cols, hipMemcpyHostToDevice, s));
	const T* dlhs = nullptr;
	if (op != SMM_OP_ASSIGN) {
		if (!lhs && m->rows) {
			setError("spmv: null lhs");
			return SMM_HIP_ERR_INVALID;
		}
		SMM_TRY(dl.alloc(m->rows));
		if (m->rows) SMM_HIP_TRY(hipMemcpyAsync(dl, lhs, sizeof(T) * m->rows, hipMemcpyHostToDevice, s));
		dlhs = dl;
	}
	SMM_TRY(launchSpmv<T>(m, op, dlhs, dx, dout, 0, nullptr, nullptr, nullptr, s));
	if (m->rows) SMM_HIP_TRY(hipMemcpyAsync(out, dout, sizeof(T) * m->rows, hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
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
	if (family != SMM_SPMV_AUTO && family != SMM_SPMV_VECTOR && family != SMM_SPMV_STREAM) {
		setError("csr_set_kernel: unknown family %d", family);
		return SMM_HIP_ERR_INVALID;
	}
	if (lanes_per_row != 0 && (lanes_per_row < 1 || lanes_per_row > 64 || (lanes_per_row & (lanes_per_row - 1)))) {
		setError("csr_set_kernel: lanes_per_row must be 0 or a power of two in 1..64");
		return SMM_HIP_ERR_INVALID;
	}
	if (family == SMM_SPMV_AUTO) {
		chooseSpmvConfig(m);
	} else {
		m->family = family;
		const double avg = m->rows > 0 ? static_cast<double>(m->nnz) / m->rows : 0.0;
		m->lanes = lanesForAvg(avg, family);
	}
	if (lanes_per_row) m->lanes = lanes_per_row;
	return SMM_HIP_OK;
}

int smm_hip_csr_get_kernel(const smm_hip_csr* m, int* family, int* lanes_per_row) {
	if (!m) {
		setError("csr_get_kernel: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	if (family) *family = m->family;
	if (lanes_per_row) *lanes_per_row = m->lanes;
	return SMM_HIP_OK;
}

int smm_hip_csr_autotune(smm_hip_csr* m) {
	if (!m) {
		setError("csr_autotune: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	if (m->rows == 0 || m->nnz == 0) return SMM_HIP_OK;
	hipStream_t s = libStream();
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
	int bestFamily = m->family, bestLanes = m->lanes;
	int status = SMM_HIP_OK;
	for (int family = SMM_SPMV_VECTOR; family <= SMM_SPMV_STREAM && status == SMM_HIP_OK; ++family) {
		const int center = lanesForAvg(avg, family);
		for (int lanes = std::max(1, center / 2); lanes <= std::min(64, center * 2) && status == SMM_HIP_OK; lanes *= 2) {
			m->family = family;
			m->lanes = lanes;
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
	m->family = bestFamily;
	m->lanes = bestLanes;
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	devFree(dx);
	devFree(dy);
	return status;
}

}  // extern "C"
