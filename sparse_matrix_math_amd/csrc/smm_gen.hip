// smm_gen.hip -- device-side generators of the benchmark matrices (BASELINE.json configs).
//
// The reference can only build a CSRMatrix through a std::map-backed TripletMatrix (ref:606-618, 1606-1641), which is
// infeasible for 5e8 entries; these kernels write start[] / positions[] / values[] directly in HBM, one lane per
// row, with start[] in closed form (no prefix sum).  sparse_matrix_math_amd/generators.py implements the same
// laws in numpy; tests check the two agree bit for bit.
#include <algorithm>

#include "smm_internal.h"

namespace smm {

constexpr int TPB = 256;
constexpr int MAXK = 32;

struct BandOffsets {
	int k;
	int d[MAXK];
};

__host__ __device__ inline unsigned long long mix64(unsigned long long z) {
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

// K distinct offsets in [1, min(max_offset, n)) drawn from splitmix64(seed), ascending
static BandOffsets drawOffsets(int n, int k, unsigned long long seed, int maxOffset) {
	BandOffsets o{};
	const long long M = std::min<long long>(maxOffset, n);
	if (M < 2) return o;
	const int want = static_cast<int>(std::min<long long>(std::min(k, MAXK), M - 1));
	unsigned long long state = seed;
	while (o.k < want) {
		state += 0x9E3779B97F4A7C15ULL;
		const int cand = 1 + static_cast<int>(mix64(state) % static_cast<unsigned long long>(M - 1));
		bool dup = false;
		for (int i = 0; i < o.k; ++i) dup = dup || o.d[i] == cand;
		if (!dup) o.d[o.k++] = cand;
	}
	std::sort(o.d, o.d + o.k);
	return o;
}

// off-diagonal value shared by A[i][j] and A[j][i]: -(0.02 + 0.98 u), u in [0,1) with 24 random bits
template <typename T>
__device__ __forceinline__ T bandValue(unsigned long long seed, int r, int k) {
	const unsigned long long h = mix64(seed + (static_cast<unsigned long long>(r) * 64ULL + static_cast<unsigned long long>(k) + 1ULL) * 0x9E3779B97F4A7C15ULL);
	const T u = static_cast<T>(static_cast<float>(h >> 40) * 5.9604644775390625e-08f);  // 2^-24, exact
	const T scaled = T(0.98) * u;
	return -(T(0.02) + scaled);
}

__device__ __host__ inline long long bandStart(long long i, long long n, const BandOffsets& o) {
	long long s = i;
	for (int k = 0; k < o.k; ++k) {
		const long long left = i - o.d[k];
		const long long cap = n - o.d[k];
		s += left > 0 ? left : 0;
		s += i < (cap > 0 ? cap : 0) ? i : (cap > 0 ? cap : 0);
	}
	return s;
}

// rows [rowBegin, rowEnd) of the n x n matrix: start[] is local (start[0] == 0), positions[] are global columns
template <typename T>
__global__ __launch_bounds__(TPB) void genBandedKernel(int n, BandOffsets o, unsigned long long seed, T diagShift, int rowBegin, int rowEnd,
                                                       int* __restrict__ start, int* __restrict__ positions, T* __restrict__ values) {
	const long long base = bandStart(rowBegin, n, o);
	for (long long i = rowBegin + static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i <= rowEnd; i += static_cast<long long>(gridDim.x) * TPB) {
		const long long s = bandStart(i, n, o) - base;
		start[i - rowBegin] = static_cast<int>(s);
		if (i == rowEnd) continue;
		const int row = static_cast<int>(i);
		long long w = s;
		T diag = diagShift;
		for (int k = o.k - 1; k >= 0; --k) {  // columns row - d_k, ascending
			const int j = row - o.d[k];
			if (j >= 0) {
				const T v = bandValue<T>(seed, j, k);
				positions[w] = j;
				values[w] = v;
				diag = diag + (-v);
				++w;
			}
		}
		const long long diagSlot = w++;
		for (int k = 0; k < o.k; ++k) {  // columns row + d_k, ascending
			const long long j = static_cast<long long>(row) + o.d[k];
			if (j < n) {
				const T v = bandValue<T>(seed, row, k);
				positions[w] = static_cast<int>(j);
				values[w] = v;
				diag = diag + (-v);
				++w;
			}
		}
		positions[diagSlot] = row;
		values[diagSlot] = diag;
	}
}

template <typename T>
__global__ __launch_bounds__(TPB) void genPoisson2dKernel(int nx, int ny, int* __restrict__ start, int* __restrict__ positions, T* __restrict__ values) {
	const long long n = static_cast<long long>(nx) * ny;
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i <= n; i += static_cast<long long>(gridDim.x) * TPB) {
		const long long s = 5 * i - min(i, static_cast<long long>(nx)) - max(0LL, i - static_cast<long long>(nx) * (ny - 1)) - (i + nx - 1) / nx - i / nx;
		start[i] = static_cast<int>(s);
		if (i == n) continue;
		const int ix = static_cast<int>(i % nx);
		const int iy = static_cast<int>(i / nx);
		long long w = s;
		if (iy > 0) { positions[w] = static_cast<int>(i - nx); values[w++] = T(-1); }
		if (ix > 0) { positions[w] = static_cast<int>(i - 1); values[w++] = T(-1); }
		positions[w] = static_cast<int>(i); values[w++] = T(4);
		if (ix < nx - 1) { positions[w] = static_cast<int>(i + 1); values[w++] = T(-1); }
		if (iy < ny - 1) { positions[w] = static_cast<int>(i + nx); values[w++] = T(-1); }
	}
}

template <typename T>
__global__ __launch_bounds__(TPB) void genStencil3dKernel(int nx, int ny, int nz, T diag, T lo, T hi, int* __restrict__ start,
                                                          int* __restrict__ positions, T* __restrict__ values) {
	const long long plane = static_cast<long long>(nx) * ny;
	const long long n = plane * nz;
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i <= n; i += static_cast<long long>(gridDim.x) * TPB) {
		const long long q = i / plane;
		const long long rem = i % plane;
		long long s = 7 * i;
		s -= min(i, plane);                                   // iz == 0
		s -= max(0LL, i - plane * (nz - 1));                  // iz == nz-1
		s -= (i + nx - 1) / nx;                               // ix == 0
		s -= i / nx;                                          // ix == nx-1
		s -= q * nx + min(rem, static_cast<long long>(nx));   // iy == 0
		s -= q * nx + max(0LL, rem - static_cast<long long>(nx) * (ny - 1));  // iy == ny-1
		start[i] = static_cast<int>(s);
		if (i == n) continue;
		const int ix = static_cast<int>(i % nx);
		const int iy = static_cast<int>((i / nx) % ny);
		const int iz = static_cast<int>(q);
		long long w = s;
		if (iz > 0) { positions[w] = static_cast<int>(i - plane); values[w++] = lo; }
		if (iy > 0) { positions[w] = static_cast<int>(i - nx); values[w++] = lo; }
		if (ix > 0) { positions[w] = static_cast<int>(i - 1); values[w++] = lo; }
		positions[w] = static_cast<int>(i); values[w++] = diag;
		if (ix < nx - 1) { positions[w] = static_cast<int>(i + 1); values[w++] = hi; }
		if (iy < ny - 1) { positions[w] = static_cast<int>(i + nx); values[w++] = hi; }
		if (iz < nz - 1) { positions[w] = static_cast<int>(i + plane); values[w++] = hi; }
	}
}

static int genGrid(long long n) { return static_cast<int>(std::max<long long>(1, std::min<long long>((n + TPB) / TPB, 8192))); }

static bool fitsInt(long long v) { return v >= 0 && v <= 2147483647LL; }

template <typename T>
static int genBanded(int n, int k, unsigned long long seed, int maxOffset, T diagShift, int rowBegin, int rowEnd, int* d_start, int* d_positions,
                     T* d_values, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	if (n < 0 || k < 0 || rowBegin < 0 || rowEnd < rowBegin || rowEnd > n || !d_start || (rowEnd > rowBegin && (!d_positions || !d_values))) {
		setError("gen_banded: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	if (!fitsInt(smm_hip_gen_banded_row_start(n, k, seed, maxOffset, rowEnd) - smm_hip_gen_banded_row_start(n, k, seed, maxOffset, rowBegin))) {
		setError("gen_banded: nnz exceeds int32 (the reference's index type, ref:1251-1257)");
		return SMM_HIP_ERR_INVALID;
	}
	const BandOffsets o = drawOffsets(n, k, seed, maxOffset);
	genBandedKernel<T><<<genGrid(rowEnd - rowBegin), TPB, 0, pickStream(stream)>>>(n, o, seed, diagShift, rowBegin, rowEnd, d_start, d_positions, d_values);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T>
static int genPoisson2d(int nx, int ny, int* d_start, int* d_positions, T* d_values, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	if (nx < 1 || ny < 1 || !d_start || !d_positions || !d_values || !fitsInt(smm_hip_gen_poisson2d_nnz(nx, ny))) {
		setError("gen_poisson2d: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	genPoisson2dKernel<T><<<genGrid(static_cast<long long>(nx) * ny), TPB, 0, pickStream(stream)>>>(nx, ny, d_start, d_positions, d_values);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T>
static int genStencil3d(int nx, int ny, int nz, T diag, T lo, T hi, int* d_start, int* d_positions, T* d_values, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	if (nx < 1 || ny < 1 || nz < 1 || !d_start || !d_positions || !d_values || !fitsInt(smm_hip_gen_stencil3d_nnz(nx, ny, nz))) {
		setError("gen_stencil3d: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	genStencil3dKernel<T><<<genGrid(static_cast<long long>(nx) * ny * nz), TPB, 0, pickStream(stream)>>>(nx, ny, nz, diag, lo, hi, d_start, d_positions, d_values);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

}  // namespace smm

using namespace smm;

extern "C" {

long long smm_hip_gen_poisson2d_nnz(int nx, int ny) {
	const long long n = static_cast<long long>(nx) * ny;
	return 5 * n - 2LL * nx - 2LL * ny;
}

long long smm_hip_gen_stencil3d_nnz(int nx, int ny, int nz) {
	const long long n = static_cast<long long>(nx) * ny * nz;
	return 7 * n - 2LL * nx * ny - 2LL * ny * nz - 2LL * nx * nz;
}

long long smm_hip_gen_banded_row_start(int n, int k, unsigned long long seed, int max_offset, int row) {
	const BandOffsets o = drawOffsets(n, k, seed, max_offset);
	return bandStart(row, n, o);
}

long long smm_hip_gen_banded_nnz(int n, int k, unsigned long long seed, int max_offset) {
	const BandOffsets o = drawOffsets(n, k, seed, max_offset);
	long long nnz = n;
	for (int i = 0; i < o.k; ++i) nnz += 2LL * std::max(0, n - o.d[i]);
	return nnz;
}

int smm_hip_gen_poisson2d_dev_f32(int nx, int ny, int* d_start, int* d_positions, float* d_values, smm_hip_stream stream) {
	return genPoisson2d<float>(nx, ny, d_start, d_positions, d_values, stream);
}
int smm_hip_gen_poisson2d_dev_f64(int nx, int ny, int* d_start, int* d_positions, double* d_values, smm_hip_stream stream) {
	return genPoisson2d<double>(nx, ny, d_start, d_positions, d_values, stream);
}
int smm_hip_gen_stencil3d_dev_f32(int nx, int ny, int nz, float diag, float lo, float hi, int* d_start, int* d_positions, float* d_values, smm_hip_stream stream) {
	return genStencil3d<float>(nx, ny, nz, diag, lo, hi, d_start, d_positions, d_values, stream);
}
int smm_hip_gen_stencil3d_dev_f64(int nx, int ny, int nz, double diag, double lo, double hi, int* d_start, int* d_positions, double* d_values, smm_hip_stream stream) {
	return genStencil3d<double>(nx, ny, nz, diag, lo, hi, d_start, d_positions, d_values, stream);
}
int smm_hip_gen_banded_dev_f32(int n, int k, unsigned long long seed, int max_offset, float diag_shift, int* d_start, int* d_positions, float* d_values, smm_hip_stream stream) {
	return genBanded<float>(n, k, seed, max_offset, diag_shift, 0, n, d_start, d_positions, d_values, stream);
}
int smm_hip_gen_banded_dev_f64(int n, int k, unsigned long long seed, int max_offset, double diag_shift, int* d_start, int* d_positions, double* d_values, smm_hip_stream stream) {
	return genBanded<double>(n, k, seed, max_offset, diag_shift, 0, n, d_start, d_positions, d_values, stream);
}

int smm_hip_gen_banded_rows_dev_f32(int n, int k, unsigned long long seed, int max_offset, float diag_shift, int row_begin, int row_end, int* d_start,
                                    int* d_positions, float* d_values, smm_hip_stream stream) {
	return genBanded<float>(n, k, seed, max_offset, diag_shift, row_begin, row_end, d_start, d_positions, d_values, stream);
}
int smm_hip_gen_banded_rows_dev_f64(int n, int k, unsigned long long seed, int max_offset, double diag_shift, int row_begin, int row_end, int* d_start,
                                    int* d_positions, double* d_values, smm_hip_stream stream) {
	return genBanded<double>(n, k, seed, max_offset, diag_shift, row_begin, row_end, d_start, d_positions, d_values, stream);
}

}  // extern "C"
