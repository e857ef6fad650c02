// smm_blas1.hip -- streaming reductions and AXPY-style updates (gfx950).
//
// dot: Vector<T>::operator* (ref:305-328) / secondNormSquared (ref:296-303).  The reference sums serially (or as
// a grain-8192 TBB tree); here every lane keeps a private sum over its packs of 16 bytes (streamMap, smm_device.h), lanes
// meet in a wave butterfly (__shfl_xor), waves meet through LDS, and each of the NPART workgroups writes one
// partial; a one-workgroup kernel adds the partials in a fixed order.  No atomics: results are bitwise
// reproducible from run to run.  Accumulation is in T, like the reference.
#include <algorithm>
#include <map>

#include "smm_device.h"
#include "smm_internal.h"

namespace smm {

constexpr int TPB = 256;

template <typename T, bool NT>
__global__ __launch_bounds__(TPB) void dotPartialsKernel(int n, const T* a, const T* b, T* __restrict__ partials, const int* __restrict__ doneFlag) {
	__shared__ T red[4];
	if (doneFlag && *doneFlag) return;
	T acc = T(0);
	if (a == b) {  // ||a||^2: read the vector once
		const T* const in[1] = {a};
		streamMap<T, NT, 1, 0>(n, in, nullptr, [&](const T(&v)[1], T(&)[1]) { acc += v[0] * v[0]; });
	} else {
		const T* const in[2] = {a, b};
		streamMap<T, NT, 2, 0>(n, in, nullptr, [&](const T(&v)[2], T(&)[1]) { acc += v[0] * v[1]; });
	}
	const T s = blockSum256(acc, red);
	if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

template <typename T>
__global__ __launch_bounds__(TPB) void sumPartialsKernel(const T* __restrict__ partials, T* __restrict__ result) {
	__shared__ T red[4];
	T acc = T(0);
	for (int i = threadIdx.x; i < NPART; i += TPB) acc += partials[i];
	const T s = blockSum256(acc, red);
	if (threadIdx.x == 0) result[0] = s;
}

template <typename T>
__global__ __launch_bounds__(TPB) void axpyKernel(int n, T alpha, const T* x, const T* y, T* out) {
	const T* const in[2] = {x, y};
	T* const o[1] = {out};
	streamMap<T, false, 2, 1>(n, in, o, [&](const T(&v)[2], T(&r)[1]) { r[0] = smmFma(alpha, v[0], v[1]); });
}

template <typename T>
__global__ __launch_bounds__(TPB) void copy2Kernel(int n, const T* src, T* d1, T* d2) {
	const T* const in[1] = {src};
	if (d2) {
		T* const o[2] = {d1, d2};
		streamMap<T, false, 1, 2>(n, in, o, [](const T(&v)[1], T(&r)[2]) { r[0] = r[1] = v[0]; });
	} else {
		T* const o[1] = {d1};
		streamMap<T, false, 1, 1>(n, in, o, [](const T(&v)[1], T(&r)[1]) { r[0] = v[0]; });
	}
}

static int gridFor(long long n) { return static_cast<int>(std::max<long long>(1, std::min<long long>((n + TPB - 1) / TPB, numCUs() * 8LL))); }

template <typename T>
int launchDotPartials(int n, const T* a, const T* b, T* partials, const int* doneFlag, hipStream_t s) {
	// non-temporal loads for vectors that are streamed from HBM anyway (tools/membw.hip: 7.0 vs 6.2 TB/s on this box)
	if (static_cast<double>(n) * sizeof(T) * 2 > 192.0 * 1024 * 1024) {
		dotPartialsKernel<T, true><<<NPART, TPB, 0, s>>>(n, a, b, partials, doneFlag);
	} else {
		dotPartialsKernel<T, false><<<NPART, TPB, 0, s>>>(n, a, b, partials, doneFlag);
	}
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T>
int launchSumPartials(const T* partials, T* result, hipStream_t s) {
	sumPartialsKernel<T><<<1, TPB, 0, s>>>(partials, result);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T>
int launchAxpy(int n, T a, const T* x, const T* y, T* out, hipStream_t s) {
	if (n <= 0) return SMM_HIP_OK;
	axpyKernel<T><<<gridFor(n), TPB, 0, s>>>(n, a, x, y, out);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T>
int launchCopy2(int n, const T* src, T* dst1, T* dst2, hipStream_t s) {
	if (n <= 0) return SMM_HIP_OK;
	copy2Kernel<T><<<gridFor(n), TPB, 0, s>>>(n, src, dst1, dst2);
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

#define SMM_INSTANTIATE(T)                                                                     \
	template int launchDotPartials<T>(int, const T*, const T*, T*, const int*, hipStream_t);   \
	template int launchSumPartials<T>(const T*, T*, hipStream_t);                              \
	template int launchAxpy<T>(int, T, const T*, const T*, T*, hipStream_t);                   \
	template int launchCopy2<T>(int, const T*, T*, T*, hipStream_t);
SMM_INSTANTIATE(float)
SMM_INSTANTIATE(double)
#undef SMM_INSTANTIATE

template <typename T>
static int dotDev(int n, const T* a, const T* b, T* d_result, hipStream_t s) {
	if (n < 0 || !d_result || (n > 0 && (!a || !b))) {
		setError("dot: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	// This entry point only enqueues, so its partial-sum buffer cannot go back to the allocator here.  One persistent buffer PER
	// STREAM (and scalar type): calls on different streams never share a buffer, calls on one stream are ordered by the stream -- and the
	// two launches of a call are enqueued under one lock, so that calls made by several host threads on the SAME stream cannot interleave
	// (A's partials, B's partials, A's sum).
	static std::map<hipStream_t, T*> buffers;
	static std::mutex mu;
	std::lock_guard<std::mutex> lock(mu);
	auto it = buffers.find(s);
	if (it == buffers.end()) {
		T* p = nullptr;
		SMM_TRY(devAlloc(reinterpret_cast<void**>(&p), NPART * sizeof(T)));
		it = buffers.emplace(s, p).first;
	}
	T* partials = it->second;
	SMM_TRY(launchDotPartials<T>(n, a, b, partials, nullptr, s));
	SMM_TRY(launchSumPartials<T>(partials, d_result, s));
	return SMM_HIP_OK;
}

template <typename T>
static int dotHost(int n, const T* a, const T* b, T* result) {
	if (n < 0 || !result || (n > 0 && (!a || !b))) {
		setError("dot: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	hipStream_t s = libStream();
	DevBuf<T> da, db, dr;
	SMM_TRY(da.alloc(n));
	SMM_TRY(dr.alloc(1));
	SMM_TRY(hostToDev(da, a, sizeof(T) * n, s));
	const T* pb = da;
	if (b != a) {
		SMM_TRY(db.alloc(n));
		SMM_TRY(hostToDev(db, b, sizeof(T) * n, s));
		pb = db;
	}
	SMM_TRY(dotDev<T>(n, da, pb, dr, s));
	SMM_HIP_TRY(hipMemcpyAsync(result, dr, sizeof(T), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

// every translation unit of the library is a code object of its own, built for the device at the FIRST launch of any of its kernels
// (5-9 ms each, measured: profiles/r04/first_spmv_setup_trace.txt); smm_hip_init touches one kernel of each hot-path unit so that
// the first SpMV of a process does not pay for it (SMM_HIP_PRELOAD=0: load lazily as before)
void preloadBlas1Unit() {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(sumPartialsKernel<float>));
	(void)hipGetLastError();
}

}  // namespace smm

using namespace smm;

extern "C" {

int smm_hip_dot_f32(int n, const float* a, const float* b, float* result) { return dotHost<float>(n, a, b, result); }
int smm_hip_dot_f64(int n, const double* a, const double* b, double* result) { return dotHost<double>(n, a, b, result); }
int smm_hip_dot_dev_f32(int n, const float* d_a, const float* d_b, float* d_result, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	return dotDev<float>(n, d_a, d_b, d_result, pickStream(stream));
}
int smm_hip_dot_dev_f64(int n, const double* d_a, const double* d_b, double* d_result, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	return dotDev<double>(n, d_a, d_b, d_result, pickStream(stream));
}

int smm_hip_axpy_dev_f32(int n, float a, const float* d_x, const float* d_y, float* d_out, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	return launchAxpy<float>(n, a, d_x, d_y, d_out, pickStream(stream));
}
int smm_hip_axpy_dev_f64(int n, double a, const double* d_x, const double* d_y, double* d_out, smm_hip_stream stream) {
	SMM_TRY(ensureInit());
	return launchAxpy<double>(n, a, d_x, d_y, d_out, pickStream(stream));
}

}  // extern "C"
