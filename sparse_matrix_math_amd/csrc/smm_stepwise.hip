// smm_stepwise.hip -- BiCGStab (ref:2191-2283) cut into stages at its global reductions, for the row-partitioned
// multi-GPU driver (sparse_matrix_math_amd/distributed.py): one process per GPU owns a contiguous range of rows, runs the
// stages below on its slice, and between a *_LOCAL and the following *_APPLY stage the host side all-reduces the few
// scalars in `sums` over RCCL.  Every scalar of the recurrence stays in device memory; nothing here synchronises.
//
// The arithmetic of each stage is the same expression, in the same order, as the single-GPU loop in smm_solvers.hip;
// only the dot products are completed across ranks.
#include <algorithm>

#include "smm_device.h"
#include "smm_internal.h"

namespace smm {

constexpr int TPB = 256;

// device-resident recurrence state: c[0] rr0, c[1] alpha, c[2] omega, c[3] beta, c[4] resL2Norm; f[0] done, f[1] iterations
template <typename T>
struct StepState {
	T c[8];
	int f[4];
};

}  // namespace smm

struct smm_hip_bicgstab_ws {
	int dtype = 0;
	int n = 0;
	void* r = nullptr;
	void* r0 = nullptr;
	void* ap = nullptr;
	void* as = nullptr;
	void* p = nullptr;  // bound by the caller: owned slice inside its halo-extended buffer
	void* s = nullptr;
	void* partials = nullptr;  // 2 * NPART
	void* sums = nullptr;      // 4 scalars the caller all-reduces
	void* state = nullptr;     // StepState<T>
	bool ownsSums = true;
};

namespace smm {

template <typename T>
__global__ __launch_bounds__(TPB) void stepInit(int n, const T* __restrict__ r, T* __restrict__ r0, T* __restrict__ p, T* __restrict__ partials) {
	__shared__ T red[4];
	T acc = T(0);
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * TPB) {
		const T v = r[i];
		r0[i] = v;  // ref:2225-2226
		p[i] = v;
		acc += v * v;  // r.r0 with r0 == r, ref:2231
	}
	const T s = blockSum256(acc, red);
	if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// sums[i] = sum of partials[i * NPART .. (i+1) * NPART) in a fixed order, i < nsets
template <typename T>
__global__ __launch_bounds__(TPB) void stepSums(const T* __restrict__ partials, int nsets, T* __restrict__ sums, const StepState<T>* st, int checkDone) {
	__shared__ T red[4];
	if (checkDone && st->f[0]) return;
	for (int k = 0; k < nsets; ++k) {
		T acc = T(0);
		for (int i = threadIdx.x; i < NPART; i += TPB) acc += partials[k * NPART + i];
		const T s = blockSum256(acc, red);
		if (threadIdx.x == 0) sums[k] = s;
	}
}

template <typename T>
__global__ void stepCoefInit(const T* __restrict__ sums, StepState<T>* st) {
	st->c[0] = sums[0];  // rr0
	st->c[4] = T(0);
	st->f[0] = 0;
	st->f[1] = 0;
}

template <typename T>
__global__ void stepCoefAlpha(const T* __restrict__ sums, StepState<T>* st) {
	if (st->f[0]) return;
	st->c[1] = st->c[0] / sums[0];  // alpha = rr0 / (ap.r0), ref:2243-2244
}

template <typename T>
__global__ __launch_bounds__(TPB) void stepUpdateS(int n, const StepState<T>* __restrict__ st, const T* ap, const T* r, T* sv) {
	if (st->f[0]) return;
	const T alpha = st->c[1];
	const T* const in[2] = {ap, r};
	T* const out[1] = {sv};
	streamMap<T, false, 2, 1>(n, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(-alpha, v[0], v[1]); });  // ref:2245-2247
}

template <typename T>
__global__ void stepCoefOmega(const T* __restrict__ sums, StepState<T>* st) {
	if (st->f[0]) return;
	st->c[2] = sums[1] / sums[0];  // omega = (as.s) / (as.as), ref:2259-2261
}

template <typename T>
__global__ __launch_bounds__(TPB) void stepUpdateXR(int n, const StepState<T>* __restrict__ st, const T* p, const T* sv, const T* as, const T* r0,
                                                    T* x, T* r, T* __restrict__ partials) {
	__shared__ T red[4];
	if (st->f[0]) return;
	const T alpha = st->c[1];
	const T omega = st->c[2];
	T acc0 = T(0), acc1 = T(0);
	const T* const in[5] = {sv, x, p, as, r0};
	T* const out[2] = {x, r};
	streamMap<T, false, 5, 2>(n, in, out, [&](const T(&v)[5], T(&o)[2]) {
		const T si = v[0];
		o[0] = smmFma(alpha, v[2], smmFma(omega, si, v[1]));  // ref:2264
		const T ri = smmFma(-omega, v[3], si);                // ref:2265
		o[1] = ri;
		acc0 += ri * ri;
		acc1 += ri * v[4];
	});
	const T s0 = blockSum256(acc0, red);
	const T s1 = blockSum256(acc1, red);
	if (threadIdx.x == 0) {
		partials[blockIdx.x] = s0;
		partials[NPART + blockIdx.x] = s1;
	}
}

template <typename T>
__global__ void stepCoefBeta(const T* __restrict__ sums, StepState<T>* st, T eps) {
	if (st->f[0]) return;
	const T rr = sums[0];
	const T newRR0 = sums[1];
	const T res = sizeof(T) == 4 ? static_cast<T>(__fsqrt_rn(static_cast<float>(rr))) : static_cast<T>(__dsqrt_rn(static_cast<double>(rr)));
	st->c[4] = res;                                             // ref:2268
	st->c[3] = (newRR0 * st->c[1]) / (st->c[0] * st->c[2]);     // beta, ref:2271
	st->c[0] = newRR0;
	st->f[1] += 1;
	if (!(res > eps)) st->f[0] = 1;                             // ref:2277
}

template <typename T>
__global__ __launch_bounds__(TPB) void stepUpdateP(int n, const StepState<T>* __restrict__ st, const T* ap, const T* r, T* p) {
	// the reference also updates p on the converging iteration (ref:2272-2274 precede the loop test); p is not an output,
	// so that last update is skipped together with every later no-op iteration
	if (st->f[0]) return;
	const T beta = st->c[3];
	const T omega = st->c[2];
	const T* const in[3] = {ap, p, r};
	T* const out[1] = {p};
	streamMap<T, false, 3, 1>(n, in, out, [&](const T(&v)[3], T(&o)[1]) { o[0] = smmFma(beta, smmFma(-omega, v[0], v[1]), v[2]); });
}

static int gridFor(long long n) { return static_cast<int>(std::max<long long>(1, std::min<long long>((n + TPB - 1) / TPB, NPART))); }

template <typename T>
static int wsCreate(int n, smm_hip_bicgstab_ws** out) {
	if (!out || n < 0) {
		setError("bicgstab_ws_create: bad arguments");
		return SMM_HIP_ERR_INVALID;
	}
	*out = nullptr;
	SMM_TRY(ensureInit());
	auto* ws = new smm_hip_bicgstab_ws();
	ws->dtype = dtypeOf<T>();
	ws->n = n;
	const size_t vb = static_cast<size_t>(std::max(1, n)) * sizeof(T);
	int st = devAlloc(&ws->r, vb);
	if (st == SMM_HIP_OK) st = devAlloc(&ws->r0, vb);
	if (st == SMM_HIP_OK) st = devAlloc(&ws->ap, vb);
	if (st == SMM_HIP_OK) st = devAlloc(&ws->as, vb);
	if (st == SMM_HIP_OK) st = devAlloc(&ws->partials, 2 * NPART * sizeof(T));
	if (st == SMM_HIP_OK) st = devAlloc(&ws->sums, 4 * sizeof(T));
	if (st == SMM_HIP_OK) st = devAlloc(&ws->state, sizeof(StepState<T>));
	if (st != SMM_HIP_OK) {
		smm_hip_bicgstab_ws_destroy(ws);
		return st;
	}
	*out = ws;
	return SMM_HIP_OK;
}

template <typename T>
static int wsStage(smm_hip_bicgstab_ws* ws, int stage, T* x, T eps, hipStream_t s) {
	const int n = ws->n;
	T* r = static_cast<T*>(ws->r);
	T* r0 = static_cast<T*>(ws->r0);
	T* ap = static_cast<T*>(ws->ap);
	T* as = static_cast<T*>(ws->as);
	T* p = static_cast<T*>(ws->p);
	T* sv = static_cast<T*>(ws->s);
	T* parts = static_cast<T*>(ws->partials);
	T* sums = static_cast<T*>(ws->sums);
	auto* st = static_cast<StepState<T>*>(ws->state);
	if (!p || !sv) {
		setError("bicgstab_ws_stage: p / s not bound");
		return SMM_HIP_ERR_INVALID;
	}
	switch (stage) {
	case SMM_STAGE_INIT_LOCAL:
		stepInit<T><<<NPART, TPB, 0, s>>>(n, r, r0, p, parts);
		stepSums<T><<<1, TPB, 0, s>>>(parts, 1, sums, st, 0);
		break;
	case SMM_STAGE_INIT_APPLY:
		stepCoefInit<T><<<1, 1, 0, s>>>(sums, st);
		break;
	case SMM_STAGE_ALPHA_LOCAL:
		stepSums<T><<<1, TPB, 0, s>>>(parts, 1, sums, st, 1);
		break;
	case SMM_STAGE_ALPHA_APPLY:
		stepCoefAlpha<T><<<1, 1, 0, s>>>(sums, st);
		stepUpdateS<T><<<gridFor(n), TPB, 0, s>>>(n, st, ap, r, sv);
		break;
	case SMM_STAGE_OMEGA_LOCAL:
		stepSums<T><<<1, TPB, 0, s>>>(parts, 2, sums, st, 1);
		break;
	case SMM_STAGE_OMEGA_APPLY:
		if (!x && n > 0) {
			setError("bicgstab_ws_stage: x is null");
			return SMM_HIP_ERR_INVALID;
		}
		stepCoefOmega<T><<<1, 1, 0, s>>>(sums, st);
		stepUpdateXR<T><<<NPART, TPB, 0, s>>>(n, st, p, sv, as, r0, x, r, parts);
		stepSums<T><<<1, TPB, 0, s>>>(parts, 2, sums, st, 1);
		break;
	case SMM_STAGE_BETA_APPLY:
		stepCoefBeta<T><<<1, 1, 0, s>>>(sums, st, eps);
		stepUpdateP<T><<<gridFor(n), TPB, 0, s>>>(n, st, ap, r, p);
		break;
	default:
		setError("bicgstab_ws_stage: unknown stage %d", stage);
		return SMM_HIP_ERR_INVALID;
	}
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

// ---- ConjugateGradient (ref:2316-2398) in stages: the workspace's r, ap and the bound p are used; c[0] = residualNormSquared,
//      c[1] = alpha, c[3] = beta, c[4] = last ||r||^2; f[0] done, f[1] iterations, f[2] SolverStatus ------------------------------
template <typename T>
__global__ __launch_bounds__(TPB) void cgStepInit(int n, const T* __restrict__ r, T* __restrict__ p, T* __restrict__ partials) {
	__shared__ T red[4];
	T acc = T(0);
	for (long long i = static_cast<long long>(blockIdx.x) * TPB + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * TPB) {
		const T v = r[i];
		p[i] = v;      // ref:2340
		acc += v * v;  // ref:2341
	}
	const T s = blockSum256(acc, red);
	if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

template <typename T>
__global__ void cgStepCoefInit(const T* __restrict__ sums, StepState<T>* st, T eps) {
	st->c[0] = sums[0];
	st->c[4] = sums[0];
	st->f[1] = 0;
	st->f[0] = 0;
	st->f[2] = SMM_SOLVER_MAX_ITERATIONS_REACHED;
	if (eps * eps > sums[0]) {  // ref:2342-2344: x stays untouched
		st->f[0] = 1;
		st->f[2] = SMM_SOLVER_SUCCESS;
	}
}

template <typename T>
__global__ void cgStepCoefAlpha(const T* __restrict__ sums, StepState<T>* st) {
	if (st->f[0]) return;
	st->c[1] = st->c[0] / sums[0];  // alpha = rr / (Ap.p), ref:2354-2358
}

template <typename T>
__global__ __launch_bounds__(TPB) void cgStepUpdateXR(int n, const StepState<T>* __restrict__ st, const T* p, const T* Ap, const T* xcur, T* x, T* r,
                                                      T* __restrict__ partials) {
	__shared__ T red[4];
	if (st->f[0]) return;
	const T alpha = st->c[1];
	T acc = T(0);
	const T* const in[4] = {p, xcur, Ap, r};
	T* const out[2] = {x, r};
	streamMap<T, false, 4, 2>(n, in, out, [&](const T(&v)[4], T(&o)[2]) {
		o[0] = smmFma(alpha, v[0], v[1]);  // ref:2372
		const T ri = smmFma(-alpha, v[2], v[3]);
		o[1] = ri;
		acc += ri * ri;
	});
	const T s = blockSum256(acc, red);
	if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

template <typename T>
__global__ void cgStepCoefBeta(const T* __restrict__ sums, StepState<T>* st, T eps) {
	if (st->f[0]) return;
	const T rrNew = sums[0];
	st->f[1] += 1;
	st->c[4] = rrNew;
	if (eps * eps > rrNew) {  // ref:2377-2379
		st->f[0] = 1;
		st->f[2] = SMM_SOLVER_SUCCESS;
	} else {
		st->c[3] = rrNew / st->c[0];  // ref:2381
		st->c[0] = rrNew;
	}
}

template <typename T>
__global__ __launch_bounds__(TPB) void cgStepUpdateP(int n, const StepState<T>* __restrict__ st, const T* r, T* p) {
	if (st->f[0]) return;
	const T beta = st->c[3];
	const T* const in[2] = {p, r};
	T* const out[1] = {p};
	streamMap<T, false, 2, 1>(n, in, out, [&](const T(&v)[2], T(&o)[1]) { o[0] = smmFma(beta, v[0], v[1]); });  // ref:2391-2393
}

template <typename T>
static int cgWsStage(smm_hip_bicgstab_ws* ws, int stage, const T* xcur, T* x, T eps, hipStream_t s) {
	const int n = ws->n;
	T* r = static_cast<T*>(ws->r);
	T* ap = static_cast<T*>(ws->ap);
	T* p = static_cast<T*>(ws->p);
	T* parts = static_cast<T*>(ws->partials);
	T* sums = static_cast<T*>(ws->sums);
	auto* st = static_cast<StepState<T>*>(ws->state);
	if (!p) {
		setError("cg_ws_stage: p not bound");
		return SMM_HIP_ERR_INVALID;
	}
	switch (stage) {
	case SMM_CG_STAGE_INIT_LOCAL:
		cgStepInit<T><<<NPART, TPB, 0, s>>>(n, r, p, parts);
		stepSums<T><<<1, TPB, 0, s>>>(parts, 1, sums, st, 0);
		break;
	case SMM_CG_STAGE_INIT_APPLY:
		cgStepCoefInit<T><<<1, 1, 0, s>>>(sums, st, eps);
		break;
	case SMM_CG_STAGE_ALPHA_LOCAL:
		stepSums<T><<<1, TPB, 0, s>>>(parts, 1, sums, st, 1);
		break;
	case SMM_CG_STAGE_ALPHA_APPLY:
		if ((!x || !xcur) && n > 0) {
			setError("cg_ws_stage: x is null");
			return SMM_HIP_ERR_INVALID;
		}
		cgStepCoefAlpha<T><<<1, 1, 0, s>>>(sums, st);
		cgStepUpdateXR<T><<<NPART, TPB, 0, s>>>(n, st, p, ap, xcur, x, r, parts);
		stepSums<T><<<1, TPB, 0, s>>>(parts, 1, sums, st, 1);
		break;
	case SMM_CG_STAGE_BETA_APPLY:
		cgStepCoefBeta<T><<<1, 1, 0, s>>>(sums, st, eps);
		cgStepUpdateP<T><<<gridFor(n), TPB, 0, s>>>(n, st, r, p);
		break;
	default:
		setError("cg_ws_stage: unknown stage %d", stage);
		return SMM_HIP_ERR_INVALID;
	}
	SMM_HIP_TRY(hipGetLastError());
	return SMM_HIP_OK;
}

template <typename T>
static int wsResult(const smm_hip_bicgstab_ws* ws, smm_hip_stream stream, int* done, int* iterations, T* resnorm) {
	StepState<T> h;
	hipStream_t s = pickStream(stream);
	SMM_HIP_TRY(hipMemcpyAsync(&h, ws->state, sizeof(h), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	if (done) *done = h.f[0];
	if (iterations) *iterations = h.f[1];
	if (resnorm) *resnorm = h.c[4];
	return SMM_HIP_OK;
}

}  // namespace smm

using namespace smm;

extern "C" {

int smm_hip_partials_count(void) { return NPART; }

int smm_hip_spmv_fused_dev_f32(const smm_hip_csr* m, int op, const float* d_lhs, const float* d_x, float* d_out, int dot_mode, const float* d_w1,
                               float* d_partials, smm_hip_stream stream) {
	if (!m) {
		setError("spmv_fused: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return launchSpmv<float>(m, op, d_lhs, d_x, d_out, dot_mode, d_w1, d_partials, nullptr, pickStream(stream));
}
int smm_hip_spmv_fused_dev_f64(const smm_hip_csr* m, int op, const double* d_lhs, const double* d_x, double* d_out, int dot_mode, const double* d_w1,
                               double* d_partials, smm_hip_stream stream) {
	if (!m) {
		setError("spmv_fused: null matrix");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return launchSpmv<double>(m, op, d_lhs, d_x, d_out, dot_mode, d_w1, d_partials, nullptr, pickStream(stream));
}

int smm_hip_finish_len(void) { return PARTS_LEN; }
int smm_hip_finish_totals_offset(void) { return PARTS_TOTALS; }

int smm_hip_spmv_fused_finish_dev_f32(const smm_hip_csr* m, int op, const float* d_lhs, const float* d_x, float* d_out, int dot_mode, const float* d_w1,
                                      float* d_finish, smm_hip_stream stream) {
	if (!m || !d_finish || dot_mode < 1 || dot_mode > 2) {
		setError("spmv_fused_finish: null matrix / buffer, or dot_mode not 1 or 2");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return launchSpmv<float>(m, op, d_lhs, d_x, d_out, dot_mode, d_w1, d_finish, nullptr, pickStream(stream), SPMV_FINISH);
}
int smm_hip_spmv_fused_finish_dev_f64(const smm_hip_csr* m, int op, const double* d_lhs, const double* d_x, double* d_out, int dot_mode, const double* d_w1,
                                      double* d_finish, smm_hip_stream stream) {
	if (!m || !d_finish || dot_mode < 1 || dot_mode > 2) {
		setError("spmv_fused_finish: null matrix / buffer, or dot_mode not 1 or 2");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return launchSpmv<double>(m, op, d_lhs, d_x, d_out, dot_mode, d_w1, d_finish, nullptr, pickStream(stream), SPMV_FINISH);
}

int smm_hip_bicgstab_ws_create_f32(int n, smm_hip_bicgstab_ws** out) { return wsCreate<float>(n, out); }
int smm_hip_bicgstab_ws_create_f64(int n, smm_hip_bicgstab_ws** out) { return wsCreate<double>(n, out); }

int smm_hip_bicgstab_ws_destroy(smm_hip_bicgstab_ws* ws) {
	if (!ws) return SMM_HIP_OK;
	devFree(ws->r);
	devFree(ws->r0);
	devFree(ws->ap);
	devFree(ws->as);
	devFree(ws->partials);
	if (ws->ownsSums) devFree(ws->sums);
	devFree(ws->state);
	delete ws;
	return SMM_HIP_OK;
}

int smm_hip_bicgstab_ws_bind(smm_hip_bicgstab_ws* ws, void* d_p, void* d_s, void* d_sums) {
	if (!ws || !d_p || !d_s) {
		setError("bicgstab_ws_bind: null argument");
		return SMM_HIP_ERR_INVALID;
	}
	ws->p = d_p;
	ws->s = d_s;
	if (d_sums) {
		if (ws->ownsSums) devFree(ws->sums);
		ws->sums = d_sums;
		ws->ownsSums = false;
	}
	return SMM_HIP_OK;
}

int smm_hip_bicgstab_ws_pointers(const smm_hip_bicgstab_ws* ws, void** d_r, void** d_r0, void** d_ap, void** d_as, void** d_partials,
                                 void** d_sums) {
	if (!ws) {
		setError("bicgstab_ws_pointers: null workspace");
		return SMM_HIP_ERR_INVALID;
	}
	if (d_r) *d_r = ws->r;
	if (d_r0) *d_r0 = ws->r0;
	if (d_ap) *d_ap = ws->ap;
	if (d_as) *d_as = ws->as;
	if (d_partials) *d_partials = ws->partials;
	if (d_sums) *d_sums = ws->sums;
	return SMM_HIP_OK;
}

int smm_hip_bicgstab_ws_stage_f32(smm_hip_bicgstab_ws* ws, int stage, float* d_x, float eps, smm_hip_stream stream) {
	if (!ws || ws->dtype != SMM_DTYPE_F32) {
		setError("bicgstab_ws_stage: null workspace or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return wsStage<float>(ws, stage, d_x, eps, pickStream(stream));
}
int smm_hip_bicgstab_ws_stage_f64(smm_hip_bicgstab_ws* ws, int stage, double* d_x, double eps, smm_hip_stream stream) {
	if (!ws || ws->dtype != SMM_DTYPE_F64) {
		setError("bicgstab_ws_stage: null workspace or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return wsStage<double>(ws, stage, d_x, eps, pickStream(stream));
}

int smm_hip_bicgstab_ws_result_f32(const smm_hip_bicgstab_ws* ws, smm_hip_stream stream, int* done, int* iterations, float* resnorm) {
	if (!ws || ws->dtype != SMM_DTYPE_F32) {
		setError("bicgstab_ws_result: null workspace or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return wsResult<float>(ws, stream, done, iterations, resnorm);
}
int smm_hip_bicgstab_ws_result_f64(const smm_hip_bicgstab_ws* ws, smm_hip_stream stream, int* done, int* iterations, double* resnorm) {
	if (!ws || ws->dtype != SMM_DTYPE_F64) {
		setError("bicgstab_ws_result: null workspace or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return wsResult<double>(ws, stream, done, iterations, resnorm);
}

int smm_hip_cg_ws_stage_f32(smm_hip_bicgstab_ws* ws, int stage, const float* d_xcur, float* d_x, float eps, smm_hip_stream stream) {
	if (!ws || ws->dtype != SMM_DTYPE_F32) {
		setError("cg_ws_stage: null workspace or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return cgWsStage<float>(ws, stage, d_xcur, d_x, eps, pickStream(stream));
}
int smm_hip_cg_ws_stage_f64(smm_hip_bicgstab_ws* ws, int stage, const double* d_xcur, double* d_x, double eps, smm_hip_stream stream) {
	if (!ws || ws->dtype != SMM_DTYPE_F64) {
		setError("cg_ws_stage: null workspace or dtype mismatch");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	return cgWsStage<double>(ws, stage, d_xcur, d_x, eps, pickStream(stream));
}

int smm_hip_cg_ws_status(const smm_hip_bicgstab_ws* ws, smm_hip_stream stream, int* solver_status) {
	if (!ws || !solver_status) {
		setError("cg_ws_status: null argument");
		return SMM_HIP_ERR_INVALID;
	}
	SMM_TRY(ensureInit());
	// f[] sits behind 8 scalars of the workspace's dtype
	const size_t off = (ws->dtype == SMM_DTYPE_F32 ? sizeof(float) : sizeof(double)) * 8 + 2 * sizeof(int);
	hipStream_t s = pickStream(stream);
	SMM_HIP_TRY(hipMemcpyAsync(solver_status, static_cast<const char*>(ws->state) + off, sizeof(int), hipMemcpyDeviceToHost, s));
	SMM_HIP_TRY(hipStreamSynchronize(s));
	return SMM_HIP_OK;
}

}  // extern "C"
