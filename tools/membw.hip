// tools/membw.hip -- read-bandwidth calibration of one MI355X: what a pure streaming read of 4 GB reaches, as a function of
// load flavour, loads in flight per lane, workgroups per CU and number of arrays.  The SpMV / dot kernels are judged
// against the best line of this table, not only against the 8 TB/s spec number.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/membw tools/membw.hip && gpurun_out/membw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                        \
	do {                                                                                \
		hipError_t e = (x);                                                             \
		if (e != hipSuccess) {                                                          \
			std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
			std::exit(1);                                                               \
		}                                                                               \
	} while (0)

// each workgroup owns contiguous chunks of U * 256 * 16 bytes, grid-strided; NT selects non-temporal loads
template <int U, bool NT>
__global__ __launch_bounds__(256) void readKernel(const f32x4* __restrict__ a, long long n16, float* __restrict__ sink) {
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
	const long long chunk = static_cast<long long>(U) * 256;
	for (long long base = static_cast<long long>(blockIdx.x) * chunk; base < n16; base += static_cast<long long>(gridDim.x) * chunk) {
		f32x4 v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const long long i = base + u * 256 + threadIdx.x;
			if (i < n16) {
				v[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
			} else {
				v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
			}
		}
#pragma unroll
		for (int u = 0; u < U; ++u) acc += v[u];
	}
	const float s = acc.x + acc.y + acc.z + acc.w;
	if (s == 123.456f) sink[0] = s;  // never true for the data used; keeps the loads alive
}

// two arrays read in lock step (the SpMV's values[] and positions[])
template <int U, bool NT>
__global__ __launch_bounds__(256) void read2Kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, long long n16, float* __restrict__ sink) {
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
	const long long chunk = static_cast<long long>(U) * 256;
	for (long long base = static_cast<long long>(blockIdx.x) * chunk; base < n16; base += static_cast<long long>(gridDim.x) * chunk) {
		f32x4 v[U], w[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const long long i = base + u * 256 + threadIdx.x;
			const bool ok = i < n16;
			v[u] = ok ? (NT ? __builtin_nontemporal_load(a + i) : a[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
			w[u] = ok ? (NT ? __builtin_nontemporal_load(b + i) : b[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
		}
#pragma unroll
		for (int u = 0; u < U; ++u) acc += v[u] * w[u];
	}
	const float s = acc.x + acc.y + acc.z + acc.w;
	if (s == 123.456f) sink[0] = s;
}

// the CG x/r update (4 reads, 2 writes, one partial sum) written the way csrc/smm_solvers.hip writes its update kernels
// (one element per lane per trip) and as 16-byte accesses with U trips in flight
typedef double f64x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void xrScalar(long long n, double alpha, const double* __restrict__ p, const double* __restrict__ Ap, const double* xcur,
                                               double* x, double* __restrict__ r, double* __restrict__ parts) {
	double acc = 0;
	for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * 256) {
		x[i] = alpha * p[i] + xcur[i];
		const double ri = -alpha * Ap[i] + r[i];
		r[i] = ri;
		acc += ri * ri;
	}
	if (acc == 123.456) parts[0] = acc;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void xrVec(long long n, double alpha, const double* __restrict__ p, const double* __restrict__ Ap, const double* xcur,
                                            double* x, double* __restrict__ r, double* __restrict__ parts) {
	double acc = 0;
	const long long nv = n / 2;
	const f64x2* pv = reinterpret_cast<const f64x2*>(p);
	const f64x2* av = reinterpret_cast<const f64x2*>(Ap);
	const f64x2* cv = reinterpret_cast<const f64x2*>(xcur);
	f64x2* xv = reinterpret_cast<f64x2*>(x);
	f64x2* rv = reinterpret_cast<f64x2*>(r);
	const long long chunk = static_cast<long long>(U) * 256;
	for (long long base = static_cast<long long>(blockIdx.x) * chunk; base < nv; base += static_cast<long long>(gridDim.x) * chunk) {
		f64x2 a[U], b[U], c[U], d[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const long long i = base + u * 256 + threadIdx.x;
			if (i < nv) {
				a[u] = NT ? __builtin_nontemporal_load(pv + i) : pv[i];
				b[u] = NT ? __builtin_nontemporal_load(av + i) : av[i];
				c[u] = NT ? __builtin_nontemporal_load(cv + i) : cv[i];
				d[u] = NT ? __builtin_nontemporal_load(rv + i) : rv[i];
			}
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const long long i = base + u * 256 + threadIdx.x;
			if (i < nv) {
				const f64x2 xn = alpha * a[u] + c[u];
				const f64x2 rn = -alpha * b[u] + d[u];
				if (NT) {
					__builtin_nontemporal_store(xn, xv + i);
					__builtin_nontemporal_store(rn, rv + i);
				} else {
					xv[i] = xn;
					rv[i] = rn;
				}
				acc += rn.x * rn.x;
				acc += rn.y * rn.y;
			}
		}
	}
	if (acc == 123.456) parts[0] = acc;
}

// The SpMV STREAM kernel's skeleton without its gathers: a persistent workgroup walks tiles of NV * 1024 elements of two
// arrays; the tile is fetched one tile ahead into registers (NV 16-byte non-temporal loads per lane and array), stored to
// LDS, and after a barrier every lane sums a strided slice of it out of LDS (like lane = row walking its row).  MODE 0: no
// LDS (sum straight from the registers); MODE 1: through LDS with the two barriers; XCD: deal contiguous eighths to
// blockIdx % 8 like the SpMV does, otherwise tiles are interleaved over all workgroups.
template <int NV, int MODE, bool XCD>
__global__ __launch_bounds__(256) void tileKernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, int nTiles, float* __restrict__ sink,
                                                  double* __restrict__ y = nullptr) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	float* sa = lds;
	float* sb = lds + NV * 1024 + 16;
	const int t = threadIdx.x;
	int tile, step, end;
	if (XCD) {
		const int g = blockIdx.x % 8;
		const int slots = (gridDim.x - g + 7) / 8;
		const int per = (nTiles + 7) / 8;
		tile = g * per + blockIdx.x / 8;
		step = slots;
		end = min(nTiles, (g + 1) * per);
	} else {
		tile = blockIdx.x;
		step = gridDim.x;
		end = nTiles;
	}
	f32x4 ra[NV], rb[NV];
	auto load = [&](int tl) {
		const long long base = static_cast<long long>(tl) * NV * 256;
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			ra[v] = __builtin_nontemporal_load(a + base + v * 256 + t);
			rb[v] = __builtin_nontemporal_load(b + base + v * 256 + t);
		}
	};
	float acc = 0.f;
	if (tile < end) load(tile);
	while (tile < end) {
		const int next = tile + step;
		if (MODE == 0) {
			f32x4 ca[NV], cb[NV];
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				ca[v] = ra[v];
				cb[v] = rb[v];
			}
			if (next < end) load(next);
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				const f32x4 p = ca[v] * cb[v];
				acc += p.x + p.y + p.z + p.w;
			}
		} else {
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				*reinterpret_cast<f32x4*>(sa + 4 * (v * 256 + t)) = ra[v];
				*reinterpret_cast<f32x4*>(sb + 4 * (v * 256 + t)) = rb[v];
			}
			__syncthreads();
			if (next < end) load(next);
			// lane = "row" of NV * 4 consecutive entries (odd stride would need padding; 4 * NV words: conflicts like a real row length)
			const int k0 = t * (NV * 4);
#pragma unroll
			for (int u = 0; u < NV * 4; ++u) acc += sa[k0 + u] * sb[k0 + u];
			// MODE 2 / 3: one 8-byte result per lane and tile, like the SpMV's out[] (plain / non-temporal store)
			if (MODE == 2) y[static_cast<long long>(tile) * 256 + t] = acc;
			if (MODE == 3) __builtin_nontemporal_store(static_cast<double>(acc), y + static_cast<long long>(tile) * 256 + t);
			if (MODE == 4 || MODE == 5) {
				// results meet in LDS (after everybody is done with the tile), then 128 lanes write the 2 KB with 16-byte stores
				__syncthreads();
				reinterpret_cast<double*>(sa)[t] = acc;
				__syncthreads();
				if (t < 128) {
					const f64x2 v = reinterpret_cast<const f64x2*>(sa)[t];
					f64x2* dst = reinterpret_cast<f64x2*>(y + static_cast<long long>(tile) * 256) + t;
					if (MODE == 5) __builtin_nontemporal_store(v, dst);
					else *dst = v;
				}
			}
			__syncthreads();
		}
		tile = next;
	}
	if (acc == 123.456f) sink[0] = acc;
}


// The same skeleton with a workgroup taking KC CONSECUTIVE tiles at a time (instead of every slots-th tile) and, WMODE 1, keeping their
// results in LDS until the KC * 2 KB of y they cover can be written in one go with 16-byte non-temporal stores (WMODE 0: 8 bytes per lane
// after every tile, as the SpMV does).  Question: is the high price of the SpMV's out[] stream (profiles/r02/membw.txt) a matter of how
// large and how contiguous the written pieces are?
template <int NV, int KC, int WMODE>
__global__ __launch_bounds__(256) void tileChunkKernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, int nTiles, float* __restrict__ sink,
                                                       double* __restrict__ y) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	float* sa = lds;
	float* sb = lds + NV * 1024 + 16;
	double* sy = reinterpret_cast<double*>(lds + 2 * (NV * 1024 + 16));
	const int t = threadIdx.x;
	const int g = blockIdx.x % 8;
	const int slots = (gridDim.x - g + 7) / 8;
	const int per = (nTiles + 7) / 8;
	const int s = blockIdx.x / 8;
	const int end = min(nTiles, (g + 1) * per);
	auto tileOf = [&](int j) { return g * per + ((j / KC) * slots + s) * KC + j % KC; };
	f32x4 ra[NV], rb[NV];
	auto load = [&](int tl) {
		const long long base = static_cast<long long>(tl) * NV * 256;
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			ra[v] = __builtin_nontemporal_load(a + base + v * 256 + t);
			rb[v] = __builtin_nontemporal_load(b + base + v * 256 + t);
		}
	};
	float acc = 0.f;
	int j = 0;
	int tile = tileOf(0);
	if (tile < end) load(tile);
	while (tile < end) {
		const int next = tileOf(j + 1);
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			*reinterpret_cast<f32x4*>(sa + 4 * (v * 256 + t)) = ra[v];
			*reinterpret_cast<f32x4*>(sb + 4 * (v * 256 + t)) = rb[v];
		}
		__syncthreads();
		if (next < end) load(next);
		const int k0 = t * (NV * 4);
#pragma unroll
		for (int u = 0; u < NV * 4; ++u) acc += sa[k0 + u] * sb[k0 + u];
		if (WMODE == 0) {
			__builtin_nontemporal_store(static_cast<double>(acc), y + static_cast<long long>(tile) * 256 + t);
		} else {
			const int slot = j % KC;
			sy[slot * 256 + t] = acc;
			const bool last = slot == KC - 1 || next >= end || next != tile + 1;
			if (last) {
				__syncthreads();
				const int pieces = (slot + 1) * 128;  // 16-byte pieces of the chunk's results
				f64x2* dst = reinterpret_cast<f64x2*>(y + static_cast<long long>(tile - slot) * 256);
				for (int q = t; q < pieces; q += 256) __builtin_nontemporal_store(reinterpret_cast<const f64x2*>(sy)[q], dst + q);
			}
		}
		__syncthreads();
		tile = next;
		++j;
	}
	if (acc == 123.456f) sink[0] = acc;
}

template <typename F>
static double timeIt(F launch, int reps) {
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int i = 0; i < 3; ++i) launch();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	for (int i = 0; i < reps; ++i) launch();
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0.f;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	CHECK(hipEventDestroy(e0));
	CHECK(hipEventDestroy(e1));
	return ms / reps;
}

template <int U, bool NT>
static void runOne(const f32x4* a, const f32x4* b, long long n16, float* sink, int cus) {
	for (int perCU : {2, 4, 8, 16}) {
		const int grid = cus * perCU;
		const double ms1 = timeIt([&] { readKernel<U, NT><<<grid, 256>>>(a, 2 * n16, sink); }, 10);
		const double ms2 = timeIt([&] { read2Kernel<U, NT><<<grid, 256>>>(a, b, n16, sink); }, 10);
		const double bytes = 2.0 * n16 * 16;
		std::printf("U=%d %s wgs/CU=%2d : one array %7.3f ms %7.1f GB/s | two arrays %7.3f ms %7.1f GB/s\n", U, NT ? "nt   " : "plain", perCU, ms1,
		            bytes / ms1 / 1e6, ms2, bytes / ms2 / 1e6);
	}
}

int main() {
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const long long n16 = (2LL << 30) / 16;  // 2 GB per array, two arrays (contiguous: the one-array runs read both as one)
	f32x4* a = nullptr;
	float* sink = nullptr;
	CHECK(hipMalloc(&a, 2 * n16 * 16));
	CHECK(hipMalloc(&sink, 64));
	CHECK(hipMemset(a, 0, 2 * n16 * 16));
	const f32x4* b = a + n16;
	std::printf("%s, %d CUs; reading 4 GB per launch\n", prop.name, cus);
	runOne<1, false>(a, b, n16, sink, cus);
	runOne<2, false>(a, b, n16, sink, cus);
	runOne<4, false>(a, b, n16, sink, cus);
	runOne<8, false>(a, b, n16, sink, cus);
	runOne<2, true>(a, b, n16, sink, cus);
	runOne<4, true>(a, b, n16, sink, cus);
	runOne<8, true>(a, b, n16, sink, cus);
	// working sets that fit the 256 MB Infinity Cache (and, at 16 MB, the eight 4 MB L2s): what the path from beyond L2 delivers
	// when HBM is not involved
	for (long long mb : {16LL, 64LL, 128LL, 192LL, 512LL}) {
		const long long m16 = mb * 1024 * 1024 / 16;
		const double ms = timeIt([&] { readKernel<8, true><<<cus * 4, 256>>>(a, m16, sink); }, 50);
		const double ms2 = timeIt([&] { readKernel<8, false><<<cus * 4, 256>>>(a, m16, sink); }, 50);
		std::printf("re-reading %4lld MB: nt %7.4f ms %8.1f GB/s | plain %7.4f ms %8.1f GB/s\n", mb, ms, m16 * 16.0 / ms / 1e6, ms2, m16 * 16.0 / ms2 / 1e6);
	}
	{
		const double bytes = 2.0 * n16 * 16;
		auto tiles = [&](auto kern, int nv, int perCU, const char* name) {
			const int nTiles = static_cast<int>(n16 / (nv * 256));
			const size_t ldsBytes = (2 * (nv * 1024 + 16)) * sizeof(float);
			const double ms = timeIt([&] { kern<<<cus * perCU, 256, ldsBytes>>>(a, b, nTiles, sink, nullptr); }, 10);
			std::printf("tile skeleton %-28s NV=%d wgs/CU=%d: %7.3f ms %7.1f GB/s\n", name, nv, perCU, ms, bytes / ms / 1e6);
		};
		for (int perCU : {2, 4}) {
			tiles(tileKernel<4, 0, false>, 4, perCU, "registers only, interleaved");
			tiles(tileKernel<4, 0, true>, 4, perCU, "registers only, XCD eighths");
			tiles(tileKernel<4, 1, false>, 4, perCU, "through LDS, interleaved");
			tiles(tileKernel<4, 1, true>, 4, perCU, "through LDS, XCD eighths");
		}
		{
			// the same skeleton writing 8 bytes per lane and tile (NV = 2: 16 KB read per 2 KB written, the 7-point fp64 stencil's ratio)
			double* y = nullptr;
			CHECK(hipMalloc(&y, (n16 / 512 + 1) * 256 * sizeof(double)));
			for (int perCU : {4, 6}) {
				const int nTiles = static_cast<int>(n16 / 512);
				const size_t ldsBytes = (2 * (2 * 1024 + 16)) * sizeof(float);
				const double wbytes = nTiles * 256.0 * 8;
				const double m1 = timeIt([&] { tileKernel<2, 1, true><<<cus * perCU, 256, ldsBytes>>>(a, b, nTiles, sink, y); }, 10);
				const double m2 = timeIt([&] { tileKernel<2, 2, true><<<cus * perCU, 256, ldsBytes>>>(a, b, nTiles, sink, y); }, 10);
				const double m3 = timeIt([&] { tileKernel<2, 3, true><<<cus * perCU, 256, ldsBytes>>>(a, b, nTiles, sink, y); }, 10);
				const double m4 = timeIt([&] { tileKernel<2, 4, true><<<cus * perCU, 256, ldsBytes>>>(a, b, nTiles, sink, y); }, 10);
				const double m5 = timeIt([&] { tileKernel<2, 5, true><<<cus * perCU, 256, ldsBytes>>>(a, b, nTiles, sink, y); }, 10);
				std::printf("tile skeleton NV=2 wgs/CU=%d: read only %.3f ms | + %.2f GB written, plain stores %.3f ms | non-temporal stores %.3f ms | 16-byte stores via LDS %.3f ms | same nt %.3f ms\n", perCU, m1, wbytes / 1e9, m2, m3, m4, m5);
			}
			for (int perCU : {4, 6}) {
				const int nTiles = static_cast<int>(n16 / 512);
				const double wbytes = nTiles * 256.0 * 8;
				auto run = [&](auto kern, int kc) {
					const size_t ldsBytes = (2 * (2 * 1024 + 16)) * sizeof(float) + static_cast<size_t>(kc) * 256 * sizeof(double);
					return timeIt([&] { kern<<<cus * perCU, 256, ldsBytes>>>(a, b, nTiles, sink, y); }, 10);
				};
				std::printf("tile skeleton NV=2 wgs/CU=%d, %.2f GB written, workgroup takes KC consecutive tiles; 8-byte nt stores per tile / KC x 2 KB from LDS at once:"
				            "  KC=1 %.3f / %.3f ms | KC=4 %.3f / %.3f ms | KC=8 %.3f / %.3f ms | KC=16 %.3f / %.3f ms\n", perCU, wbytes / 1e9,
				            run(tileChunkKernel<2, 1, 0>, 1), run(tileChunkKernel<2, 1, 1>, 1), run(tileChunkKernel<2, 4, 0>, 4), run(tileChunkKernel<2, 4, 1>, 4),
				            run(tileChunkKernel<2, 8, 0>, 8), run(tileChunkKernel<2, 8, 1>, 8), run(tileChunkKernel<2, 16, 0>, 16), run(tileChunkKernel<2, 16, 1>, 16));
			}
			CHECK(hipFree(y));
		}
		for (int perCU : {4, 8}) {
			tiles(tileKernel<2, 1, true>, 2, perCU, "through LDS, XCD eighths");
			tiles(tileKernel<2, 1, false>, 2, perCU, "through LDS, interleaved");
		}
	}
	{
		// 4 vectors of 1 GiB (134 M doubles, the 512^3 grid): 4 reads + 2 writes = 6.44 GB per launch
		const long long n = 134217728LL;
		double* v = reinterpret_cast<double*>(a);
		double *p = v, *Ap = v + n, *x = v + 2 * n, *r = v + 3 * n;
		double* parts = reinterpret_cast<double*>(sink);
		const double bytes = 6.0 * n * 8;
		for (int perCU : {4, 8, 16}) {
			const int grid = cus * perCU;
			const double t0 = timeIt([&] { xrScalar<<<grid, 256>>>(n, 1e-9, p, Ap, x, x, r, parts); }, 5);
			const double t1 = timeIt([&] { xrVec<1, false><<<grid, 256>>>(n, 1e-9, p, Ap, x, x, r, parts); }, 5);
			const double t2 = timeIt([&] { xrVec<2, false><<<grid, 256>>>(n, 1e-9, p, Ap, x, x, r, parts); }, 5);
			const double t3 = timeIt([&] { xrVec<2, true><<<grid, 256>>>(n, 1e-9, p, Ap, x, x, r, parts); }, 5);
			const double t4 = timeIt([&] { xrVec<4, true><<<grid, 256>>>(n, 1e-9, p, Ap, x, x, r, parts); }, 5);
			std::printf("x/r update fp64 n=2^27 wgs/CU=%2d: scalar %.3f ms %.0f GB/s | 16B U1 %.3f ms %.0f | U2 %.3f ms %.0f | U2 nt %.3f ms %.0f | U4 nt %.3f ms %.0f\n", perCU, t0,
			            bytes / t0 / 1e6, t1, bytes / t1 / 1e6, t2, bytes / t2 / 1e6, t3, bytes / t3 / 1e6, t4, bytes / t4 / 1e6);
		}
	}
	CHECK(hipFree(a));
	CHECK(hipFree(sink));
	return 0;
}
