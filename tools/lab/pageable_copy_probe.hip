// pageable_copy_probe.hip -- how long do the first copies between PAGEABLE host memory and the device take in a fresh process?
// (round 5: the second host-pointer solve of a process lost ~20 ms; tools/r05_copy_probe.sh)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
__global__ void touch(double* p, size_t n) {
	for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] += 1.0;
}
int main(int argc, char** argv) {
	const size_t n = argc > 1 ? atol(argv[1]) : 1259712;
	const int warm = argc > 2 ? atoi(argv[2]) : 0;  // 1: a small D2H + H2D round trip first; 2: a same-size round trip on a scratch buffer first
	hipStream_t s;
	hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	double *d1, *d2;
	hipMalloc(&d1, n * 8);
	hipMalloc(&d2, n * 8);
	std::vector<double> big(50 * n, 1.0);  // the "matrix": a large first H2D like csr_create's
	double* dbig;
	hipMalloc(&dbig, big.size() * 8);
	auto t = clk::now();
	hipMemcpyAsync(dbig, big.data(), big.size() * 8, hipMemcpyHostToDevice, s);
	hipStreamSynchronize(s);
	printf("matrix H2D %zu MB: %.3f ms\n", big.size() * 8 >> 20, ms(t));
	if (warm) {
		std::vector<double> w(warm == 1 ? 4096 : n, 0.0);
		t = clk::now();
		hipMemcpyAsync(w.data(), d1, w.size() * 8, hipMemcpyDeviceToHost, s);
		hipStreamSynchronize(s);
		printf("warm D2H %zu KB: %.3f ms\n", w.size() * 8 >> 10, ms(t));
		t = clk::now();
		hipMemcpyAsync(d1, w.data(), w.size() * 8, hipMemcpyHostToDevice, s);
		hipStreamSynchronize(s);
		printf("warm H2D: %.3f ms\n", ms(t));
	}
	const int how = argc > 3 ? atoi(argv[3]) : 0;  // 0 direct (the runtime pins the user's pages), 1 hipHostRegister around the call, 2 staged through a pinned buffer
	double* stage = nullptr;
	if (how == 2) hipHostMalloc(reinterpret_cast<void**>(&stage), 2 * n * 8, hipHostMallocDefault);
	std::vector<double> b(n, 1.0), x(n, 0.0);
	for (int rep = 0; rep < 6; ++rep) {
		std::vector<double> rhs = b;
		std::fill(x.begin(), x.end(), 0.0);
		t = clk::now();
		if (how == 1) {
			hipHostRegister(rhs.data(), n * 8, hipHostRegisterDefault);
			hipHostRegister(x.data(), n * 8, hipHostRegisterDefault);
		}
		if (how == 2) {
			memcpy(stage, rhs.data(), n * 8);
			hipMemcpyAsync(d1, stage, n * 8, hipMemcpyHostToDevice, s);
			memcpy(stage + n, x.data(), n * 8);
			hipMemcpyAsync(d2, stage + n, n * 8, hipMemcpyHostToDevice, s);
		} else {
			hipMemcpyAsync(d1, rhs.data(), n * 8, hipMemcpyHostToDevice, s);
			hipMemcpyAsync(d2, x.data(), n * 8, hipMemcpyHostToDevice, s);
		}
		hipStreamSynchronize(s);
		const double h2d = ms(t);
		t = clk::now();
		for (int k = 0; k < 200; ++k) touch<<<1024, 256, 0, s>>>(d2, n);
		hipStreamSynchronize(s);
		const double work = ms(t);
		t = clk::now();
		if (how == 2) {
			hipMemcpyAsync(stage, d2, n * 8, hipMemcpyDeviceToHost, s);
			hipStreamSynchronize(s);
			memcpy(x.data(), stage, n * 8);
		} else {
			hipMemcpyAsync(x.data(), d2, n * 8, hipMemcpyDeviceToHost, s);
			hipStreamSynchronize(s);
		}
		if (how == 1) {
			hipHostUnregister(rhs.data());
			hipHostUnregister(x.data());
		}
		printf("rep %d: H2D b, x %.3f ms | 200 kernels %.3f ms | D2H x %.3f ms\n", rep, h2d, work, ms(t));
	}
	return 0;
}
