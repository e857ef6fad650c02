// test_loader.cpp -- the host side of the drop-in header that needs no GPU: the Matrix Market / dense-text loaders (ref:2507-2669),
// TripletMatrix -> CSR assembly and the error reporting of the hot-path wrappers when there is no device.  Built twice: plain, and
// with -fsanitize=address,undefined (tests/test_cpp_dropin.py runs both under `pytest -m "not gpu"`).
#include "dropin_checks.h"

template <typename T>
static void testNoDeviceIsObservable() {
	// Without a GPU every hot-path call must fail LOUDLY through the channels the header documents -- never silently do nothing
	SMM::TripletMatrix<T> t(2, 2);
	t.addEntry(0, 0, 2);
	t.addEntry(1, 1, 4);
	SMM::CSRMatrix<T> m(t);
	T x[2] = {1, 1}, y[2] = {7, 7};
	m.rMult(x, y);
	if (SMM::lastHipStatus() == SMM_HIP_OK) {  // a GPU is present: the product ran
		CHECK(y[0] == T(2) && y[1] == T(4));
		return;
	}
	CHECK(SMM::lastHipStatus() == SMM_HIP_ERR_NO_DEVICE);
	CHECK(std::isnan(y[0]) && std::isnan(y[1]));
	SMM::Vector<T> a(4, 1), b(4, 2);
	CHECK(std::isnan(a * b));
	T rhs[2] = {2, 4}, sol[2] = {0, 0};
	CHECK(SMM::ConjugateGradient<T>(m, rhs, sol, sol, -1, T(1e-6)) == SMM::SolverStatus::DIVERGED);
	CHECK(SMM::lastHipStatus() == SMM_HIP_ERR_NO_DEVICE);
	CHECK(SMM::BiCGStab<T>(m, rhs, sol, -1, T(1e-6)) == SMM::SolverStatus::DIVERGED);
	auto sgs = m.template getPreconditioner<SMM::SolverPreconditioner::SYMMETRIC_GAUS_SEIDEL>();
	CHECK(sgs.apply(rhs, sol) != 0);
}

int main() {
	testLoader<float>(false);
	testLoader<double>(false);
	testNoDeviceIsObservable<float>();
	testNoDeviceIsObservable<double>();
	std::printf("%d checks, %d failed\n", g_checks, g_failed);
	return g_failed ? 1 : 0;
}
