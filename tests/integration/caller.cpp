// A user program written against the REFERENCE header (vasil-pashov/sparse_matrix_math, include/sparse_matrix_math.h) -- nothing in it
// knows about the GPU.  tests/test_integration_patch.py compiles it twice against the patched header (tests/integration/smm_with_hip.patch
// = INTEGRATION.md section B): without -DSMM_WITH_HIP the patch is inert and the reference's own CPU code answers; with it every call
// below goes through libsmm_hip.so, and the only addition is the status line (SMM::lastHipStatus() exists only in the patched build).
#include <cmath>
#include <cstdio>

#include "sparse_matrix_math.h"

#if defined(SMM_WITH_HIP)
#define HIP_STATUS() SMM::lastHipStatus()
#else
#define HIP_STATUS() 0
#endif

int main() {
	const int n = 12;  // 1-D Poisson: tridiagonal 2, -1
	SMM::TripletMatrix<double> t(n, n);
	for (int i = 0; i < n; ++i) {
		t.addEntry(i, i, 2.0);
		if (i > 0) t.addEntry(i, i - 1, -1.0);
		if (i + 1 < n) t.addEntry(i, i + 1, -1.0);
	}
	SMM::CSRMatrix<double> m(t);
	SMM::Vector<double> ones(n, 1.0), b(n, 0.0), x(n, 0.0);
	m.rMult(ones, b);  // b = A * 1 = (1, 0, ..., 0, 1)
	std::printf("rMult b0 %.17g b1 %.17g hip %d\n", b[0], b[1], HIP_STATUS());
	const double dot = b * ones;
	std::printf("dot %.17g hip %d\n", dot, HIP_STATUS());
	const SMM::SolverStatus cg = SMM::ConjugateGradient<double>(m, b, x, x, -1, 1e-12);
	std::printf("cg status %d x0 %.12f hip %d\n", static_cast<int>(cg), x[0], HIP_STATUS());
	x.fill(0.0);
	const SMM::SolverStatus bi = SMM::BiCGStab<double>(m, b, x, -1, 1e-12);
	std::printf("bicgstab status %d x0 %.12f hip %d\n", static_cast<int>(bi), x[0], HIP_STATUS());
	x.fill(0.0);
	using SGS = SMM::CSRMatrix<double>::SGSPreconditioner;
	const SGS M = m.getPreconditioner<SMM::SolverPreconditioner::SYMMETRIC_GAUS_SEIDEL>();
	const SMM::SolverStatus bs = SMM::BiCGStab<SGS, double>(m, b, x, -1, 1e-12, M);
	std::printf("bicgstab+sgs status %d x0 %.12f hip %d\n", static_cast<int>(bs), x[0], HIP_STATUS());
	SMM::Vector<double> y(n, 0.0);
	const int applied = M.apply(b, y);
	std::printf("sgs apply rc %d y0 %.12f hip %d\n", applied, y[0], HIP_STATUS());
	m.updateEntry(0, 0, 4.0);  // an edit: the device mirror must not serve stale values
	m.rMult(ones, b);
	std::printf("rMult after edit b0 %.17g hip %d\n", b[0], HIP_STATUS());
	return 0;
}
