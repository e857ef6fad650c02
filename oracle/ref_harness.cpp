// ref_harness.cpp -- thin extern "C" shell around the REAL reference header, used only to pin the oracle.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes
// /root/reference/include/sparse_matrix_math.h by path at build time (oracle/Makefile, target _ref) and is
// compiled into oracle/_ref/libsmm_ref.so, which is git-ignored.  It builds only where /root/reference is
// mounted (the build container); tests that need it skip when the library is absent.
//
// Compile recipe (the header does not build with g++ or plain clang):
//   /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O2 -fdelayed-template-parsing -ffp-contract=off
// Single-threaded reference only: oneTBB headers are not installed, so SMM_MULTITHREADING cannot be built.
//
// CSRMatrix can only be filled through a std::map-backed TripletMatrix in the reference; to load raw CSR
// arrays the harness opens the private members with the usual test-only trick (std headers first).
#include <algorithm>
#include <cassert>
#include <cctype>
#include <cinttypes>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include <memory>
#include <unordered_map>
#include <utility>
#include <vector>

#define private public
#include "sparse_matrix_math.h"
#undef private

namespace {

template <typename T>
SMM::CSRMatrix<T>* makeCsr(int rows, int cols, const int* start, const int* positions, const T* values) {
	auto* m = new SMM::CSRMatrix<T>();
	const int nnz = start[rows];
	m->values.reset(new T[nnz > 0 ? nnz : 1]);
	m->positions.reset(new int[nnz > 0 ? nnz : 1]);
	m->start.reset(new int[rows + 1]);
	std::copy(values, values + nnz, m->values.get());
	std::copy(positions, positions + nnz, m->positions.get());
	std::copy(start, start + rows + 1, m->start.get());
	m->denseRowCount = rows;
	m->denseColCount = cols;
	// firstActiveStart exactly as fillArrays leaves it (include/sparse_matrix_math.h:1619-1628)
	int first = -1;
	for (int i = 0; i < rows; ++i) {
		if (first == -1 && start[i + 1] != 0) {
			first = i;
		}
	}
	m->firstActiveStart = first == -1 ? rows : first;
	return m;
}

// Jacobi is not in the reference; this harness-side type plugs into the reference's BiCGStab template
// through the preconditioner concept `int apply(const T*, T*) const` (:2218, :2235, :2251).
template <typename T>
struct HarnessJacobi {
	const T* diag;
	int rows;
	int apply(const T* rhs, T* x) const noexcept {
		for (int i = 0; i < rows; ++i) {
			x[i] = rhs[i] / diag[i];
		}
		return 0;
	}
};

template <typename T>
void spmv(void* h, int op, const T* lhs, const T* x, T* out) {
	auto* m = static_cast<SMM::CSRMatrix<T>*>(h);
	if (op == 0) {
		m->rMult(x, out);
	} else if (op == 1) {
		m->rMultAdd(lhs, x, out);
	} else {
		m->rMultSub(lhs, x, out);
	}
}

template <typename T>
int bicgstab(void* h, T* b, T* x, int maxIterations, T eps, int precond, const T* diag) {
	auto* m = static_cast<SMM::CSRMatrix<T>*>(h);
	if (precond == 0) {
		return static_cast<int>(SMM::BiCGStab<T>(*m, b, x, maxIterations, eps));
	}
	if (precond == 3) {
		using SGS = typename SMM::CSRMatrix<T>::SGSPreconditioner;
		const SGS M = m->template getPreconditioner<SMM::SolverPreconditioner::SYMMETRIC_GAUS_SEIDEL>();
		return static_cast<int>(SMM::BiCGStab<SGS, T>(*m, b, x, maxIterations, eps, M));
	}
	if (precond == 1) {
		HarnessJacobi<T> M{diag, m->getDenseRowCount()};
		return static_cast<int>(SMM::BiCGStab<HarnessJacobi<T>, T>(*m, b, x, maxIterations, eps, M));
	}
	return -1;
}

// The reference's own BiCGStab template with the reference's own SGSPreconditioner taken from ANOTHER matrix (`hm`): with hm = the
// block-diagonal part of A this is what SMM_PRECOND_BLOCK_SGS must reproduce.  The template takes any preconditioner object (:2191-2199).
template <typename T>
int bicgstabSgsOf(void* h, void* hm, T* b, T* x, int maxIterations, T eps) {
	auto* a = static_cast<SMM::CSRMatrix<T>*>(h);
	auto* m = static_cast<SMM::CSRMatrix<T>*>(hm);
	using SGS = typename SMM::CSRMatrix<T>::SGSPreconditioner;
	const SGS M = m->template getPreconditioner<SMM::SolverPreconditioner::SYMMETRIC_GAUS_SEIDEL>();
	return static_cast<int>(SMM::BiCGStab<SGS, T>(*a, b, x, maxIterations, eps, M));
}

}  // namespace

extern "C" {

int ref_bicgstab_sgs_of_f32(void* h, void* hm, float* b, float* x, int maxIterations, float eps) {
	return bicgstabSgsOf<float>(h, hm, b, x, maxIterations, eps);
}
int ref_bicgstab_sgs_of_f64(void* h, void* hm, double* b, double* x, int maxIterations, double eps) {
	return bicgstabSgsOf<double>(h, hm, b, x, maxIterations, eps);
}

void* ref_csr_create_f32(int rows, int cols, const int* start, const int* positions, const float* values) {
	return makeCsr<float>(rows, cols, start, positions, values);
}
void* ref_csr_create_f64(int rows, int cols, const int* start, const int* positions, const double* values) {
	return makeCsr<double>(rows, cols, start, positions, values);
}
void ref_csr_destroy_f32(void* h) { delete static_cast<SMM::CSRMatrix<float>*>(h); }
void ref_csr_destroy_f64(void* h) { delete static_cast<SMM::CSRMatrix<double>*>(h); }

void ref_spmv_f32(void* h, int op, const float* lhs, const float* x, float* out) { spmv<float>(h, op, lhs, x, out); }
void ref_spmv_f64(void* h, int op, const double* lhs, const double* x, double* out) { spmv<double>(h, op, lhs, x, out); }

float ref_dot_f32(int n, const float* a, const float* b) {
	SMM::Vector<float> va(n), vb(n);
	std::copy(a, a + n, va.begin());
	std::copy(b, b + n, vb.begin());
	return va * vb;
}
double ref_dot_f64(int n, const double* a, const double* b) {
	SMM::Vector<double> va(n), vb(n);
	std::copy(a, a + n, va.begin());
	std::copy(b, b + n, vb.begin());
	return va * vb;
}

int ref_cg_f32(void* h, const float* b, const float* x0, float* x, int maxIterations, float eps) {
	return static_cast<int>(SMM::ConjugateGradient<float>(*static_cast<SMM::CSRMatrix<float>*>(h), b, x0, x, maxIterations, eps));
}
int ref_cg_f64(void* h, const double* b, const double* x0, double* x, int maxIterations, double eps) {
	return static_cast<int>(SMM::ConjugateGradient<double>(*static_cast<SMM::CSRMatrix<double>*>(h), b, x0, x, maxIterations, eps));
}

int ref_bicgstab_f32(void* h, float* b, float* x, int maxIterations, float eps, int precond, const float* diag) {
	return bicgstab<float>(h, b, x, maxIterations, eps, precond, diag);
}
int ref_bicgstab_f64(void* h, double* b, double* x, int maxIterations, double eps, int precond, const double* diag) {
	return bicgstab<double>(h, b, x, maxIterations, eps, precond, diag);
}

int ref_bicgsymmetric_f32(void* h, float* b, float* x, int maxIterations, float eps) {
	return static_cast<int>(SMM::BiCGSymmetric<float>(*static_cast<SMM::CSRMatrix<float>*>(h), b, x, maxIterations, eps));
}
int ref_bicgsymmetric_f64(void* h, double* b, double* x, int maxIterations, double eps) {
	return static_cast<int>(SMM::BiCGSymmetric<double>(*static_cast<SMM::CSRMatrix<double>*>(h), b, x, maxIterations, eps));
}

int ref_sgs_apply_f32(void* h, const float* rhs, float* x) {
	auto* m = static_cast<SMM::CSRMatrix<float>*>(h);
	return m->getPreconditioner<SMM::SolverPreconditioner::SYMMETRIC_GAUS_SEIDEL>().apply(rhs, x);
}
int ref_sgs_apply_f64(void* h, const double* rhs, double* x) {
	auto* m = static_cast<SMM::CSRMatrix<double>*>(h);
	return m->getPreconditioner<SMM::SolverPreconditioner::SYMMETRIC_GAUS_SEIDEL>().apply(rhs, x);
}

// IC0: factor values are copied out (ic0val, nnz long) so the oracle's factorization can be compared too
int ref_ic0_f32(void* h, float* ic0val, const float* rhs, float* x) {
	auto* m = static_cast<SMM::CSRMatrix<float>*>(h);
	SMM::CSRMatrix<float>::IC0Preconditioner M(*m);
	const int err = M.init();
	if (err) return err;
	std::copy(M.ic0Val.get(), M.ic0Val.get() + m->getNonZeroCount(), ic0val);
	return M.apply(rhs, x);
}
int ref_ic0_f64(void* h, double* ic0val, const double* rhs, double* x) {
	auto* m = static_cast<SMM::CSRMatrix<double>*>(h);
	SMM::CSRMatrix<double>::IC0Preconditioner M(*m);
	const int err = M.init();
	if (err) return err;
	std::copy(M.ic0Val.get(), M.ic0Val.get() + m->getNonZeroCount(), ic0val);
	return M.apply(rhs, x);
}

int ref_pcg_ic0_f32(void* h, const float* b, const float* x0, float* x, int maxIterations, float eps) {
	auto* m = static_cast<SMM::CSRMatrix<float>*>(h);
	SMM::CSRMatrix<float>::IC0Preconditioner M(*m);
	if (M.init()) return -1;
	return static_cast<int>(SMM::ConjugateGradient<float>(*m, b, x0, x, maxIterations, eps, M));
}
int ref_pcg_ic0_f64(void* h, const double* b, const double* x0, double* x, int maxIterations, double eps) {
	auto* m = static_cast<SMM::CSRMatrix<double>*>(h);
	SMM::CSRMatrix<double>::IC0Preconditioner M(*m);
	if (M.init()) return -1;
	return static_cast<int>(SMM::ConjugateGradient<double>(*m, b, x0, x, maxIterations, eps, M));
}

// Reference Matrix Market loader (symmetric only, :2531-2609): returns the CSR arrays of the loaded matrix
// so fixtures can be cut from the reference's own test assets.  Caller passes capacity; returns nnz or <0.
int ref_load_mtx_f64(const char* path, int* rows, int* cols, int cap, int* start, int* positions, double* values) {
	SMM::CSRMatrix<double> m;
	const SMM::MatrixLoadStatus st = SMM::loadMatrix(path, m);
	if (st != SMM::MatrixLoadStatus::SUCCESS) return -1 - static_cast<int>(st);
	const int nnz = m.getNonZeroCount();
	*rows = m.getDenseRowCount();
	*cols = m.getDenseColCount();
	if (nnz > cap) return -100;
	std::copy(m.start.get(), m.start.get() + *rows + 1, start);
	std::copy(m.positions.get(), m.positions.get() + nnz, positions);
	std::copy(m.values.get(), m.values.get() + nnz, values);
	return nnz;
}

int ref_version(void) { return SMM_MAJOR_VERSION * 10000 + SMM_MINOR_VERSION * 100 + SMM_PATCH_VERSION; }

}  // extern "C"
