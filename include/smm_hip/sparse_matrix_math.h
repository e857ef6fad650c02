// smm_hip/sparse_matrix_math.h -- the reference's C++ API for the hot path, re-implemented on top of the C ABI of
// libsmm_hip.so (include/smm_hip.h).  A program written against vasil-pashov/sparse_matrix_math's
// include/sparse_matrix_math.h ("ref" below) that only uses
//
//     SMM::Vector, SMM::TripletMatrix, SMM::CSRMatrix (init / rMult / rMultAdd / rMultSub / getters / iteration /
//     getPreconditioner), SMM::SolverStatus, SMM::SolverPreconditioner, SMM::ConjugateGradient (plain and IC0),
//     SMM::BiCGStab (plain and preconditioned), SMM::BiCGSymmetric, SMM::loadMatrix
//
// compiles against this header unchanged and runs those calls on an MI355X: same names, same argument order and meaning,
// same return values (SolverStatus; int != 0 on failure for init / apply).  Matrix assembly (TripletMatrix, CSR arrays)
// stays on the host exactly as in the reference; the CSR arrays are mirrored to the GPU the first time a hot-path call
// needs them.  Nothing here computes on the CPU: every rMult* / solver / apply call goes through libsmm_hip.so.
//
// FAILURES OF THE GPU PATH ARE OBSERVABLE (the reference's signatures have no room for them, so they are reported beside them):
//   * SMM::lastHipStatus() -- the SMM_HIP_* status of the last hot-path call of this thread (0 = ok), text in smm_hip_last_error();
//   * rMult / rMultAdd / rMultSub fill `out` with NaN, Vector::operator* returns NaN, solvers return SolverStatus::DIVERGED
//     (and lastHipStatus() != 0 tells that apart from a numerical divergence), apply() / init() return non-zero;
//   * compile with -DSMM_HIP_ABORT_ON_ERROR to print the message and abort() instead.
//
// Written from the documented behaviour of the reference; no reference source is reproduced here.
// Additions the reference lacks: CSRMatrix::init(rows, cols, start, positions, values) (raw CSR arrays, ref can only be
// filled through a std::map), JacobiPreconditioner, a working ILU0Preconditioner, Matrix Market `general` matrices.
#pragma once

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../smm_hip.h"

#define SMM_MAJOR_VERSION 0
#define SMM_MINOR_VERSION 2
#define SMM_PATCH_VERSION 0

namespace SMM {

// ---- C ABI dispatch on T ------------------------------------------------------------------------------------------
namespace detail {
template <typename T>
struct Abi;
template <>
struct Abi<float> {
	static int create(int r, int c, const int* s, const int* p, const float* v, smm_hip_csr** o) { return smm_hip_csr_create_f32(r, c, s, p, v, o); }
	static int spmv(const smm_hip_csr* m, int op, const float* l, const float* x, float* o) { return smm_hip_spmv_f32(m, op, l, x, o); }
	static int dot(int n, const float* a, const float* b, float* r) { return smm_hip_dot_f32(n, a, b, r); }
	static int cg(const smm_hip_csr* a, const float* b, const float* x0, float* x, int it, float eps, const smm_hip_precond* M, int* st) {
		return smm_hip_cg_f32(a, b, x0, x, it, eps, M, st, nullptr, nullptr);
	}
	static int bicgstab(const smm_hip_csr* a, float* b, float* x, int it, float eps, const smm_hip_precond* M, int* st) {
		return smm_hip_bicgstab_f32(a, b, x, it, eps, M, st, nullptr, nullptr);
	}
	static int bicgsym(const smm_hip_csr* a, float* b, float* x, int it, float eps, int* st) { return smm_hip_bicgsymmetric_f32(a, b, x, it, eps, st, nullptr); }
	static int apply(const smm_hip_precond* M, const float* r, float* x) { return smm_hip_precond_apply_f32(M, r, x); }
};
template <>
struct Abi<double> {
	static int create(int r, int c, const int* s, const int* p, const double* v, smm_hip_csr** o) { return smm_hip_csr_create_f64(r, c, s, p, v, o); }
	static int spmv(const smm_hip_csr* m, int op, const double* l, const double* x, double* o) { return smm_hip_spmv_f64(m, op, l, x, o); }
	static int dot(int n, const double* a, const double* b, double* r) { return smm_hip_dot_f64(n, a, b, r); }
	static int cg(const smm_hip_csr* a, const double* b, const double* x0, double* x, int it, double eps, const smm_hip_precond* M, int* st) {
		return smm_hip_cg_f64(a, b, x0, x, it, eps, M, st, nullptr, nullptr);
	}
	static int bicgstab(const smm_hip_csr* a, double* b, double* x, int it, double eps, const smm_hip_precond* M, int* st) {
		return smm_hip_bicgstab_f64(a, b, x, it, eps, M, st, nullptr, nullptr);
	}
	static int bicgsym(const smm_hip_csr* a, double* b, double* x, int it, double eps, int* st) { return smm_hip_bicgsymmetric_f64(a, b, x, it, eps, st, nullptr); }
	static int apply(const smm_hip_precond* M, const double* r, double* x) { return smm_hip_precond_apply_f64(M, r, x); }
};
inline int& statusSlot() noexcept {
	static thread_local int st = SMM_HIP_OK;
	return st;
}
// every hot-path call funnels its ABI status through here
inline int note(int abi) noexcept {
	statusSlot() = abi;
#ifdef SMM_HIP_ABORT_ON_ERROR
	if (abi != SMM_HIP_OK) {
		std::fprintf(stderr, "smm_hip: error %d: %s\n", abi, smm_hip_last_error());
		std::abort();
	}
#endif
	return abi;
}
template <typename T>
inline void fillNaN(T* p, int n) noexcept {
	for (int i = 0; p && i < n; ++i) p[i] = std::numeric_limits<T>::quiet_NaN();
}
}  // namespace detail

// SMM_HIP_* status of the calling thread's last hot-path call (rMult*, dot / norms, solvers, preconditioner init / apply); 0 = ok
inline int lastHipStatus() noexcept { return detail::statusSlot(); }

// ---- Vector<T> (ref:42-381): host buffer that decays to T*, dot product and norms on the GPU ---------------------------
template <typename T>
class Vector {
public:
	using Iterator = T*;
	using ConstIterator = const T*;
	Vector() noexcept = default;
	explicit Vector(const int size) noexcept : buf(static_cast<size_t>(size > 0 ? size : 0)) {}
	Vector(const int size, const T val) noexcept : buf(static_cast<size_t>(size > 0 ? size : 0), val) {}
	Vector(std::initializer_list<T> l) : buf(l) {}
	void init(const int size) { buf.assign(static_cast<size_t>(size), T()); }
	void init(const int size, const T val) { buf.assign(static_cast<size_t>(size), val); }
	int getSize() const noexcept { return static_cast<int>(buf.size()); }
	operator T*() noexcept { return buf.data(); }
	operator const T*() const noexcept { return buf.data(); }
	T& operator[](const int i) { return buf[static_cast<size_t>(i)]; }
	const T& operator[](const int i) const { return buf[static_cast<size_t>(i)]; }
	Iterator begin() noexcept { return buf.data(); }
	Iterator end() noexcept { return buf.data() + buf.size(); }
	ConstIterator begin() const noexcept { return buf.data(); }
	ConstIterator end() const noexcept { return buf.data() + buf.size(); }
	void fill(const T v) { std::fill(buf.begin(), buf.end(), v); }
	Vector& operator+=(const Vector& o) {
		for (size_t i = 0; i < buf.size(); ++i) buf[i] += o.buf[i];
		return *this;
	}
	Vector& operator-=(const Vector& o) {
		for (size_t i = 0; i < buf.size(); ++i) buf[i] -= o.buf[i];
		return *this;
	}
	// dot product (ref:305-328) and norms (ref:287-303) -- reductions of the hot path, computed on the GPU
	const T operator*(const Vector& o) const {
		T r = T(0);
		if (detail::note(detail::Abi<T>::dot(getSize(), buf.data(), o.buf.data(), &r)) != SMM_HIP_OK) r = std::numeric_limits<T>::quiet_NaN();
		return r;
	}
	T secondNormSquared() const { return (*this) * (*this); }
	T secondNorm() const { return std::sqrt(secondNormSquared()); }

private:
	std::vector<T> buf;
};

// ---- TripletMatrix<T> (ref:383-684): ordered COO assembly; duplicates are summed ---------------------------------------
template <typename T>
class TripletEl {
public:
	TripletEl(int r, int c, T v) : row(r), col(c), value(v) {}
	int getRow() const noexcept { return row; }
	int getCol() const noexcept { return col; }
	T getValue() const noexcept { return value; }

private:
	int row, col;
	T value;
};

template <typename T>
class TripletMatrix {
public:
	TripletMatrix() noexcept = default;
	TripletMatrix(int rows, int cols) noexcept : denseRowCount(rows), denseColCount(cols) {}
	TripletMatrix(int rows, int cols, int /*numTriplets*/) noexcept : denseRowCount(rows), denseColCount(cols) {}
	void init(int rows, int cols, int /*numTriplets*/ = 0) {
		denseRowCount = rows;
		denseColCount = cols;
		data.clear();
	}
	int getNonZeroCount() const noexcept { return static_cast<int>(data.size()); }
	int getDenseRowCount() const noexcept { return denseRowCount; }
	int getDenseColCount() const noexcept { return denseColCount; }
	// entries with the same (row, col) add up (ref:612-617)
	void addEntry(int row, int col, T value) { data[key(row, col)] += value; }
	T getValue(int row, int col) const {
		auto it = data.find(key(row, col));
		return it == data.end() ? T(0) : it->second;
	}
	bool updateEntry(int row, int col, T value) {
		auto it = data.find(key(row, col));
		if (it == data.end()) return false;
		it->second = value;
		return true;
	}
	// row-major, column-ascending traversal
	template <typename F>
	void forEach(F&& f) const {
		for (const auto& kv : data) f(static_cast<int>(kv.first >> 32), static_cast<int>(kv.first & 0xFFFFFFFFu), kv.second);
	}

private:
	static uint64_t key(int row, int col) { return (static_cast<uint64_t>(static_cast<uint32_t>(row)) << 32) | static_cast<uint32_t>(col); }
	std::map<uint64_t, T> data;
	int denseRowCount = 0, denseColCount = 0;
};

// ref:1002-1006; JACOBI and the BLOCK_ forms (ILU0 / SGS of the block-diagonal part of A, smm_hip.h) are additions
enum class SolverPreconditioner { NONE, SYMMETRIC_GAUS_SEIDEL, ILU0, JACOBI, BLOCK_ILU0, BLOCK_SGS };
enum class SolverStatus { SUCCESS = 0, DIVERGED, MAX_ITERATIONS_REACHED };                       // ref:2010-2014

// ---- CSRMatrix<T> (ref:1010-1641) -------------------------------------------------------------------------------------------
template <typename T>
class CSRMatrix {
public:
	using value_type = T;

	class ConstElement {
	public:
		ConstElement(const CSRMatrix* m, int row, int idx) : m(m), row(row), idx(idx) {}
		int getRow() const noexcept { return row; }
		int getCol() const noexcept { return m->positions[idx]; }
		T getValue() const noexcept { return m->values[idx]; }

	private:
		const CSRMatrix* m;
		int row, idx;
	};
	// forward iterator over all stored elements in row-major order (what the reference's tests use: `for (const auto& el : m)`)
	class ConstIterator {
	public:
		ConstIterator(const CSRMatrix* m, int row, int idx) : m(m), row(row), idx(idx) { skipEmpty(); }
		ConstElement operator*() const { return ConstElement(m, row, idx); }
		ConstIterator& operator++() {
			++idx;
			skipEmpty();
			return *this;
		}
		bool operator!=(const ConstIterator& o) const { return idx != o.idx; }
		bool operator==(const ConstIterator& o) const { return idx == o.idx; }

	private:
		void skipEmpty() {
			while (row < m->denseRowCount && idx >= m->start[row + 1]) ++row;
		}
		const CSRMatrix* m;
		int row, idx;
	};

	// Every preconditioner wraps a device-side smm_hip_precond; `int apply(const T* rhs, T* x) const` as in ref:1173-1235.
	class PreconditionerBase {
	public:
		PreconditionerBase(const PreconditionerBase&) = delete;
		PreconditionerBase& operator=(const PreconditionerBase&) = delete;
		PreconditionerBase(PreconditionerBase&& o) noexcept : m(o.m), kind(o.kind), h(o.h) { o.h = nullptr; }
		~PreconditionerBase() { smm_hip_precond_destroy(h); }
		// non-zero on structural failure (missing / tiny diagonal, empty row, non-SPD pivot), like ref:1668-1693
		int init() const noexcept {
			if (h) return 0;
			const smm_hip_csr* dev = m->device();
			if (!dev) return 1;
			return detail::note(smm_hip_precond_create(dev, kind, &h)) == SMM_HIP_OK ? 0 : 1;
		}
		int apply(const T* rhs, T* x) const noexcept {
			if (init()) return 1;
			return detail::note(detail::Abi<T>::apply(h, rhs, x)) == SMM_HIP_OK ? 0 : 1;
		}
		const smm_hip_precond* handle() const noexcept { return init() ? nullptr : h; }

	protected:
		PreconditionerBase(const CSRMatrix& m, int kind) noexcept : m(&m), kind(kind) {}
		const CSRMatrix* m;
		int kind;
		mutable smm_hip_precond* h = nullptr;
	};
	class IDPreconditioner {  // ref:1166-1170
	public:
		int apply(const T*, T*) const noexcept { return 0; }
		const smm_hip_precond* handle() const noexcept { return nullptr; }
	};
	class SGSPreconditioner : public PreconditionerBase {  // ref:1173-1186
	public:
		SGSPreconditioner(const CSRMatrix& m) noexcept : PreconditionerBase(m, SMM_PRECOND_SGS) {}
		SGSPreconditioner(SGSPreconditioner&&) noexcept = default;
	};
	class JacobiPreconditioner : public PreconditionerBase {  // addition
	public:
		JacobiPreconditioner(const CSRMatrix& m) noexcept : PreconditionerBase(m, SMM_PRECOND_JACOBI) {}
		JacobiPreconditioner(JacobiPreconditioner&&) noexcept = default;
	};
	class ILU0Preconditioner : public PreconditionerBase {  // ref:1189-1212 (declared there, not usable)
	public:
		ILU0Preconditioner(const CSRMatrix& m) noexcept : PreconditionerBase(m, SMM_PRECOND_ILU0) {}
		ILU0Preconditioner(ILU0Preconditioner&&) noexcept = default;
		int validate() noexcept { return this->init(); }
	};
	class IC0Preconditioner : public PreconditionerBase {  // ref:1216-1235
	public:
		IC0Preconditioner(const CSRMatrix& m) noexcept : PreconditionerBase(m, SMM_PRECOND_IC0) {}
		IC0Preconditioner(IC0Preconditioner&&) noexcept = default;
	};
	// additions: ILU0 / SGS of the block-diagonal part of A (blocks of <= 1024 rows -- bricks of the grid when the matrix is a grid
	// stencil, runs of consecutive rows otherwise --, every block's sweeps cut to <= 16 dependent levels; smm_hip.h,
	// SMM_PRECOND_BLOCK_*, smm_hip_precond_create_block_ex for other sizes / cuts / partitions)
	class BlockILU0Preconditioner : public PreconditionerBase {
	public:
		BlockILU0Preconditioner(const CSRMatrix& m) noexcept : PreconditionerBase(m, SMM_PRECOND_BLOCK_ILU0) {}
		BlockILU0Preconditioner(BlockILU0Preconditioner&&) noexcept = default;
		int validate() noexcept { return this->init(); }
	};
	class BlockSGSPreconditioner : public PreconditionerBase {
	public:
		BlockSGSPreconditioner(const CSRMatrix& m) noexcept : PreconditionerBase(m, SMM_PRECOND_BLOCK_SGS) {}
		BlockSGSPreconditioner(BlockSGSPreconditioner&&) noexcept = default;
	};

	CSRMatrix() noexcept = default;
	CSRMatrix(const TripletMatrix<T>& triplet) noexcept { init(triplet); }
	CSRMatrix(const CSRMatrix&) = delete;
	CSRMatrix& operator=(const CSRMatrix&) = delete;
	CSRMatrix(CSRMatrix&& o) noexcept { *this = std::move(o); }
	CSRMatrix& operator=(CSRMatrix&& o) noexcept {
		release();
		values = std::move(o.values);
		positions = std::move(o.positions);
		start = std::move(o.start);
		denseRowCount = o.denseRowCount;
		denseColCount = o.denseColCount;
		firstActiveStart = o.firstActiveStart;
		dev = o.dev;
		o.dev = nullptr;
		return *this;
	}
	~CSRMatrix() { release(); }

	// ref:1326-1349 / 1606-1641: count per row, prefix sum, scatter in map order (row-major, columns ascending)
	int init(const TripletMatrix<T>& triplet) noexcept {
		release();
		denseRowCount = triplet.getDenseRowCount();
		denseColCount = triplet.getDenseColCount();
		const int nnz = triplet.getNonZeroCount();
		values.reset(new T[nnz > 0 ? nnz : 1]);
		positions.reset(new int[nnz > 0 ? nnz : 1]);
		start.reset(new int[denseRowCount + 1]());
		triplet.forEach([&](int r, int, T) { start[r + 1]++; });
		for (int i = 0; i < denseRowCount; ++i) start[i + 1] += start[i];
		int k = 0;
		triplet.forEach([&](int, int c, T v) {
			positions[k] = c;
			values[k] = v;
			++k;
		});
		computeFirstActive();
		return 0;
	}
	// addition: adopt raw CSR arrays (copied); columns must ascend inside each row
	int init(int rows, int cols, const int* startIn, const int* positionsIn, const T* valuesIn) noexcept {
		release();
		denseRowCount = rows;
		denseColCount = cols;
		const int nnz = startIn[rows];
		values.reset(new T[nnz > 0 ? nnz : 1]);
		positions.reset(new int[nnz > 0 ? nnz : 1]);
		start.reset(new int[rows + 1]);
		std::copy(startIn, startIn + rows + 1, start.get());
		std::copy(positionsIn, positionsIn + nnz, positions.get());
		std::copy(valuesIn, valuesIn + nnz, values.get());
		computeFirstActive();
		return 0;
	}
	int getNonZeroCount() const noexcept { return start ? start[denseRowCount] : 0; }
	int getDenseRowCount() const noexcept { return denseRowCount; }
	int getDenseColCount() const noexcept { return denseColCount; }
	ConstIterator begin() const noexcept { return ConstIterator(this, 0, 0); }
	ConstIterator end() const noexcept { return ConstIterator(this, denseRowCount, getNonZeroCount()); }
	T getValue(int row, int col) const noexcept {
		const int* b = positions.get() + start[row];
		const int* e = positions.get() + start[row + 1];
		const int* it = std::lower_bound(b, e, col);
		return it != e && *it == col ? values[it - positions.get()] : T(0);
	}

	// ---- the hot path: out = op(lhs, A * mult) on the GPU (ref:1458-1515) ----
	void rMult(const T* const mult, T* const res) const noexcept { spmv(SMM_OP_ASSIGN, nullptr, mult, res); }
	void rMultAdd(const T* const lhs, const T* const mult, T* const out) const noexcept { spmv(SMM_OP_ADD, lhs, mult, out); }
	void rMultSub(const T* const lhs, const T* const mult, T* const out) const noexcept { spmv(SMM_OP_SUB, lhs, mult, out); }

	// ref:1643-1651.  WHAT THE KINDS COST ON THE GPU (measured, MI355X, BiCGStab to 1e-8 on the 1.26 M-row convection-diffusion problem of
	// BASELINE config 5; INTEGRATION.md "What a preconditioner costs"): the reference's own kind, SYMMETRIC_GAUS_SEIDEL, and ILU0 are EXACT
	// triangular sweeps -- bit-identical to the sequential loops (ref:1658-1713), and bound by one memory-fabric round trip per dependency
	// level: 79 / 67 iterations but ~200 / ~167 ms, against ~23 ms for 321 iterations with NONE.  They are kept because they are the
	// reference's semantics, not because they are fast.  The kinds that WIN on a GPU are BLOCK_ILU0 / BLOCK_SGS (the same algorithms on
	// the block-diagonal part of A, one wavefront per block: ~18 ms, 105 iterations, create included) and JACOBI (folded into the SpMV
	// rows: the price of NONE).  A caller of getPreconditioner<SYMMETRIC_GAUS_SEIDEL>() (test/cpp/bicgstab.cpp:160-162) gets the slow,
	// exact one -- switching to BLOCK_SGS is a one-word change, but a different preconditioner (different iteration counts).
	template <SolverPreconditioner precond>
	decltype(auto) getPreconditioner() const noexcept {
		if constexpr (precond == SolverPreconditioner::NONE) {
			return IDPreconditioner();
		} else if constexpr (precond == SolverPreconditioner::SYMMETRIC_GAUS_SEIDEL) {
			return SGSPreconditioner(*this);
		} else if constexpr (precond == SolverPreconditioner::ILU0) {
			return ILU0Preconditioner(*this);
		} else if constexpr (precond == SolverPreconditioner::BLOCK_ILU0) {
			return BlockILU0Preconditioner(*this);
		} else if constexpr (precond == SolverPreconditioner::BLOCK_SGS) {
			return BlockSGSPreconditioner(*this);
		} else {
			return JacobiPreconditioner(*this);
		}
	}

	// device mirror of the three arrays, created on first use; nullptr when there is no GPU
	const smm_hip_csr* device() const noexcept {
		if (!dev && start) {
			if (detail::note(detail::Abi<T>::create(denseRowCount, denseColCount, start.get(), positions.get(), values.get(), &dev)) != SMM_HIP_OK) dev = nullptr;
		}
		return dev;
	}
	// Tuning only (results unchanged): SpMV kernel family and lanes per row for this matrix, e.g. SMM_SPMV_PATTERN for stencil /
	// banded matrices.  Returns the ABI status: non-zero when the matrix does not qualify (the previous choice stays).
	int setSpmvKernel(const int family, const int lanesPerRow = 0) const noexcept {
		(void)device();
		return dev ? smm_hip_csr_set_kernel(dev, family, lanesPerRow) : SMM_HIP_ERR_NO_DEVICE;
	}
	// call after editing values/positions in place through the raw accessors below
	void invalidateDevice() noexcept {
		smm_hip_csr_destroy(dev);
		dev = nullptr;
	}
	const T* rawValues() const noexcept { return values.get(); }
	const int* rawPositions() const noexcept { return positions.get(); }
	const int* rawStart() const noexcept { return start.get(); }

private:
	void spmv(int op, const T* lhs, const T* mult, T* out) const noexcept {
		const smm_hip_csr* d = device();  // a failed mirror has already noted its status
		if (!d || detail::note(detail::Abi<T>::spmv(d, op, lhs, mult, out)) != SMM_HIP_OK) detail::fillNaN(out, denseRowCount);
	}
	void computeFirstActive() noexcept {  // ref:1619-1628
		firstActiveStart = denseRowCount;
		for (int i = 0; i < denseRowCount; ++i) {
			if (start[i + 1] != 0) {
				firstActiveStart = i;
				break;
			}
		}
	}
	void release() noexcept {
		smm_hip_csr_destroy(dev);
		dev = nullptr;
	}
	// the reference's layout (ref:1243-1259)
	std::unique_ptr<T[]> values;
	std::unique_ptr<int[]> positions;
	std::unique_ptr<int[]> start;
	int denseRowCount = 0;
	int denseColCount = 0;
	int firstActiveStart = 0;
	mutable smm_hip_csr* dev = nullptr;
};

// ---- solvers ---------------------------------------------------------------------------------------------------------------
namespace detail {
inline SolverStatus toStatus(int abi, int solver) {
	// no GPU / HIP failure: DIVERGED, with lastHipStatus() != 0 (a numerical DIVERGED leaves it 0) and the text in smm_hip_last_error()
	if (note(abi) != SMM_HIP_OK) return SolverStatus::DIVERGED;
	return static_cast<SolverStatus>(solver);
}
template <typename T>
struct Functor;
template <>
struct Functor<float> {
	static int run(const smm_hip_csr* a, float* b, float* x, int it, float eps, smm_hip_apply_fn_f32 fn, void* user, int* st) {
		return smm_hip_bicgstab_functor_f32(a, b, x, it, eps, fn, user, st, nullptr, nullptr);
	}
};
template <>
struct Functor<double> {
	static int run(const smm_hip_csr* a, double* b, double* x, int it, double eps, smm_hip_apply_fn_f64 fn, void* user, int* st) {
		return smm_hip_bicgstab_functor_f64(a, b, x, it, eps, fn, user, st, nullptr, nullptr);
	}
};
}  // namespace detail

// ref:2316-2398
template <typename T>
inline SolverStatus ConjugateGradient(const CSRMatrix<T>& a, const T* const b, const T* const x0, T* const x, int maxIterations, T eps) {
	int st = 0;
	const smm_hip_csr* d = a.device();
	const int rc = d ? detail::Abi<T>::cg(d, b, x0, x, maxIterations, eps, nullptr, &st) : SMM_HIP_ERR_NO_DEVICE;
	return detail::toStatus(rc, st);
}

// ref:2414-2505
template <typename T>
inline SolverStatus ConjugateGradient(const CSRMatrix<T>& a, const T* const b, const T* const x0, T* const x, int maxIterations, T eps,
                                      const typename CSRMatrix<T>::IC0Preconditioner& M) {
	int st = 0;
	const smm_hip_csr* d = a.device();
	const smm_hip_precond* h = M.handle();
	const int rc = d && h ? detail::Abi<T>::cg(d, b, x0, x, maxIterations, eps, h, &st) : SMM_HIP_ERR_NO_DEVICE;
	return detail::toStatus(rc, st);
}

// ref:2191-2283.  CSRMatrix<T>'s own preconditioner classes live on the GPU and run inside the device-resident loop.  ANY other type
// with the reference's `int apply(const T* rhs, T* x) const` (ref:2199, 2218, 2235, 2251) is accepted as well: the loop then runs
// its SpMVs, reductions and updates on the device and calls the functor on the host around two PCIe copies per apply (the slow
// path: smm_hip_bicgstab_functor_*).
template <typename Preconditioner, typename T>
inline SolverStatus BiCGStab(const CSRMatrix<T>& a, T* b, T* x, int maxIterations, T eps, const Preconditioner& preconditioner) {
	int st = 0;
	const smm_hip_csr* d = a.device();
	if (!d) return SolverStatus::DIVERGED;
	if constexpr (std::is_same<Preconditioner, typename CSRMatrix<T>::IDPreconditioner>::value ||
	              std::is_base_of<typename CSRMatrix<T>::PreconditionerBase, Preconditioner>::value) {
		const smm_hip_precond* h = preconditioner.handle();
		constexpr bool precondition = !std::is_same<Preconditioner, typename CSRMatrix<T>::IDPreconditioner>::value;
		if (precondition && !h) return SolverStatus::DIVERGED;
		const int rc = detail::Abi<T>::bicgstab(d, b, x, maxIterations, eps, h, &st);
		return detail::toStatus(rc, st);
	} else {
		auto trampoline = +[](void* user, const T* rhs, T* out) -> int { return static_cast<const Preconditioner*>(user)->apply(rhs, out); };
		const int rc = detail::Functor<T>::run(d, b, x, maxIterations, eps, trampoline, const_cast<Preconditioner*>(&preconditioner), &st);
		return detail::toStatus(rc, st);
	}
}

// ref:2294-2303
template <typename T>
inline SolverStatus BiCGStab(const CSRMatrix<T>& a, T* b, T* x, int maxIterations, T eps) {
	return BiCGStab(a, b, x, maxIterations, eps, typename CSRMatrix<T>::IDPreconditioner());
}

// ref:2021-2102
template <typename T>
inline SolverStatus BiCGSymmetric(const CSRMatrix<T>& a, T* b, T* x, int maxIterations, T eps) {
	int st = 0;
	const smm_hip_csr* d = a.device();
	const int rc = d ? detail::Abi<T>::bicgsym(d, b, x, maxIterations, eps, &st) : SMM_HIP_ERR_NO_DEVICE;
	return detail::toStatus(rc, st);
}

// ---- file loaders (ref:2507-2669) ---------------------------------------------------------------------------------------------
// Same enumerators in the same order as ref:2507-2522; additions follow them.
enum class MatrixLoadStatus {
	SUCCESS = 0,
	FAILED_TO_OPEN_FILE,
	FAILED_TO_OPEN_FILE_UNKNOWN_FORMAT,
	FAILED_TO_PARSE_FILE,
	PARSE_ERROR_MMX_FILE_MISSING_BANNER,
	PARSE_ERROR_MMX_FILE_UNSUPPORTED_TYPE,
	PARSE_ERROR_MMX_FILE_UNSUPPORTED_FORMAT,
	PARSE_ERROR_MMX_FILE_UNSUPPORTED_EL_TYPE,
	PARSE_ERROR_MMX_FILE_UNSUPPORTED_STRUCTURE,
	// additions
	PARSE_ERROR_INDEX_OUT_OF_RANGE,  // an entry outside rows x cols (undefined behaviour in the reference)
	MATRIX_TOO_LARGE                 // more than 2^31 - 1 stored entries after mirroring: start[] is int (ref:1251-1257)
};

namespace detail {
struct MmEntry {
	int row, col;
	double value;
};
struct MmHeader {
	int rows = 0, cols = 0;
	long long declared = 0;
	bool pattern = false, symmetric = false;
};
inline void lower(std::string& s) {
	for (char& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
}
// Banner, comments and size line exactly as ref:2537-2581 reads them: `%%MatrixMarket` case-sensitive, the four qualifiers
// case-insensitive; `coordinate` only; `real` / `integer` (ref) and `pattern` (addition: every entry is 1); `symmetric` (ref) and
// `general` (addition).
inline MatrixLoadStatus readMmHeader(std::istream& file, MmHeader& h) {
	std::string banner, matrix, format, type, structure;
	file >> banner;
	if (banner != "%%MatrixMarket") return MatrixLoadStatus::PARSE_ERROR_MMX_FILE_MISSING_BANNER;
	file >> matrix;
	lower(matrix);
	if (matrix != "matrix") return MatrixLoadStatus::PARSE_ERROR_MMX_FILE_UNSUPPORTED_TYPE;
	file >> format;
	lower(format);
	if (format != "coordinate") return MatrixLoadStatus::PARSE_ERROR_MMX_FILE_UNSUPPORTED_FORMAT;
	file >> type;
	lower(type);
	h.pattern = type == "pattern";
	if (!h.pattern && type != "real" && type != "integer") return MatrixLoadStatus::PARSE_ERROR_MMX_FILE_UNSUPPORTED_EL_TYPE;
	file >> structure;
	lower(structure);
	h.symmetric = structure == "symmetric";
	if (!h.symmetric && structure != "general") return MatrixLoadStatus::PARSE_ERROR_MMX_FILE_UNSUPPORTED_STRUCTURE;
	while (file.peek() == '%' || std::isspace(file.peek())) file.ignore(std::numeric_limits<std::streamsize>::max(), '\n');
	file >> h.rows >> h.cols >> h.declared;
	if (file.fail() || h.rows < 0 || h.cols < 0) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
	return MatrixLoadStatus::SUCCESS;
}
// Entries until end of file, like the reference's loop (ref:2584-2608: the declared count only sizes the container); every
// entry goes to sink(row, col, value) with 0-based indices, off-diagonal entries of a symmetric file also mirrored (ref:2598-2601).
template <typename Sink>
inline MatrixLoadStatus readMmEntries(std::istream& file, const MmHeader& h, Sink&& sink) {
	while (std::isspace(file.peek())) file.get();
	while (!file.eof() && file.peek() != std::char_traits<char>::eof()) {
		int row = 0, col = 0;
		double value = 1.0;
		file >> row >> col;
		if (!h.pattern) file >> value;
		if (file.fail()) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
		if (row < 1 || col < 1 || row > h.rows || col > h.cols) return MatrixLoadStatus::PARSE_ERROR_INDEX_OUT_OF_RANGE;
		sink(row - 1, col - 1, value);
		if (h.symmetric && row != col) sink(col - 1, row - 1, value);
		while (std::isspace(file.peek())) file.get();
	}
	return MatrixLoadStatus::SUCCESS;
}
}  // namespace detail

// ref:2531-2609, same signature: Matrix Market coordinate file into a TripletMatrix (duplicates add up, ref:612-617)
template <typename T>
inline MatrixLoadStatus loadMatrixMarketMatrix(const char* filepath, TripletMatrix<T>& out) {
	std::ifstream file(filepath);
	if (!file.is_open()) return MatrixLoadStatus::FAILED_TO_OPEN_FILE;
	detail::MmHeader h;
	const MatrixLoadStatus st = detail::readMmHeader(file, h);
	if (st != MatrixLoadStatus::SUCCESS) return st;
	out.init(h.rows, h.cols, static_cast<int>(std::min<long long>(h.declared, std::numeric_limits<int>::max())));
	return detail::readMmEntries(file, h, [&](int r, int c, double v) { out.addEntry(r, c, static_cast<T>(v)); });
}

// Addition: the same file DIRECT TO CSR, without the std::map of the triplet form (ref:606-618 costs ~50 bytes and a tree insertion
// per entry: 5e8 entries do not fit).  Entries are collected, stably sorted by (row, column) and duplicates added in file order --
// the same sums in the same order as TripletMatrix::addEntry forms them -- and the three CSR arrays are written in one pass.
template <typename T>
inline MatrixLoadStatus loadMatrixMarketMatrix(const char* filepath, CSRMatrix<T>& out) {
	std::ifstream file(filepath);
	if (!file.is_open()) return MatrixLoadStatus::FAILED_TO_OPEN_FILE;
	detail::MmHeader h;
	MatrixLoadStatus st = detail::readMmHeader(file, h);
	if (st != MatrixLoadStatus::SUCCESS) return st;
	std::vector<detail::MmEntry> entries;
	if (h.declared > 0) entries.reserve(static_cast<size_t>(h.symmetric ? 2 * h.declared : h.declared));
	st = detail::readMmEntries(file, h, [&](int r, int c, double v) { entries.push_back({r, c, v}); });
	if (st != MatrixLoadStatus::SUCCESS) return st;
	std::stable_sort(entries.begin(), entries.end(), [](const detail::MmEntry& a, const detail::MmEntry& b) { return a.row != b.row ? a.row < b.row : a.col < b.col; });
	std::vector<int> start(static_cast<size_t>(h.rows) + 1, 0), positions;
	std::vector<T> values;
	positions.reserve(entries.size());
	values.reserve(entries.size());
	for (size_t i = 0; i < entries.size();) {
		size_t j = i;
		T sum = T(0);  // TripletMatrix::addEntry: data[key] += value starting from T() (ref:612-617)
		for (; j < entries.size() && entries[j].row == entries[i].row && entries[j].col == entries[i].col; ++j) sum += static_cast<T>(entries[j].value);
		if (positions.size() == static_cast<size_t>(std::numeric_limits<int>::max())) return MatrixLoadStatus::MATRIX_TOO_LARGE;
		positions.push_back(entries[i].col);
		values.push_back(sum);
		start[static_cast<size_t>(entries[i].row) + 1]++;
		i = j;
	}
	for (int r = 0; r < h.rows; ++r) start[static_cast<size_t>(r) + 1] += start[static_cast<size_t>(r)];
	return out.init(h.rows, h.cols, start.data(), positions.data(), values.data()) == 0 ? MatrixLoadStatus::SUCCESS : MatrixLoadStatus::FAILED_TO_PARSE_FILE;
}

// The dense text format the reference's saveDenseText writes and ref:2611-2643 reads: `rows cols { {a, b, ...}, {...}, ... }`.
// Written from that format description with a parser of its own: the file is read whole and scanned once as a token stream --
// numbers by strtod, `{` `}` tracked as a nesting depth, commas and white space skipped -- and the shape is CHECKED (exactly `rows`
// inner groups of exactly `cols` numbers, a closed outer group), which the format allows and a stream of >> / ignore calls cannot:
// STRICTER than the reference, which reads `cols` numbers per row and skips whatever else the line holds (INTEGRATION.md lists the
// difference).  Text after the closing brace is ignored, as there.  Zeros are not stored.
template <typename T>
inline MatrixLoadStatus loadSMMDTMatrix(const char* filepath, TripletMatrix<T>& out) {
	std::FILE* f = std::fopen(filepath, "rb");
	if (!f) return MatrixLoadStatus::FAILED_TO_OPEN_FILE;
	std::string text;
	char buf[1 << 16];
	for (size_t got; (got = std::fread(buf, 1, sizeof(buf), f)) > 0;) text.append(buf, got);
	std::fclose(f);
	const char* p = text.c_str();
	const char* const end = p + text.size();
	long dims[2] = {0, 0};
	for (long& d : dims) {  // the two leading integers
		char* after = nullptr;
		d = std::strtol(p, &after, 10);
		if (after == p || d < 0 || d > std::numeric_limits<int>::max()) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
		p = after;
	}
	const int rows = static_cast<int>(dims[0]), cols = static_cast<int>(dims[1]);
	out.init(rows, cols, 0);
	int depth = 0, row = -1, col = 0;  // depth 1: between rows; depth 2: inside row `row`, `col` numbers read so far
	bool closed = false;
	while (p < end && !closed) {
		const char c = *p;
		if (c == '{') {
			if (++depth > 2) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
			if (depth == 2) {
				if (++row >= rows) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
				col = 0;
			}
			++p;
		} else if (c == '}') {
			if (depth == 2 && col != cols) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
			if (--depth < 0) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
			closed = depth == 0;
			++p;
		} else if (c == ',' || std::isspace(static_cast<unsigned char>(c))) {
			++p;
		} else {
			// a number as `file >> val` with val of type T reads it (ref:2629-2634): parsed straight to T -- strtof for float, so that the
			// rounding is the one-step rounding of the reference and not double -> float --, plain decimal notation only (operator>> takes
			// neither "inf" / "nan" nor hexadecimal floats), and the zero test is made on the T value (a number that underflows to 0 in T
			// is not stored, as in the reference)
			const char* q = p + ((*p == '+' || *p == '-') ? 1 : 0);
			if (!(std::isdigit(static_cast<unsigned char>(*q)) || *q == '.') || (q[0] == '0' && (q[1] == 'x' || q[1] == 'X'))) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
			char* after = nullptr;
			T v;
			if (sizeof(T) == sizeof(float)) v = static_cast<T>(std::strtof(p, &after));
			else v = static_cast<T>(std::strtod(p, &after));
			if (after == p || depth != 2 || col >= cols) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
			if (v != T(0)) out.addEntry(row, col, v);
			++col;
			p = after;
		}
	}
	if (!closed || row + 1 != rows) return MatrixLoadStatus::FAILED_TO_PARSE_FILE;
	return MatrixLoadStatus::SUCCESS;
}

// ref:2645-2656: by file extension
template <typename T>
inline MatrixLoadStatus loadMatrix(const char* filepath, TripletMatrix<T>& out) {
	const char* dot = std::strrchr(filepath, '.');
	if (dot && std::strcmp(dot + 1, "mtx") == 0) return loadMatrixMarketMatrix(filepath, out);
	if (dot && std::strcmp(dot + 1, "smmdt") == 0) return loadSMMDTMatrix(filepath, out);
	return MatrixLoadStatus::FAILED_TO_OPEN_FILE_UNKNOWN_FORMAT;
}

// ref:2658-2669.  `.mtx` goes direct to CSR (above); the dense text format is small by nature and keeps the triplet route.
template <typename T>
inline MatrixLoadStatus loadMatrix(const char* filepath, CSRMatrix<T>& out) {
	const char* dot = std::strrchr(filepath, '.');
	if (dot && std::strcmp(dot + 1, "mtx") == 0) return loadMatrixMarketMatrix(filepath, out);
	TripletMatrix<T> triplet;
	const MatrixLoadStatus status = loadMatrix(filepath, triplet);
	if (status != MatrixLoadStatus::SUCCESS) return status;
	out.init(triplet);
	return MatrixLoadStatus::SUCCESS;
}

}  // namespace SMM
