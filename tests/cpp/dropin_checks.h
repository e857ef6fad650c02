// dropin_checks.h -- check machinery and the file-loader checks shared by test_dropin.cpp (GPU) and test_loader.cpp (host only,
// also built with AddressSanitizer + UBSan).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "smm_hip/sparse_matrix_math.h"

static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                                   \
	do {                                                                              \
		++g_checks;                                                                   \
		if (!(cond)) {                                                                \
			++g_failed;                                                               \
			std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);             \
		}                                                                             \
	} while (0)

template <typename T>
static bool approx(T ref, T got, double eps) {
	return std::fabs(static_cast<double>(ref) - static_cast<double>(got)) <= eps * std::max(1.0, std::fabs(static_cast<double>(ref)));
}

template <typename T>
constexpr T l2Eps() { return std::is_same<T, float>::value ? T(1e-4) : T(1e-8); }
template <typename T>
constexpr double infEps() { return std::is_same<T, float>::value ? 1e-4 : 1e-8; }

static void writeFile(const std::string& path, const std::string& text) {
	std::ofstream f(path);
	f << text;
}

template <typename T>
static void testLoader(const bool withGpu) {
	using St = SMM::MatrixLoadStatus;
	const std::string path = "/tmp/smm_hip_dropin_test.mtx";
	// the data of the reference's own loader fixture (test/assets/load_symmetric_test.mtx) and the expectations of
	// test/cpp/csr.cpp:787-830: 8 stored entries (off-diagonals mirrored, the explicit zero kept), no trailing newline
	writeFile(path, "%%MatrixMarket matrix coordinate real symmetric\n5 5 6\n1 1 3\n2 2 12\n2 5 34\n3 3 -0.3\n5 5 -4\n4 3 0");
	{
		SMM::CSRMatrix<T> csr;
		CHECK(SMM::loadMatrix(path.c_str(), csr) == St::SUCCESS);
		CHECK(csr.getDenseRowCount() == 5 && csr.getDenseColCount() == 5 && csr.getNonZeroCount() == 8);
		const T want[5][5] = {{3, 0, 0, 0, 0}, {0, 12, 0, 0, 34}, {0, 0, T(-0.3), 0, 0}, {0, 0, 0, 0, 0}, {0, 34, 0, 0, -4}};
		for (int i = 0; i < 5; ++i) {
			for (int j = 0; j < 5; ++j) CHECK(csr.getValue(i, j) == want[i][j]);
		}
		// the triplet route (ref:2531, same signature) gives the same matrix
		SMM::TripletMatrix<T> trip;
		CHECK(SMM::loadMatrixMarketMatrix(path.c_str(), trip) == St::SUCCESS);
		CHECK(trip.getNonZeroCount() == 8);
		SMM::CSRMatrix<T> viaTriplet(trip);
		CHECK(viaTriplet.getNonZeroCount() == csr.getNonZeroCount());
		for (int k = 0; k < csr.getNonZeroCount(); ++k) {
			CHECK(viaTriplet.rawPositions()[k] == csr.rawPositions()[k] && viaTriplet.rawValues()[k] == csr.rawValues()[k]);
		}
		for (int r = 0; r <= 5; ++r) CHECK(viaTriplet.rawStart()[r] == csr.rawStart()[r]);
	}
	// `general` (addition): nothing is mirrored; entries in any order; duplicates add up in file order like TripletMatrix::addEntry
	writeFile(path, "%%MatrixMarket MATRIX Coordinate Real General\n% comment\n%another\n\n3 3 6\n3 1 0.25\n1 1 4.0\n1 2 1.0\n2 2 3.0\n3 3 5\n1 2 0.5\n");
	{
		SMM::CSRMatrix<T> g;
		CHECK(SMM::loadMatrix(path.c_str(), g) == St::SUCCESS);
		CHECK(g.getNonZeroCount() == 5 && g.getValue(0, 1) == T(1.5) && g.getValue(1, 0) == T(0) && g.getValue(2, 0) == T(0.25));
		if (withGpu) {
			T b[3] = {T(5.5), 3, T(5.25)}, x[3] = {0, 0, 0};
			CHECK(SMM::BiCGStab<T>(g, b, x, -1, l2Eps<T>()) == SMM::SolverStatus::SUCCESS);
			for (int i = 0; i < 3; ++i) CHECK(approx(T(1), x[i], infEps<T>()));
		}
	}
	// `pattern` (addition): every stored entry is 1; symmetric pattern mirrors
	writeFile(path, "%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n1 1\n3 1\n2 2\n");
	{
		SMM::CSRMatrix<T> pm;
		CHECK(SMM::loadMatrix(path.c_str(), pm) == St::SUCCESS);
		CHECK(pm.getNonZeroCount() == 4 && pm.getValue(0, 2) == T(1) && pm.getValue(2, 0) == T(1) && pm.getValue(1, 1) == T(1) && pm.getValue(2, 2) == T(0));
	}
	// symmetric file with a repeated off-diagonal: both mirrored copies add up (ref:2598-2601 + 612-617)
	writeFile(path, "%%MatrixMarket matrix coordinate integer symmetric\n2 2 3\n2 1 2\n2 1 3\n1 1 7\n");
	{
		SMM::CSRMatrix<T> d;
		CHECK(SMM::loadMatrix(path.c_str(), d) == St::SUCCESS);
		CHECK(d.getNonZeroCount() == 3 && d.getValue(0, 1) == T(5) && d.getValue(1, 0) == T(5) && d.getValue(0, 0) == T(7));
	}
	// the reference's status codes, by name (ref:2507-2522)
	SMM::CSRMatrix<T> e;
	SMM::TripletMatrix<T> et;
	writeFile(path, "%MatrixMarket matrix coordinate real symmetric\n1 1 1\n1 1 1\n");
	CHECK(SMM::loadMatrix(path.c_str(), e) == St::PARSE_ERROR_MMX_FILE_MISSING_BANNER);
	CHECK(SMM::loadMatrix(path.c_str(), et) == St::PARSE_ERROR_MMX_FILE_MISSING_BANNER);
	writeFile(path, "%%MatrixMarket vector coordinate real symmetric\n1 1 1\n1 1 1\n");
	CHECK(SMM::loadMatrix(path.c_str(), e) == St::PARSE_ERROR_MMX_FILE_UNSUPPORTED_TYPE);
	writeFile(path, "%%MatrixMarket matrix array real symmetric\n1 1\n1\n");
	CHECK(SMM::loadMatrix(path.c_str(), e) == St::PARSE_ERROR_MMX_FILE_UNSUPPORTED_FORMAT);
	writeFile(path, "%%MatrixMarket matrix coordinate complex symmetric\n1 1 1\n1 1 1 0\n");
	CHECK(SMM::loadMatrix(path.c_str(), e) == St::PARSE_ERROR_MMX_FILE_UNSUPPORTED_EL_TYPE);
	writeFile(path, "%%MatrixMarket matrix coordinate real skew-symmetric\n2 2 1\n2 1 1\n");
	CHECK(SMM::loadMatrix(path.c_str(), e) == St::PARSE_ERROR_MMX_FILE_UNSUPPORTED_STRUCTURE);
	writeFile(path, "%%MatrixMarket matrix coordinate real general\n2 2 1\n2 x 1\n");
	CHECK(SMM::loadMatrix(path.c_str(), e) == St::FAILED_TO_PARSE_FILE);
	writeFile(path, "%%MatrixMarket matrix coordinate real general\n2 2\n");
	CHECK(SMM::loadMatrix(path.c_str(), e) == St::FAILED_TO_PARSE_FILE);
	writeFile(path, "%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n");
	CHECK(SMM::loadMatrix(path.c_str(), e) == St::PARSE_ERROR_INDEX_OUT_OF_RANGE);
	CHECK(SMM::loadMatrix("/nonexistent/file.mtx", e) == St::FAILED_TO_OPEN_FILE);
	CHECK(SMM::loadMatrix("/tmp/smm_hip_dropin_test.txt", e) == St::FAILED_TO_OPEN_FILE_UNKNOWN_FORMAT);
	CHECK(static_cast<int>(St::PARSE_ERROR_MMX_FILE_UNSUPPORTED_STRUCTURE) == 8);  // same numbering as the reference
	std::remove(path.c_str());
	// dense text format (ref:2611-2643) as saveDenseText writes it (ref:1930-2008)
	const std::string dpath = "/tmp/smm_hip_dropin_test.smmdt";
	writeFile(dpath, "3 4\n{\n{1.500000,0,2.000000,0},\n{0,0,0,0},\n{0,3.000000,0,-1.250000}\n}\n");
	{
		SMM::TripletMatrix<T> t;
		CHECK(SMM::loadMatrix(dpath.c_str(), t) == St::SUCCESS);
		CHECK(t.getDenseRowCount() == 3 && t.getDenseColCount() == 4 && t.getNonZeroCount() == 4);
		CHECK(t.getValue(0, 0) == T(1.5) && t.getValue(0, 2) == T(2) && t.getValue(2, 1) == T(3) && t.getValue(2, 3) == T(-1.25) && t.getValue(1, 1) == T(0));
		SMM::CSRMatrix<T> m;
		CHECK(SMM::loadMatrix(dpath.c_str(), m) == St::SUCCESS);
		CHECK(m.getNonZeroCount() == 4 && m.getValue(2, 3) == T(-1.25));
	}
	// the shape is checked: a short row, a missing row, an unclosed outer group, a number outside a row, too deep a nesting
	for (const char* bad : {"2 2\n{\n{1,2},\n{3}\n}\n", "2 2\n{\n{1,2}\n}\n", "2 2\n{\n{1,2},\n{3,4}\n", "2 2\n{ 5 {1,2},{3,4} }\n", "1 1\n{ { {1} } }\n",
	                        "2 2\n{\n{1,2,3},\n{4,5}\n}\n", "x 2\n{}\n", "2 2\n{\n{1,2},\n{3,4},\n{5,6}\n}\n"}) {
		writeFile(dpath, bad);
		SMM::TripletMatrix<T> t;
		CHECK(SMM::loadMatrix(dpath.c_str(), t) == St::FAILED_TO_PARSE_FILE);
	}
	writeFile(dpath, "0 0\n{\n}\n");
	{
		SMM::TripletMatrix<T> t;
		CHECK(SMM::loadMatrix(dpath.c_str(), t) == St::SUCCESS);
		CHECK(t.getNonZeroCount() == 0);
	}
	// numbers are read as `file >> val` with val of type T reads them (ref:2629-2634): one rounding straight to T (0.1 + 2^-28 is the
	// classic double-rounding case: to double first, then to float, lands one ulp off), no "inf" / "nan" / hexadecimal floats, a value
	// that underflows to zero in T is not stored, and text after the closing brace is ignored as in the reference
	writeFile(dpath, "1 3\n{\n{0.100000003725290298461914062500000001, 1e-60, 2}\n}\ntrailing text\n");
	{
		SMM::TripletMatrix<T> t;
		CHECK(SMM::loadMatrix(dpath.c_str(), t) == St::SUCCESS);
		const T expect = sizeof(T) == sizeof(float) ? static_cast<T>(std::strtof("0.100000003725290298461914062500000001", nullptr))
		                                            : static_cast<T>(std::strtod("0.100000003725290298461914062500000001", nullptr));
		CHECK(t.getValue(0, 0) == expect);
		CHECK(t.getNonZeroCount() == (sizeof(T) == sizeof(float) ? 2 : 3));  // 1e-60 is zero in float
	}
	for (const char* bad : {"1 1\n{\n{nan}\n}\n", "1 1\n{\n{inf}\n}\n", "1 1\n{\n{0x1p3}\n}\n", "1 1\n{\n{-inf}\n}\n"}) {
		writeFile(dpath, bad);
		SMM::TripletMatrix<T> t;
		CHECK(SMM::loadMatrix(dpath.c_str(), t) == St::FAILED_TO_PARSE_FILE);
	}
	std::remove(dpath.c_str());
}

