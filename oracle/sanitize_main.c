/*
 * sanitize_main.c -- drives every function of the CPU oracle (smm_oracle.c) on small matrices so that the whole restatement runs
 * under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle sanitize`; tests/test_oracle.py::test_oracle_under_sanitizers).
 * TEST INFRASTRUCTURE like the rest of oracle/: it checks the checker, nothing in the product links it.
 * The matrices: 2-D 5-point Poisson 12 x 12 (SPD) and a non-symmetric convection-diffusion variant of it, plus a matrix with empty
 * rows for the SpMV edge cases (ref:1479-1483).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smm_oracle.h"

static int failures = 0;
#define EXPECT(cond)                                                      \
	do {                                                                  \
		if (!(cond)) {                                                    \
			++failures;                                                   \
			printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
		}                                                                 \
	} while (0)

typedef struct {
	int rows, nnz;
	int *start, *positions;
	double* values;
	float* valuesf;
} Csr;

/* 5-point stencil on an N x N grid; lower neighbours weigh `lo`, upper `hi` (lo == hi == -1: the Laplacian) */
static Csr stencil(int N, double lo, double hi) {
	Csr m;
	m.rows = N * N;
	m.start = (int*)malloc(sizeof(int) * (size_t)(m.rows + 1));
	m.positions = (int*)malloc(sizeof(int) * (size_t)(5 * m.rows));
	m.values = (double*)malloc(sizeof(double) * (size_t)(5 * m.rows));
	int k = 0;
	for (int j = 0; j < N; ++j) {
		for (int i = 0; i < N; ++i) {
			const int r = j * N + i;
			m.start[r] = k;
			if (j > 0) { m.positions[k] = r - N; m.values[k++] = lo; }
			if (i > 0) { m.positions[k] = r - 1; m.values[k++] = lo; }
			m.positions[k] = r; m.values[k++] = 4.0;
			if (i + 1 < N) { m.positions[k] = r + 1; m.values[k++] = hi; }
			if (j + 1 < N) { m.positions[k] = r + N; m.values[k++] = hi; }
		}
	}
	m.start[m.rows] = k;
	m.nnz = k;
	m.valuesf = (float*)malloc(sizeof(float) * (size_t)k);
	for (int q = 0; q < k; ++q) m.valuesf[q] = (float)m.values[q];
	return m;
}

static void release(Csr* m) {
	free(m->start);
	free(m->positions);
	free(m->values);
	free(m->valuesf);
}

static void rowSums(const Csr* m, double* b, float* bf) {
	for (int r = 0; r < m->rows; ++r) {
		double s = 0;
		for (int k = m->start[r]; k < m->start[r + 1]; ++k) s += m->values[k];
		b[r] = s;
		bf[r] = (float)s;
	}
}

int main(void) {
	const int N = 12;
	Csr spd = stencil(N, -1.0, -1.0);
	Csr cd = stencil(N, -1.3, -0.7);
	const int n = spd.rows;
	double *b = malloc(sizeof(double) * (size_t)n), *x = malloc(sizeof(double) * (size_t)n), *y = malloc(sizeof(double) * (size_t)n);
	double* f = malloc(sizeof(double) * (size_t)spd.nnz);
	float *bf = malloc(sizeof(float) * (size_t)n), *xf = malloc(sizeof(float) * (size_t)n), *yf = malloc(sizeof(float) * (size_t)n);
	float* ff = malloc(sizeof(float) * (size_t)spd.nnz);
	int it = 0;
	double res = 0;
	float resf = 0;

	/* SpMV, all three ops, serial and OpenMP, in place and out of place */
	rowSums(&spd, b, bf);
	for (int op = 0; op < 3; ++op) {
		for (int i = 0; i < n; ++i) { x[i] = 1.0; y[i] = 2.0; xf[i] = 1.f; yf[i] = 2.f; }
		smm_oracle_spmv_f64(n, spd.start, spd.positions, spd.values, op, y, x, y);
		smm_oracle_spmv_f32(n, spd.start, spd.positions, spd.valuesf, op, yf, xf, yf);
		const double want = op == 0 ? b[5] : op == 1 ? 2.0 + b[5] : 2.0 - b[5];
		EXPECT(fabs(y[5] - want) < 1e-12 && fabsf(yf[5] - (float)want) < 1e-5f);
		for (int i = 0; i < n; ++i) { y[i] = 2.0; yf[i] = 2.f; }
		smm_oracle_omp_spmv_f64(n, spd.start, spd.positions, spd.values, op, y, x, y);
		smm_oracle_omp_spmv_f32(n, spd.start, spd.positions, spd.valuesf, op, yf, xf, yf);
		EXPECT(fabs(y[5] - want) < 1e-12);
	}
	{ /* leading, inner and trailing empty rows */
		int start[6] = {0, 0, 2, 2, 3, 3}, pos[3] = {0, 4, 2};
		double val[3] = {1.5, -2.0, 4.0}, xx[5] = {1, 2, 3, 4, 5}, lhs[5] = {9, 9, 9, 9, 9}, out[5];
		smm_oracle_spmv_f64(5, start, pos, val, SMM_ORACLE_OP_SUB, lhs, xx, out);
		EXPECT(out[0] == 9 && out[1] == 9 - (1.5 - 10.0) && out[2] == 9 && out[3] == 9 - 12.0 && out[4] == 9);
		EXPECT(smm_oracle_first_active_start(5, start) == 1);
	}
	/* reductions */
	for (int i = 0; i < n; ++i) { x[i] = 0.5 + i; xf[i] = (float)x[i]; }
	EXPECT(fabs(smm_oracle_dot_f64(n, x, x) - smm_oracle_nrm2sq_f64(n, x)) < 1e-9);
	EXPECT(fabs(smm_oracle_dot_tbbshape_f64(n, x, x) - smm_oracle_omp_dot_f64(n, x, x)) < 1e-6);
	EXPECT(fabsf(smm_oracle_dot_f32(n, xf, xf) - smm_oracle_nrm2sq_f32(n, xf)) < 1.f);
	(void)smm_oracle_dot_tbbshape_f32(n, xf, xf);
	(void)smm_oracle_omp_dot_f32(n, xf, xf);
	/* CG (serial, OpenMP), PCG + IC0, BiCGSymmetric on the SPD matrix: x -> 1 */
	memset(x, 0, sizeof(double) * (size_t)n);
	EXPECT(smm_oracle_cg_f64(n, spd.start, spd.positions, spd.values, b, x, x, -1, 1e-10, &it, &res) == SMM_ORACLE_SUCCESS);
	EXPECT(fabs(x[n / 2] - 1.0) < 1e-8 && it > 0);
	memset(xf, 0, sizeof(float) * (size_t)n);
	EXPECT(smm_oracle_cg_f32(n, spd.start, spd.positions, spd.valuesf, bf, xf, xf, -1, 1e-4f, &it, &resf) == SMM_ORACLE_SUCCESS);
	memset(x, 0, sizeof(double) * (size_t)n);
	EXPECT(smm_oracle_omp_cg_f64(n, spd.start, spd.positions, spd.values, b, x, x, -1, 1e-10, &it, &res) == SMM_ORACLE_SUCCESS);
	memset(xf, 0, sizeof(float) * (size_t)n);
	(void)smm_oracle_omp_cg_f32(n, spd.start, spd.positions, spd.valuesf, bf, xf, xf, 5, 0.f, &it, &resf);
	EXPECT(smm_oracle_ic0_factorize_f64(n, spd.start, spd.positions, spd.values, f) == 0);
	EXPECT(smm_oracle_ic0_apply_f64(n, spd.start, spd.positions, f, b, y) == 0);
	memset(x, 0, sizeof(double) * (size_t)n);
	EXPECT(smm_oracle_pcg_ic0_f64(n, spd.start, spd.positions, spd.values, f, b, x, x, -1, 1e-10, &it, &res) == SMM_ORACLE_SUCCESS);
	EXPECT(fabs(x[3] - 1.0) < 1e-8);
	EXPECT(smm_oracle_ic0_factorize_f32(n, spd.start, spd.positions, spd.valuesf, ff) == 0);
	EXPECT(smm_oracle_ic0_apply_f32(n, spd.start, spd.positions, ff, bf, yf) == 0);
	memset(xf, 0, sizeof(float) * (size_t)n);
	(void)smm_oracle_pcg_ic0_f32(n, spd.start, spd.positions, spd.valuesf, ff, bf, xf, xf, 4, 0.f, &it, &resf);
	memset(x, 0, sizeof(double) * (size_t)n);
	rowSums(&spd, b, bf);
	EXPECT(smm_oracle_bicgsymmetric_f64(n, spd.start, spd.positions, spd.values, b, x, -1, 1e-10, &it) == SMM_ORACLE_SUCCESS);
	memset(xf, 0, sizeof(float) * (size_t)n);
	(void)smm_oracle_bicgsymmetric_f32(n, spd.start, spd.positions, spd.valuesf, bf, xf, 5, 1e-4f, &it);
	/* BiCGStab none / Jacobi / ILU0 / SGS (serial) and the OpenMP port on the non-symmetric matrix */
	for (int kind = 0; kind <= 3; ++kind) {
		const double* pv = NULL;
		const float* pvf = NULL;
		if (kind == SMM_ORACLE_PRECOND_JACOBI) {
			EXPECT(smm_oracle_jacobi_setup_f64(n, cd.start, cd.positions, cd.values, y) == 0);
			EXPECT(smm_oracle_jacobi_setup_f32(n, cd.start, cd.positions, cd.valuesf, yf) == 0);
			pv = y;
			pvf = yf;
		} else if (kind == SMM_ORACLE_PRECOND_ILU0) {
			EXPECT(smm_oracle_ilu0_factorize_f64(n, cd.start, cd.positions, cd.values, f) == 0);
			EXPECT(smm_oracle_ilu0_factorize_f32(n, cd.start, cd.positions, cd.valuesf, ff) == 0);
			pv = f;
			pvf = ff;
		}
		rowSums(&cd, b, bf);
		memset(x, 0, sizeof(double) * (size_t)n);
		EXPECT(smm_oracle_bicgstab_f64(n, cd.start, cd.positions, cd.values, b, x, -1, 1e-10, kind, pv, &it, &res) == SMM_ORACLE_SUCCESS);
		EXPECT(fabs(x[n - 1] - 1.0) < 1e-7);
		memset(xf, 0, sizeof(float) * (size_t)n);
		(void)smm_oracle_bicgstab_f32(n, cd.start, cd.positions, cd.valuesf, bf, xf, 6, 1e-30f, kind, pvf, &it, &resf);
		EXPECT(it == 6);
	}
	rowSums(&cd, b, bf);
	memset(x, 0, sizeof(double) * (size_t)n);
	EXPECT(smm_oracle_omp_bicgstab_f64(n, cd.start, cd.positions, cd.values, b, x, -1, 1e-10, &it, &res) == SMM_ORACLE_SUCCESS);
	memset(xf, 0, sizeof(float) * (size_t)n);
	(void)smm_oracle_omp_bicgstab_f32(n, cd.start, cd.positions, cd.valuesf, bf, xf, 3, 0.f, &it, &resf);
	/* stand-alone applies */
	EXPECT(smm_oracle_sgs_apply_f64(n, cd.start, cd.positions, cd.values, b, x) == 0);
	EXPECT(smm_oracle_sgs_apply_f32(n, cd.start, cd.positions, cd.valuesf, bf, xf) == 0);
	EXPECT(smm_oracle_ilu0_apply_f64(n, cd.start, cd.positions, f, b, x) == 0);
	EXPECT(smm_oracle_ilu0_apply_f32(n, cd.start, cd.positions, ff, bf, xf) == 0);
	EXPECT(smm_oracle_jacobi_apply_f64(n, y, b, x) == 0);
	EXPECT(smm_oracle_jacobi_apply_f32(n, yf, bf, xf) == 0);
	smm_oracle_omp_set_threads(2);
	EXPECT(smm_oracle_omp_max_threads() >= 1);
	(void)smm_oracle_uses_std_fma();

	release(&spd);
	release(&cd);
	free(b); free(x); free(y); free(f); free(bf); free(xf); free(yf); free(ff);
	printf("oracle under sanitizers: %d failures\n", failures);
	return failures ? 1 : 0;
}
