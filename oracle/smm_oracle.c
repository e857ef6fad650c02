/*
 * smm_oracle.c -- instantiates the CPU oracle for float and double.
 * TEST INFRASTRUCTURE ONLY: see smm_oracle.h for what may load this and for the parity status.
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fopenmp -shared -fPIC smm_oracle.c -lm   (oracle/Makefile)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "smm_oracle.h"

/* firstActiveStart: index of the first row with an entry, rows when the matrix is empty
 * (CSRMatrix<T>::fillArrays, include/sparse_matrix_math.h:1619-1628) */
int smm_oracle_first_active_start(int rows, const int* start) {
	for (int i = 0; i < rows; ++i) {
		if (start[i + 1] != 0) {
			return i;
		}
	}
	return rows;
}

int smm_oracle_omp_max_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

void smm_oracle_omp_set_threads(int n) {
#ifdef _OPENMP
	if (n > 0) omp_set_num_threads(n);
#else
	(void)n;
#endif
}

int smm_oracle_uses_std_fma(void) {
#ifdef SMM_WITH_STD_FMA
	return 1;
#else
	return 0;
#endif
}

#define T float
#define FN(name) name##_f32
#define SQRT sqrtf
#define FABS fabsf
#define FMA fmaf
#include "smm_oracle_impl.inc"
#undef T
#undef FN
#undef SQRT
#undef FABS
#undef FMA

#define T double
#define FN(name) name##_f64
#define SQRT sqrt
#define FABS fabs
#define FMA fma
#include "smm_oracle_impl.inc"
#undef T
#undef FN
#undef SQRT
#undef FABS
#undef FMA
