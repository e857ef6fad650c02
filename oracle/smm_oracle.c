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

/* The team size is handed to every parallel region explicitly (num_threads(SMM_ORACLE_NT) in smm_oracle_impl.inc): two OpenMP run-times
 * can live in one process (the gcc build's libgomp, the clang-built timing flavour's libomp, torch's own), and a call to
 * omp_set_num_threads() binds to whichever the dynamic loader finds first -- not necessarily the one that runs this library's regions. */
static int smm_oracle_threads_ = 0; /* 0: the run-time's default */

int smm_oracle_omp_max_threads(void) {
#ifdef _OPENMP
	return smm_oracle_threads_ > 0 ? smm_oracle_threads_ : omp_get_max_threads();
#else
	return 1;
#endif
}

void smm_oracle_omp_set_threads(int n) {
	if (n > 0) smm_oracle_threads_ = n;
}

#define SMM_ORACLE_NT smm_oracle_omp_max_threads()

/* form of the OpenMP port's SpMV row loop: 1 = one row at a time (the reference's loop under a parallel for), 2 / 4 = that many rows
 * walked in lock step (independent multiply-add chains; every row's own sum keeps its order: same bits).  tools/cpu_port_ab.py times
 * them against the real reference on the GPU box's host; the default is what won there. */
static int smm_oracle_spmv_form_ = 2; /* profiles/r04/cpu_port_ab.txt: 10 M rows, one core of an EPYC 9575F: 2 rows 2.80 it/s, 1 row 2.60, 4 rows 2.47; the real reference 2.67 */
void smm_oracle_omp_set_spmv_form(int rows_in_lock_step) {
	if (rows_in_lock_step == 1 || rows_in_lock_step == 2 || rows_in_lock_step == 4) smm_oracle_spmv_form_ = rows_in_lock_step;
}

int smm_oracle_uses_std_fma(void) {
#ifdef SMM_WITH_STD_FMA
	return 1;
#else
	return 0;
#endif
}

/* Level cut of the block preconditioners (an addition; no counterpart in the reference).  Inside block b (rows bounds[b] ..
 * bounds[b+1]) the forward sweep of a triangular solve gives row i the level  lo(i) = 0 when it keeps no in-block entry left of
 * the diagonal, else 1 + the largest lo(j) over the entries (i, j), j < i, it keeps; the backward sweep defines up(i) the same
 * way over the entries right of the diagonal, rows descending.  With a cap C > 0 an entry (i, j), j < i, is KEPT only while
 * lo(j) < C - 1, and an entry (i, j), j > i, only while up(j) < C - 1: no row of either sweep then sits deeper than level C - 1,
 * whatever the matrix.  Rows are visited in the sweep's own order (ascending for lo, descending for up), so the rule is a plain
 * recurrence.  keep[k] = 1 for every stored entry that takes part in M (the diagonal, and the in-block entries the rule keeps),
 * 0 for entries that couple two blocks or that the cap drops; M is then the block preconditioner of the matrix made of the kept
 * entries.  cap <= 0: no cap (keep = the block-diagonal part).  Returns the deepest level + 1 over both sweeps and all blocks. */
int smm_oracle_block_level_cut(int rows, const int* start, const int* positions, int nblocks, const int* bounds, int cap,
                               unsigned char* keep) {
	int deepest = 0;
	int* lvl = (int*)malloc(sizeof(int) * (size_t)(rows > 0 ? rows : 1));
	if (!lvl) return -1;
	for (int k = 0; k < start[rows]; ++k) keep[k] = 0;
	for (int b = 0; b < nblocks; ++b) {
		const int r0 = bounds[b], r1 = bounds[b + 1];
		for (int i = r0; i < r1; ++i) { /* forward sweep */
			int lv = 0;
			for (int k = start[i]; k < start[i + 1]; ++k) {
				const int j = positions[k];
				if (j < r0 || j >= i) continue;
				if (cap > 0 && lvl[j] >= cap - 1) continue;
				keep[k] = 1;
				if (lvl[j] + 1 > lv) lv = lvl[j] + 1;
			}
			lvl[i] = lv;
			if (lv + 1 > deepest) deepest = lv + 1;
		}
		for (int i = r1 - 1; i >= r0; --i) { /* backward sweep */
			int lv = 0;
			for (int k = start[i]; k < start[i + 1]; ++k) {
				const int j = positions[k];
				if (j == i) keep[k] = 1;
				if (j <= i || j >= r1) continue;
				if (cap > 0 && lvl[j] >= cap - 1) continue;
				keep[k] = 1;
				if (lvl[j] + 1 > lv) lv = lvl[j] + 1;
			}
			lvl[i] = lv;
			if (lv + 1 > deepest) deepest = lv + 1;
		}
	}
	free(lvl);
	return deepest;
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
