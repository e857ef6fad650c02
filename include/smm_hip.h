/*
 * smm_hip.h -- C ABI of libsmm_hip.so: the MI355X (gfx950) implementation of the CSR SpMV + Krylov inner
 * loop behind SMM::ConjugateGradient / SMM::BiCGStab over SMM::CSRMatrix
 * (vasil-pashov/sparse_matrix_math v0.2.0, include/sparse_matrix_math.h -- cited below as "ref").
 *
 * The reference is a header-only C++ template library with no FFI seam, so this header
 * IS the drop-in boundary: plain pointers and sizes, no C++ / torch types.  Each entry point names the
 * reference interface it replaces.  include/smm_hip/sparse_matrix_math.h layers the reference's own C++
 * signatures (SMM::CSRMatrix<T>::rMult, SMM::ConjugateGradient, SMM::BiCGStab, ...) on top of these calls;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns an int shim status: SMM_HIP_OK (0) or a negative SMM_HIP_ERR_* code.  Nothing
 *     throws.  The reference's own SolverStatus (ref:2010-2014) is returned through *solver_status.
 *   - `_f32` / `_f64` select T = float / double (the reference's template parameter).  Index arrays are
 *     int32 exactly as in the reference (ref:1243-1259).
 *   - functions without `_dev` take HOST pointers, like the reference's API: they copy in, run on the GPU,
 *     copy out and synchronise before returning.
 *   - `_dev` functions take DEVICE pointers and a hipStream_t (as void*, NULL = the null stream); they only enqueue
 *     work unless they must return a value to the host (solvers synchronise the stream before returning their
 *     status).  Temporaries are recycled in stream order: drive one matrix from one stream at a time.
 *     Host-pointer functions run on a private non-blocking stream of the library.
 *   - there is NO CPU fallback: without a HIP device every call fails with SMM_HIP_ERR_NO_DEVICE.
 *   - rounding: multiply-adds are a*x+b (two roundings) like the reference's default _smm_fma (ref:28-36);
 *     a library built with -DSMM_WITH_STD_FMA uses fma(a,x,b) instead.  smm_hip_uses_std_fma() tells which.
 */
#ifndef SMM_HIP_H
#define SMM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMM_HIP_VERSION_MAJOR 0
#define SMM_HIP_VERSION_MINOR 1

/* shim status */
#define SMM_HIP_OK 0
#define SMM_HIP_ERR_INVALID (-1)   /* bad argument (null pointer, dtype mismatch, negative size, aliasing x==out) */
#define SMM_HIP_ERR_HIP (-2)       /* HIP runtime failure; text in smm_hip_last_error() */
#define SMM_HIP_ERR_NO_DEVICE (-3) /* no HIP device / smm_hip_init not possible */
#define SMM_HIP_ERR_PRECOND (-4)   /* structural failure in a preconditioner (missing / tiny diagonal, empty row): the
                                      reference's non-zero return of apply()/init() (ref:1668-1693) */
#define SMM_HIP_ERR_NOMEM (-5)
#define SMM_HIP_ERR_COMM (-6)      /* multi-GPU communicator failure (RCCL error, missing librccl, failed host callback) */

/* SolverStatus, ref:2010-2014 */
#define SMM_SOLVER_SUCCESS 0
#define SMM_SOLVER_DIVERGED 1
#define SMM_SOLVER_MAX_ITERATIONS_REACHED 2

/* SpMV op: rMult / rMultAdd / rMultSub, ref:1501-1515 */
#define SMM_OP_ASSIGN 0 /* out = A*x          */
#define SMM_OP_ADD 1    /* out = lhs + A*x    */
#define SMM_OP_SUB 2    /* out = lhs - A*x    */

/* Preconditioner kinds.  NONE = IDPreconditioner (ref:1166-1170), SGS = SGSPreconditioner (ref:1173-1186,
 * SolverPreconditioner::SYMMETRIC_GAUS_SEIDEL ref:1002-1006).  JACOBI is absent from the reference and ILU0 is
 * declared but unusable there (ref:1189-1212, 1715-1790); both are additions (BASELINE.json north_star).
 * IC0 = IC0Preconditioner (ref:1216-1235), used by the preconditioned ConjugateGradient overload. */
#define SMM_PRECOND_NONE 0
#define SMM_PRECOND_JACOBI 1
#define SMM_PRECOND_ILU0 2
#define SMM_PRECOND_SGS 3
#define SMM_PRECOND_IC0 4
/* Block-diagonal forms (additions, no counterpart in the reference).  The rows are cut into contiguous blocks; M is the ILU0 /
 * SGS preconditioner of the block-diagonal part of A: an entry that couples two blocks is dropped from M only (the solver keeps
 * multiplying with all of A).  Inside a block the arithmetic is the global preconditioner's, entry for entry -- BLOCK_SGS is
 * SGSPreconditioner::apply (ref:1658-1713) of the block-diagonal matrix, and with one block covering all rows both kinds give the
 * global preconditioner's bits.  What it buys on MI355X: a block is factorised and swept by ONE wavefront out of LDS, all blocks at
 * once, one launch per apply -- no dependency chain across the chip (csrc/smm_precond_block.hip). */
#define SMM_PRECOND_BLOCK_ILU0 5
#define SMM_PRECOND_BLOCK_SGS 6

#define SMM_DTYPE_F32 0
#define SMM_DTYPE_F64 1
#define SMM_DTYPE_I64 2 /* only as the element type handed to smm_hip_host_allreduce_fn */

/* SpMV kernel families (smm_hip_csr_set_kernel).  AUTO picks from nnz/row, and for large matrices tries PATTERN (below). */
#define SMM_SPMV_AUTO 0
#define SMM_SPMV_VECTOR 1 /* L lanes of a wavefront per row, wave shuffle reduction */
#define SMM_SPMV_STREAM 2 /* row blocks staged through LDS with 16-byte coalesced loads, row-sequential sums */
/* For matrices whose entries take their columns from a limited set of offsets relative to the row (stencil, banded and band-numbered
 * mesh matrices).  positions[] (4 bytes per entry) is replaced, in one of two encodings (smm_hip_csr_pattern_info tells which):
 *   MASKS  <= 64 distinct offsets and rows of <= 64 entries: one 64-bit mask per ROW, verified against EVERY entry on the device before
 *          the family is used; an SpMV streams only values[] (half the bytes for fp32);
 *          When in addition every entry of a diagonal holds the same value (constant-coefficient stencils: the Laplacians) -- CONST --
 *          values[] is not read either: 24 bytes per fp64 row instead of 104 -- 20 for grid-shaped matrices of >= 2^21 rows, which run the
 *          2.5-D kernel (csrc/smm_spmv_march.hip: a plane's window of x in LDS, the planes above and below in registers, 32-bit masks);
 *   CODES  <= 65536 distinct offsets: one 16-bit index into the matrix's sorted offset dictionary per ENTRY (6 instead of 8 bytes per
 *          fp32 entry), built on the device from all entries.
 * Same result bit for bit as the other families at the same lanes_per_row.  Selected explicitly (smm_hip_csr_set_kernel returns
 * SMM_HIP_ERR_INVALID when the matrix fits neither encoding) or by AUTO: the first SpMV of a matrix with >= 2^25 stored entries and
 * <= 64 entries per row on average runs the analysis once, on the caller's stream, and switches the matrix over when it passes
 * (smm_hip_csr_get_kernel then reports SMM_SPMV_PATTERN); a matrix that does not fit stays with STREAM.
 * Environment: SMM_HIP_AUTO_PATTERN=0 keeps AUTO on STREAM, SMM_HIP_AUTO_PATTERN_MIN_NNZ moves the threshold, SMM_HIP_AUTO_DICT=0 keeps
 * AUTO from using the CODES encoding, SMM_HIP_PATTERN_CONST=0 turns the constant-diagonal form off. */
#define SMM_SPMV_PATTERN 3
#define SMM_PATTERN_NONE 0  /* smm_hip_csr_pattern_info: not analysed, or the matrix fits neither encoding */
#define SMM_PATTERN_MASKS 1
#define SMM_PATTERN_CODES 2
#define SMM_PATTERN_CONST 3 /* MASKS, and every diagonal holds one value (verified bit for bit against every entry): with one lane per
                             * row an SpMV reads the row's mask, x and <= 32 numbers -- no positions[], no values[] (the Laplacians) */

typedef struct smm_hip_csr smm_hip_csr;         /* device-resident CSRMatrix<T> (ref:1243-1259) */
typedef struct smm_hip_precond smm_hip_precond; /* device-resident preconditioner */
typedef void* smm_hip_stream;                   /* hipStream_t with HIP's own meaning: NULL = the null (default) stream */

/* ---- runtime ------------------------------------------------------------------------------------------- */
/* Select the HIP device this process uses (one process per GPU) and create the library stream.  Idempotent. */
int smm_hip_init(int device);
int smm_hip_shutdown(void);
/* Text of the last failure on the calling thread ("" if none). */
const char* smm_hip_last_error(void);
int smm_hip_uses_std_fma(void);
/* name: device name (may be NULL), cus: compute units, hbm_bytes: total device memory */
int smm_hip_device_info(char* name, size_t name_cap, int* cus, size_t* hbm_bytes);
/* Blocks until everything enqueued on `stream` has finished. */
int smm_hip_stream_synchronize(smm_hip_stream stream);

/* TEST HOOK: the next device allocation of the library of at least min_bytes bytes fails ONCE with SMM_HIP_ERR_NOMEM, as if the device
 * were full (0 disarms it).  What the tests use to check that an optional step which cannot get its memory -- the automatic PATTERN
 * analysis -- leaves the caller's SpMV / solve untouched. */
int smm_hip_debug_fail_next_alloc(size_t min_bytes);
/* ---- live kernel timing (bench.py roofline) ------------------------------------------------------------------
 * When enabled every SpMV launch -- standalone or inside a solver loop -- is bracketed by a pair of HIP events on the
 * stream it is launched on.  smm_hip_profile_read waits for the recorded events, returns the summed SpMV kernel time
 * in milliseconds and the number of launches since the last reset, and optionally resets the tally. */
int smm_hip_profile_enable(int on);
int smm_hip_profile_read(double* spmv_ms, long long* spmv_launches, int reset);
/* The row-partitioned SpMV (smm_hip_dist_*): while profiling is on, every halo exchange leaves a pair of events -- the end of the local
 * block A_loc on the caller's stream, the end of the exchange on the communicator's stream.  exposed_ms = the sum over the pairs of
 * max(0, exchange end - A_loc end): the part of the exchanges the local block did NOT cover; pairs = the number of exchanges. */
int smm_hip_profile_read_waits(double* exposed_ms, long long* pairs, int reset);

/* ---- CSRMatrix<T> (ref:1243-1259; replaces CSRMatrix::init(TripletMatrix) ref:1326-1349 as the way in) ---- */
/* Copies the three host arrays of a CSRMatrix (values[nnz], positions[nnz] ascending per row, start[rows+1])
 * to the device.  The host arrays stay owned by the caller. */
int smm_hip_csr_create_f32(int rows, int cols, const int* start, const int* positions, const float* values, smm_hip_csr** out);
int smm_hip_csr_create_f64(int rows, int cols, const int* start, const int* positions, const double* values, smm_hip_csr** out);
/* Wraps arrays that already live in device memory (no copy; the caller keeps them alive and unchanged). */
int smm_hip_csr_create_dev_f32(int rows, int cols, const int* d_start, const int* d_positions, const float* d_values, smm_hip_csr** out);
int smm_hip_csr_create_dev_f64(int rows, int cols, const int* d_start, const int* d_positions, const double* d_values, smm_hip_csr** out);
int smm_hip_csr_destroy(smm_hip_csr* m);
/* getDenseRowCount / getDenseColCount / getNonZeroCount (ref:1351-1364) + dtype + firstActiveStart (ref:1619-1628) */
int smm_hip_csr_info(const smm_hip_csr* m, int* rows, int* cols, int* nnz, int* dtype, int* first_active_start);
/* Force a SpMV kernel family / lanes-per-row (0 = heuristic).  Tuning knob, not needed for correctness. */
int smm_hip_csr_set_kernel(smm_hip_csr* m, int family, int lanes_per_row);
int smm_hip_csr_get_kernel(const smm_hip_csr* m, int* family, int* lanes_per_row);
/* The STREAM family's tile table as the last SpMV built it (0 tiles before the first SpMV or for the other families): number of
 * tiles, the nonzeros / rows a tile was cut for, and whether the launches go to spmvTileKernel (1: the pieces of a row in different
 * waves -- 2 or 4 lanes per row, the benchmark matrix) or to the pipelined spmvStreamKernel (0).  Diagnostics for tests and benches. */
int smm_hip_csr_tile_info(const smm_hip_csr* m, int* tiles, int* tile_nnz_cap, int* tile_max_rows, int* tile_kernel);
/* The PATTERN family's encoding of this matrix (SMM_PATTERN_*) and the number of distinct offsets it found; NONE / 0 before the
 * analysis has run (the first SpMV of a large matrix, or smm_hip_csr_set_kernel(m, SMM_SPMV_PATTERN, ...)). */
int smm_hip_csr_pattern_info(const smm_hip_csr* m, int* encoding, int* offsets);
/* Which kernel the next SpMV of this matrix launches and what one launch of it has to move: `name` receives the kernel's template name
 * as a profiler prints it, without the template arguments ("spmvTileKernel", "spmvPatternTileKernel", "spmvPatternConstKernel" ...;
 * name_cap bytes, always terminated); *bytes_per_launch the bytes of the layout THAT kernel reads and writes once -- for the CSR
 * kernels SURVEY section 8d's B_spmv = nnz (s + 4) + (rows + 1) 4 + cols s + rows s, for the PATTERN encodings what they store instead of
 * positions[] (8 bytes per row of mask; 2 bytes per entry of code; no values[] for CONST) + start[] where the kernel reads it + x +
 * out.  Benchmarks price a kernel's launch time with THIS number, never with another layout's.  Diagnostics, like tile_info. */
int smm_hip_csr_kernel_desc(const smm_hip_csr* m, char* name, int name_cap, long long* bytes_per_launch);
/* From how many rows grid-shaped matrices are served by the 2.5-D kernels (csrc/smm_spmv_march.hip): constant diagonals (default
 * 2^21) and values read (default 6 x 2^20 fp64 / 2^24 fp32) -- below, the gather / wave kernels are as fast or faster
 * (profiles/r04/march_threshold.txt).
 * -1 restores a default.  Applies to matrices analysed afterwards.  Tuning knob; the tests use it to run the kernels on small grids. */
int smm_hip_set_march_min_rows(long long const_diagonals_rows, long long values_read_rows);
/* Test / measurement knob: from how many BYTES PER VECTOR the unpreconditioned ConjugateGradient defers its x update (csrc/smm_solvers.hip,
 * cgLazyXP: the last eight directions are kept and x is brought up to date every eighth iteration -- the reference's roundings in the
 * reference's order, bit for bit, 0.875 vector pass less per iteration, eight more vectors of device memory).  Default 64 MB (where five
 * vectors no longer fit the Infinity Cache); a negative value restores the default; SMM_HIP_CG_LAZY_X=0 in the environment turns it off. */
int smm_hip_set_cg_lazy_x_min_bytes(long long bytes);
/* Test / measurement knob: 0 keeps ConjugateGradient from forming its next direction inside the 2.5-D SpMV kernel (MarchFuse,
 * csrc/smm_spmv_march.hip; on by default wherever the deferred x update is on and that kernel serves the matrix): same bits either way. */
int smm_hip_set_cg_fuse_p(int on);
/* allow = 0: a matrix with constant diagonals keeps reading values[] (the MASKS kernels); 1 (default): CONST where it applies.  For
 * measurements of one against the other; the results are the same bits either way. */
int smm_hip_csr_pattern_allow_const(smm_hip_csr* m, int allow);
/* Times the candidate SpMV configurations on this matrix and keeps the fastest. */
int smm_hip_csr_autotune(smm_hip_csr* m);

/* ---- SpMV: CSRMatrix<T>::rMult / rMultAdd / rMultSub (ref:1458-1515) -------------------------------------- */
/* out[i] = op(lhs[i], sum_k values[k]*x[positions[k]]); empty rows give op(lhs[i],0) (ref:1479-1483);
 * out may alias lhs, x must not alias out (ref:1503).  lhs is ignored for SMM_OP_ASSIGN. */
int smm_hip_spmv_f32(const smm_hip_csr* m, int op, const float* lhs, const float* x, float* out);
int smm_hip_spmv_f64(const smm_hip_csr* m, int op, const double* lhs, const double* x, double* out);
int smm_hip_spmv_dev_f32(const smm_hip_csr* m, int op, const float* d_lhs, const float* d_x, float* d_out, smm_hip_stream stream);
int smm_hip_spmv_dev_f64(const smm_hip_csr* m, int op, const double* d_lhs, const double* d_x, double* d_out, smm_hip_stream stream);

/* ---- reductions: Vector<T>::operator* (ref:305-328), secondNormSquared (ref:296-303) ---------------------- */
int smm_hip_dot_f32(int n, const float* a, const float* b, float* result);
int smm_hip_dot_f64(int n, const double* a, const double* b, double* result);
/* d_result: one T in device memory */
int smm_hip_dot_dev_f32(int n, const float* d_a, const float* d_b, float* d_result, smm_hip_stream stream);
int smm_hip_dot_dev_f64(int n, const double* d_a, const double* d_b, double* d_result, smm_hip_stream stream);

/* ---- AXPY-style updates of the solver loops (device pointers; scalars by value) ---------------------------
 * y = a*x + y_in form used by every update loop of ref:2245-2247, 2263-2267, 2272-2274, 2362-2394:
 *   smm_hip_axpby_dev: out[i] = _smm_fma(a, x[i], y[i])      (out may alias x or y) */
int smm_hip_axpy_dev_f32(int n, float a, const float* d_x, const float* d_y, float* d_out, smm_hip_stream stream);
int smm_hip_axpy_dev_f64(int n, double a, const double* d_x, const double* d_y, double* d_out, smm_hip_stream stream);

/* ---- solvers ------------------------------------------------------------------------------------------------
 * smm_hip_cg_*  replaces  SolverStatus ConjugateGradient(const CSRMatrix<T>& a, const T* b, const T* x0, T* x,
 *                                                      int maxIterations, T eps)            (ref:2316-2398)
 *   maxIterations == -1 means rows (not clamped otherwise); convergence test eps*eps > ||r||^2; when the
 *   initial residual already passes, x is NOT written (ref:2342-2344).  x may alias x0.
 *   With M != NULL (kind SMM_PRECOND_IC0) it replaces the IC0 overload (ref:2414-2505).
 * smm_hip_bicgstab_*  replaces  SolverStatus BiCGStab(const CSRMatrix<T>& a, T* b, T* x, int maxIterations, T eps
 *                                                   [, const Preconditioner& M])          (ref:2191-2303)
 *   x is in/out; maxIterations is clamped to rows, -1 means rows; the loop body always runs once; status is
 *   SUCCESS unless iterations > maxIterations (ref:2277-2282); M == NULL is the IDPreconditioner overload.
 * Additive outputs (may be NULL; the reference API has no equivalent): iterations = loop passes executed,
 * resnorm = last ||r||^2 (cg) or ||r|| (bicgstab) the loop computed.
 */
int smm_hip_cg_f32(const smm_hip_csr* a, const float* b, const float* x0, float* x, int maxIterations, float eps,
                   const smm_hip_precond* M, int* solver_status, int* iterations, float* resnorm2);
int smm_hip_cg_f64(const smm_hip_csr* a, const double* b, const double* x0, double* x, int maxIterations, double eps,
                   const smm_hip_precond* M, int* solver_status, int* iterations, double* resnorm2);
int smm_hip_cg_dev_f32(const smm_hip_csr* a, const float* d_b, const float* d_x0, float* d_x, int maxIterations, float eps,
                       const smm_hip_precond* M, smm_hip_stream stream, int* solver_status, int* iterations, float* resnorm2);
int smm_hip_cg_dev_f64(const smm_hip_csr* a, const double* d_b, const double* d_x0, double* d_x, int maxIterations, double eps,
                       const smm_hip_precond* M, smm_hip_stream stream, int* solver_status, int* iterations, double* resnorm2);

/* Register-resident ConjugateGradient (csrc/smm_resident.hip).  An unpreconditioned CG whose matrix fits the register file of the chip
 * (rows <= 512 * CUs * {8, 4, 2, 1} for rows of at most {5, 9, 16, 27} entries: BASELINE config 2 does) runs as ONE launch with the
 * matrix held in registers and two grid-wide barriers per iteration, instead of three launches per iteration that re-read the matrix.
 * Same algorithm, same per-row arithmetic; the global sums add the rows in a different (fixed) partition, so alpha / beta differ from the
 * three-launch loop in the last bits.  mode: SMM_CG_RESIDENT_OFF never, _AUTO when it fits (default; falls back silently otherwise),
 * _REQUIRE fail with SMM_HIP_ERR_INVALID when it does not apply (tests, measurements); any other value only queries.  Returns the
 * previous mode.  Initial value from the environment variable SMM_HIP_CG_RESIDENT (0 / 1 / 2). */
#define SMM_CG_RESIDENT_OFF 0
#define SMM_CG_RESIDENT_AUTO 1
#define SMM_CG_RESIDENT_REQUIRE 2
int smm_hip_cg_resident(int mode);

/* Single-launch BiCGStab (csrc/smm_resident_bicg.hip).  A BiCGStab without a preconditioner, or with the library's Jacobi, whose matrix is in
 * the PATTERN family's row-mask encoding (banded / stencil matrices: what the solvers adopt from 2^20 stored entries) with at most 16
 * column offsets and whose vectors fit the register file (rows <= 512 * CUs * 12 in fp64, * 24 in fp32: BASELINE config 5 does) runs as
 * ONE launch: r, p, s, A p, A s in registers, x and r0 in LDS, five grid-wide barriers per iteration instead of seven dependent launches.
 * Same per-row arithmetic and update expressions as the loop; the global sums add the rows in another (fixed) partition, so alpha / omega /
 * beta differ from the loop's in the last bits.  mode as for smm_hip_cg_resident (OFF / AUTO / REQUIRE; any other value only queries);
 * returns the previous mode.  Initial value from the environment variable SMM_HIP_BICGSTAB_RESIDENT (0 / 1 / 2). */
int smm_hip_bicgstab_resident(int mode);

int smm_hip_bicgstab_f32(const smm_hip_csr* a, float* b, float* x, int maxIterations, float eps,
                         const smm_hip_precond* M, int* solver_status, int* iterations, float* resnorm);
int smm_hip_bicgstab_f64(const smm_hip_csr* a, double* b, double* x, int maxIterations, double eps,
                         const smm_hip_precond* M, int* solver_status, int* iterations, double* resnorm);
int smm_hip_bicgstab_dev_f32(const smm_hip_csr* a, const float* d_b, float* d_x, int maxIterations, float eps,
                             const smm_hip_precond* M, smm_hip_stream stream, int* solver_status, int* iterations, float* resnorm);
int smm_hip_bicgstab_dev_f64(const smm_hip_csr* a, const double* d_b, double* d_x, int maxIterations, double eps,
                             const smm_hip_precond* M, smm_hip_stream stream, int* solver_status, int* iterations, double* resnorm);

/* The generic form of the reference's template `BiCGStab<Preconditioner, T>` (ref:2191-2199): ANY preconditioner, given as a host
 * function with the reference's contract `int apply(const T* rhs, T* x)` on HOST vectors of length rows (non-zero = failure; rhs
 * and x never alias).  SpMV, reductions and updates run on the device as in smm_hip_bicgstab_*; every apply costs two PCIe copies of
 * one vector and two stream drains, so this is the slow path for preconditioners the library does not have.  A failing apply ends
 * the solve with SMM_HIP_ERR_PRECOND. */
typedef int (*smm_hip_apply_fn_f32)(void* user, const float* rhs, float* x);
typedef int (*smm_hip_apply_fn_f64)(void* user, const double* rhs, double* x);
int smm_hip_bicgstab_functor_f32(const smm_hip_csr* a, float* b, float* x, int maxIterations, float eps, smm_hip_apply_fn_f32 apply, void* user,
                                 int* solver_status, int* iterations, float* resnorm);
int smm_hip_bicgstab_functor_f64(const smm_hip_csr* a, double* b, double* x, int maxIterations, double eps, smm_hip_apply_fn_f64 apply, void* user,
                                 int* solver_status, int* iterations, double* resnorm);

/* BiCGSymmetric (ref:2021-2102): same kernels as CG plus the DIVERGED heuristics (ref:2056-2058, 2079-2081) */
int smm_hip_bicgsymmetric_f32(const smm_hip_csr* a, float* b, float* x, int maxIterations, float eps, int* solver_status, int* iterations);
int smm_hip_bicgsymmetric_f64(const smm_hip_csr* a, double* b, double* x, int maxIterations, double eps, int* solver_status, int* iterations);

/* ---- preconditioners: `int apply(const T* rhs, T* x) const noexcept` (ref:1173-1235) --------------------------
 * create: replaces CSRMatrix<T>::getPreconditioner<kind>() (ref:1643-1651) / IC0Preconditioner::init (ref:1798).
 * The matrix must outlive the preconditioner (the reference holds a const CSRMatrix&).  Structural failures
 * (missing or |d|<1e-5 diagonal, empty row, non-SPD IC0 pivot) return SMM_HIP_ERR_PRECOND. */
int smm_hip_precond_create(const smm_hip_csr* a, int kind, smm_hip_precond** out);
/* The block kinds with a chosen block size: blocks of at most block_rows rows (64 .. 2048; 0 = the default, 1024) and at most 8192
 * stored entries; smm_hip_precond_create uses the default.  The partition is a property of the handle:
 * smm_hip_precond_block_bounds returns nblocks + 1 numbers (bounds[0] = 0, bounds[nblocks] = rows) -- row numbers when the blocks are
 * runs of consecutive rows, positions in the row order of smm_hip_precond_block_rows when they are bricks of a grid (see
 * smm_hip_precond_create_block_ex: grid stencils get bricks by default). */
int smm_hip_precond_create_block(const smm_hip_csr* a, int kind, int block_rows, smm_hip_precond** out);
/* ... and a chosen LEVEL CUT.  Inside a block the forward sweep gives every row a level (0 when it keeps no entry left of the
 * diagonal, else 1 + the deepest level of the rows its kept entries point to), the backward sweep likewise; entries that point to
 * a row of level level_cap - 1 are dropped from M (like the entries that couple two blocks), so no sweep of any block runs deeper
 * than level_cap dependent levels.  level_cap: -1 = the default (16), 0 = no cut (M = the block-diagonal part of A exactly), else
 * 2 .. 4095.  The rule is a recurrence in the sweep's own row order; the tests' CPU checker states it sequentially
 * (block_level_cut) and the device result is compared with it bit for bit.  smm_hip_precond_create / _create_block use the default. */
int smm_hip_precond_create_block_capped(const smm_hip_csr* a, int kind, int block_rows, int level_cap, smm_hip_precond** out);
/* ... and a chosen PARTITION.  CONTIGUOUS: blocks are runs of consecutive rows (cut greedily from row 0).  BRICKS: for a matrix that is
 * a 2-D / 3-D grid stencil in natural order -- its entries use the offsets {0, +-1, +-nx[, +-nx ny]}, found and verified by the PATTERN
 * analysis -- the blocks are bricks of the grid (16 x 8 x 8 points for 1024 rows; squares in 2-D), so that M keeps the couplings of all
 * grid directions inside a block; the rows of a block keep their natural order.  Any other matrix: SMM_HIP_ERR_INVALID.  AUTO (what
 * every other create call uses): bricks where they apply, contiguous otherwise (SMM_HIP_BLOCK_BRICKS=0: always contiguous).
 * smm_hip_precond_block_rows returns the rows block by block (order[bounds[b] .. bounds[b+1]) are block b's rows; the identity for
 * contiguous blocks) and the brick's extent along the grid axes ({0, 0, 0} for contiguous blocks); either pointer may be NULL. */
#define SMM_BLOCKS_AUTO 0
#define SMM_BLOCKS_CONTIGUOUS 1
#define SMM_BLOCKS_BRICKS 2
int smm_hip_precond_create_block_ex(const smm_hip_csr* a, int kind, int block_rows, int level_cap, int partition, smm_hip_precond** out);
int smm_hip_precond_block_rows(const smm_hip_precond* M, int* order, size_t count, int* brick);
/* Bytes one apply reads per row besides rhs / x (/ w1): the fixed-size record of the lower and of the upper sweep, and the row-order
 * entries of a brick partition.  (Rows with more in-block entries than a record holds continue in overflow lists, not counted.) */
int smm_hip_precond_block_record_bytes(const smm_hip_precond* M, int* lower, int* upper, int* order);
int smm_hip_precond_block_level_cap(const smm_hip_precond* M, int* level_cap);
int smm_hip_precond_block_count(const smm_hip_precond* M, int* nblocks);
int smm_hip_precond_block_bounds(const smm_hip_precond* M, int* bounds, size_t count);
int smm_hip_precond_destroy(smm_hip_precond* M);
int smm_hip_precond_info(const smm_hip_precond* M, int* kind, int* levels_lower, int* levels_upper);
/* How the two triangular sweeps of SGS / ILU0 / IC0 run (same numbers bit for bit either way):
 * LEVELS: one launch per dependency level (runs of small levels share a launch); SYNCFREE: one launch per sweep, rows wait on
 * per-entry ready values.  AUTO = SYNCFREE.  A sweep that fails to finish (cannot happen; bounded for safety) makes the solve
 * or apply that used it return SMM_HIP_ERR_HIP. */
#define SMM_SWEEP_AUTO 0
#define SMM_SWEEP_LEVELS 1
#define SMM_SWEEP_SYNCFREE 2
#define SMM_SWEEP_SYNCFREE_XCD 3 /* SYNCFREE with every wavefront of a sweep on ONE XCD: rows meet in that XCD's L2 */
int smm_hip_precond_set_sweep(smm_hip_precond* M, int mode);
/* x = M^-1 rhs; rhs must not alias x (ref:1667) */
int smm_hip_precond_apply_f32(const smm_hip_precond* M, const float* rhs, float* x);
int smm_hip_precond_apply_f64(const smm_hip_precond* M, const double* rhs, double* x);
int smm_hip_precond_apply_dev_f32(const smm_hip_precond* M, const float* d_rhs, float* d_x, smm_hip_stream stream);
int smm_hip_precond_apply_dev_f64(const smm_hip_precond* M, const double* d_rhs, double* d_x, smm_hip_stream stream);
/* x = M^-1 (A v), A being the matrix M was created for: the operator a preconditioned loop applies twice per pass (ref:2234-2235,
 * 2250-2251).  BLOCK_ILU0 / BLOCK_SGS form A v inside the apply's launch, each row summed in the order of its stored entries (the
 * reference's rMult, ref:1484-1489) -- no SpMV launch, A v never travels through memory; every other kind runs the SpMV and then the
 * apply.  v must not alias x. */
int smm_hip_precond_apply_spmv_f32(const smm_hip_precond* M, const float* v, float* x);
int smm_hip_precond_apply_spmv_f64(const smm_hip_precond* M, const double* v, double* x);
int smm_hip_precond_apply_spmv_dev_f32(const smm_hip_precond* M, const float* d_v, float* d_x, smm_hip_stream stream);
int smm_hip_precond_apply_spmv_dev_f64(const smm_hip_precond* M, const double* d_v, double* d_x, smm_hip_stream stream);
/* The asynchronous apply cannot report a triangular sweep that failed to finish (the escape bound of the synchronisation-free sweeps:
 * it then publishes NaN and raises a sticky flag).  This call synchronises `stream`, returns SMM_HIP_ERR_HIP when a sweep applied on it
 * since the last call tripped the bound, and clears the flag.  The solver entry points call it themselves before they return. */
int smm_hip_precond_take_error(const smm_hip_precond* M, smm_hip_stream stream);
/* copies the factor values (ILU0 / IC0 / BLOCK_ILU0: nnz values on A's pattern -- for BLOCK_ILU0 the entries that couple two blocks
 * keep A's value; JACOBI: rows diagonal entries) to the host */
int smm_hip_precond_values_f32(const smm_hip_precond* M, float* out, size_t count);
int smm_hip_precond_values_f64(const smm_hip_precond* M, double* out, size_t count);

/* ---- stage-wise BiCGStab for row-partitioned multi-GPU solves (one process per GPU) -------------------------------
 * The loop of ref:2191-2283 cut at its global reductions.  Each rank owns rows [row_begin, row_end) of A and the matching
 * slices of every vector.  Between a *_LOCAL stage and the following *_APPLY stage the caller all-reduces (sum) the first
 * 1 or 2 scalars of the workspace's `sums` buffer across ranks (RCCL); before each SpMV it exchanges the x-vector halo.
 * Every stage only enqueues kernels; all scalars of the recurrence stay on the device.
 *
 *   r = b - A x  (caller, SpMV)                       INIT_LOCAL  -> all-reduce sums[0]   -> INIT_APPLY
 *   ap = A p  with dot_mode 1, w1 = r0 (caller)       ALPHA_LOCAL -> all-reduce sums[0]   -> ALPHA_APPLY  (alpha, s)
 *   as = A s  with dot_mode 2, w1 = s  (caller)       OMEGA_LOCAL -> all-reduce sums[0:2] -> OMEGA_APPLY  (omega, x, r)
 *                                                                 -> all-reduce sums[0:2] -> BETA_APPLY   (res, beta, p)
 */
#define SMM_STAGE_INIT_LOCAL 1
#define SMM_STAGE_INIT_APPLY 2
#define SMM_STAGE_ALPHA_LOCAL 3
#define SMM_STAGE_ALPHA_APPLY 4
#define SMM_STAGE_OMEGA_LOCAL 5
#define SMM_STAGE_OMEGA_APPLY 6
#define SMM_STAGE_BETA_APPLY 7
typedef struct smm_hip_bicgstab_ws smm_hip_bicgstab_ws;
/* number of partial sums per reduced quantity a fused SpMV writes (d_partials holds 2x this many T) */
int smm_hip_partials_count(void);
/* SpMV with the dot products of the fresh out[] fused into the epilogue: dot_mode 0 none; 1: out.w1 -> partials[0..P);
 * 2: out.out -> partials[0..P) and out.w1 -> partials[P..2P), P = smm_hip_partials_count() */
int smm_hip_spmv_fused_dev_f32(const smm_hip_csr* m, int op, const float* d_lhs, const float* d_x, float* d_out, int dot_mode,
                               const float* d_w1, float* d_partials, smm_hip_stream stream);
int smm_hip_spmv_fused_dev_f64(const smm_hip_csr* m, int op, const double* d_lhs, const double* d_x, double* d_out, int dot_mode,
                               const double* d_w1, double* d_partials, smm_hip_stream stream);
/* The same with the reduction FINISHED in the launch: the workgroup that ends last adds the partial sums (same fixed order, same bits as
 * a separate summing kernel) and leaves the totals at d_finish[smm_hip_finish_totals_offset() + {0, 1}] (dot_mode 1: out.w1; dot_mode 2:
 * out.out, out.w1).  d_finish: smm_hip_finish_len() elements, zeroed ONCE by the caller before the first use (it carries the arrival
 * counter between launches); one buffer per stream.  What the row-partitioned solvers all-reduce right behind their SpMV. */
int smm_hip_finish_len(void);
int smm_hip_finish_totals_offset(void);
int smm_hip_spmv_fused_finish_dev_f32(const smm_hip_csr* m, int op, const float* d_lhs, const float* d_x, float* d_out, int dot_mode,
                                      const float* d_w1, float* d_finish, smm_hip_stream stream);
int smm_hip_spmv_fused_finish_dev_f64(const smm_hip_csr* m, int op, const double* d_lhs, const double* d_x, double* d_out, int dot_mode,
                                      const double* d_w1, double* d_finish, smm_hip_stream stream);
/* workspace for n local rows: owns r, r0, ap, as, the partial-sum buffer, `sums` (4 scalars) and the recurrence state */
int smm_hip_bicgstab_ws_create_f32(int n, smm_hip_bicgstab_ws** out);
int smm_hip_bicgstab_ws_create_f64(int n, smm_hip_bicgstab_ws** out);
int smm_hip_bicgstab_ws_destroy(smm_hip_bicgstab_ws* ws);
/* p and s live in the caller's halo-extended buffers (they are SpMV inputs): bind the owned slices.  d_sums (optional,
 * >= 4 scalars) replaces the workspace's own `sums` buffer, e.g. with memory the caller's collective library can see. */
int smm_hip_bicgstab_ws_bind(smm_hip_bicgstab_ws* ws, void* d_p, void* d_s, void* d_sums);
int smm_hip_bicgstab_ws_pointers(const smm_hip_bicgstab_ws* ws, void** d_r, void** d_r0, void** d_ap, void** d_as, void** d_partials,
                                 void** d_sums);
int smm_hip_bicgstab_ws_stage_f32(smm_hip_bicgstab_ws* ws, int stage, float* d_x, float eps, smm_hip_stream stream);
int smm_hip_bicgstab_ws_stage_f64(smm_hip_bicgstab_ws* ws, int stage, double* d_x, double eps, smm_hip_stream stream);
/* synchronises `stream` and returns the loop state: done flag, iterations executed, last ||r|| */
int smm_hip_bicgstab_ws_result_f32(const smm_hip_bicgstab_ws* ws, smm_hip_stream stream, int* done, int* iterations, float* resnorm);
int smm_hip_bicgstab_ws_result_f64(const smm_hip_bicgstab_ws* ws, smm_hip_stream stream, int* done, int* iterations, double* resnorm);

/* ConjugateGradient (ref:2316-2398) in the same stage-wise form, on the same workspace type (r, ap and the bound p are used):
 *   r = b - A x0 (caller)                          CG_INIT_LOCAL  -> all-reduce sums[0] -> CG_INIT_APPLY  (early exit test)
 *   Ap = A p with dot_mode 1, w1 = p (caller)      CG_ALPHA_LOCAL -> all-reduce sums[0] -> CG_ALPHA_APPLY (alpha, x, r, local ||r||^2)
 *                                                                 -> all-reduce sums[0] -> CG_BETA_APPLY  (test, beta, p)
 * d_xcur is x0 on the first iteration and x afterwards (ref:2351, 2395); x is only written once the loop runs. */
#define SMM_CG_STAGE_INIT_LOCAL 1
#define SMM_CG_STAGE_INIT_APPLY 2
#define SMM_CG_STAGE_ALPHA_LOCAL 3
#define SMM_CG_STAGE_ALPHA_APPLY 4
#define SMM_CG_STAGE_BETA_APPLY 5
int smm_hip_cg_ws_stage_f32(smm_hip_bicgstab_ws* ws, int stage, const float* d_xcur, float* d_x, float eps, smm_hip_stream stream);
int smm_hip_cg_ws_stage_f64(smm_hip_bicgstab_ws* ws, int stage, const double* d_xcur, double* d_x, double eps, smm_hip_stream stream);
/* synchronises `stream`; SolverStatus of the stage-wise CG (iterations / ||r||^2 / done come from smm_hip_bicgstab_ws_result_*) */
int smm_hip_cg_ws_status(const smm_hip_bicgstab_ws* ws, smm_hip_stream stream, int* solver_status);

/* ---- multi-GPU: rows range-partitioned over the GPUs of one node, ONE PROCESS PER GPU ------------------------------------------
 * The reference is a single-process CPU library (no counterpart).  Rank g owns rows
 * [bounds[g], bounds[g+1]) of the n_global x n_global matrix -- its slice of values / positions / start (ref:1243-1259) with
 * GLOBAL column numbers and a local start[] (start[0] == 0) -- and the matching slices of b and x.  Each SpMV first fetches the
 * x-vector halo from the owning ranks (point to point), each dot product is completed by an all-reduce of 1-2 scalars; both run
 * on a side stream of the communicator while the caller's stream computes what does not depend on them.
 *
 * Communicators:
 *   smm_hip_comm_create_rccl   RCCL over xGMI.  Rank 0 calls smm_hip_comm_unique_id and hands the 128 bytes to every rank (any
 *                              out-of-band channel: torch.distributed, MPI, a file); then every rank calls create_rccl.  librccl is
 *                              resolved with dlopen at this point, so single-GPU users of the library never need it.
 *   smm_hip_comm_create_host   the caller moves the bytes: two callbacks (sum all-reduce of a small host array; a batch of sends and
 *                              receives of host buffers).  For tests / rehearsals with several ranks on one GPU, where RCCL cannot run.
 *   smm_hip_comm_create_self   a single rank (no communication).
 * The library's device (smm_hip_init) is the one the communicator uses.  Calls on a communicator and on the distributed matrices
 * built on it are collective: every rank makes the same calls in the same order. */
typedef struct smm_hip_comm smm_hip_comm;
typedef struct smm_hip_dist_csr smm_hip_dist_csr;
#define SMM_HIP_COMM_ID_BYTES 128
#define SMM_COMM_SELF 0
#define SMM_COMM_RCCL 1
#define SMM_COMM_HOST 2
/* in-place sum over all ranks of buf[0..count); dtype is SMM_DTYPE_F32 / F64 / I64.  Return 0 on success. */
typedef int (*smm_hip_host_allreduce_fn)(void* user, void* buf, int count, int dtype);
/* post n_recv receives and n_send sends of host buffers (peer rank, pointer, byte count each) and wait for all of them */
typedef int (*smm_hip_host_sendrecv_fn)(void* user, int n_send, const int* send_peer, void* const* send_buf, const size_t* send_bytes,
                                        int n_recv, const int* recv_peer, void* const* recv_buf, const size_t* recv_bytes);
int smm_hip_comm_unique_id(void* id /* SMM_HIP_COMM_ID_BYTES */);
int smm_hip_comm_create_rccl(int rank, int world, const void* id, smm_hip_comm** out);
int smm_hip_comm_create_host(int rank, int world, smm_hip_host_allreduce_fn allreduce, smm_hip_host_sendrecv_fn sendrecv, void* user,
                             smm_hip_comm** out);
int smm_hip_comm_create_self(smm_hip_comm** out);
int smm_hip_comm_destroy(smm_hip_comm* comm);
int smm_hip_comm_info(const smm_hip_comm* comm, int* rank, int* world, int* kind);
/* the communicator's size as RCCL itself reports it (ncclCommCount); 0 for the other kinds */
int smm_hip_comm_rccl_ranks(const smm_hip_comm* comm, int* count);
/* Failure containment: ncclCommInitRank and every wait on a stream that carries RCCL work are bounded by SMM_HIP_COMM_TIMEOUT_S
 * seconds (environment, default 180); on a time-out -- or on any failure inside a distributed call -- the communicator is aborted
 * (ncclCommAbort), the call returns SMM_HIP_ERR_COMM and every later call on it fails at once: the peers then run into their own
 * bounded wait instead of hanging in the next collective.  The process should exit. */
/* runs every collective the solvers use once and checks the results (collective) */
int smm_hip_comm_selftest(smm_hip_comm* comm);
/* bounds[0..world]: contiguous row ranges with ~equal nonzeros, from a host start[rows+1] */
int smm_hip_partition_rows_by_nnz(const int* start, int rows, int world, int* bounds);
/* This rank's rows as DEVICE arrays (d_start local, d_positions global columns).  Collective: exchanges the column ranges,
 * plans the halo, splits the rows on the device into A_loc (owned columns) and A_rem (halo columns).  The arrays are copied. */
int smm_hip_dist_csr_create_dev_f32(smm_hip_comm* comm, int n_global, const int* bounds, const int* d_start, const int* d_positions,
                                    const float* d_values, smm_hip_dist_csr** out);
int smm_hip_dist_csr_create_dev_f64(smm_hip_comm* comm, int n_global, const int* bounds, const int* d_start, const int* d_positions,
                                    const double* d_values, smm_hip_dist_csr** out);
int smm_hip_dist_csr_destroy(smm_hip_dist_csr* A);
int smm_hip_dist_csr_info(const smm_hip_dist_csr* A, int* n_local, int* ext_len, int* own_offset, int* halo_elements, long long* nnz_loc,
                          long long* nnz_rem);
/* Pieces the halo of this matrix is exchanged in (1 unless SMM_HIP_HALO_CHUNKS asked for 2 .. 4 on EVERY rank when the matrix was created):
 * with k pieces every SpMV issues k exchanges back to back, and the remote block is cut by columns into k parts, part j starting as soon as
 * piece j has landed (row sums are then formed as ((loc + rem_0) + rem_1) + ...: deterministic, rounding differs from the one-piece form). */
int smm_hip_dist_csr_halo_chunks(const smm_hip_dist_csr* D, int* chunks);
/* How this matrix's halo and scalars travel (decided collectively when it was created):
 *   p2p          1: peer to peer (csrc/smm_p2p.h) -- every rank maps every rank's block of fine-grained device memory (hipIpcMemHandle; over
 *                xGMI between GPUs), pushes the boundary slices of a vector straight into the neighbours' landing areas and completes the
 *                dot products by one single-workgroup kernel that writes into / reads from per-rank slots: no collective is launched inside
 *                the loop.  r06: the DEFAULT between processes -- taken whenever every rank could map every peer and passed the self-test
 *                through every path at create time; SMM_HIP_P2P=0 on any rank keeps every rank with the communicator's collectives (0).
 *                2: the hybrid -- the scalars through the slots, the halo through the communicator's grouped send / receive (the halo part of
 *                the self-test failed on some rank, or SMM_HIP_P2P_HALO=0).  Ranks that are threads of ONE process always get 0.
 *                Results: the halo is pure data movement (same bits); the scalars are added in rank order on every rank (deterministic,
 *                identical on all ranks).
 *   relays       relay ranks per halo segment (SMM_HIP_P2P_RELAYS; default world - 4, i.e. 4 at 8 ranks): the segment's direct_share goes
 *                over the link src -> dst, the rest in equal shares src -> relay -> dst over links a nearest-neighbour exchange leaves idle.
 *   halo_first   1: the rows of an updated vector that a peer receives are produced by a small launch of their own and the exchange is
 *                posted right behind it, before the bulk of the update runs (SMM_HIP_HALO_FIRST=0 turns it off); same bits either way. */
int smm_hip_dist_csr_options(const smm_hip_dist_csr* D, int* p2p, int* relays, int* halo_first, double* direct_share);
/* How many SpMVs with a halo this matrix has run so far as ONE launch (csrc/smm_spmv_split.hip: the local half of a workgroup's rows, a
 * bounded wait for the word the exchange raises, the remote half; out[] written once) and how many as TWO (A_loc, then A_rem behind the
 * exchange's event).  One launch is taken whenever both local blocks are in the row-mask encoding with values read at 1 / 2 / 4 lanes per
 * row (what the solvers adopt from 2^20 entries); SMM_HIP_SPLIT_SPMV=0 keeps the two launches.  Same bits either way. */
int smm_hip_dist_csr_matvec_forms(const smm_hip_dist_csr* D, long long* one_launch, long long* two_launches);
/* A THIN remote block: when at most an eighth of this rank's rows hold an entry in another rank's columns (the slabs of a 3-D grid: the two
 * boundary planes) those rows are listed at creation (*rows; 0: the block is not thin) and the second half of an SpMV is a launch over the
 * listed rows only -- out[row] (+|-)= A_rem[row] . halo, the row's entries in order -- instead of a pass over all of out[]; the dot products
 * ride in the local launch and the thin one adds what its rows change (o . w = a . w + d . w,  o . o = a . a + d (a + o)).  *matvecs: SpMVs
 * run in this form so far (they are not counted by smm_hip_dist_csr_matvec_forms).  Taken when the remote block's kernel reads one lane per
 * row and no Jacobi division rides in the epilogue; out[] has the bits of the general form, the dot products differ in the order of their
 * additions.  SMM_HIP_THIN_REMOTE=0 at creation keeps the general second launch. */
int smm_hip_dist_csr_thin_remote(const smm_hip_dist_csr* D, int* rows, long long* matvecs);
/* SpMVs of smm_hip_dist_cg_* on this matrix that formed the next direction themselves (p = beta p_old + r in the load phase of the 2.5-D
 * constant-diagonal kernel, csrc/smm_spmv_march.hip): taken when x is deferred (vectors beyond the caches), the local block is served by that
 * kernel with non-temporal outputs and the remote block is empty or thin; the halo of r then travels instead of the direction's and every rank
 * forms the halo of the new direction itself -- the owner's expression on the owner's operands.  Same bits as the loop that forms p in a
 * launch of its own (smm_hip_set_cg_fuse_p(0)). */
int smm_hip_dist_csr_cg_fused(const smm_hip_dist_csr* D, long long* matvecs);
/* Milliseconds that workgroup 0 of this matrix's one-launch SpMVs has spent, in total, waiting for the word its halo exchange raises after
 * finishing the local half of its rows (100 MHz device clock): what the exchanges cost beyond the compute that ran beside them -- the one-launch
 * form's counterpart of smm_hip_profile_read_waits (`exposed_comm_ms` of `bench.py --gpus N`).  Synchronises the device; reset != 0 clears it. */
int smm_hip_dist_csr_split_wait(smm_hip_dist_csr* D, double* waited_ms, int reset);
/* The peer-to-peer plan of one rank as plain numbers: host arithmetic only (no device, no communicator), the same function the set-up uses.
 * needs[2 q], needs[2 q + 1] = the column range [cmin, cmax) rank q's rows touch; bounds[0 .. world] = the row partition.  Writes records of
 * 8 values into out (capacity in values; *out_count = records): push {0, dst, relay or -1, first global column, position at dst / relay,
 * count, path, relay job}, forward {1, src, dst, staging position, landing position at dst, count, path, job}, land {2, src, offset in the
 * halo-extended vector, landing position, count, paths, 0, 0}.  For tests (tests/test_p2p_plan_cpu.py replays a world of 8 on the CPU). */
int smm_hip_dist_p2p_plan(int world, int rank, const long long* needs, const int* bounds, int relays, double direct_share, long long* out, int out_capacity,
                          int* out_count);
/* the two local blocks (owned by A): a_loc is the square diagonal block a block-Jacobi preconditioner is built on
 * (smm_hip_precond_create(a_loc, kind, &M)); both accept smm_hip_csr_set_kernel */
int smm_hip_dist_csr_local_block(const smm_hip_dist_csr* A, smm_hip_csr** a_loc, smm_hip_csr** a_rem);
/* out = op(lhs, A x) on the owned rows; d_x, d_lhs, d_out are this rank's slices (n_local).  rMult / rMultAdd / rMultSub, ref:1458-1515 */
int smm_hip_dist_spmv_dev_f32(smm_hip_dist_csr* A, int op, const float* d_lhs, const float* d_x, float* d_out, smm_hip_stream stream);
int smm_hip_dist_spmv_dev_f64(smm_hip_dist_csr* A, int op, const double* d_lhs, const double* d_x, double* d_out, smm_hip_stream stream);
/* BiCGStab (ref:2191-2303) / ConjugateGradient (ref:2316-2398) with the semantics of smm_hip_bicgstab_dev_* / smm_hip_cg_dev_*; every
 * rank gets the same status / iterations / resnorm.  M_loc (may be NULL): JACOBI / ILU0 / SGS of this rank's diagonal block, applied
 * block-Jacobi by rank -- the same preconditioner as on one GPU only for JACOBI or world == 1. */
int smm_hip_dist_bicgstab_dev_f32(smm_hip_dist_csr* A, const float* d_b, float* d_x, int maxIterations, float eps, const smm_hip_precond* M_loc,
                                  smm_hip_stream stream, int* solver_status, int* iterations, float* resnorm);
int smm_hip_dist_bicgstab_dev_f64(smm_hip_dist_csr* A, const double* d_b, double* d_x, int maxIterations, double eps, const smm_hip_precond* M_loc,
                                  smm_hip_stream stream, int* solver_status, int* iterations, double* resnorm);
int smm_hip_dist_cg_dev_f32(smm_hip_dist_csr* A, const float* d_b, const float* d_x0, float* d_x, int maxIterations, float eps,
                            smm_hip_stream stream, int* solver_status, int* iterations, float* resnorm2);
int smm_hip_dist_cg_dev_f64(smm_hip_dist_csr* A, const double* d_b, const double* d_x0, double* d_x, int maxIterations, double eps,
                            smm_hip_stream stream, int* solver_status, int* iterations, double* resnorm2);

/* ---- synthetic workload generators (BASELINE.json configs; device-side so 5e8-entry matrices need no host
 *      std::map as in ref:606-618).  d_start[rows+1], d_positions[nnz], d_values[nnz] are DEVICE arrays sized by
 *      the *_nnz query.  Same laws as sparse_matrix_math_amd.generators (numpy), bit for bit. --------------- */
long long smm_hip_gen_poisson2d_nnz(int nx, int ny);
long long smm_hip_gen_stencil3d_nnz(int nx, int ny, int nz);
long long smm_hip_gen_banded_nnz(int n, int k, unsigned long long seed, int max_offset);
int smm_hip_gen_poisson2d_dev_f32(int nx, int ny, int* d_start, int* d_positions, float* d_values, smm_hip_stream stream);
int smm_hip_gen_poisson2d_dev_f64(int nx, int ny, int* d_start, int* d_positions, double* d_values, smm_hip_stream stream);
/* 7-point stencil: diagonal `diag`, lower neighbours `lo`, upper neighbours `hi` (Laplacian: 6,-1,-1;
 * convection-diffusion stand-in for atmosmodd: 6,-1-c,-1+c) */
int smm_hip_gen_stencil3d_dev_f32(int nx, int ny, int nz, float diag, float lo, float hi, int* d_start, int* d_positions, float* d_values, smm_hip_stream stream);
int smm_hip_gen_stencil3d_dev_f64(int nx, int ny, int nz, double diag, double lo, double hi, int* d_start, int* d_positions, double* d_values, smm_hip_stream stream);
/* banded-random symmetric strictly diagonally dominant matrix: A[i][i] = diag_shift +
 * sum|offdiag| (SURVEY's law is diag_shift = 1; A*1 = diag_shift*1, so 1/diag_shift sets the condition number) */
int smm_hip_gen_banded_dev_f32(int n, int k, unsigned long long seed, int max_offset, float diag_shift, int* d_start, int* d_positions, float* d_values, smm_hip_stream stream);
int smm_hip_gen_banded_dev_f64(int n, int k, unsigned long long seed, int max_offset, double diag_shift, int* d_start, int* d_positions, double* d_values, smm_hip_stream stream);
/* rows [row_begin, row_end) only -- what one rank of a row-partitioned run owns: d_start[row_end - row_begin + 1] is local
 * (d_start[0] == 0), d_positions hold GLOBAL columns.  smm_hip_gen_banded_row_start(row) is start[row] of the full matrix
 * in closed form, so the local nnz is row_start(row_end) - row_start(row_begin). */
long long smm_hip_gen_banded_row_start(int n, int k, unsigned long long seed, int max_offset, int row);
int smm_hip_gen_banded_rows_dev_f32(int n, int k, unsigned long long seed, int max_offset, float diag_shift, int row_begin, int row_end,
                                    int* d_start, int* d_positions, float* d_values, smm_hip_stream stream);
int smm_hip_gen_banded_rows_dev_f64(int n, int k, unsigned long long seed, int max_offset, double diag_shift, int row_begin, int row_end,
                                    int* d_start, int* d_positions, double* d_values, smm_hip_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* SMM_HIP_H */
