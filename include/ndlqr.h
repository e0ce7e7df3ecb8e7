/*
 * ndlqr.h -- public C API of the MI355X-native nested-dissection LQR solver.
 *
 * Drop-in boundary for the `ndlqr_*` API of bjack205/rsLQR. Every declaration below names the
 * reference interface it replaces (file:line under /root/reference/src). Struct layouts of the
 * caller-visible types are kept field-for-field (callers poke at them directly, e.g.
 * `solver->soln->data`, `solver->nvars`, `solver->num_threads`, `solver->profile`).
 * One header carries the whole API; the per-module header names of the reference
 * (solver.h, solve.h, nested_dissection.h, ...) exist next to this file as one-line forwards.
 *
 * Where the work happens: everything numerical (leaf solves, separator inner products,
 * Cholesky, triangular solves, Schur updates, the dense Matrix* helpers) runs on the GPU
 * through the C-ABI shim declared in ndlqr_hip.h. There is no CPU fallback: with no HIP
 * device these entry points return NDLQR_ERR_NO_DEVICE (-2) and print to stderr.
 */
#ifndef NDLQR_H_
#define NDLQR_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h> /* (the reference's nested_dissection.h:15 pulls it in; callers rely on that: test/parallel_test.c:120) */

#ifdef __cplusplus
extern "C" {
#endif

#define NDLQR_OK 0
#define NDLQR_ERR_INVALID (-1)   /* reference convention: -1 on NULL / bad argument */
#define NDLQR_ERR_NO_DEVICE (-2) /* additive: no usable HIP device / HIP runtime error */
#define NDLQR_ERR_NOT_SPD (-3)   /* additive: a Cholesky pivot was <= 0 on the device */

/* ------------------------------------------------------------------ matrix.h:71-75 */
typedef struct {
  int rows;
  int cols;
  double* data; /* column-major */
} Matrix;

Matrix NewMatrix(int rows, int cols);                       /* matrix.h:84 */
int MatrixSetConst(Matrix* mat, double val);                /* matrix.h:93 */
int FreeMatrix(Matrix* mat);                                /* matrix.h:104 */
int MatrixNumElements(const Matrix* mat);                   /* matrix.h:113 */
int MatrixGetLinearIndex(const Matrix* mat, int row, int col);
double* MatrixGetElement(const Matrix* mat, int row, int col);
double* MatrixGetElementTranspose(const Matrix* mat, int row, int col, bool istranposed);
int MatrixSetElement(Matrix* mat, int row, int col, double val);
int MatrixCopy(Matrix* dest, Matrix* src);
int MatrixCopyTranspose(Matrix* dest, Matrix* src);
int MatrixScaleByConst(Matrix* mat, double alpha);
double MatrixNormedDifference(Matrix* A, Matrix* B);
int MatrixFlatten(Matrix* mat);
int MatrixFlattenToRow(Matrix* mat);
int PrintMatrix(const Matrix* mat);
int PrintRowVector(const Matrix* mat);

/* ------------------------------------------------------------------ linalg.h:53-153 */
typedef struct {
  char uplo;    /* 'L' */
  int success;  /* 0 = factorisation succeeded */
  char lib;     /* 'H' = HIP device backend (reference: 'B','E','I') */
  void* fact;   /* unused (Eigen-only in the reference) */
  int is_freed;
} CholeskyInfo;

/* Backend selectors of the reference's header (src/linalg.h:17-39; callers print them,
 * test/sample_problem_test.c:172): none of its CPU backends exists here. */
static const int kUseMKL = 0;
static const int kUseEigen = 0;
static const int kUseClap = 0;
static const int kUseBLAS = 0;

enum MatrixLinearAlgebraLibrary { libBLAS = 0, libMKL = 1, libEigen = 2, libInternal = 3, libHIP = 4 };

CholeskyInfo DefaultCholeskyInfo(void);
void FreeFactorization(CholeskyInfo* cholinfo);
/* Dense helpers (linalg.h:83-153). Device-backed: operands are staged to HBM, the kernel of
 * ndlqr_hip.h runs, results are copied back. Meant for tests and small glue, not hot loops. */
int MatrixAddition(Matrix* A, Matrix* B, double alpha);
int MatrixCholeskyFactorize(Matrix* mat);
int MatrixCholeskyFactorizeWithInfo(Matrix* mat, CholeskyInfo* cholinfo);
int MatrixCholeskySolve(Matrix* A, Matrix* b);
int MatrixCholeskySolveWithInfo(Matrix* A, Matrix* b, CholeskyInfo* cholinfo);
void MatrixMultiply(Matrix* A, Matrix* B, Matrix* C, bool tA, bool tB, double alpha, double beta);
void MatrixSymmetricMultiply(Matrix* Asym, Matrix* B, Matrix* C, double alpha, double beta);
void MatrixCopyDiagonal(Matrix* dest, Matrix* src);
enum MatrixLinearAlgebraLibrary MatrixGetLinearAlgebraLibrary(void);
void MatrixPrintLinearAlgebraLibrary(void);
/* The names of the reference's internal backend (src/linalg_custom.h:44-159; its tests call them directly,
 * test/linalg_custom_test.c:11-208). Same argument meaning and return codes, same device kernels as the
 * Matrix* functions above. */
static const int clap_kCholeskySuccess = 0;
static const int clap_kCholeskyFail = -1;
int clap_MatrixAddition(Matrix* A, Matrix* B, double alpha);                 /* B += alpha A */
int clap_MatrixScale(Matrix* A, double alpha);                               /* A *= alpha */
int clap_MatrixMultiply(Matrix* A, Matrix* B, Matrix* C, bool tA, bool tB, double alpha, double beta);
int clap_MatrixTransposeMultiply(Matrix* A, Matrix* B, Matrix* C);           /* C = A' B */
int clap_SymmetricMatrixMultiply(Matrix* Asym, Matrix* B, Matrix* C, double alpha, double beta);
int clap_AddDiagonal(Matrix* A, double alpha);                               /* A += alpha I */
int clap_CholeskyFactorize(Matrix* A);                                       /* lower, in place; -1: pivot <= 0 */
int clap_CholeskySolve(Matrix* L, Matrix* b);                                /* b <- (L L')^-1 b */
int clap_LowerTriBackSub(Matrix* L, Matrix* b, bool istransposed);           /* b <- L^-1 b or L^-T b */

/* ------------------------------------------------------------------ utils.h / linalg_utils.h */
bool IsPowerOfTwo(int x);
static inline int PowerOfTwo(int x) { return 1 << x; }
int LogOfTwo(int x);
int ReadFile(const char* filename, char** out, int* len);
void MatrixLinAlgTimeStart(void);
void MatrixLinAlgTimeStop(void);
void MatrixLinAlgTimeReset(void);
double MatrixGetLinAlgTimeMilliseconds(void);

/* ------------------------------------------------------------------ lqr_data.h:54-132 */
/* One knot point: 0.5 x'Qx + q'x + 0.5 u'Ru + r'u + c ;  x+ = Ax + Bu + d.
 * Q and R are DIAGONALS (n and m entries). A (n x n), B (n x m) column-major.
 * Q is the base pointer of one allocation [Q R q r c A B d] (lqr_data.c:24-49). */
typedef struct {
  int nstates;
  int ninputs;
  double* Q;
  double* R;
  double* q;
  double* r;
  double* c;
  double* A;
  double* B;
  double* d;
} LQRData;

int ndlqr_InitializeLQRData(LQRData* lqrdata, double* Q, double* R, double* q, double* r,
                            double c, double* A, double* B, double* d);
LQRData* ndlqr_NewLQRData(int nstates, int ninputs);
int ndlqr_FreeLQRData(LQRData* lqrdata);
int ndlqr_CopyLQRData(LQRData* dest, LQRData* src);
Matrix ndlqr_GetA(LQRData* lqrdata);
Matrix ndlqr_GetB(LQRData* lqrdata);
Matrix ndlqr_Getd(LQRData* lqrdata);
Matrix ndlqr_GetQ(LQRData* lqrdata);
Matrix ndlqr_GetR(LQRData* lqrdata);
Matrix ndlqr_Getq(LQRData* lqrdata);
Matrix ndlqr_Getr(LQRData* lqrdata);
void ndlqr_PrintLQRData(LQRData* lqrdata);

/* ------------------------------------------------------------------ lqr_problem.h:31-68 */
typedef struct {
  int nhorizon;
  double* x0;
  LQRData** lqrdata;
} LQRProblem;

int ndlqr_InitializeLQRProblem(LQRProblem* lqrproblem, double* x0, LQRData** lqrdata);
LQRProblem* ndlqr_NewLQRProblem(int nstates, int ninputs, int nhorizon);
int ndlqr_FreeLQRProblem(LQRProblem* lqrprob);

/* ------------------------------------------------------------------ json_utils.h:46-76 */
/* Own parser (cJSON is not a dependency). 2-D arrays are arrays of columns; "index" is 1-based. */
LQRData* ndlqr_ReadLQRDataJSONFile(const char* filename);
LQRProblem* ndlqr_ReadLQRProblemJSONFile(const char* filename);
Matrix ReadMatrixJSONFile(const char* filename, const char* name);

/* ------------------------------------------------------------------ binary_tree.h:21-69 */
typedef struct {
  int start; /* inclusive */
  int stop;  /* inclusive for left/right_inds, as the reference fills them */
} UnitRange;

typedef struct BinaryNode_s BinaryNode;
struct BinaryNode_s {
  int idx;
  int level;
  int levelidx;
  UnitRange left_inds;
  UnitRange right_inds;
  BinaryNode* parent;
  BinaryNode* left_child;
  BinaryNode* right_child;
};

typedef struct {
  BinaryNode* root;
  BinaryNode* node_list;
  int num_elements;
  int depth;
} OrderedBinaryTree;

OrderedBinaryTree ndlqr_BuildTree(int nhorizon);
int ndlqr_FreeTree(OrderedBinaryTree* tree);
int ndlqr_GetIndexFromLeaf(const OrderedBinaryTree* tree, int leaf, int level);
int ndlqr_GetIndexLevel(const OrderedBinaryTree* tree, int index);
int ndlqr_GetIndexAtLevel(const OrderedBinaryTree* tree, int index, int level);

/* ------------------------------------------------------------------ nddata.h:40-145 */
typedef struct {
  Matrix lambda; /* (n, w) */
  Matrix state;  /* (n, w) */
  Matrix input;  /* (m, w) */
} NdFactor;

typedef struct {
  int nstates;
  int ninputs;
  int nsegments; /* nhorizon - 1 */
  int depth;
  int width;
  double* data;      /* host mirror, reference layout: block (k,level) at (k + N*level)*(2n+m)*w */
  NdFactor* factors;
} NdData;

Matrix ndlqr_GetLambdaFactor(NdFactor* factor);
Matrix ndlqr_GetStateFactor(NdFactor* factor);
Matrix ndlqr_GetInputFactor(NdFactor* factor);
NdData* ndlqr_NewNdData(int nstates, int ninputs, int nhorizon, int width);
int ndlqr_FreeNdData(NdData* nddata);
int ndlqr_GetNdFactor(NdData* nddata, int index, int level, NdFactor** factor);
void ndlqr_ResetNdData(NdData* nddata);

/* ------------------------------------------------------------------ cholesky_factors.h:30-92 */
typedef struct {
  int depth;
  int nhorizon;
  CholeskyInfo* cholinfo;
  int numfacts;
} NdLqrCholeskyFactors;

NdLqrCholeskyFactors* ndlqr_NewCholeskyFactors(int depth, int nhorizon);
int ndlqr_FreeCholeskyFactors(NdLqrCholeskyFactors* cholfacts);
int ndlqr_GetQFactorizon(NdLqrCholeskyFactors* cholfacts, int index, CholeskyInfo** cholfact);
int ndlqr_GetRFactorizon(NdLqrCholeskyFactors* cholfacts, int index, CholeskyInfo** cholfact);
int ndlqr_GetSFactorization(NdLqrCholeskyFactors* cholfacts, int leaf, int level,
                            CholeskyInfo** cholfact);

/* ------------------------------------------------------------------ solver.h:31-226 */
typedef struct {
  double t_total_ms;
  double t_leaves_ms;
  double t_products_ms;  /* device: separator kernels (products + Cholesky + solves fused) */
  double t_cholesky_ms;  /* device: always 0 (fused into the separator kernels) */
  double t_cholsolve_ms; /* device: always 0 (fused into the separator kernels) */
  double t_shur_ms;      /* device: Schur-update kernels + solution sweep */
  int num_threads;
} NdLqrProfile;

NdLqrProfile ndlqr_NewNdLqrProfile(void);
void ndlqr_ResetProfile(NdLqrProfile* prof);
void ndlqr_CopyProfile(NdLqrProfile* dest, NdLqrProfile* src);
void ndlqr_PrintProfile(NdLqrProfile* profile);
void ndlqr_CompareProfile(NdLqrProfile* base, NdLqrProfile* prof);

typedef struct {
  int nstates;
  int ninputs;
  int nhorizon;
  int depth;
  int nvars;
  OrderedBinaryTree tree;
  Matrix* diagonals; /* (nhorizon, 2): dense Q_k, R_k host mirrors */
  NdData* data;      /* host mirror of the KKT coupling blocks */
  NdData* fact;      /* host mirror of the factorisation (filled by ndlqr_SyncFactorsToHost, or by every ndlqr_Solve under ndlqr_SetFactorMirroring) */
  NdData* soln;      /* rhs in, solution out (always synced by ndlqr_Solve) */
  NdLqrCholeskyFactors* cholfacts;
  double solve_time_ms;
  double linalg_time_ms;
  NdLqrProfile profile;
  int num_threads;   /* accepted and reported; irrelevant on the device */
  /* ---- appended (not in the reference) ---- */
  void* device_ctx;  /* opaque: batch-of-1 device solver */
  unsigned device_flags;     /* NDLQR_FLAG_* used by ndlqr_Solve (ndlqr_SetDeviceFlags; default 0) */
  int device_profiling;      /* ndlqr_SetDeviceProfiling: -1 (default) first solve only, 1 every solve, 0 never */
  int device_profiled;       /* a profiled solve has filled the per-kernel buckets of `profile` */
  NdLqrProfile device_split; /* ... which are kept here for the solves that replay the captured graph */
  int mirror_fact;           /* ndlqr_SetFactorMirroring: ndlqr_Solve leaves the factorisation in `fact` like the reference */
} NdLqrSolver;

NdLqrSolver* ndlqr_NewNdLqrSolver(int nstates, int ninputs, int nhorizon);
int ndlqr_FreeNdLqrSolver(NdLqrSolver* solver);
int ndlqr_InitializeWithLQRProblem(const LQRProblem* lqrprob, NdLqrSolver* solver);
void ndlqr_ResetSolver(NdLqrSolver* solver);
void ndlqr_PrintSolveSummary(NdLqrSolver* solver);
int ndlqr_GetNumVars(NdLqrSolver* solver);
int ndlqr_SetNumThreads(NdLqrSolver* solver, int num_threads);
int ndlqr_GetNumThreads(NdLqrSolver* solver);
int ndlqr_PrintSolveProfile(NdLqrSolver* solver);
NdLqrProfile ndlqr_GetProfile(NdLqrSolver* solver);

/* ------------------------------------------------------------------ solve.h:37-72 */
/* Factor + substitute on the device. Returns 0; additionally NDLQR_ERR_NO_DEVICE /
 * NDLQR_ERR_NOT_SPD (the reference always returns 0, solve.c:189). */
int ndlqr_Solve(NdLqrSolver* solver);
Matrix ndlqr_GetSolution(NdLqrSolver* solver);
int ndlqr_CopySolution(NdLqrSolver* solver, double* soln);
/* additive: per-kernel HIP-event profiling of ndlqr_Solve (fills the buckets of solver->profile like the
 * reference's always-on profiler, src/solve.c:15-25,184-188). Default: the FIRST solve of a solver is profiled (eager
 * launches, an event pair per kernel); every later one replays one captured hipGraph -- copies up, launch chain, copy
 * down: a third of the wall time of a small solve -- and reports its own t_total_ms / solve_time_ms beside the
 * per-kernel split of the last profiled solve. on = 1: profile every solve; on = 0: never (buckets stay 0). */
int ndlqr_SetDeviceProfiling(NdLqrSolver* solver, int on);
/* additive: NDLQR_FLAG_* bits ndlqr_Solve runs with (default 0 = fast mode, solution only; the same
 * launch sequence ndlqr_SolveBatch times). NDLQR_FLAG_STRICT_FP reproduces the reference's default
 * build bit for bit, NDLQR_FLAG_KEEP_FACT materialises the factor array during every solve. */
int ndlqr_SetDeviceFlags(NdLqrSolver* solver, unsigned flags);
/* additive: copy the device factorisation into solver->fact->data (reference layout). Without
 * NDLQR_FLAG_KEEP_FACT on the solve the (still resident) problem is factored once more for it. */
int ndlqr_SyncFactorsToHost(NdLqrSolver* solver);
/* additive: on = 1 makes every ndlqr_Solve of this solver end the way the reference's does (src/solve.c:120-131,
 * src/nddata.h:83-93): with the complete factorisation in solver->fact->data -- the solve runs with
 * NDLQR_FLAG_KEEP_FACT and the factor array comes down with the solution (N K (2n+m) n doubles: 1.5 MB at (6,3,256)).
 * Default 0 (solution only; ndlqr_SyncFactorsToHost on demand), or 1 when the environment has
 * NDLQR_SOLVE_MIRRORS_FACT=1 at ndlqr_NewNdLqrSolver -- for callers that read solver->fact behind an unmodified
 * ndlqr_Solve. */
int ndlqr_SetFactorMirroring(NdLqrSolver* solver, int on);

/* ------------------------------------------------------------------ nested_dissection.h:39-147 */
/* Stage functions on the host mirrors; each runs its dense math through the device-backed
 * Matrix* helpers above (tests / debugging; the hot path is ndlqr_Solve / ndlqr_SolveBatch). */
int ndlqr_SolveLeaf(NdLqrSolver* solver, int index);
int ndlqr_SolveLeaves(NdLqrSolver* solver);
int ndlqr_FactorInnerProduct(NdData* data, NdData* fact, int index, int data_level,
                             int fact_level);
int ndlqr_SolveCholeskyFactor(NdData* fact, CholeskyInfo* cholinfo, int index, int level,
                              int upper_level);
bool ndlqr_ShouldCalcLambda(OrderedBinaryTree* tree, int index, int i);
int ndlqr_UpdateShurFactor(NdData* fact, NdData* soln, int index, int i, int level,
                           int upper_level, bool calc_lambda);
int ndlqr_ComputeShurCompliment(NdLqrSolver* solver, int index, int level, int upper_level);

/* ================================================================== additive: batch API */
/*
 * A batch of independent LQR problems of identical (nstates, ninputs, nhorizon) solved in one
 * launch sequence on one GPU (SURVEY.md 8b "New, additive"). ndlqr_Solve == batch of 1.
 *
 * Flat host layout accepted by ndlqr_InitializeBatchFlat, problem p at offset p*stride:
 *   A [batch][N][n*n] column-major, B [batch][N][n*m] column-major,
 *   Q,q,d [batch][N][n], R,r [batch][N][m], x0 [batch][n]
 * (every knot carries every field, like LQRData; A,B,R,r,d of the last knot are unused).
 */
typedef struct NdLqrBatchSolver NdLqrBatchSolver;

#define NDLQR_FLAG_STRICT_FP 1u   /* separate mul/add (no FMA): bit-reproduces the reference's
                                     default build; slower. Default: fused multiply-add. */
#define NDLQR_FLAG_GENERIC 2u     /* force the runtime-sized kernels even when a size-specialised
                                     variant exists (cross-check) */
#define NDLQR_FLAG_PROFILE 4u     /* bracket every kernel with HIP events */
#define NDLQR_FLAG_KEEP_FACT 8u   /* also materialise the complete factor array on the device (what
                                     the reference leaves in solver->fact); needed by
                                     ndlqr_CopyBatchFactors. Off: only the solution is produced. */

#define NDLQR_FLAG_KEEP_RECORDS 16u /* fast mode: keep what a right-hand-side re-solve needs (separator
                                     records + Cholesky factors, ~3.5 KB per knot at (12,4)) without
                                     materialising the factor array: ndlqr_SolveBatchRhsOnly works,
                                     ndlqr_CopyBatchFactors does not. Size-specialised shapes and every
                                     other one up to 128 states; beyond (the knot-based kernels) the factor
                                     array is kept instead, as with NDLQR_FLAG_KEEP_FACT. */
/* Reach of the modes by block size (runtime-sized kernels; the size-specialised instances of
 * rslqr_amd/csrc/small_instances.def support every mode): EVERY mode works for every block size the device memory holds
 * (round 4; tested up to (256,32)). What changes with the size is the speed: the default fast mode and
 * NDLQR_FLAG_KEEP_RECORDS run the separator-only schedule on the matrix cores up to 128 states (n + m + 4 staged columns
 * within the 160 KB of LDS; (128,32) and everything beyond: the knot-based kernels). NDLQR_FLAG_STRICT_FP and
 * NDLQR_FLAG_KEEP_FACT always run the knot-based kernels, whose separator kernel keeps S-bar and the whole right-hand-side
 * panel in LDS up to about 82 states (tile-filling block sizes, n a multiple of 16, up to 112) and in global memory beyond
 * (one scratch pair per level-0 separator, allocated by the first such solve; NDLQR_ERR_INVALID with
 * ndlqr_hip_last_error() = "global scratch ... does not fit" if the device is too small for the batch). The factor-based
 * rhs-only re-solve reads the factor where it lies beyond ~140 states. */

NdLqrBatchSolver* ndlqr_NewBatchSolver(int nstates, int ninputs, int nhorizon, int batch,
                                       int device);
int ndlqr_FreeBatchSolver(NdLqrBatchSolver* bs);
int ndlqr_BatchSetFlags(NdLqrBatchSolver* bs, unsigned flags);
unsigned ndlqr_BatchGetFlags(const NdLqrBatchSolver* bs);
int ndlqr_InitializeBatch(NdLqrBatchSolver* bs, const LQRProblem* const* probs, int count);
int ndlqr_InitializeBatchFlat(NdLqrBatchSolver* bs, const double* A, const double* B,
                              const double* Q, const double* R, const double* q,
                              const double* r, const double* d, const double* x0);
/* Same flat layout, but DEVICE pointers (the problem is produced on the GPU): packed by a kernel
 * on the solver's stream, no host round trip. */
int ndlqr_InitializeBatchFlatDevice(NdLqrBatchSolver* bs, const double* dA, const double* dB,
                                    const double* dQ, const double* dR, const double* dq,
                                    const double* dr, const double* dd, const double* dx0);
/* Seeded synthetic problems, problem p seeded with seed0 + p; generated on the
 * host, packed and uploaded. */
int ndlqr_InitializeBatchSynthetic(NdLqrBatchSolver* bs, uint64_t seed0);
int ndlqr_SolveBatch(NdLqrBatchSolver* bs);      /* launch + wait */
/* Factor / solve split (MPC re-solves): replace q, r, d, x0 (flat layout as above) and run only
 * the solution sweep against the factorisation cached by the last ndlqr_SolveBatch; needs
 * NDLQR_FLAG_KEEP_FACT (or, in fast mode up to 128 states, the lighter
 * NDLQR_FLAG_KEEP_RECORDS) to have been set for that solve. A, B, Q, R are those of that solve. */
int ndlqr_BatchSetRhsFlat(NdLqrBatchSolver* bs, const double* q, const double* r, const double* d,
                          const double* x0);
int ndlqr_SolveBatchRhsOnly(NdLqrBatchSolver* bs);
/* additive: nrhs sets of right-hand sides for the whole batch against the factorisation kept by the last ndlqr_SolveBatch
 * with NDLQR_FLAG_KEEP_RECORDS -- q, d [nrhs][batch][N][n], r [nrhs][batch][N][m], x0 [nrhs][batch][n] (the flat layout above
 * with a leading [nrhs]) in, solutions [nrhs][batch][nvars] out (host arrays; blocking). The reference solves one
 * right-hand side per factorisation (src/nddata.h:70-75); here e.g. ONE problem of (12,4,256) takes 1024 right-hand sides
 * (sampled initial states / cost offsets of one model) at the rate its right-hand sides and solutions move through the
 * HBM. Size-specialised shapes on the level-per-launch schedule (batch x N / 4 > 2048, or NDLQR_TREE=0 in the environment). */
int ndlqr_SolveBatchMultiRhs(NdLqrBatchSolver* bs, int nrhs, const double* q, const double* r, const double* d,
                             const double* x0, double* soln);
/* The same for a slice of every solution -- knots [knot0, knot0 + nknots), blocks = NDLQR_SOLN_* (below) ->
 * out [nrhs][batch][nknots][width]; only those knots are computed by the last launch and brought down. */
int ndlqr_SolveBatchMultiRhsSlices(NdLqrBatchSolver* bs, int nrhs, const double* q, const double* r, const double* d,
                                   const double* x0, int knot0, int nknots, unsigned blocks, double* out);
int ndlqr_SolveBatchAsync(NdLqrBatchSolver* bs); /* enqueue on the solver's stream */
int ndlqr_BatchSynchronize(NdLqrBatchSolver* bs);
/* One MPC step, asynchronous: new q, r, d, x0 (flat host layout as above) up, factor + solve against the resident
 * A, B, Q, R, the solutions [batch][nvars] down into `soln` (the array ndlqr_CopyBatchSolutions fills); q, r, d may each
 * be NULL: that part of the right-hand side stays what its most recent writer -- ndlqr_InitializeBatch*,
 * ndlqr_BatchSetRhsFlat or an earlier step -- left (an MPC iteration often replaces x0 alone; full steps and x0-only
 * steps may be mixed freely: the library keeps track of which buffer set's copy is behind in what). Consecutive
 * steps alternate between the two buffer sets of the solve pipeline, so the transfers of one step run beside the
 * kernels of the other; `soln` of a step is complete after ndlqr_BatchSynchronize, or -- one step behind --
 * ndlqr_BatchSynchronizePrevious. Host arrays from ndlqr_HostAlloc (pinned) keep the copies asynchronous;
 * pageable memory works but blocks. What the reference does per MPC iteration with ndlqr_ResetSolver +
 * ndlqr_InitializeWithLQRProblem + ndlqr_Solve + ndlqr_CopySolution (src/solve.h:20-32).
 * With NDLQR_FLAG_KEEP_RECORDS (size-specialised shapes on the level-per-launch schedule): a step never changes A, B, Q,
 * R, so the first step factors and every further one is the right-hand-side re-solve on the kept records -- 0.51 instead
 * of 0.68 ms per step of 1024 x (12,4,256) with x0 up and u of knot 0 down -- until new inputs are uploaded; the steps
 * are then stream-ordered on one buffer set (results equal a full solve's to rounding, not bit for bit). */
int ndlqr_BatchStepAsync(NdLqrBatchSolver* bs, const double* q, const double* r, const double* d,
                         const double* x0, double* soln);
/* Waits for the step before the most recent one; NDLQR_ERR_NOT_SPD when a Cholesky pivot of that step (or an earlier,
 * unreported one) was not positive -- its `soln` is then not a solution. */
int ndlqr_BatchSynchronizePrevious(NdLqrBatchSolver* bs);
/* What a step brings down: knots [knot0, knot0 + nknots) of every problem, of each knot the blocks of `blocks`, in
 * the reference's order lambda, state, input -> soln = [batch][nknots][width], width = n per NDLQR_SOLN_LAMBDA /
 * NDLQR_SOLN_STATE + m for NDLQR_SOLN_INPUT. nknots = 0: back to every solution [batch][nvars] (the default; what
 * ndlqr_CopySolution hands back, src/solve.c:192-201). An MPC loop that applies u_0 asks for (0, 1, NDLQR_SOLN_INPUT):
 * 32 KB instead of 59 MB per 1024 problems of (12,4,256). ndlqr_CopyBatchSolutionSlices: the same slice of the most
 * recent solve, synchronously.
 * NDLQR_SOLN_ONLY (or-ed into `blocks`): the caller wants NOTHING but the selection, so a step may skip the part of the
 * back-substitution that produces the other knots -- (0, 1, NDLQR_SOLN_INPUT | NDLQR_SOLN_ONLY) is the MPC step that
 * computes u_0 alone: the forward pass over the whole horizon, the top-down sweep, and the eight knots around knot 0
 * instead of all N (0.63 -> 0.43 ms per 1024 x (12,4,256), 0.48 -> 0.30 with NDLQR_FLAG_KEEP_RECORDS). After such a step
 * the solver holds only that slice: ndlqr_CopyBatchSolutionSlices inside it works, everything that needs the whole
 * vector (ndlqr_CopyBatchSolution(s), ndlqr_BatchKKTResidual, the device-side pack) returns -1 until the next solve or
 * step without the bit. ((64,16,512) x 256: 11.6 -> 9.4 ms, 8.0 -> 5.5 with NDLQR_FLAG_KEEP_RECORDS.) The knot-based
 * kernels (strict mode, KEEP_FACT, beyond 128 states) compute everything regardless. */
#define NDLQR_SOLN_LAMBDA 1u
#define NDLQR_SOLN_STATE 2u
#define NDLQR_SOLN_INPUT 4u
#define NDLQR_SOLN_ONLY 8u
int ndlqr_BatchSetStepSelection(NdLqrBatchSolver* bs, int knot0, int nknots, unsigned blocks);
/* The same economy for a loop that replaces A, B, Q, R too (ndlqr_InitializeBatchFlat[Device] per iteration): factor +
 * solve of the resident problems, computing and delivering knots [knot0, knot0 + nknots) alone -- out =
 * [batch][nknots][width] in host, pinned or device memory; asynchronous like ndlqr_SolveBatchAsync (complete after
 * ndlqr_BatchSynchronize), and like a step with NDLQR_SOLN_ONLY it leaves nothing but that slice behind. */
int ndlqr_SolveBatchSlicesAsync(NdLqrBatchSolver* bs, int knot0, int nknots, unsigned blocks, double* out);
int ndlqr_CopyBatchSolutionSlices(NdLqrBatchSolver* bs, int knot0, int nknots, unsigned blocks, double* out);
/* Time-axis sharding: one problem (or a small batch) over G ranks, rank g on knots [g N / G, (g + 1) N / G) -- for jobs
 * with fewer problems than GPUs (details and limits: ndlqr_hip.h). Per solve, on every rank:
 * Factor -> ExportTop(buf) -> sum buf over the ranks -> ImportTop(buf) -> Finish -> ndlqr_BatchSynchronize. */
int ndlqr_BatchTimeShardTopDoubles(NdLqrBatchSolver* bs, int G);
int ndlqr_BatchTimeShardFactor(NdLqrBatchSolver* bs, int g, int G);
int ndlqr_BatchTimeShardExportTop(NdLqrBatchSolver* bs, int G, double* buf);
int ndlqr_BatchTimeShardImportTop(NdLqrBatchSolver* bs, int G, const double* buf);
int ndlqr_BatchTimeShardFinish(NdLqrBatchSolver* bs, int g, int G);
void* ndlqr_HostAlloc(size_t bytes); /* pinned host memory (NULL: no device / no memory) */
void ndlqr_HostFree(void* p);
/* Device memory, for loops that live on the GPU: ndlqr_BatchStepAsync takes q, r, d, x0 and soln in the solver's device
 * memory as they are (no transfer at all; any mix with host pointers works). ndlqr_DeviceCopy: synchronous, any direction. */
void* ndlqr_DeviceAlloc(size_t bytes);
void ndlqr_DeviceFree(void* p);
int ndlqr_DeviceCopy(void* dst, const void* src, size_t bytes);
int ndlqr_BatchNumVars(const NdLqrBatchSolver* bs);
int ndlqr_BatchSize(const NdLqrBatchSolver* bs);
int ndlqr_CopyBatchSolution(NdLqrBatchSolver* bs, int p, double* soln);    /* nvars doubles */
int ndlqr_CopyBatchSolutions(NdLqrBatchSolver* bs, double* soln);          /* batch*nvars */
/* the same [batch][nvars] array written to DEVICE memory by a kernel on the solver's stream
 * (asynchronous; e.g. the send buffer of an all_gather of the shards' solutions) */
int ndlqr_CopyBatchSolutionsDevice(NdLqrBatchSolver* bs, double* dsoln);
int ndlqr_CopyBatchFactors(NdLqrBatchSolver* bs, int p, double* fact);     /* reference layout */
int ndlqr_BatchCholeskyFailures(NdLqrBatchSolver* bs);
/* KKT residual ||K z - b||_2 and ||b||_2 of every problem's resident solution against its raw
 * data, evaluated on the device (batch doubles each; bnorm may be NULL). */
int ndlqr_BatchKktResiduals(NdLqrBatchSolver* bs, double* res, double* bnorm);
double ndlqr_BatchSolveTimeMs(const NdLqrBatchSolver* bs); /* HIP-event time of last solve */
void* ndlqr_BatchDeviceContext(NdLqrBatchSolver* bs);      /* NdlqrHipCtx* (ndlqr_hip.h) */

/* Seeded synthetic problem generator (host, bit-reproducible). */
int ndlqr_GenerateSyntheticFlat(int nstates, int ninputs, int nhorizon, uint64_t seed, double* A,
                                double* B, double* Q, double* R, double* q, double* r, double* d,
                                double* x0);
LQRProblem* ndlqr_NewSyntheticLQRProblem(int nstates, int ninputs, int nhorizon, uint64_t seed);

const char* ndlqr_Version(void);

#ifdef __cplusplus
}
#endif
#endif /* NDLQR_H_ */
