/*
 * ndlqr_hip.h -- thin C-ABI shim between the plain-C host library and the HIP kernels.
 *
 * Plain pointers and sizes only. This is the device boundary that SURVEY.md (section 1) inserts
 * between the reference's L2 driver (src/solve.c) and its L1/L0 numerics
 * (src/nested_dissection.c, src/linalg_custom.c): everything behind these calls runs on
 * gfx950. Implemented in rslqr_amd/csrc/ndlqr_hip.hip.
 *
 * Device data model (all fp64):
 *   inputs  AB  [batch][N][n][n+m]   row i of knot k = [A_k(i,:) | B_k(i,:)]   (A,B row-major)
 *           QR  [batch][N][n+m]      diag(Q_k) then diag(R_k)
 *           rhs [batch][N][2n+m]     initial right-hand side, already negated like
 *                                    src/solver.c:188-190: knot k = [-(x0 | d_{k-1}); -q_k; -r_k]
 *                                    (kept untouched by the solve, so solves can be repeated)
 *   state   F   [batch][K][N][2n+m][n]  factor block (level p, knot k), ROW-major; rows
 *                                       0..n-1 lambda, n..2n-1 state, 2n..2n+m-1 input.
 *                                       (reference: column-major sub-blocks, src/nddata.c:40-53)
 *           z   [batch][N][2n+m]     rhs in / solution out, same order as the reference
 *           info[batch+1]            non-positive Cholesky pivots per problem, batch total last (cumulative
 *                                    over solves; ndlqr_hip_cholesky_failures reports those since the last
 *                                    synchronisation)
 */
#ifndef NDLQR_HIP_H_
#define NDLQR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct NdlqrHipCtx NdlqrHipCtx;

int ndlqr_hip_device_count(void);
const char* ndlqr_hip_last_error(void);

/* replaces the allocation half of ndlqr_NewNdLqrSolver (src/solver.c:61-96) for a batch */
NdlqrHipCtx* ndlqr_hip_create(int nstates, int ninputs, int nhorizon, int batch, int device);
/* ... with creation options: NDLQR_CREATE_NO_PAD keeps the caller's block size on the device (a block size without
 * a size-specialised instance otherwise runs zero-padded inside one, with another array layout: raw device pointers
 * -- ndlqr_hip_device_pointers -- need the caller's own). The environment variable NDLQR_NO_PAD=1 does the same for
 * every context of the process. */
#define NDLQR_CREATE_NO_PAD 1u
NdlqrHipCtx* ndlqr_hip_create_ex(int nstates, int ninputs, int nhorizon, int batch, int device, unsigned create_flags);
void ndlqr_hip_destroy(NdlqrHipCtx* ctx);
int ndlqr_hip_set_flags(NdlqrHipCtx* ctx, unsigned flags); /* NDLQR_FLAG_* of ndlqr.h */
unsigned ndlqr_hip_get_flags(const NdlqrHipCtx* ctx);
/* Use an externally owned hipStream_t (e.g. torch's current stream); NULL = own stream. */
int ndlqr_hip_set_stream(NdlqrHipCtx* ctx, void* hip_stream);
void* ndlqr_hip_get_stream(NdlqrHipCtx* ctx);

/* H2D of packed inputs for problems [p0, p0+count) (host layout = device layout above).
 * Replaces the data movement of ndlqr_InitializeWithLQRProblem (src/solver.c:122-194); the
 * zero-padded KKT `data` array is never materialised (kernels read A,B,Q,R and the rhs directly). */
int ndlqr_hip_upload_inputs(NdlqrHipCtx* ctx, int p0, int count, const double* AB,
                            const double* QR, const double* rhs);
/* The same packing done ON the device from flat arrays that already live in HBM (reference
 * layout: A [batch][N][n*n] and B [batch][N][n*m] column-major, Q,q,d [batch][N][n], R,r [batch][N][m],
 * x0 [batch][n]); asynchronous on the context's stream. The caller orders the pack kernel behind
 * whatever produced the inputs: synchronise the producer stream first, or make it the context's
 * stream with ndlqr_hip_set_stream. Invalidates a cached factorisation / cached records. */
int ndlqr_hip_pack_flat_device(NdlqrHipCtx* ctx, const double* A, const double* B, const double* Q,
                               const double* R, const double* q, const double* r, const double* d,
                               const double* x0);
/* Device pointers for zero-copy producers (order: AB, QR, rhs, F, z). Allocates the factor array and
 * sets the pipeline depth to 1, so that z is THE solution buffer from then on. */
int ndlqr_hip_device_pointers(NdlqrHipCtx* ctx, void** out5);

/* ndlqr_Solve (src/solve.c:38-190) for the whole batch: leaf kernel + one kernel per tree level
 * (inner products, Cholesky, triangular solves, Schur updates, rhs sweep fused). Async on the
 * context's stream; HIP events bracket the sequence. */
int ndlqr_hip_solve_async(NdlqrHipCtx* ctx);
int ndlqr_hip_synchronize(NdlqrHipCtx* ctx);
/* One-shot solve from / into pinned host staging owned by the context -- what the drop-in ndlqr_Solve (host memory in,
 * host memory out on every call, src/solve.c:38-201) runs: ndlqr_hip_staged_io hands out the staging (AB, QR, rhs in
 * the packed layout above but in the CALLER's block size, z = [batch][N][2n+m] coming back), the caller packs into
 * it, ndlqr_hip_solve_staged replays one captured graph -- the three copies up, the launch chain, the copy down --
 * and waits for it: one launch and one synchronisation per call. Stream-ordered (pipeline depth 1 from the first
 * call on). Returns like ndlqr_hip_synchronize; ndlqr_hip_cholesky_failures tells about pivots. */
int ndlqr_hip_staged_io(NdlqrHipCtx* ctx, double** AB, double** QR, double** rhs, double** z);
int ndlqr_hip_solve_staged(NdlqrHipCtx* ctx);
/* Solve pipeline. Depth 2 (default; NDLQR_PIPELINE): consecutive ndlqr_hip_solve_async calls of one
 * context alternate between two sets of output buffers (records, accumulators, solution), each on its
 * own stream, so that a solve starts while the previous one is still in its thinly populated upper tree
 * levels / its HBM-bound back-substitution. Every solve is complete; ndlqr_hip_synchronize waits for all
 * of them and the download functions return the most recent one. Applies to the schedules that keep
 * nothing but the solution (no factor array, no kept records, no per-kernel events, own stream); depth 1
 * = strictly stream-ordered solves. Inputs must not be replaced while solves are in flight: the upload /
 * pack functions wait for them first. */
int ndlqr_hip_set_pipeline_depth(NdlqrHipCtx* ctx, int depth);
int ndlqr_hip_pipeline_depth(const NdlqrHipCtx* ctx);
/* Factor / solve split: new right-hand side(s) against the factorisation cached by the last
 * ndlqr_hip_solve_async with NDLQR_FLAG_KEEP_FACT (the reference's solution sweep,
 * src/solve.c:137-182, alone). rhs layout as in ndlqr_hip_upload_inputs. */
int ndlqr_hip_upload_rhs(NdlqrHipCtx* ctx, int p0, int count, const double* rhs);
int ndlqr_hip_solve_rhs_async(NdlqrHipCtx* ctx);
double ndlqr_hip_last_solve_ms(NdlqrHipCtx* ctx); /* valid after synchronize */
/* Several right-hand sides per problem against one kept factorisation each (the reference's NdData holds a single
 * right-hand side, src/nddata.h:70-75): nrhs sets of right-hand sides for the whole batch, flat HOST arrays in the layout of
 * ndlqr_BatchSetRhsFlat with a leading [nrhs] -- q, d [nrhs][batch][N][n], r [nrhs][batch][N][m], x0 [nrhs][batch][n] --,
 * solutions [nrhs][batch][nvars] into soln. Needs the records of a solve with NDLQR_FLAG_KEEP_RECORDS on a size-specialised
 * shape in the level-per-launch form (batch x N / 4 > 2048 or NDLQR_TREE=0), else NDLQR_ERR_INVALID. Blocking;
 * ndlqr_hip_last_solve_ms then reports the device time of the solve kernels alone. */
int ndlqr_hip_solve_multi_rhs(NdlqrHipCtx* ctx, int nrhs, const double* q, const double* r, const double* d,
                              const double* x0, double* soln);
/* ... of which only knots [knot0, knot0 + nknots), blocks `blocks` (NDLQR_SOLN_*) are computed and brought down:
 * out = [nrhs][batch][nknots][width] (u of knot 0 for a thousand sampled initial states of one model: 32 KB instead of 59 MB) */
int ndlqr_hip_solve_multi_rhs_slices(NdlqrHipCtx* ctx, int nrhs, const double* q, const double* r, const double* d,
                                     const double* x0, int knot0, int nknots, unsigned blocks, double* out);

/* One MPC step, asynchronous: a new right-hand side up (flat host arrays in the reference's layout: q, d
 * [batch][N][n], r [batch][N][m], x0 [batch][n] -- what ndlqr_InitializeWithLQRProblem reads from the problem,
 * src/solver.c:141-190), packed and negated by a kernel, factor + solve, the solutions [batch][nvars] (the layout of
 * ndlqr_hip_download_solutions, src/solve.c:192-201) down into `soln`. q, r, d may each be NULL: that part of the
 * right-hand side stays what its most recent writer left -- an upload, the device-side packing or an earlier step (an
 * MPC iteration often replaces x0 alone). There is ONE logical right-hand side; each buffer set of the pipeline has a
 * copy, and a solve or step that lands on a set whose copy is behind in a part it does not itself replace first
 * copies that part over (only a change of flow does: full steps followed by x0-only steps, a plain solve behind
 * steps). Everything is ordered on the stream of the
 * step's buffer set, so with the two-deep pipeline the copies of one step run beside the kernels of the other. Give
 * pinned host memory (ndlqr_hip_host_alloc): copies from / to pageable memory are staged by the runtime and block.
 * Pointers into THIS device's memory are taken as they are -- the pack kernels read q, r, d, x0 and write `soln` there
 * directly, nothing crosses the host link: the step of a loop whose states and inputs live on the GPU.
 * `soln` of a step is complete after ndlqr_hip_synchronize (every step) or, one step behind,
 * ndlqr_hip_synchronize_previous (the step before the most recent one; NDLQR_ERR_NOT_SPD when that step met a
 * non-positive pivot). */
int ndlqr_hip_step_async(NdlqrHipCtx* ctx, const double* q, const double* r, const double* d, const double* x0,
                         double* soln);
int ndlqr_hip_synchronize_previous(NdlqrHipCtx* ctx);
/* What a step brings down (ndlqr.h: ndlqr_BatchSetStepSelection) / the same slice of the latest solve, synchronously. */
int ndlqr_hip_set_step_selection(NdlqrHipCtx* ctx, int knot0, int nknots, unsigned blocks);
/* Factor + solve of the resident problems delivering a slice alone (ndlqr.h: ndlqr_SolveBatchSlicesAsync). */
int ndlqr_hip_solve_slices_async(NdlqrHipCtx* ctx, int knot0, int nknots, unsigned blocks, double* out);
int ndlqr_hip_download_selection(NdlqrHipCtx* ctx, int knot0, int nknots, unsigned blocks, double* out);
/* Time-axis sharding: ONE problem (or a small batch) solved by G ranks, each working on a chunk of
 * N / G consecutive knots of the horizon -- for jobs with fewer problems than GPUs; the batch axis stays the sharding
 * unit otherwise. Every rank holds the whole problem's inputs (upload as usual) and runs, for its chunk g:
 *     ndlqr_hip_time_shard_factor(ctx, g, G)      bottom kernel + the tree levels inside the chunk (asynchronous)
 *     ndlqr_hip_time_shard_export(ctx, G, buf)    the G - 1 accumulator slots between the chunks, packed
 *                                                 [G - 1][batch][slot] (ndlqr_hip_time_shard_top_doubles); waits
 *     -- the caller sums `buf` over the ranks: one all-reduce (RCCL over xGMI when every rank has its own GPU) --
 *     ndlqr_hip_time_shard_import(ctx, G, buf)
 *     ndlqr_hip_time_shard_finish(ctx, g, G)      the top log2(G) levels (redundantly on every rank: no multipliers
 *                                                 travel back), top-down sweep, back-substitution of the chunk
 * followed by ndlqr_hip_synchronize. The solution array then holds the knots [g N / G, (g + 1) N / G) of every problem
 * (ndlqr_hip_download_solutions hands back whole vectors: the other knots are stale). buf: host or device memory.
 * Default fast mode, size-specialised block sizes with a matrix-core instance, G a power of two, N / G >= 16,
 * N <= 64 K / (8 n) knots. The loop being split: src/solve.c:68-134 (factor), :137-182 (solve). */
int ndlqr_hip_time_shard_top_doubles(NdlqrHipCtx* ctx, int G);
int ndlqr_hip_time_shard_factor(NdlqrHipCtx* ctx, int g, int G);
int ndlqr_hip_time_shard_export(NdlqrHipCtx* ctx, int G, double* buf);
int ndlqr_hip_time_shard_import(NdlqrHipCtx* ctx, int G, const double* buf);
int ndlqr_hip_time_shard_finish(NdlqrHipCtx* ctx, int g, int G);
/* Pinned host memory for the transfer functions (hipHostMalloc): copies from / to it are asynchronous and run at the
 * rate of the host link. NULL when no device / no memory. */
void* ndlqr_hip_host_alloc(size_t bytes);
void ndlqr_hip_host_free(void* p);
/* device memory of the current device / a synchronous copy between any two of host, pinned and device memory: for
 * callers without HIP headers whose MPC loop lives on the GPU (ndlqr_hip_step_async takes device pointers) */
void* ndlqr_hip_device_alloc(size_t bytes);
void ndlqr_hip_device_free(void* p);
int ndlqr_hip_copy(void* dst, const void* src, size_t bytes);

/* D2H. soln: count*nvars doubles, nvars = (2n+m)N - m (src/solve.c:192-201).
 * fact: one problem, converted to the reference's NdData layout, N*K*(2n+m)*n doubles. */
int ndlqr_hip_download_solutions(NdlqrHipCtx* ctx, int p0, int count, double* soln);
int ndlqr_hip_download_rhs_blocks(NdlqrHipCtx* ctx, int p, double* z_full); /* N*(2n+m) */
int ndlqr_hip_download_factors(NdlqrHipCtx* ctx, int p, double* fact); /* needs NDLQR_FLAG_KEEP_FACT */
int ndlqr_hip_factors_valid(const NdlqrHipCtx* ctx); /* 1: the device holds the factor array of the last solve */
/* Solutions of the whole batch packed as [batch][nvars] into DEVICE memory `dst` (e.g. the send
 * buffer of an RCCL all_gather of the shards' solutions), by a kernel on the context's stream. */
int ndlqr_hip_pack_solutions_device(NdlqrHipCtx* ctx, double* dst);
/* Residual of the resident solution against the raw problem, per problem, computed on the device:
 * res[b] = ||K z - b||_2, bnorm[b] = ||b||_2 (bnorm may be NULL); the rows are those of the
 * reference's KKT system (src/solver.c:122-194). batch doubles each. */
int ndlqr_hip_kkt_residual(NdlqrHipCtx* ctx, double* res, double* bnorm);
int ndlqr_hip_cholesky_failures(NdlqrHipCtx* ctx);
/* Name of the launch sequence the last solve used (for reports): "reduced", "reduced-fused2" (the (12,4) instance; NDLQR_FUSE2=1/0), "reduced-tree",
 * "reduced-records", "knot-lean", "knot-strict", "knot-keep", "generic-reduced",
 * "generic-reduced-records", "generic-lean",
 * "generic-strict", "generic-keep" (DESIGN.md section 3). */
const char* ndlqr_hip_schedule(const NdlqrHipCtx* ctx);

/* Per-kernel profile (NDLQR_FLAG_PROFILE): HIP-event durations accumulated since the last
 * reset, one slot per kernel kind. Returns number of slots / fills name, total ms, launches. */
int ndlqr_hip_profile_slots(NdlqrHipCtx* ctx);
int ndlqr_hip_profile_get(NdlqrHipCtx* ctx, int slot, char* name, int name_cap, double* total_ms,
                          int* launches);
int ndlqr_hip_profile_reset(NdlqrHipCtx* ctx);

/* Dense helpers behind Matrix* of ndlqr.h (src/linalg.c:55-190 -> src/linalg_custom.c).
 * Host pointers in, host pointers out; column-major. Return 0, or -1 for a failed Cholesky. */
int ndlqr_hip_gemm(int tA, int tB, int m, int n, int k, double alpha, const double* A, int lda,
                   const double* B, int ldb, double beta, double* C, int ldc);
int ndlqr_hip_potrf_lower(int n, double* A, int lda);
int ndlqr_hip_potrs_lower(int n, int nrhs, const double* L, int ldl, double* B, int ldb);
/* one triangular substitution alone: L x = b (transposed = 0) or L' x = b (clap_LowerTriBackSub,
 * src/linalg_custom.c:113-132) */
int ndlqr_hip_trsv_lower(int n, int nrhs, const double* L, int ldl, double* B, int ldb, int transposed);

#ifdef __cplusplus
}
#endif
#endif /* NDLQR_HIP_H_ */
