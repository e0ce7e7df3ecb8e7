/*
 * batch.c -- NdLqrBatchSolver: host side of the batched device solve (plain C).
 *
 * New relative to the reference (which solves one problem per ndlqr_Solve call and
 * parallelises inside it, src/solve.c:50-183): a batch axis of independent problems is the
 * unit of GPU parallelism. This file only packs caller data into the device input layout of
 * ndlqr_hip.h and drives the shim; all numerics are in ndlqr_hip.hip.
 *
 * Packing = the data movement of ndlqr_InitializeWithLQRProblem (src/solver.c:122-194) without
 * materialising the zero-padded `data` array: A_k, B_k go in row-major (which is exactly the
 * reference's C.state = A', C.input = B' stored column-major, src/solver.c:149-152), the rhs is
 * negated as in src/solver.c:188-190.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ndlqr.h"
#include "ndlqr_hip.h"

struct NdLqrBatchSolver {
  int n, m, N, K, batch, nvars, device;
  NdlqrHipCtx* ctx;
  /* host staging for `chunk` problems */
  int chunk;
  double* hAB;
  double* hQR;
  double* hrhs;
};

static size_t ab_doubles(const NdLqrBatchSolver* bs) { return (size_t)bs->N * bs->n * (bs->n + bs->m); }
static size_t qr_doubles(const NdLqrBatchSolver* bs) { return (size_t)bs->N * (bs->n + bs->m); }
static size_t rhs_doubles(const NdLqrBatchSolver* bs) { return (size_t)bs->N * (2 * bs->n + bs->m); }

NdLqrBatchSolver* ndlqr_NewBatchSolver(int nstates, int ninputs, int nhorizon, int batch,
                                       int device) {
  if (nstates <= 0 || ninputs <= 0 || batch <= 0) return NULL;
  if (nhorizon < 2 || !IsPowerOfTwo(nhorizon)) {
    fprintf(stderr, "ERROR: horizon must be a power of two >= 2, got %d.\n", nhorizon);
    return NULL;
  }
  NdlqrHipCtx* ctx = ndlqr_hip_create(nstates, ninputs, nhorizon, batch, device);
  if (!ctx) {
    fprintf(stderr, "ERROR: cannot create the HIP device context: %s\n", ndlqr_hip_last_error());
    return NULL;
  }
  NdLqrBatchSolver* bs = (NdLqrBatchSolver*)calloc(1, sizeof(*bs));
  if (!bs) { ndlqr_hip_destroy(ctx); return NULL; }
  bs->n = nstates; bs->m = ninputs; bs->N = nhorizon; bs->K = LogOfTwo(nhorizon);
  bs->batch = batch; bs->device = device; bs->ctx = ctx;
  bs->nvars = (2 * nstates + ninputs) * nhorizon - ninputs;
  /* stage at most ~64 MB of packed inputs at a time */
  size_t per_problem = sizeof(double) * (ab_doubles(bs) + qr_doubles(bs) + rhs_doubles(bs));
  size_t chunk = (64u << 20) / per_problem;
  if (chunk < 1) chunk = 1;
  if (chunk > (size_t)batch) chunk = (size_t)batch;
  bs->chunk = (int)chunk;
  bs->hAB = (double*)malloc(sizeof(double) * ab_doubles(bs) * chunk);
  bs->hQR = (double*)malloc(sizeof(double) * qr_doubles(bs) * chunk);
  bs->hrhs = (double*)malloc(sizeof(double) * rhs_doubles(bs) * chunk);
  if (!bs->hAB || !bs->hQR || !bs->hrhs) { ndlqr_FreeBatchSolver(bs); return NULL; }
  return bs;
}

int ndlqr_FreeBatchSolver(NdLqrBatchSolver* bs) {
  if (!bs) return NDLQR_ERR_INVALID;
  if (bs->ctx) ndlqr_hip_destroy(bs->ctx);
  free(bs->hAB); free(bs->hQR); free(bs->hrhs);
  free(bs);
  return NDLQR_OK;
}

int ndlqr_BatchSetFlags(NdLqrBatchSolver* bs, unsigned flags) {
  return bs ? ndlqr_hip_set_flags(bs->ctx, flags) : NDLQR_ERR_INVALID;
}
unsigned ndlqr_BatchGetFlags(const NdLqrBatchSolver* bs) { return bs ? ndlqr_hip_get_flags(bs->ctx) : 0u; }
int ndlqr_BatchNumVars(const NdLqrBatchSolver* bs) { return bs ? bs->nvars : NDLQR_ERR_INVALID; }
int ndlqr_BatchSize(const NdLqrBatchSolver* bs) { return bs ? bs->batch : NDLQR_ERR_INVALID; }
void* ndlqr_BatchDeviceContext(NdLqrBatchSolver* bs) { return bs ? (void*)bs->ctx : NULL; }

/* One knot into staging slot `slot`. Ak,Bk column-major; Qk,Rk diagonals. */
static void pack_knot(const NdLqrBatchSolver* bs, int slot, int k, const double* Ak,
                      const double* Bk, const double* Qk, const double* Rk) {
  const int n = bs->n, m = bs->m, w = n + m;
  double* AB = bs->hAB + (size_t)slot * ab_doubles(bs) + (size_t)k * n * w;
  double* QR = bs->hQR + (size_t)slot * qr_doubles(bs) + (size_t)k * w;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) AB[i * w + j] = Ak[i + n * j];
    for (int j = 0; j < m; ++j) AB[i * w + n + j] = Bk[i + n * j];
  }
  memcpy(QR, Qk, sizeof(double) * n);
  memcpy(QR + n, Rk, sizeof(double) * m);
}

/* rhs block of knot k: [-(x0 | d_{k-1}) ; -q_k ; -r_k], u-slot of the last knot stays 0
 * (src/solver.c:141-190). */
static void pack_rhs(const NdLqrBatchSolver* bs, int slot, int k, const double* lam_src,
                     const double* qk, const double* rk) {
  const int n = bs->n, m = bs->m;
  double* z = bs->hrhs + (size_t)slot * rhs_doubles(bs) + (size_t)k * (2 * n + m);
  for (int i = 0; i < n; ++i) z[i] = -lam_src[i];
  for (int i = 0; i < n; ++i) z[n + i] = -qk[i];
  for (int i = 0; i < m; ++i) z[2 * n + i] = (k < bs->N - 1) ? -rk[i] : 0.0;
}

static int flush(NdLqrBatchSolver* bs, int p0, int count) {
  return ndlqr_hip_upload_inputs(bs->ctx, p0, count, bs->hAB, bs->hQR, bs->hrhs);
}

int ndlqr_InitializeBatch(NdLqrBatchSolver* bs, const LQRProblem* const* probs, int count) {
  if (!bs || !probs || count != bs->batch) return NDLQR_ERR_INVALID;
  const int N = bs->N;
  int slot = 0, p0 = 0;
  for (int p = 0; p < count; ++p) {
    const LQRProblem* prob = probs[p];
    if (!prob || prob->nhorizon != N) return NDLQR_ERR_INVALID; /* src/solver.c:125 */
    for (int k = 0; k < N; ++k) {
      const LQRData* l = prob->lqrdata[k];
      if (l->nstates != bs->n || l->ninputs != bs->m) return NDLQR_ERR_INVALID; /* :142-143 */
      pack_knot(bs, slot, k, l->A, l->B, l->Q, l->R);
      pack_rhs(bs, slot, k, k == 0 ? prob->x0 : prob->lqrdata[k - 1]->d, l->q, l->r);
    }
    if (++slot == bs->chunk || p == count - 1) {
      int err = flush(bs, p0, slot);
      if (err) return err;
      p0 += slot;
      slot = 0;
    }
  }
  return NDLQR_OK;
}

int ndlqr_InitializeBatchFlat(NdLqrBatchSolver* bs, const double* A, const double* B,
                              const double* Q, const double* R, const double* q,
                              const double* r, const double* d, const double* x0) {
  if (!bs || !A || !B || !Q || !R || !q || !r || !d || !x0) return NDLQR_ERR_INVALID;
  const size_t n = (size_t)bs->n, m = (size_t)bs->m, N = (size_t)bs->N;
  int slot = 0, p0 = 0;
  for (int p = 0; p < bs->batch; ++p) {
    const double* Ap = A + p * N * n * n; const double* Bp = B + p * N * n * m;
    const double* Qp = Q + p * N * n; const double* Rp = R + p * N * m;
    const double* qp = q + p * N * n; const double* rp = r + p * N * m;
    const double* dp = d + p * N * n; const double* x0p = x0 + p * n;
    for (size_t k = 0; k < N; ++k) {
      pack_knot(bs, slot, (int)k, Ap + k * n * n, Bp + k * n * m, Qp + k * n, Rp + k * m);
      pack_rhs(bs, slot, (int)k, k == 0 ? x0p : dp + (k - 1) * n, qp + k * n, rp + k * m);
    }
    if (++slot == bs->chunk || p == bs->batch - 1) {
      int err = flush(bs, p0, slot);
      if (err) return err;
      p0 += slot;
      slot = 0;
    }
  }
  return NDLQR_OK;
}

int ndlqr_InitializeBatchFlatDevice(NdLqrBatchSolver* bs, const double* dA, const double* dB,
                                    const double* dQ, const double* dR, const double* dq,
                                    const double* dr, const double* dd, const double* dx0) {
  if (!bs) return NDLQR_ERR_INVALID;
  return ndlqr_hip_pack_flat_device(bs->ctx, dA, dB, dQ, dR, dq, dr, dd, dx0);
}

/* Synthetic problems of one staging chunk are generated and packed side by side on host threads
 * (one splitmix64 stream per problem, so the result does not depend on the thread count): the
 * (64,16,512) x 256 family is 37 s of Box-Muller draws on one core. */
typedef struct {
  NdLqrBatchSolver* bs;
  uint64_t seed0;
  int p0, count, tid, nthreads, err;
} SynthJob;

static void* synth_worker(void* arg) {
  SynthJob* job = (SynthJob*)arg;
  NdLqrBatchSolver* bs = job->bs;
  const size_t n = (size_t)bs->n, m = (size_t)bs->m, N = (size_t)bs->N;
  double* buf = (double*)malloc(sizeof(double) * (N * (n * n + n * m + 3 * n + 2 * m) + n));
  if (!buf) { job->err = NDLQR_ERR_INVALID; return NULL; }
  double* A = buf; double* B = A + N * n * n; double* Q = B + N * n * m; double* R = Q + N * n;
  double* q = R + N * m; double* r = q + N * n; double* d = r + N * m; double* x0 = d + N * n;
  for (int slot = job->tid; slot < job->count; slot += job->nthreads) {
    ndlqr_GenerateSyntheticFlat(bs->n, bs->m, bs->N, job->seed0 + (uint64_t)(job->p0 + slot), A, B, Q, R, q, r, d, x0);
    for (size_t k = 0; k < N; ++k) {
      pack_knot(bs, slot, (int)k, A + k * n * n, B + k * n * m, Q + k * n, R + k * m);
      pack_rhs(bs, slot, (int)k, k == 0 ? x0 : d + (k - 1) * n, q + k * n, r + k * m);
    }
  }
  free(buf);
  return NULL;
}

static int host_threads(void) {
  const char* env = getenv("NDLQR_HOST_THREADS");
  if (env && atoi(env) > 0) return atoi(env);
  cpu_set_t set;
  int t = 1;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) t = CPU_COUNT(&set);
  return t > 16 ? 16 : (t < 1 ? 1 : t);
}

int ndlqr_InitializeBatchSynthetic(NdLqrBatchSolver* bs, uint64_t seed0) {
  if (!bs) return NDLQR_ERR_INVALID;
  enum { MAXT = 64 };
  int nthreads = host_threads();
  if (nthreads > MAXT) nthreads = MAXT;
  for (int p0 = 0; p0 < bs->batch; p0 += bs->chunk) {
    const int count = bs->batch - p0 < bs->chunk ? bs->batch - p0 : bs->chunk;
    const int nt = count < nthreads ? count : nthreads;
    pthread_t th[MAXT];
    SynthJob job[MAXT];
    int started = 0;
    for (int t = 0; t < nt; ++t) {
      job[t] = (SynthJob){bs, seed0, p0, count, t, nt, NDLQR_OK};
      if (t > 0 && pthread_create(&th[t], NULL, synth_worker, &job[t]) == 0) ++started;
      else if (t > 0) { job[t].tid = -1; }
    }
    /* (a thread that could not be started: its share is done here, after this thread's own) */
    synth_worker(&job[0]);
    int err = job[0].err;
    for (int t = 1; t < nt; ++t) {
      if (job[t].tid < 0) { job[t].tid = t; synth_worker(&job[t]); }
      else pthread_join(th[t], NULL);
      if (job[t].err) err = job[t].err;
    }
    (void)started;
    if (!err) err = flush(bs, p0, count);
    if (err) return err;
  }
  return NDLQR_OK;
}

int ndlqr_BatchSetRhsFlat(NdLqrBatchSolver* bs, const double* q, const double* r, const double* d,
                          const double* x0) {
  if (!bs || !q || !r || !d || !x0) return NDLQR_ERR_INVALID;
  const size_t n = (size_t)bs->n, m = (size_t)bs->m, N = (size_t)bs->N;
  int slot = 0, p0 = 0;
  for (int p = 0; p < bs->batch; ++p) {
    const double* qp = q + p * N * n; const double* rp = r + p * N * m;
    const double* dp = d + p * N * n; const double* x0p = x0 + p * n;
    for (size_t k = 0; k < N; ++k)
      pack_rhs(bs, slot, (int)k, k == 0 ? x0p : dp + (k - 1) * n, qp + k * n, rp + k * m);
    if (++slot == bs->chunk || p == bs->batch - 1) {
      int err = ndlqr_hip_upload_rhs(bs->ctx, p0, slot, bs->hrhs);
      if (err) return err;
      p0 += slot;
      slot = 0;
    }
  }
  return NDLQR_OK;
}

int ndlqr_SolveBatchRhsOnly(NdLqrBatchSolver* bs) {
  if (!bs) return NDLQR_ERR_INVALID;
  int err = ndlqr_hip_solve_rhs_async(bs->ctx);
  if (err) return err;
  return ndlqr_hip_synchronize(bs->ctx);
}

int ndlqr_SolveBatchMultiRhsSlices(NdLqrBatchSolver* bs, int nrhs, const double* q, const double* r, const double* d,
                                   const double* x0, int knot0, int nknots, unsigned blocks, double* out) {
  return bs ? ndlqr_hip_solve_multi_rhs_slices(bs->ctx, nrhs, q, r, d, x0, knot0, nknots, blocks, out) : NDLQR_ERR_INVALID;
}
int ndlqr_SolveBatchMultiRhs(NdLqrBatchSolver* bs, int nrhs, const double* q, const double* r, const double* d,
                             const double* x0, double* soln) {
  return bs ? ndlqr_hip_solve_multi_rhs(bs->ctx, nrhs, q, r, d, x0, soln) : NDLQR_ERR_INVALID;
}

int ndlqr_SolveBatchAsync(NdLqrBatchSolver* bs) {
  return bs ? ndlqr_hip_solve_async(bs->ctx) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchSynchronize(NdLqrBatchSolver* bs) {
  return bs ? ndlqr_hip_synchronize(bs->ctx) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchStepAsync(NdLqrBatchSolver* bs, const double* q, const double* r, const double* d,
                         const double* x0, double* soln) {
  return bs ? ndlqr_hip_step_async(bs->ctx, q, r, d, x0, soln) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchSynchronizePrevious(NdLqrBatchSolver* bs) {
  return bs ? ndlqr_hip_synchronize_previous(bs->ctx) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchSetStepSelection(NdLqrBatchSolver* bs, int knot0, int nknots, unsigned blocks) {
  return bs ? ndlqr_hip_set_step_selection(bs->ctx, knot0, nknots, blocks) : NDLQR_ERR_INVALID;
}
int ndlqr_SolveBatchSlicesAsync(NdLqrBatchSolver* bs, int knot0, int nknots, unsigned blocks, double* out) {
  return bs ? ndlqr_hip_solve_slices_async(bs->ctx, knot0, nknots, blocks, out) : NDLQR_ERR_INVALID;
}
int ndlqr_CopyBatchSolutionSlices(NdLqrBatchSolver* bs, int knot0, int nknots, unsigned blocks, double* out) {
  return bs ? ndlqr_hip_download_selection(bs->ctx, knot0, nknots, blocks, out) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchTimeShardTopDoubles(NdLqrBatchSolver* bs, int G) {
  return bs ? ndlqr_hip_time_shard_top_doubles(bs->ctx, G) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchTimeShardFactor(NdLqrBatchSolver* bs, int g, int G) {
  return bs ? ndlqr_hip_time_shard_factor(bs->ctx, g, G) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchTimeShardExportTop(NdLqrBatchSolver* bs, int G, double* buf) {
  return bs ? ndlqr_hip_time_shard_export(bs->ctx, G, buf) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchTimeShardImportTop(NdLqrBatchSolver* bs, int G, const double* buf) {
  return bs ? ndlqr_hip_time_shard_import(bs->ctx, G, buf) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchTimeShardFinish(NdLqrBatchSolver* bs, int g, int G) {
  return bs ? ndlqr_hip_time_shard_finish(bs->ctx, g, G) : NDLQR_ERR_INVALID;
}
void* ndlqr_HostAlloc(size_t bytes) { return ndlqr_hip_host_alloc(bytes); }
void ndlqr_HostFree(void* p) { ndlqr_hip_host_free(p); }
void* ndlqr_DeviceAlloc(size_t bytes) { return ndlqr_hip_device_alloc(bytes); }
void ndlqr_DeviceFree(void* p) { ndlqr_hip_device_free(p); }
int ndlqr_DeviceCopy(void* dst, const void* src, size_t bytes) { return ndlqr_hip_copy(dst, src, bytes); }

int ndlqr_SolveBatch(NdLqrBatchSolver* bs) {
  if (!bs) return NDLQR_ERR_INVALID;
  int err = ndlqr_hip_solve_async(bs->ctx);
  if (err) return err;
  err = ndlqr_hip_synchronize(bs->ctx);
  if (err) return err;
  return ndlqr_hip_cholesky_failures(bs->ctx) > 0 ? NDLQR_ERR_NOT_SPD : NDLQR_OK;
}
double ndlqr_BatchSolveTimeMs(const NdLqrBatchSolver* bs) {
  return bs ? ndlqr_hip_last_solve_ms(bs->ctx) : -1.0;
}
int ndlqr_BatchCholeskyFailures(NdLqrBatchSolver* bs) {
  return bs ? ndlqr_hip_cholesky_failures(bs->ctx) : NDLQR_ERR_INVALID;
}
int ndlqr_BatchKktResiduals(NdLqrBatchSolver* bs, double* res, double* bnorm) {
  if (!bs || !res) return NDLQR_ERR_INVALID;
  return ndlqr_hip_kkt_residual(bs->ctx, res, bnorm);
}
int ndlqr_CopyBatchSolution(NdLqrBatchSolver* bs, int p, double* soln) {
  if (!bs || !soln || p < 0 || p >= bs->batch) return NDLQR_ERR_INVALID;
  int err = ndlqr_hip_download_solutions(bs->ctx, p, 1, soln);
  return err ? err : bs->nvars;
}
int ndlqr_CopyBatchSolutions(NdLqrBatchSolver* bs, double* soln) {
  if (!bs || !soln) return NDLQR_ERR_INVALID;
  int err = ndlqr_hip_download_solutions(bs->ctx, 0, bs->batch, soln);
  return err ? err : bs->nvars;
}
int ndlqr_CopyBatchSolutionsDevice(NdLqrBatchSolver* bs, double* dsoln) {
  if (!bs || !dsoln) return NDLQR_ERR_INVALID;
  int err = ndlqr_hip_pack_solutions_device(bs->ctx, dsoln);
  return err ? err : bs->nvars;
}
int ndlqr_CopyBatchFactors(NdLqrBatchSolver* bs, int p, double* fact) {
  if (!bs || !fact || p < 0 || p >= bs->batch) return NDLQR_ERR_INVALID;
  return ndlqr_hip_download_factors(bs->ctx, p, fact);
}
