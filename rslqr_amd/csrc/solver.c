/*
 * solver.c -- NdLqrSolver object and ndlqr_Solve (plain C host driver).
 *
 * Replaces, with the same caller-visible behaviour (paths under /root/reference/src):
 *   profile helpers                     solver.c:11-59
 *   ndlqr_NewNdLqrSolver / Free / Reset solver.c:61-120
 *   ndlqr_InitializeWithLQRProblem      solver.c:122-194   (host mirrors, same contents)
 *   summary / getters                   solver.c:196-226
 *   ndlqr_Solve / GetSolution / Copy    solve.c:38-201
 *
 * ndlqr_Solve here = pack host mirrors -> H2D -> device kernels (ndlqr_hip.h) -> D2H of the
 * solution. The device context is a batch-of-1 NdLqrBatchSolver created on first use, so the
 * container-only parts of the API (New / Initialize / Reset / getters) work on a machine with
 * no GPU, and ndlqr_Solve fails loudly there (NDLQR_ERR_NO_DEVICE). There is no CPU solve path.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ndlqr.h"
#include "ndlqr_hip.h"


/* ======================================================================= profile */

NdLqrProfile ndlqr_NewNdLqrProfile(void) {
  NdLqrProfile prof;
  memset(&prof, 0, sizeof(prof));
  prof.num_threads = -1;
  return prof;
}

void ndlqr_ResetProfile(NdLqrProfile* prof) {
  const int keep = prof->num_threads;
  memset(prof, 0, sizeof(*prof));
  prof->num_threads = keep;
}

void ndlqr_CopyProfile(NdLqrProfile* dest, NdLqrProfile* src) { *dest = *src; }

void ndlqr_PrintProfile(NdLqrProfile* profile) {
  printf("Solved with %d threads\n", profile->num_threads);
  printf("Solve Total:    %.3f ms\n", profile->t_total_ms);
  printf("Solve Leaves:   %.3f ms\n", profile->t_leaves_ms);
  printf("Solve Products: %.3f ms\n", profile->t_products_ms);
  printf("Solve Cholesky: %.3f ms\n", profile->t_cholesky_ms);
  printf("Solve Solve:    %.3f ms\n", profile->t_cholsolve_ms);
  printf("Solve Shur:     %.3f ms\n", profile->t_shur_ms);
}

static void print_ratio(const char* label, double base, double other) {
  printf("%s%.3f / %.3f (%.2f speedup)\n", label, base, other, base / other);
}

void ndlqr_CompareProfile(NdLqrProfile* base, NdLqrProfile* prof) {
  printf("Num Threads:     %d / %d\n", base->num_threads, prof->num_threads);
  print_ratio("Solve Total:     ", base->t_total_ms, prof->t_total_ms);
  print_ratio("Solve Leaves:    ", base->t_leaves_ms, prof->t_leaves_ms);
  print_ratio("Solve Products:  ", base->t_products_ms, prof->t_products_ms);
  print_ratio("Solve Cholesky:  ", base->t_cholesky_ms, prof->t_cholesky_ms);
  print_ratio("Solve CholSolve: ", base->t_cholsolve_ms, prof->t_cholsolve_ms);
  print_ratio("Solve Shur Comp: ", base->t_shur_ms, prof->t_shur_ms);
}

/* ======================================================================= construction */

NdLqrSolver* ndlqr_NewNdLqrSolver(int nstates, int ninputs, int nhorizon) {
  if (nstates <= 0 || ninputs <= 0 || nhorizon < 2 || !IsPowerOfTwo(nhorizon)) {
    fprintf(stderr, "ERROR: need nstates, ninputs > 0 and a power-of-two horizon >= 2.\n");
    return NULL;
  }
  NdLqrSolver* s = (NdLqrSolver*)calloc(1, sizeof(NdLqrSolver));
  if (!s) return NULL;
  const int n = nstates, m = ninputs, N = nhorizon;
  s->nstates = n;
  s->ninputs = m;
  s->nhorizon = N;
  s->tree = ndlqr_BuildTree(N);
  s->depth = s->tree.depth;
  s->nvars = (2 * n + m) * N - m;

  /* dense Q_k (n x n) and R_k (m x m) views over one slab, Q_0 first (solver.c:66-77) */
  const size_t per_knot = (size_t)n * n + (size_t)m * m;
  double* slab = (double*)calloc(per_knot * (size_t)N, sizeof(double));
  s->diagonals = (Matrix*)malloc(sizeof(Matrix) * 2 * (size_t)N);
  if (slab && s->diagonals) {
    for (int k = 0; k < N; ++k) {
      Matrix Qk = {n, n, slab + per_knot * (size_t)k};
      Matrix Rk = {m, m, Qk.data + (size_t)n * n};
      s->diagonals[2 * k] = Qk;
      s->diagonals[2 * k + 1] = Rk;
    }
  }
  s->data = ndlqr_NewNdData(n, m, N, n);
  s->fact = ndlqr_NewNdData(n, m, N, n);
  s->soln = ndlqr_NewNdData(n, m, N, 1);
  s->cholfacts = ndlqr_NewCholeskyFactors(s->depth, N);
  s->profile = ndlqr_NewNdLqrProfile();
  s->num_threads = 1;
  s->device_ctx = NULL;
  s->device_flags = 0u;
  {
    const char* mf = getenv("NDLQR_SOLVE_MIRRORS_FACT");
    s->mirror_fact = (mf && atoi(mf) != 0) ? 1 : 0;
  }
  s->device_profiling = -1;
  s->device_profiled = 0;
  s->device_split = ndlqr_NewNdLqrProfile();
  if (!slab || !s->diagonals || !s->data || !s->fact || !s->soln || !s->cholfacts ||
      !s->tree.node_list) {
    free(slab);
    if (s->diagonals) { free(s->diagonals); s->diagonals = NULL; }
    ndlqr_FreeNdLqrSolver(s);
    return NULL;
  }
  return s;
}

int ndlqr_FreeNdLqrSolver(NdLqrSolver* solver) {
  if (!solver) return -1;
  if (solver->device_ctx) ndlqr_FreeBatchSolver((NdLqrBatchSolver*)solver->device_ctx);
  ndlqr_FreeTree(&solver->tree);
  if (solver->data) ndlqr_FreeNdData(solver->data);
  if (solver->fact) ndlqr_FreeNdData(solver->fact);
  if (solver->soln) ndlqr_FreeNdData(solver->soln);
  if (solver->cholfacts) ndlqr_FreeCholeskyFactors(solver->cholfacts);
  if (solver->diagonals) {
    free(solver->diagonals[0].data); /* slab base, like solver.c:115 */
    free(solver->diagonals);
  }
  free(solver);
  return 0;
}

void ndlqr_ResetSolver(NdLqrSolver* solver) {
  ndlqr_ResetNdData(solver->data);
  ndlqr_ResetNdData(solver->fact);
  ndlqr_ResetNdData(solver->soln);
  ndlqr_ResetProfile(&solver->profile);
  for (int e = 0; e < 2 * solver->nhorizon; ++e) MatrixSetConst(&solver->diagonals[e], 0.0);
}

/* ======================================================================= KKT assembly */

static void set_diagonal(Matrix* M, const double* diag) {
  MatrixSetConst(M, 0.0);
  for (int i = 0; i < M->rows; ++i) M->data[i + M->rows * i] = diag[i];
}

int ndlqr_InitializeWithLQRProblem(const LQRProblem* lqrprob, NdLqrSolver* solver) {
  if (!lqrprob || !solver) return -1;
  const int n = solver->nstates, m = solver->ninputs, N = solver->nhorizon;
  if (lqrprob->nhorizon != N) return -1;

  NdFactor* zblk;
  ndlqr_GetNdFactor(solver->soln, 0, 0, &zblk);
  memcpy(zblk->lambda.data, lqrprob->x0, sizeof(double) * (size_t)n);

  for (int k = 0; k < N; ++k) {
    LQRData* knot = lqrprob->lqrdata[k];
    const bool last = (k == N - 1);
    if (!last && (knot->nstates != n || knot->ninputs != m)) return -1;
    ndlqr_GetNdFactor(solver->soln, k, 0, &zblk);
    memcpy(zblk->state.data, knot->q, sizeof(double) * (size_t)n);
    set_diagonal(&solver->diagonals[2 * k], knot->Q);
    if (last) break;

    memcpy(zblk->input.data, knot->r, sizeof(double) * (size_t)m);
    set_diagonal(&solver->diagonals[2 * k + 1], knot->R);

    /* coupling through separator k (tree level of node k): this knot contributes [A' ; B'],
     * the next one [-I ; 0]; the next knot's lambda slot of the rhs gets d_k. */
    const int lvl = ndlqr_GetIndexLevel(&solver->tree, k);
    NdFactor *Cthis, *Cnext, *znext;
    ndlqr_GetNdFactor(solver->data, k, lvl, &Cthis);
    ndlqr_GetNdFactor(solver->data, k + 1, lvl, &Cnext);
    ndlqr_GetNdFactor(solver->soln, k + 1, 0, &znext);
    Matrix A = ndlqr_GetA(knot), B = ndlqr_GetB(knot);
    MatrixCopyTranspose(&Cthis->state, &A);
    MatrixCopyTranspose(&Cthis->input, &B);
    MatrixSetConst(&Cnext->state, 0.0);
    for (int i = 0; i < n; ++i) Cnext->state.data[i + n * i] = -1.0;
    MatrixSetConst(&Cnext->input, 0.0);
    memcpy(znext->lambda.data, knot->d, sizeof(double) * (size_t)n);
  }

  for (int e = 0; e < solver->nvars; ++e) solver->soln->data[e] *= -1; /* solver.c:188-190 */
  return 0;
}

/* ======================================================================= reporting */

void ndlqr_PrintSolveSummary(NdLqrSolver* solver) {
  printf("rsLQR Solve Summary (MI355X / HIP backend)\n");
  printf("------------------------------------------\n");
  printf("  Device solve time:  %f ms\n", solver->solve_time_ms);
  printf("  Wall time incl. H2D/D2H: %f ms\n", solver->profile.t_total_ms);
  printf("  Host threads setting: %d (unused on the device).\n", solver->num_threads);
  printf("  ");
  MatrixPrintLinearAlgebraLibrary();
}

int ndlqr_GetNumVars(NdLqrSolver* solver) { return solver->nvars; }

int ndlqr_SetNumThreads(NdLqrSolver* solver, int num_threads) {
  if (!solver) return -1;
  solver->num_threads = num_threads;
  return 0;
}

int ndlqr_GetNumThreads(NdLqrSolver* solver) { return solver ? solver->num_threads : -1; }

int ndlqr_PrintSolveProfile(NdLqrSolver* solver) {
  if (!solver) return -1;
  ndlqr_PrintProfile(&solver->profile);
  return 0;
}

NdLqrProfile ndlqr_GetProfile(NdLqrSolver* solver) { return solver->profile; }

/* ======================================================================= solve */

static double wall_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static NdLqrBatchSolver* device_solver(NdLqrSolver* solver) {
  if (!solver->device_ctx) {
    solver->device_ctx =
        ndlqr_NewBatchSolver(solver->nstates, solver->ninputs, solver->nhorizon, 1, -1);
  }
  return (NdLqrBatchSolver*)solver->device_ctx;
}

/* [A | B] rows, diagonals and right-hand side of the host mirrors in the packed layout of ndlqr_hip.h (column j of the
 * column-major A' block is row j of A: the mirrors already are in device row order) */
static void pack_from_mirrors(const NdLqrSolver* solver, double* AB, double* QR, double* rhs) {
  const int n = solver->nstates, m = solver->ninputs, N = solver->nhorizon, w = n + m;
  const NdData* data = solver->data;
  for (int k = 0; k < N; ++k) {
    double* ab = AB + (size_t)k * n * w;
    double* qr = QR + (size_t)k * w;
    if (k < N - 1) {
      int lvl = 0;
      for (int t = k; t & 1; t >>= 1) ++lvl;
      const NdFactor* C = data->factors + (k + N * lvl);
      for (int i = 0; i < n; ++i) {
        memcpy(ab + i * w, C->state.data + (size_t)n * i, sizeof(double) * n);
        memcpy(ab + i * w + n, C->input.data + (size_t)m * i, sizeof(double) * m);
      }
    } else {
      memset(ab, 0, sizeof(double) * n * w);
    }
    for (int i = 0; i < n; ++i) qr[i] = solver->diagonals[2 * k].data[i + n * i];
    for (int i = 0; i < m; ++i) qr[n + i] = (k < N - 1) ? solver->diagonals[2 * k + 1].data[i + m * i] : 1.0;
  }
  memcpy(rhs, solver->soln->data, sizeof(double) * (size_t)N * (2 * n + m));
}

int ndlqr_Solve(NdLqrSolver* solver) {
  if (!solver) return NDLQR_ERR_INVALID;
  const double t0 = wall_ms();
  NdLqrBatchSolver* bs = device_solver(solver);
  if (!bs) {
    fprintf(stderr, "ndlqr_Solve: no HIP device context (%s); there is no CPU fallback.\n",
            ndlqr_hip_last_error());
    return NDLQR_ERR_NO_DEVICE;
  }
  NdlqrHipCtx* ctx = (NdlqrHipCtx*)ndlqr_BatchDeviceContext(bs);
  /* The same launch sequence as the batch API (default: fast mode, solution only). The
   * factorisation the reference leaves in solver->fact is produced on demand by
   * ndlqr_SyncFactorsToHost; ndlqr_SetDeviceFlags selects strict mode / KEEP_FACT up front.
   * Per-kernel events (eager launches) on the first solve of the solver, or on every one when asked for; otherwise
   * the whole call -- inputs up, launch chain, solution down -- is one captured graph (ndlqr_hip_solve_staged). */
  const int profiled = solver->device_profiling > 0 || (solver->device_profiling < 0 && !solver->device_profiled);
  unsigned want = solver->device_flags & ~NDLQR_FLAG_PROFILE;
  if (solver->mirror_fact) want |= NDLQR_FLAG_KEEP_FACT; /* the reference's ndlqr_Solve leaves fact behind (src/solve.c:120-131) */
  if (profiled) want |= NDLQR_FLAG_PROFILE;
  ndlqr_hip_set_flags(ctx, want);
  if (profiled) ndlqr_hip_profile_reset(ctx);
  double *hAB, *hQR, *hrhs, *hz;
  int err = ndlqr_hip_staged_io(ctx, &hAB, &hQR, &hrhs, &hz);
  if (err) return err;
  pack_from_mirrors(solver, hAB, hQR, hrhs);
  err = ndlqr_hip_solve_staged(ctx);
  if (err) return err;
  if (ndlqr_hip_cholesky_failures(ctx) > 0) err = NDLQR_ERR_NOT_SPD;
  /* full rhs blocks (N*(2n+m)) so the unused trailing u_N slot mirrors the device too */
  memcpy(solver->soln->data, hz, sizeof(double) * (size_t)solver->nhorizon * (2 * solver->nstates + solver->ninputs));
  if (solver->mirror_fact) {
    const int ferr = ndlqr_CopyBatchFactors(bs, 0, solver->fact->data);
    if (ferr && !err) err = ferr;
  }

  solver->solve_time_ms = ndlqr_BatchSolveTimeMs(bs);
  solver->linalg_time_ms = 0.0; /* the reference's global LA timer is compiled out by default too */
  ndlqr_ResetProfile(&solver->profile);
  /* The device kernels fuse the reference's phases (src/solve.c:15-25,76-116) differently; the
   * buckets are filled per kernel kind (see NdLqrProfile in ndlqr.h):
   *   t_leaves_ms    leaf kernel / bottom kernel (leaf phase fused with tree levels 0-1)
   *   t_products_ms  separator kernels (inner products + Cholesky + triangular solves in one kernel)
   *   t_shur_ms      Schur-update kernels and the solution sweep (apply / back-substitution)
   *   t_cholesky_ms, t_cholsolve_ms   always 0: never separate kernels on the device */
  if (profiled) {
    NdLqrProfile split = ndlqr_NewNdLqrProfile();
    const int slots = ndlqr_hip_profile_slots(ctx);
    for (int sl = 0; sl < slots; ++sl) {
      char name[64];
      double ms = 0;
      int launches = 0;
      if (ndlqr_hip_profile_get(ctx, sl, name, (int)sizeof(name), &ms, &launches) != 0) continue;
      if (strncmp(name, "leaf", 4) == 0 || strncmp(name, "bottom", 6) == 0) split.t_leaves_ms += ms;
      else if (strncmp(name, "separator", 9) == 0 || strncmp(name, "upper", 5) == 0 || strncmp(name, "top", 3) == 0)
        split.t_products_ms += ms;
      else split.t_shur_ms += ms;
    }
    solver->device_split = split;
    solver->device_profiled = 1;
  }
  if (solver->device_profiling != 0) { /* (the split of the last profiled solve; this call's own total) */
    solver->profile.t_leaves_ms = solver->device_split.t_leaves_ms;
    solver->profile.t_products_ms = solver->device_split.t_products_ms;
    solver->profile.t_shur_ms = solver->device_split.t_shur_ms;
  }
  solver->profile.t_total_ms = wall_ms() - t0;
  solver->profile.num_threads = solver->num_threads;
  return err;
}

Matrix ndlqr_GetSolution(NdLqrSolver* solver) {
  Matrix soln = {solver->nvars, 1, solver->soln->data};
  return soln;
}

int ndlqr_CopySolution(NdLqrSolver* solver, double* soln) {
  if (!solver) return -1;
  memcpy(soln, solver->soln->data, sizeof(double) * (size_t)solver->nvars);
  return solver->nvars;
}

int ndlqr_SetDeviceProfiling(NdLqrSolver* solver, int on) {
  if (!solver) return -1;
  solver->device_profiling = on ? 1 : 0;
  return 0;
}

int ndlqr_SetFactorMirroring(NdLqrSolver* solver, int on) {
  if (!solver) return -1;
  solver->mirror_fact = on ? 1 : 0;
  return 0;
}

int ndlqr_SetDeviceFlags(NdLqrSolver* solver, unsigned flags) {
  if (!solver) return -1;
  solver->device_flags = flags;
  return 0;
}

int ndlqr_SyncFactorsToHost(NdLqrSolver* solver) {
  if (!solver || !solver->device_ctx) return NDLQR_ERR_INVALID;
  NdLqrBatchSolver* bs = (NdLqrBatchSolver*)solver->device_ctx;
  NdlqrHipCtx* ctx = (NdlqrHipCtx*)ndlqr_BatchDeviceContext(bs);
  if (!ndlqr_hip_factors_valid(ctx)) {
    /* the last ndlqr_Solve ran without NDLQR_FLAG_KEEP_FACT: its inputs are still resident on the
     * device, so factor once more with the factor array materialised */
    const unsigned flags = ndlqr_hip_get_flags(ctx);
    ndlqr_hip_set_flags(ctx, (flags | NDLQR_FLAG_KEEP_FACT) & ~NDLQR_FLAG_PROFILE);
    int err = ndlqr_SolveBatch(bs);
    ndlqr_hip_set_flags(ctx, flags);
    if (err && err != NDLQR_ERR_NOT_SPD) return err;
  }
  return ndlqr_CopyBatchFactors(bs, 0, solver->fact->data);
}

const char* ndlqr_Version(void) { return "rslqr_amd 0.1 (gfx950)"; }
