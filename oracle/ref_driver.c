/*
 * ref_driver.c -- thin driver linked against the REAL reference sources.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/Makefile). This file contains no reference code: it
 * includes the reference's own headers from /root/reference/src at build time and calls
 * the reference's public functions. The resulting oracle/_ref/libref.so is used
 *   (1) to validate oracle/ndlqr_oracle.c (tests/test_oracle_vs_reference.py),
 *   (2) to generate the golden fixtures in tests/golden/ (tests/golden/make_golden.py),
 *   (3) optionally as bench.py's cpu_baseline ("kind": "reference").
 *
 * The reference's problem containers (lqr_data.c / lqr_problem.c / json_utils.c) need cJSON,
 * which is absent here, and are NOT part of the hot path; this driver fills the plain
 * LQRProblem / LQRData structs declared in the reference headers itself.
 */
#include <stdlib.h>
#include <string.h>

#include "ndlqr_ref_includes.h"

/* Build an LQRProblem from flat arrays (layout documented in ndlqr_oracle.c). */
static LQRProblem* make_problem(int n, int m, int N, const double* A, const double* B,
                                const double* Q, const double* R, const double* q,
                                const double* r, const double* d, const double* x0) {
  LQRProblem* p = (LQRProblem*)malloc(sizeof(LQRProblem));
  p->nhorizon = N;
  p->x0 = (double*)malloc(sizeof(double) * n);
  memcpy(p->x0, x0, sizeof(double) * n);
  p->lqrdata = (LQRData**)malloc(sizeof(LQRData*) * N);
  for (int k = 0; k < N; ++k) {
    LQRData* l = (LQRData*)malloc(sizeof(LQRData));
    int total = 2 * n + 2 * m + 1 + n * n + n * m + n;
    double* buf = (double*)calloc(total, sizeof(double));
    l->nstates = n; l->ninputs = m;
    l->Q = buf; l->R = l->Q + n; l->q = l->R + m; l->r = l->q + n; l->c = l->r + m;
    l->A = l->c + 1; l->B = l->A + n * n; l->d = l->B + n * m;
    memcpy(l->Q, Q + (size_t)k * n, sizeof(double) * n);
    memcpy(l->R, R + (size_t)k * m, sizeof(double) * m);
    memcpy(l->q, q + (size_t)k * n, sizeof(double) * n);
    memcpy(l->r, r + (size_t)k * m, sizeof(double) * m);
    memcpy(l->A, A + (size_t)k * n * n, sizeof(double) * n * n);
    memcpy(l->B, B + (size_t)k * n * m, sizeof(double) * n * m);
    memcpy(l->d, d + (size_t)k * n, sizeof(double) * n);
    p->lqrdata[k] = l;
  }
  return p;
}
static void free_problem(LQRProblem* p) {
  for (int k = 0; k < p->nhorizon; ++k) { free(p->lqrdata[k]->Q); free(p->lqrdata[k]); }
  free(p->lqrdata); free(p->x0); free(p);
}

/* (exported for ref_riccati_driver.c) */
LQRProblem* ref_make_problem(int n, int m, int N, const double* A, const double* B, const double* Q,
                             const double* R, const double* q, const double* r, const double* d,
                             const double* x0) {
  return make_problem(n, m, N, A, B, Q, R, q, r, d, x0);
}
void ref_free_problem(LQRProblem* p) { free_problem(p); }

/* New + Initialize; returns the reference's own NdLqrSolver*. */
void* ref_new_solver(int n, int m, int N, const double* A, const double* B, const double* Q,
                     const double* R, const double* q, const double* r, const double* d,
                     const double* x0) {
  NdLqrSolver* s = ndlqr_NewNdLqrSolver(n, m, N);
  if (!s) return NULL;
  LQRProblem* p = make_problem(n, m, N, A, B, Q, R, q, r, d, x0);
  int err = ndlqr_InitializeWithLQRProblem(p, s);
  free_problem(p);
  if (err) { ndlqr_FreeNdLqrSolver(s); return NULL; }
  return s;
}
int ref_reinit(void* sv, int n, int m, int N, const double* A, const double* B, const double* Q,
               const double* R, const double* q, const double* r, const double* d,
               const double* x0) {
  NdLqrSolver* s = (NdLqrSolver*)sv;
  ndlqr_ResetSolver(s);
  LQRProblem* p = make_problem(n, m, N, A, B, Q, R, q, r, d, x0);
  int err = ndlqr_InitializeWithLQRProblem(p, s);
  free_problem(p);
  return err;
}
void ref_free_solver(void* s) { ndlqr_FreeNdLqrSolver((NdLqrSolver*)s); }
int ref_solve(void* sv, int nthreads) {
  NdLqrSolver* s = (NdLqrSolver*)sv;
  ndlqr_SetNumThreads(s, nthreads);
  return ndlqr_Solve(s);
}
double ref_solve_time_ms(void* sv) { return ((NdLqrSolver*)sv)->solve_time_ms; }
double* ref_soln(void* sv) { return ((NdLqrSolver*)sv)->soln->data; }
double* ref_fact(void* sv) { return ((NdLqrSolver*)sv)->fact->data; }
double* ref_data(void* sv) { return ((NdLqrSolver*)sv)->data->data; }
void* ref_fact_nd(void* sv) { return ((NdLqrSolver*)sv)->fact; }
void* ref_data_nd(void* sv) { return ((NdLqrSolver*)sv)->data; }
void* ref_soln_nd(void* sv) { return ((NdLqrSolver*)sv)->soln; }
void* ref_tree(void* sv) { return &((NdLqrSolver*)sv)->tree; }
void* ref_cholfacts(void* sv) { return ((NdLqrSolver*)sv)->cholfacts; }
int ref_nvars(void* sv) { return ((NdLqrSolver*)sv)->nvars; }
int ref_depth(void* sv) { return ((NdLqrSolver*)sv)->depth; }
void ref_profile(void* sv, double* out7) {
  NdLqrProfile p = ((NdLqrSolver*)sv)->profile;
  out7[0] = p.t_total_ms; out7[1] = p.t_leaves_ms; out7[2] = p.t_products_ms;
  out7[3] = p.t_cholesky_ms; out7[4] = p.t_cholsolve_ms; out7[5] = p.t_shur_ms;
  out7[6] = p.num_threads;
}

/* Cholesky of the separator block exactly as src/solve.c:87-98 does it inline. */
int ref_factor_separator(void* sv, int leaf, int level) {
  NdLqrSolver* s = (NdLqrSolver*)sv;
  int index = ndlqr_GetIndexFromLeaf(&s->tree, leaf, level);
  NdFactor* F;
  ndlqr_GetNdFactor(s->fact, index + 1, level, &F);
  Matrix Sbar = F->lambda;
  CholeskyInfo* cholinfo;
  ndlqr_GetSFactorization(s->cholfacts, leaf, level, &cholinfo);
  return MatrixCholeskyFactorizeWithInfo(&Sbar, cholinfo);
}
int ref_solve_chol_factor(void* sv, int leaf, int level, int upper_level) {
  NdLqrSolver* s = (NdLqrSolver*)sv;
  int index = ndlqr_GetIndexFromLeaf(&s->tree, leaf, level);
  CholeskyInfo* cholinfo;
  ndlqr_GetSFactorization(s->cholfacts, leaf, level, &cholinfo);
  return ndlqr_SolveCholeskyFactor(s->fact, cholinfo, index, level, upper_level);
}

/*
 * Batch timing for bench.py cpu_baseline "reference" leg: `count` problems stored back to
 * back; each is Reset+Initialize'd (untimed) then ndlqr_Solve'd with `nthreads`
 * (timed by the solver's own omp_get_wtime bracket, src/solve.c:40,184).
 * Returns summed solve ms.
 */
double ref_bench(int n, int m, int N, int count, int reps, const double* A, const double* B,
                 const double* Q, const double* R, const double* q, const double* r,
                 const double* d, const double* x0, int nthreads) {
  size_t sA = (size_t)N * n * n, sB = (size_t)N * n * m, sn = (size_t)N * n, sm = (size_t)N * m;
  NdLqrSolver* s = ndlqr_NewNdLqrSolver(n, m, N);
  if (!s) return -1.0;
  double total = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (int p = 0; p < count; ++p) {
      ref_reinit(s, n, m, N, A + p * sA, B + p * sB, Q + p * sn, R + p * sm, q + p * sn,
                 r + p * sm, d + p * sn, x0 + (size_t)p * n);
      ndlqr_SetNumThreads(s, nthreads);
      ndlqr_Solve(s);
      total += s->solve_time_ms;
    }
  }
  ndlqr_FreeNdLqrSolver(s);
  return total;
}

/*
 * Throughput mode for bench.py: `nthreads` host threads, each with its own reference solver,
 * each running whole solves with one thread inside (the nested region of ndlqr_Solve gets a
 * team of 1). Returns wall ms of the whole loop (reset + initialise included; they are
 * O(N n^2) against the O(N K^2 n^3) solve).
 */
#include <omp.h>
double ref_bench_throughput(int n, int m, int N, int count, int reps, const double* A,
                            const double* B, const double* Q, const double* R, const double* q,
                            const double* r, const double* d, const double* x0, int nthreads) {
  size_t sA = (size_t)N * n * n, sB = (size_t)N * n * m, sn = (size_t)N * n, sm = (size_t)N * m;
  NdLqrSolver** ss = (NdLqrSolver**)malloc(sizeof(NdLqrSolver*) * nthreads);
  for (int t = 0; t < nthreads; ++t) ss[t] = ndlqr_NewNdLqrSolver(n, m, N);
  double t0 = omp_get_wtime();
  for (int rep = 0; rep < reps; ++rep) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int p = 0; p < count; ++p) {
      NdLqrSolver* s = ss[omp_get_thread_num()];
      ref_reinit(s, n, m, N, A + p * sA, B + p * sB, Q + p * sn, R + p * sm, q + p * sn,
                 r + p * sm, d + p * sn, x0 + (size_t)p * n);
      ndlqr_SetNumThreads(s, 1);
      ndlqr_Solve(s);
    }
  }
  double ms = (omp_get_wtime() - t0) * 1e3;
  for (int t = 0; t < nthreads; ++t) ndlqr_FreeNdLqrSolver(ss[t]);
  free(ss);
  return ms;
}
