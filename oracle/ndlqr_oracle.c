/*
 * ndlqr_oracle.c -- CPU ORACLE for the rsLQR nested-dissection LQR solve.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The shipped library (rslqr_amd/csrc) never links, includes or calls anything here.
 *
 * It restates, in plain C99 with flat column-major arrays, the algorithm of the
 * reference bjack205/rsLQR (paths relative to /root/reference):
 *   - KKT assembly            src/solver.c:122-194
 *   - leaf solve              src/nested_dissection.c:10-105
 *   - separator inner product src/nested_dissection.c:114-134
 *   - separator Cholesky      src/solve.c:87-98  -> src/linalg_custom.c:88-111
 *   - Cholesky solve          src/nested_dissection.c:136-152 -> src/linalg_custom.c:113-138
 *   - Schur update            src/nested_dissection.c:154-177
 *   - level schedule          src/solve.c:38-190
 *   - storage layout          src/nddata.c:15-65,82-96
 * Floating-point operations are issued in the same order as the reference's
 * default "internal routines" backend (ijk GEMM with beta applied first, left-looking
 * Cholesky, column-by-column substitution), so results agree with the reference to
 * rounding (compilers may contract a*b+c differently).
 *
 * PARITY PIN: checked against (a) the reference's golden vectors lqr_prob.json["soln"],
 * lqr_prob_256.json["soln"], sample_problem.json{b, E01*, E11*, E02*, E12*, soln} and the
 * literals of test/nested_dissection_test.c, and (b) the reference itself compiled
 * from /root/reference/src into oracle/_ref/libref.so (see oracle/Makefile) --
 * tests/test_oracle_vs_reference.py and tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int n, m, N, K;     /* nstates, ninputs, nhorizon, depth=log2(N) */
  int fb;             /* doubles per matrix factor block: (2n+m)*n */
  int zb;             /* doubles per rhs block: 2n+m */
  double* data;       /* "C" blocks,  N*K*fb  (src/solver.c:87)  */
  double* fact;       /* "F" blocks,  N*K*fb  (src/solver.c:88)  */
  double* soln;       /* rhs/solution, N*zb   (src/solver.c:89)  */
  double* diag;       /* per knot: dense Q (n*n) then dense R (m*m) (src/solver.c:66-77) */
  int* chol_info;     /* 0 ok / -1 failed, 2N + (N-1) slots (src/cholesky_factors.c:6-36) */
} OracleSolver;

/* ---- index helpers ---------------------------------------------------------------- */

/* block (k, level) of a width-n NdData: src/nddata.c:82-96 */
static double* blk(const OracleSolver* s, double* base, int k, int level) {
  return base + ((size_t)k + (size_t)s->N * level) * s->fb;
}
static double* zblk(const OracleSolver* s, int k) { return s->soln + (size_t)k * s->zb; }
static double* diagQ(const OracleSolver* s, int k) {
  return s->diag + (size_t)k * (s->n * s->n + s->m * s->m);
}
static double* diagR(const OracleSolver* s, int k) { return diagQ(s, k) + s->n * s->n; }

/* level of tree node k = number of trailing one bits (src/binary_tree.c:9-37) */
int oracle_index_level(int k) {
  int l = 0;
  while (k & 1) { ++l; k >>= 1; }
  return l;
}
/* src/binary_tree.c:65-69 */
int oracle_index_from_leaf(int leaf, int level) { return (1 << level) * (2 * leaf + 1) - 1; }
/* src/binary_tree.c:89-106 (k = N-1 folds into the node of k-1) */
int oracle_index_at_level(int k, int level) {
  int base = (k >> (level + 1)) << (level + 1);
  return base + (1 << level) - 1;
}
/* src/nested_dissection.c:173-177 */
int oracle_should_calc_lambda(int sep_index, int level, int i) {
  int left_start = sep_index - ((1 << level) - 1);
  int right_start = sep_index + 1;
  int is_start = (i == left_start) || (i == right_start);
  return !is_start || i == 0;
}

/* ---- dense helpers: src/linalg_custom.c ------------------------------------------- */

/* C = alpha*op(A)*op(B) + beta*C ; column-major; src/linalg_custom.c:20-43 */
static void gemm(const double* A, int Ar, int Ac, const double* B, int Br, int Bc, double* C,
                 int Cr, int tA, int tB, double alpha, double beta) {
  int n = tA ? Ac : Ar;
  int m = tA ? Ar : Ac;
  int p = tB ? Br : Bc;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < p; ++j) {
      double* Cij = C + i + (size_t)Cr * j;
      *Cij *= beta;
      for (int k = 0; k < m; ++k) {
        double Aik = tA ? A[k + (size_t)Ar * i] : A[i + (size_t)Ar * k];
        double Bkj = tB ? B[j + (size_t)Br * k] : B[k + (size_t)Br * j];
        *Cij += alpha * Aik * Bkj;
      }
    }
  }
}

/* in-place lower Cholesky; src/linalg_custom.c:88-111 */
static int chol(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    for (int k = 0; k < j; ++k) {
      for (int i = j; i < n; ++i) {
        A[i + (size_t)n * j] -= A[i + (size_t)n * k] * A[j + (size_t)n * k];
      }
    }
    double Ajj = A[j + (size_t)n * j];
    if (Ajj <= 0) return -1;
    double ajj = sqrt(Ajj);
    for (int i = j; i < n; ++i) A[i + (size_t)n * j] /= ajj;
  }
  return 0;
}

/* src/linalg_custom.c:113-132 */
static void tri_sub(const double* L, int n, double* b, int cols, int transposed) {
  for (int j_ = 0; j_ < n; ++j_) {
    int j = transposed ? n - j_ - 1 : j_;
    for (int k = 0; k < cols; ++k) {
      double* xjk = b + j + (size_t)n * k;
      *xjk /= L[j + (size_t)n * j];
      for (int i_ = j_ + 1; i_ < n; ++i_) {
        int i = transposed ? i_ - (j_ + 1) : i_;
        double Lij = transposed ? L[j + (size_t)n * i] : L[i + (size_t)n * j];
        b[i + (size_t)n * k] -= Lij * (*xjk);
      }
    }
  }
}
/* src/linalg_custom.c:134-138 */
static void chol_solve(const double* L, int n, double* b, int cols) {
  tri_sub(L, n, b, cols, 0);
  tri_sub(L, n, b, cols, 1);
}

/* ---- solver object ----------------------------------------------------------------- */

static int is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

OracleSolver* oracle_new(int n, int m, int N) {
  if (n <= 0 || m <= 0 || N < 2 || !is_pow2(N)) return NULL; /* src/nddata.c:16-21 */
  OracleSolver* s = (OracleSolver*)calloc(1, sizeof(*s));
  s->n = n; s->m = m; s->N = N;
  s->K = 0; while ((1 << s->K) < N) ++s->K;
  s->fb = (2 * n + m) * n;
  s->zb = 2 * n + m;
  s->data = (double*)calloc((size_t)N * s->K * s->fb, sizeof(double));
  s->fact = (double*)calloc((size_t)N * s->K * s->fb, sizeof(double));
  s->soln = (double*)calloc((size_t)N * s->zb, sizeof(double));
  s->diag = (double*)calloc((size_t)N * (n * n + m * m), sizeof(double));
  s->chol_info = (int*)calloc((size_t)3 * N, sizeof(int));
  return s;
}
void oracle_free(OracleSolver* s) {
  if (!s) return;
  free(s->data); free(s->fact); free(s->soln); free(s->diag); free(s->chol_info); free(s);
}
/* src/solver.c:98-106 */
void oracle_reset(OracleSolver* s) {
  memset(s->data, 0, sizeof(double) * (size_t)s->N * s->K * s->fb);
  memset(s->fact, 0, sizeof(double) * (size_t)s->N * s->K * s->fb);
  memset(s->soln, 0, sizeof(double) * (size_t)s->N * s->zb);
  memset(s->diag, 0, sizeof(double) * (size_t)s->N * (s->n * s->n + s->m * s->m));
}
double* oracle_data(OracleSolver* s) { return s->data; }
double* oracle_fact(OracleSolver* s) { return s->fact; }
double* oracle_soln(OracleSolver* s) { return s->soln; }
double* oracle_diag(OracleSolver* s) { return s->diag; }
int oracle_nvars(const OracleSolver* s) { return s->zb * s->N - s->m; }
int oracle_depth(const OracleSolver* s) { return s->K; }
int oracle_chol_failures(const OracleSolver* s) {
  int c = 0;
  for (int i = 0; i < 3 * s->N; ++i) c += s->chol_info[i] != 0;
  return c;
}

/*
 * KKT assembly, src/solver.c:122-194.
 * Flat inputs, per knot k (k = 0..N-1): A[k] n*n col-major, B[k] n*m col-major,
 * Q[k] n (diagonal), R[k] m (diagonal), q[k] n, r[k] m, d[k] n; x0 n.
 * (The reference's LQRData always carries all fields for every knot, including the
 * unused A,B,R,r,d of the last one.)
 */
int oracle_initialize(OracleSolver* s, const double* A, const double* B, const double* Q,
                      const double* R, const double* q, const double* r, const double* d,
                      const double* x0) {
  const int n = s->n, m = s->m, N = s->N;
  memcpy(zblk(s, 0), x0, sizeof(double) * n); /* soln(0).lambda = x0 */
  int k;
  for (k = 0; k < N - 1; ++k) {
    int level = oracle_index_level(k);
    double* C = blk(s, s->data, k, level);
    double* Cx = C + n * n;      /* state block  n x n */
    double* Cu = C + 2 * n * n;  /* input block  m x n */
    const double* Ak = A + (size_t)k * n * n;
    const double* Bk = B + (size_t)k * n * m;
    for (int i = 0; i < n; ++i)     /* Cx = A' */
      for (int j = 0; j < n; ++j) Cx[i + n * j] = Ak[j + n * i];
    for (int i = 0; i < m; ++i)     /* Cu = B' */
      for (int j = 0; j < n; ++j) Cu[i + m * j] = Bk[j + n * i];
    double* z = zblk(s, k);
    memcpy(z + n, q + (size_t)k * n, sizeof(double) * n);
    memcpy(z + 2 * n, r + (size_t)k * m, sizeof(double) * m);
    double* Qd = diagQ(s, k);
    double* Rd = diagR(s, k);
    memset(Qd, 0, sizeof(double) * n * n);
    memset(Rd, 0, sizeof(double) * m * m);
    for (int i = 0; i < n; ++i) Qd[i + n * i] = Q[(size_t)k * n + i];
    for (int i = 0; i < m; ++i) Rd[i + m * i] = R[(size_t)k * m + i];
    /* next time step */
    double* C2 = blk(s, s->data, k + 1, level);
    double* C2x = C2 + n * n;
    double* C2u = C2 + 2 * n * n;
    memset(C2x, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) C2x[i + n * i] = -1.0;
    memset(C2u, 0, sizeof(double) * m * n);
    memcpy(zblk(s, k + 1), d + (size_t)k * n, sizeof(double) * n);
  }
  /* terminal step */
  memcpy(zblk(s, k) + n, q + (size_t)k * n, sizeof(double) * n);
  double* Qd = diagQ(s, k);
  memset(Qd, 0, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i) Qd[i + n * i] = Q[(size_t)k * n + i];
  /* negate rhs */
  int nvars = oracle_nvars(s);
  for (int i = 0; i < nvars; ++i) s->soln[i] *= -1;
  return 0;
}

/* src/nested_dissection.c:10-105 */
int oracle_solve_leaf(OracleSolver* s, int k) {
  const int n = s->n, m = s->m, N = s->N;
  double* Qd = diagQ(s, k);
  double* Rd = diagR(s, k);
  double* z = zblk(s, k);
  if (k == 0) {
    double* C = blk(s, s->data, 0, 0);
    double* F = blk(s, s->fact, 0, 0);
    double *Fy = F, *Fx = F + n * n, *Fu = F + 2 * n * n;
    const double *Cx = C + n * n, *Cu = C + 2 * n * n;
    for (int i = 0; i < n * n; ++i) Fy[i] = Cx[i];
    for (int i = 0; i < n * n; ++i) Fy[i] *= -1.0;
    for (int i = 0; i < n * n; ++i) Fx[i] = 0.0;
    memcpy(Fu, Cu, sizeof(double) * m * n);
    s->chol_info[1] = chol(Rd, m);
    chol_solve(Rd, m, Fu, n);
    chol_solve(Rd, m, z + 2 * n, 1);
    /* rhs: zy = -Q zy - zx ; zx = -zy_old  (uses data(0,0).lambda as scratch) */
    double* tmp = C; /* C->lambda column 0 */
    memcpy(tmp, z, sizeof(double) * n);
    memcpy(z, z + n, sizeof(double) * n);
    gemm(Qd, n, n, tmp, n, 1, z, n, 0, 0, -1.0, -1.0);
    memcpy(z + n, tmp, sizeof(double) * n);
    for (int i = 0; i < n; ++i) z[n + i] *= -1.0;
    s->chol_info[0] = chol(Qd, n);
  } else {
    int level = 0;
    s->chol_info[2 * k] = chol(Qd, n);
    if (k < N - 1) {
      level = oracle_index_level(k);
      double* C = blk(s, s->data, k, level);
      double* F = blk(s, s->fact, k, level);
      s->chol_info[2 * k + 1] = chol(Rd, m);
      chol_solve(Rd, m, z + 2 * n, 1);
      memcpy(F + n * n, C + n * n, sizeof(double) * n * n);
      chol_solve(Qd, n, F + n * n, n);
      memcpy(F + 2 * n * n, C + 2 * n * n, sizeof(double) * m * n);
      chol_solve(Rd, m, F + 2 * n * n, n);
    }
    chol_solve(Qd, n, z + n, 1);
    int prev_level = oracle_index_level(k - 1);
    double* C = blk(s, s->data, k, prev_level);
    double* F = blk(s, s->fact, k, prev_level);
    memcpy(F + n * n, C + n * n, sizeof(double) * n * n);
    chol_solve(Qd, n, F + n * n, n);
    for (int i = 0; i < m * n; ++i) F[2 * n * n + i] = 0.0;
  }
  return 0;
}

/* which: 0 = matrix factors (width n), 1 = rhs (width 1). src/nested_dissection.c:114-134 */
int oracle_inner_product(OracleSolver* s, int which, int index, int data_level, int fact_level) {
  const int n = s->n, m = s->m;
  const double* C1 = blk(s, s->data, index, data_level);
  const double* C2 = blk(s, s->data, index + 1, data_level);
  double *F1, *F2;
  int w;
  if (which == 0) {
    F1 = blk(s, s->fact, index, fact_level);
    F2 = blk(s, s->fact, index + 1, fact_level);
    w = n;
  } else {
    F1 = zblk(s, index);
    F2 = zblk(s, index + 1);
    w = 1;
  }
  double* S = F2; /* lambda block of F2 */
  gemm(C1 + n * n, n, n, F1 + n * w, n, w, S, n, 1, 0, 1.0, -1.0);
  gemm(C1 + 2 * n * n, m, n, F1 + 2 * n * w, m, w, S, n, 1, 0, 1.0, 1.0);
  gemm(C2 + n * n, n, n, F2 + n * w, n, w, S, n, 1, 0, 1.0, 1.0);
  gemm(C2 + 2 * n * n, m, n, F2 + 2 * n * w, m, w, S, n, 1, 0, 1.0, 1.0);
  return 0;
}

static int s_slot(const OracleSolver* s, int leaf, int level) { /* src/cholesky_factors.c:65-82 */
  int idx = 2 * s->N;
  for (int l = 0; l < level; ++l) idx += 1 << (s->K - l - 1);
  return idx + leaf;
}

/* src/solve.c:87-98 */
int oracle_factor_separator(OracleSolver* s, int leaf, int level) {
  int index = oracle_index_from_leaf(leaf, level);
  double* Sbar = blk(s, s->fact, index + 1, level);
  int info = chol(Sbar, s->n);
  s->chol_info[s_slot(s, leaf, level)] = info;
  return info;
}

/* src/nested_dissection.c:136-152 */
int oracle_solve_chol_factor(OracleSolver* s, int index, int level, int upper_level) {
  const double* Sbar = blk(s, s->fact, index + 1, level);
  double* f = blk(s, s->fact, index + 1, upper_level);
  chol_solve(Sbar, s->n, f, s->n);
  return 0;
}

/* src/solve.c:152-170 */
int oracle_solve_chol_rhs(OracleSolver* s, int index, int level) {
  const double* Sbar = blk(s, s->fact, index + 1, level);
  chol_solve(Sbar, s->n, zblk(s, index + 1), 1);
  return 0;
}

/* which: 0 -> g in fact column upper_level, 1 -> g in soln. src/nested_dissection.c:154-171 */
int oracle_update_schur(OracleSolver* s, int which, int index, int i, int level, int upper_level,
                        int calc_lambda) {
  const int n = s->n, m = s->m;
  const double* F = blk(s, s->fact, i, level);
  const double* f;
  double* g;
  int w;
  if (which == 0) {
    f = blk(s, s->fact, index + 1, upper_level);
    g = blk(s, s->fact, i, upper_level);
    w = n;
  } else {
    f = zblk(s, index + 1);
    g = zblk(s, i);
    w = 1;
  }
  if (calc_lambda) gemm(F, n, n, f, n, w, g, n, 0, 0, -1.0, 1.0);
  gemm(F + n * n, n, n, f, n, w, g + n * w, n, 0, 0, -1.0, 1.0);
  gemm(F + 2 * n * n, m, n, f, n, w, g + 2 * n * w, m, 0, 0, -1.0, 1.0);
  return 0;
}

/* src/nested_dissection.c:179-192 */
int oracle_compute_schur_compliment(OracleSolver* s, int index, int level, int upper_level) {
  int left_start = index - ((1 << level) - 1);
  int right_stop = index + (1 << level);
  for (int i = left_start; i <= right_stop; ++i) {
    int cl = oracle_should_calc_lambda(index, level, i);
    oracle_update_schur(s, upper_level == 0 ? 1 : 0, index, i, level, upper_level, cl);
  }
  return 0;
}

/* static contiguous chunking, src/solve.c:27-36 */
static void get_work(int total, int nthreads, int tid, int* start, int* stop) {
  int per = total / nthreads;
  *start = per * tid;
  *stop = (tid == nthreads - 1) ? total : per * (tid + 1);
}

/*
 * Level-synchronous schedule, src/solve.c:38-190. Same phases and barriers.
 * stop_after_level >= 0 stops the factor phase after that level (test hook); -1 = full solve.
 */
int oracle_solve_ex(OracleSolver* s, int nthreads, int stop_after_level, int do_solve_phase) {
  const int K = s->K, N = s->N;
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
#ifdef _OPENMP
    int nt = omp_get_num_threads();
    int tid = omp_get_thread_num();
#else
    int nt = 1, tid = 0;
#endif
    int a, b;
    get_work(N, nt, tid, &a, &b);
    for (int k = a; k < b; ++k) oracle_solve_leaf(s, k);
#ifdef _OPENMP
#pragma omp barrier
#endif
    for (int level = 0; level < K; ++level) {
      if (stop_after_level >= 0 && level > stop_after_level) break;
      int numleaves = 1 << (K - level - 1);
      int cur_depth = K - level;
      get_work(numleaves * cur_depth, nt, tid, &a, &b);
      for (int i = a; i < b; ++i) {
        int leaf = i / cur_depth;
        int upper = level + (i % cur_depth);
        oracle_inner_product(s, 0, oracle_index_from_leaf(leaf, level), level, upper);
      }
#ifdef _OPENMP
#pragma omp barrier
#endif
      get_work(numleaves, nt, tid, &a, &b);
      for (int leaf = a; leaf < b; ++leaf) oracle_factor_separator(s, leaf, level);
#ifdef _OPENMP
#pragma omp barrier
#endif
      int upper_levels = cur_depth - 1;
      get_work(numleaves * upper_levels, nt, tid, &a, &b);
      for (int i = a; i < b; ++i) {
        int leaf = i / upper_levels;
        int upper = level + 1 + (i % upper_levels);
        oracle_solve_chol_factor(s, oracle_index_from_leaf(leaf, level), level, upper);
      }
#ifdef _OPENMP
#pragma omp barrier
#endif
      get_work(N * upper_levels, nt, tid, &a, &b);
      for (int i = a; i < b; ++i) {
        int k = i / upper_levels;
        int upper = level + 1 + (i % upper_levels);
        int index = oracle_index_at_level(k, level);
        int cl = oracle_should_calc_lambda(index, level, k);
        oracle_update_schur(s, 0, index, k, level, upper, cl);
      }
#ifdef _OPENMP
#pragma omp barrier
#endif
    }
    if (do_solve_phase) {
      for (int level = 0; level < K; ++level) {
        int numleaves = 1 << (K - level - 1);
        get_work(numleaves, nt, tid, &a, &b);
        for (int leaf = a; leaf < b; ++leaf)
          oracle_inner_product(s, 1, oracle_index_from_leaf(leaf, level), level, 0);
#ifdef _OPENMP
#pragma omp barrier
#endif
        get_work(numleaves, nt, tid, &a, &b);
        for (int leaf = a; leaf < b; ++leaf)
          oracle_solve_chol_rhs(s, oracle_index_from_leaf(leaf, level), level);
#ifdef _OPENMP
#pragma omp barrier
#endif
        get_work(N, nt, tid, &a, &b);
        for (int k = a; k < b; ++k) {
          int index = oracle_index_at_level(k, level);
          int cl = oracle_should_calc_lambda(index, level, k);
          oracle_update_schur(s, 1, index, k, level, 0, cl);
        }
#ifdef _OPENMP
#pragma omp barrier
#endif
      }
    }
  }
  return 0;
}

int oracle_solve(OracleSolver* s, int nthreads) { return oracle_solve_ex(s, nthreads, -1, 1); }

static double now_ms(void) {
#ifdef _OPENMP
  return omp_get_wtime() * 1e3;
#else
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
#endif
}

/*
 * One-shot convenience: new + initialize + solve + copy out.
 * soln_out: N*(2n+m) doubles (first nvars are the solution), fact_out (may be NULL): N*K*fb.
 * Returns solve wall time in ms (solve only, like src/solve.c:40,184) through *ms.
 */
int oracle_solve_flat(int n, int m, int N, const double* A, const double* B, const double* Q,
                      const double* R, const double* q, const double* r, const double* d,
                      const double* x0, int nthreads, double* soln_out, double* fact_out,
                      double* ms) {
  OracleSolver* s = oracle_new(n, m, N);
  if (!s) return -1;
  oracle_initialize(s, A, B, Q, R, q, r, d, x0);
  double t0 = now_ms();
  oracle_solve(s, nthreads);
  double t1 = now_ms();
  if (ms) *ms = t1 - t0;
  if (soln_out) memcpy(soln_out, s->soln, sizeof(double) * (size_t)N * s->zb);
  if (fact_out) memcpy(fact_out, s->fact, sizeof(double) * (size_t)N * s->K * s->fb);
  int fails = oracle_chol_failures(s);
  oracle_free(s);
  return fails;
}

/*
 * CPU-baseline timing helper (bench.py cpu_baseline "port" leg).
 * Problems are stored back to back in the flat arrays (count of them).
 *   mode 0: "reference semantics" -- one solve at a time, nthreads inside the solve
 *   mode 1: "throughput"          -- omp parallel for over problems, 1 thread per solve
 * Re-initialises before every solve (excluded from the timed sum in mode 0).
 * Returns total solve-only milliseconds (mode 0) or wall ms of the whole loop incl.
 * initialise (mode 1) through *ms.
 */
int oracle_bench(int n, int m, int N, int count, int reps, const double* A, const double* B,
                 const double* Q, const double* R, const double* q, const double* r,
                 const double* d, const double* x0, int nthreads, int mode, double* ms) {
  size_t sA = (size_t)N * n * n, sB = (size_t)N * n * m, sn = (size_t)N * n, sm = (size_t)N * m;
  double total = 0;
  if (mode == 0) {
    OracleSolver* s = oracle_new(n, m, N);
    if (!s) return -1;
    for (int rep = 0; rep < reps; ++rep) {
      for (int p = 0; p < count; ++p) {
        oracle_reset(s);
        oracle_initialize(s, A + p * sA, B + p * sB, Q + p * sn, R + p * sm, q + p * sn,
                          r + p * sm, d + p * sn, x0 + (size_t)p * n);
        double t0 = now_ms();
        oracle_solve(s, nthreads);
        total += now_ms() - t0;
      }
    }
    oracle_free(s);
  } else {
    OracleSolver** ss = (OracleSolver**)malloc(sizeof(OracleSolver*) * nthreads);
    for (int t = 0; t < nthreads; ++t) ss[t] = oracle_new(n, m, N);
    double t0 = now_ms();
    for (int rep = 0; rep < reps; ++rep) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
      for (int p = 0; p < count; ++p) {
#ifdef _OPENMP
        OracleSolver* s = ss[omp_get_thread_num()];
#else
        OracleSolver* s = ss[0];
#endif
        oracle_reset(s);
        oracle_initialize(s, A + p * sA, B + p * sB, Q + p * sn, R + p * sm, q + p * sn,
                          r + p * sm, d + p * sn, x0 + (size_t)p * n);
        oracle_solve_ex(s, 1, -1, 1);
      }
    }
    total = now_ms() - t0;
    for (int t = 0; t < nthreads; ++t) oracle_free(ss[t]);
    free(ss);
  }
  if (ms) *ms = total;
  return 0;
}

/*
 * Secondary witness: KKT residual ||K z - b||_2 from the raw problem.
 * z = [lam_1 x_1 u_1 ... lam_N x_N] (1-based knots), length (2n+m)N - m.
 * Rows:  -x_1 = -x0 ;  Q x_k + q_k - lam_k + A_k' lam_{k+1} = 0 ;
 *        R u_k + r_k + B_k' lam_{k+1} = 0 ;  A_k x_k + B_k u_k + d_k - x_{k+1} = 0.
 * Also returns ||b||_2 through *bnorm (b = [x0; q; r; d ...]).
 */
double oracle_kkt_residual(int n, int m, int N, const double* A, const double* B,
                           const double* Q, const double* R, const double* q, const double* r,
                           const double* d, const double* x0, const double* z, double* bnorm) {
  const int zb = 2 * n + m;
  double res = 0, bn = 0;
  for (int i = 0; i < n; ++i) {
    double e = z[n + i] - x0[i];
    res += e * e; bn += x0[i] * x0[i];
  }
  for (int k = 0; k < N; ++k) {
    const double* lam = z + (size_t)k * zb;
    const double* x = lam + n;
    const double* u = x + n;
    const double* lam_next = lam + zb;
    const double* x_next = lam_next + n;
    const double* Ak = A + (size_t)k * n * n;
    const double* Bk = B + (size_t)k * n * m;
    for (int i = 0; i < n; ++i) {
      double e = Q[(size_t)k * n + i] * x[i] + q[(size_t)k * n + i] - lam[i];
      if (k < N - 1)
        for (int j = 0; j < n; ++j) e += Ak[j + n * i] * lam_next[j];
      res += e * e; bn += q[(size_t)k * n + i] * q[(size_t)k * n + i];
    }
    if (k < N - 1) {
      for (int i = 0; i < m; ++i) {
        double e = R[(size_t)k * m + i] * u[i] + r[(size_t)k * m + i];
        for (int j = 0; j < n; ++j) e += Bk[j + n * i] * lam_next[j];
        res += e * e; bn += r[(size_t)k * m + i] * r[(size_t)k * m + i];
      }
      for (int i = 0; i < n; ++i) {
        double e = d[(size_t)k * n + i] - x_next[i];
        for (int j = 0; j < n; ++j) e += Ak[i + n * j] * x[j];
        for (int j = 0; j < m; ++j) e += Bk[i + n * j] * u[j];
        res += e * e; bn += d[(size_t)k * n + i] * d[(size_t)k * n + i];
      }
    }
  }
  if (bnorm) *bnorm = sqrt(bn);
  return sqrt(res);
}
