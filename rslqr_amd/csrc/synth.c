/*
 * synth.c -- seeded synthetic LQR problems (host, plain C).
 *
 * The reference ships only four JSON fixtures (no generator), so the
 * benchmark family of BASELINE.json (nx,nu,N,batch) is generated here, following SURVEY.md 8d:
 * a splitmix64 stream per problem; draws, in this order:
 *   x0_i ~ N(0,1)                                    i < n
 *   per knot k = 0..N-1:
 *     for i < n:  Q_i ~ U[0.5,2], q_i ~ N(0,1), d_i ~ 0.1 N(0,1)
 *     for i < m:  R_i ~ U[0.01,0.1], r_i ~ N(0,1)
 *     G (n x n, column-major order) ~ N(0,1)/sqrt(n);  A = (1 - h g) I + h (G - G')/2, h = g = 0.1
 *     B (n x m, column-major order) ~ h N(0,1)
 * Time-varying on purpose: nothing can be constant-folded. Normals are Box-Muller (cos branch)
 * on 53-bit uniforms, so the stream is reproducible from the seed alone.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ndlqr.h"

typedef struct { uint64_t state; } Rng;

static uint64_t next_u64(Rng* g) {
  uint64_t z = (g->state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double uniform01(Rng* g) { return (double)(next_u64(g) >> 11) * 0x1.0p-53; }
static double uniform(Rng* g, double lo, double hi) { return lo + (hi - lo) * uniform01(g); }
static double normal(Rng* g) {
  const double u1 = ((double)(next_u64(g) >> 11) + 1.0) * 0x1.0p-53; /* (0,1] */
  const double u2 = uniform01(g);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

int ndlqr_GenerateSyntheticFlat(int nstates, int ninputs, int nhorizon, uint64_t seed, double* A,
                                double* B, double* Q, double* R, double* q, double* r, double* d,
                                double* x0) {
  const int n = nstates, m = ninputs, N = nhorizon;
  if (n <= 0 || m <= 0 || N <= 0) return NDLQR_ERR_INVALID;
  const double h = 0.1, gamma = 0.1;
  Rng g = {seed};
  double* G = (double*)malloc(sizeof(double) * (size_t)n * n);
  if (!G) return NDLQR_ERR_INVALID;
  for (int i = 0; i < n; ++i) x0[i] = normal(&g);
  const double gscale = 1.0 / sqrt((double)n);
  for (int k = 0; k < N; ++k) {
    double* Qk = Q + (size_t)k * n; double* qk = q + (size_t)k * n; double* dk = d + (size_t)k * n;
    double* Rk = R + (size_t)k * m; double* rk = r + (size_t)k * m;
    double* Ak = A + (size_t)k * n * n; double* Bk = B + (size_t)k * n * m;
    for (int i = 0; i < n; ++i) {
      Qk[i] = uniform(&g, 0.5, 2.0);
      qk[i] = normal(&g);
      dk[i] = 0.1 * normal(&g);
    }
    for (int i = 0; i < m; ++i) {
      Rk[i] = uniform(&g, 0.01, 0.1);
      rk[i] = normal(&g);
    }
    for (int e = 0; e < n * n; ++e) G[e] = normal(&g) * gscale;
    for (int c = 0; c < n; ++c)
      for (int rr = 0; rr < n; ++rr) {
        double a = h * 0.5 * (G[rr + n * c] - G[c + n * rr]);
        if (rr == c) a += 1.0 - h * gamma;
        Ak[rr + n * c] = a;
      }
    for (int e = 0; e < n * m; ++e) Bk[e] = h * normal(&g);
  }
  free(G);
  return NDLQR_OK;
}

LQRProblem* ndlqr_NewSyntheticLQRProblem(int nstates, int ninputs, int nhorizon, uint64_t seed) {
  const size_t n = (size_t)nstates, m = (size_t)ninputs, N = (size_t)nhorizon;
  LQRProblem* prob = ndlqr_NewLQRProblem(nstates, ninputs, nhorizon);
  if (!prob) return NULL;
  double* buf = (double*)malloc(sizeof(double) * (N * (n * n + n * m + 3 * n + 2 * m) + n));
  if (!buf) { ndlqr_FreeLQRProblem(prob); return NULL; }
  double* A = buf; double* B = A + N * n * n; double* Q = B + N * n * m; double* R = Q + N * n;
  double* q = R + N * m; double* r = q + N * n; double* d = r + N * m; double* x0 = d + N * n;
  ndlqr_GenerateSyntheticFlat(nstates, ninputs, nhorizon, seed, A, B, Q, R, q, r, d, x0);
  for (size_t k = 0; k < N; ++k)
    ndlqr_InitializeLQRData(prob->lqrdata[k], Q + k * n, R + k * m, q + k * n, r + k * m, 0.0,
                            A + k * n * n, B + k * n * m, d + k * n);
  memcpy(prob->x0, x0, sizeof(double) * n);
  free(buf);
  return prob;
}
