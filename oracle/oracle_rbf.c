/*
 * oracle_rbf.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * RBF interpolation is NOT implemented in the reference (README:18-26 lists it
 * as future work), so this harness is the composition SURVEY.md 3.3/3.4 fixes:
 *   fill   Phi_ij = phi(|x_i - x_j|) with libm exp/log           (no reference code)
 *   solve  Gaussian, Wendland: gsl_linalg_cholesky_decomp1 + _svx (linalg/cholesky.c:88,163)
 *          TPS:      gsl_linalg_LU_decomp + _svx (not SPD)       (linalg/lu.c:59,166)
 *   eval   s(y) = sum_j w_j phi(|y - x_j|), j ascending          (no reference code)
 * The RBF kernels are therefore "parity unpinned" by any reference test; the
 * solvers underneath are pinned (tests/test_oracle_linalg.py).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

double oracle_rbf_phi(int kind, double eps, double r2)
{
  if (kind == ORACLE_RBF_GAUSSIAN) return exp(-(eps * eps) * r2);
  if (kind == ORACLE_RBF_WENDLAND) {                 /* (1 - eps r)_+^4 (4 eps r + 1), Wendland C2 */
    const double t = eps * sqrt(r2), u = 1.0 - t;
    if (u <= 0.0) return 0.0;
    return (u * u) * (u * u) * (4.0 * t + 1.0);
  }
  if (r2 == 0.0) return 0.0;
  return 0.5 * r2 * log(r2);
}

static double dist2(const double *a, const double *b, int dim)
{
  double s = 0.0;
  for (int c = 0; c < dim; c++) { double d = a[c] - b[c]; s += d * d; }
  return s;
}

void oracle_rbf_fill(int kind, double eps, const double *x, size_t n, int dim, size_t tda,
                     double *phi, size_t lda)
{
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < n; j++)
      phi[i * lda + j] = oracle_rbf_phi(kind, eps, dist2(x + i * tda, x + j * tda, dim));
}

int oracle_rbf_solve(int kind, double eps, const double *x, size_t n, int dim, size_t tda,
                     const double *f, double *w)
{
  double *phi = (double *)malloc(n * n * sizeof(double));
  if (!phi) return ORACLE_FAILURE;
  oracle_rbf_fill(kind, eps, x, n, dim, tda, phi, n);
  memcpy(w, f, n * sizeof(double));
  int status;
  if (kind != ORACLE_RBF_TPS) {                      /* positive definite kernels: Gaussian, Wendland */
    status = oracle_cholesky_decomp1(n, phi, n);
    if (status == ORACLE_SUCCESS) status = oracle_cholesky_svx(n, phi, n, w);
  } else {
    size_t *perm = (size_t *)malloc(n * sizeof(size_t));
    int signum;
    status = oracle_lu_decomp(n, phi, n, perm, &signum);
    if (status == ORACLE_SUCCESS) status = oracle_lu_svx(n, phi, n, perm, w);
    free(perm);
  }
  free(phi);
  return status;
}

void oracle_rbf_eval(int kind, double eps, const double *x, size_t n, int dim, size_t tda,
                     const double *w, const double *y, size_t m, size_t ytda, double *s)
{
  for (size_t k = 0; k < m; k++) {
    double acc = 0.0;
    for (size_t j = 0; j < n; j++)
      acc += w[j] * oracle_rbf_phi(kind, eps, dist2(y + k * ytda, x + j * tda, dim));
    s[k] = acc;
  }
}

/* Ordinary kriging with covariance C(h) = phi(h) and nugget eta (README:24 lists kriging as future work: no
   reference code, PARITY UNPINNED; the solver underneath is the pinned gsl_linalg_cholesky_decomp1 + _svx, with
   gsl_linalg_pcholesky_decomp + _svx as the semi-definite fall-back).  Dual form:
       [K 1; 1^T 0] [w; mu] = [f; 0],  K = Phi + eta I      ->  a = K^-1 f, b = K^-1 1, mu = (1^T a) / (1^T b), w = a - mu b
       s(y) = mu + sum_j w_j phi(|y - x_j|)                                                                          */
int oracle_krige_solve(int kind, double eps, double nugget, const double *x, size_t n, int dim, size_t tda,
                       const double *f, double *w, double *mean)
{
  double *phi = (double *)malloc(n * n * sizeof(double)), *b = (double *)malloc(n * sizeof(double));
  if (!phi || !b) { free(phi); free(b); return ORACLE_FAILURE; }
  oracle_rbf_fill(kind, eps, x, n, dim, tda, phi, n);
  for (size_t i = 0; i < n; i++) { phi[i * n + i] += nugget; b[i] = 1.0; }
  memcpy(w, f, n * sizeof(double));
  int status = oracle_cholesky_decomp1(n, phi, n);
  if (status == ORACLE_SUCCESS) {
    oracle_cholesky_svx(n, phi, n, w);
    oracle_cholesky_svx(n, phi, n, b);
  } else {                                            /* semi-definite: pivoted LDL^T (linalg/pcholesky.c:71-229) */
    size_t *perm = (size_t *)malloc(n * sizeof(size_t));
    oracle_rbf_fill(kind, eps, x, n, dim, tda, phi, n);
    for (size_t i = 0; i < n; i++) phi[i * n + i] += nugget;
    status = oracle_pcholesky_decomp(n, phi, n, perm);
    if (status == ORACLE_SUCCESS) { oracle_pcholesky_svx(n, phi, n, perm, w); oracle_pcholesky_svx(n, phi, n, perm, b); }
    free(perm);
  }
  if (status == ORACLE_SUCCESS) {
    double sa = 0.0, sb = 0.0;
    for (size_t i = 0; i < n; i++) { sa += w[i]; sb += b[i]; }
    const double mu = sa / sb;
    for (size_t i = 0; i < n; i++) w[i] = w[i] - mu * b[i];
    *mean = mu;
  }
  free(phi); free(b);
  return status;
}

void oracle_krige_eval(int kind, double eps, double mean, const double *x, size_t n, int dim, size_t tda,
                       const double *w, const double *y, size_t m, size_t ytda, double *s)
{
  oracle_rbf_eval(kind, eps, x, n, dim, tda, w, y, m, ytda, s);
  for (size_t k = 0; k < m; k++) s[k] = s[k] + mean;
}

/* Thin-plate spline with its affine tail ("N + d + 1 with affine augmentation"; no
   reference code, PARITY UNPINNED): the saddle system
       [Phi P; P^T 0] [w; c] = [f; 0],   P = [1, x] in raw coordinates,
   solved the way the reference would solve an indefinite system: gsl_linalg_LU_decomp + _svx (linalg/lu.c:59-201,
   pinned) of the full (n + dim + 1) matrix;  s(y) = sum_j w_j phi(|y - x_j|) (j ascending) + c_0 + sum_a c_a y_a. */
int oracle_rbf_solve_affine(int kind, double eps, const double *x, size_t n, int dim, size_t tda,
                            const double *f, double *w, double *c)
{
  const size_t k = (size_t)dim + 1, na = n + k;
  double *a = (double *)calloc(na * na, sizeof(double)), *rhs = (double *)calloc(na, sizeof(double));
  size_t *perm = (size_t *)malloc(na * sizeof(size_t));
  if (!a || !rhs || !perm) { free(a); free(rhs); free(perm); return ORACLE_FAILURE; }
  oracle_rbf_fill(kind, eps, x, n, dim, tda, a, na);
  for (size_t i = 0; i < n; i++) {
    for (size_t q = 0; q < k; q++) {
      const double v = q == 0 ? 1.0 : x[i * tda + q - 1];
      a[i * na + n + q] = v;
      a[(n + q) * na + i] = v;
    }
    rhs[i] = f[i];
  }
  int signum;
  int status = oracle_lu_decomp(na, a, na, perm, &signum);
  if (status == ORACLE_SUCCESS) status = oracle_lu_svx(na, a, na, perm, rhs);
  if (status == ORACLE_SUCCESS) {
    memcpy(w, rhs, n * sizeof(double));
    for (size_t q = 0; q < k; q++) c[q] = rhs[n + q];
  }
  free(a); free(rhs); free(perm);
  return status;
}

void oracle_rbf_eval_affine(int kind, double eps, const double *c, const double *x, size_t n, int dim, size_t tda,
                            const double *w, const double *y, size_t m, size_t ytda, double *s)
{
  oracle_rbf_eval(kind, eps, x, n, dim, tda, w, y, m, ytda, s);
  for (size_t k = 0; k < m; k++) {
    double t = c[0];
    for (int a = 0; a < dim; a++) t += c[1 + a] * y[k * ytda + a];
    s[k] = s[k] + t;
  }
}
