/*
 * oracle_linalg.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Unblocked dense solvers with the reference's loop order / summation order:
 *   LU with partial pivoting      linalg/lu.c:59-124  (kij, strict '>' pivot search)
 *   LU solve                      linalg/lu.c:166-201 (permute, unit-lower fwd, upper back)
 *   gaxpy Cholesky                linalg/cholesky.c:88-131 (dgemv per column, cblas/source_gemv_r.h:60-75)
 *   Cholesky solve                linalg/cholesky.c:163-185 (trsv Lower/NoTrans, Lower/Trans)
 *   trsv loops                    cblas/source_trsv_r.h:33-129
 *   permute                       permutation/permute_source.c:140 (x_new[i] = x_old[p[i]])
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>

int oracle_lu_decomp(size_t n, double *a, size_t lda, size_t *perm, int *signum)
{
  *signum = 1;
  for (size_t i = 0; i < n; i++) perm[i] = i;
  if (n == 0) return ORACLE_SUCCESS;

  for (size_t j = 0; j + 1 < n; j++) {
    /* pivot = first row attaining the column maximum (strict >, lu.c:82-93) */
    double big = fabs(a[j * lda + j]);
    size_t piv = j;
    for (size_t i = j + 1; i < n; i++) {
      double v = fabs(a[i * lda + j]);
      if (v > big) { big = v; piv = i; }
    }
    if (piv != j) {
      for (size_t k = 0; k < n; k++) {             /* matrix/swap_source.c:21 */
        double tmp = a[j * lda + k];
        a[j * lda + k] = a[piv * lda + k];
        a[piv * lda + k] = tmp;
      }
      size_t tp = perm[j]; perm[j] = perm[piv]; perm[piv] = tp;
      *signum = -*signum;
    }
    double ajj = a[j * lda + j];
    if (ajj != 0.0) {
      for (size_t i = j + 1; i < n; i++) {
        double l = a[i * lda + j] / ajj;
        a[i * lda + j] = l;
        for (size_t k = j + 1; k < n; k++)
          a[i * lda + k] = a[i * lda + k] - l * a[j * lda + k];
      }
    }
  }
  return ORACLE_SUCCESS;
}

int oracle_lu_singular(size_t n, const double *lu, size_t lda)
{
  for (size_t i = 0; i < n; i++)
    if (lu[i * lda + i] == 0) return 1;
  return 0;
}

/* forward substitution, row-major Lower/NoTrans (source_trsv_r.h:56-79) */
static void trsv_lower_notrans(size_t n, const double *a, size_t lda, double *x, int nonunit)
{
  if (n == 0) return;
  if (nonunit) x[0] = x[0] / a[0];
  for (size_t i = 1; i < n; i++) {
    double tmp = x[i];
    for (size_t j = 0; j < i; j++) tmp -= a[lda * i + j] * x[j];
    x[i] = nonunit ? tmp / a[lda * i + i] : tmp;
  }
}

/* back substitution, row-major Upper/NoTrans (source_trsv_r.h:33-55) */
static void trsv_upper_notrans(size_t n, const double *a, size_t lda, double *x, int nonunit)
{
  if (n == 0) return;
  if (nonunit) x[n - 1] = x[n - 1] / a[lda * (n - 1) + (n - 1)];
  for (size_t i = n - 1; i > 0 && i--;) {
    double tmp = x[i];
    for (size_t j = i + 1; j < n; j++) tmp -= a[lda * i + j] * x[j];
    x[i] = nonunit ? tmp / a[lda * i + i] : tmp;
  }
}

/* back substitution with the transpose of a lower factor (source_trsv_r.h:106-129) */
static void trsv_lower_trans(size_t n, const double *a, size_t lda, double *x, int nonunit)
{
  if (n == 0) return;
  if (nonunit) x[n - 1] = x[n - 1] / a[lda * (n - 1) + (n - 1)];
  for (size_t i = n - 1; i > 0 && i--;) {
    double tmp = x[i];
    for (size_t j = i + 1; j < n; j++) tmp -= a[lda * j + i] * x[j];
    x[i] = nonunit ? tmp / a[lda * i + i] : tmp;
  }
}

int oracle_lu_svx(size_t n, const double *lu, size_t lda, const size_t *perm, double *x)
{
  if (oracle_lu_singular(n, lu, lda)) return ORACLE_EDOM;   /* lu.c:181-184 */
  /* gsl_permute_vector: x <- P x, i.e. new[i] = old[perm[i]] */
  double stackbuf[16];
  double *tmp = n <= 16 ? stackbuf : (double *)malloc(n * sizeof(double));
  for (size_t i = 0; i < n; i++) tmp[i] = x[perm[i]];
  for (size_t i = 0; i < n; i++) x[i] = tmp[i];
  if (tmp != stackbuf) free(tmp);
  trsv_lower_notrans(n, lu, lda, x, 0);
  trsv_upper_notrans(n, lu, lda, x, 1);
  return ORACLE_SUCCESS;
}

int oracle_cholesky_decomp1(size_t n, double *a, size_t lda)
{
  /* keep the original matrix in the strict upper triangle (cholesky.c:103,
     matrix/swap_source.c:213: upper(i<j) <- lower) */
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < i; j++) a[j * lda + i] = a[i * lda + j];

  for (size_t j = 0; j < n; j++) {
    if (j > 0) {
      /* v = A(j:n,j) ; v -= A(j:n,0:j) * A(j,0:j)^T   (dgemv NoTrans, alpha=-1, beta=1):
         per row: temp = sum_k x[k]*A[i][k] (k ascending), y += alpha*temp */
      for (size_t i = j; i < n; i++) {
        double temp = 0.0;
        for (size_t k = 0; k < j; k++) temp += a[j * lda + k] * a[i * lda + k];
        a[i * lda + j] += -1.0 * temp;
      }
    }
    double ajj = a[j * lda + j];
    if (ajj <= 0.0) return ORACLE_EDOM;            /* cholesky.c:120-123 */
    ajj = sqrt(ajj);
    double inv = 1.0 / ajj;                          /* gsl_vector_scale(v, 1/ajj) */
    for (size_t i = j; i < n; i++) a[i * lda + j] *= inv;
  }
  return ORACLE_SUCCESS;
}

int oracle_cholesky_svx(size_t n, const double *llt, size_t lda, double *x)
{
  trsv_lower_notrans(n, llt, lda, x, 1);           /* L c = b   (cholesky.c:178) */
  trsv_lower_trans(n, llt, lda, x, 1);             /* L^T x = c (cholesky.c:181) */
  return ORACLE_SUCCESS;
}

/* ---- solver breadth behind the same facade ---------------------------- */

/* linalg/cholesky.c:312-338 (scale) + :355-388 (scale_apply) + :392-429 (decomp2) */
int oracle_cholesky_decomp2(size_t n, double *a, size_t lda, double *s)
{
  for (size_t i = 0; i < n; i++) {
    const double aii = a[i * lda + i];
    s[i] = aii <= 0.0 ? 1.0 : 1.0 / sqrt(aii);
  }
  for (size_t j = 0; j < n; j++)
    for (size_t i = j; i < n; i++) a[i * lda + j] *= s[i] * s[j];
  return oracle_cholesky_decomp1(n, a, lda);
}

/* linalg/cholesky.c:431-462: x *= S; L c = x; L^T x = c; x *= S */
int oracle_cholesky_svx2(size_t n, const double *llt, size_t lda, const double *s, double *x)
{
  for (size_t i = 0; i < n; i++) x[i] *= s[i];
  oracle_cholesky_svx(n, llt, lda, x);
  for (size_t i = 0; i < n; i++) x[i] *= s[i];
  return ORACLE_SUCCESS;
}

/* linalg/cholesky.c:541-582: 1-norm of the ORIGINAL matrix kept in the strict upper triangle of LLT,
   diagonal rebuilt from the rows of L (ddot, k ascending) */
static double cholesky_norm1(size_t n, const double *llt, size_t lda, double *work)
{
  double max = 0.0;
  for (size_t j = 0; j < n; j++) {
    double sum = 0.0, ajj = 0.0;
    for (size_t k = 0; k <= j; k++) ajj += llt[j * lda + k] * llt[j * lda + k];
    for (size_t i = 0; i < j; i++) {
      const double v = fabs(llt[i * lda + j]);
      sum += v;
      work[i] += v;
    }
    work[j] = sum + fabs(ajj);
  }
  for (size_t i = 0; i < n; i++) if (work[i] > max) max = work[i];
  return max;
}

static double asum(size_t n, const double *x) { double r = 0.0; for (size_t i = 0; i < n; i++) r += fabs(x[i]); return r; }

/* linalg/condest.c:95-188 (Hager / Higham estimator of |A^-1|_1, at most 5 iterations); solve(ctx, x): x := A^-1 x */
static double invnorm1(size_t n, void (*solve)(void *, double *), void *ctx, double *work)
{
  double *x = work, *v = work + n, *xi = work + 2 * n;
  for (size_t i = 0; i < n; i++) x[i] = 1.0 / (double)n;
  for (size_t i = 0; i < n; i++) v[i] = x[i];
  solve(ctx, v);
  double gamma = asum(n, v), gamma_old;
  for (size_t i = 0; i < n; i++) xi[i] = v[i] >= 0.0 ? 1 : -1;
  for (size_t i = 0; i < n; i++) x[i] = xi[i];
  solve(ctx, x);
  for (size_t k = 0; k < 5; k++) {
    size_t j = 0;                                   /* idamax: first index of the largest |x| (source_iamax_r.h) */
    double big = 0.0;
    for (size_t i = 0; i < n; i++) if (fabs(x[i]) > big) { big = fabs(x[i]); j = i; }
    for (size_t i = 0; i < n; i++) v[i] = 0.0;
    v[j] = 1.0;
    solve(ctx, v);
    gamma_old = gamma;
    gamma = asum(n, v);
    int same = 1;
    for (size_t i = 0; i < n; i++) if ((v[i] >= 0.0) != (xi[i] >= 0.0)) { same = 0; break; }
    if (same || gamma < gamma_old) break;
    for (size_t i = 0; i < n; i++) xi[i] = v[i] >= 0.0 ? 1 : -1;
    for (size_t i = 0; i < n; i++) x[i] = xi[i];
    solve(ctx, x);
  }
  double temp = 1.0;
  for (size_t i = 0; i < n; i++) { x[i] = temp * (1.0 + (double)i / ((double)n - 1.0)); temp = -temp; }
  solve(ctx, x);
  temp = 2.0 * asum(n, x) / (3.0 * (double)n);
  if (temp > gamma) gamma = temp;
  return gamma;
}

struct chol_ctx { size_t n, lda; const double *m; const size_t *perm; };
static void chol_solve_cb(void *c, double *x) { struct chol_ctx *q = (struct chol_ctx *)c; oracle_cholesky_svx(q->n, q->m, q->lda, x); }

/* gsl_linalg_cholesky_rcond (cholesky.c:499-537), A^-1 = L^-T L^-1 (cholesky.c:584-604) */
int oracle_cholesky_rcond(size_t n, const double *llt, size_t lda, double *rcond, double *work)
{
  *rcond = 0.0;
  if (n == 0) return ORACLE_SUCCESS;
  const double anorm = cholesky_norm1(n, llt, lda, work);
  if (anorm == 0.0) return ORACLE_SUCCESS;
  struct chol_ctx c = {n, lda, llt, NULL};
  const double gamma = invnorm1(n, chol_solve_cb, &c, work);
  if (gamma != 0.0) *rcond = (1.0 / anorm) / gamma;
  return ORACLE_SUCCESS;
}

/* linalg/lu.c:204-252: work = A x - b (dgemv, source_gemv_r.h:60-75: y = beta y, then += alpha * dot);
   LU delta = work; x -= delta */
int oracle_lu_refine(size_t n, const double *a, size_t lda, const double *lu, size_t ldlu, const size_t *perm,
                     const double *b, double *x, double *work)
{
  if (oracle_lu_singular(n, lu, ldlu)) return ORACLE_EDOM;
  for (size_t i = 0; i < n; i++) work[i] = b[i];
  for (size_t i = 0; i < n; i++) work[i] *= -1.0;
  for (size_t i = 0; i < n; i++) {
    double temp = 0.0;
    for (size_t j = 0; j < n; j++) temp += x[j] * a[lda * i + j];
    work[i] += 1.0 * temp;
  }
  const int st = oracle_lu_svx(n, lu, ldlu, perm, work);
  for (size_t i = 0; i < n; i++) x[i] += -1.0 * work[i];          /* daxpy(-1, work, x) */
  return st;
}

/* linalg/pcholesky.c:71-130: outer-product LDL^T with diagonal pivoting (Golub & Van Loan alg. 4.2.2);
   lower triangle holds L (unit diagonal implied) and D on the diagonal; original kept in the upper triangle */
int oracle_pcholesky_decomp(size_t n, double *a, size_t lda, size_t *perm)
{
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < i; j++) a[j * lda + i] = a[i * lda + j];          /* transpose_tricpy('L', 0) */
  for (size_t i = 0; i < n; i++) perm[i] = i;
  for (size_t k = 0; k < n; k++) {
    size_t j = k;                                     /* gsl_vector_max_index over the diagonal k..n-1: first maximum */
    double max = a[k * lda + k];
    for (size_t i = k; i < n; i++) if (a[i * lda + i] > max) { max = a[i * lda + i]; j = i; }
    { const size_t t = perm[k]; perm[k] = perm[j]; perm[j] = t; }
    if (k != j) {                                     /* cholesky_common.c:34-86, lower triangle only */
      const size_t ii = k, jj = j;
      for (size_t c = 0; c < ii; c++) { double t = a[ii * lda + c]; a[ii * lda + c] = a[jj * lda + c]; a[jj * lda + c] = t; }
      for (size_t c = ii + 1; c < jj; c++) { double t = a[jj * lda + c]; a[jj * lda + c] = a[c * lda + ii]; a[c * lda + ii] = t; }
      for (size_t c = jj + 1; c < n; c++) { double t = a[c * lda + ii]; a[c * lda + ii] = a[c * lda + jj]; a[c * lda + jj] = t; }
      { double t = a[ii * lda + ii]; a[ii * lda + ii] = a[jj * lda + jj]; a[jj * lda + jj] = t; }
    }
    if (k + 1 < n) {
      const double alpha = a[k * lda + k], alphainv = 1.0 / alpha;
      /* dsyr(Lower, -alphainv, v, m): m[i][j] += v[j] * (-alphainv * v[i]), j <= i  (source_syr.h:46-56) */
      for (size_t i = k + 1; i < n; i++) {
        const double tmp = -alphainv * a[i * lda + k];
        for (size_t c = k + 1; c <= i; c++) a[i * lda + c] += a[c * lda + k] * tmp;
      }
      for (size_t i = k + 1; i < n; i++) a[i * lda + k] *= alphainv;
    }
  }
  return ORACLE_SUCCESS;
}

/* linalg/pcholesky.c:190-229: x = P b; L w = x (unit); y = w / D; L^T z = y (unit); x = P^T z */
int oracle_pcholesky_svx(size_t n, const double *ldlt, size_t lda, const size_t *perm, double *x)
{
  double *tmp = (double *)malloc((n ? n : 1) * sizeof(double));
  for (size_t i = 0; i < n; i++) tmp[i] = x[perm[i]];
  for (size_t i = 0; i < n; i++) x[i] = tmp[i];
  trsv_lower_notrans(n, ldlt, lda, x, 0);
  for (size_t i = 0; i < n; i++) x[i] /= ldlt[i * lda + i];
  trsv_lower_trans(n, ldlt, lda, x, 0);
  for (size_t i = 0; i < n; i++) tmp[perm[i]] = x[i];                           /* permute_vector_inverse */
  for (size_t i = 0; i < n; i++) x[i] = tmp[i];
  free(tmp);
  return ORACLE_SUCCESS;
}


/* linalg/pcholesky.c:231-273: keep A in the upper triangle, scale (cholesky.c:312-388), pivoted LDL^T of S A S */
int oracle_pcholesky_decomp2(size_t n, double *a, size_t lda, size_t *perm, double *s)
{
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < i; j++) a[j * lda + i] = a[i * lda + j];
  for (size_t i = 0; i < n; i++) {
    const double aii = a[i * lda + i];
    s[i] = aii <= 0.0 ? 1.0 : 1.0 / sqrt(aii);
  }
  for (size_t j = 0; j < n; j++)
    for (size_t i = j; i < n; i++) a[i * lda + j] *= s[i] * s[j];
  /* pcholesky_decomp(copy_uplo = 0): the shared decomposition copies the lower triangle up first, which would
     overwrite the saved original -- save and restore the strict upper triangle around it */
  double *up = (double *)malloc(n * n * sizeof(double));
  if (!up) return ORACLE_FAILURE;
  for (size_t i = 0; i < n; i++) for (size_t j = i + 1; j < n; j++) up[i * n + j] = a[i * lda + j];
  const int st = oracle_pcholesky_decomp(n, a, lda, perm);
  for (size_t i = 0; i < n; i++) for (size_t j = i + 1; j < n; j++) a[i * lda + j] = up[i * n + j];
  free(up);
  return st;
}

/* linalg/pcholesky.c:314-353 */
int oracle_pcholesky_svx2(size_t n, const double *ldlt, size_t lda, const size_t *perm, const double *s, double *x)
{
  for (size_t i = 0; i < n; i++) x[i] *= s[i];
  oracle_pcholesky_svx(n, ldlt, lda, perm, x);
  for (size_t i = 0; i < n; i++) x[i] *= s[i];
  return ORACLE_SUCCESS;
}

static void pchol_solve_cb(void *c, double *x) { struct chol_ctx *q = (struct chol_ctx *)c; oracle_pcholesky_svx(q->n, q->m, q->lda, q->perm, x); }

/* linalg/pcholesky.c:472-580: 1-norm of the matrix kept in the strict upper triangle with its diagonal rebuilt
   from L D L^T (in pivoted order, then un-permuted), times the estimate of |A^-1|_1; work: 3 n */
int oracle_pcholesky_rcond(size_t n, const double *ldlt, size_t lda, const size_t *perm, double *rcond, double *work)
{
  *rcond = 0.0;
  if (n == 0) return ORACLE_SUCCESS;
  double *diag = work + n, *tmp = work + 2 * n;
  for (size_t j = 0; j < n; j++) {
    double ajj = ldlt[j * lda + j];
    for (size_t i = 0; i < j; i++) { const double di = ldlt[i * lda + i], l = ldlt[j * lda + i]; ajj += di * l * l; }
    tmp[j] = ajj;
  }
  for (size_t i = 0; i < n; i++) diag[perm[i]] = tmp[i];          /* gsl_permute_vector_inverse */
  for (size_t i = 0; i < n; i++) work[i] = 0.0;
  double anorm = 0.0;
  for (size_t j = 0; j < n; j++) {
    double sum = 0.0;
    for (size_t i = 0; i < j; i++) { const double v = fabs(ldlt[i * lda + j]); sum += v; work[i] += v; }
    work[j] = sum + fabs(diag[j]);
  }
  for (size_t i = 0; i < n; i++) if (work[i] > anorm) anorm = work[i];
  if (anorm == 0.0) return ORACLE_SUCCESS;
  struct chol_ctx c = {n, lda, ldlt, perm};
  const double gamma = invnorm1(n, pchol_solve_cb, &c, work);
  if (gamma != 0.0) *rcond = (1.0 / anorm) / gamma;
  return ORACLE_SUCCESS;
}
