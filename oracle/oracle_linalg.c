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
