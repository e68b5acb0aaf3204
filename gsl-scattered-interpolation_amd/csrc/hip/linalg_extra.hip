/*
 * linalg_extra.hip -- solver breadth behind the same facade.  Compiled with
 * -ffp-contract=off: the pivoted LDL^T chooses its pivots by comparing diagonal entries, so its arithmetic is
 * the reference's, operation for operation (same pivots, same bits as the CPU restatement).
 *
 * Replaces (reference file:line):
 *   gsl_linalg_cholesky_decomp2      linalg/cholesky.c:392-429   = scale (:312-338) + scale_apply (:355-388) + decomp1
 *   gsl_linalg_cholesky_svx2/solve2  linalg/cholesky.c:431-497   x *= S; L c = x; L^T x = c; x *= S
 *   gsl_linalg_cholesky_rcond        linalg/cholesky.c:499-537   1-norm of the original matrix kept in the strict upper
 *                                                                triangle (:541-582) x the Hager / Higham estimate of
 *                                                                |A^-1|_1 (linalg/condest.c:95-188, <= 5 iterations)
 *   gsl_linalg_LU_refine             linalg/lu.c:204-252         work = A x - b; LU delta = work; x -= delta
 *   gsl_linalg_pcholesky_decomp      linalg/pcholesky.c:71-154   outer-product LDL^T with diagonal pivoting
 *                                                                (Golub & Van Loan alg. 4.2.2): P A P^T = L D L^T
 *   gsl_linalg_pcholesky_svx/solve   linalg/pcholesky.c:156-229
 * The O(N^3) work of decomp2 is the MFMA Cholesky of chol.hip; rcond and LU_refine are a handful of the
 * blocked triangular sweeps; the pivoted LDL^T is Level-2 by nature (every step needs the freshly updated
 * diagonal): two launches per column, HBM-bound rank-1 updates -- the fallback for semi-definite / nuggeted
 * kernel matrices that plain Cholesky refuses, not a fast path.
 */
#include "common.h"
#include <math.h>
#include <stdlib.h>

/* ------------------------------------------------------------------------ */
/* scaled Cholesky                                                           */
__global__ void __launch_bounds__(256)
chol_scale_kernel(const double *__restrict__ a, size_t lda, size_t n, double *__restrict__ s)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double aii = a[i * lda + i];
  s[i] = aii <= 0.0 ? 1.0 : 1.0 / sqrt(aii);          /* cholesky.c:330-335 */
}

__global__ void __launch_bounds__(256)
chol_scale_apply_kernel(double *__restrict__ a, size_t lda, size_t n, const double *__restrict__ s)
{
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j > i || i >= n) return;                         /* lower triangle, diagonal included (cholesky.c:373-384) */
  a[i * lda + j] *= s[i] * s[j];
}

__global__ void __launch_bounds__(256)
vec_mul_kernel(double *__restrict__ x, const double *__restrict__ s, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= s[i];
}

extern "C" int gsl_sinterp_hip_cholesky_decomp2(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, double *d_s,
                                                int *h_info)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, lda >= n && n <= 65535u * 256u, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_a && d_s), ST_EFAULT);
  if (h_info) *h_info = 0;
  if (n == 0) return ST_SUCCESS;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(chol_scale_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)d_a, lda, n, d_s);
  hipLaunchKernelGGL(chol_scale_apply_kernel, dim3(nb, (unsigned)n), dim3(256), 0, ctx->stream, d_a, lda, n, (const double *)d_s);
  LAUNCH_CHECK(ctx);
  return gsl_sinterp_hip_cholesky_decomp1(ctx, n, d_a, lda, h_info);   /* copies the (scaled) lower triangle into the upper one first */
}

extern "C" int gsl_sinterp_hip_cholesky_svx2(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_llt, size_t lda, const double *d_s,
                                             double *d_x)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_llt && d_s && d_x), ST_EFAULT);
  if (n == 0) return ST_SUCCESS;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(vec_mul_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_x, d_s, n);
  LAUNCH_CHECK(ctx);
  int st = gsl_sinterp_hip_cholesky_svx(ctx, n, d_llt, lda, d_x);
  if (st) return st;
  hipLaunchKernelGGL(vec_mul_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_x, d_s, n);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

/* ------------------------------------------------------------------------ */
/* reciprocal condition number                                               */
/* column j of |A| summed: the strict upper triangle holds the original matrix (rows i < j of column j, and -- by
   symmetry -- row j right of the diagonal for the entries below it); A_jj = sum_k L_jk^2.  One workgroup per j. */
__global__ void __launch_bounds__(256)
chol_norm1_kernel(const double *__restrict__ llt, size_t lda, size_t n, unsigned long long *__restrict__ out)
{
  __shared__ double s_red[4];
  const size_t j = blockIdx.x;
  double acc = 0.0, ajj = 0.0;
  for (size_t i = threadIdx.x; i < j; i += 256) acc += fabs(llt[i * lda + j]);          /* column j above the diagonal */
  for (size_t c = j + 1 + threadIdx.x; c < n; c += 256) acc += fabs(llt[j * lda + c]);  /* row j right of it = column j below */
  for (size_t k = threadIdx.x; k <= j; k += 256) { const double l = llt[j * lda + k]; ajj += l * l; }
  acc += ajj;                                                                            /* |A_jj| = A_jj >= 0 */
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    atomicMax(out, (unsigned long long)__double_as_longlong(t));                         /* t >= 0: bit order = value order */
  }
}

static double host_asum(const double *x, size_t n) { double r = 0.0; for (size_t i = 0; i < n; i++) r += fabs(x[i]); return r; }

/* condest.c:95-188 (Hager / Higham estimate of |A^-1|_1, at most 5 iterations) with x := A^-1 x done on the device by
   `solve(d_v)`; the O(N) vector work of the estimator (signs, argmax, 1-norms) stays on the host: at most 8 solves of
   one vector each */
template <class Solve>
static int invnorm1_device(gsl_sinterp_hip_ctx *ctx, size_t n, Solve solve, double *gamma_out)
{
  double *d_v = NULL;
  double *x = (double *)malloc(3 * n * sizeof(double));
  if (!x) return sinterp_fail(ctx, ST_ENOMEM, "rcond: host workspace", hipSuccess, __FILE__, __LINE__);
  double *v = x + n, *xi = x + 2 * n;
  int st = gsl_sinterp_hip_malloc(ctx, (void **)&d_v, n * sizeof(double));
  auto ainv = [&](double *h) -> int {
    int s = gsl_sinterp_hip_h2d(ctx, d_v, h, n * sizeof(double));
    if (!s) s = solve(d_v);
    if (!s) s = gsl_sinterp_hip_d2h(ctx, h, d_v, n * sizeof(double));
    return s;
  };
  double gamma = 0.0, gamma_old;
  if (!st) {
    for (size_t i = 0; i < n; i++) x[i] = 1.0 / (double)n;
    memcpy(v, x, n * sizeof(double));
    st = ainv(v);
  }
  if (!st) {
    gamma = host_asum(v, n);
    for (size_t i = 0; i < n; i++) xi[i] = v[i] >= 0.0 ? 1 : -1;
    memcpy(x, xi, n * sizeof(double));
    st = ainv(x);
  }
  for (size_t k = 0; k < 5 && !st; k++) {
    size_t j = 0;
    double big = 0.0;
    for (size_t i = 0; i < n; i++) if (fabs(x[i]) > big) { big = fabs(x[i]); j = i; }   /* idamax: first maximum */
    memset(v, 0, n * sizeof(double));
    v[j] = 1.0;
    st = ainv(v);
    if (st) break;
    gamma_old = gamma;
    gamma = host_asum(v, n);
    bool same = true;
    for (size_t i = 0; i < n; i++) if ((v[i] >= 0.0) != (xi[i] >= 0.0)) { same = false; break; }
    if (same || gamma < gamma_old) break;
    for (size_t i = 0; i < n; i++) xi[i] = v[i] >= 0.0 ? 1 : -1;
    memcpy(x, xi, n * sizeof(double));
    st = ainv(x);
  }
  if (!st) {
    double temp = 1.0;
    for (size_t i = 0; i < n; i++) { x[i] = temp * (1.0 + (double)i / ((double)n - 1.0)); temp = -temp; }
    st = ainv(x);
    if (!st) {
      temp = 2.0 * host_asum(x, n) / (3.0 * (double)n);
      if (temp > gamma) gamma = temp;
    }
  }
  gsl_sinterp_hip_free(ctx, d_v);
  free(x);
  *gamma_out = gamma;
  return st;
}

extern "C" int gsl_sinterp_hip_cholesky_rcond(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_llt, size_t lda, double *h_rcond)
{
  REQUIRE(ctx, ctx != NULL && h_rcond != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  *h_rcond = 0.0;
  if (n == 0) return ST_SUCCESS;
  REQUIRE(ctx, d_llt != NULL, ST_EFAULT);
  unsigned long long *d_norm = (unsigned long long *)((char *)ctx->d_scratch + 256);
  HIP_OK(ctx, hipMemsetAsync(d_norm, 0, sizeof *d_norm, ctx->stream));
  hipLaunchKernelGGL(chol_norm1_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_llt, lda, n, d_norm);
  LAUNCH_CHECK(ctx);
  unsigned long long bits = 0;
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(&bits, d_norm, sizeof bits, hipMemcpyDeviceToHost));
  double anorm;
  memcpy(&anorm, &bits, sizeof anorm);
  if (anorm == 0.0) return ST_SUCCESS;                  /* cholesky.c:523-524 */

  double gamma = 0.0;
  int st = invnorm1_device(ctx, n, [&](double *d_v) -> int { return gsl_sinterp_hip_cholesky_svx(ctx, n, d_llt, lda, d_v); }, &gamma);
  if (!st && gamma != 0.0) *h_rcond = (1.0 / anorm) / gamma;
  return st;
}

/* ------------------------------------------------------------------------ */
/* LU refinement                                                             */
/* work_i = sum_j A_ij x_j - b_i : one workgroup per row */
__global__ void __launch_bounds__(256)
residual_kernel(const double *__restrict__ a, size_t lda, size_t n, const double *__restrict__ x, const double *__restrict__ b,
                double *__restrict__ work)
{
  __shared__ double s_red[4];
  const size_t i = blockIdx.x;
  double acc = 0.0;
  for (size_t j = threadIdx.x; j < n; j += 256) acc += a[i * lda + j] * x[j];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) work[i] = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) - b[i];
}

__global__ void __launch_bounds__(256)
vec_sub_kernel(double *__restrict__ x, const double *__restrict__ d, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = x[i] - d[i];                          /* daxpy(-1, work, x), lu.c:246 */
}

extern "C" int gsl_sinterp_hip_lu_refine(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_a, size_t lda, const double *d_lu,
                                         size_t ldlu, const int *d_perm, const double *d_b, double *d_x, double *d_work)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, lda >= n && ldlu >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_a && d_lu && d_perm && d_b && d_x && d_work), ST_EFAULT);
  if (n == 0) return ST_SUCCESS;
  hipLaunchKernelGGL(residual_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_a, lda, n, (const double *)d_x, d_b, d_work);
  LAUNCH_CHECK(ctx);
  const int st = gsl_sinterp_hip_lu_svx(ctx, n, d_lu, ldlu, d_perm, d_work);     /* GSL_EDOM when LU is singular (lu.c:231-234) */
  if (st) return st;
  hipLaunchKernelGGL(vec_sub_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_x, (const double *)d_work, n);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

/* ------------------------------------------------------------------------ */
/* pivoted LDL^T                                                              */
/* step k, part 1 (one workgroup): j = first index of the largest remaining diagonal entry (gsl_vector_max_index,
   vector/minmax_source.c:112-139); swap perm[k], perm[j]; swap row/column k and j of the lower triangle
   (cholesky_common.c:34-86); copy the new column k (rows k+1..n-1) to v[]; publish 1/alpha. */
__global__ void __launch_bounds__(1024)
pchol_pivot_kernel(double *__restrict__ a, size_t lda, size_t n, size_t k, int *__restrict__ perm, double *__restrict__ v,
                   double *__restrict__ alphainv_out)
{
  __shared__ double s_val[16];
  __shared__ unsigned s_idx[16];
  __shared__ unsigned s_j;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double best = -INFINITY;
  unsigned bi = 0xffffffffu;
  for (size_t i = k + tid; i < n; i += 1024) {
    const double d = a[i * lda + i];
    if (d > best || bi == 0xffffffffu) { best = d; bi = (unsigned)i; }     /* ascending i per thread: keeps the first maximum */
  }
  auto better = [](double v1, unsigned i1, double v2, unsigned i2) {        /* (v1,i1) beats (v2,i2)? larger value, then smaller index */
    if (i1 == 0xffffffffu) return false;
    if (i2 == 0xffffffffu) return true;
    return v1 > v2 || (v1 == v2 && i1 < i2);
  };
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_xor(best, off);
    const unsigned oi = __shfl_xor(bi, off);
    if (better(ov, oi, best, bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    double bv = s_val[0];
    unsigned bj = s_idx[0];
    for (int w = 1; w < 16; w++) if (better(s_val[w], s_idx[w], bv, bj)) { bv = s_val[w]; bj = s_idx[w]; }
    s_j = bj;
    const int t = perm[k]; perm[k] = perm[bj]; perm[bj] = t;
  }
  __syncthreads();
  const size_t ii = k, jj = s_j;
  if (jj != ii) {
    for (size_t c = tid; c < ii; c += 1024) { const double t = a[ii * lda + c]; a[ii * lda + c] = a[jj * lda + c]; a[jj * lda + c] = t; }
    for (size_t c = ii + 1 + tid; c < jj; c += 1024) { const double t = a[jj * lda + c]; a[jj * lda + c] = a[c * lda + ii]; a[c * lda + ii] = t; }
    for (size_t c = jj + 1 + tid; c < n; c += 1024) { const double t = a[c * lda + ii]; a[c * lda + ii] = a[c * lda + jj]; a[c * lda + jj] = t; }
    if (tid == 0) { const double t = a[ii * lda + ii]; a[ii * lda + ii] = a[jj * lda + jj]; a[jj * lda + jj] = t; }
  }
  __syncthreads();                                      /* the swaps above are visible to this workgroup's later loads */
  for (size_t i = k + 1 + tid; i < n; i += 1024) v[i] = a[i * lda + k];
  if (tid == 0) *alphainv_out = 1.0 / a[k * lda + k];   /* pcholesky.c:112-113 */
}

/* step k, part 2: m -= v v^T / alpha on the lower triangle (dsyr, cblas/source_syr.h:46-56: m_ic += v_c * (-alphainv * v_i),
   c <= i) and v /= alpha (pcholesky.c:121-124).  One row per blockIdx.y, 256 columns per blockIdx.x. */
__global__ void __launch_bounds__(256)
pchol_update_kernel(double *__restrict__ a, size_t lda, size_t n, size_t k, const double *__restrict__ v,
                    const double *__restrict__ alphainv_in)
{
  const size_t i = k + 1 + blockIdx.y;
  const size_t c = k + 1 + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double alphainv = *alphainv_in;
  const double vi = v[i];
  if (c <= i) {
    const double tmp = -alphainv * vi;
    a[i * lda + c] += v[c] * tmp;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a[i * lda + k] = vi * alphainv;
}

__global__ void pchol_init_perm_kernel(int *perm, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) perm[i] = (int)i;
}

__global__ void __launch_bounds__(256)
tricpy_l2u_kernel(double *__restrict__ a, size_t lda, size_t n)
{
  /* upper(j, i) <- lower(i, j), j < i (matrix/swap_source.c:213); simple form: one element per thread */
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j < i && i < n) a[j * lda + i] = a[i * lda + j];
}

static int pcholesky_decomp_impl(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *d_perm, bool copy_uplo);

extern "C" int gsl_sinterp_hip_pcholesky_decomp(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *d_perm)
{
  return pcholesky_decomp_impl(ctx, n, d_a, lda, d_perm, true);
}

/* pcholesky_decomp(copy_uplo, A, p) of linalg/pcholesky.c:71-154 */
static int pcholesky_decomp_impl(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *d_perm, bool copy_uplo)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, lda >= n && n <= 65535, ST_EINVAL);       /* grid.y = rows of the trailing block */
  REQUIRE(ctx, n == 0 || (d_a && d_perm), ST_EFAULT);
  if (n == 0) return ST_SUCCESS;
  void *buf = NULL;
  int st = sinterp_aux(ctx, (n + 8) * sizeof(double), &buf);
  if (st) return st;
  double *v = (double *)buf, *alphainv = v + n;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (copy_uplo) hipLaunchKernelGGL(tricpy_l2u_kernel, dim3(nb, (unsigned)n), dim3(256), 0, ctx->stream, d_a, lda, n);   /* pcholesky.c:91-95 */
  hipLaunchKernelGGL(pchol_init_perm_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_perm, n);
  for (size_t k = 0; k < n; k++) {
    hipLaunchKernelGGL(pchol_pivot_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_a, lda, n, k, d_perm, v, alphainv);
    const size_t rest = n - k - 1;
    if (rest)
      hipLaunchKernelGGL(pchol_update_kernel, dim3((unsigned)((rest + 255) / 256), (unsigned)rest), dim3(256), 0, ctx->stream, d_a,
                         lda, n, k, (const double *)v, (const double *)alphainv);
  }
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

__global__ void permute_gather_d_kernel(const double *__restrict__ src, const int *__restrict__ perm, double *__restrict__ dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];                     /* gsl_permute_vector: out[i] = in[p[i]] */
}
__global__ void permute_scatter_d_kernel(const double *__restrict__ src, const int *__restrict__ perm, double *__restrict__ dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[perm[i]] = src[i];                     /* gsl_permute_vector_inverse: out[p[i]] = in[i] */
}
__global__ void diag_div_kernel(double *__restrict__ x, const double *__restrict__ ldlt, size_t lda, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = x[i] / ldlt[i * lda + i];           /* gsl_vector_div(x, D), pcholesky.c:218 */
}

extern "C" int gsl_sinterp_hip_pcholesky_svx(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_ldlt, size_t lda, const int *d_perm,
                                             double *d_x)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  EXCLUSIVE_SECTION(ctx);
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_ldlt && d_perm && d_x), ST_EFAULT);
  if (n == 0) return ST_SUCCESS;
  void *d_tmp = NULL;
  int st = sinterp_workspace(ctx, 2 * n * sizeof(double), &d_tmp);
  if (st) return st;
  double *t0 = (double *)d_tmp, *t1 = t0 + n;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(permute_gather_d_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)d_x, d_perm, t0, n);   /* x := P b */
  LAUNCH_CHECK(ctx);
  st = sinterp_trsv(ctx, n, d_ldlt, lda, t0, t1, 0, 1);           /* L w = P b, unit lower */
  if (st) return st;
  hipLaunchKernelGGL(diag_div_kernel, dim3(nb), dim3(256), 0, ctx->stream, t1, d_ldlt, lda, n);                            /* D y = w */
  LAUNCH_CHECK(ctx);
  st = sinterp_trsv(ctx, n, d_ldlt, lda, t1, t0, 1, 1);           /* L^T z = y, unit */
  if (st) return st;
  hipLaunchKernelGGL(permute_scatter_d_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)t0, d_perm, d_x, n);  /* x = P^T z */
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}


/* ------------------------------------------------------------------------ */
/* gsl_linalg_pcholesky_decomp2 / _svx2 / _rcond (linalg/pcholesky.c:231-353, 472-580) */
extern "C" int gsl_sinterp_hip_pcholesky_decomp2(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *d_perm, double *d_s)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, lda >= n && n <= 65535, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_a && d_perm && d_s), ST_EFAULT);
  if (n == 0) return ST_SUCCESS;
  const unsigned nb = (unsigned)((n + 255) / 256);
  /* the UNSCALED matrix goes to the strict upper triangle first (:251-252), then the lower triangle is scaled (:254-262) */
  hipLaunchKernelGGL(tricpy_l2u_kernel, dim3(nb, (unsigned)n), dim3(256), 0, ctx->stream, d_a, lda, n);
  hipLaunchKernelGGL(chol_scale_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)d_a, lda, n, d_s);
  hipLaunchKernelGGL(chol_scale_apply_kernel, dim3(nb, (unsigned)n), dim3(256), 0, ctx->stream, d_a, lda, n, (const double *)d_s);
  LAUNCH_CHECK(ctx);
  return pcholesky_decomp_impl(ctx, n, d_a, lda, d_perm, false);
}

extern "C" int gsl_sinterp_hip_pcholesky_svx2(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_ldlt, size_t lda, const int *d_perm,
                                              const double *d_s, double *d_x)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_ldlt && d_perm && d_s && d_x), ST_EFAULT);
  if (n == 0) return ST_SUCCESS;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(vec_mul_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_x, d_s, n);      /* x := S b */
  LAUNCH_CHECK(ctx);
  int st = gsl_sinterp_hip_pcholesky_svx(ctx, n, d_ldlt, lda, d_perm, d_x);
  if (st) return st;
  hipLaunchKernelGGL(vec_mul_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_x, d_s, n);      /* x = S x~ */
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

/* A_jj of the pivoted matrix rebuilt from L D L^T: D_j + sum_{i<j} D_i L_ji^2 (pcholesky.c:536-551); one workgroup per j */
__global__ void __launch_bounds__(256)
pchol_diag_kernel(const double *__restrict__ ldlt, size_t lda, size_t n, double *__restrict__ out)
{
  __shared__ double s_red[4];
  const size_t j = blockIdx.x;
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < j; i += 256) { const double l = ldlt[j * lda + i]; acc += ldlt[i * lda + i] * l * l; }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[j] = ldlt[j * lda + j] + ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
}

/* column j of |A|: the strict upper triangle holds the original matrix, the diagonal comes from diag[] */
__global__ void __launch_bounds__(256)
pchol_norm1_kernel(const double *__restrict__ ldlt, size_t lda, size_t n, const double *__restrict__ diag, unsigned long long *__restrict__ out)
{
  __shared__ double s_red[4];
  const size_t j = blockIdx.x;
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < j; i += 256) acc += fabs(ldlt[i * lda + j]);
  for (size_t c = j + 1 + threadIdx.x; c < n; c += 256) acc += fabs(ldlt[j * lda + c]);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + fabs(diag[j]);
    atomicMax(out, (unsigned long long)__double_as_longlong(t));
  }
}

extern "C" int gsl_sinterp_hip_pcholesky_rcond(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_ldlt, size_t lda, const int *d_perm,
                                               double *h_rcond)
{
  REQUIRE(ctx, ctx != NULL && h_rcond != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  *h_rcond = 0.0;
  if (n == 0) return ST_SUCCESS;
  REQUIRE(ctx, d_ldlt != NULL && d_perm != NULL, ST_EFAULT);
  void *buf = NULL;
  int st = sinterp_aux(ctx, 2 * n * sizeof(double), &buf);
  if (st) return st;
  double *dp = (double *)buf, *diag = dp + n;
  unsigned long long *d_norm = (unsigned long long *)((char *)ctx->d_scratch + 256);
  const unsigned nb = (unsigned)((n + 255) / 256);
  HIP_OK(ctx, hipMemsetAsync(d_norm, 0, sizeof *d_norm, ctx->stream));
  hipLaunchKernelGGL(pchol_diag_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_ldlt, lda, n, dp);
  hipLaunchKernelGGL(permute_scatter_d_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)dp, d_perm, diag, n);   /* permute_vector_inverse */
  hipLaunchKernelGGL(pchol_norm1_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_ldlt, lda, n, (const double *)diag, d_norm);
  LAUNCH_CHECK(ctx);
  unsigned long long bits = 0;
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(&bits, d_norm, sizeof bits, hipMemcpyDeviceToHost));
  double anorm;
  memcpy(&anorm, &bits, sizeof anorm);
  if (anorm == 0.0) return ST_SUCCESS;                  /* pcholesky.c:497-498 */
  double gamma = 0.0;
  st = invnorm1_device(ctx, n, [&](double *d_v) -> int { return gsl_sinterp_hip_pcholesky_svx(ctx, n, d_ldlt, lda, d_perm, d_v); }, &gamma);
  if (!st && gamma != 0.0) *h_rcond = (1.0 / anorm) / gamma;
  return st;
}
