/*
 * chol.hip -- SPD factorisation and solve for the Gaussian RBF system.
 *
 * Replaces gsl_linalg_cholesky_decomp1 (linalg/cholesky.c:88-131) and
 * gsl_linalg_cholesky_svx (linalg/cholesky.c:163-185).  Same contract: in place,
 * L in the lower triangle (diagonal included), the original matrix kept in the
 * strict upper triangle (cholesky.c:103), GSL_EDOM when a pivot is <= 0
 * (cholesky.c:120-123), columns scaled by 1/sqrt(a_jj) (cholesky.c:125-126).
 *
 * The reference is an unblocked Level-2 gaxpy sweep.  Here the factorisation is
 * a recursive "tall panel" Cholesky:
 *     panel(j0, w):  if w <= 32: base kernel   (diag potrf + trsm of all rows below, fused)
 *                    else: panel(j0, w/2)
 *                          A[j0+w/2:N, j0+w/2:j0+w] -= A[j0+w/2:N, j0:j0+w/2] * A[j0+w/2:j0+w, j0:j0+w/2]^T
 *                                                   (gemm.hip, fp64 MFMA, lower part only on the diagonal tile)
 *                          panel(j0+w/2, w - w/2)
 * so every O(N^3) flop runs in the MFMA GEMM with K = w/2 and the serial part
 * is N/32 single-wave 32x32 factorizations, recomputed redundantly by every
 * workgroup of the base kernel instead of being a launch of their own.
 */
#include "common.h"
#include <math.h>

#define CB 32 /* base panel width */

__device__ __forceinline__ double lane_bcast(double v, int src)
{
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

/* ------------------------------------------------------------------------ */
/* base: factor the nb x nb diagonal block (nb <= 32) and solve the rows below */
__global__ void __launch_bounds__(256)
chol_base_kernel(double *__restrict__ A, size_t lda, size_t n, size_t j0, int nb, int *__restrict__ info,
                 double *__restrict__ diag_store)
{
  __shared__ double sL[CB][CB + 1];
  __shared__ double sInv[CB];
  const int tid = threadIdx.x;

  if (tid < 64) {                                  /* one wave factors the diagonal block */
    const int lane = tid;
    double a[CB];
#pragma unroll
    for (int k = 0; k < CB; k++) a[k] = (lane < nb && k <= lane && k < nb) ? A[(j0 + lane) * lda + j0 + k] : 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < CB; j++) {
      double v = a[j];
#pragma unroll
      for (int k = 0; k < j; k++) v = fma(-a[k], lane_bcast(a[k], j), v);
      double d = lane_bcast(v, j);
      if (j < nb && !(d > 0.0)) {                   /* cholesky.c:120-123 */
        if (lane == 0 && blockIdx.x == 0) atomicCAS(info, 0, (int)(j0 + j + 1));
        bad = true;
        d = 1.0;
      }
      if (j >= nb) d = 1.0;
      const double sd = sqrt(d);
      const double inv = 1.0 / sd;
      a[j] = (lane == j) ? sd : v * inv;
      if (lane == 0) sInv[j] = inv;
    }
    (void)bad;
    if (lane < CB) {
#pragma unroll
      for (int k = 0; k < CB; k++) sL[lane][k] = (k <= lane) ? a[k] : 0.0;
      /* The factored diagonal block is NOT written into A here: workgroups of this
         launch that start late must still read the unfactored block.  It goes to a
         side buffer and is copied into A after the last panel (nothing in the
         factorisation reads a diagonal block of L again). */
      if (blockIdx.x == 0) {
        double *dst = diag_store + (j0 / CB) * (CB * CB) + lane * CB;
#pragma unroll
        for (int k = 0; k < CB; k++) dst[k] = (k <= lane) ? a[k] : 0.0;
      }
    }
  }
  __syncthreads();

  /* rows below the diagonal block: x L^T = b, one row per thread, row in registers */
  const size_t row = j0 + nb + (size_t)blockIdx.x * blockDim.x + tid;
  if (row >= n) return;
  double *p = A + row * lda + j0;
  double x[CB];
  if (nb == CB && ((((uintptr_t)p) & 15) == 0)) {
#pragma unroll
    for (int k = 0; k < CB; k += 2) {
      double2 t = *reinterpret_cast<const double2 *>(p + k);
      x[k] = t.x; x[k + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < CB; k++) x[k] = k < nb ? p[k] : 0.0;
  }
#pragma unroll
  for (int j = 0; j < CB; j++) {
    double v = x[j];
#pragma unroll
    for (int k = 0; k < j; k++) v = fma(-x[k], sL[j][k], v);
    x[j] = v * sInv[j];
  }
  if (nb == CB && ((((uintptr_t)p) & 15) == 0)) {
#pragma unroll
    for (int k = 0; k < CB; k += 2) *reinterpret_cast<double2 *>(p + k) = make_double2(x[k], x[k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < CB; k++) if (k < nb) p[k] = x[k];
  }
}

__global__ void __launch_bounds__(256)
chol_diag_writeback_kernel(double *__restrict__ A, size_t lda, size_t n, const double *__restrict__ diag_store)
{
  const size_t j0 = (size_t)blockIdx.x * CB;
  for (int e = threadIdx.x; e < CB * CB; e += 256) {
    const int r = e / CB, k = e % CB;
    if (k <= r && j0 + r < n) A[(j0 + r) * lda + j0 + k] = diag_store[(size_t)blockIdx.x * (CB * CB) + e];
  }
}

/* upper(i<j) <- lower(j,i)  (matrix/swap_source.c:213, cholesky.c:103) */
__global__ void __launch_bounds__(256)
tricpy_lower_to_upper_kernel(double *__restrict__ A, size_t lda, size_t n)
{
  __shared__ double tile[32][33];
  const size_t bi = blockIdx.y, bj = blockIdx.x;   /* source tile (bi,bj) with bi >= bj */
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const size_t i = bi * 32 + r, j = bj * 32 + tx;
    tile[r][tx] = (i < n && j < n) ? A[i * lda + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const size_t j = bj * 32 + r, i = bi * 32 + tx;  /* destination element (j, i), j < i */
    if (i < n && j < n && j < i) A[j * lda + i] = tile[tx][r];
  }
}

static int chol_panel(gsl_sinterp_hip_ctx *ctx, double *A, size_t lda, size_t n, size_t j0, size_t w, int *d_info,
                      double *d_diag)
{
  if (w <= CB) {
    const size_t below = n - j0 - w;
    const unsigned grid = (unsigned)((below + 255) / 256) + (below == 0 ? 1u : 0u);
    hipLaunchKernelGGL(chol_base_kernel, dim3(grid ? grid : 1), dim3(256), 0, ctx->stream, A, lda, n, j0, (int)w, d_info, d_diag);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  size_t w1 = ((w / 2 + CB - 1) / CB) * CB;
  if (w1 >= w) w1 = w - CB;
  int st = chol_panel(ctx, A, lda, n, j0, w1, d_info, d_diag);
  if (st) return st;
  const size_t r0 = j0 + w1;
  st = sinterp_gemm_minus(ctx, n - r0, w - w1, w1, A + r0 * lda + j0, lda, A + r0 * lda + j0, lda, 0,
                          A + r0 * lda + r0, lda, 1);
  if (st) return st;
  return chol_panel(ctx, A, lda, n, r0, w - w1, d_info, d_diag);
}

extern "C" int gsl_sinterp_hip_cholesky_decomp1(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *h_info)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || d_a, ST_EFAULT);
  if (h_info) *h_info = 0;
  if (n == 0) return ST_SUCCESS;
  int *d_info = (int *)ctx->d_scratch + 8;
  HIP_OK(ctx, hipMemsetAsync(d_info, 0, sizeof(int), ctx->stream));
  const unsigned nt = (unsigned)((n + 31) / 32);
  hipLaunchKernelGGL(tricpy_lower_to_upper_kernel, dim3(nt, nt), dim3(256), 0, ctx->stream, d_a, lda, n);
  LAUNCH_CHECK(ctx);
  const size_t nblk = (n + CB - 1) / CB;
  void *d_diag = NULL;
  int st = sinterp_workspace(ctx, nblk * CB * CB * sizeof(double), &d_diag);
  if (st) return st;
  st = chol_panel(ctx, d_a, lda, n, 0, n, d_info, (double *)d_diag);
  if (st) return st;
  hipLaunchKernelGGL(chol_diag_writeback_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, d_a, lda, n,
                     (const double *)d_diag);
  LAUNCH_CHECK(ctx);
  int info = 0;
  HIP_OK(ctx, hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  if (h_info) *h_info = info;
  if (info) return sinterp_fail(ctx, ST_EDOM, "cholesky_decomp1: matrix is not positive definite", hipSuccess, __FILE__, __LINE__);
  return ST_SUCCESS;
}

/* ------------------------------------------------------------------------ */
/* blocked triangular sweeps shared with lu.hip.  One launch per 64-wide block
   column J: every workgroup solves the 64x64 diagonal system redundantly in
   its first wave, then updates its slice of the remaining right-hand side.
     mode 0: Lower, NoTrans, forward   (source_trsv_r.h:56-79)    rows i > J:  b_i -= sum_j T[i][j] x_j
     mode 1: Lower, Trans,  backward   (source_trsv_r.h:106-129)  cols i < J:  b_i -= sum_j T[j][i] x_j
     mode 2: Upper, NoTrans, backward  (source_trsv_r.h:33-55)    rows i < J:  b_i -= sum_j T[i][j] x_j   */
#define TS 64

__global__ void __launch_bounds__(256)
trsv_sweep_kernel(const double *__restrict__ T, size_t ldt, size_t n, double *__restrict__ b,
                  double *__restrict__ xout, size_t j0, int nb, int mode, int unit)
{
  __shared__ double sx[TS];
  __shared__ double sD[TS][TS + 1];
  const int tid = threadIdx.x;
  /* diagonal block into LDS (as the matrix of the small system to solve) */
  for (int e = tid; e < TS * TS; e += 256) {
    const int r = e / TS, c = e % TS;
    double v = 0.0;
    if (r < nb && c < nb) v = T[(j0 + r) * ldt + j0 + c];
    sD[r][c] = v;
  }
  __syncthreads();
  if (tid < 64) {
    const int lane = tid;
    double bi = lane < nb ? b[j0 + lane] : 0.0;
    if (mode == 0) {            /* forward with D = lower(sD) */
      for (int j = 0; j < nb; j++) {
        double xj = lane_bcast(bi, j);
        if (!unit) xj = xj / sD[j][j];
        if (lane == j) bi = xj;
        if (lane > j) bi -= sD[lane][j] * xj;
      }
    } else if (mode == 1) {     /* backward with D^T, D lower: x_j then b_i -= D[j][i] x_j for i < j */
      for (int j = nb - 1; j >= 0; j--) {
        double xj = lane_bcast(bi, j);
        if (!unit) xj = xj / sD[j][j];
        if (lane == j) bi = xj;
        if (lane < j) bi -= sD[j][lane] * xj;
      }
    } else {                    /* backward with D upper */
      for (int j = nb - 1; j >= 0; j--) {
        double xj = lane_bcast(bi, j);
        if (!unit) xj = xj / sD[j][j];
        if (lane == j) bi = xj;
        if (lane < j) bi -= sD[lane][j] * xj;
      }
    }
    /* solved block goes to xout, never back into b: late workgroups of this launch re-read b[J] */
    if (lane < nb) { sx[lane] = bi; if (blockIdx.x == 0) xout[j0 + lane] = bi; }
  }
  __syncthreads();

  /* remaining right-hand side */
  size_t i;
  if (mode == 0) { i = j0 + nb + (size_t)blockIdx.x * 256 + tid; if (i >= n) return; }
  else { i = (size_t)blockIdx.x * 256 + tid; if (i >= j0) return; }
  double acc = 0.0;
  if (mode == 1) {
    for (int j = 0; j < nb; j++) acc = fma(T[(j0 + j) * ldt + i], sx[j], acc);   /* coalesced across lanes */
  } else {
    const double *row = T + i * ldt + j0;
    for (int j = 0; j < nb; j++) acc = fma(row[j], sx[j], acc);
  }
  b[i] -= acc;
}

int sinterp_trsv(gsl_sinterp_hip_ctx *ctx, size_t n, const double *T, size_t ldt, double *b, double *xout, int mode,
                 int unit)
{
  if (n == 0) return ST_SUCCESS;
  const size_t nblk = (n + TS - 1) / TS;
  for (size_t t = 0; t < nblk; t++) {
    const size_t blk = (mode == 0) ? t : nblk - 1 - t;
    const size_t j0 = blk * TS;
    const int nb = (int)((n - j0) < TS ? (n - j0) : TS);
    const size_t rest = (mode == 0) ? n - j0 - nb : j0;
    const unsigned grid = (unsigned)((rest + 255) / 256);
    hipLaunchKernelGGL(trsv_sweep_kernel, dim3(grid ? grid : 1), dim3(256), 0, ctx->stream, T, ldt, n, b, xout, j0, nb, mode, unit);
    LAUNCH_CHECK(ctx);
  }
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_cholesky_svx(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_llt, size_t lda, double *d_x)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_llt && d_x), ST_EFAULT);
  void *d_tmp = NULL;
  int st = sinterp_workspace(ctx, n * sizeof(double), &d_tmp);
  if (st) return st;
  st = sinterp_trsv(ctx, n, d_llt, lda, d_x, (double *)d_tmp, 0, 0);   /* L c = b     (cholesky.c:178) */
  if (st) return st;
  return sinterp_trsv(ctx, n, d_llt, lda, (double *)d_tmp, d_x, 1, 0); /* L^T x = c   (cholesky.c:181) */
}
