/*
 * solve.hip -- "init" of an RBF interpolant on the device: Phi fill + dense solve
 * Phi w = f, choosing the route by kernel class:
 *
 *   route 1  Gaussian (SPD):  Cholesky + two triangular sweeps
 *            = gsl_linalg_cholesky_decomp1 + _svx (linalg/cholesky.c:88,163).
 *   route 2  thin-plate spline (symmetric, indefinite, conditionally positive definite of
 *            order 2): the reference route would be gsl_linalg_LU_decomp + _svx
 *            (linalg/lu.c:59,166), whose partial pivoting is an inherently serial N-step
 *            reduction chain.  Phi is positive definite on the complement of the d+1
 *            polynomials P = [1, (x-mean)/std], so  B = Phi + s P P^T  with s = c |Phi|_inf / N
 *            is SPD (c = 4; threshold measured between 1 and 2) and
 *                Phi^-1 f = B^-1 f + B^-1 P (I/s - P^T B^-1 P)^-1 P^T B^-1 f      (Woodbury)
 *            i.e. one MFMA Cholesky, d+2 right-hand sides through the blocked sweeps and a
 *            (d+1)x(d+1) system on the host.  Interpolated values agree with the LU route to
 *            ~1e-13 relative (weights to ~1e-9, the cond(Phi)*eps level at which any two
 *            backward-stable solvers differ -- SURVEY.md section 7).
 *   route 3  if B is not SPD even with c = 32: pivoted LU (lu.hip), the reference route.
 */
#include "common.h"
#include <math.h>
#include <stdlib.h>

#define SV_MAXK 4 /* dim + 1 <= 4 polynomial columns */

/* statistics + polynomial block: one workgroup.  Y holds the right-hand sides
   (vector 0 = f, vectors 1..k = P columns), Pk keeps a copy of P. */
__global__ void __launch_bounds__(1024)
poly_block_kernel(const double *__restrict__ x, size_t n, int dim, size_t xtda, const double *__restrict__ f,
                  double *__restrict__ Y, double *__restrict__ Pk, double *__restrict__ stats)
{
  __shared__ double s_red[2 * 3][16];
  __shared__ double s_mean[3], s_inv[3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double sum[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
  for (size_t i = tid; i < n; i += 1024)
    for (int c = 0; c < dim; c++) { const double v = x[i * xtda + c]; sum[c] += v; sq[c] = fma(v, v, sq[c]); }
  for (int c = 0; c < 3; c++) {
    double a = sum[c], b = sq[c];
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if (lane == 0) { s_red[2 * c][wave] = a; s_red[2 * c + 1][wave] = b; }
  }
  __syncthreads();
  if (tid < 3) {
    double a = 0, b = 0;
    for (int w = 0; w < 16; w++) { a += s_red[2 * tid][w]; b += s_red[2 * tid + 1][w]; }
    const double mean = a / (double)n;
    double var = b / (double)n - mean * mean;
    if (!(var > 0)) var = 1.0;
    s_mean[tid] = mean;
    s_inv[tid] = 1.0 / sqrt(var);
    if (stats) { stats[tid] = mean; stats[3 + tid] = s_inv[tid]; }     /* the affine tail is reported in raw coordinates */
  }
  __syncthreads();
  for (size_t i = tid; i < n; i += 1024) {
    Y[i] = f[i];
    Y[n + i] = 1.0; Pk[i] = 1.0;
    for (int c = 0; c < dim; c++) {
      const double v = (x[i * xtda + c] - s_mean[c]) * s_inv[c];
      Y[(size_t)(c + 2) * n + i] = v;
      Pk[(size_t)(c + 1) * n + i] = v;
    }
  }
}

/* |Phi|_inf = max_i sum_j |Phi_ij| : one workgroup per row, max through the bit pattern */
__global__ void __launch_bounds__(256)
row_norm_kernel(const double *__restrict__ phi, size_t lda, size_t n, unsigned long long *__restrict__ out)
{
  __shared__ double s_red[4];
  const size_t i = blockIdx.x;
  double a = 0;
  for (size_t j = threadIdx.x; j < n; j += 256) a += fabs(phi[i * lda + j]);
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    atomicMax(out, (unsigned long long)__double_as_longlong(t));     /* t >= 0: bit order == value order */
  }
}

/* Phi_ij += (c |Phi|_inf / n) * sum_a P_a[i] P_a[j] */
__global__ void __launch_bounds__(256)
poly_shift_kernel(double *__restrict__ phi, size_t lda, size_t n, const double *__restrict__ Pk, int k, double cmul,
                  const unsigned long long *__restrict__ norm_bits)
{
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t i = blockIdx.y;
  if (j >= n) return;
  const double s = cmul * __longlong_as_double((long long)*norm_bits) / (double)n;
  double acc = 0.0;
  for (int a = 0; a < k; a++) acc = fma(Pk[(size_t)a * n + i], Pk[(size_t)a * n + j], acc);
  phi[i * lda + j] = fma(s, acc, phi[i * lda + j]);
}

/* G[a][b] = sum_i P_a[i] Y_b[i]  (a < k, b < k+1): one workgroup per entry */
__global__ void __launch_bounds__(256)
gram_kernel(const double *__restrict__ Pk, const double *__restrict__ Y, size_t n, int k, double *__restrict__ G)
{
  __shared__ double s_red[4];
  const int a = blockIdx.x / (k + 1), b = blockIdx.x % (k + 1);
  double acc = 0;
  for (size_t i = threadIdx.x; i < n; i += 256) acc = fma(Pk[(size_t)a * n + i], Y[(size_t)b * n + i], acc);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) G[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ void __launch_bounds__(256)
combine_kernel(const double *__restrict__ Y, size_t n, int k, const double *__restrict__ coef, double *__restrict__ w)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v = Y[i];
  for (int a = 0; a < k; a++) v = fma(coef[a], Y[(size_t)(a + 1) * n + i], v);
  w[i] = v;
}

/* tiny dense solve on the host (partial pivoting) */
static int host_solve(int k, double *S, double *g)
{
  for (int j = 0; j < k; j++) {
    int p = j;
    for (int i = j + 1; i < k; i++) if (fabs(S[i * k + j]) > fabs(S[p * k + j])) p = i;
    if (S[p * k + j] == 0.0) return 1;
    if (p != j) { for (int c = 0; c < k; c++) { double t = S[j * k + c]; S[j * k + c] = S[p * k + c]; S[p * k + c] = t; } double t = g[j]; g[j] = g[p]; g[p] = t; }
    for (int i = j + 1; i < k; i++) {
      const double l = S[i * k + j] / S[j * k + j];
      for (int c = j; c < k; c++) S[i * k + c] -= l * S[j * k + c];
      g[i] -= l * g[j];
    }
  }
  for (int i = k - 1; i >= 0; i--) {
    double t = g[i];
    for (int c = i + 1; c < k; c++) t -= S[i * k + c] * g[c];
    g[i] = t / S[i * k + i];
  }
  return 0;
}

static int rbf_solve_impl(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim, size_t xtda,
                          double *d_phi, size_t lda, double *d_w, int *h_route, bool keep_upper, double *h_poly = NULL);

extern "C" int gsl_sinterp_hip_rbf_solve(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n,
                                         int dim, size_t xtda, double *d_phi, size_t lda, double *d_w, int *h_route)
{
  /* d_phi is scratch here: the Gaussian route fills and factors the lower triangle only */
  return rbf_solve_impl(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda, d_w, h_route, false);
}

/* h_poly != NULL (thin-plate spline only): the affine-augmented system
       [Phi P; P^T 0] [w; c] = [f; 0],   P = [1, x],      s(y) = sum_j w_j phi(|y - x_j|) + c_0 + sum_a c_a y_a
   -- the standard thin-plate spline ("N + d + 1 with affine augmentation").  P^T w = 0
   makes Phi w = (Phi + s P P^T) w = B w, so with the SAME shifted SPD matrix B and the same d + 2 right-hand sides
       w = B^-1 f - B^-1 P c,     (P^T B^-1 P) c = P^T B^-1 f :
   one MFMA Cholesky, the blocked sweeps, a (d+1) x (d+1) system on the host (routes 9 / 10 mirror 2 / 3).  h_poly
   receives c in RAW coordinates (the solve uses the standardised columns of poly_block_kernel). */
static int rbf_solve_affine_lu(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim, size_t xtda,
                               double *d_phi, size_t lda, double *d_w, double *h_poly);

static int rbf_solve_impl(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim, size_t xtda,
                          double *d_phi, size_t lda, double *d_w, int *h_route, bool keep_upper, double *h_poly)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  EXCLUSIVE_SECTION(ctx);
  REQUIRE(ctx, dim >= 1 && dim <= 3 && xtda >= (size_t)dim && lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_x && d_phi && d_w), ST_EFAULT);
  if (h_route) *h_route = 0;
  if (n == 0) return ST_SUCCESS;
  /* thin-plate spline: the shift and the row norms need the full matrix; Gaussian: the Cholesky reads the lower
     triangle only, the upper one (the "original kept above the diagonal" of cholesky.c:103) only when asked for */
  const bool spd = kind != GSL_SINTERP_RBF_TPS;        /* Gaussian, Wendland: positive definite kernels */
  static const bool no_fused = getenv("GSL_SINTERP_NO_FUSED_SHIFT") && getenv("GSL_SINTERP_NO_FUSED_SHIFT")[0] == '1';
  const bool force_lu = getenv("GSL_SINTERP_FORCE_LU") && getenv("GSL_SINTERP_FORCE_LU")[0] == '1';
  /* thin-plate spline on the SPD route: the matrix is written once, shifted (sinterp_tps_fill_shifted, below) */
  const bool fused = !spd && !no_fused && !force_lu;
  int st = fused ? ST_SUCCESS : sinterp_rbf_fill_ex(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda, spd && !keep_upper);
  if (st) return st;
  int info = 0;

  if (spd) {
    /* the fill (and the shift) write both triangles; the forward substitution rides along with the factorisation */
    st = sinterp_cholesky_factor_solve_sym(ctx, n, d_phi, lda, &info, d_w, n, 1);
    if (st) return st;
    if (h_route) *h_route = 1;
    return ST_SUCCESS;
  }

  /* ---- conditionally positive definite kernel: shifted SPD system + Woodbury */
  const int k = dim + 1;
  void *aux = NULL;
  st = sinterp_aux(ctx, ((size_t)(2 * k + 1) * n + 64) * sizeof(double), &aux);
  if (st) return st;
  double *Y = (double *)aux;                      /* (k+1) x n : f, P columns -> B^-1 [f P] */
  double *Pk = Y + (size_t)(k + 1) * n;           /* k x n     : P                          */
  double *G = Pk + (size_t)k * n;                 /* k x (k+1) Gram block, then coefficients */
  unsigned long long *d_norm = (unsigned long long *)(G + 32);

  double cmul = 4.0;
  for (int attempt = 0; attempt < 2 && !force_lu; attempt++, cmul *= 8.0) {
    if (attempt > 0 && !fused) {
      st = gsl_sinterp_hip_rbf_fill(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda);
      if (st) return st;
    }
    hipLaunchKernelGGL(poly_block_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_x, n, dim, xtda, (const double *)d_w, Y, Pk, G + 24);
    if (fused) {
      /* |Phi|_inf from the coordinates, then Phi + s P P^T written in one pass */
      st = sinterp_tps_fill_shifted(ctx, d_x, n, dim, xtda, d_phi, lda, Pk, k, cmul, d_norm);
      if (st) return st;
    } else {
      HIP_OK(ctx, hipMemsetAsync(d_norm, 0, sizeof(unsigned long long), ctx->stream));
      hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, (const double *)d_phi, lda, n, d_norm);
      hipLaunchKernelGGL(poly_shift_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, ctx->stream, d_phi, lda, n,
                         (const double *)Pk, k, cmul, (const unsigned long long *)d_norm);
    }
    LAUNCH_CHECK(ctx);
    st = sinterp_cholesky_factor_solve_sym(ctx, n, d_phi, lda, &info, Y, n, k + 1);   /* the fill (and the shift) write both triangles */
    if (st == ST_EDOM) continue;                  /* not SPD with this shift: larger shift (Y is rebuilt), then LU */
    if (st) return st;
    hipLaunchKernelGGL(gram_kernel, dim3((unsigned)(k * (k + 1))), dim3(256), 0, ctx->stream, (const double *)Pk, (const double *)Y,
                       n, k, G);
    LAUNCH_CHECK(ctx);
    /* one read-back: the Gram block, the column statistics (G + 24) and the norm (G + 32) are neighbours */
    double hbuf[40];
    unsigned long long hnorm = 0;
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    HIP_OK(ctx, hipMemcpy(hbuf, G, sizeof(double) * 33, hipMemcpyDeviceToHost));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    const double *hG = hbuf;
    memcpy(&hnorm, &hbuf[32], sizeof hnorm);
    double nrm;
    memcpy(&nrm, &hnorm, sizeof nrm);
    const double s = cmul * nrm / (double)n;
    if (h_poly) {
      /* (P^T B^-1 P) c = P^T B^-1 f, w = B^-1 f - B^-1 P c */
      double M[SV_MAXK * SV_MAXK], g[SV_MAXK];
      for (int a = 0; a < k; a++) {
        g[a] = hG[a * (k + 1)];
        for (int b = 0; b < k; b++) M[a * k + b] = hG[a * (k + 1) + 1 + b];
      }
      if (host_solve(k, M, g)) break;             /* degenerate (collinear centres): the reference route reports it */
      double neg[SV_MAXK];
      for (int a = 0; a < k; a++) neg[a] = -g[a];
      HIP_OK(ctx, hipMemcpyAsync(G, neg, sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)Y, n, k,
                         (const double *)G, d_w);
      LAUNCH_CHECK(ctx);
      HIP_OK(ctx, hipStreamSynchronize(ctx->stream));   /* neg[] is a stack buffer */
      /* standardised columns (x - mean) inv  ->  raw coordinates */
      const double *mean = hbuf + 24, *inv = hbuf + 27;
      h_poly[0] = g[0];
      for (int a = 0; a < dim; a++) { h_poly[1 + a] = g[1 + a] * inv[a]; h_poly[0] -= g[1 + a] * inv[a] * mean[a]; }
      if (h_route) *h_route = 9;
      return ST_SUCCESS;
    }
    /* S c = g with S = I/s - P^T B^-1 P, g = P^T B^-1 f   (G column 0 = g, columns 1..k = P^T B^-1 P) */
    double S[SV_MAXK * SV_MAXK], g[SV_MAXK];
    for (int a = 0; a < k; a++) {
      g[a] = hG[a * (k + 1)];
      for (int b = 0; b < k; b++) S[a * k + b] = (a == b ? 1.0 / s : 0.0) - hG[a * (k + 1) + 1 + b];
    }
    if (host_solve(k, S, g)) break;               /* degenerate correction system: use LU */
    HIP_OK(ctx, hipMemcpyAsync(G, g, sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)Y, n, k,
                       (const double *)G, d_w);
    LAUNCH_CHECK(ctx);
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));   /* g[] is a stack buffer */
    if (h_route) *h_route = 2;
    return ST_SUCCESS;
  }

  /* ---- reference route: pivoted LU */
  if (h_poly) {
    st = rbf_solve_affine_lu(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda, d_w, h_poly);
    if (!st && h_route) *h_route = 10;
    return st;
  }
  st = gsl_sinterp_hip_rbf_fill(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda);
  if (st) return st;
  int *d_perm = (int *)Pk;                         /* n ints fit in the k*n doubles of Pk */
  int signum = 0;
  st = gsl_sinterp_hip_lu_decomp(ctx, n, d_phi, lda, d_perm, &signum);
  if (st) return st;
  if (h_route) *h_route = 3;
  return gsl_sinterp_hip_lu_svx(ctx, n, d_phi, lda, d_perm, d_w);
}


/* border of the augmented matrix: rows / columns n .. n+k-1 = [1, x] (raw coordinates), zero corner; rhs tail = 0 */
__global__ void __launch_bounds__(256)
affine_border_kernel(double *__restrict__ phi, size_t lda, size_t n, const double *__restrict__ x, int dim, size_t xtda,
                     double *__restrict__ rhs)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int k = dim + 1;
  if (i < n) {
    for (int a = 0; a < k; a++) {
      const double v = a == 0 ? 1.0 : x[i * xtda + a - 1];
      phi[i * lda + n + a] = v;
      phi[(n + a) * lda + i] = v;
    }
  }
  if (i < (size_t)k) {
    for (int a = 0; a < k; a++) phi[(n + i) * lda + n + a] = 0.0;
    rhs[n + i] = 0.0;
  }
}

/* the reference route of the augmented system: gsl_linalg_LU_decomp + _svx (linalg/lu.c:59-201) of the (n + d + 1) matrix */
static int rbf_solve_affine_lu(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim, size_t xtda,
                               double *d_phi, size_t lda, double *d_w, double *h_poly)
{
  const int k = dim + 1;
  const size_t na = n + (size_t)k;
  if (lda < na) return sinterp_fail(ctx, ST_EINVAL, "rbf_solve_affine: the pivoted-LU route needs d_phi with n + dim + 1 rows and lda >= n + dim + 1", hipSuccess, __FILE__, __LINE__);
  int st = gsl_sinterp_hip_rbf_fill(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda);
  if (st) return st;
  double *d_rhs = NULL;
  int *d_perm = NULL;
  st = gsl_sinterp_hip_malloc(ctx, (void **)&d_rhs, na * sizeof(double));
  if (!st) st = gsl_sinterp_hip_malloc(ctx, (void **)&d_perm, na * sizeof(int));
  if (!st) {
    hipError_t e = hipMemcpyAsync(d_rhs, d_w, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) st = sinterp_fail(ctx, ST_EFAILED, "rbf_solve_affine: copy", e, __FILE__, __LINE__);
  }
  if (!st) {
    hipLaunchKernelGGL(affine_border_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_phi, lda, n, d_x, dim, xtda, d_rhs);
    if (hipGetLastError() != hipSuccess) st = sinterp_fail(ctx, ST_EFAILED, "rbf_solve_affine: border", hipSuccess, __FILE__, __LINE__);
  }
  int signum = 0;
  if (!st) st = gsl_sinterp_hip_lu_decomp(ctx, na, d_phi, lda, d_perm, &signum);
  if (!st) st = gsl_sinterp_hip_lu_svx(ctx, na, d_phi, lda, d_perm, d_rhs);
  if (!st) {
    hipError_t e = hipMemcpyAsync(d_w, d_rhs, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(h_poly, d_rhs + n, (size_t)k * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) st = sinterp_fail(ctx, ST_EFAILED, "rbf_solve_affine: read back", e, __FILE__, __LINE__);
  }
  gsl_sinterp_hip_free(ctx, d_rhs); gsl_sinterp_hip_free(ctx, d_perm);
  return st;
}

extern "C" int gsl_sinterp_hip_rbf_solve_affine(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim,
                                                size_t xtda, double *d_phi, size_t lda, double *d_w, double *h_poly, int *h_route)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  REQUIRE(ctx, kind == GSL_SINTERP_RBF_TPS, ST_EINVAL);      /* conditionally positive definite of order 2: the kernel the tail belongs to */
  REQUIRE(ctx, h_poly != NULL, ST_EFAULT);
  REQUIRE(ctx, n >= (size_t)dim + 1, ST_EINVAL);
  for (int a = 0; a < SV_MAXK; a++) h_poly[a] = 0.0;
  return rbf_solve_impl(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda, d_w, h_route, false, h_poly);
}

/* s[k] += c_0 + sum_a c_a y[k][a] */
__global__ void __launch_bounds__(256)
add_poly_kernel(double *__restrict__ s, size_t m, const double *__restrict__ y, size_t ytda, int dim, double c0, double c1, double c2,
                double c3)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
    double t = fma(c1, y[k * ytda], c0);
    if (dim > 1) t = fma(c2, y[k * ytda + 1], t);
    if (dim > 2) t = fma(c3, y[k * ytda + 2], t);
    s[k] = s[k] + t;
  }
}

extern "C" int gsl_sinterp_hip_rbf_eval_affine(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *h_poly, const double *d_x,
                                               size_t n, int dim, size_t xtda, const double *d_w, const double *d_y, size_t m,
                                               size_t ytda, double *d_s, unsigned long long model_id)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  REQUIRE(ctx, h_poly != NULL, ST_EFAULT);
  int st = gsl_sinterp_hip_rbf_eval_model(ctx, kind, eps, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, model_id);
  if (st || m == 0) return st;
  size_t blocks = (m + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_poly_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_s, m, d_y, ytda, dim, h_poly[0], h_poly[1],
                     dim > 1 ? h_poly[2] : 0.0, dim > 2 ? h_poly[3] : 0.0);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}


/* explicit solver choice + optional condition estimate (include/gsl_sinterp_hip.h) */
extern "C" int gsl_sinterp_hip_rbf_solve_ex(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim,
                                            size_t xtda, double *d_phi, size_t lda, double *d_w, int solver, double *h_rcond,
                                            int *h_route)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  EXCLUSIVE_SECTION(ctx);
  REQUIRE(ctx, solver >= GSL_SINTERP_SOLVER_DEFAULT && solver <= GSL_SINTERP_SOLVER_LU_REFINE, ST_EINVAL);
  if (h_rcond) *h_rcond = NAN;
  if (h_route) *h_route = 0;
  if (solver == GSL_SINTERP_SOLVER_DEFAULT) {
    if (!h_rcond || kind == GSL_SINTERP_RBF_TPS)
      return gsl_sinterp_hip_rbf_solve(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda, d_w, h_route);
    int route = 0;
    int st = rbf_solve_impl(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda, d_w, &route, true);
    if (h_route) *h_route = route;
    if (st || route != 1) return st;
    return gsl_sinterp_hip_cholesky_rcond(ctx, n, d_phi, lda, h_rcond);     /* d_phi holds L + the original above the diagonal */
  }
  REQUIRE(ctx, dim >= 1 && dim <= 3 && xtda >= (size_t)dim && lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_x && d_phi && d_w), ST_EFAULT);
  if (n == 0) return ST_SUCCESS;
  int st = gsl_sinterp_hip_rbf_fill(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda);
  if (st) return st;
  if (solver == GSL_SINTERP_SOLVER_CHOLESKY2) {
    void *aux = NULL;
    st = sinterp_aux(ctx, n * sizeof(double), &aux);
    if (st) return st;
    int info = 0;
    st = gsl_sinterp_hip_cholesky_decomp2(ctx, n, d_phi, lda, (double *)aux, &info);
    if (st) return st;
    if (h_route) *h_route = 4;
    if (h_rcond) { st = gsl_sinterp_hip_cholesky_rcond(ctx, n, d_phi, lda, h_rcond); if (st) return st; }
    return gsl_sinterp_hip_cholesky_svx2(ctx, n, d_phi, lda, (const double *)aux, d_w);
  }
  if (solver == GSL_SINTERP_SOLVER_PCHOLESKY) {
    int *d_perm = NULL;
    st = gsl_sinterp_hip_malloc(ctx, (void **)&d_perm, n * sizeof(int));
    if (!st) st = gsl_sinterp_hip_pcholesky_decomp(ctx, n, d_phi, lda, d_perm);
    if (!st) st = gsl_sinterp_hip_pcholesky_svx(ctx, n, d_phi, lda, d_perm, d_w);
    if (!st) st = gsl_sinterp_hip_sync(ctx);
    gsl_sinterp_hip_free(ctx, d_perm);
    if (!st && h_route) *h_route = 5;
    return st;
  }
  /* LU + one refinement step: A (copy), LU (in d_phi), b (copy of the right-hand side) */
  double *d_copy = NULL, *d_b = NULL, *d_work = NULL;
  int *d_perm = NULL, signum = 0;
  st = gsl_sinterp_hip_malloc(ctx, (void **)&d_copy, n * n * sizeof(double));
  if (!st) st = gsl_sinterp_hip_malloc(ctx, (void **)&d_b, 2 * n * sizeof(double));
  if (!st) st = gsl_sinterp_hip_malloc(ctx, (void **)&d_perm, n * sizeof(int));
  if (!st) {
    d_work = d_b + n;
    hipError_t e = hipMemcpy2DAsync(d_copy, n * sizeof(double), d_phi, lda * sizeof(double), n * sizeof(double), n,
                                    hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_b, d_w, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) st = sinterp_fail(ctx, ST_EFAILED, "rbf_solve_ex: copies", e, __FILE__, __LINE__);
  }
  if (!st) st = gsl_sinterp_hip_lu_decomp(ctx, n, d_phi, lda, d_perm, &signum);
  if (!st) st = gsl_sinterp_hip_lu_svx(ctx, n, d_phi, lda, d_perm, d_w);
  if (!st) st = gsl_sinterp_hip_lu_refine(ctx, n, d_copy, n, d_phi, lda, d_perm, d_b, d_w, d_work);
  if (!st) st = gsl_sinterp_hip_sync(ctx);
  gsl_sinterp_hip_free(ctx, d_copy); gsl_sinterp_hip_free(ctx, d_b); gsl_sinterp_hip_free(ctx, d_perm);
  if (!st && h_route) *h_route = 6;
  return st;
}


/* ------------------------------------------------------------------------ */
/* Ordinary kriging (the reference's README:24 lists kriging as future work): the dual form on the covariance
   C(h) = phi(h) with a nugget,
       [K 1; 1^T 0] [w; mu] = [f; 0],  K = Phi + nugget I,      s(y) = mu + sum_j w_j phi(|y - x_j|),
   solved as  a = K^-1 f, b = K^-1 1, mu = (1^T a) / (1^T b), w = a - mu b : ONE factorisation, two right-hand sides
   through the blocked sweeps.  K is SPD for a positive definite kernel (route 7: the MFMA Cholesky); a
   semi-definite K (duplicate sites with nugget 0) fails there with GSL_EDOM and goes through the pivoted LDL^T
   of linalg/pcholesky.c (route 8). */
__global__ void __launch_bounds__(256)
diag_add_kernel(double *__restrict__ a, size_t lda, size_t n, double v)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i * lda + i] += v;
}

__global__ void __launch_bounds__(256)
krige_rhs_kernel(const double *__restrict__ f, size_t n, double *__restrict__ Y, double *__restrict__ ones)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { Y[i] = f[i]; Y[n + i] = 1.0; ones[i] = 1.0; }
}

__global__ void __launch_bounds__(256)
add_const_kernel(double *__restrict__ s, size_t m, double c)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) s[k] = s[k] + c;
}

extern "C" int gsl_sinterp_hip_krige_solve(gsl_sinterp_hip_ctx *ctx, int kind, double eps, double nugget, const double *d_x, size_t n,
                                           int dim, size_t xtda, double *d_phi, size_t lda, double *d_w, double *h_mean,
                                           int *h_route)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  EXCLUSIVE_SECTION(ctx);
  REQUIRE(ctx, dim >= 1 && dim <= 3 && xtda >= (size_t)dim && lda >= n && nugget >= 0.0, ST_EINVAL);
  REQUIRE(ctx, kind == GSL_SINTERP_RBF_GAUSSIAN || kind == GSL_SINTERP_RBF_WENDLAND, ST_EINVAL);   /* covariances: positive definite kernels */
  REQUIRE(ctx, h_mean != NULL && (n == 0 || (d_x && d_phi && d_w)), ST_EFAULT);
  if (h_route) *h_route = 0;
  *h_mean = 0.0;
  if (n == 0) return ST_SUCCESS;
  void *aux = NULL;
  int st = sinterp_aux(ctx, (3 * n + 64) * sizeof(double), &aux);
  if (st) return st;
  double *Y = (double *)aux, *ones = Y + 2 * n, *G = ones + n;
  const unsigned nb = (unsigned)((n + 255) / 256);
  int route = 7;
  for (int attempt = 0; attempt < 2; attempt++) {
    st = gsl_sinterp_hip_rbf_fill(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda);       /* both triangles: the pivoted route needs them */
    if (st) return st;
    if (nugget != 0.0) hipLaunchKernelGGL(diag_add_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_phi, lda, n, nugget);
    hipLaunchKernelGGL(krige_rhs_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)d_w, n, Y, ones);
    LAUNCH_CHECK(ctx);
    if (attempt == 0) {
      int info = 0;
      st = sinterp_cholesky_factor_solve_sym(ctx, n, d_phi, lda, &info, Y, n, 2);
      if (st == ST_EDOM) continue;                                                      /* only semi-definite: pivoted LDL^T (Y is rebuilt) */
      if (st) return st;
      break;
    }
    route = 8;
    int *d_perm = NULL;
    st = gsl_sinterp_hip_malloc(ctx, (void **)&d_perm, n * sizeof(int));
    if (!st) st = gsl_sinterp_hip_pcholesky_decomp(ctx, n, d_phi, lda, d_perm);
    if (!st) st = gsl_sinterp_hip_pcholesky_svx(ctx, n, d_phi, lda, d_perm, Y);
    if (!st) st = gsl_sinterp_hip_pcholesky_svx(ctx, n, d_phi, lda, d_perm, Y + n);
    if (!st) st = gsl_sinterp_hip_sync(ctx);
    gsl_sinterp_hip_free(ctx, d_perm);
    if (st) return st;
  }
  hipLaunchKernelGGL(gram_kernel, dim3(2), dim3(256), 0, ctx->stream, (const double *)ones, (const double *)Y, n, 1, G);   /* 1^T a, 1^T b */
  LAUNCH_CHECK(ctx);
  double hG[2] = {0.0, 0.0};
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(hG, G, sizeof hG, hipMemcpyDeviceToHost));
  if (!(hG[1] != 0.0) || hG[1] != hG[1])
    return sinterp_fail(ctx, ST_EDOM, "krige_solve: 1^T K^-1 1 = 0 (degenerate covariance matrix)", hipSuccess, __FILE__, __LINE__);
  const double mu = hG[0] / hG[1], coef = -mu;
  HIP_OK(ctx, hipMemcpyAsync(G, &coef, sizeof coef, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(combine_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)Y, n, 1, (const double *)G, d_w);
  LAUNCH_CHECK(ctx);
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));                                        /* coef is a stack variable */
  *h_mean = mu;
  if (h_route) *h_route = route;
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_krige_eval(gsl_sinterp_hip_ctx *ctx, int kind, double eps, double mean, const double *d_x, size_t n,
                                          int dim, size_t xtda, const double *d_w, const double *d_y, size_t m, size_t ytda,
                                          double *d_s, unsigned long long model_id)
{
  int st = gsl_sinterp_hip_rbf_eval_model(ctx, kind, eps, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, model_id);
  if (st || m == 0) return st;
  size_t blocks = (m + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_const_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_s, m, mean);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}
