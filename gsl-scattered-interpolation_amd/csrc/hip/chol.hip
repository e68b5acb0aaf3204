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
#include <stdlib.h>

#ifdef SINTERP_DIAG_PROF
__device__ unsigned long long g_diag_ts[80];
#define TSTAMP(i) do { if (threadIdx.x == 0) g_diag_ts[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int gsl_sinterp_hip_debug_diag_ts(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_ts), sizeof(unsigned long long) * 80); }
#endif
#include "chol_potrf.h"

#define TRSV_MAXR 5   /* right-hand sides solved together (f + the d+1 polynomial columns) */


/* ------------------------------------------------------------------------ */
/* base: factor the nb x nb diagonal block (nb <= 32) and solve the rows below */
__global__ void __launch_bounds__(256)
chol_base_kernel(double *__restrict__ A, size_t lda, size_t n, size_t j0, int nb, int *__restrict__ info,
                 double *__restrict__ diag_store)
{
  __shared__ double sL[CB][CB + 1];
  __shared__ double sInv[CB];
  const int tid = threadIdx.x;

  /* this thread's row of the panel: the loads are issued before the serial factorisation so
     their latency hides behind it */
  const size_t row = j0 + nb + (size_t)blockIdx.x * blockDim.x + tid;
  double *p = A + row * lda + j0;
  double x[CB];
  const bool vecrow = nb == CB && ((((uintptr_t)p) & 15) == 0) && ((lda & 1) == 0);
  if (row < n) {
    if (vecrow) {
#pragma unroll
      for (int k = 0; k < CB; k += 2) { const double2 t = *reinterpret_cast<const double2 *>(p + k); x[k] = t.x; x[k + 1] = t.y; }
    } else {
#pragma unroll
      for (int k = 0; k < CB; k++) x[k] = k < nb ? p[k] : 0.0;
    }
  }

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
      /* sqrt(d) and 1/sqrt(d) from one v_rsq_f64 seed + FMA refinement: this pair sits on
         the serial chain of the panel, a libm sqrt followed by an IEEE divide would cost
         ~10x the cycles.  Both results end within 1 ulp of cholesky.c:125-126's values. */
      double y = __builtin_amdgcn_rsq(d);
      y = y * fma(-0.5 * d * y, y, 1.5);
      y = y * fma(-0.5 * d * y, y, 1.5);
      double sd = d * y;
      sd = fma(fma(-sd, sd, d), 0.5 * y, sd);
      const double inv = fma(fma(-sd, y, 1.0), y, y);
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
  if (row >= n) return;
#pragma unroll
  for (int j = 0; j < CB; j++) {
    double v = x[j];
#pragma unroll
    for (int k = 0; k < j; k++) v = fma(-x[k], sL[j][k], v);
    x[j] = v * sInv[j];
  }
  if (vecrow) {
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

/* ------------------------------------------------------------------------ */
/* 128-wide panels: two launches per panel instead of the 4 base + 3 small-K launches of the
   32-wide recursion.
     chol_diag128_kernel  (one workgroup): factors the 128x128 diagonal block in LDS -- four
        32-wide steps of {single-wave potrf32 on registers, row-per-thread TRSM, MFMA trailing
        update} -- writes L in place and the inverses of its four 32x32 diagonal blocks (forward
        substitution, overlapped with the TRSM phase) to a side buffer;
     chol_trsm128_kernel  (64 rows per workgroup, 16 per wave): X = B L^-T by block substitution,
        every step an MFMA product:  X_c = (B_c - sum_{p<c} X_p L_cp^T) Dinv_c^T,  c = 0..3,
        B tile, the off-diagonal blocks of L and the Dinv blocks staged in LDS.
   Multiplying by explicitly inverted 32x32 diagonal blocks of a Cholesky factor is the standard
   GPU formulation of the panel TRSM (the sweeps further down do the same with 64x64 blocks).
   LDS layout of triangles: packed 32x32 blocks of pitch 34 doubles -- 34 = 2 mod 4 makes the
   (row = lane&15, k = lane>>4) MFMA fragment reads conflict free. */

/* Forward substitution folded into the factorisation (round 4).  With right-hand sides f (nrhs vectors, fb + q * ldf),
   the driver keeps them "one panel ahead": when panel p is factored, f_p (rows j0 .. j0+127) already carries the updates of
   every earlier panel, so  y_p = L_pp^-1 f_p  is final (cblas/source_trsv_r.h:56-79 computes the same sums row by row);
   chol_trsm16_kernel then subtracts L[R, p] y_p from the entries below.  After the last panel fb holds L^-1 f and the
   solve only needs its backward sweep.  Here: block substitution with the inverted 32 x 32 diagonal blocks that the
   factorisation leaves in Dv, one right-hand side per wave (wave-local: no workgroup barrier), lanes = 32 rows x 2 halves
   of the K range. */
__device__ __forceinline__ void diag128_forward(const double *S, const double *Dv, double *fs, double *tt, int tid, int nrhs)
{
  const int lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
  for (int q = wave; q < nrhs; q += 4) {
    double *y = fs + q * PB, *t = tt + q * CB;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double sum = 0.0;
#pragma unroll
      for (int kb = 0; kb < i; kb++) {
        const double *Lr = S + pblk(i, kb) + r * PQ + half * 16;
        const double *yk = y + kb * 32 + half * 16;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          const double2 l2 = *reinterpret_cast<const double2 *>(Lr + k);
          sum = fma(l2.x, yk[k], sum);
          sum = fma(l2.y, yk[k + 1], sum);
        }
      }
      sum += __shfl_xor(sum, 32);
      if (half == 0) t[r] = y[i * 32 + r] - sum;
      __builtin_amdgcn_wave_barrier();
      /* y_i = Dinv_i t  (Dinv lower triangular: exact zeros above the diagonal) */
      const double *Wr = Dv + i * PBLK + (half * 16) * PQ + r;           /* Dinv[r][c] = Dt[c * PQ + r] (stored transposed) */
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < 16; c++) acc = fma(Wr[c * PQ], t[half * 16 + c], acc);
      acc += __shfl_xor(acc, 32);
      if (half == 0) y[i * 32 + r] = acc;
      __builtin_amdgcn_wave_barrier();
    }
  }
}

__global__ void __launch_bounds__(256)
chol_diag128_kernel(double *__restrict__ A, size_t lda, size_t j0, int *__restrict__ info, double *__restrict__ diag_store,
                    double *__restrict__ Dinvg, double *__restrict__ fb, size_t ldf, int nrhs)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *S = sm;                       /* 10 packed blocks of the lower triangle */
  double *Dv = S + 10 * PBLK;           /* 4 blocks: inverses of the diagonal blocks */
  double *fs = Dv + 4 * PBLK;           /* [TRSV_MAXR][128] right-hand sides of the folded forward substitution */
  double *tt = fs + 5 * PB;             /* [TRSV_MAXR][32] */
  const int tid = threadIdx.x;
  double *Ab = A + j0 * lda + j0;
  TSTAMP(0);

  /* load the lower triangle (whole 32x32 blocks, coalesced along k): all 40 loads of a thread are
     issued before the first LDS store (one memory round trip, not 40) */
  {
    double v[40], fv[3];
    const int r8 = tid >> 5, k = tid & 31;
#pragma unroll
    for (int t = 0; t < 40; t++) {
      constexpr int BI[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3}, BJ[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
      const int b = t >> 2, r = (t & 3) * 8 + r8;
      v[t] = Ab[(size_t)(BI[b] * 32 + r) * lda + BJ[b] * 32 + k];
    }
#pragma unroll
    for (int t = 0; t < 3; t++) { const int e = t * 256 + tid; fv[t] = e < nrhs * PB ? fb[(size_t)(e >> 7) * ldf + j0 + (e & 127)] : 0.0; }
#pragma unroll
    for (int t = 0; t < 40; t++) S[(t >> 2) * PBLK + ((t & 3) * 8 + r8) * PQ + k] = v[t];
#pragma unroll
    for (int t = 0; t < 16; t++) { const int r = (t & 3) * 8 + r8; Dv[(t >> 2) * PBLK + r * PQ + k] = r == k ? 1.0 : 0.0; }   /* potrf32: identity */
#pragma unroll
    for (int t = 0; t < 3; t++) { const int e = t * 256 + tid; if (e < nrhs * PB) fs[e] = fv[t]; }
  }
  __syncthreads();
  TSTAMP(1);

  potrf128_lds<4>(S, Dv, tid, info, j0);

  TSTAMP(18);
  if (nrhs > 0) {
    diag128_forward(S, Dv, fs, tt, tid, nrhs);
    __syncthreads();
  }
  /* L -> A (lower part only); the diagonal 32-blocks also -> diag_store in the format of
     chol_base_kernel, so that chol_diag_writeback_kernel rewrites the same values */
  {
    const int r8 = tid >> 5, k = tid & 31;
#pragma unroll
    for (int t = 0; t < 40; t++) {
      constexpr int BI[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3}, BJ[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
      const int b = t >> 2, r = (t & 3) * 8 + r8, bi = BI[b], bj = BJ[b];
      const double v = S[b * PBLK + r * PQ + k];
      if (bi != bj || k <= r) Ab[(size_t)(bi * 32 + r) * lda + bj * 32 + k] = v;
      if (bi == bj && diag_store) diag_store[(j0 / CB + bi) * (CB * CB) + r * CB + k] = (k <= r) ? v : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int b = t >> 2, r = (t & 3) * 8 + r8;
      Dinvg[b * 1024 + r * 32 + k] = Dv[b * PBLK + k * PQ + r];             /* row-major Dinv from the transposed LDS image */
    }
#pragma unroll
    for (int t = 0; t < 3; t++) { const int e = t * 256 + tid; if (e < nrhs * PB) fb[(size_t)(e >> 7) * ldf + j0 + (e & 127)] = fs[e]; }
  }
  TSTAMP(19);
}

/* rows below a 128-wide diagonal block: X = B L^-T in place (round 4).
   Block substitution over the four 32-column blocks c:   Y_c = B_c - sum_{p<c} X_p L_cp^T,   X_c = Y_c Dinv_c^T.
   The round-2 kernel gave every wave a 16-row strip for all 128 columns: 144 dependent MFMAs per wave (3.8 us) whatever
   the height of the panel, and its 64 rows per workgroup left most CUs idle below ~16 k rows.  Here a workgroup owns
   16 RF rows, RF chosen by the driver so that the launch is about one workgroup per CU:
     RF = 1 (panels up to 4 k rows): the four waves split every product as (f, h) = (16-column fragment of the block, half
             of the K range); the partial sums meet in LDS and are added when read back as A operands -- 32 / 40 MFMAs per wave;
     RF = 2, 4: wave (f, g) owns the row fragments g, g + 2 with the whole K range (no partial sums), 80 MFMAs per row fragment.
   7 workgroup barriers.  Every operand that does not depend on an earlier step (B tile, the L blocks and Dinv blocks as MFMA
   B fragments) is fetched from global / L2 straight into registers at kernel start: one memory round trip, no LDS staging
   (with 16 rows per workgroup at 16 k rows those fragment loads alone were 83 MB per launch: hence RF).
   Folded forward substitution (nrhs > 0, see diag128_forward): f[R] -= X_R y_p while X_R passes through LDS on its way out. */
#define TRX 130
#define TRY 34
template <int RF>
__global__ void __launch_bounds__(256)
chol_trsm16_kernel(double *__restrict__ A, size_t lda, size_t n, size_t j0, const double *__restrict__ Dinvg, size_t row_start,
                   double *__restrict__ fb, size_t ldf, int nrhs)
{
  constexpr bool KSPLIT = RF == 1;
  constexpr int NP = KSPLIT ? 2 : 1;              /* planes of partial sums */
  constexpr int NRF = KSPLIT ? 1 : RF / 2;        /* row fragments per wave */
  constexpr int ROWS = 16 * RF;
  constexpr int LCH = KSPLIT ? 4 : 8;             /* K chunks (of 4) per 32-wide block in the L products */
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *Xp = sm;                                /* [NP][ROWS][TRX] */
  double *Yp = Xp + NP * ROWS * TRX;              /* [NP][ROWS][TRY] */
  double *ys = Yp + NP * ROWS * TRY;              /* [TRSV_MAXR][PB] */
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = wave & 1, g = wave >> 1, fr = lane & 15, fq = lane >> 4;
  const int plane = KSPLIT ? g : 0;
  const size_t row0 = row_start + (size_t)blockIdx.x * ROWS;
  const int dch = KSPLIT ? (f ? 4 : 2) : (f ? 8 : 4);       /* K chunks of this wave in the Dinv products: K = 16 (f + 1) */
  const int dk0 = KSPLIT ? dch * g : 0, lk0 = KSPLIT ? 4 * g : 0;

  /* ---- every global load of the kernel, issued before the first use */
  double a0[NRF][8], dv[4][8], lb[6][LCH], yv[3];
  double4_t cinit[NRF][3];
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int i = 0; i < 8; i++) dv[c][i] = i < dch ? Dinvg[c * 1024 + (16 * f + fr) * 32 + 4 * (dk0 + i) + fq] : 0.0;
#pragma unroll
  for (int c = 1; c < 4; c++)
#pragma unroll
    for (int p = 0; p < c; p++)
#pragma unroll
      for (int i = 0; i < LCH; i++)
        lb[c * (c - 1) / 2 + p][i] = A[(j0 + 32 * c + 16 * f + fr) * lda + j0 + 32 * p + 4 * (lk0 + i) + fq];
#pragma unroll
  for (int t = 0; t < NRF; t++) {
    const int rf = KSPLIT ? 0 : g + 2 * t;
    const size_t arow = row0 + 16 * rf + fr < n ? row0 + 16 * rf + fr : n - 1;
    const double *Arow = A + arow * lda + j0;               /* this lane's row as MFMA A operand */
#pragma unroll
    for (int i = 0; i < 8; i++) a0[t][i] = i < dch ? Arow[4 * (dk0 + i) + fq] : 0.0;
#pragma unroll
    for (int c = 1; c < 4; c++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const size_t r = row0 + 16 * rf + fq + 4 * rg < n ? row0 + 16 * rf + fq + 4 * rg : n - 1;
        cinit[t][c - 1][rg] = plane == 0 ? A[r * lda + j0 + 32 * c + 16 * f + fr] : 0.0;
      }
  }
#pragma unroll
  for (int t = 0; t < 3; t++) { const int e = t * 256 + tid; yv[t] = e < nrhs * PB ? fb[(size_t)(e >> 7) * ldf + j0 + (e & 127)] : 0.0; }
#pragma unroll
  for (int t = 0; t < 3; t++) { const int e = t * 256 + tid; if (e < nrhs * PB) ys[e] = yv[t]; }

  double facc[RF][TRSV_MAXR];
#pragma unroll
  for (int t = 0; t < RF; t++)
#pragma unroll
    for (int q = 0; q < TRSV_MAXR; q++) facc[t][q] = 0.0;
  const int sr = tid >> 4, sc = (tid & 15) * 2;             /* store phase: row (+ 16 per pass) and column pair of this thread */

#pragma unroll
  for (int c = 0; c < 4; c++) {
    if (c > 0) {
#pragma unroll
      for (int t = 0; t < NRF; t++) {
        const int rf = KSPLIT ? 0 : g + 2 * t;
        double4_t acc = cinit[t][c - 1];
#pragma unroll
        for (int p = 0; p < c; p++)
#pragma unroll
          for (int i = 0; i < LCH; i++) {
            const int k = 32 * p + 4 * (lk0 + i) + fq;
            double a = Xp[(16 * rf + fr) * TRX + k];
            if constexpr (KSPLIT) a += Xp[ROWS * TRX + (16 * rf + fr) * TRX + k];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, lb[c * (c - 1) / 2 + p][i], acc, 0, 0, 0);
          }
#pragma unroll
        for (int rg = 0; rg < 4; rg++) Yp[plane * ROWS * TRY + (16 * rf + fq + 4 * rg) * TRY + 16 * f + fr] = acc[rg];
      }
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NRF; t++) {
      const int rf = KSPLIT ? 0 : g + 2 * t;
      double4_t x = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (i < dch) {
          const int k = 4 * (dk0 + i) + fq;
          double a = a0[t][i];
          if (c > 0) {
            a = Yp[(16 * rf + fr) * TRY + k];
            if constexpr (KSPLIT) a += Yp[ROWS * TRY + (16 * rf + fr) * TRY + k];
          }
          x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, dv[c][i], x, 0, 0, 0);
        }
      }
#pragma unroll
      for (int rg = 0; rg < 4; rg++) Xp[plane * ROWS * TRX + (16 * rf + fq + 4 * rg) * TRX + 32 * c + 16 * f + fr] = x[rg];
    }
    __syncthreads();
    /* X_c on its way out: 16 RF rows x 32 columns, two columns per thread and pass */
#pragma unroll
    for (int t = 0; t < RF; t++) {
      const int r = sr + 16 * t;
      double2 xv = *reinterpret_cast<const double2 *>(&Xp[r * TRX + 32 * c + sc]);
      if constexpr (KSPLIT) {
        const double2 x1 = *reinterpret_cast<const double2 *>(&Xp[ROWS * TRX + r * TRX + 32 * c + sc]);
        xv.x += x1.x; xv.y += x1.y;
      }
      const size_t grow = row0 + r;
      if (grow < n) *reinterpret_cast<double2 *>(A + grow * lda + j0 + 32 * c + sc) = xv;
#pragma unroll
      for (int q = 0; q < TRSV_MAXR; q++)
        if (q < nrhs) facc[t][q] = fma(xv.y, ys[q * PB + 32 * c + sc + 1], fma(xv.x, ys[q * PB + 32 * c + sc], facc[t][q]));
    }
  }
  if (nrhs > 0) {
#pragma unroll
    for (int t = 0; t < RF; t++) {
      const size_t grow = row0 + sr + 16 * t;
#pragma unroll
      for (int q = 0; q < TRSV_MAXR; q++) {
        if (q < nrhs) {
          double v = facc[t][q];
          v += __shfl_xor(v, 1);
          v += __shfl_xor(v, 2);
          v += __shfl_xor(v, 4);
          v += __shfl_xor(v, 8);
          if ((tid & 15) == 0 && grow < n) fb[(size_t)q * ldf + grow] -= v;
        }
      }
    }
  }
}

template <int RF>
static int launch_trsm16(gsl_sinterp_hip_ctx *ctx, double *A, size_t lda, size_t n, size_t j0, const double *d_linv, size_t row_start,
                         double *fb, size_t ldf, int nrhs)
{
  constexpr int NP = RF == 1 ? 2 : 1;
  const size_t lds = (size_t)(NP * 16 * RF * (TRX + TRY) + TRSV_MAXR * PB) * sizeof(double);
  { int ast = sinterp_func_lds(ctx, (const void *)chol_trsm16_kernel<RF>, (int)lds); if (ast) return ast; }
  const size_t below = n - row_start;
  hipLaunchKernelGGL(chol_trsm16_kernel<RF>, dim3((unsigned)((below + 16 * RF - 1) / (16 * RF))), dim3(256), lds, ctx->stream, A, lda, n, j0,
                     d_linv, row_start, fb, ldf, nrhs);
  return ST_SUCCESS;
}

/* tall panels (more than 4 k rows below the block; round 2): 64 rows per workgroup, wave w owns rows 16w..16w+15 for all
   four 32-column steps, so the steps need no workgroup barrier; B tile, L blocks and Dinv blocks staged in LDS with
   coalesced 16-byte loads.  144 dependent MFMAs per wave, which only pays when there is a workgroup for every CU. */
__global__ void __launch_bounds__(256)
chol_trsm128_kernel(double *__restrict__ A, size_t lda, size_t n, size_t j0, const double *__restrict__ Dinvg, size_t row_start,
                    double *__restrict__ fb, size_t ldf, int nrhs)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *Bt = sm;                     /* [64][TR_LD] */
  double *Lb = Bt + 64 * TR_LD;        /* 6 off-diagonal blocks of L: (bi, bj) at bi(bi-1)/2 + bj */
  double *Dvb = Lb + 6 * PBLK;         /* 4 inverted diagonal blocks */
  double *ys = Dvb + 4 * PBLK;         /* [TRSV_MAXR][PB]: y_p of the folded forward substitution */
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const size_t row0 = row_start + (size_t)blockIdx.x * 64;    /* rows [row_start, n): the whole panel below the block, or a slice of it */
  {
    /* one round trip: every global load is issued before the first LDS store */
    double2 vb[16];
    double vl[24], vd[16], yv[3];
    const int r8 = tid >> 5, k = tid & 31;
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int e = t * 256 + tid, r = e >> 6, k2 = (e & 63) * 2;
      const size_t grow = row0 + r < n ? row0 + r : n - 1;
      vb[t] = *reinterpret_cast<const double2 *>(A + grow * lda + j0 + k2);
    }
#pragma unroll
    for (int t = 0; t < 24; t++) {
      constexpr int BI[6] = {1, 2, 2, 3, 3, 3}, BJ[6] = {0, 0, 1, 0, 1, 2};
      const int b = t >> 2, r = (t & 3) * 8 + r8;
      vl[t] = A[(j0 + BI[b] * 32 + r) * lda + j0 + BJ[b] * 32 + k];
    }
#pragma unroll
    for (int t = 0; t < 16; t++) vd[t] = Dinvg[(t >> 2) * 1024 + ((t & 3) * 8 + r8) * 32 + k];
#pragma unroll
    for (int t = 0; t < 3; t++) { const int e = t * 256 + tid; yv[t] = e < nrhs * PB ? fb[(size_t)(e >> 7) * ldf + j0 + (e & 127)] : 0.0; }
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int e = t * 256 + tid, r = e >> 6, k2 = (e & 63) * 2;
      Bt[r * TR_LD + k2] = vb[t].x; Bt[r * TR_LD + k2 + 1] = vb[t].y;
    }
#pragma unroll
    for (int t = 0; t < 3; t++) { const int e = t * 256 + tid; if (e < nrhs * PB) ys[e] = yv[t]; }
#pragma unroll
    for (int t = 0; t < 24; t++) Lb[(t >> 2) * PBLK + ((t & 3) * 8 + r8) * PQ + k] = vl[t];
#pragma unroll
    for (int t = 0; t < 16; t++) Dvb[(t >> 2) * PBLK + ((t & 3) * 8 + r8) * PQ + k] = vd[t];
  }
  __syncthreads();

  double *arow = Bt + (wave * 16 + fr) * TR_LD + fq;        /* A-operand view of this wave's rows */
  double *drow = Bt + (wave * 16 + fq) * TR_LD + fr;        /* accumulator (D layout) view */
#pragma unroll
  for (int c = 0; c < 4; c++) {
    double4_t acc[2];
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) acc[f][rg] = drow[4 * rg * TR_LD + c * 32 + f * 16];
#pragma unroll
    for (int p = 0; p < c; p++) {
      const double *lb = Lb + (c * (c - 1) / 2 + p) * PBLK + fr * PQ + fq;
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        const double a = -arow[p * 32 + kk * 4];
#pragma unroll
        for (int f = 0; f < 2; f++) acc[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lb[f * 16 * PQ + kk * 4], acc[f], 0, 0, 0);
      }
    }
    /* Y -> LDS (own rows), then X_c = Y Dinv_c^T (Dinv lower triangular: fragment f needs K = 16(f+1)) */
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) drow[4 * rg * TR_LD + c * 32 + f * 16] = acc[f][rg];
    const double *db = Dvb + c * PBLK + fr * PQ + fq;
#pragma unroll
    for (int f = 0; f < 2; f++) {
      acc[f] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < (f + 1) * 4; kk++)
        acc[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[c * 32 + kk * 4], db[f * 16 * PQ + kk * 4], acc[f], 0, 0, 0);
    }
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        drow[4 * rg * TR_LD + c * 32 + f * 16] = acc[f][rg];
        const size_t grow = row0 + wave * 16 + fq + 4 * rg;
        if (grow < n) A[grow * lda + j0 + c * 32 + f * 16 + fr] = acc[f][rg];
      }
  }
  if (nrhs > 0) {
    /* folded forward substitution: f[R] -= X_R y_p, X_R is in Bt now (every wave wrote its own rows) */
    __syncthreads();
    const int r = tid >> 2, qd = tid & 3;
    const size_t grow = row0 + r;
    for (int q = 0; q < nrhs; q++) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 32; k++) v = fma(Bt[r * TR_LD + qd * 32 + k], ys[q * PB + qd * 32 + k], v);
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      if (qd == 0 && grow < n) fb[(size_t)q * ldf + grow] -= v;
    }
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

/* width of the left part of a panel of width w > CB: half, rounded up to a 128-column boundary while
   the panel is wider than 128 (so the recursion ends in exact 128-wide panels), to 32 below */
static size_t chol_split(size_t w)
{
  const size_t unit = w > PB ? PB : CB;
  size_t w1 = ((w / 2 + unit - 1) / unit) * unit;
  if (w1 >= w) w1 = w - unit;
  return w1;
}

/* fb != NULL: nrhs right-hand sides (fb + q * ldf) ride along -- only ever passed when every leaf of the recursion is a
   128-wide panel (chol_fold_applicable), see diag128_forward */
static int chol_panel(gsl_sinterp_hip_ctx *ctx, double *A, size_t lda, size_t n, size_t j0, size_t w, int *d_info,
                      double *d_diag, double *fb, size_t ldf, int nrhs)
{
  int st;
  static const bool no_p128 = getenv("GSL_SINTERP_NO_PANEL128") && getenv("GSL_SINTERP_NO_PANEL128")[0] == '1';
  if (w == PB && !no_p128 && (lda & 1) == 0 && ((((uintptr_t)(A + j0 * lda + j0)) & 15) == 0)) {
    double *d_linv = d_diag + ((n + CB - 1) / CB) * (CB * CB);   /* 4 inverted 32x32 blocks */
    const size_t lds_diag = (size_t)(14 * PBLK + TRSV_MAXR * (PB + CB)) * sizeof(double);
    { int ast = sinterp_func_lds(ctx, (const void *)chol_diag128_kernel, (int)lds_diag); if (ast) return ast; }
    /* every panel of this factorisation is 128 wide (n a multiple of 128): the blocks are written straight into A and the
       write-back launch of the 32-wide path is skipped, so the side copy is not needed either */
    hipLaunchKernelGGL(chol_diag128_kernel, dim3(1), dim3(256), lds_diag, ctx->stream, A, lda, j0, d_info,
                       n % PB == 0 ? (double *)NULL : d_diag, d_linv, fb, ldf, nrhs);
    const size_t below = n - j0 - w;
    if (below) {
      /* 16 rows per workgroup while that is at most about one workgroup per CU, 64-row strips above
         (developer override: GSL_SINTERP_TRSM_RF = 1, 2, 4 forces the 16 RF-row kernel, 64 the strip kernel) */
      static const int force_rf = getenv("GSL_SINTERP_TRSM_RF") ? atoi(getenv("GSL_SINTERP_TRSM_RF")) : 0;
      int ast = 0;
      if (force_rf == 1 || (!force_rf && below <= 4096)) ast = launch_trsm16<1>(ctx, A, lda, n, j0, d_linv, j0 + PB, fb, ldf, nrhs);
      else if (force_rf == 2) ast = launch_trsm16<2>(ctx, A, lda, n, j0, d_linv, j0 + PB, fb, ldf, nrhs);
      else if (force_rf == 4) ast = launch_trsm16<4>(ctx, A, lda, n, j0, d_linv, j0 + PB, fb, ldf, nrhs);
      else {
        const size_t lds_trsm = (size_t)(64 * TR_LD + 10 * PBLK + TRSV_MAXR * PB) * sizeof(double);
        ast = sinterp_func_lds(ctx, (const void *)chol_trsm128_kernel, (int)lds_trsm);
        if (!ast)
          hipLaunchKernelGGL(chol_trsm128_kernel, dim3((unsigned)((below + 63) / 64)), dim3(256), lds_trsm, ctx->stream, A, lda, n, j0,
                             (const double *)d_linv, j0 + PB, fb, ldf, nrhs);
      }
      if (ast) return ast;
    }
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  if (w <= PB) {
    if (fb) return sinterp_fail(ctx, ST_EFAILED, "cholesky: folded forward substitution on a panel that is not 128 wide", hipSuccess, __FILE__, __LINE__);
  }
  if (w <= CB) {
    const size_t below = n - j0 - w;
    const unsigned grid = (unsigned)((below + 255) / 256) + (below == 0 ? 1u : 0u);
    hipLaunchKernelGGL(chol_base_kernel, dim3(grid ? grid : 1), dim3(256), 0, ctx->stream, A, lda, n, j0, (int)w, d_info, d_diag);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  const size_t w1 = chol_split(w);
  st = chol_panel(ctx, A, lda, n, j0, w1, d_info, d_diag, fb, ldf, nrhs);
  if (st) return st;
  const size_t r0 = j0 + w1, w2 = w - w1;
  st = sinterp_gemm_minus(ctx, n - r0, w2, w1, A + r0 * lda + j0, lda, A + r0 * lda + j0, lda, 0, A + r0 * lda + r0, lda, 1);
  if (st) return st;
  return chol_panel(ctx, A, lda, n, r0, w2, d_info, d_diag, fb, ldf, nrhs);
}

/* every leaf of the recursion is a 128-wide panel: the forward substitution can ride along */
static bool chol_fold_applicable(size_t n, const double *d_a, size_t lda, int nrhs)
{
  static const bool no_p128 = getenv("GSL_SINTERP_NO_PANEL128") && getenv("GSL_SINTERP_NO_PANEL128")[0] == '1';
  static const bool no_fold = getenv("GSL_SINTERP_NO_FOLD") && getenv("GSL_SINTERP_NO_FOLD")[0] == '1';
  return !no_p128 && !no_fold && nrhs >= 1 && nrhs <= TRSV_MAXR && n >= PB && n % PB == 0 && (lda & 1) == 0 && ((((uintptr_t)d_a) & 15) == 0);
}


__global__ void chol_zero_info_kernel(int *info) { *info = 0; }

/* symmetric_input: both triangles of d_a hold the matrix (the RBF fill writes it that way), so the
   copy that preserves the original in the strict upper triangle (cholesky.c:103) is already there */
/* fb / ldf / nrhs: optional right-hand sides for the folded forward substitution; *h_folded = 1 when fb holds L^-1 f on return */
/* defer_info: return right after the launches; the caller reads the pivot status later with chol_read_info (one host
   round trip for the factorisation AND whatever the caller enqueues behind it) */
static int chol_read_info(gsl_sinterp_hip_ctx *ctx, size_t n, int *h_info)
{
  int *d_info = (int *)ctx->d_scratch + 8;
  int info = 0;
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(&info, d_info, sizeof(int), hipMemcpyDeviceToHost));
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  if (h_info) *h_info = info;
  if (info) {
    snprintf(ctx->err, sizeof ctx->err, "cholesky_decomp1: matrix is not positive definite (pivot %d of %zu <= 0)", info, n);
    return ST_EDOM;
  }
  return ST_SUCCESS;
}

static int cholesky_decomp1_impl(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *h_info, bool symmetric_input,
                                 double *fb, size_t ldf, int nrhs, int *h_folded, bool defer_info = false)
{
  if (h_folded) *h_folded = 0;
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  EXCLUSIVE_SECTION(ctx);                         /* launches kernels that spin on sibling workgroups */
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || d_a, ST_EFAULT);
  if (h_info) *h_info = 0;
  if (n == 0) return ST_SUCCESS;
  int *d_info = (int *)ctx->d_scratch + 8;
  const size_t nblk = (n + CB - 1) / CB;
  void *d_diag = NULL;
  /* diagonal blocks + the inverted 32 x 32 blocks of EVERY 128-wide panel (look-ahead: a panel's row solve on the side
     stream still reads them while the chain factors the next panel) */
  const size_t n_pan = (n + PB - 1) / PB;
  int st = sinterp_workspace(ctx, (nblk * CB * CB + (n_pan + 1) * (4 * 1024)) * sizeof(double), &d_diag);
  if (st) return st;
  st = sinterp_streamk_prepare(ctx);
  if (st) return st;
  const bool fold = fb != NULL && chol_fold_applicable(n, d_a, lda, nrhs) && ldf >= n;
  if (!fold) { fb = NULL; nrhs = 0; }
  int replayed = 0;
  /* graph key: the right-hand-side buffer (16-byte aligned or not, its low bits are free below bit 3), their count and stride */
  const void *gkey = (const void *)(((uintptr_t)fb & ~(uintptr_t)7) ^ (uintptr_t)(symmetric_input ? 1 : 0) ^ ((uintptr_t)nrhs << 1) ^ ((uintptr_t)ldf << 44));
  st = sinterp_graph_try_launch(ctx, 0, n, lda, d_a, gkey, &replayed);
  if (st) return st;
  if (!replayed) {
    hipStream_t saved;
    st = sinterp_capture_begin(ctx, &saved);
    if (st) return st;
    /* a kernel, not a memset node: a memset node inside a replayed graph was observed not to be
       ordered/visible like the kernels around it (stale flags on the second launch of a sweep graph) */
    hipLaunchKernelGGL(chol_zero_info_kernel, dim3(1), dim3(1), 0, ctx->stream, d_info);
    hipError_t me = hipSuccess;
    const unsigned nt = (unsigned)((n + 31) / 32);
    if (!symmetric_input) hipLaunchKernelGGL(tricpy_lower_to_upper_kernel, dim3(nt, nt), dim3(256), 0, ctx->stream, d_a, lda, n);
    st = chol_panel(ctx, d_a, lda, n, 0, n, d_info, (double *)d_diag, fb, ldf, nrhs);
    {
      static const bool no_p128w = getenv("GSL_SINTERP_NO_PANEL128") && getenv("GSL_SINTERP_NO_PANEL128")[0] == '1';
      const bool all128 = !no_p128w && n % PB == 0 && (lda & 1) == 0 && ((((uintptr_t)d_a) & 15) == 0);
      if (!all128)
        hipLaunchKernelGGL(chol_diag_writeback_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, d_a, lda, n,
                           (const double *)d_diag);
    }
    int st2 = sinterp_capture_end(ctx, saved, 0, n, lda, d_a, gkey);
    if (me != hipSuccess) return sinterp_fail(ctx, ST_EFAILED, "zero info", me, __FILE__, __LINE__);
    if (st) return st;
    if (st2) return st2;
    LAUNCH_CHECK(ctx);
  }
  if (h_folded) *h_folded = fold ? 1 : 0;
  if (defer_info) return ST_SUCCESS;
  st = chol_read_info(ctx, n, h_info);
  if (st && h_folded) *h_folded = 0;
  return st;
}

extern "C" int gsl_sinterp_hip_cholesky_decomp1(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *h_info)
{
  return cholesky_decomp1_impl(ctx, n, d_a, lda, h_info, false, NULL, 0, 0, NULL);
}

int sinterp_cholesky_decomp1_sym(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *h_info)
{
  return cholesky_decomp1_impl(ctx, n, d_a, lda, h_info, true, NULL, 0, 0, NULL);
}

/* factor (symmetric input) and solve: d_x (nrhs vectors, d_x + q * ldx) <- (L L^T)^-1 d_x.  When every panel is 128 wide the
   forward substitution rides along with the factorisation and only the backward sweep remains (cholesky.c:178-181 does
   L c = b, then L^T x = c). */
int sinterp_cholesky_factor_solve_sym(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *h_info, double *d_x, size_t ldx,
                                      int nrhs)
{
  EXCLUSIVE_SECTION(ctx);
  int folded = 0;
  /* the sweeps are enqueued behind the factorisation without waiting for its pivot status: a failed factorisation
     leaves NaN in L, the sweeps carry them through (their hand-offs wait on epochs, not on values), and the one
     host round trip at the end reports GSL_EDOM */
  int st = cholesky_decomp1_impl(ctx, n, d_a, lda, h_info, true, d_x, ldx, nrhs, &folded, true);
  if (st) return st;
  if (!folded) st = sinterp_cholesky_svx_multi(ctx, n, d_a, lda, d_x, ldx, nrhs);
  else st = sinterp_trsv_multi(ctx, n, d_a, lda, d_x, d_x, ldx, nrhs, 1, 0);                /* L^T x = c, in place */
  if (st) return st;
  return chol_read_info(ctx, n, h_info);
}

/* ------------------------------------------------------------------------ */
/* blocked triangular sweeps shared with lu.hip.  One launch per 64-wide block
   column J: every workgroup solves the 64x64 diagonal system redundantly in
   its first wave, then updates its slice of the remaining right-hand side.
     mode 0: Lower, NoTrans, forward   (source_trsv_r.h:56-79)    rows i > J:  b_i -= sum_j T[i][j] x_j
     mode 1: Lower, Trans,  backward   (source_trsv_r.h:106-129)  cols i < J:  b_i -= sum_j T[j][i] x_j
     mode 2: Upper, NoTrans, backward  (source_trsv_r.h:33-55)    rows i < J:  b_i -= sum_j T[i][j] x_j   */
#define TS 64

/* right-hand side r lives at b + r*ldb (solved blocks at xout + r*ldb) */
__global__ void __launch_bounds__(256)
trsv_sweep_kernel(const double *__restrict__ T, size_t ldt, size_t n, double *__restrict__ b,
                  double *__restrict__ xout, size_t ldb, int nrhs, size_t j0, int nb, int mode, int unit)
{
  __shared__ double sx[TRSV_MAXR][TS];
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
  {
    /* wave w solves right-hand sides w, w+4, ... (one system per wave at a time) */
    const int lane = tid & 63, wave = tid >> 6;
    /* x_j = b_j / d_j sits on the serial chain: one parallel divide per lane up front,
       then q = b r, q += (b - d q) r  (a correctly rounded quotient in all but rare
       half-ulp cases) instead of an IEEE divide per step */
    const double dl = (lane < nb && !unit) ? sD[lane][lane] : 1.0;
    const double rl = 1.0 / dl;
    for (int r = wave; r < nrhs; r += 4) {
      double bi = lane < nb ? b[r * ldb + j0 + lane] : 0.0;
#define TRSV_STEP(J, COEF, COND)                                   \
      {                                                            \
        const double bj = lane_bcast(bi, (J));                     \
        double xj = bj;                                            \
        if (!unit) {                                               \
          const double dj = lane_bcast(dl, (J)), rj = lane_bcast(rl, (J)); \
          const double q = bj * rj;                                \
          xj = fma(fma(-dj, q, bj), rj, q);                        \
        }                                                          \
        if (lane == (J)) bi = xj;                                  \
        if (COND) bi = fma(-(COEF), xj, bi);                       \
      }
      if (mode == 0) {            /* forward with D = lower(sD) */
        for (int j = 0; j < nb; j++) TRSV_STEP(j, sD[lane][j], lane > j)
      } else if (mode == 1) {     /* backward with D^T, D lower: x_j then b_i -= D[j][i] x_j for i < j */
        for (int j = nb - 1; j >= 0; j--) TRSV_STEP(j, sD[j][lane], lane < j)
      } else {                    /* backward with D upper */
        for (int j = nb - 1; j >= 0; j--) TRSV_STEP(j, sD[lane][j], lane < j)
      }
#undef TRSV_STEP
      /* solved block goes to xout, never back into b: late workgroups of this launch re-read b[J] */
      if (lane < nb) { sx[r][lane] = bi; if (blockIdx.x == 0) xout[r * ldb + j0 + lane] = bi; }
    }
  }
  __syncthreads();

  /* remaining right-hand side */
  if (mode == 1) {
    /* column access T[j0+j][i]: 64 columns per workgroup, the nb rows split over the 4 waves */
    __shared__ double s_part[TRSV_MAXR][4][64];
    const int il = tid & 63, jg = tid >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + il;
    double acc[TRSV_MAXR];
#pragma unroll
    for (int r = 0; r < TRSV_MAXR; r++) acc[r] = 0.0;
    if (i < j0) {
      const int jb = jg * 16;
      double t[16];
#pragma unroll
      for (int jj = 0; jj < 16; jj++)                                /* 16 independent loads in flight */
        t[jj] = (jb + jj < nb) ? T[(j0 + jb + jj) * ldt + i] : 0.0;  /* coalesced across lanes */
#pragma unroll
      for (int jj = 0; jj < 16; jj++)
#pragma unroll
        for (int r = 0; r < TRSV_MAXR; r++) if (r < nrhs) acc[r] = fma(t[jj], sx[r][jb + jj], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < TRSV_MAXR; r++) if (r < nrhs) s_part[r][jg][il] = acc[r];
    __syncthreads();
    if (jg == 0 && i < j0) {
#pragma unroll
      for (int r = 0; r < TRSV_MAXR; r++)
        if (r < nrhs) b[r * ldb + i] -= (s_part[r][0][il] + s_part[r][1][il]) + (s_part[r][2][il] + s_part[r][3][il]);
    }
  } else {
    /* row access T[i][j0..j0+nb): 8 lanes share a row (8 x 64 B contiguous), 32 rows per pass */
    const int lane = tid & 63, wave = tid >> 6;
    const int rsub = lane >> 3, c8 = (lane & 7) * 8;
    const size_t base = (mode == 0) ? j0 + nb : 0;
    const size_t limit = (mode == 0) ? n : j0;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const size_t i = base + ((size_t)blockIdx.x * 2 + pass) * 32 + wave * 8 + rsub;
      double acc[TRSV_MAXR];
#pragma unroll
      for (int r = 0; r < TRSV_MAXR; r++) acc[r] = 0.0;
      if (i < limit) {
        const double *row = T + i * ldt + j0 + c8;
        double t[8];
        if (nb == TS && ((((uintptr_t)row) & 15) == 0)) {
#pragma unroll
          for (int k = 0; k < 8; k += 2) { const double2 v = *reinterpret_cast<const double2 *>(row + k); t[k] = v.x; t[k + 1] = v.y; }
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) t[k] = (c8 + k < nb) ? row[k] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < TRSV_MAXR; r++)
          if (r < nrhs) {
#pragma unroll
            for (int k = 0; k < 8; k++) acc[r] = fma(t[k], sx[r][c8 + k], acc[r]);
          }
      }
#pragma unroll
      for (int r = 0; r < TRSV_MAXR; r++) {
        if (r < nrhs) {
          double a = acc[r];
          a += __shfl_xor(a, 1);
          a += __shfl_xor(a, 2);
          a += __shfl_xor(a, 4);
          if (i < limit && (lane & 7) == 0) b[r * ldb + i] -= a;
        }
      }
    }
  }
}

/* ---- inverse of every 64x64 diagonal block, all blocks in parallel -----------------------
   The block is read as a LOWER triangular matrix Lb (for an upper factor: its transpose).
   Lane c solves Lb x = e_c by forward substitution with x in registers (no cross-lane
   traffic, Lb broadcast from LDS).  Dinv[blk][i][c] = (Lb^-1)[i][c]. */
__global__ void __launch_bounds__(64)
tri_inv_kernel(const double *__restrict__ T, size_t ldt, size_t n, int upper, int unit, double *__restrict__ Dinv)
{
  __shared__ double sL[TS][TS + 1];
  const size_t j0 = (size_t)blockIdx.x * TS;
  const int nb = (int)((n - j0) < TS ? (n - j0) : TS);
  const int lane = threadIdx.x;
  /* all 64 row loads in flight at once (clamped addresses, selected afterwards): as a loop of conditional loads the
     block arrived in 64 dependent round trips (45 of the kernel's 52 us) */
  {
    double tv[TS];
    const int lc = lane < nb ? lane : nb - 1;
#pragma unroll
    for (int r = 0; r < TS; r++) {
      const int rc = r < nb ? r : nb - 1;
      tv[r] = upper ? T[(j0 + lc) * ldt + j0 + rc] : T[(j0 + rc) * ldt + j0 + lc];
    }
#pragma unroll
    for (int r = 0; r < TS; r++) {
      double v = (r == lane) ? 1.0 : 0.0;                     /* identity padding past nb */
      if (r < nb && lane < nb) v = lane < r ? tv[r] : (lane == r ? (unit ? 1.0 : tv[r]) : 0.0);
      sL[r][lane] = v;
    }
  }
  __syncthreads();
  double x[TS];
#pragma unroll
  for (int i = 0; i < TS; i++) {
    double v = (i == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; k++) v = fma(-sL[i][k], x[k], v);
    x[i] = v / sL[i][i];
  }
  double *out = Dinv + (size_t)blockIdx.x * TS * TS;
#pragma unroll
  for (int i = 0; i < TS; i++) out[(size_t)i * TS + lane] = x[i];   /* coalesced: row i, column = lane */
}

/* sweep step with precomputed diagonal-block inverses: x_J = W b_J, W = Dinv (mode 0) or Dinv^T
   (modes 1, 2), then the remaining right-hand side is updated exactly as in trsv_sweep_kernel.
   The matrix entries of the update are fetched BEFORE x_J is formed, so the launch has one
   global round trip on its critical path. */
__global__ void __launch_bounds__(256)
trsv_sweep_inv_kernel(const double *__restrict__ T, size_t ldt, size_t n, double *__restrict__ b,
                      double *__restrict__ xout, size_t ldb, int nrhs, size_t j0, int nb, int mode,
                      const double *__restrict__ Dinv)
{
  __shared__ double sx[TRSV_MAXR][TS];
  __shared__ double sb[TRSV_MAXR][TS];
  __shared__ double sW[TS][TS + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  /* ---- prefetch this thread's slice of the update operand */
  double t[16];
  size_t irow = 0;
  bool have = false;
  const int rsub = lane >> 3, c8 = (lane & 7) * 8;
  if (mode == 1) {
    irow = (size_t)blockIdx.x * 64 + lane;
    have = irow < j0;
#pragma unroll
    for (int jj = 0; jj < 16; jj++)
      t[jj] = (have && wave * 16 + jj < nb) ? T[(j0 + wave * 16 + jj) * ldt + irow] : 0.0;
  } else {
    const size_t base = (mode == 0) ? j0 + nb : 0;
    const size_t limit = (mode == 0) ? n : j0;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const size_t i = base + ((size_t)blockIdx.x * 2 + pass) * 32 + wave * 8 + rsub;
      const double *row = T + i * ldt + j0 + c8;
#pragma unroll
      for (int k = 0; k < 8; k++) t[pass * 8 + k] = (i < limit && c8 + k < nb) ? row[k] : 0.0;
    }
  }
  /* ---- x_J = W b_J */
  const double *D = Dinv + (j0 / TS) * (size_t)(TS * TS);
  for (int e = tid; e < TS * TS; e += 256) {
    const int r = e / TS, c = e % TS;
    sW[r][c] = D[e];
  }
  for (int e = tid; e < nrhs * TS; e += 256) {
    const int r = e / TS, c = e % TS;
    sb[r][c] = c < nb ? b[r * ldb + j0 + c] : 0.0;
  }
  __syncthreads();
  for (int r = wave; r < nrhs; r += 4) {
    double acc = 0.0;
    if (mode == 0) { for (int c = 0; c < TS; c++) acc = fma(sW[lane][c], sb[r][c], acc); }
    else { for (int c = 0; c < TS; c++) acc = fma(sW[c][lane], sb[r][c], acc); }
    sx[r][lane] = acc;
    if (blockIdx.x == 0 && lane < nb) xout[r * ldb + j0 + lane] = acc;
  }
  __syncthreads();
  /* ---- remaining right-hand side */
  if (mode == 1) {
    __shared__ double s_part[TRSV_MAXR][4][64];
#pragma unroll
    for (int r = 0; r < TRSV_MAXR; r++) {
      if (r < nrhs) {
        double acc = 0.0;
#pragma unroll
        for (int jj = 0; jj < 16; jj++) acc = fma(t[jj], sx[r][wave * 16 + jj], acc);
        s_part[r][wave][lane] = acc;
      }
    }
    __syncthreads();
    if (wave == 0 && have) {
#pragma unroll
      for (int r = 0; r < TRSV_MAXR; r++)
        if (r < nrhs) b[r * ldb + irow] -= (s_part[r][0][lane] + s_part[r][1][lane]) + (s_part[r][2][lane] + s_part[r][3][lane]);
    }
  } else {
    const size_t base = (mode == 0) ? j0 + nb : 0;
    const size_t limit = (mode == 0) ? n : j0;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const size_t i = base + ((size_t)blockIdx.x * 2 + pass) * 32 + wave * 8 + rsub;
#pragma unroll
      for (int r = 0; r < TRSV_MAXR; r++) {
        if (r < nrhs) {
          double a = 0.0;
#pragma unroll
          for (int k = 0; k < 8; k++) a = fma(t[pass * 8 + k], sx[r][c8 + k], a);
          a += __shfl_xor(a, 1);
          a += __shfl_xor(a, 2);
          a += __shfl_xor(a, 4);
          if (i < limit && (lane & 7) == 0) b[r * ldb + i] -= a;
        }
      }
    }
  }
}

/* ---- the whole sweep in ONE launch ----------------------------------------------------------
   A sweep is a chain of nblk dependent block steps; as one launch per step it costs ~10 us per
   step of pure launch/drain latency (C3: 512 steps per solve).  Here workgroup w owns block w of
   the sweep order (and w + G, ... when there are more blocks than CUs): it streams the tiles
   T(I, J) of its block row against the already published x_J -- waiting on a per-block flag, the
   tile itself is fetched before the wait -- then x_I = W_I (b_I - sum), publishes x_I with
   agent-scope (sc1) stores and raises flag[I].  A workgroup only ever waits on blocks earlier in
   the sweep order, which belong to workgroups dispatched before it: no deadlock.  b is read only. */
__global__ void __launch_bounds__(256)
trsv_dataflow_kernel(const double *__restrict__ T, size_t ldt, size_t n, const double *__restrict__ b, double *xout, size_t ldb,
                     int nrhs, int mode, const double *__restrict__ Dinv, unsigned *tf, unsigned long long *xq, unsigned nblk)
{
  /* tf[0] is stable for the whole launch: only the workgroup of the LAST block of the sweep advances it,
     after every other block has been published (it has consumed them all).
     Hand-off of x: every entry is published as two 8-byte words {epoch | low half}, {epoch | high half}
     (aligned 8-byte accesses are single-copy atomic), so a consumer that polls the entry itself needs ONE
     memory round trip per dependent step instead of flag-then-data, and a word can never be mistaken for
     one of an earlier sweep. */
  const unsigned want = __hip_atomic_load(tf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const size_t npad = (size_t)nblk * TS;
  __shared__ double sx[2][TRSV_MAXR][TS];
  __shared__ double sW[TS][TS + 1];
  __shared__ double srhs[TRSV_MAXR][TS];
  __shared__ double s_part[TRSV_MAXR][4][TS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rsub = lane >> 3, c8 = (lane & 7) * 8;
  for (unsigned t = blockIdx.x; t < nblk; t += gridDim.x) {
    const unsigned I = (mode == 0) ? t : nblk - 1 - t;
    const size_t i0 = (size_t)I * TS;
    const int nbI = (int)((n - i0) < TS ? (n - i0) : TS);
    __syncthreads();                                     /* previous block of this workgroup fully done with LDS */
    for (int e = tid; e < TS * TS; e += 256) sW[e / TS][e % TS] = Dinv[(size_t)I * (TS * TS) + e];
    /* this block's right-hand side, fetched now: it is only needed after the last x_J has arrived, and
       a load issued there would sit on the critical path of the whole sweep */
    double bpre[(TRSV_MAXR * TS + 255) / 256];
#pragma unroll
    for (int q = 0; q < (TRSV_MAXR * TS + 255) / 256; q++) {
      const int e = q * 256 + tid, r = e / TS, c = e % TS;
      bpre[q] = (e < nrhs * TS && c < nbI) ? b[r * ldb + i0 + c] : 0.0;
    }
    double acc[TRSV_MAXR][2];
#pragma unroll
    for (int r = 0; r < TRSV_MAXR; r++) acc[r][0] = acc[r][1] = 0.0;
    for (unsigned sstep = 0; sstep < t; sstep++) {
      const unsigned J = (mode == 0) ? sstep : nblk - 1 - sstep;
      const size_t j0 = (size_t)J * TS;
      /* this thread's slice of the tile: independent of x_J, in flight while we wait for the flag */
      double tt[16];
      if (mode == 1) {
        /* T(J rows, I columns), used transposed: lane = column, wave = 16-row group */
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
          const size_t jr = j0 + wave * 16 + jj;
          tt[jj] = (jr < n && lane < nbI) ? T[jr * ldt + i0 + lane] : 0.0;
        }
      } else {
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          const size_t i = i0 + pass * 32 + wave * 8 + rsub;
          const double *row = T + i * ldt + j0 + c8;
          if (i < n && j0 + TS <= n && ((((uintptr_t)row) & 15) == 0)) {
#pragma unroll
            for (int k = 0; k < 8; k += 2) { const double2 v = *reinterpret_cast<const double2 *>(row + k); tt[pass * 8 + k] = v.x; tt[pass * 8 + k + 1] = v.y; }
          } else {
#pragma unroll
            for (int k = 0; k < 8; k++) tt[pass * 8 + k] = (i < n && j0 + c8 + k < n) ? row[k] : 0.0;
          }
        }
      }
      const int buf = sstep & 1;
      for (int e = tid; e < nrhs * TS; e += 256) {        /* nrhs can be 5: more entries than threads */
        const int r = e / TS, c = e % TS;
        double v = 0.0;
        if (j0 + c < n) {
          const unsigned long long *q = xq + 2 * ((size_t)r * npad + j0 + c);
          unsigned long long w0, w1;
          for (;;) {
            w0 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w0 >> 32) == want && (unsigned)(w1 >> 32) == want) break;
            __builtin_amdgcn_s_sleep(1);
          }
          v = __hiloint2double((int)(unsigned)w1, (int)(unsigned)w0);
        }
        sx[buf][r][c] = v;
      }
      __syncthreads();
      if (mode == 1) {
#pragma unroll
        for (int r = 0; r < TRSV_MAXR; r++)
          if (r < nrhs) {
#pragma unroll
            for (int jj = 0; jj < 16; jj++) acc[r][0] = fma(tt[jj], sx[buf][r][wave * 16 + jj], acc[r][0]);
          }
      } else {
#pragma unroll
        for (int r = 0; r < TRSV_MAXR; r++)
          if (r < nrhs) {
#pragma unroll
            for (int pass = 0; pass < 2; pass++)
#pragma unroll
              for (int k = 0; k < 8; k++) acc[r][pass] = fma(tt[pass * 8 + k], sx[buf][r][c8 + k], acc[r][pass]);
          }
      }
    }
    /* reduce the partial sums: b_I - sum -> srhs */
    if (mode == 1) {
#pragma unroll
      for (int r = 0; r < TRSV_MAXR; r++) if (r < nrhs) s_part[r][wave][lane] = acc[r][0];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < (TRSV_MAXR * TS + 255) / 256; q++) {
        const int e = q * 256 + tid, r = e / TS, c = e % TS;
        if (e < nrhs * TS) {
          const double sum = (s_part[r][0][c] + s_part[r][1][c]) + (s_part[r][2][c] + s_part[r][3][c]);
          srhs[r][c] = (c < nbI) ? bpre[q] - sum : 0.0;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < TRSV_MAXR; r++)
        if (r < nrhs) {
#pragma unroll
          for (int pass = 0; pass < 2; pass++) {
            double a = acc[r][pass];
            a += __shfl_xor(a, 1);
            a += __shfl_xor(a, 2);
            a += __shfl_xor(a, 4);
            const int row = pass * 32 + wave * 8 + rsub;
            if ((lane & 7) == 0) srhs[r][row] = -a;
          }
        }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < (TRSV_MAXR * TS + 255) / 256; q++) {
        const int e = q * 256 + tid, r = e / TS, c = e % TS;
        if (e < nrhs * TS) srhs[r][c] = (c < nbI) ? bpre[q] + srhs[r][c] : 0.0;
      }
    }
    __syncthreads();
    /* x_I = W srhs  (W = Dinv for the forward sweep, Dinv^T for the backward ones) */
    for (int r = wave; r < nrhs; r += 4) {
      double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
      if (mode == 0) {
#pragma unroll
        for (int c = 0; c < TS; c += 4) {
          p0 = fma(sW[lane][c], srhs[r][c], p0); p1 = fma(sW[lane][c + 1], srhs[r][c + 1], p1);
          p2 = fma(sW[lane][c + 2], srhs[r][c + 2], p2); p3 = fma(sW[lane][c + 3], srhs[r][c + 3], p3);
        }
      } else {
#pragma unroll
        for (int c = 0; c < TS; c += 4) {
          p0 = fma(sW[c][lane], srhs[r][c], p0); p1 = fma(sW[c + 1][lane], srhs[r][c + 1], p1);
          p2 = fma(sW[c + 2][lane], srhs[r][c + 2], p2); p3 = fma(sW[c + 3][lane], srhs[r][c + 3], p3);
        }
      }
      if (lane < nbI) {
        const double xv = (p0 + p1) + (p2 + p3);
        unsigned long long *q = xq + 2 * ((size_t)r * npad + i0 + lane);
        const unsigned long long tag = (unsigned long long)want << 32;
        __hip_atomic_store(q, tag | (unsigned)__double2loint(xv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, tag | (unsigned)__double2hiint(xv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xout[r * ldb + i0 + lane] = xv;
      }
    }
    if (t == nblk - 1) {                                 /* sweep complete: every other block was consumed above */
      __syncthreads();
      if (tid == 0) __hip_atomic_store(tf, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static int trsv_launches(gsl_sinterp_hip_ctx *ctx, size_t n, const double *T, size_t ldt, double *b, double *xout,
                         size_t ldb, int nrhs, int mode, int unit, double *d_inv, int inv_ready)
{
  const size_t nblk = (n + TS - 1) / TS;
  /* single-block systems keep the exact substitution of the reference (cblas/source_trsv_r.h);
     larger ones multiply by the inverted 64x64 diagonal blocks */
  const bool use_inv = nblk > 1 && d_inv != NULL;
  if (use_inv && !inv_ready) {
    hipLaunchKernelGGL(tri_inv_kernel, dim3((unsigned)nblk), dim3(64), 0, ctx->stream, T, ldt, n, mode == 2 ? 1 : 0, unit, d_inv);
    LAUNCH_CHECK(ctx);
  }
  static const bool no_df = getenv("GSL_SINTERP_NO_DATAFLOW_TRSV") && getenv("GSL_SINTERP_NO_DATAFLOW_TRSV")[0] == '1';
  if (use_inv && !no_df && ctx->sk_wgs > 0 && ctx->d_tf && ctx->tf_count >= nblk + 1 && ctx->d_xq) {
    const unsigned G = (unsigned)(nblk < (size_t)ctx->sk_wgs ? nblk : (size_t)ctx->sk_wgs);
    hipLaunchKernelGGL(trsv_dataflow_kernel, dim3(G), dim3(256), 0, ctx->stream, T, ldt, n, (const double *)b, xout, ldb, nrhs, mode,
                       (const double *)d_inv, ctx->d_tf, ctx->d_xq, (unsigned)nblk);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  for (size_t t = 0; t < nblk; t++) {
    const size_t blk = (mode == 0) ? t : nblk - 1 - t;
    const size_t j0 = blk * TS;
    const int nb = (int)((n - j0) < TS ? (n - j0) : TS);
    const size_t rest = (mode == 0) ? n - j0 - nb : j0;
    const unsigned grid = (unsigned)((rest + 63) / 64);
    if (use_inv)
      hipLaunchKernelGGL(trsv_sweep_inv_kernel, dim3(grid ? grid : 1), dim3(256), 0, ctx->stream, T, ldt, n, b, xout, ldb, nrhs,
                         j0, nb, mode, (const double *)d_inv);
    else
      hipLaunchKernelGGL(trsv_sweep_kernel, dim3(grid ? grid : 1), dim3(256), 0, ctx->stream, T, ldt, n, b, xout, ldb, nrhs, j0,
                         nb, mode, unit);
    LAUNCH_CHECK(ctx);
  }
  return ST_SUCCESS;
}

/* inv_ready: the inverted diagonal blocks of this very triangle are still in the context's buffer
   (the backward sweep of a Cholesky solve right after its forward sweep) */
__global__ void __launch_bounds__(256)
vec_copy_kernel(const double *__restrict__ src, double *__restrict__ dst, size_t count)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) dst[i] = src[i];
}

/* b == xout (in place): the right-hand sides are copied to the context's second grow-only buffer by a kernel inside the
   same captured graph, and the sweep reads the copy */
static int trsv_multi_ex(gsl_sinterp_hip_ctx *ctx, size_t n, const double *T, size_t ldt, double *b, double *xout, size_t ldb,
                         int nrhs, int mode, int unit, int inv_ready)
{
  if (n == 0 || nrhs == 0) return ST_SUCCESS;
  if (nrhs > TRSV_MAXR) return sinterp_fail(ctx, ST_EINVAL, "trsv: too many right-hand sides", hipSuccess, __FILE__, __LINE__);
  /* one cached graph per direction: slot 2 forward, slot 3 backward */
  const int slot = mode == 0 ? 2 : 3;
  const size_t key_lda = ((ldt * 8 + (size_t)mode * 2 + (size_t)unit) * 8 + (size_t)nrhs) ^ (ldb << 40) ^ ((size_t)inv_ready << 62);
  const void *p1 = (const void *)((uintptr_t)b ^ ((uintptr_t)xout << 1));
  int replayed = 0;
  int st = sinterp_graph_try_launch(ctx, slot, n, key_lda, T, p1, &replayed);
  if (st || replayed) return st;
  void *d_inv = NULL;
  const size_t inv_bytes = ((n + TS - 1) / TS) * TS * TS * sizeof(double);
  const bool in_place = b == xout;
  if (in_place && inv_ready) return sinterp_fail(ctx, ST_EINVAL, "trsv: in-place sweep cannot reuse inverted blocks", hipSuccess, __FILE__, __LINE__);
  /* the copy of an in-place sweep's right-hand sides lives behind the inverted blocks (a buffer whose growth already
     invalidates the cached sweep graphs) */
  st = sinterp_invbuf(ctx, inv_bytes + (in_place ? (size_t)nrhs * ldb * sizeof(double) : 0), &d_inv);
  if (st) return st;
  {
    /* epoch + per-block flags of the dataflow sweep: grown (and zeroed) outside capture only */
    const size_t need = (n + TS - 1) / TS + 1;
    if (need > ctx->tf_count) {
      HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_tf) { HIP_OK(ctx, hipFree(ctx->d_tf)); ctx->d_tf = NULL; ctx->tf_count = 0; }
      const size_t cnt = need < 1024 ? 1024 : need * 2;
      HIP_OK(ctx, hipMalloc((void **)&ctx->d_tf, cnt * sizeof(unsigned)));
      HIP_OK(ctx, hipMemset(ctx->d_tf, 0, cnt * sizeof(unsigned)));
      if (ctx->d_xq) { HIP_OK(ctx, hipFree(ctx->d_xq)); ctx->d_xq = NULL; }
      const size_t xq_bytes = (size_t)TRSV_MAXR * cnt * TS * 2 * sizeof(unsigned long long);   /* epoch 0 everywhere */
      HIP_OK(ctx, hipMalloc((void **)&ctx->d_xq, xq_bytes));
      HIP_OK(ctx, hipMemset(ctx->d_xq, 0, xq_bytes));
      HIP_OK(ctx, hipDeviceSynchronize());
      ctx->tf_count = cnt;
      for (int i = 2; i < 4; i++)                       /* cached sweep graphs hold the old pointer */
        if (ctx->graph[i].exec) { (void)hipGraphExecDestroy(ctx->graph[i].exec); ctx->graph[i].exec = NULL; }
    }
  }
  st = sinterp_streamk_prepare(ctx);                   /* CU count (the dataflow sweep needs co-resident workgroups) */
  if (st) return st;
  double *b_src = in_place ? (double *)((char *)d_inv + inv_bytes) : b;
  hipStream_t saved;
  st = sinterp_capture_begin(ctx, &saved);
  if (st) return st;
  if (in_place) {
    const size_t cnt = (size_t)(nrhs - 1) * ldb + n;
    hipLaunchKernelGGL(vec_copy_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)b, b_src, cnt);
  }
  st = trsv_launches(ctx, n, T, ldt, b_src, xout, ldb, nrhs, mode, unit, (double *)d_inv, inv_ready);
  int st2 = sinterp_capture_end(ctx, saved, slot, n, key_lda, T, p1);
  return st ? st : st2;
}

int sinterp_trsv_multi(gsl_sinterp_hip_ctx *ctx, size_t n, const double *T, size_t ldt, double *b, double *xout, size_t ldb,
                       int nrhs, int mode, int unit)
{
  return trsv_multi_ex(ctx, n, T, ldt, b, xout, ldb, nrhs, mode, unit, 0);
}

int sinterp_trsv(gsl_sinterp_hip_ctx *ctx, size_t n, const double *T, size_t ldt, double *b, double *xout, int mode,
                 int unit)
{
  return sinterp_trsv_multi(ctx, n, T, ldt, b, xout, n, 1, mode, unit);
}

/* X <- (L L^T)^-1 X for nrhs vectors stored at d_x + r*ldx */
int sinterp_cholesky_svx_multi(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_llt, size_t lda, double *d_x, size_t ldx,
                               int nrhs)
{
  EXCLUSIVE_SECTION(ctx);
  void *d_tmp = NULL;
  int st = sinterp_workspace(ctx, (size_t)nrhs * ldx * sizeof(double), &d_tmp);
  if (st) return st;
  st = sinterp_trsv_multi(ctx, n, d_llt, lda, d_x, (double *)d_tmp, ldx, nrhs, 0, 0);   /* L c = b     (cholesky.c:178) */
  if (st) return st;
  return trsv_multi_ex(ctx, n, d_llt, lda, (double *)d_tmp, d_x, ldx, nrhs, 1, 0, 1);   /* L^T x = c   (cholesky.c:181); same inverted blocks */
}

extern "C" int gsl_sinterp_hip_cholesky_svx(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_llt, size_t lda, double *d_x)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  EXCLUSIVE_SECTION(ctx);
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_llt && d_x), ST_EFAULT);
  return sinterp_cholesky_svx_multi(ctx, n, d_llt, lda, d_x, n, 1);
}
