/*
 * lu.hip -- general dense solve for kernels that are not SPD (thin-plate
 * spline: zero diagonal, indefinite).
 *
 * Replaces gsl_linalg_LU_decomp (linalg/lu.c:59-124) and gsl_linalg_LU_svx
 * (linalg/lu.c:166-201).  Same contract: P A = L U in place with partial
 * pivoting, the pivot of a column is the FIRST row attaining max |a| (strict
 * '>' scan, lu.c:82-93), multipliers stored below the diagonal, the permutation
 * returned in gsl_permutation form (row i of PA is row perm[i] of A) with
 * signum = (-1)^swaps; a zero pivot leaves its column untouched (lu.c:105).
 *
 * The reference is an unblocked kij sweep.  Here: recursive panel LU (Toledo):
 *     lu(j0, w):  if w <= 8: pivoted base kernel on the tall panel
 *                 else lu(j0, w/2); swap rows of the right half; U12 <- L11^-1 U12;
 *                      A22 -= A21 U12 (gemm.hip, fp64 MFMA); lu(j0+w/2, w-w/2);
 *                      swap rows of the left half
 * The pivot search stays a per-column reduction (inherent to partial pivoting);
 * everything else is Level-3.
 */
#include "common.h"
#include <math.h>
#include <stdlib.h>

#define LB 8            /* base panel width */
#define LU_THREADS 1024
#define TB 16           /* triangular-solve base */

/* ------------------------------------------------------------------------ */
/* base: tall panel A[j0:n, j0:j0+w], one workgroup, rows strided over threads */
__global__ void __launch_bounds__(LU_THREADS)
lu_base_kernel(double *__restrict__ A, size_t lda, size_t n, size_t j0, int w, int *__restrict__ ipiv)
{
  __shared__ double s_val[LU_THREADS / 64];
  __shared__ unsigned long long s_row[LU_THREADS / 64];
  __shared__ double s_prow[LB];
  __shared__ unsigned long long s_piv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int j = 0; j < w; j++) {
    const size_t col = j0 + j;
    /* ---- pivot search over rows col..n-1: max |a|, first row on ties ---- */
    double best = -1.0;
    unsigned long long brow = ~0ULL;
    for (size_t i = col + tid; i < n; i += LU_THREADS) {
      const double v = fabs(A[i * lda + col]);
      if (v > best) { best = v; brow = i; }        /* ascending i per thread: keeps the first */
    }
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(best, off);
      const unsigned long long orow = __shfl_xor(brow, off);
      if (ov > best || (ov == best && orow < brow)) { best = ov; brow = orow; }
    }
    if (lane == 0) { s_val[wave] = best; s_row[wave] = brow; }
    __syncthreads();
    if (tid == 0) {
      double bv = s_val[0]; unsigned long long br = s_row[0];
      for (int k = 1; k < LU_THREADS / 64; k++)
        if (s_val[k] > bv || (s_val[k] == bv && s_row[k] < br)) { bv = s_val[k]; br = s_row[k]; }
      /* NaN column or empty: keep the diagonal row */
      if (br == ~0ULL) br = col;
      s_piv = br;
      ipiv[col] = (int)br;
    }
    __syncthreads();
    const size_t piv = (size_t)s_piv;
    /* ---- swap rows col <-> piv inside the panel, publish the pivot row ---- */
    if (tid < w) {
      const double a = A[col * lda + j0 + tid];
      const double b = A[piv * lda + j0 + tid];
      if (piv != col) { A[col * lda + j0 + tid] = b; A[piv * lda + j0 + tid] = a; }
      s_prow[tid] = b;
    }
    __syncthreads();
    const double ajj = s_prow[j];
    if (ajj != 0.0) {                               /* lu.c:105 */
      for (size_t i = col + 1 + tid; i < n; i += LU_THREADS) {
        double *row = A + i * lda + j0;
        const double l = row[j] / ajj;
        row[j] = l;
        for (int k = j + 1; k < w; k++) row[k] = row[k] - l * s_prow[k];
      }
    }
    __syncthreads();
  }
}


/* ------------------------------------------------------------------------ */
/* base, register-resident: the whole tall panel (<= 1024*R rows x 8 columns) lives
   in the VGPRs of ONE workgroup for the duration of its 8 column steps, so a
   column step costs two workgroup barriers instead of round trips through L2.
   Row i of the panel belongs to thread (i - j0) % 1024, slot (i - j0) / 1024. */
template <int R, int NTH = LU_THREADS>
__global__ void __launch_bounds__(NTH)
lu_base_reg_kernel(double *__restrict__ A, size_t lda, size_t n, size_t j0, int w, int *__restrict__ ipiv)
{
  __shared__ double s_val[NTH / 64];
  __shared__ unsigned int s_row[NTH / 64];
  __shared__ double s_prow[LB], s_crow[LB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  double a[R][LB];
#pragma unroll
  for (int s = 0; s < R; s++) {
    const size_t row = j0 + tid + (size_t)NTH * s;
    const double *p = A + row * lda + j0;
    if (row < n && w == LB && ((((uintptr_t)p) & 15) == 0)) {
#pragma unroll
      for (int k = 0; k < LB; k += 2) { double2 t = *reinterpret_cast<const double2 *>(p + k); a[s][k] = t.x; a[s][k + 1] = t.y; }
    } else {
#pragma unroll
      for (int k = 0; k < LB; k++) a[s][k] = (row < n && k < w) ? p[k] : 0.0;
    }
  }

  const unsigned n32 = (unsigned)n, j032 = (unsigned)j0;       /* n < 2^31 (ipiv is int) */
#pragma unroll
  for (int j = 0; j < LB; j++) {
    if (j < w) {                                    /* w is uniform */
      const unsigned col = j032 + (unsigned)j;
      /* row indices are re-derived from a laundered thread id in every column step: kept live across the eight
         unrolled steps (with the 64 panel registers of R = 4) they were spilled and reloaded on the dependent path */
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const unsigned rbase = j032 + (unsigned)tl;
      /* pivot search: max |a|, smallest row on ties (lu.c:82-93) */
      double best = -1.0;
      unsigned brow = 0xffffffffu;
#pragma unroll
      for (int s = 0; s < R; s++) {
        const unsigned row = rbase + (unsigned)NTH * s;
        const double v = fabs(a[s][j]);
        if (row < n32 && row >= col && v > best) { best = v; brow = row; }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off);
        const unsigned orow = __shfl_xor(brow, off);
        if (ov > best || (ov == best && orow < brow)) { best = ov; brow = orow; }
      }
      if (lane == 0) { s_val[wave] = best; s_row[wave] = brow; }
      __syncthreads();
      /* second level: lane k of every wave takes wave k's candidate, a 4-step butterfly finishes it (one LDS round
         trip + shuffles; read one after the other by a scalarised loop the NTH / 64 candidates cost 16 dependent LDS
         round trips per column step).  Same comparator: larger |a|, smaller row on ties; candidates are never NaN */
      constexpr int NWV = NTH / 64;
      double bv = lane < NWV ? s_val[lane] : -1.0;
      unsigned piv = lane < NWV ? s_row[lane] : 0xffffffffu;
#pragma unroll
      for (int off = NWV / 2; off > 0; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const unsigned orow = __shfl_xor(piv, off);
        if (ov > bv || (ov == bv && orow < piv)) { bv = ov; piv = orow; }
      }
      piv = (unsigned)__builtin_amdgcn_readfirstlane((int)piv);
      if (piv == 0xffffffffu) piv = col;            /* NaN column: keep the diagonal row */
      if (tid == 0) ipiv[col] = (int)piv;
      /* publish the pivot row and the current row, then swap them */
      const unsigned own_p = (piv - j032) % NTH, slot_p = (piv - j032) / NTH;
      const unsigned own_c = (unsigned)j % NTH;       /* col - j0 = j < 1024: slot 0 */
#pragma unroll
      for (int s = 0; s < R; s++)
        if ((unsigned)tl == own_p && (unsigned)s == slot_p) {
#pragma unroll
          for (int k = 0; k < LB; k++) s_prow[k] = a[s][k];
        }
      if ((unsigned)tl == own_c) {
#pragma unroll
        for (int k = 0; k < LB; k++) s_crow[k] = a[0][k];
      }
      __syncthreads();
      if (piv != col) {
#pragma unroll
        for (int s = 0; s < R; s++)
          if ((unsigned)tl == own_p && (unsigned)s == slot_p) {
#pragma unroll
            for (int k = 0; k < LB; k++) a[s][k] = s_crow[k];
          }
        if ((unsigned)tl == own_c) {
#pragma unroll
          for (int k = 0; k < LB; k++) a[0][k] = s_prow[k];
        }
      }
      const double ajj = s_prow[j];
      if (ajj != 0.0) {                              /* lu.c:105 */
#pragma unroll
        for (int s = 0; s < R; s++) {
          const unsigned row = rbase + (unsigned)NTH * s;
          if (row < n32 && row > col) {
            const double l = a[s][j] / ajj;
            a[s][j] = l;
#pragma unroll
            for (int k = j + 1; k < LB; k++) a[s][k] = a[s][k] - l * s_prow[k];
          }
          /* one row at a time: interleaved, the IEEE divide expansions of all R rows keep ~10 temporaries each live */
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      /* s_val / s_prow are rewritten only after the next iteration's first barrier
         resp. between its two barriers: no third barrier needed */
    }
  }

#pragma unroll
  for (int s = 0; s < R; s++) {
    const size_t row = j0 + tid + (size_t)NTH * s;
    if (row >= n) continue;
    double *p = A + row * lda + j0;
    if (w == LB && ((((uintptr_t)p) & 15) == 0)) {
#pragma unroll
      for (int k = 0; k < LB; k += 2) *reinterpret_cast<double2 *>(p + k) = make_double2(a[s][k], a[s][k + 1]);
    } else {
#pragma unroll
      for (int k = 0; k < LB; k++) if (k < w) p[k] = a[s][k];
    }
  }
}


/* ------------------------------------------------------------------------ */
/* Block-panel kernel: ONE workgroup (512 threads, 2 waves per SIMD) factors a whole
   block panel A[j0:n, j0:j0+wb], wb <= 64, of at most 512*R rows.
     - thread t owns physical rows j0 + t + 512*s (s < R); rows never move while the
       block is being factored: each row carries its current POSITION (what its row
       index would be after the reference's swaps, lu.c:95-101) in a register, which
       is also the tie-break key of the pivot search (first row attaining the max).
     - the block is swept right-looking in slabs of 8 columns held in registers:
       8 pivoted column steps (two workgroup barriers each), then U12 of the slab
       (8 x remaining columns, forward substitution) and the rank-8 update of the
       remaining columns of every active row.
     - at the end rows whose position differs from their physical row (<= 2*wb of
       them) are moved through LDS, so the block leaves in the reference's layout
       and ipiv holds the reference's swap sequence for the outer laswp kernels.   */
#define BW 64
#define BT 512

template <int R>
__global__ void __launch_bounds__(BT)
lu_block_kernel(double *__restrict__ A, size_t lda, size_t n, size_t j0, int wb, int *__restrict__ ipiv)
{
  __shared__ double s_val[BT / 64];
  __shared__ unsigned s_pos[BT / 64];
  __shared__ double s_prow[8];
  __shared__ double s_Lss[8][8];
  __shared__ double s_U[8][BW];
  __shared__ unsigned s_pivphys[8];
  __shared__ int s_nmoved;
  __shared__ unsigned s_dest[2 * BW];
  __shared__ double s_stage[2 * BW][BW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a[R][8];
  unsigned pos[R];
  unsigned retired = 0;
#pragma unroll
  for (int s = 0; s < R; s++) pos[s] = (unsigned)(j0 + tid + (size_t)BT * s);
  if (tid == 0) s_nmoved = 0;

  const int nsub = (wb + 7) / 8;
  for (int sub = 0; sub < nsub; sub++) {
    const size_t c0 = j0 + 8 * (size_t)sub;
    const int ws = (wb - 8 * sub) < 8 ? (wb - 8 * sub) : 8;
    /* ---- slab into registers */
#pragma unroll
    for (int s = 0; s < R; s++) {
      const size_t row = j0 + tid + (size_t)BT * s;
      const double *p = A + row * lda + c0;
      if (row < n && ws == 8 && ((((uintptr_t)p) & 15) == 0)) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) { const double2 t = *reinterpret_cast<const double2 *>(p + k); a[s][k] = t.x; a[s][k + 1] = t.y; }
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) a[s][k] = (row < n && k < ws) ? p[k] : 0.0;
      }
    }
    /* ---- 8 pivoted column steps */
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (j < ws) {
        const unsigned col = (unsigned)(c0 + j);
        double best = -1.0;
        unsigned bpos = 0xffffffffu;
#pragma unroll
        for (int s = 0; s < R; s++) {
          const size_t row = j0 + tid + (size_t)BT * s;
          const double v = fabs(a[s][j]);
          const bool act = row < n && !((retired >> s) & 1u);
          if (act && (v > best || (v == best && pos[s] < bpos))) { best = v; bpos = pos[s]; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const double ov = __shfl_xor(best, off);
          const unsigned op = __shfl_xor(bpos, off);
          if (ov > best || (ov == best && op < bpos)) { best = ov; bpos = op; }
        }
        if (lane == 0) { s_val[wave] = best; s_pos[wave] = bpos; }
        __syncthreads();
        double bv = s_val[0];
        unsigned q = s_pos[0];
#pragma unroll
        for (int k = 1; k < BT / 64; k++) {
          const double ov = s_val[k];
          const unsigned op = s_pos[k];
          if (ov > bv || (ov == bv && op < q)) { bv = ov; q = op; }
        }
        if (q == 0xffffffffu) q = col;               /* all-NaN column: keep the row in place (lu.c:82-93) */
        /* the pivot row publishes its slab, takes position col; the row that sat at col moves to q.
           Exactly one thread of the workgroup owns the pivot: find its slot cheaply, then run the
           publish code behind a branch so the other 255 threads skip it. */
        int sl = -1;
#pragma unroll
        for (int s = 0; s < R; s++) {
          const size_t row = j0 + tid + (size_t)BT * s;
          const bool act = row < n && !((retired >> s) & 1u);
          if (act && pos[s] == q) sl = s;
          else if (act && pos[s] == col) pos[s] = q;   /* displaced row (only exists when q != col) */
        }
        if (sl >= 0) {
#pragma unroll
          for (int s = 0; s < R; s++) {
            if (s == sl) {
#pragma unroll
              for (int k = 0; k < 8; k++) s_prow[k] = a[s][k];
#pragma unroll
              for (int m = 0; m < 8; m++) s_Lss[j][m] = (m < j) ? a[s][m] : 0.0;
              pos[s] = col;
            }
          }
          s_pivphys[j] = (unsigned)(j0 + tid + (size_t)BT * sl);
          ipiv[col] = (int)q;
          retired |= 1u << sl;
        }
        __syncthreads();
        const double ajj = s_prow[j];
        if (ajj != 0.0) {                             /* lu.c:105 */
          const double rcp = 1.0 / ajj;
#pragma unroll
          for (int s = 0; s < R; s++) {
            const size_t row = j0 + tid + (size_t)BT * s;
            if (row < n && !((retired >> s) & 1u)) {
              const double qq = a[s][j] * rcp;         /* l = a/ajj, correctly rounded via one residual step */
              const double l = fma(fma(-ajj, qq, a[s][j]), rcp, qq);
              a[s][j] = l;
#pragma unroll
              for (int k = j + 1; k < 8; k++) a[s][k] = fma(-l, s_prow[k], a[s][k]);
            }
          }
        }
      }
    }
    /* ---- slab back to memory (multipliers, and U entries of the retired rows) */
#pragma unroll
    for (int s = 0; s < R; s++) {
      const size_t row = j0 + tid + (size_t)BT * s;
      if (row >= n) continue;
      double *p = A + row * lda + c0;
      if (ws == 8 && ((((uintptr_t)p) & 15) == 0)) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) *reinterpret_cast<double2 *>(p + k) = make_double2(a[s][k], a[s][k + 1]);
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < ws) p[k] = a[s][k];
      }
    }
    /* ---- remaining columns of the block: U12 of this slab, then the rank-8 update */
    const int nr = wb - 8 * sub - ws;
    if (nr > 0) {                                     /* ws == 8 here */
      __syncthreads();                                /* s_pivphys / s_Lss complete */
      if (tid < nr) {
        double u[8];
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = A[(size_t)s_pivphys[i] * lda + c0 + 8 + tid];
#pragma unroll
        for (int i = 1; i < 8; i++)
#pragma unroll
          for (int m = 0; m < i; m++) u[i] = fma(-s_Lss[i][m], u[m], u[i]);
#pragma unroll
        for (int i = 0; i < 8; i++) { s_U[i][tid] = u[i]; A[(size_t)s_pivphys[i] * lda + c0 + 8 + tid] = u[i]; }
      }
      __syncthreads();
      /* rows are processed G at a time so that G x 4 16-byte loads are in flight per lane
         before the first FMA needs its data (the update is latency-bound otherwise) */
      constexpr int G = R < 4 ? R : 4;
#pragma unroll
      for (int g0 = 0; g0 < R; g0 += G) {
        bool act[G];
        double *p[G];
        bool any = false;
#pragma unroll
        for (int gi = 0; gi < G; gi++) {
          const size_t row = j0 + tid + (size_t)BT * (g0 + gi);
          act[gi] = row < n && !((retired >> (g0 + gi)) & 1u);
          p[gi] = A + row * lda + c0 + 8;
          any |= act[gi];
        }
        if (!any) continue;
        const bool vec = ((((uintptr_t)p[0]) & 15) == 0) && ((lda & 1) == 0);
        for (int cc = 0; cc < nr; cc += 8) {
          double x[G][8];
          const bool full = vec && (cc + 8 <= nr);
#pragma unroll
          for (int gi = 0; gi < G; gi++) {
            if (act[gi] && full) {
#pragma unroll
              for (int c = 0; c < 8; c += 2) { const double2 t = *reinterpret_cast<const double2 *>(p[gi] + cc + c); x[gi][c] = t.x; x[gi][c + 1] = t.y; }
            } else {
#pragma unroll
              for (int c = 0; c < 8; c++) x[gi][c] = (act[gi] && cc + c < nr) ? p[gi][cc + c] : 0.0;
            }
          }
#pragma unroll
          for (int k = 0; k < 8; k++) {
            double uk[8];
#pragma unroll
            for (int c = 0; c < 8; c++) uk[c] = s_U[k][(cc + c) & (BW - 1)];
#pragma unroll
            for (int gi = 0; gi < G; gi++)
#pragma unroll
              for (int c = 0; c < 8; c++) x[gi][c] = fma(-a[g0 + gi][k], uk[c], x[gi][c]);
          }
#pragma unroll
          for (int gi = 0; gi < G; gi++) {
            if (!act[gi]) continue;
            if (full) {
#pragma unroll
              for (int c = 0; c < 8; c += 2) *reinterpret_cast<double2 *>(p[gi] + cc + c) = make_double2(x[gi][c], x[gi][c + 1]);
            } else {
#pragma unroll
              for (int c = 0; c < 8; c++) if (cc + c < nr) p[gi][cc + c] = x[gi][c];
            }
          }
        }
      }
    }
    __syncthreads();                                  /* U rows visible to their owners' next slab load */
  }

  /* ---- move the displaced rows (block columns only; the outer laswp handles the rest) */
#pragma unroll
  for (int s = 0; s < R; s++) {
    const size_t row = j0 + tid + (size_t)BT * s;
    if (row < n && pos[s] != (unsigned)row) {
      const int slot = atomicAdd(&s_nmoved, 1);
      s_dest[slot] = pos[s];
      const double *p = A + row * lda + j0;
      for (int c = 0; c < wb; c++) s_stage[slot][c] = p[c];
    }
  }
  __syncthreads();
  const int nmoved = s_nmoved;
  for (int e = tid; e < nmoved * wb; e += BT) {
    const int slot = e / wb, c = e % wb;
    A[(size_t)s_dest[slot] * lda + j0 + c] = s_stage[slot][c];
  }
}

/* apply the row interchanges k = k0..k1-1 (row k <-> ipiv[k]) to columns c0..c0+nc-1 */
__global__ void __launch_bounds__(256)
laswp_kernel(double *__restrict__ A, size_t lda, size_t c0, size_t nc, const int *__restrict__ ipiv, size_t k0, size_t k1)
{
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nc) return;
  double *col = A + c0 + c;
  for (size_t k = k0; k < k1; k++) {
    const size_t p = (size_t)ipiv[k];
    if (p != k) {
      const double a = col[k * lda], b = col[p * lda];
      col[k * lda] = b; col[p * lda] = a;
    }
  }
}

/* B <- L^-1 B, L unit lower nb x nb (nb <= 32) at A[r0.., r0..], B = A[r0:r0+nb, c0:c0+nc];
   one column of B per thread, kept in registers */
__global__ void __launch_bounds__(256)
trsm_unit_lower_base_kernel(double *__restrict__ A, size_t lda, size_t r0, int nb, size_t c0, size_t nc)
{
  __shared__ double sL[TB][TB + 1];
  for (int e = threadIdx.x; e < TB * TB; e += 256) {
    const int r = e / TB, k = e % TB;
    sL[r][k] = (r < nb && k < r) ? A[(r0 + r) * lda + r0 + k] : 0.0;
  }
  __syncthreads();
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nc) return;
  double *col = A + r0 * lda + c0 + c;
  double x[TB];
#pragma unroll
  for (int r = 0; r < TB; r++) x[r] = r < nb ? col[(size_t)r * lda] : 0.0;
#pragma unroll
  for (int r = 1; r < TB; r++) {
    double v = x[r];
#pragma unroll
    for (int k = 0; k < r; k++) v -= sL[r][k] * x[k];
    x[r] = v;
  }
#pragma unroll
  for (int r = 0; r < TB; r++) if (r < nb) col[(size_t)r * lda] = x[r];
}

static int trsm_unit_lower(gsl_sinterp_hip_ctx *ctx, double *A, size_t lda, size_t r0, size_t nb, size_t c0, size_t nc)
{
  if (nb == 0 || nc == 0) return ST_SUCCESS;
  if (nb <= TB) {
    hipLaunchKernelGGL(trsm_unit_lower_base_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, A, lda,
                       r0, (int)nb, c0, nc);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  size_t n1 = ((nb / 2 + TB - 1) / TB) * TB;
  if (n1 >= nb) n1 = nb - TB;
  int st = trsm_unit_lower(ctx, A, lda, r0, n1, c0, nc);
  if (st) return st;
  /* B2 -= L21 * X1 :  L21 = A[r0+n1 : r0+nb, r0 : r0+n1],  X1 = A[r0 : r0+n1, c0 : c0+nc] */
  st = sinterp_gemm_minus(ctx, nb - n1, nc, n1, A + (r0 + n1) * lda + r0, lda, A + r0 * lda + c0, lda, 1,
                          A + (r0 + n1) * lda + c0, lda, 0);
  if (st) return st;
  return trsm_unit_lower(ctx, A, lda, r0 + n1, nb - n1, c0, nc);
}

static int lu_panel(gsl_sinterp_hip_ctx *ctx, double *A, size_t lda, size_t n, size_t j0, size_t w, int *d_ipiv)
{
  const size_t prow = n - j0;
  /* measured on MI355X: the single-workgroup block kernel wins while the panel is short
     (<= 1024 rows: its dependent global round trips are few); taller panels go through the
     recursion down to 8-column register-resident panels */
  /* developer override (tools/time_lu.py): tallest panel the single-workgroup block kernel takes */
  static const size_t block_rows = getenv("GSL_SINTERP_LU_BLOCK_ROWS") ? (size_t)atol(getenv("GSL_SINTERP_LU_BLOCK_ROWS")) : (size_t)BT * 2;
  if (w <= BW && prow <= block_rows && prow <= (size_t)BT * 8) {
    if (prow <= (size_t)BT)
      hipLaunchKernelGGL(lu_block_kernel<1>, dim3(1), dim3(BT), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv);
    else if (prow <= (size_t)BT * 2)
      hipLaunchKernelGGL(lu_block_kernel<2>, dim3(1), dim3(BT), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv);
    else if (prow <= (size_t)BT * 4)
      hipLaunchKernelGGL(lu_block_kernel<4>, dim3(1), dim3(BT), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv);
    else
      hipLaunchKernelGGL(lu_block_kernel<8>, dim3(1), dim3(BT), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  if (w <= LB) {
    const size_t rows = n - j0;
    if (rows <= (size_t)LU_THREADS)
      hipLaunchKernelGGL(lu_base_reg_kernel<1>, dim3(1), dim3(LU_THREADS), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv);
    else if (rows <= (size_t)LU_THREADS * 2)
      hipLaunchKernelGGL(lu_base_reg_kernel<2>, dim3(1), dim3(LU_THREADS), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv);
    else if (rows <= (size_t)LU_THREADS * 4) {
      /* 512 threads x 8 rows: with 1024 threads the 64 panel registers of 4 rows per thread do not fit the 128-VGPR
         budget of a 16-wave workgroup (scratch on the dependent path), and the per-wave work of a column step
         (reductions, pivot-row exchange, scalar control) is paid by twice as many waves */
      static const int th = getenv("GSL_SINTERP_LU_BASE_THREADS") ? atoi(getenv("GSL_SINTERP_LU_BASE_THREADS")) : 512;
      if (th == 1024)
        hipLaunchKernelGGL(lu_base_reg_kernel<4>, dim3(1), dim3(LU_THREADS), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv);
      else
        hipLaunchKernelGGL((lu_base_reg_kernel<8, 512>), dim3(1), dim3(512), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv);
    }
    else   /* taller than the register file of one CU: panel stays in L2 */
      hipLaunchKernelGGL(lu_base_kernel, dim3(1), dim3(LU_THREADS), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  const size_t unit = (prow <= block_rows && prow <= (size_t)BT * 8 && w > BW) ? BW : LB;   /* split on block-kernel boundaries */
  size_t w1 = ((w / 2 + unit - 1) / unit) * unit;
  if (w1 >= w) w1 = w - unit;
  const size_t w2 = w - w1, c1 = j0 + w1;
  int st = lu_panel(ctx, A, lda, n, j0, w1, d_ipiv);
  if (st) return st;
  hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((w2 + 255) / 256)), dim3(256), 0, ctx->stream, A, lda, c1, w2, d_ipiv, j0, c1);
  LAUNCH_CHECK(ctx);
  st = trsm_unit_lower(ctx, A, lda, j0, w1, c1, w2);             /* U12 = L11^-1 A12 */
  if (st) return st;
  st = sinterp_gemm_minus(ctx, n - c1, w2, w1, A + c1 * lda + j0, lda, A + j0 * lda + c1, lda, 1,
                          A + c1 * lda + c1, lda, 0);           /* A22 -= A21 U12 */
  if (st) return st;
  st = lu_panel(ctx, A, lda, n, c1, w2, d_ipiv);
  if (st) return st;
  hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((w1 + 255) / 256)), dim3(256), 0, ctx->stream, A, lda, j0, w1, d_ipiv, c1,
                     c1 + w2 < n ? c1 + w2 : n);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_lu_decomp(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *d_perm,
                                         int *h_signum)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  EXCLUSIVE_SECTION(ctx);
  REQUIRE(ctx, lda >= n && n < 2147483647ULL, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_a && d_perm), ST_EFAULT);
  if (h_signum) *h_signum = 1;
  if (n == 0) return ST_SUCCESS;
  int replayed = 0;
  int st = sinterp_streamk_prepare(ctx);               /* the N.N updates run on the stream-K kernel (buffers: outside capture) */
  if (st) return st;
  st = sinterp_graph_try_launch(ctx, 1, n, lda, d_a, d_perm, &replayed);
  if (st) return st;
  if (!replayed) {
    hipStream_t saved;
    st = sinterp_capture_begin(ctx, &saved);
    if (st) return st;
    st = lu_panel(ctx, d_a, lda, n, 0, n, d_perm);                /* d_perm holds LAPACK-style ipiv for now */
    int st2 = sinterp_capture_end(ctx, saved, 1, n, lda, d_a, d_perm);
    if (st) return st;
    if (st2) return st2;
  }
  /* ipiv -> gsl_permutation content + signum (lu.c:95-101) */
  int *h_ipiv = (int *)malloc(n * sizeof(int));
  int *h_perm = (int *)malloc(n * sizeof(int));
  if (!h_ipiv || !h_perm) { free(h_ipiv); free(h_perm); return sinterp_fail(ctx, ST_ENOMEM, "lu_decomp: host buffers", hipSuccess, __FILE__, __LINE__); }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(h_ipiv, d_perm, n * sizeof(int), hipMemcpyDeviceToHost);
  int sign = 1;
  if (e == hipSuccess) {
    for (size_t i = 0; i < n; i++) h_perm[i] = (int)i;
    for (size_t k = 0; k < n; k++) {
      const int p = h_ipiv[k];
      if (p != (int)k && p >= 0 && (size_t)p < n) { int t = h_perm[k]; h_perm[k] = h_perm[p]; h_perm[p] = t; sign = -sign; }
    }
    e = hipMemcpy(d_perm, h_perm, n * sizeof(int), hipMemcpyHostToDevice);
  }
  free(h_ipiv); free(h_perm);
  if (e != hipSuccess) return sinterp_fail(ctx, ST_EFAILED, "lu_decomp: pivot transfer", e, __FILE__, __LINE__);
  if (h_signum) *h_signum = sign;
  return ST_SUCCESS;
}

__global__ void permute_gather_kernel(const double *__restrict__ src, const int *__restrict__ perm, double *__restrict__ dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];                 /* permutation/permute_source.c:140 */
}

__global__ void lu_singular_kernel(const double *__restrict__ lu, size_t lda, size_t n, int *__restrict__ flag)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && lu[i * lda + i] == 0.0) atomicExch(flag, 1);   /* linear_simplex_util.h:14-26 / lu.c:181 */
}

extern "C" int gsl_sinterp_hip_lu_svx(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_lu, size_t lda,
                                      const int *d_perm, double *d_x)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  EXCLUSIVE_SECTION(ctx);
  REQUIRE(ctx, lda >= n, ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_lu && d_perm && d_x), ST_EFAULT);
  if (n == 0) return ST_SUCCESS;
  int *d_flag = (int *)ctx->d_scratch + 16;
  HIP_OK(ctx, hipMemsetAsync(d_flag, 0, sizeof(int), ctx->stream));
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(lu_singular_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_lu, lda, n, d_flag);
  LAUNCH_CHECK(ctx);
  int flag = 0;
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost));
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  if (flag) return sinterp_fail(ctx, ST_EDOM, "lu_svx: matrix is singular", hipSuccess, __FILE__, __LINE__);

  void *d_tmp = NULL;
  int st = sinterp_workspace(ctx, n * sizeof(double), &d_tmp);
  if (st) return st;
  double *tmp = (double *)d_tmp;
  hipLaunchKernelGGL(permute_gather_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_x, d_perm, tmp, n);   /* tmp = P b */
  LAUNCH_CHECK(ctx);
  st = sinterp_trsv(ctx, n, d_lu, lda, tmp, d_x, 0, 1);     /* L c = P b, unit lower: solved blocks -> d_x   */
  if (st) return st;
  /* back substitution reads d_x as right-hand side, writes solved blocks to tmp */
  st = sinterp_trsv(ctx, n, d_lu, lda, d_x, tmp, 2, 0);     /* U x = c */
  if (st) return st;
  HIP_OK(ctx, hipMemcpyAsync(d_x, tmp, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return ST_SUCCESS;
}
