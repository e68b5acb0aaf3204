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

#ifdef SINTERP_DIAG_PROF
__device__ unsigned long long g_lu_ts[128];
#define LU_TSTAMP(i) do { if (threadIdx.x == 0 && j0 == 0) g_lu_ts[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int gsl_sinterp_hip_debug_lu_ts(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lu_ts), sizeof(unsigned long long) * 128); }
__device__ unsigned long long g_lc_ts[6 * 64 + 2];
#define LC_TSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && j0 == 0) g_lc_ts[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int gsl_sinterp_hip_debug_lc_ts(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lc_ts), sizeof(unsigned long long) * (6 * 64 + 2)); }
#else
#define LU_TSTAMP(i) do { } while (0)
#define LC_TSTAMP(i) do { } while (0)
#endif
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
      double v = fabs(A[i * lda + col]);
      if (i == col && v != v) v = INFINITY;        /* a NaN diagonal keeps itself: nothing compares greater (lu.c:82-93) */
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

  LU_TSTAMP(0);
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
        double v = fabs(a[s][j]);
        if (row == col && v != v) v = INFINITY;     /* a NaN diagonal keeps itself (lu.c:82-93) */
        if (row < n32 && row >= col && v > best) { best = v; brow = row; }
      }
      LU_TSTAMP(2 + 6 * j);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off);
        const unsigned orow = __shfl_xor(brow, off);
        if (ov > best || (ov == best && orow < brow)) { best = ov; brow = orow; }
      }
      LU_TSTAMP(3 + 6 * j);
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
      LU_TSTAMP(4 + 6 * j);
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
      LU_TSTAMP(5 + 6 * j);
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
      LU_TSTAMP(6 + 6 * j);
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
      LU_TSTAMP(7 + 6 * j);
      /* s_val / s_prow are rewritten only after the next iteration's first barrier
         resp. between its two barriers: no third barrier needed */
    }
  }

  LU_TSTAMP(60);
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
  LU_TSTAMP(61);
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
          double v = fabs(a[s][j]);
          if (pos[s] == col && v != v) v = INFINITY;   /* a NaN diagonal keeps itself (lu.c:82-93) */
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

/* ------------------------------------------------------------------------ */
/* Cooperative panel kernel (round 4): G = ceil(rows / 256) workgroups factor ONE 64-wide panel A[j0:n, j0:j0+wb] together.
   A single workgroup cannot feed a tall panel (measured: loading 4096 rows x 64 B takes one CU 20 us -- ~200 outstanding
   lines per microsecond --, a third of the 8-column base kernel's 63 us, the store another fifth), and the recursion between
   8-column panels costs ~3500 launches per factorisation.  Here every thread keeps ONE row of the panel (64 doubles) in
   registers for all 64 column steps, rows never move (each carries its POSITION, i.e. its row index after the reference's
   swaps, lu.c:95-101; the copy-back writes a row where its position says), and a column step is
       local candidate (DPP wave reduction, one barrier)  ->  ONE grid-wide exchange  ->  rank-1 update in registers.
   Exchange: every workgroup publishes its candidate {|a|, position, the candidate row's 64 values} into its slot (parity
   double-buffered) as 8-byte words {tag | 32 payload bits} with agent-scope stores; every workgroup polls ALL slots (the
   rows speculatively with the headers: one memory round trip per column, no flag-then-data) and picks the same winner
   with the reference's rule -- larger |a|, first position on ties, NaN never beats the diagonal row, a NaN diagonal keeps
   itself (lu.c:82-93, strict '>').  tag = generation * 64 + column + 1; the generation counter is advanced by workgroup 0
   when the panel is done, so a replayed graph never mistakes an old word for a new one.  Two slots per workgroup suffice: a
   workgroup can only publish column j + 2 after every other one has published j + 1, i.e. has finished reading j.
   Deadlock: the G <= 64 workgroups (one per CU, ~300 VGPRs) are co-resident; every poll loop is bounded and a time-out
   raises a sticky abort word that ends all later waits at once (the host then reports GSL_EFAILED). */
#define LC_W 64
#define LC_ROWS 256
#define LC_GMAX 64
#define LC_SLOT_WORDS 136                      /* 128 row words + 3 header words, padded */
#define LC_POLL_MAX (1 << 22)

__device__ __forceinline__ void lc_better(double &k, unsigned &p, double ok, unsigned op)
{
  if (ok > k || (ok == k && op < p)) { k = ok; p = op; }
}
template <int CTRL, int RMASK>
__device__ __forceinline__ void lc_dpp_step(double &k, unsigned &p)
{
  const int lo = __double2loint(k), hi = __double2hiint(k);
  const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, RMASK, 0xF, false);
  const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, RMASK, 0xF, false);
  const unsigned op = (unsigned)__builtin_amdgcn_update_dpp((int)p, (int)p, CTRL, RMASK, 0xF, false);
  lc_better(k, p, __hiloint2double(ohi, olo), op);
}
/* wave-wide argmax of (key, smaller position on ties); the result is uniform */
__device__ __forceinline__ void lc_wave_argmax(double &k, unsigned &p)
{
  lc_dpp_step<0xB1, 0xF>(k, p);                /* quad_perm [1,0,3,2] */
  lc_dpp_step<0x4E, 0xF>(k, p);                /* quad_perm [2,3,0,1] */
  lc_dpp_step<0x141, 0xF>(k, p);               /* row_half_mirror */
  lc_dpp_step<0x140, 0xF>(k, p);               /* row_mirror: all 16 lanes of a row agree */
  lc_dpp_step<0x142, 0xA>(k, p);               /* row_bcast15 into rows 1, 3 */
  lc_dpp_step<0x143, 0xC>(k, p);               /* row_bcast31 into rows 2, 3: lane 63 holds the wave's result */
  const int lo = __builtin_amdgcn_readlane(__double2loint(k), 63), hi = __builtin_amdgcn_readlane(__double2hiint(k), 63);
  k = __hiloint2double(hi, lo);
  p = (unsigned)__builtin_amdgcn_readlane((int)p, 63);
}

/* winner of four (key, position) candidates, all eight values loaded before the first comparison (the sequential
   "if better" form compiled to four dependent LDS round trips with branches in between) */
__device__ __forceinline__ void lc_pick4(const double *k, const unsigned *p, double &bk, unsigned &bp, int &bw)
{
  const double k0 = k[0], k1 = k[1], k2 = k[2], k3 = k[3];
  const unsigned p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
  const bool b01 = k1 > k0 || (k1 == k0 && p1 < p0);
  const double ka = b01 ? k1 : k0; const unsigned pa = b01 ? p1 : p0; const int wa = b01 ? 1 : 0;
  const bool b23 = k3 > k2 || (k3 == k2 && p3 < p2);
  const double kb = b23 ? k3 : k2; const unsigned pb = b23 ? p3 : p2; const int wb = b23 ? 3 : 2;
  const bool bb = kb > ka || (kb == ka && pb < pa);
  bk = bb ? kb : ka; bp = bb ? pb : pa; bw = bb ? wb : wa;
}

struct LcShared {
  double wrow[4][LC_W];                        /* the four waves' candidate rows */
  double cand[4][LC_W];                        /* after the exchange: each wave's best polled row */
  __attribute__((aligned(16))) double wkey[4], ckey[4];
  __attribute__((aligned(16))) unsigned wpos[4], cpos[4];
};

template <int J>
__device__ __forceinline__ void lc_step(double (&a)[LC_W], unsigned &pos, const bool valid, const unsigned j0, const int wb, LcShared &sh, const unsigned bid,
                                        unsigned long long *__restrict__ slots, unsigned *__restrict__ ctl, const unsigned tagbase,
                                        const unsigned G, int *__restrict__ ipiv, bool &dead)
{
  if constexpr (J < LC_W) {
    /* Agent scope throughout.  Tried: all workers on ONE XCD (every 8th block of an 8 x larger grid works; XCC_IDs exchanged and
       compared at kernel start) with workgroup-scope (sc0) stores / loads so that the XCD's L2 would be the point of coherence
       instead of memory -- the polls never saw the other CUs' stores (with and without `buffer_inv sc0` before each poll): every
       exchange timed out.  Removed. */
    constexpr int SC = __HIP_MEMORY_SCOPE_AGENT;
    if (J < wb) {                                           /* wb is uniform */
      const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
      const unsigned c = j0 + (unsigned)J, tag = tagbase + (unsigned)J + 1u;
      LC_TSTAMP(6 * J);
      /* ---- local candidate ---- */
      const double v = fabs(a[J]);
      const bool active = valid && pos >= c;
      double key = !active ? -2.0 : (pos == c ? (v != v ? INFINITY : v) : (v == v ? v : -1.0));
      unsigned kp = active ? pos : 0xffffffffu;
      const unsigned mypos = kp;
      lc_wave_argmax(key, kp);
      if (active && mypos == kp) {                          /* this wave's candidate row (positions are unique) */
#pragma unroll
        for (int k = 0; k < LC_W; k += 2) *reinterpret_cast<double2 *>(&sh.wrow[wave][k]) = make_double2(a[k], a[k + 1]);
      }
      if (lane == 0) { sh.wkey[wave] = key; sh.wpos[wave] = kp; }
      __syncthreads();
      LC_TSTAMP(6 * J + 1);
      /* ---- publish the workgroup's candidate: wave 0 the row, wave 1 the header (the other waves go straight to the poll) ---- */
      if (wave < 2) {
        double bk; unsigned bp; int bw;
        lc_pick4(sh.wkey, sh.wpos, bk, bp, bw);
        unsigned long long *slot = slots + ((size_t)(J & 1) * LC_GMAX + bid) * LC_SLOT_WORDS;
        const unsigned long long th = (unsigned long long)tag << 32;
        if (wave == 0) {
          const double x = sh.wrow[bw][lane];
          __hip_atomic_store(slot + 2 * lane, th | (unsigned)__double2loint(x), __ATOMIC_RELAXED, SC);
          __hip_atomic_store(slot + 2 * lane + 1, th | (unsigned)__double2hiint(x), __ATOMIC_RELAXED, SC);
        } else if (wave == 1 && lane < 3) {
          const unsigned w32 = lane == 0 ? (unsigned)__double2loint(bk) : (lane == 1 ? (unsigned)__double2hiint(bk) : bp);
          __hip_atomic_store(slot + 128 + lane, th | w32, __ATOMIC_RELAXED, SC);
        }
      }
      LC_TSTAMP(6 * J + 2);
      /* ---- poll every slot (wave w takes slots w, w + 4, ...; four at a time, all their loads in flight together):
              headers and rows in the same round trip.  (Every wave polling ALL slots itself, which saves the second
              barrier, was slower: 327 vs 257 us per panel -- 4 x the polling traffic per CU.) ---- */
      double gk = -3.0, grow = 0.0;
      unsigned gp = 0xffffffffu;
      for (unsigned gb = (unsigned)wave; gb < G; gb += 16) {
        unsigned long long w0[4], w1[4], wh[4];
        int spins = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const unsigned g = gb + 4u * q;
            if (g < G) {
              const unsigned long long *slot = slots + ((size_t)(J & 1) * LC_GMAX + g) * LC_SLOT_WORDS;
              w0[q] = __hip_atomic_load(slot + 2 * lane, __ATOMIC_RELAXED, SC);
              w1[q] = __hip_atomic_load(slot + 2 * lane + 1, __ATOMIC_RELAXED, SC);
              wh[q] = __hip_atomic_load(slot + 128 + (lane < 3 ? lane : 0), __ATOMIC_RELAXED, SC);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (gb + 4u * q < G) ok = ok && (unsigned)(w0[q] >> 32) == tag && (unsigned)(w1[q] >> 32) == tag && (unsigned)(wh[q] >> 32) == tag;
          if (__all(ok) || dead) break;
          if (++spins > LC_POLL_MAX || ((spins & 255) == 0 && __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            dead = true;
            if (lane == 0) __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (gb + 4u * q < G) {
            const int klo = __builtin_amdgcn_readlane((int)(unsigned)wh[q], 0), khi = __builtin_amdgcn_readlane((int)(unsigned)wh[q], 1);
            const double ok_ = __hiloint2double(khi, klo);
            const unsigned op_ = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wh[q], 2);
            if (ok_ > gk || (ok_ == gk && op_ < gp)) { gk = ok_; gp = op_; grow = __hiloint2double((int)(unsigned)w1[q], (int)(unsigned)w0[q]); }
          }
      }
      LC_TSTAMP(6 * J + 3);
      sh.cand[wave][lane] = grow;
      if (lane == 0) { sh.ckey[wave] = gk; sh.cpos[wave] = gp; }
      __syncthreads();
      double bk; unsigned piv; int bw;
      lc_pick4(sh.ckey, sh.cpos, bk, piv, bw);
      if (piv == 0xffffffffu) piv = c;                      /* cannot happen (the diagonal row is always a candidate) */
      if (bid == 0 && tid == 0) ipiv[c] = (int)piv;
      LC_TSTAMP(6 * J + 4);
      /* ---- the swap, in positions ---- */
      if (valid) { if (pos == piv) pos = c; else if (pos == c) pos = piv; }
      const double *prow = sh.cand[bw];
      const double ajj = prow[J];
      if (valid && pos > c && ajj != 0.0) {                 /* lu.c:105 */
        const double l = a[J] / ajj;
        a[J] = l;
#pragma unroll
        for (int k = J + 1; k < LC_W; k++) a[k] = a[k] - l * prow[k];
      }
    }
    LC_TSTAMP(6 * J + 5);
    lc_step<J + 1>(a, pos, valid, j0, wb, sh, bid, slots, ctl, tagbase, G, ipiv, dead);
  }
}

__global__ void __launch_bounds__(LC_ROWS)
lu_coop_kernel(double *__restrict__ A, size_t lda, size_t n, size_t j0, int wb, int *__restrict__ ipiv,
               unsigned long long *__restrict__ slots, unsigned *__restrict__ ctl)
{
  __shared__ LcShared sh;
  const int tid = threadIdx.x;
  const unsigned G = gridDim.x, bid = blockIdx.x;
  const unsigned gen = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   /* stable until workgroup 0 ends */
  const unsigned tagbase = gen * 64u;
  const size_t row = j0 + (size_t)bid * LC_ROWS + tid;
  const bool valid = row < n;
  unsigned pos = (unsigned)row;
  double a[LC_W];
  if (valid) {
    const double *p = A + row * lda + j0;                   /* 16-byte aligned: lda even, j0 even (host checks) */
#pragma unroll
    for (int k = 0; k < LC_W; k += 2) {
      if (k + 1 < wb) { const double2 t = *reinterpret_cast<const double2 *>(p + k); a[k] = t.x; a[k + 1] = t.y; }
      else { a[k] = k < wb ? p[k] : 0.0; a[k + 1] = 0.0; }
    }
  } else {
#pragma unroll
    for (int k = 0; k < LC_W; k++) a[k] = 0.0;
  }
  bool dead = false;
  LC_TSTAMP(384);
  lc_step<0>(a, pos, valid, (unsigned)j0, wb, sh, bid, slots, ctl, tagbase, G, ipiv, dead);
  /* every workgroup has published the last column, hence loaded its rows: positions can be written over them */
  if (valid) {
    double *p = A + (size_t)pos * lda + j0;
#pragma unroll
    for (int k = 0; k < LC_W; k += 2) {
      if (k + 1 < wb) *reinterpret_cast<double2 *>(p + k) = make_double2(a[k], a[k + 1]);
      else if (k < wb) p[k] = a[k];
    }
  }
  LC_TSTAMP(385);
  if (bid == 0 && tid == 0) __hip_atomic_store(ctl, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

/* 64-row base of the same solve (round 4): X = L^-1 B for a unit-lower 64 x 64 block, one column of B per thread, 16 rows at
   a time in registers: a block first takes the updates of the solved values above it (kept in LDS, one runtime loop with 16
   independent FMAs per solved value), then its own 16 x 16 triangle.  L transposed in LDS: a step reads its column of
   multipliers as uniform 16-byte words.  One launch instead of the seven (four 16-row bases + three products) the 16-row
   recursion spent on a 64-row block: ~1470 -> ~320 launches for the U12 solves of an N = 4096 factorisation.  (All 64 values
   in registers with the solve fully unrolled: the scheduler hoists the 2016 LDS reads, 15 KB of scratch per lane, 259 us.) */
#define TB64 64
__global__ void __launch_bounds__(64)
trsm_unit_lower64_kernel(double *__restrict__ A, size_t lda, size_t r0, int nb, size_t c0, size_t nc)
{
  __shared__ __attribute__((aligned(16))) double sLt[TB64][TB64 + 2];   /* sLt[k][r] = L[r][k], r > k; pitch 66: a column store spreads over the banks */
  __shared__ double xs[TB64][64];                        /* solved values of this workgroup's 64 columns */
  const int tid = threadIdx.x;
  const size_t c = (size_t)blockIdx.x * 64 + tid;
  const bool live = c < nc;
  double *col = A + r0 * lda + c0 + (live ? c : 0);
  /* this thread's whole column, requested before anything else (four dependent load latencies otherwise) */
  double xr[TB64 / 16][16];
#pragma unroll
  for (int rb = 0; rb < TB64 / 16; rb++) {
#pragma unroll
    for (int r = 0; r < 16; r++) xr[rb][r] = (live && rb * 16 + r < nb) ? col[(size_t)(rb * 16 + r) * lda] : 0.0;
  }
  /* L transposed into LDS: 64 coalesced row loads (lane = column), all in flight together, then written down the columns of
     sLt (padded pitch: the 64 lanes of a store spread over 16 banks instead of one).  Lane = row with a loop over its 64 multipliers kept
     eight uncoalesced loads in flight at a time: ~8 of the kernel's 20 us. */
  {
    double lv[TB64];
#pragma unroll
    for (int r = 0; r < TB64; r++) lv[r] = (r < nb && tid < r) ? A[(r0 + r) * lda + r0 + tid] : 0.0;
#pragma unroll
    for (int r = 0; r < TB64; r++) sLt[tid][r] = lv[r];
  }
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int rb = 0; rb < TB64 / 16; rb++) {
    if (rb * 16 < nb) {
      double (&x)[16] = xr[rb];
#pragma unroll 4
      for (int k = 0; k < rb * 16; k++) {                /* the blocks above: 16 updates per solved value (4 in flight) */
        const double xk = xs[k][tid];
        const double *l = &sLt[k][rb * 16];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = fma(-l[r], xk, x[r]);
      }
#pragma unroll
      for (int k = 0; k < 15; k++) {                     /* this block's 16 x 16 triangle */
        const double *l = &sLt[rb * 16 + k][rb * 16];
#pragma unroll
        for (int r = k + 1; r < 16; r++) x[r] = fma(-l[r], x[k], x[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        xs[rb * 16 + r][tid] = x[r];
        if (rb * 16 + r < nb && rb * 16 + r > 0) col[(size_t)(rb * 16 + r) * lda] = x[r];     /* row 0 is unchanged */
      }
    }
  }
}

/* the interchanges of ONE 64-wide panel (k0 <= k < k1) applied to every column outside it, [0, k0) and [k1', n): run right
   after the panel kernel ("eager"), so a thread's dependent chain is 64 swaps; the recursion's own calls walked up to 2048
   swaps per thread (408 us for the top level: 160 GB/s).  Equivalent: the columns to the right are not touched by anything
   else before the swaps the recursion would have applied to them. */
__global__ void __launch_bounds__(64)
laswp_outside_kernel(double *__restrict__ A, size_t lda, size_t n, size_t pc0, size_t pw, const int *__restrict__ ipiv, size_t k0, size_t k1)
{
  size_t c = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= n - pw) return;
  if (c >= pc0) c += pw;                                  /* skip the panel's own columns */
  double *col = A + c;
  for (size_t k = k0; k < k1; k++) {
    const size_t p = (size_t)ipiv[k];
    if (p != k) {
      const double a = col[k * lda], b = col[p * lda];
      col[k * lda] = b; col[p * lda] = a;
    }
  }
}

static int trsm_unit_lower(gsl_sinterp_hip_ctx *ctx, double *A, size_t lda, size_t r0, size_t nb, size_t c0, size_t nc)
{
  if (nb == 0 || nc == 0) return ST_SUCCESS;
  static const bool no64 = getenv("GSL_SINTERP_NO_TRSM64") && getenv("GSL_SINTERP_NO_TRSM64")[0] == '1';
  if (!no64 && nb <= TB64 && nb > TB) {
    hipLaunchKernelGGL(trsm_unit_lower64_kernel, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, ctx->stream, A, lda,
                       r0, (int)nb, c0, nc);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  if (nb <= TB) {
    hipLaunchKernelGGL(trsm_unit_lower_base_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, A, lda,
                       r0, (int)nb, c0, nc);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  const size_t tb = (!no64 && nb > TB64) ? TB64 : TB;
  size_t n1 = ((nb / 2 + tb - 1) / tb) * tb;
  if (n1 >= nb) n1 = nb - tb;
  int st = trsm_unit_lower(ctx, A, lda, r0, n1, c0, nc);
  if (st) return st;
  /* B2 -= L21 * X1 :  L21 = A[r0+n1 : r0+nb, r0 : r0+n1],  X1 = A[r0 : r0+n1, c0 : c0+nc] */
  st = sinterp_gemm_minus(ctx, nb - n1, nc, n1, A + (r0 + n1) * lda + r0, lda, A + r0 * lda + c0, lda, 1,
                          A + (r0 + n1) * lda + c0, lda, 0);
  if (st) return st;
  return trsm_unit_lower(ctx, A, lda, r0 + n1, nb - n1, c0, nc);
}

/* the cooperative panel kernel applies: buffers in place, aligned rows, at most LC_GMAX workgroups */
static bool lu_coop_ok(const gsl_sinterp_hip_ctx *ctx, size_t lda, size_t n, size_t j0)
{
  static const bool off = getenv("GSL_SINTERP_NO_LU_COOP") && getenv("GSL_SINTERP_NO_LU_COOP")[0] == '1';
  return !off && ctx->d_lu_coop != NULL && (lda & 1) == 0 && (j0 & 1) == 0 && n - j0 <= (size_t)LC_GMAX * LC_ROWS;
}

static int lu_panel(gsl_sinterp_hip_ctx *ctx, double *A, size_t lda, size_t n, size_t j0, size_t w, int *d_ipiv, bool eager)
{
  const size_t kend = j0 + w < n ? j0 + w : n;
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
    if (eager && n > w)
      hipLaunchKernelGGL(laswp_outside_kernel, dim3((unsigned)((n - w + 63) / 64)), dim3(64), 0, ctx->stream, A, lda, n, j0, w, d_ipiv, j0, kend);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  /* tall 64-wide panels: the cooperative kernel (one 64-column step of the recursion = one launch) */
  if (w <= LC_W && lu_coop_ok(ctx, lda, n, j0)) {          /* (the shorter ones went to the block kernel above) */
    const unsigned G = (unsigned)((prow + LC_ROWS - 1) / LC_ROWS);
    hipLaunchKernelGGL(lu_coop_kernel, dim3(G), dim3(LC_ROWS), 0, ctx->stream, A, lda, n, j0, (int)w, d_ipiv,
                       (unsigned long long *)((char *)ctx->d_lu_coop + 64), (unsigned *)ctx->d_lu_coop);
    if (eager && n > w)
      hipLaunchKernelGGL(laswp_outside_kernel, dim3((unsigned)((n - w + 63) / 64)), dim3(64), 0, ctx->stream, A, lda, n, j0, w, d_ipiv, j0, kend);
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
  const size_t unit = (w > BW && ((prow <= block_rows && prow <= (size_t)BT * 8) || lu_coop_ok(ctx, lda, n, j0))) ? BW : LB;   /* split on 64-wide panel boundaries */
  size_t w1 = ((w / 2 + unit - 1) / unit) * unit;
  if (w1 >= w) w1 = w - unit;
  const size_t w2 = w - w1, c1 = j0 + w1;
  int st = lu_panel(ctx, A, lda, n, j0, w1, d_ipiv, eager);
  if (st) return st;
  if (!eager) {
    hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((w2 + 255) / 256)), dim3(256), 0, ctx->stream, A, lda, c1, w2, d_ipiv, j0, c1);
    LAUNCH_CHECK(ctx);
  }
  st = trsm_unit_lower(ctx, A, lda, j0, w1, c1, w2);             /* U12 = L11^-1 A12 */
  if (st) return st;
  st = sinterp_gemm_minus(ctx, n - c1, w2, w1, A + c1 * lda + j0, lda, A + j0 * lda + c1, lda, 1,
                          A + c1 * lda + c1, lda, 0);           /* A22 -= A21 U12 */
  if (st) return st;
  st = lu_panel(ctx, A, lda, n, c1, w2, d_ipiv, eager);
  if (st) return st;
  if (!eager) {
    hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((w1 + 255) / 256)), dim3(256), 0, ctx->stream, A, lda, j0, w1, d_ipiv, c1,
                       c1 + w2 < n ? c1 + w2 : n);
    LAUNCH_CHECK(ctx);
  }
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
  if (!ctx->d_lu_coop) {                               /* [generation, abort | slots], zeroed once: tags start at 1 */
    const size_t bytes = 64 + (size_t)2 * LC_GMAX * LC_SLOT_WORDS * sizeof(unsigned long long);
    HIP_OK(ctx, hipMalloc(&ctx->d_lu_coop, bytes));
    HIP_OK(ctx, hipMemset(ctx->d_lu_coop, 0, bytes));
    const unsigned gen0 = 1;                             /* tag 0 = never written */
    HIP_OK(ctx, hipMemcpy(ctx->d_lu_coop, &gen0, sizeof gen0, hipMemcpyHostToDevice));
  }
  st = sinterp_graph_try_launch(ctx, 1, n, lda, d_a, d_perm, &replayed);
  if (st) return st;
  if (!replayed) {
    hipStream_t saved;
    st = sinterp_capture_begin(ctx, &saved);
    if (st) return st;
    /* every leaf a 64-wide panel (block or cooperative kernel): its interchanges go to the other columns right away */
    const bool eager = lu_coop_ok(ctx, lda, n, 0);
    st = lu_panel(ctx, d_a, lda, n, 0, n, d_perm, eager);         /* d_perm holds LAPACK-style ipiv for now */
    int st2 = sinterp_capture_end(ctx, saved, 1, n, lda, d_a, d_perm);
    if (st) return st;
    if (st2) return st2;
  }
  /* ipiv -> gsl_permutation content + signum (lu.c:95-101) */
  int *h_ipiv = (int *)malloc(n * sizeof(int));
  int *h_perm = (int *)malloc(n * sizeof(int));
  if (!h_ipiv || !h_perm) { free(h_ipiv); free(h_perm); return sinterp_fail(ctx, ST_ENOMEM, "lu_decomp: host buffers", hipSuccess, __FILE__, __LINE__); }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  unsigned h_ctl[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(h_ctl, ctx->d_lu_coop, sizeof h_ctl, hipMemcpyDeviceToHost);
  if (e == hipSuccess && h_ctl[1] != 0) {             /* a grid-wide exchange timed out: the result is not a factorisation */
    (void)hipMemset((char *)ctx->d_lu_coop + 4, 0, 4);
    free(h_ipiv); free(h_perm);
    return sinterp_fail(ctx, ST_EFAILED, "lu_decomp: the cooperative panel kernel timed out waiting for a workgroup", hipSuccess, __FILE__, __LINE__);
  }
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
