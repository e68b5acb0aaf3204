/*
 * chol_potrf.h -- the in-LDS factorisation of a 128 x 128 diagonal block (potrf128), shared by the launch-per-panel
 * driver (chol.hip: chol_diag128_kernel) and the persistent task-DAG driver (chol_dag.hip: the chain workgroup).
 * Reference semantics: gsl_linalg_cholesky_decomp1, linalg/cholesky.c:88-131.
 */
#ifndef SINTERP_CHOL_POTRF_H
#define SINTERP_CHOL_POTRF_H

#define CB 32 /* base panel width */
#ifndef TSTAMP
#define TSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ double lane_bcast(double v, int src)
{
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

typedef double double4_t __attribute__((ext_vector_type(4)));
#define PB 128
#define PQ 34
#define PBLK (32 * PQ)
#define TR_LD 130     /* pitch of the 64 x 128 tile of the trsm kernel (= 2 mod 4) */

__device__ __forceinline__ int pblk(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * PBLK; }

/* 16x16 fragment (fi, fj) of  acc += sgn * A * B^T,  A and B row-major [32][PQ] blocks, K = 32 */
__device__ __forceinline__ double4_t frag_nt(const double *Ab, const double *Bb, int fi, int fj, int lane, double4_t acc, double sgn)
{
  const double *ap = Ab + (fi * 16 + (lane & 15)) * PQ + (lane >> 4);
  const double *bp = Bb + (fj * 16 + (lane & 15)) * PQ + (lane >> 4);
#pragma unroll
  for (int kk = 0; kk < 8; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * ap[kk * 4], bp[kk * 4], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ double4_t frag_load(const double *Cb, int fi, int fj, int lane)
{
  double4_t c;
#pragma unroll
  for (int rg = 0; rg < 4; rg++) c[rg] = Cb[(fi * 16 + (lane >> 4) + 4 * rg) * PQ + fj * 16 + (lane & 15)];
  return c;
}
__device__ __forceinline__ void frag_store(double *Cb, int fi, int fj, int lane, double4_t c)
{
#pragma unroll
  for (int rg = 0; rg < 4; rg++) Cb[(fi * 16 + (lane >> 4) + 4 * rg) * PQ + fj * 16 + (lane & 15)] = c[rg];
}


/* potrf32 on one wave: lane = row, the row's 32 entries in registers, left-looking; compile-time
   recursion over the columns (straight-line code, no branch per column: a failing pivot is recorded
   and reported once at the end).
   Column J+1 of row i is  a_i[J+1] - sum_{k<=J} L[i][k] L[J+1][k].  Row J+1 of L has to reach every
   lane: a readlane pair per entry (the obvious way) makes the kernel issue bound.  Instead every
   finished column is stored to the LDS image of the block (one ds_write_b64 per column), and row
   J+1's entries k < J -- final before column J starts -- come back as uniform-address ds_read_b128
   (two entries per instruction), issued before column J's rsq chain and consumed in its latency
   shadow (a wave issues in order: the FMAs are interleaved by hand between the ~8 dependent chain
   ops, sched_barrier pins the order).  Only the k = J term needs a readlane. */
template <int J, int SLOT>
__device__ __forceinline__ void potrf32_fill(const double (&a)[CB], const double (&rp)[CB], double rlast, double (&p)[2])
{
  /* partial dot product of column J+1: terms k < J-1 use row entries fetched one column ago (rp), the
     term k = J-1 the entry fetched at the start of this column (rlast), consumed last; two accumulators
     (the wave issues in order and an fp64 FMA occupies the VALU for 4 cycles: two chains already hide its
     latency, and every extra accumulator is one more add on the dependent path into the next column) */
  if constexpr (J + 1 < CB && SLOT >= 2 && SLOT <= 6 && J >= 2) {
#pragma unroll
    for (int k = ((SLOT - 2) * (J - 1)) / 5; k < ((SLOT - 1) * (J - 1)) / 5; k++) p[k & 1] = fma(-a[k], rp[k], p[k & 1]);
  }
  if constexpr (J + 1 < CB && SLOT == 7 && J >= 1) p[(J - 1) & 1] = fma(-a[J - 1], rlast, p[(J - 1) & 1]);
  __builtin_amdgcn_sched_barrier(0);
}

/* rp: entries k < J-1 of row J+1 of L, fetched from the LDS image during column J-1 */
template <int J>
__device__ __forceinline__ void potrf32_cols(double (&a)[CB], int lane, double cur, int &badcol, const double *D, double *colp,
                                             int cstride, const double (&rp)[CB])
{
  if constexpr (J < CB) {
    TSTAMP(32 + J);
    double v = cur;
    if constexpr (J > 0) v = fma(-a[J - 1], lane_bcast(a[J - 1], J), v);
    double d = lane_bcast(v, J);
    /* LDS reads issued now: the one entry of row J+1 that column J-1 just produced (used in the last
       fill slot of this column), and row J+2's entries k < J for the NEXT column -- a full column
       (~200 cycles) ahead of their use, so the ~70-cycle LDS latency never stalls the in-order wave */
    double rlast = 0.0;
    if constexpr (J >= 1 && J + 1 < CB) rlast = D[(J + 1) * PQ + (J - 1)];
    double rn[CB];
    if constexpr (J + 2 < CB) {
#pragma unroll
      for (int k = 0; k < J; k += 2) {
        const double2 t = *reinterpret_cast<const double2 *>(D + (J + 2) * PQ + k);
        rn[k] = t.x;
        if (k + 1 < CB) rn[k + 1] = t.y;
      }
    }
    /* cholesky.c:120-123: a pivot <= 0 is RECORDED (first failing column wins), off the dependent chain: the
       rsq below then yields inf / NaN, which flows through the rest of the block; the caller reports GSL_EDOM
       and the content of a failed factorisation is unspecified (as in the reference, which stops mid-way).
       (the test itself is issued after the rsq, below) */
    /* 1/sqrt(d) = y0 (1 - r)^(-1/2), r = 1 - d y0^2 with the v_rsq_f64 seed y0 (|r| <~ 2^-21):
       y0 (1 + r/2 + 3 r^2/8) is exact to r^3 ~ 1e-19 -- four dependent ops after the seed instead
       of the seven of a Newton step plus correction.  The diagonal entry sqrt(d) = d / sqrt(d) is the
       SAME product v * inv every row forms (lane J holds v = d), so no lane needs a special case; it
       ends within ~2 ulp of cholesky.c:125-126's sqrt (a separate residual correction for that one
       entry cost 9 instructions per column on the wave's in-order issue path). */
    double p[2] = {0.0, 0.0};
    if constexpr (J + 1 < CB) p[0] = a[J + 1];
    const double y0 = __builtin_amdgcn_rsq(d);
    __builtin_amdgcn_sched_barrier(0);
    const bool ok = d > 0.0;                            /* in the latency shadow of the rsq */
    badcol = (!ok && badcol == 0) ? J + 1 : badcol;
    potrf32_fill<J, 0>(a, rp, rlast, p);
    const double t = d * y0;
    potrf32_fill<J, 1>(a, rp, rlast, p);
    const double rr = fma(-t, y0, 1.0);
    potrf32_fill<J, 2>(a, rp, rlast, p);
    const double s1 = fma(0.375, rr, 0.5), u = y0 * rr;
    potrf32_fill<J, 3>(a, rp, rlast, p);
    const double inv = fma(u, s1, y0);
    potrf32_fill<J, 4>(a, rp, rlast, p);
    a[J] = v * inv;
    potrf32_fill<J, 5>(a, rp, rlast, p);
    colp[J * cstride] = a[J];                           /* column J of L (row reads of later columns) / of L^-1 */
    potrf32_fill<J, 6>(a, rp, rlast, p);
    potrf32_fill<J, 7>(a, rp, rlast, p);
    potrf32_cols<J + 1>(a, lane, p[0] + p[1], badcol, D, colp, cstride, rn);
  }
}


/* potrf128 of the lower triangle held in S (10 packed 32 x 32 blocks, pitch PQ), in place; Dv receives the inverses of
   the four diagonal blocks.  Called by ALL NW * 64 threads of the workgroup (it synchronises them); info: first
   failing column (1-based, j0 + column), set with atomicCAS when a pivot is <= 0. */
template <int NW>
__device__ __forceinline__ void potrf128_lds(double *S, double *Dv, int tid, int *__restrict__ info, size_t j0)
{
  const int lane = tid & 63, wave = tid >> 6;
  /* Schedule.  The dependent chain is potrf32(0) -> TRSM of column 0 -> update of block (1,1) -> potrf32(1) -> ...;
     a potrf32 occupies ONE wave for ~6 us.  Everything that is not on that chain -- the updates of the blocks
     below the next diagonal block -- is done by waves 1..3 WHILE wave 0 factors the next diagonal block:
         P(0) | T(0) | U1(0) | P(1) || U2(0) | T(1) | U1(1) | P(2) || U2(1) | T(2) | U1(2) | P(3)
     T(jb): X = B Dinv^T for the blocks below D_jb; U1(jb): block (jb+1,jb+1) -= blk blk^T (its three lower
     fragments, one per wave); U2(jb): the remaining blocks (bi >= jb+2) -= blk(bi,jb) blk(bj,jb)^T. */
  auto potrf_block = [&](int jb) {
    double *D = S + pblk(jb, jb);
    /* potrf32 (see potrf32_cols).  Lanes 0..31 hold the rows of the diagonal block.  Lanes 32..63
       produce its inverse with the SAME instruction stream: column c of L^-1 obeys
          x_c[J] = (I[J][c] - sum_{k<J} L[J][k] x_c[k]) / L[J][J],
       which is the left-looking update of a "row" whose data is row c of the identity. */
    const bool is_row = lane < CB;
    const int c = lane - CB;
    double a[CB];
    {
      /* branch-free: every lane reads a whole row of the block (lanes 32..63 the row of lane - 32, discarded) as 16
         unconditional ds_read_b128, then selects; a per-entry `k <= lane ? D[..] : 0` compiles to 32 exec-masked
         branches with a full LDS round trip each (~3k cycles per block, measured as the gap between the column
         stamps and the block total) */
      const double *rowp = D + (lane & (CB - 1)) * PQ;
      double v[CB];
#pragma unroll
      for (int k = 0; k < CB; k += 2) { const double2 t = *reinterpret_cast<const double2 *>(rowp + k); v[k] = t.x; v[k + 1] = t.y; }
#pragma unroll
      for (int k = 0; k < CB; k++) a[k] = is_row ? ((k <= lane) ? v[k] : 0.0) : ((k == c) ? 1.0 : 0.0);
    }
    double *colp = is_row ? D + lane * PQ : Dv + jb * PBLK + c;   /* entry J of this lane's vector: colp[J * cstride] */
    const int cstride = is_row ? 1 : PQ;
    int badcol = 0;
    double r0[CB];                                      /* nothing prefetched before column 0 */
    potrf32_cols<0>(a, lane, a[0], badcol, D, colp, cstride, r0);
    /* every column was written to D as it was produced (colp); the strict upper triangle of a diagonal
       block is never read afterwards (TRSM uses Dv, the write-back masks k <= r) */
    if (badcol && lane == 0) atomicCAS(info, 0, (int)(j0 + jb * 32 + badcol));
  };
  auto update_frag = [&](int jb, int bi, int bj, int f) {
    double *Cb = S + pblk(bi, bj);
    double4_t c = frag_load(Cb, f >> 1, f & 1, lane);
    c = frag_nt(S + pblk(bi, jb), S + pblk(bj, jb), f >> 1, f & 1, lane, c, -1.0);
    frag_store(Cb, f >> 1, f & 1, lane, c);
  };

  /* jb = -1 is the prologue P(0); one call site of the (large, fully unrolled) potrf32 body */
  for (int jb = -1; jb < 3; jb++) {
    if (jb >= 0) {
    /* T(jb): rows below, X = B Dinv^T on MFMA, one 16-row strip (both column fragments) per unit; Dinv is
       lower triangular, so the first 16 columns need K = 16 only.  Strips 0, 1 (block (jb+1, jb)) are on the chain. */
    for (int u = wave; u < (3 - jb) * 2; u += NW) {
      double *Bb = S + pblk(jb + 1 + (u >> 1), jb);
      const int fi = u & 1;
      const double *ap = Bb + (fi * 16 + (lane & 15)) * PQ + (lane >> 4);
      const double *bp = Dv + jb * PBLK + (lane & 15) * PQ + (lane >> 4);
      double4_t x0 = (double4_t){0.0, 0.0, 0.0, 0.0}, x1 = x0;
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        const double av = ap[kk * 4];
        if (kk < 4) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[kk * 4], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[16 * PQ + kk * 4], x1, 0, 0, 0);
      }
      frag_store(Bb, fi, 0, lane, x0);
      frag_store(Bb, fi, 1, lane, x1);
    }
    __syncthreads();
    TSTAMP(4 + jb * 4);
    /* U1(jb): the next diagonal block, lower fragments (0,0), (1,0), (1,1) on waves 0..2 */
    if (wave < 3) update_frag(jb, jb + 1, jb + 1, wave == 0 ? 0 : wave + 1);
    __syncthreads();
    TSTAMP(5 + jb * 4);
    }
    /* P(jb+1) on wave 0  ||  U2(jb) on waves 1..3 */
    if (wave == 0) {
      potrf_block(jb + 1);
    } else if (jb >= 0) {
      int unit = 0;
      for (int bi = jb + 2; bi < 4; bi++)
        for (int bj = jb + 1; bj <= bi; bj++)
          for (int f = 0; f < 4; f++) {
            if (bi == bj && f == 1) continue;           /* strictly upper fragment of a diagonal block: never read */
            if (unit++ % (NW - 1) == wave - 1) update_frag(jb, bi, bj, f);
          }
    }
    __syncthreads();
    TSTAMP(6 + jb * 4);
  }

}

#endif
