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
/* per-column stamps inside potrf32 serialise its LDS prefetch (s_memtime waits on lgkmcnt): only with -DSINTERP_DIAG_PROF_COLS */
#if defined(SINTERP_DIAG_PROF_COLS)
#define TSTAMP_COL(i) TSTAMP(i)
#else
#define TSTAMP_COL(i) do { } while (0)
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
   recursion over the columns (straight-line code, no branch per column: a failing pivot shows up as a
   NaN diagonal and is reported once at the end).

   Round 4: the pivot chain runs on UNIFORM values.  Write V_J[i] = A[i][J] - sum_{k<J} L[i][k] L[J][k]
   (lane i), d_J = V_J[J], inv_J = 1/sqrt(d_J), L[i][J] = V_J[i] inv_J.  The round-2 kernel went
   lane values -> readlane -> d_J -> rsq chain -> lane values -> readlane -> ... : two cross-lane hops
   (~26 cycles each) sat on the dependent path of every column.  Here every lane carries the chain
   redundantly:
       uL      = e inv_{J-1}                e = V_{J-1}[J]        (= L[J][J-1])
       d_J     = q - uL^2                   q = (A[J][J] - sum_{k<=J-2} L[J][k]^2), lane J's partial dot product
       inv_J   = rsq chain of d_J
   e and q are readlanes of values that are final one column EARLIER (V_{J-1} needs inv_{J-2} only, the partial dot
   product needs the columns <= J-2), so their latency is off the chain; what remains on it is 2 + 1 + 4 dependent
   fp64 ops (~50 cycles).  The column time is then the wave's in-order issue: ~20 instructions + the J-term dot
   product of the next column (terms k <= J-2 against row J+1 of L from the LDS image -- uniform ds_read_b128,
   fetched a column ahead --, the term k = J-1 against e2 inv_{J-1}, e2 = V_{J-1}[J+1], a third readlane).
   d_J formed this way is bit-identical to V_J[J] (same operands, same fma).  */
/* Order pins.  The column body below is written in the order the wave should issue it (it issues in order); the
   compiler's schedulers otherwise sink the LDS prefetch down to its use a column later (an LDS round trip on the
   dependent path of every column) and the pivot readlane behind the chain tail.  An empty volatile asm that takes values
   as read-write operands is a join point: every producer of those values comes before it, every consumer after, and
   the "memory" clobber keeps the LDS accesses on their side.  No instruction is emitted. */
#define PIN_V2(x, y)       asm volatile("" : "+v"(x), "+v"(y) : : "memory")
#define PIN_V4(x, y, z, w) asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : : "memory")
#define PIN_SSV(x, y, z)   asm volatile("" : "+s"(x), "+s"(y), "+v"(z) : : "memory")
#define PIN_SV(x, y)       asm volatile("" : "+s"(x), "+v"(y) : : "memory")

/* Columns [K0, K0 + 16) of the 32-wide block: the dot products run over k in [K0, J) only -- the terms k < K0 of the second
   half were applied to the LDS image by potrf32_split (MFMA), which halves the average dot-product length of the block. */
#define HB 16
template <int K0, int J>
__device__ __forceinline__ void potrf32_step(double (&a)[CB], double vprev, double cur, double inv_prev, double e, double e2, double q,
                                             const double *D, double *colp, const double (&rp)[CB])
{
  constexpr int END = K0 + HB;
  if constexpr (J < END) {
    TSTAMP_COL(32 + J);
    /* ---- head of the chain: d_J and the rsq seed */
    double v, d, uL2 = 0.0, acol = 0.0;
    if constexpr (J == K0) {
      v = cur;
      d = lane_bcast(v, J);
    } else {
      const double uL = e * inv_prev;                   /* L[J][J-1], uniform */
      d = fma(-uL, uL, q);                              /* uniform; = V_J[J] bit for bit */
      acol = vprev * inv_prev;                          /* column J-1 of L (lanes >= 32: of L^-1) */
      v = fma(-acol, uL, cur);                          /* V_J */
      uL2 = e2 * inv_prev;                              /* L[J+1][J-1], uniform */
    }
    double y0 = __builtin_amdgcn_rsq(d);
    PIN_V4(y0, v, uL2, acol);
    /* ---- in the shadow of the rsq: column J-1 to the LDS image, the prefetch for the NEXT column's dot product (row
       J+2's entries K0 <= k < J: columns <= J-1 are in the image, the store precedes these reads in the wave's in-order
       LDS queue), and the two broadcasts of V_J the next column starts from */
    if constexpr (J > K0) { a[J - 1] = acol; colp[J - 1] = acol; }
    double rn[CB];
    if constexpr (J + 2 < END) {
#pragma unroll
      for (int k = K0; k < J; k += 2) {
        const double2 t2 = *reinterpret_cast<const double2 *>(D + (J + 2) * PQ + k);
        rn[k] = t2.x;
        if (k + 1 < CB) rn[k + 1] = t2.y;
      }
    }
    double en = 0.0, e2n = 0.0, qn = 0.0, curn = 0.0;
    if constexpr (J + 1 < END) en = lane_bcast(v, J + 1);
    if constexpr (J + 2 < END) e2n = lane_bcast(v, J + 2);
    /* ---- partial dot product of column J+1: terms K0 <= k <= J-1 (two accumulators), then lane J+1's value of it */
    if constexpr (J + 1 < END) {
      double p0 = a[J + 1], p1 = 0.0;
      if constexpr (J >= K0 + 1) p1 = -acol * uL2;      /* the term k = J-1 opens the second accumulator */
      PIN_SSV(en, e2n, p0);
#pragma unroll
      for (int k = K0; k + 1 < J; k++) {
        if (k & 1) p1 = fma(-a[k], rp[k], p1);
        else p0 = fma(-a[k], rp[k], p0);
      }
      curn = (J >= K0 + 1) ? p0 + p1 : p0;
      qn = lane_bcast(curn, J + 1);
      PIN_SV(qn, y0);
    }
    /* ---- tail of the chain (q's readlane latency hides behind it).
       1/sqrt(d) = y0 (1 - r)^(-1/2), r = 1 - d y0^2 with the v_rsq_f64 seed y0 (|r| <~ 2^-21):
       y0 (1 + r/2 + 3 r^2/8) is exact to r^3 ~ 1e-19 -- four dependent ops after the seed.  The diagonal entry
       sqrt(d) = d / sqrt(d) is the SAME product v * inv every row forms (lane J holds v = d); it ends within
       ~2 ulp of cholesky.c:125-126's sqrt.  (Moving the store / prefetch / broadcasts into the stalls between these
       four dependent ops was measured: 22.1 vs 21.2 us per block -- the reads then return too late for the next fill.) */
    const double t = d * y0;
    const double rr = fma(-t, y0, 1.0);
    const double s1 = fma(0.375, rr, 0.5), u = y0 * rr;
    const double inv = fma(u, s1, y0);
    potrf32_step<K0, J + 1>(a, v, curn, inv, en, e2n, qn, D, colp, rn);
  } else {
    a[END - 1] = vprev * inv_prev;
    colp[END - 1] = a[END - 1];
  }
}

/* Between the two halves of a potrf32 (same wave, no barrier: the wave's LDS queue is in order): with X = rows 16..31,
   columns 0..15 of the block (L_10, final) and W = the inverse of the first 16 x 16 block (lanes 32..47 produced it),
       D[16:32, 16:32]   -=  X X^T          the k < 16 terms of every later column's dot product,
       Dt[c][16:32]       =  -(X W)[:, c]    the same terms for the inverse columns c < 16
   -- 8 MFMAs instead of 16 x 16 FMA terms on the wave's in-order issue path.  D: the block (row-major, pitch PQ);
   Dt: its inverse stored TRANSPOSED (Dt[c * PQ + k] = (L^-1)[k][c]). */
__device__ __forceinline__ void potrf32_split(double *D, double *Dt, int lane)
{
  const int fr = lane & 15, fq = lane >> 4;
  const double *ap = D + (HB + fr) * PQ + fq;          /* A operand: X[row = fr][k = fq + 4 kk]; also the B operand of X X^T */
  const double *bp = Dt + fr * PQ + fq;                /* B operand of X W: W[k][c = fr] = Dt[c * PQ + k] */
  double4_t cu, cw = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int rg = 0; rg < 4; rg++) cu[rg] = D[(HB + fq + 4 * rg) * PQ + HB + fr];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    const double av = ap[kk * 4], na = -av;
    cu = __builtin_amdgcn_mfma_f64_16x16x4f64(na, av, cu, 0, 0, 0);
    cw = __builtin_amdgcn_mfma_f64_16x16x4f64(na, bp[kk * 4], cw, 0, 0, 0);
  }
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    D[(HB + fq + 4 * rg) * PQ + HB + fr] = cu[rg];
    Dt[fr * PQ + HB + fq + 4 * rg] = cw[rg];
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
    double *Dt = Dv + jb * PBLK;                        /* the inverse, TRANSPOSED: Dt[c * PQ + k] = (L^-1)[k][c]; identity on entry */
    /* this lane's vector: a row of the block, or (lanes >= 32) row c of the identity -- both plain 16-byte LDS reads of
       colp[0 .. 31], no selects.  The entries right of the diagonal (the symmetric copy of the input) are NOT masked:
       they only ever flow into entries right of the diagonal of the same row, which nothing reads. */
    double *colp = is_row ? D + lane * PQ : Dt + (lane - CB) * PQ;
    double a[CB];
#pragma unroll
    for (int k = 0; k < CB; k += 2) { const double2 t = *reinterpret_cast<const double2 *>(colp + k); a[k] = t.x; a[k + 1] = t.y; }
    double r0[CB];                                      /* nothing prefetched before the first column of a half */
    potrf32_step<0, 0>(a, 0.0, a[0], 0.0, 0.0, 0.0, 0.0, D, colp, r0);
    potrf32_split(D, Dt, lane);
#pragma unroll
    for (int k = HB; k < CB; k += 2) { const double2 t = *reinterpret_cast<const double2 *>(colp + k); a[k] = t.x; a[k + 1] = t.y; }
    potrf32_step<HB, HB>(a, 0.0, a[HB], 0.0, 0.0, 0.0, 0.0, D, colp, r0);
    /* every column was written to the LDS images as it was produced (colp).
       cholesky.c:120-123: a pivot d <= 0 makes the rsq chain produce NaN (d < 0: rsq = NaN; d = 0: 0 * inf), which
       flows into every later column, so the FIRST row whose diagonal entry is not a positive number is the failing
       column; the caller reports GSL_EDOM and the content of a failed factorisation is unspecified (as in the
       reference, which stops mid-way).  One LDS read + one ballot per block instead of a test per column. */
    const double dg = D[(lane & (CB - 1)) * PQ + (lane & (CB - 1))];
    const unsigned long long badm = __ballot(is_row && !(dg > 0.0));
    if (badm && lane == 0) atomicCAS(info, 0, (int)(j0 + jb * 32 + __ffsll((long long)badm)));
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
      const double *bp = Dv + jb * PBLK + (lane >> 4) * PQ + (lane & 15);     /* Dinv[n][k] = Dt[k * PQ + n] */
      double4_t x0 = (double4_t){0.0, 0.0, 0.0, 0.0}, x1 = x0;
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        const double av = ap[kk * 4];
        if (kk < 4) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[kk * 4 * PQ], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[kk * 4 * PQ + 16], x1, 0, 0, 0);
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
