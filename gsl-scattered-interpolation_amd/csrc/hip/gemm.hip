/*
 * gemm.hip -- the Level-3 building block of the dense solves: C -= A * op(B) on
 * v_mfma_f64_16x16x4_f64 (gfx950).  The reference factorizations are Level-2
 * (linalg/cholesky.c:105-116 gaxpy via cblas/source_gemv_r.h:60-75;
 * linalg/lu.c:105-119 rank-1 updates); the blocked drivers in chol.hip / lu.hip
 * move all O(N^3) work here (the role gsl_blas_dsyrk / dgemm / dtrsm,
 * blas/blas.c:1334,1649,2105, would play in a blocked GSL routine).
 *
 * Tiling (one workgroup = 256 threads = 4 waves, 2x2):
 *   block tile 128 x 128, K step 16, double-buffered LDS, register-staged
 *   prefetch of the next K step issued before the MFMAs of the current one;
 *   each wave owns a 64 x 64 sub-tile = 4 x 4 MFMA fragments (16 accumulators
 *   of 4 f64 -> 128 VGPRs), 64 MFMAs per K step.
 * LDS images:
 *   A (and B when stored [n][k]) as [row][16 + 2]: the fragment read
 *   (row = lane&15, k = lane>>4) is a conflict-free ds_read_b64 because the
 *   row pitch (18 doubles) is = 2 mod 4;
 *   B stored [k][n] as [k][128 + 16]: consecutive lanes read consecutive n.
 * MFMA operand maps (cdna_hip_programming.md section 3): A[i=lane&15][k=lane>>4],
 *   B[k=lane>>4][j=lane&15], D[row=(lane>>4)+4*reg][col=lane&15].
 */
#include "common.h"
#include <stdlib.h>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define GT_BM 128
#define GT_BN 128
#define GT_BK 16
#define GT_LDA (GT_BK + 2)      /* 18 */
#define GT_LDBN (GT_BN + 16)    /* 144 */
#define GT_ASZ (GT_BM * GT_LDA) /* 2304 doubles */
#define GT_BSZ 2304             /* max(128*18, 16*144) */

struct GemmArgs {
  size_t m, n, k;
  const double *A; size_t lda;
  const double *B; size_t ldb;
  double *C; size_t ldc;
  int lower_only;
  int tiles_m, tiles_n;
  unsigned n_active;   /* tiles that are launched: all, or the lower trapezoid when lower_only */
};

__device__ __forceinline__ double2 ld2(const double *p, bool ok0, bool ok1, bool vec)
{
  if (vec && ok1) return *reinterpret_cast<const double2 *>(p);
  double2 v;
  v.x = ok0 ? p[0] : 0.0;
  v.y = ok1 ? p[1] : 0.0;
  return v;
}

template <int B_IS_KN, bool FULL>
__global__ void __launch_bounds__(256, 2)
gemm_minus_kernel(GemmArgs g)
{
  __shared__ __attribute__((aligned(16))) double sA[2][GT_ASZ];
  __shared__ __attribute__((aligned(16))) double sB[2][GT_BSZ];

  /* XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give
     each XCD a contiguous run of tiles (bijective for any grid size) */
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  const unsigned tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  int tm, tn;
  if (!g.lower_only) {
    tm = (int)(tile / g.tiles_n); tn = (int)(tile % g.tiles_n);
  } else {
    /* only the lower trapezoid is launched (row tm holds min(tm+1, tiles_n) tiles), so every XCD
       gets the same number of equally heavy tiles; consecutive ids share the A row panel */
    const unsigned tri = (unsigned)g.tiles_n * (unsigned)(g.tiles_n + 1) / 2;
    if (tile < tri) {
      tm = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
      while ((unsigned)(tm + 1) * (unsigned)(tm + 2) / 2 <= tile) tm++;
      while ((unsigned)tm * (unsigned)(tm + 1) / 2 > tile) tm--;
      tn = (int)(tile - (unsigned)tm * (unsigned)(tm + 1) / 2);
    } else {
      const unsigned t2 = tile - tri;
      tm = g.tiles_n + (int)(t2 / g.tiles_n); tn = (int)(t2 % g.tiles_n);
    }
  }

  const size_t row0 = (size_t)tm * GT_BM, col0 = (size_t)tn * GT_BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  const bool vecA = ((g.lda & 1) == 0) && ((((uintptr_t)g.A) & 15) == 0);
  const bool vecB = ((g.ldb & 1) == 0) && ((((uintptr_t)g.B) & 15) == 0);

  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  double2 ra[4], rb[4];

  auto fetch = [&](size_t k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + 256 * i;
      if (FULL) {   /* whole tiles, 16-byte aligned operands: no predicates */
        { const int rr = c >> 3, kc = (c & 7) * 2;
          ra[i] = *reinterpret_cast<const double2 *>(g.A + (row0 + rr) * g.lda + k0 + kc); }
        if (B_IS_KN) { const int kk = c >> 6, nc = (c & 63) * 2;
          rb[i] = *reinterpret_cast<const double2 *>(g.B + (k0 + kk) * g.ldb + col0 + nc); }
        else { const int rr = c >> 3, kc = (c & 7) * 2;
          rb[i] = *reinterpret_cast<const double2 *>(g.B + (col0 + rr) * g.ldb + k0 + kc); }
        continue;
      }
      { /* A: 128 rows x 8 chunks of 2 */
        const int rr = c >> 3, kc = (c & 7) * 2;
        const size_t grow = row0 + rr, gk = k0 + kc;
        const bool okr = grow < g.m;
        ra[i] = ld2(g.A + grow * g.lda + gk, okr && gk < g.k, okr && gk + 1 < g.k, vecA);
      }
      if (B_IS_KN) { /* B[k][n]: 16 rows x 64 chunks of 2 */
        const int kk = c >> 6, nc = (c & 63) * 2;
        const size_t gk = k0 + kk, gcol = col0 + nc;
        const bool okk = gk < g.k;
        rb[i] = ld2(g.B + gk * g.ldb + gcol, okk && gcol < g.n, okk && gcol + 1 < g.n, vecB);
      } else { /* B[n][k]: 128 rows x 8 chunks of 2 */
        const int rr = c >> 3, kc = (c & 7) * 2;
        const size_t gn = col0 + rr, gk = k0 + kc;
        const bool okn = gn < g.n;
        rb[i] = ld2(g.B + gn * g.ldb + gk, okn && gk < g.k, okn && gk + 1 < g.k, vecB);
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + 256 * i;
      { const int rr = c >> 3, kc = (c & 7) * 2;
        *reinterpret_cast<double2 *>(&sA[buf][rr * GT_LDA + kc]) = ra[i]; }
      if (B_IS_KN) { const int kk = c >> 6, nc = (c & 63) * 2;
        *reinterpret_cast<double2 *>(&sB[buf][kk * GT_LDBN + nc]) = rb[i]; }
      else { const int rr = c >> 3, kc = (c & 7) * 2;
        *reinterpret_cast<double2 *>(&sB[buf][rr * GT_LDA + kc]) = rb[i]; }
    }
  };

  const size_t nsteps = (g.k + GT_BK - 1) / GT_BK;
  fetch(0);
  stash(0);
  __syncthreads();

  for (size_t s = 0; s < nsteps; s++) {
    const int buf = (int)(s & 1);
    if (s + 1 < nsteps) fetch((s + 1) * GT_BK);   /* in flight under the MFMAs */
    const double *a_base = &sA[buf][(wr * 64 + fr) * GT_LDA + fq];
    const double *b_base = B_IS_KN ? &sB[buf][fq * GT_LDBN + wc * 64 + fr] : &sB[buf][(wc * 64 + fr) * GT_LDA + fq];
#pragma unroll
    for (int kk = 0; kk < GT_BK / 4; kk++) {
      double af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) af[i] = a_base[i * 16 * GT_LDA + kk * 4];
#pragma unroll
      for (int j = 0; j < 4; j++) bf[j] = B_IS_KN ? b_base[kk * 4 * GT_LDBN + j * 16] : b_base[j * 16 * GT_LDA + kk * 4];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nsteps) stash(buf ^ 1);           /* other buffer: last read before the previous barrier */
    __syncthreads();
  }

  /* epilogue: C -= acc.  D[row=(lane>>4)+4*reg][col=lane&15] */
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const size_t gcol = col0 + wc * 64 + j * 16 + fr;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const size_t grow = row0 + wr * 64 + i * 16 + fq + 4 * rg;
        if ((FULL || (grow < g.m && gcol < g.n)) && (!g.lower_only || gcol <= grow)) {
          double *p = g.C + grow * g.ldc + gcol;
          *p = *p - acc[i][j][rg];
        }
      }
    }
  }
}


/* ------------------------------------------------------------------------ */
/* Full-tile NT variant with direct-to-LDS loads (global_load_lds_dwordx4) and a 3-stage ring:
   the operand tiles of K-steps s+1 and s+2 are in flight while step s feeds the MFMAs, no VGPR
   staging, no ds_write pass, one raw s_barrier + one counted vmcnt wait per K-step.
   LDS image per stage and operand: 128 rows x 16 doubles, unpadded (a DMA wave-instruction writes
   1 KiB linearly = 8 rows), 16-byte slot index XOR-swizzled with (row>>1)&7 -- applied to the
   per-lane SOURCE address of the DMA and to the fragment read, so the (row = lane&15,
   k = lane>>4) ds_read_b64 stays bank-conflict free. */
#define DM_STAGES 3

__device__ __forceinline__ void dma16(const double *gsrc, double *ldst)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
}

/* WR = wave rows: 2 -> 128x128 tile, 4 waves, 1 wave/SIMD;  4 -> 256x128 tile, 8 waves, 2 waves/SIMD
   (a wave's DMA issue / wait gaps are then covered by its SIMD partner's MFMAs). */
template <int WR>
__global__ void __launch_bounds__(128 * WR, 1)
gemm_minus_dma_nt_kernel(GemmArgs g)
{
  constexpr int BM = 64 * WR;
  constexpr int NW = 2 * WR;                           /* waves */
  constexpr int A_TILE = BM * GT_BK, B_TILE = GT_BN * GT_BK;
  constexpr int A_CH = BM / 8 / NW, B_CH = (GT_BN / 8) / NW;   /* 1-KiB chunks per wave: 4,4 or 4,2 */
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *sA = smem;                                   /* [DM_STAGES][A_TILE] */
  double *sB = smem + DM_STAGES * A_TILE;

  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  const unsigned tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  int tm, tn;
  if (!g.lower_only) {
    tm = (int)(tile / g.tiles_n); tn = (int)(tile % g.tiles_n);
  } else {
    /* row tm of tiles holds min(R*(tm+1), tiles_n) active tiles, R = BM/128 */
    constexpr unsigned R = BM / GT_BN;
    const unsigned full_rows = (unsigned)g.tiles_n / R;             /* rows of the triangular part */
    const unsigned tri = R * full_rows * (full_rows + 1) / 2;
    if (tile < tri) {
      const double t = (double)tile / (double)R;
      tm = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
      while (R * (unsigned)(tm + 1) * (unsigned)(tm + 2) / 2 <= tile) tm++;
      while (R * (unsigned)tm * (unsigned)(tm + 1) / 2 > tile) tm--;
      tn = (int)(tile - R * (unsigned)tm * (unsigned)(tm + 1) / 2);
    } else {
      const unsigned t2 = tile - tri;
      tm = (int)full_rows + (int)(t2 / g.tiles_n); tn = (int)(t2 % g.tiles_n);
    }
  }
  const size_t row0 = (size_t)tm * BM, col0 = (size_t)tn * GT_BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  /* DMA source addresses of this lane */
  const double *srcA[A_CH], *srcB[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; i++) {
    const int rr = (wave * A_CH + i) * 8 + (lane >> 3);
    srcA[i] = g.A + (row0 + rr) * g.lda + (((lane & 7) ^ ((rr >> 1) & 7)) << 1);
  }
#pragma unroll
  for (int i = 0; i < B_CH; i++) {
    const int rr = (wave * B_CH + i) * 8 + (lane >> 3);
    srcB[i] = g.B + (col0 + rr) * g.ldb + (((lane & 7) ^ ((rr >> 1) & 7)) << 1);
  }
  auto issue = [&](int stage, size_t k0) {
#pragma unroll
    for (int i = 0; i < A_CH; i++) dma16(srcA[i] + k0, sA + stage * A_TILE + (wave * A_CH + i) * 128);
#pragma unroll
    for (int i = 0; i < B_CH; i++) dma16(srcB[i] + k0, sB + stage * B_TILE + (wave * B_CH + i) * 128);
  };

  /* fragment read offsets (doubles) within a stage: row*16 + ((slot ^ sw)*2 + (k&1)) */
  const int sw = (fr >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) koff[kk] = (((kk * 2 + (fq >> 1)) ^ sw) << 1) + (fq & 1);
  const int arow = (wr * 64 + fr) * GT_BK, brow = (wc * 64 + fr) * GT_BK;

  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const size_t nsteps = g.k / GT_BK;
  issue(0, 0);
  if (nsteps > 1) issue(1, GT_BK);

  for (size_t s = 0; s < nsteps; s++) {
    /* stage s has landed once at most the DMAs of stage s+1 are still outstanding */
    if (s + 1 < nsteps) {
      if (A_CH + B_CH == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nsteps) issue((int)((s + 2) % DM_STAGES), (s + 2) * GT_BK);   /* ring slot read at step s-1 */
    const double *a_base = sA + (s % DM_STAGES) * A_TILE + arow;
    const double *b_base = sB + (s % DM_STAGES) * B_TILE + brow;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      double af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) af[i] = a_base[i * 16 * GT_BK + koff[kk]];
#pragma unroll
      for (int j = 0; j < 4; j++) bf[j] = b_base[j * 16 * GT_BK + koff[kk]];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const size_t gcol = col0 + wc * 64 + j * 16 + fr;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const size_t grow = row0 + wr * 64 + i * 16 + fq + 4 * rg;
        if (!g.lower_only || gcol <= grow) {
          double *p = g.C + grow * g.ldc + gcol;
          *p = *p - acc[i][j][rg];
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------ */
/* Stream-K form of the DMA kernel.  With one workgroup per tile the launch time is quantised in
   whole tiles per CU: 8192^2 lower / 256x128 tiles = 1056 tiles = 4.1 rounds on 256 CUs (a fifth,
   almost empty round costs 20 %), 2048^2 lower = 136 tiles of 128x128 leaves 120 CUs idle.  Here
   the launch is G persistent workgroups (G <= #CUs) and the unit of work is one K-step of one
   tile: workgroup w owns the contiguous range [total*w/G, total*(w+1)/G) of the (tile, K-step)
   space, so every CU gets the same number of MFMA steps whatever the tile count.
   A tile whose K range is cut is finished by the workgroup holding its FIRST K-step (the owner);
   the others store their accumulators to their private slot and raise a flag.  A non-owner
   segment is always the first thing its workgroup does, so the owner (which reaches the tile
   last) finds the flags already set and never waits on a workgroup that itself waits.  Partials
   are added in ascending workgroup order: the summation order is a pure function of the shape,
   results are reproducible run to run.
   Partials and flags are exchanged with agent-scope atomic loads/stores (sc1: coherent across
   the 8 XCD L2s) and an explicit vmcnt(0) + barrier before the flag, so no L2 writeback /
   invalidate is needed.  The owner clears the flag it consumed: launches on one stream are
   serialised, so the buffers are reusable by the next launch (and by a hipGraph replay). */
struct StreamK {
  unsigned steps;               /* K-steps per tile */
  unsigned total;               /* tiles * steps */
  unsigned base, rem;           /* total = G*base + rem: workgroup w starts at w*base + min(w, rem) */
  unsigned dp_rounds;           /* whole tiles first: round r gives tile r*G + w to workgroup w; the (tile, K-step) space
                                   of total/steps tiles AFTER those dp_rounds*G tiles is then split as above */
  double *partial;              /* [G][BM*128] */
  unsigned *flags;              /* [G] */
};

template <int BM, int BN>
__device__ __forceinline__ void decode_tile(const GemmArgs &g, unsigned tile, int &tm, int &tn)
{
  if (!g.lower_only) {
    tm = (int)(tile / g.tiles_n); tn = (int)(tile % g.tiles_n);
  } else {
    constexpr unsigned R = BM / BN;
    const unsigned full_rows = (unsigned)g.tiles_n / R;
    const unsigned tri = R * full_rows * (full_rows + 1) / 2;
    if (tile < tri) {
      const double t = (double)tile / (double)R;
      tm = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
      while (R * (unsigned)(tm + 1) * (unsigned)(tm + 2) / 2 <= tile) tm++;
      while (R * (unsigned)tm * (unsigned)(tm + 1) / 2 > tile) tm--;
      tn = (int)(tile - R * (unsigned)tm * (unsigned)(tm + 1) / 2);
    } else {
      const unsigned t2 = tile - tri;
      tm = (int)full_rows + (int)(t2 / g.tiles_n); tn = (int)(t2 % g.tiles_n);
    }
  }
}

/* Tile configurations (block tile BM x BN, wave tile WM x WN):
     256x128 / 64x64, 8 waves, 144 KiB LDS -- large updates, 2 waves per SIMD;
     128x128 / 64x64, 4 waves,  96 KiB
      64x64  / 32x32, 4 waves,  48 KiB     -- updates with few 128-tiles (the K <= 512 levels of the
                                             recursions): 4x the workgroups, a 128-wide panel update
                                             is otherwise one 13.7 us tile per CU on a fraction of the CUs */
/* SS consecutive 16-wide K sub-steps form one "group" = the unit between two barriers (and the unit of the
   stream-K split); ST groups are resident in the LDS ring.  SS = 1, ST = 3 is the pipeline described above.
   The small 64x64 tile spends only 0.43 us of MFMA work per 16-wide step -- less than a DMA round trip and
   comparable to a barrier -- so it runs SS = 4, ST = 2 (128 KiB): a K = 128 update has its WHOLE operand
   panel in flight from the first instruction and crosses two barriers instead of eight. */
/* PIPE (ST = 3, SS = 1 only): the K loop is software-pipelined ACROSS its barrier.  In the plain loop every wave
   meets at the barrier with an empty MFMA queue and must then wait out an LDS round trip before its first
   MFMA of the new step: a bubble of a few hundred cycles per 8192-cycle step on every SIMD.  Here the last
   quarter of a step's MFMAs (its fragments are already in registers) is issued AFTER the barrier and after the
   first fragment loads of the next step, so the matrix pipe stays fed through the rendezvous; the barrier also
   moves one quarter-step earlier relative to the DMA ring, which now runs three steps ahead instead of two. */
/* BKN: the B operand is stored [k][n] (C -= A B, the N.N updates of the LU route) instead of [n][k].  Its LDS image per
   step is then 16 k-rows of BN doubles, each row one linear 1-KiB DMA wave-instruction (BN = 128), rows pitched
   BN + 16 doubles apart so that the (k = lane>>4, n = lane&15) fragment read -- 16 consecutive doubles per k-row,
   k-rows 32 banks apart -- is conflict-free; no swizzle needed. */
template <int BM, int BN, int WM, int WN, int SS = 1, int ST = DM_STAGES, bool PIPE = false, bool BKN = false>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
gemm_minus_streamk_kernel(GemmArgs g, StreamK x)
{
  static_assert(ST == 3 || ST == 2, "ring depth");
  static_assert(!PIPE || (ST == 3 && SS == 1), "pipelined loop: three-deep ring of single steps");
  static_assert(!BKN || BN == 128, "k-row = one 1-KiB DMA instruction");
  constexpr int WCOLS = BN / WN;                        /* waves along n */
  constexpr int NW = (BM / WM) * WCOLS;
  constexpr int NT = NW * 64;                           /* threads */
  constexpr int FM = WM / 16, FN = WN / 16;             /* MFMA fragments per wave */
  constexpr int BPITCH = BN + 16;                       /* BKN: doubles between the k-rows of the B image */
  constexpr int A_TILE = BM * GT_BK, B_TILE = BKN ? GT_BK * BPITCH : BN * GT_BK;
  constexpr int A_CH = BM / 8 / NW, B_CH = (BN / 8) / NW;
  static_assert(A_CH >= 1 && B_CH >= 1 && A_CH * 8 * NW == BM && B_CH * 8 * NW == BN, "tile / wave split");
  constexpr int PER_GROUP = SS * (A_CH + B_CH);         /* DMA wave-instructions a wave issues per group */
  static_assert(PER_GROUP <= 32, "vmcnt immediate");
  constexpr size_t GK = (size_t)GT_BK * SS;             /* K extent of a group */
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *sA = smem;
  double *sB = smem + ST * SS * A_TILE;

  const unsigned G = gridDim.x, bid = blockIdx.x;
  const unsigned q = G / 8, r = G % 8, xcd = bid % 8;
  const unsigned gl = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;   /* XCD-contiguous ranges */
  auto start_of = [&](unsigned w) -> unsigned { return w * x.base + (w < x.rem ? w : x.rem); };
  unsigned it = __builtin_amdgcn_readfirstlane(start_of(gl));
  const unsigned it_end = __builtin_amdgcn_readfirstlane(start_of(gl + 1));
  unsigned tile = __builtin_amdgcn_readfirstlane(it / x.steps);
  const unsigned dp_tiles = x.dp_rounds * G;
  unsigned round = 0;

  /* Whole-tile rounds first: the 32 workgroups of an XCD hold 32 consecutive tiles of the same round and
     walk K more or less in step, so operand panels are shared in their L2; then the stream-K remainder. */
  for (;;) {
    const bool dp = round < x.dp_rounds;
    if (!dp && it >= it_end) break;
    const unsigned tile_first = dp ? 0u : tile * x.steps, tile_end = tile_first + x.steps;
    const unsigned s0 = dp ? 0u : it - tile_first;
    const unsigned s1 = dp ? x.steps : (it_end < tile_end ? it_end - tile_first : x.steps);
    int tm, tn;
    {
      const unsigned tid_ = dp ? round * G + gl : dp_tiles + tile;
      decode_tile<BM, BN>(g, tid_, tm, tn);
    }
    const size_t row0 = (size_t)tm * BM, col0 = (size_t)tn * BN;

    double4_t acc[FM][FN];
    /* 64 x 64 tiles (the K <= 256 levels: one short tile per workgroup): the owner fetches its C tile BEFORE the K loop
       -- 16 values per thread, in flight with the first DMA groups -- so the epilogue is a plain store instead of a
       read-modify-write whose load latency nothing covers (small: the K = 128 level of C3 1.228 -> 1.209 ms, C4 init
       7.41 -> 7.26 ms, C2 init 2.87 -> 2.81 ms) */
    constexpr bool CPRE = BM == 64 && BN == 64;
    double4_t cpre[CPRE ? FM : 1][CPRE ? FN : 1];
    if constexpr (CPRE) {
      if (s0 == 0) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int wr = wave / WCOLS, wc = wave % WCOLS;
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int i = 0; i < FM; i++)
#pragma unroll
          for (int j = 0; j < FN; j++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++)
              cpre[i][j][rg] = g.C[(row0 + wr * WM + i * 16 + fq + 4 * rg) * g.ldc + col0 + wc * WN + j * 16 + fr];
      }
    }
    {
      /* Everything lane-dependent is derived from a laundered thread id INSIDE the segment: left to
         itself the compiler hoists it all out of the while loop, runs out of VGPRs (the 8-wave
         variant has 256 including the 128 accumulators) and spills operands of the K loop -- and
         a scratch reload's s_waitcnt vmcnt(0) also drains the DMA ring. */
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, wave = tid >> 6;
      const int wr = wave / WCOLS, wc = wave % WCOLS;
      const int fr = lane & 15, fq = lane >> 4;
      const int sw = (fr >> 1) & 7;
      int koff[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) koff[kk] = (((kk * 2 + (fq >> 1)) ^ sw) << 1) + (fq & 1);
      const int arow = (wr * WM + fr) * GT_BK, brow = BKN ? fq * BPITCH + wc * WN + fr : (wc * WN + fr) * GT_BK;

      const double *srcA[A_CH], *srcB[B_CH];
#pragma unroll
      for (int i = 0; i < A_CH; i++) {
        const int rr = (wave * A_CH + i) * 8 + (lane >> 3);
        srcA[i] = g.A + (row0 + rr) * g.lda + (((lane & 7) ^ ((rr >> 1) & 7)) << 1);
      }
#pragma unroll
      for (int i = 0; i < B_CH; i++) {
        const int rr = (wave * B_CH + i) * 8 + (lane >> 3);
        if constexpr (BKN) srcB[i] = g.B + (size_t)(wave * B_CH + i) * g.ldb + col0 + lane * 2;   /* chunk = k-row of the step */
        else srcB[i] = g.B + (col0 + rr) * g.ldb + (((lane & 7) ^ ((rr >> 1) & 7)) << 1);
      }
      /* one group = SS sub-steps, each its own [rows][16] swizzled image (slot = stage * SS + sub-step) */
      auto issue = [&](int stage, size_t k0) {
#pragma unroll
        for (int ss = 0; ss < SS; ss++) {
#pragma unroll
          for (int i = 0; i < A_CH; i++) dma16(srcA[i] + k0 + ss * GT_BK, sA + (stage * SS + ss) * A_TILE + (wave * A_CH + i) * 128);
#pragma unroll
          for (int i = 0; i < B_CH; i++) {
            if constexpr (BKN) dma16(srcB[i] + (k0 + ss * GT_BK) * g.ldb, sB + (stage * SS + ss) * B_TILE + (wave * B_CH + i) * BPITCH);
            else dma16(srcB[i] + k0 + ss * GT_BK, sB + (stage * SS + ss) * B_TILE + (wave * B_CH + i) * 128);
          }
        }
      };

#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

      __syncthreads();                                   /* the ring is free: previous segment fully read */
      if constexpr (PIPE) {
        auto wait_keep = [&](int groups) {               /* all but the last `groups` issued groups have landed */
          if (groups >= 2) {
            if constexpr (PER_GROUP == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (PER_GROUP == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          } else if (groups == 1) {
            if constexpr (PER_GROUP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (PER_GROUP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto load_frag = [&](int stage, int kk, double (&af)[FM], double (&bf)[FN]) {
          const double *a_base = sA + stage * A_TILE + arow;
          const double *b_base = sB + stage * B_TILE + brow;
#pragma unroll
          for (int i = 0; i < FM; i++) af[i] = a_base[i * 16 * GT_BK + koff[kk]];
#pragma unroll
          for (int j = 0; j < FN; j++) bf[j] = BKN ? b_base[kk * 4 * BPITCH + j * 16] : b_base[j * 16 * GT_BK + koff[kk]];
        };
        auto mma = [&](const double (&af)[FM], const double (&bf)[FN]) {
#pragma unroll
          for (int i = 0; i < FM; i++)
#pragma unroll
            for (int j = 0; j < FN; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        };
        const unsigned nst = s1 - s0;
        issue(0, (size_t)s0 * GK);
        if (nst > 1) issue(1, (size_t)(s0 + 1) * GK);
        if (nst > 2) issue(2, (size_t)(s0 + 2) * GK);
        wait_keep(nst > 2 ? 2 : (int)nst - 1);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        double af0[FM], bf0[FN], af1[FM], bf1[FN];
        load_frag(0, 0, af0, bf0);
        /* all steps but the last: the rendezvous sits between the third and the fourth quarter of the step */
        for (unsigned rel = 0; rel + 1 < nst; rel++) {
          const int stage = (int)(rel % 3);
          load_frag(stage, 1, af1, bf1);
          mma(af0, bf0);
          load_frag(stage, 2, af0, bf0);
          mma(af1, bf1);
          load_frag(stage, 3, af1, bf1);
          mma(af0, bf0);
          /* (stage, 3) is in af1/bf1; every LDS read of `stage` has returned; stage+1 has landed for this wave once
             only the group issued after it (rel+2, if it exists) is outstanding; after the barrier that holds for
             every wave, and `stage` may be refilled */
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          /* tell the compiler's s_waitcnt pass that af1 / bf1 are settled HERE (it cannot see through the asm wait
             above and would otherwise put a full lgkmcnt(0) between the next step's loads and the deferred MFMAs) */
#pragma unroll
          for (int i = 0; i < FM; i++) asm volatile("" : "+v"(af1[i]));
#pragma unroll
          for (int j = 0; j < FN; j++) asm volatile("" : "+v"(bf1[j]));
          wait_keep(rel + 2 < nst ? 1 : 0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if (rel + 3 < nst) issue(stage, (size_t)(s0 + rel + 3) * GK);
          load_frag((int)((rel + 1) % 3), 0, af0, bf0);
          __builtin_amdgcn_sched_barrier(0);             /* keep the loads in front of the deferred MFMAs */
          mma(af1, bf1);                                 /* runs while the new fragments fly (counted lgkmcnt: no merge of paths here) */
        }
        {
          const int stage = (int)((nst - 1) % 3);
          load_frag(stage, 1, af1, bf1);
          mma(af0, bf0);
          load_frag(stage, 2, af0, bf0);
          mma(af1, bf1);
          load_frag(stage, 3, af1, bf1);
          mma(af0, bf0);
          mma(af1, bf1);
        }
      } else {
      issue(0, (size_t)s0 * GK);
      if (s0 + 1 < s1) issue(1, (size_t)(s0 + 1) * GK);
      for (unsigned s = s0; s < s1; s++) {
        const unsigned rel = s - s0;
        /* group s has landed once at most the DMAs of group s+1 are still outstanding */
        if (s + 1 < s1) {
          if constexpr (PER_GROUP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else if constexpr (PER_GROUP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else if constexpr (PER_GROUP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else { static_assert(PER_GROUP == 4 || PER_GROUP == 6 || PER_GROUP == 8 || PER_GROUP == 16, "vmcnt immediate"); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ST == 3) {
          if (s + 2 < s1) issue((int)((rel + 2) % 3), (size_t)(s + 2) * GK);   /* ring slot read at step s-1 */
        }
        const int stage = (int)(rel % ST);
#pragma unroll
        for (int ss = 0; ss < SS; ss++) {
          const double *a_base = sA + (stage * SS + ss) * A_TILE + arow;
          const double *b_base = sB + (stage * SS + ss) * B_TILE + brow;
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            double af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; i++) af[i] = a_base[i * 16 * GT_BK + koff[kk]];
#pragma unroll
            for (int j = 0; j < FN; j++) bf[j] = BKN ? b_base[kk * 4 * BPITCH + j * 16] : b_base[j * 16 * GT_BK + koff[kk]];
#pragma unroll
            for (int i = 0; i < FM; i++)
#pragma unroll
              for (int j = 0; j < FN; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
          }
        }
        if constexpr (ST == 2) {
          /* two-deep ring: the stage just read is refilled with group s+2 once every wave has left it */
          if (s + 2 < s1) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue(stage, (size_t)(s + 2) * GK);
          }
        }
      }
      }
    }

    /* epilogue: its own laundered copies, so nothing of it is live across the K loop */
    int tid = threadIdx.x;
    size_t row0e = row0, col0e = col0;
    asm volatile("" : "+v"(tid), "+s"(row0e), "+s"(col0e));
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int fr = lane & 15, fq = lane >> 4;
    if (s0 != 0) {
      /* not the owner: publish the partial tile */
      double *pp = x.partial + (size_t)gl * (BM * BN) + tid;
#pragma unroll
      for (int i = 0; i < FM; i++) {
#pragma unroll
        for (int j = 0; j < FN; j++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++)
            __hip_atomic_store(pp + ((i * FN + j) * 4 + rg) * NT, acc[i][j][rg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(x.flags + gl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (s1 < x.steps) {
        for (unsigned w = gl + 1; w < G; w++) {
          if (start_of(w) >= tile_end) break;
          if (tid == 0)
            while (__hip_atomic_load(x.flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
          __syncthreads();
          const double *pp = x.partial + (size_t)w * (BM * BN) + tid;
#pragma unroll
          for (int i = 0; i < FM; i++) {
#pragma unroll
            for (int j = 0; j < FN; j++)
#pragma unroll
              for (int rg = 0; rg < 4; rg++)
                acc[i][j][rg] += __hip_atomic_load(pp + ((i * FN + j) * 4 + rg) * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);            /* 16 loads in flight at a time */
          }
          if (tid == 0) __hip_atomic_store(x.flags + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int i = 0; i < FM; i++) {
#pragma unroll
        for (int j = 0; j < FN; j++) {
          const size_t gcol = col0e + wc * WN + j * 16 + fr;
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            const size_t grow = row0e + wr * WM + i * 16 + fq + 4 * rg;
            if (!g.lower_only || gcol <= grow) {
              double *p = g.C + grow * g.ldc + gcol;
              if constexpr (CPRE) *p = cpre[i][j][rg] - acc[i][j][rg];
              else *p = *p - acc[i][j][rg];
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (dp) round++;
    else { it = tile_first + s1; tile++; }
  }
}

int sinterp_streamk_prepare(gsl_sinterp_hip_ctx *ctx)
{
  if (ctx->sk_wgs) return ST_SUCCESS;
  static const bool off = getenv("GSL_SINTERP_NO_STREAMK") && getenv("GSL_SINTERP_NO_STREAMK")[0] == '1';
  if (off) return ST_SUCCESS;
  int cus = 0;
  HIP_OK(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
  if (cus <= 0) return ST_SUCCESS;
  HIP_OK(ctx, hipMalloc((void **)&ctx->d_sk_partial, (size_t)cus * 256 * GT_BN * sizeof(double)));
  HIP_OK(ctx, hipMalloc((void **)&ctx->d_sk_flags, (size_t)cus * 2 * sizeof(unsigned)));   /* 2 x: the two-workgroups-per-CU variant */
  HIP_OK(ctx, hipMemset(ctx->d_sk_flags, 0, (size_t)cus * 2 * sizeof(unsigned)));
  HIP_OK(ctx, hipDeviceSynchronize());
  ctx->sk_wgs = cus;
  return ST_SUCCESS;
}

/* ------------------------------------------------------------------------ */
/* Small-K, skinny-N update (the K = 32 / 64 levels of the recursions): C[m x BN] -= A[m x K] B^T,
   B stored [BN][K], K <= 64.  Everything a workgroup needs -- its 128 x K slice of A, all of B and
   its C tile -- is fetched in ONE global round trip; the A slice is negated on its way into LDS
   and the accumulators start as C, so the epilogue is a plain store. */
template <int BN>
__global__ void __launch_bounds__(256)
gemm_minus_smallk_kernel(GemmArgs g)
{
  constexpr int NF = BN / 16;                  /* n fragments per wave */
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int K = (int)g.k, LD = K + 2;
  double *sA = smem;                           /* [128][LD] */
  double *sB = smem + 128 * LD;                /* [BN][LD]  */
  const size_t row0 = (size_t)blockIdx.x * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int kch = K / 2;                       /* 16-byte chunks per row */

  /* C tile -> accumulators (row clamped, masked later) */
  double4_t acc[2][NF];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NF; j++) {
      const size_t gcol = j * 16 + fr;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const size_t grow = row0 + wave * 32 + i * 16 + fq + 4 * rg;
        const size_t rcl = grow < g.m ? grow : g.m - 1;
        acc[i][j][rg] = g.C[rcl * g.ldc + gcol];
      }
    }
  /* operands -> LDS */
  for (int c = tid; c < 128 * kch; c += 256) {
    const int rr = c / kch, kc = (c % kch) * 2;
    const size_t grow = row0 + rr, rcl = grow < g.m ? grow : g.m - 1;
    const double2 v = *reinterpret_cast<const double2 *>(g.A + rcl * g.lda + kc);
    *reinterpret_cast<double2 *>(&sA[rr * LD + kc]) = make_double2(-v.x, -v.y);
  }
  for (int c = tid; c < BN * kch; c += 256) {
    const int rr = c / kch, kc = (c % kch) * 2;
    *reinterpret_cast<double2 *>(&sB[rr * LD + kc]) = *reinterpret_cast<const double2 *>(g.B + (size_t)rr * g.ldb + kc);
  }
  __syncthreads();
  const double *a_base = sA + (wave * 32 + fr) * LD + fq;
  const double *b_base = sB + fr * LD + fq;
  for (int kk = 0; kk < K; kk += 4) {
    double af[2], bf[NF];
#pragma unroll
    for (int i = 0; i < 2; i++) af[i] = a_base[i * 16 * LD + kk];
#pragma unroll
    for (int j = 0; j < NF; j++) bf[j] = b_base[j * 16 * LD + kk];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < NF; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NF; j++) {
      const size_t gcol = j * 16 + fr;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const size_t grow = row0 + wave * 32 + i * 16 + fq + 4 * rg;
        if (grow < g.m && (!g.lower_only || gcol <= grow)) g.C[grow * g.ldc + gcol] = acc[i][j][rg];
      }
    }
}

int sinterp_gemm_minus(gsl_sinterp_hip_ctx *ctx, size_t m, size_t n, size_t k, const double *A, size_t lda,
                       const double *B, size_t ldb, int b_is_kn, double *C, size_t ldc, int lower_only)
{
  if (m == 0 || n == 0 || k == 0) return ST_SUCCESS;
  GemmArgs g;
  g.m = m; g.n = n; g.k = k; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.lower_only = lower_only;
  g.tiles_m = (int)((m + GT_BM - 1) / GT_BM);
  g.tiles_n = (int)((n + GT_BN - 1) / GT_BN);
  if (lower_only && g.tiles_m < g.tiles_n) g.tiles_n = g.tiles_m;   /* columns right of the square part are all above the diagonal */
  unsigned grid = (unsigned)g.tiles_m * (unsigned)g.tiles_n;
  if (lower_only) {
    const unsigned tn_ = (unsigned)g.tiles_n;
    grid = tn_ * (tn_ + 1) / 2 + ((unsigned)g.tiles_m - tn_) * tn_;
  }
  g.n_active = grid;
  const bool full = (m % GT_BM == 0) && (n % GT_BN == 0) && (k % GT_BK == 0) && ((lda & 1) == 0) && ((ldb & 1) == 0) &&
                    ((((uintptr_t)A) & 15) == 0) && ((((uintptr_t)B) & 15) == 0);
  /* skinny small-K update: one round trip */
  if (!b_is_kn && (n == 32 || n == 64) && k <= 64 && (k % 4) == 0 && k >= 4 && ((lda & 1) == 0) && ((ldb & 1) == 0) &&
      ((((uintptr_t)A) & 15) == 0) && ((((uintptr_t)B) & 15) == 0)) {
    const size_t lds = (size_t)(128 + n) * (k + 2) * sizeof(double);
    if (n == 32) {
      { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_smallk_kernel<32>, 160 * 1024); if (ast) return ast; }
      hipLaunchKernelGGL(gemm_minus_smallk_kernel<32>, dim3((unsigned)g.tiles_m), dim3(256), lds, ctx->stream, g);
    } else {
      { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_smallk_kernel<64>, 160 * 1024); if (ast) return ast; }
      hipLaunchKernelGGL(gemm_minus_smallk_kernel<64>, dim3((unsigned)g.tiles_m), dim3(256), lds, ctx->stream, g);
    }
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  static const bool no_dma = getenv("GSL_SINTERP_NO_DMA_GEMM") && getenv("GSL_SINTERP_NO_DMA_GEMM")[0] == '1';
  if (full && !b_is_kn && k >= 4 * GT_BK && !no_dma) {
    static const bool no_w8 = getenv("GSL_SINTERP_NO_GEMM8") && getenv("GSL_SINTERP_NO_GEMM8")[0] == '1';
    if (ctx->sk_wgs > 0) {
      /* stream-K: G persistent workgroups share the (tile, K-step) space evenly */
      StreamK x;
      x.steps = (unsigned)(k / GT_BK);                   /* groups per tile; rescaled below for the grouped 64x64 variant */
      x.partial = ctx->d_sk_partial;
      x.flags = ctx->d_sk_flags;
      GemmArgs h = g;
      unsigned tiles = grid;
      int cfg = 1;                                       /* 0: 256x128, 1: 128x128, 2: 64x64 */
      if (!no_w8 && (m % 256 == 0) && (!lower_only || (g.tiles_n % 2) == 0)) {
        GemmArgs h8 = g;
        h8.tiles_m = (int)(m / 256);
        unsigned t8 = (unsigned)h8.tiles_m * (unsigned)h8.tiles_n;
        if (lower_only) {
          const unsigned fr_ = (unsigned)h8.tiles_n / 2;
          t8 = 2 * fr_ * (fr_ + 1) / 2 + ((unsigned)h8.tiles_m - fr_) * (unsigned)h8.tiles_n;
        }
        /* the big tile only pays when every CU gets a few K-steps of it */
        if ((unsigned long long)t8 * x.steps >= 16ull * (unsigned)ctx->sk_wgs) { cfg = 0; h = h8; tiles = t8; }
      }
      static const bool no_t64 = getenv("GSL_SINTERP_NO_GEMM64") && getenv("GSL_SINTERP_NO_GEMM64")[0] == '1';
      if (cfg == 1 && !no_t64 && 2u * grid <= (unsigned)ctx->sk_wgs) {
        /* 128-tiles for at most half of the CUs: quarter tiles, 4x the workgroups (measured: with
           more tiles than that the extra prologues/epilogues cost more than the idle CUs) */
        cfg = 2;
        h.tiles_m = (int)(m / 64); h.tiles_n = (int)(n / 64);
        if (lower_only && h.tiles_m < h.tiles_n) h.tiles_n = h.tiles_m;
        tiles = (unsigned)h.tiles_m * (unsigned)h.tiles_n;
        if (lower_only) { const unsigned tn_ = (unsigned)h.tiles_n; tiles = tn_ * (tn_ + 1) / 2 + ((unsigned)h.tiles_m - tn_) * tn_; }
      }
      /* developer override (tools/gemm_cfg_sweep.py): force the tile configuration where the shape allows it */
      if (getenv("GSL_SINTERP_GEMM_CFG")) {
        const int want_cfg = atoi(getenv("GSL_SINTERP_GEMM_CFG"));
        if (want_cfg == 1 || (want_cfg == 2 && m % 64 == 0 && n % 64 == 0)) {
          cfg = want_cfg; h = g; tiles = grid;
          if (want_cfg == 2) {
            h.tiles_m = (int)(m / 64); h.tiles_n = (int)(n / 64);
            if (lower_only && h.tiles_m < h.tiles_n) h.tiles_n = h.tiles_m;
            tiles = (unsigned)h.tiles_m * (unsigned)h.tiles_n;
            if (lower_only) { const unsigned tn_ = (unsigned)h.tiles_n; tiles = tn_ * (tn_ + 1) / 2 + ((unsigned)h.tiles_m - tn_) * tn_; }
          }
        }
      }
      static const bool no_group = getenv("GSL_SINTERP_NO_GEMM_GROUP") && getenv("GSL_SINTERP_NO_GEMM_GROUP")[0] == '1';
      const bool grouped = cfg == 2 && !no_group && (k % (4 * GT_BK)) == 0;
      if (grouped) x.steps = (unsigned)(k / (4 * GT_BK));
      unsigned long long total64 = (unsigned long long)tiles * x.steps;
      unsigned long long want = total64 / (grouped ? 4 : 16);   /* >= 16 K-steps (of 16) per workgroup ... */
      if (want < tiles) want = tiles;                     /* ... but never fewer workgroups than tiles */
      const unsigned long long wg_cap = (unsigned long long)ctx->sk_wgs;
      if (want > wg_cap) want = wg_cap;
      const unsigned G = (unsigned)(want ? want : 1);
      /* many tiles: all but the last full round (and the remainder) as whole tiles */
      static const bool no_hybrid = getenv("GSL_SINTERP_NO_HYBRID_SK") && getenv("GSL_SINTERP_NO_HYBRID_SK")[0] == '1';
      x.dp_rounds = (!no_hybrid && tiles / G >= 2) ? tiles / G - 1 : 0;
      total64 = (unsigned long long)(tiles - x.dp_rounds * G) * x.steps;
      if (total64 < 0x7fffffffull) {
      x.total = (unsigned)total64; x.base = x.total / G; x.rem = x.total % G;
      static const bool no_pipe = getenv("GSL_SINTERP_NO_GEMM_PIPE") && getenv("GSL_SINTERP_NO_GEMM_PIPE")[0] == '1';
      if (cfg == 0 && !no_pipe) {
        const size_t lds = (size_t)DM_STAGES * (256 + 128) * GT_BK * sizeof(double);
        { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_streamk_kernel<256, 128, 64, 64, 1, 3, true>, (int)lds); if (ast) return ast; }
        hipLaunchKernelGGL((gemm_minus_streamk_kernel<256, 128, 64, 64, 1, 3, true>), dim3(G), dim3(512), lds, ctx->stream, h, x);
      } else if (cfg == 0) {
        const size_t lds = (size_t)DM_STAGES * (256 + 128) * GT_BK * sizeof(double);
        { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_streamk_kernel<256, 128, 64, 64>, (int)lds); if (ast) return ast; }
        hipLaunchKernelGGL((gemm_minus_streamk_kernel<256, 128, 64, 64>), dim3(G), dim3(512), lds, ctx->stream, h, x);
      } else if (cfg == 1 && !no_pipe) {
        const size_t lds = (size_t)DM_STAGES * (128 + 128) * GT_BK * sizeof(double);
        { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_streamk_kernel<128, 128, 64, 64, 1, 3, true>, (int)lds); if (ast) return ast; }
        hipLaunchKernelGGL((gemm_minus_streamk_kernel<128, 128, 64, 64, 1, 3, true>), dim3(G), dim3(256), lds, ctx->stream, h, x);
      } else if (cfg == 1) {
        const size_t lds = (size_t)DM_STAGES * (128 + 128) * GT_BK * sizeof(double);
        { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_streamk_kernel<128, 128, 64, 64>, (int)lds); if (ast) return ast; }
        hipLaunchKernelGGL((gemm_minus_streamk_kernel<128, 128, 64, 64>), dim3(G), dim3(256), lds, ctx->stream, h, x);
      } else if (grouped) {
        const size_t lds = (size_t)2 * 4 * (64 + 64) * GT_BK * sizeof(double);       /* 128 KiB */
        { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_streamk_kernel<64, 64, 32, 32, 4, 2>, (int)lds); if (ast) return ast; }
        hipLaunchKernelGGL((gemm_minus_streamk_kernel<64, 64, 32, 32, 4, 2>), dim3(G), dim3(256), lds, ctx->stream, h, x);
      } else {
        const size_t lds = (size_t)DM_STAGES * (64 + 64) * GT_BK * sizeof(double);
        { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_streamk_kernel<64, 64, 32, 32>, (int)lds); if (ast) return ast; }
        hipLaunchKernelGGL((gemm_minus_streamk_kernel<64, 64, 32, 32>), dim3(G), dim3(256), lds, ctx->stream, h, x);
      }
      LAUNCH_CHECK(ctx);
      return ST_SUCCESS;
      }
    }
    /* 256x128 tiles (8 waves) when the rows split evenly and there is enough work for every CU */
    const bool w8 = !no_w8 && (m % 256 == 0) && (!lower_only || (g.tiles_n % 2) == 0) && grid >= 1024;
    if (w8) {
      GemmArgs h = g;
      h.tiles_m = (int)(m / 256);
      unsigned grid8 = (unsigned)h.tiles_m * (unsigned)h.tiles_n;
      if (lower_only) {
        const unsigned fr_ = (unsigned)h.tiles_n / 2;
        grid8 = 2 * fr_ * (fr_ + 1) / 2 + ((unsigned)h.tiles_m - fr_) * (unsigned)h.tiles_n;
      }
      const size_t lds8 = (size_t)DM_STAGES * (256 + 128) * GT_BK * sizeof(double);   /* 144 KiB */
      { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_dma_nt_kernel<4>, (int)lds8); if (ast) return ast; }
      hipLaunchKernelGGL(gemm_minus_dma_nt_kernel<4>, dim3(grid8), dim3(512), lds8, ctx->stream, h);
      LAUNCH_CHECK(ctx);
      return ST_SUCCESS;
    }
    const size_t lds = (size_t)DM_STAGES * (128 + 128) * GT_BK * sizeof(double);     /* 96 KiB */
    { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_dma_nt_kernel<2>, (int)lds); if (ast) return ast; }
    hipLaunchKernelGGL(gemm_minus_dma_nt_kernel<2>, dim3(grid), dim3(256), lds, ctx->stream, g);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
  static const bool no_kn = getenv("GSL_SINTERP_NO_KN_STREAMK") && getenv("GSL_SINTERP_NO_KN_STREAMK")[0] == '1';
  if (b_is_kn && full && !lower_only && !no_dma && !no_kn && ctx->sk_wgs > 0 && k >= 4 * GT_BK) {
    /* C -= A B with B stored [k][n] (the N.N updates of the LU route): the stream-K DMA pipeline with a [k][n] B image */
    StreamK x;
    x.steps = (unsigned)(k / GT_BK);
    x.partial = ctx->d_sk_partial; x.flags = ctx->d_sk_flags;
    GemmArgs h = g;
    unsigned tiles = grid;
    bool big = false;
    if ((m % 256) == 0) {
      const unsigned t8 = (unsigned)(m / 256) * (unsigned)g.tiles_n;
      if ((unsigned long long)t8 * x.steps >= 16ull * (unsigned)ctx->sk_wgs) { big = true; h.tiles_m = (int)(m / 256); tiles = t8; }
    }
    unsigned long long total64 = (unsigned long long)tiles * x.steps;
    unsigned long long want = total64 / 16;
    if (want < tiles) want = tiles;
    if (want > (unsigned long long)ctx->sk_wgs) want = (unsigned long long)ctx->sk_wgs;
    const unsigned G = (unsigned)(want ? want : 1);
    x.dp_rounds = (tiles / G >= 2) ? tiles / G - 1 : 0;
    total64 = (unsigned long long)(tiles - x.dp_rounds * G) * x.steps;
    if (total64 < 0x7fffffffull) {
      x.total = (unsigned)total64; x.base = x.total / G; x.rem = x.total % G;
      if (big) {
        const size_t lds = (size_t)DM_STAGES * (256 * GT_BK + GT_BK * (128 + 16)) * sizeof(double);      /* 150 KiB */
        { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_streamk_kernel<256, 128, 64, 64, 1, 3, true, true>, (int)lds); if (ast) return ast; }
        hipLaunchKernelGGL((gemm_minus_streamk_kernel<256, 128, 64, 64, 1, 3, true, true>), dim3(G), dim3(512), lds, ctx->stream, h, x);
      } else {
        const size_t lds = (size_t)DM_STAGES * (128 * GT_BK + GT_BK * (128 + 16)) * sizeof(double);
        { int ast = sinterp_func_lds(ctx, (const void *)gemm_minus_streamk_kernel<128, 128, 64, 64, 1, 3, true, true>, (int)lds); if (ast) return ast; }
        hipLaunchKernelGGL((gemm_minus_streamk_kernel<128, 128, 64, 64, 1, 3, true, true>), dim3(G), dim3(256), lds, ctx->stream, h, x);
      }
      LAUNCH_CHECK(ctx);
      return ST_SUCCESS;
    }
  }
  if (b_is_kn) {
    if (full) hipLaunchKernelGGL((gemm_minus_kernel<1, true>), dim3(grid), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_minus_kernel<1, false>), dim3(grid), dim3(256), 0, ctx->stream, g);
  } else {
    if (full) hipLaunchKernelGGL((gemm_minus_kernel<0, true>), dim3(grid), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_minus_kernel<0, false>), dim3(grid), dim3(256), 0, ctx->stream, g);
  }
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_gemm_minus(gsl_sinterp_hip_ctx *ctx, size_t m, size_t n, size_t k, const double *d_a,
                                          size_t lda, const double *d_b, size_t ldb, int b_is_kn, double *d_c,
                                          size_t ldc, int lower_only)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  EXCLUSIVE_SECTION(ctx);
  REQUIRE(ctx, lda >= k && ldc >= n && ldb >= (b_is_kn ? n : k), ST_EINVAL);
  REQUIRE(ctx, (m == 0 || n == 0 || k == 0) || (d_a && d_b && d_c), ST_EFAULT);
  int st = sinterp_streamk_prepare(ctx);
  if (st) return st;
  return sinterp_gemm_minus(ctx, m, n, k, d_a, lda, d_b, ldb, b_is_kn, d_c, ldc, lower_only);
}
