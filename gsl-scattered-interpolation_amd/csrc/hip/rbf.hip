/*
 * rbf.hip -- radial-kernel matrix fill and the N x M evaluation sweep.
 *
 * The reference holds no RBF code (README:18-26); SURVEY.md 3.3/3.4 fixes what
 * is computed:  Phi_ij = phi(|x_i - x_j|)  and  s(y_k) = sum_j w_j phi(|y_k - x_j|)
 * with j ascending (one accumulator per target, so the summation order is the
 * CPU oracle's; only exp/log rounding and FMA contraction differ).
 *
 *   fill : HBM-write bound (8 B per entry), 16-byte stores, centres via cache.
 *   eval : fp64-VALU bound (~N x 16 ops per target); centre tiles + weights
 *          staged through LDS and broadcast to the wave, TPT targets per lane
 *          for ILP; exp / log evaluated with 256-entry LDS tables and short
 *          polynomials (<= 2 ulp, far inside the 1e-10 parity tolerance).
 */
#include "common.h"
#include <math.h>
#include <stdlib.h>

#define TBL_BITS 8
#define TBL_N (1 << TBL_BITS)

/* tables live in global memory (built once per context on first use) and are
   copied into LDS by each workgroup */
#define LOG_BITS 8
#define LOG_N (1 << LOG_BITS)
#define LOG_COPIES 8                  /* LDS replicas of the (1/c, ln c) table: one per PAIR of lanes of a ds_read_b128 pass */
#define LOG_LDS (LOG_N * LOG_COPIES * 2)   /* doubles: 32 KiB */
struct RbfTables {
  double exp2_frac[TBL_N];      /* 2^(i/256)                          */
  double log_pair[LOG_N][2];    /* {1/c_i, ln c_i}, c_i = (1 + (i+0.5)/LOG_N)/2 */
};

__device__ RbfTables g_rbf_tables;
static bool g_tables_ready[64] = {false};

static int ensure_tables(gsl_sinterp_hip_ctx *ctx)
{
  if (ctx->device < 64 && g_tables_ready[ctx->device]) return ST_SUCCESS;
  static RbfTables h;
  for (int i = 0; i < TBL_N; i++) h.exp2_frac[i] = exp2((double)i / TBL_N);
  for (int i = 0; i < LOG_N; i++) {
    double c = 0.5 * (1.0 + ((double)i + 0.5) / LOG_N);   /* bin midpoint of the frexp mantissa in [1/2, 1) */
    h.log_pair[i][0] = 1.0 / c;
    h.log_pair[i][1] = log(c);
  }
  HIP_OK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_rbf_tables), &h, sizeof h, 0, hipMemcpyHostToDevice));
  if (ctx->device < 64) g_tables_ready[ctx->device] = true;
  return ST_SUCCESS;
}

/* 2^t for t <= 0 (and moderate t > 0): t*256 = k + f, |f| <= 1/2;
   2^t = 2^(k>>8) * T[k&255] * exp(f ln2/256), degree-4 Taylor (|arg| <= 1.36e-3,
   truncation 3.8e-17) */
__device__ __forceinline__ double exp2_tbl(double t, const double *__restrict__ tbl)
{
  t = fmax(t, -1100.0);
  const double ts = t * (double)TBL_N;
  const double kf = rint(ts);
  const double f = ts - kf;                       /* exact */
  const int k = (int)kf;
  const double a = f * (0.693147180559945309417232 / TBL_N);
  double p = fma(a, 1.0 / 24.0, 1.0 / 6.0);
  p = fma(p, a, 0.5);
  p = fma(p, a, 1.0);
  p = fma(p, a, 1.0);
  return ldexp(tbl[k & (TBL_N - 1)] * p, k >> TBL_BITS);
}

/* ln(v), v >= 0 finite: v = 2^e m with m in [1/2, 1) from v_frexp_mant_f64 / v_frexp_exp_i32_f64
   (one instruction each; splitting the high word with integer ops costs five more).  Table index =
   top 8 mantissa bits; c_i = (1 + (i+0.5)/256)/2 is the midpoint of m's bin, u = m/c_i - 1,
   |u| <= 2^-9, log1p(u) to u^5 (|u|^6/6 < 1e-17);  ln v = e ln2 + ln c_i + log1p(u).
   (Until round 3: 128 entries, |u| <= 2^-8, one more term -- the larger table trades one FMA of the ~22 VALU
   instructions per pair for nothing: C2 sweep 2.87 -> 2.80 ms.)
   The lookup is data dependent per lane; a plain LDS table costs ~3x in bank conflicts (measured:
   31 % of the TPS sweep).  The table is therefore stored as 8 interleaved copies of the 16-byte
   pair {1/c_i, ln c_i}: row i is 128 bytes = 32 banks, lane l reads copy l & 7, so a 16-lane pass of the
   ds_read_b128 meets at most a two-way conflict (lanes l and l + 8) whatever the indices are -- 16 cycles per
   wave and pair against >= 80 of VALU work (16 copies of 128 entries, conflict free, were the same 32 KiB).
   v = 0 gives a finite value (m = 0 -> u = -1), which the callers multiply by r^2 = 0. */
template <int COPIES>
__device__ __forceinline__ double log_tbl(double v, const double *__restrict__ lt_lane)
{
  const int idx = (__double2hiint(v) >> (20 - LOG_BITS)) & (LOG_N - 1);
  const double2 t = *reinterpret_cast<const double2 *>(lt_lane + idx * (COPIES * 2));
  const double m = __builtin_amdgcn_frexp_mant(v);
  const int e = __builtin_amdgcn_frexp_exp(v);
  const double u = fma(m, t.x, -1.0);
  double p = fma(u, 0.2, -0.25);
  p = fma(p, u, 1.0 / 3.0);
  p = fma(p, u, -0.5);
  p = fma(p, u, 1.0);
  return fma((double)e, 0.693147180559945309417232, fma(p, u, t.y));
}

template <int KIND, int COPIES>
__device__ __forceinline__ double phi_r2(double r2, double coef, const double *__restrict__ t0,
                                         const double *__restrict__ lt_lane)
{
  if (KIND == GSL_SINTERP_RBF_GAUSSIAN) {
    return exp2_tbl(r2 * coef, t0);                /* coef = -eps^2 log2(e) */
  } else if (KIND == GSL_SINTERP_RBF_WENDLAND) {
    /* coef = eps; exactly 0 at and beyond the support radius (u <= 0); a NaN distance stays NaN */
    const double t = coef * sqrt(r2), u = 1.0 - t, u2 = u * u;
    return u <= 0.0 ? 0.0 : (u2 * u2) * fma(4.0, t, 1.0);
  } else {
    /* r^2 ln r = 0.5 r^2 ln r^2; the 0.5 is folded into the caller's weight (coef = 0.5 in fill).
       r2 = 0 (target on a centre): log_tbl returns a finite value, the product is exactly 0 */
    return (coef * r2) * log_tbl<COPIES>(r2, lt_lane);
  }
}

template <int COPIES>
__device__ __forceinline__ void load_tables(double *s_t0, double *s_lt, int kind)
{
  if (kind == GSL_SINTERP_RBF_GAUSSIAN) {
    for (int i = threadIdx.x; i < TBL_N; i += blockDim.x) s_t0[i] = g_rbf_tables.exp2_frac[i];
  } else {
    /* eight loads in flight: one at a time, the 16 trips of the replicated table are 16 global round trips (~25 us) per workgroup */
#pragma unroll 8
    for (int i = threadIdx.x; i < LOG_N * COPIES * 2; i += blockDim.x) s_lt[i] = g_rbf_tables.log_pair[i / (COPIES * 2)][i & 1];
  }
}

/* ------------------------------------------------------------------------ */
/* fill: block = 256 threads -> 16 rows x 128 cols, 2 columns (16 B) per lane */
/* SHIFT (thin-plate spline, round 4): the entry leaves as Phi_ij + s sum_a P_a[i] P_a[j] with s = cmul |Phi|_inf / n read from
   norm_bits (tps_rownorm_kernel) -- the shifted SPD matrix of solve.hip in ONE pass over HBM; until round 3 the plain fill was
   followed by a read of the whole matrix for its norm and a read-modify-write for the shift (61 + 61 us at N = 4096). */
template <int KIND, int DIM, bool SHIFT = false>
__global__ void __launch_bounds__(256)
rbf_fill_kernel(double coef, const double *__restrict__ x, size_t n, size_t xtda, double *__restrict__ phi, size_t lda, int lower_only,
                const double *__restrict__ Pk = nullptr, int pk = 0, double cmul = 0.0, const unsigned long long *__restrict__ norm_bits = nullptr)
{
  /* lower_only: tiles that lie entirely above the diagonal are not written (the Cholesky route reads the lower
     triangle only; half of the HBM writes of the fill) */
  if (lower_only && (size_t)blockIdx.x * 128 > (size_t)blockIdx.y * 16 + 15) return;
  __shared__ double s_t0[KIND == GSL_SINTERP_RBF_GAUSSIAN ? TBL_N : 1];
  __shared__ __attribute__((aligned(16))) double s_lt[KIND == GSL_SINTERP_RBF_TPS ? LOG_N * 2 : 2];   /* one copy: the fill is HBM-write bound */
  load_tables<1>(s_t0, s_lt, KIND);
  const double *lt_lane = s_lt;
  __syncthreads();
  const size_t j0 = ((size_t)blockIdx.x * 64 + (threadIdx.x & 63)) * 2;
  const size_t ibase = (size_t)blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
  if (j0 >= n) return;
  const bool two = j0 + 1 < n;
  double xa[DIM], xb[DIM];
#pragma unroll
  for (int c = 0; c < DIM; c++) { xa[c] = x[j0 * xtda + c]; xb[c] = two ? x[(j0 + 1) * xtda + c] : 0.0; }
  double pja[4] = {0, 0, 0, 0}, pjb[4] = {0, 0, 0, 0}, pir[4][4] = {{0}}, sh = 0.0;
  if (SHIFT) {
    sh = cmul * __longlong_as_double((long long)*norm_bits) / (double)n;
#pragma unroll
    for (int a = 0; a < 4; a++)
      if (a < pk) { pja[a] = Pk[(size_t)a * n + j0]; pjb[a] = two ? Pk[(size_t)a * n + j0 + 1] : 0.0; }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int a = 0; a < 4; a++) pir[r][a] = (a < pk && ibase + r < n) ? Pk[(size_t)a * n + ibase + r] : 0.0;   /* uniform over the wave; all in flight before the rows */
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const size_t i = ibase + r;
    if (i >= n) break;
    double ra = 0.0, rb = 0.0;
#pragma unroll
    for (int c = 0; c < DIM; c++) {
      const double xi = x[i * xtda + c];
      const double da = xi - xa[c], db = xi - xb[c];
      ra = fma(da, da, ra); rb = fma(db, db, rb);
    }
    double va = phi_r2<KIND, 1>(ra, coef, s_t0, lt_lane);
    double vb = phi_r2<KIND, 1>(rb, coef, s_t0, lt_lane);
    if (KIND == GSL_SINTERP_RBF_TPS) { va = ra > 0.0 ? va : 0.0; vb = rb > 0.0 ? vb : 0.0; }   /* phi(0) = 0 exactly in the matrix */
    if (SHIFT) {
      double acca = 0.0, accb = 0.0;
#pragma unroll
      for (int a = 0; a < 4; a++)
        if (a < pk) { acca = fma(pir[r][a], pja[a], acca); accb = fma(pir[r][a], pjb[a], accb); }
      va = fma(sh, acca, va); vb = fma(sh, accb, vb);
    }
    double *dst = phi + i * lda + j0;
    /* non-temporal: 1 GB of matrix is written once and read back by the factorisation long after (C3 init 32.66 -> 32.55 ms) */
    if (two && ((((uintptr_t)dst) & 15) == 0)) { typedef double v2d __attribute__((ext_vector_type(2))); v2d vv = {va, vb}; __builtin_nontemporal_store(vv, reinterpret_cast<v2d *>(dst)); }
    else { dst[0] = va; if (two) dst[1] = vb; }
  }
}

/* A target with a NaN coordinate has r^2 = NaN for every centre: the naive sum (oracle) is NaN.  The
   Gaussian sweeps compare r^2 against the cut-off (false for NaN), so they restore the NaN at the end. */
template <int DIM>
__device__ __forceinline__ bool nan_target(const double (&yy)[DIM])
{
  bool isn = false;
#pragma unroll
  for (int c = 0; c < DIM; c++) isn |= (yy[c] != yy[c]);
  return isn;
}

/* ------------------------------------------------------------------------ */
/* eval sweep: TPT targets per lane, centre tile of TJ entries in LDS          */
#define EV_THREADS 256
#define EV_TJ 512

template <int KIND, int DIM, int TPT>
__global__ void __launch_bounds__(EV_THREADS)
rbf_eval_kernel(double coef, const double *__restrict__ x, size_t n, size_t xtda, const double *__restrict__ w,
                const double *__restrict__ y, size_t m, size_t ytda, double *__restrict__ s, const int *__restrict__ perm,
                const unsigned *__restrict__ omap)
{
  __shared__ double s_t0[KIND == GSL_SINTERP_RBF_GAUSSIAN ? TBL_N : 1];
  __shared__ __attribute__((aligned(16))) double s_lt[KIND == GSL_SINTERP_RBF_TPS ? LOG_LDS : 2];
  __shared__ double s_c[EV_TJ * (DIM + 1)];       /* per centre: x[0..DIM-1], w */
  load_tables<LOG_COPIES>(s_t0, s_lt, KIND);
  const double *lt_lane = s_lt + (threadIdx.x & (LOG_COPIES - 1)) * 2;

  /* slot i of the (optionally cell-sorted) order -> target index; a lane's TPT targets are
     ADJACENT slots so they are spatial neighbours too */
  const size_t k0 = (((size_t)blockIdx.x * EV_THREADS) + threadIdx.x) * TPT;
  size_t kidx[TPT];
  double yy[TPT][DIM], acc[TPT];
#pragma unroll
  for (int t = 0; t < TPT; t++) {
    const size_t slot = k0 + (size_t)t;
    kidx[t] = slot < m ? (perm ? (size_t)perm[slot] : slot) : m;
    acc[t] = 0.0;
#pragma unroll
    for (int c = 0; c < DIM; c++) yy[t][c] = kidx[t] < m ? y[kidx[t] * ytda + c] : 0.0;
  }

  for (size_t jt = 0; jt < n; jt += EV_TJ) {
    const int cnt = (int)((n - jt) < (size_t)EV_TJ ? (n - jt) : (size_t)EV_TJ);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt; e += EV_THREADS) {
#pragma unroll
      for (int c = 0; c < DIM; c++) s_c[e * (DIM + 1) + c] = x[(jt + e) * xtda + c];
      s_c[e * (DIM + 1) + DIM] = (KIND == GSL_SINTERP_RBF_TPS ? 0.5 : 1.0) * w[jt + e];
    }
    __syncthreads();
#pragma unroll 2
    for (int e = 0; e < cnt; e++) {
      double xc[DIM];
#pragma unroll
      for (int c = 0; c < DIM; c++) xc[c] = s_c[e * (DIM + 1) + c];
      const double wj = s_c[e * (DIM + 1) + DIM];
      double r2[TPT];
#pragma unroll
      for (int t = 0; t < TPT; t++) {
        r2[t] = 0.0;
#pragma unroll
        for (int c = 0; c < DIM; c++) { const double d = yy[t][c] - xc[c]; r2[t] = fma(d, d, r2[t]); }
      }
      bool take[TPT];
      if (KIND == GSL_SINTERP_RBF_GAUSSIAN) {
        /* Every distance is computed; a target takes a term only when it is above 2^-72 of the
           kernel's maximum -- a function of the (target, centre) pair alone, so the value does not
           depend on which targets share the wave (bit-reproducible whatever the target order).
           The exponential is skipped when NO lane of the wave takes the term (wave-uniform branch).
           Dropped terms are < 2^-72 |w_j| each, i.e. a relative error <= N max|w| 2.1e-22 on O(1)
           values -- ten orders below the 1e-10 parity tolerance (tests/test_gpu_rbf.py). */
        bool need = false;
#pragma unroll
        for (int t = 0; t < TPT; t++) { take[t] = r2[t] * coef > -72.0; need |= take[t]; }
        if (__builtin_amdgcn_ballot_w64(need) == 0) continue;
      }
      if (KIND == GSL_SINTERP_RBF_WENDLAND) {
        /* outside the support the term is exactly 0: skipping it changes nothing (wave-uniform skip of the sqrt) */
        bool need = false;
#pragma unroll
        for (int t = 0; t < TPT; t++) { take[t] = r2[t] * (coef * coef) < 1.0; need |= take[t]; }
        if (__builtin_amdgcn_ballot_w64(need) == 0) continue;
      }
#pragma unroll
      for (int t = 0; t < TPT; t++) {
        const double a = fma(wj, phi_r2<KIND, LOG_COPIES>(r2[t], KIND == GSL_SINTERP_RBF_TPS ? 1.0 : coef, s_t0, lt_lane), acc[t]);
        acc[t] = (KIND != GSL_SINTERP_RBF_TPS && !take[t]) ? acc[t] : a;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TPT; t++)
    if (kidx[t] < m) s[omap ? (size_t)omap[kidx[t]] : kidx[t]] = (KIND != GSL_SINTERP_RBF_TPS && nan_target<DIM>(yy[t])) ? NAN : acc[t];
}

/* ------------------------------------------------------------------------ */
/* Gaussian sweep with tile culling.  At the shape parameters this path is used with
   (eps ~ 2 N^(1/d)) a target only sees centres within r = sqrt(72 ln 2)/eps of itself -- a few per
   cent of the cloud -- yet the plain sweep still computes every distance.  Here the centres are
   put in Morton cell order (sort.hip), packed as {x, w} and cut into tiles of CT consecutive
   (hence spatially compact) centres with a bounding box each; a workgroup -- whose 512 targets are
   neighbours too, thanks to the target sort -- first collects the tiles whose box comes within
   the cut-off of ITS targets' box (one ballot per 64 tiles, kept as bit masks: ascending order,
   no atomics) and then runs the usual inner loop over those tiles only.  The criterion is the
   one of the per-centre early-out (term < 2^-72 of the kernel maximum), applied to a lower bound
   of the distance, so only terms below that bound are dropped; the summation order is the (fixed)
   Morton order of the centres instead of their input order. */
/* Tile size CT (template parameter: 8, 16 or 32 centres; 128 until round 2).  A target tests every centre of the tiles its
   workgroup keeps; smaller tiles hug the cut-off disc more closely.  Sweep phase, ms -- round 2: C4 (2-D, N = 8192, M = 1e7)
   2.41 / 1.97 / 1.73 / 1.58 and C3 (3-D, N = 16384, M = 1e6) 1.96 / 1.75 / 1.70 / 1.74 for CT = 128 / 64 / 32 / 16; round 3 (targets
   physically reordered by Morton cell, a workgroup = a 2 x 2 block of target cells): C4 1.41 / 1.25 / 1.18 and C3 1.29 / 1.30 / 1.34
   for CT = 32 / 16 / 8; with the kept tiles batched CULL_STAGE centres per LDS stage (one pair of barriers per 64 centres instead
   of per tile): C4 1.25 / 1.19 / 1.18 for CT = 16 / 8 / 4, C3 1.255 / 1.253 / 1.29 for CT = 32 / 16 / 8.  So: 8 in two dimensions,
   32 otherwise -- and the next larger size while the tile count would exceed
   CULL_MAX_TILES (the kept-tile bit mask in LDS).  The tile size changes which centres are TESTED, never which terms a target
   takes nor their order: results are bit-identical for every CT. */
#define CULL_MAX_TILES 8192
#ifndef CULL_STAGE
#define CULL_STAGE 64       /* centres per LDS stage of the culled sweep (a multiple of every tile size) */
#endif
static inline int cull_tile_size(int dim, size_t n)
{
  int ct = dim == 2 ? 8 : 32;
  while (ct < 32 && (n + ct - 1) / ct > CULL_MAX_TILES) ct *= 2;
  return ct;
}

#define CT_THREADS 64
template <int DIM, int CT>
__global__ void __launch_bounds__(CT_THREADS)
centre_pack_kernel(const double *__restrict__ x, size_t n, size_t xtda, const double *__restrict__ w,
                   const int *__restrict__ perm, double *__restrict__ xs, double *__restrict__ tbox)
{
  constexpr int NW = (CT + 63) / 64;
  __shared__ double s_lo[DIM][NW], s_hi[DIM][NW];
  const size_t i = (size_t)blockIdx.x * CT + threadIdx.x;
  const bool ok = threadIdx.x < CT && i < n;
  double v[DIM];
  if (ok) {
    const size_t j = (size_t)perm[i];
#pragma unroll
    for (int c = 0; c < DIM; c++) { v[c] = x[j * xtda + c]; xs[i * (DIM + 1) + c] = v[c]; }
    xs[i * (DIM + 1) + DIM] = w[j];
  }
#pragma unroll
  for (int c = 0; c < DIM; c++) {
    double lo = ok ? v[c] : INFINITY, hi = ok ? v[c] : -INFINITY;
    for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_xor(lo, off)); hi = fmax(hi, __shfl_xor(hi, off)); }
    if ((threadIdx.x & 63) == 0) { s_lo[c][threadIdx.x >> 6] = lo; s_hi[c][threadIdx.x >> 6] = hi; }
  }
  __syncthreads();
  if (threadIdx.x < DIM) {
    const int c = threadIdx.x;
    double l = s_lo[c][0], h = s_hi[c][0];
    for (int w = 1; w < NW; w++) { l = fmin(l, s_lo[c][w]); h = fmax(h, s_hi[c][w]); }
    tbox[(size_t)blockIdx.x * (2 * DIM) + 2 * c] = l;
    tbox[(size_t)blockIdx.x * (2 * DIM) + 2 * c + 1] = h;
  }
}

#ifdef SINTERP_DIAG_PROF
/* developer build only (make prof): pair counters of the culled sweep -- [0] pair-lanes the kernel evaluated (every lane
   of a wave pays for a centre that ANY of its targets takes), [1] pairs inside the cut-off (the useful ones), [2] centres
   staged in LDS x targets of the workgroup (what survives the tile culling).  tools/gauss_pairs.py reads them. */
__device__ unsigned long long g_cull_stats[4];
extern "C" int gsl_sinterp_hip_debug_cull_stats(unsigned long long *out, int reset)
{
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cull_stats), sizeof(unsigned long long) * 4);
  if (e == hipSuccess && reset) { unsigned long long z[4] = {0, 0, 0, 0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_cull_stats), z, sizeof z); }
  return (int)e;
}
#define CULL_STAT(i, v) st_##i += (v)
#else
#define CULL_STAT(i, v) do { } while (0)
#endif

/* KIND = Gaussian: cut-off 2^-72 of the kernel maximum (coef = -eps^2 log2 e); KIND = Wendland: the support
   radius itself (coef = eps), so culling drops terms that are exactly 0 */
#ifndef CULL_THREADS
#define CULL_THREADS 128   /* 256 / 128 / 64 threads: C3 sweep 1.34 / 1.26 / 1.27 ms, C4 1.71 ms throughout */
#endif
template <int KIND, int DIM, int TPT, int CT>
__global__ void __launch_bounds__(CULL_THREADS)
rbf_eval_gauss_cull_kernel(double coef, const double *__restrict__ xs, size_t n, const double *__restrict__ tbox, unsigned ntiles,
                           const double *__restrict__ y, size_t m, size_t ytda, double *__restrict__ s, const int *__restrict__ perm,
                const unsigned *__restrict__ omap)
{
  __shared__ double s_t0[TBL_N];
  __shared__ __attribute__((aligned(16))) double s_c[CULL_STAGE * (DIM + 1)];
  constexpr int NWV = CULL_THREADS / 64;
  __shared__ double s_blo[DIM][NWV], s_bhi[DIM][NWV];
  __shared__ unsigned long long s_mask[CULL_MAX_TILES / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < TBL_N; i += CULL_THREADS) s_t0[i] = g_rbf_tables.exp2_frac[i];

  const size_t k0 = (((size_t)blockIdx.x * CULL_THREADS) + tid) * TPT;
  size_t kidx[TPT];
  double yy[TPT][DIM], acc[TPT];
#pragma unroll
  for (int t = 0; t < TPT; t++) {
    const size_t slot = k0 + (size_t)t;
    kidx[t] = slot < m ? (perm ? (size_t)perm[slot] : slot) : m;
    acc[t] = 0.0;
#pragma unroll
    for (int c = 0; c < DIM; c++) yy[t][c] = kidx[t] < m ? y[kidx[t] * ytda + c] : 0.0;
  }
  /* bounding box of this workgroup's targets */
#pragma unroll
  for (int c = 0; c < DIM; c++) {
    double lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int t = 0; t < TPT; t++) if (kidx[t] < m) { lo = fmin(lo, yy[t][c]); hi = fmax(hi, yy[t][c]); }
    for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_xor(lo, off)); hi = fmax(hi, __shfl_xor(hi, off)); }
    if (lane == 0) { s_blo[c][wave] = lo; s_bhi[c][wave] = hi; }
  }
  __syncthreads();
  double blo[DIM], bhi[DIM];
#pragma unroll
  for (int c = 0; c < DIM; c++) {
    blo[c] = s_blo[c][0]; bhi[c] = s_bhi[c][0];
#pragma unroll
    for (int w = 1; w < NWV; w++) { blo[c] = fmin(blo[c], s_blo[c][w]); bhi[c] = fmax(bhi[c], s_bhi[c][w]); }
  }
  /* tiles within the cut-off of the box: one bit per tile */
  const unsigned nmask = (ntiles + 63) / 64;
  for (unsigned base = 0; base < ntiles; base += CULL_THREADS) {
    const unsigned t = base + tid;
    bool keep = false;
    if (t < ntiles) {
      double d2 = 0.0;
#pragma unroll
      for (int c = 0; c < DIM; c++) {
        const double tl = tbox[(size_t)t * (2 * DIM) + 2 * c], th = tbox[(size_t)t * (2 * DIM) + 2 * c + 1];
        const double g = fmax(0.0, fmax(tl - bhi[c], blo[c] - th));
        d2 = fma(g, g, d2);
      }
      keep = KIND == GSL_SINTERP_RBF_GAUSSIAN ? d2 * coef > -72.0 : d2 * (coef * coef) < 1.0;
    }
    const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
    if (lane == 0 && (base / 64 + wave) < nmask) s_mask[base / 64 + wave] = b;
  }
  __syncthreads();

#ifdef SINTERP_DIAG_PROF
  unsigned long long st_0 = 0, st_1 = 0, st_2 = 0;
#endif
  /* the kept tiles, ascending, CULL_STAGE centres per LDS stage (several small tiles share one pair of barriers) */
  unsigned mi = 0;
  unsigned long long mask = nmask ? s_mask[0] : 0ULL;
  for (;;) {
    int cnt = 0;
    __syncthreads();                                       /* the previous stage has been consumed */
    while (cnt + CT <= CULL_STAGE) {
      while (!mask && mi + 1 < nmask) mask = s_mask[++mi];
      if (!mask) break;
      const unsigned t = mi * 64 + (unsigned)__builtin_ctzll(mask);
      mask &= mask - 1;
      const size_t c0 = (size_t)t * CT;
      const int tc = (int)((n - c0) < (size_t)CT ? (n - c0) : (size_t)CT);
      for (int e = tid; e < tc * (DIM + 1); e += CULL_THREADS) s_c[cnt * (DIM + 1) + e] = xs[c0 * (DIM + 1) + e];
      cnt += tc;
    }
    if (cnt == 0) break;
    __syncthreads();
    {
#pragma unroll 2
      for (int e = 0; e < cnt; e++) {
        double xc[DIM];
#pragma unroll
        for (int c = 0; c < DIM; c++) xc[c] = s_c[e * (DIM + 1) + c];
        const double wj = s_c[e * (DIM + 1) + DIM];
        double r2[TPT];
        bool take[TPT], need = false;
#pragma unroll
        for (int tt = 0; tt < TPT; tt++) {
          r2[tt] = 0.0;
#pragma unroll
          for (int c = 0; c < DIM; c++) { const double d = yy[tt][c] - xc[c]; r2[tt] = fma(d, d, r2[tt]); }
          take[tt] = KIND == GSL_SINTERP_RBF_GAUSSIAN ? r2[tt] * coef > -72.0 : r2[tt] * (coef * coef) < 1.0;
          need |= take[tt];
        }
        CULL_STAT(2, TPT);
        if (__builtin_amdgcn_ballot_w64(need) == 0) continue;
        CULL_STAT(0, TPT);
#pragma unroll
        for (int tt = 0; tt < TPT; tt++) CULL_STAT(1, take[tt] ? 1 : 0);
        /* per-target criterion (see rbf_eval_kernel): a culled tile holds only centres every target of
           the workgroup would reject, so the value is the sum over the centres with term > 2^-72, in
           Morton order -- independent of the workgroup / wave the target landed in */
#pragma unroll
        for (int tt = 0; tt < TPT; tt++) {
          const double a = fma(wj, phi_r2<KIND, 1>(r2[tt], coef, s_t0, (const double *)NULL), acc[tt]);
          acc[tt] = take[tt] ? a : acc[tt];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TPT; t++)
    if (kidx[t] < m) s[omap ? (size_t)omap[kidx[t]] : kidx[t]] = nan_target<DIM>(yy[t]) ? NAN : acc[t];
#ifdef SINTERP_DIAG_PROF
  for (int off = 32; off > 0; off >>= 1) { st_0 += __shfl_xor(st_0, off); st_1 += __shfl_xor(st_1, off); st_2 += __shfl_xor(st_2, off); }
  if (lane == 0) { atomicAdd(&g_cull_stats[0], st_0); atomicAdd(&g_cull_stats[1], st_1); atomicAdd(&g_cull_stats[2], st_2); }
#endif
}

template <int KIND, int TPT>
static int launch_eval_cull(gsl_sinterp_hip_ctx *ctx, double coef, const double *d_x, size_t n, int dim, size_t xtda, const double *d_w,
                            const double *d_y, size_t m, size_t ytda, double *d_s, const int *d_perm, const unsigned *d_omap,
                            unsigned long long model_id)
{
  /* the packed centres depend on the model only: reuse them when the caller vouches for the model (model_id != 0) */
  const bool cached = model_id != 0 && ctx->cent_key.id == model_id && ctx->cent_key.x == d_x && ctx->cent_key.w == d_w &&
                      ctx->cent_key.n == n && ctx->cent_key.xtda == xtda && ctx->cent_key.dim == dim && ctx->cent_key.kind == KIND;
  int *d_cperm = NULL;
  int st = ST_SUCCESS;
  if (!cached) {
    ctx->cent_key.id = 0;
    st = sinterp_sort_centres(ctx, d_x, n, xtda, dim, 8, &d_cperm);
    if (st) return st;
  }
  const int ct = cull_tile_size(dim, n);
  const unsigned ntiles = (unsigned)((n + ct - 1) / ct);
  void *buf = NULL;
  st = sinterp_centbuf(ctx, (n * (size_t)(dim + 1) + (size_t)ntiles * 2 * dim) * sizeof(double), &buf);
  if (st) return st;
  double *xs = (double *)buf, *tbox = xs + n * (size_t)(dim + 1);
  const size_t per_block = (size_t)CULL_THREADS * TPT;
  dim3 grid((unsigned)((m + per_block - 1) / per_block));
#define CULL_LAUNCH(D, C)                                                                                                              \
  do {                                                                                                                                 \
    if (!cached) hipLaunchKernelGGL((centre_pack_kernel<D, C>), dim3(ntiles), dim3(CT_THREADS), 0, ctx->stream, d_x, n, xtda, d_w,    \
                                    (const int *)d_cperm, xs, tbox);                                                                   \
    hipLaunchKernelGGL((rbf_eval_gauss_cull_kernel<KIND, D, TPT, C>), grid, dim3(CULL_THREADS), 0, ctx->stream, coef, (const double *)xs, \
                       n, (const double *)tbox, ntiles, d_y, m, ytda, d_s, d_perm, d_omap);                                            \
  } while (0)
  switch (dim) {
    case 1: CULL_LAUNCH(1, 32); break;
    case 2:
      if (ct == 8) CULL_LAUNCH(2, 8);
      else if (ct == 16) CULL_LAUNCH(2, 16);
      else CULL_LAUNCH(2, 32);
      break;
    default: CULL_LAUNCH(3, 32); break;
  }
#undef CULL_LAUNCH
  LAUNCH_CHECK(ctx);
  if (model_id != 0 && !cached) {
    ctx->cent_key.id = model_id; ctx->cent_key.x = d_x; ctx->cent_key.w = d_w; ctx->cent_key.n = n; ctx->cent_key.xtda = xtda;
    ctx->cent_key.dim = dim; ctx->cent_key.kind = KIND;
  }
  return ST_SUCCESS;
}

/* ------------------------------------------------------------------------ */
static double kernel_coef(int kind, double eps)
{
  if (kind == GSL_SINTERP_RBF_WENDLAND) return eps;
  return kind == GSL_SINTERP_RBF_GAUSSIAN ? -(eps * eps) * 1.44269504088896340735992 : 0.5;
}

static bool known_kind(int kind)
{
  return kind == GSL_SINTERP_RBF_GAUSSIAN || kind == GSL_SINTERP_RBF_TPS || kind == GSL_SINTERP_RBF_WENDLAND;
}

template <int KIND>
static int launch_fill(gsl_sinterp_hip_ctx *ctx, double coef, const double *d_x, size_t n, int dim, size_t xtda,
                       double *d_phi, size_t lda, int lower_only)
{
  dim3 grid((unsigned)((n + 127) / 128), (unsigned)((n + 15) / 16));
  switch (dim) {
    case 1: hipLaunchKernelGGL((rbf_fill_kernel<KIND, 1>), grid, dim3(256), 0, ctx->stream, coef, d_x, n, xtda, d_phi, lda, lower_only); break;
    case 2: hipLaunchKernelGGL((rbf_fill_kernel<KIND, 2>), grid, dim3(256), 0, ctx->stream, coef, d_x, n, xtda, d_phi, lda, lower_only); break;
    default: hipLaunchKernelGGL((rbf_fill_kernel<KIND, 3>), grid, dim3(256), 0, ctx->stream, coef, d_x, n, xtda, d_phi, lda, lower_only); break;
  }
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

#define TN_R 2   /* rows per wave: 8 per workgroup, n / 8 workgroups (16 rows per workgroup left one wave per SIMD: 76 us at N = 4096) */
/* |Phi|_inf = max_i sum_j |phi(|x_i - x_j|)| from the coordinates (nothing of the matrix is read): TN_R rows per wave, a
   wave's lanes stride over the columns, fixed summation order (lane partial sums, then the butterfly) -> reproducible */
template <int DIM>
__global__ void __launch_bounds__(256)
tps_rownorm_kernel(double coef, const double *__restrict__ x, size_t n, size_t xtda, unsigned long long *__restrict__ out)
{
  __shared__ double s_t0[1];
  __shared__ __attribute__((aligned(16))) double s_lt[LOG_LDS];        /* the sweep's replicated table: this kernel is VALU / LDS bound */
  load_tables<LOG_COPIES>(s_t0, s_lt, GSL_SINTERP_RBF_TPS);
  const double *lt_lane = s_lt + (threadIdx.x & (LOG_COPIES - 1)) * 2;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t ibase = (size_t)blockIdx.x * (4 * TN_R) + (threadIdx.x >> 6) * TN_R;
  double xi[TN_R][DIM], acc[TN_R];
#pragma unroll
  for (int r = 0; r < TN_R; r++) {
    acc[r] = 0.0;
#pragma unroll
    for (int c = 0; c < DIM; c++) xi[r][c] = ibase + r < n ? x[(ibase + r) * xtda + c] : 0.0;
  }
#pragma unroll 8
  for (size_t j = lane; j < n; j += 64) {          /* eight columns' coordinates in flight (one at a time: a load latency per step) */
    double xj[DIM];
#pragma unroll
    for (int c = 0; c < DIM; c++) xj[c] = x[j * xtda + c];
#pragma unroll
    for (int r = 0; r < TN_R; r++) {
      double r2 = 0.0;
#pragma unroll
      for (int c = 0; c < DIM; c++) { const double d = xi[r][c] - xj[c]; r2 = fma(d, d, r2); }
      const double v = phi_r2<GSL_SINTERP_RBF_TPS, LOG_COPIES>(r2, coef, s_t0, lt_lane);
      acc[r] += r2 > 0.0 ? fabs(v) : 0.0;
    }
  }
#pragma unroll
  for (int r = 0; r < TN_R; r++) {
    double a = acc[r];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0 && ibase + r < n) atomicMax(out, (unsigned long long)__double_as_longlong(a));   /* a >= 0: bit order == value order */
  }
}

/* thin-plate spline matrix with the shift of the SPD route applied in flight (both triangles): d_norm receives the bit
   pattern of |Phi|_inf, Pk = k standardised polynomial columns of length n */
int sinterp_tps_fill_shifted(gsl_sinterp_hip_ctx *ctx, const double *d_x, size_t n, int dim, size_t xtda, double *d_phi, size_t lda,
                             const double *d_Pk, int k, double cmul, unsigned long long *d_norm)
{
  REQUIRE(ctx, dim >= 1 && dim <= 3 && xtda >= (size_t)dim && lda >= n && (n + 15) / 16 <= 65535, ST_EINVAL);
  int st = ensure_tables(ctx);
  if (st) return st;
  const double coef = kernel_coef(GSL_SINTERP_RBF_TPS, 0.0);
  HIP_OK(ctx, hipMemsetAsync(d_norm, 0, sizeof(unsigned long long), ctx->stream));
  const dim3 ngrid((unsigned)((n + 4 * TN_R - 1) / (4 * TN_R))), grid((unsigned)((n + 127) / 128), (unsigned)((n + 15) / 16));
  switch (dim) {
    case 1:
      hipLaunchKernelGGL(tps_rownorm_kernel<1>, ngrid, dim3(256), 0, ctx->stream, coef, d_x, n, xtda, d_norm);
      hipLaunchKernelGGL((rbf_fill_kernel<GSL_SINTERP_RBF_TPS, 1, true>), grid, dim3(256), 0, ctx->stream, coef, d_x, n, xtda, d_phi, lda, 0, d_Pk, k, cmul, (const unsigned long long *)d_norm);
      break;
    case 2:
      hipLaunchKernelGGL(tps_rownorm_kernel<2>, ngrid, dim3(256), 0, ctx->stream, coef, d_x, n, xtda, d_norm);
      hipLaunchKernelGGL((rbf_fill_kernel<GSL_SINTERP_RBF_TPS, 2, true>), grid, dim3(256), 0, ctx->stream, coef, d_x, n, xtda, d_phi, lda, 0, d_Pk, k, cmul, (const unsigned long long *)d_norm);
      break;
    default:
      hipLaunchKernelGGL(tps_rownorm_kernel<3>, ngrid, dim3(256), 0, ctx->stream, coef, d_x, n, xtda, d_norm);
      hipLaunchKernelGGL((rbf_fill_kernel<GSL_SINTERP_RBF_TPS, 3, true>), grid, dim3(256), 0, ctx->stream, coef, d_x, n, xtda, d_phi, lda, 0, d_Pk, k, cmul, (const unsigned long long *)d_norm);
      break;
  }
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_rbf_fill(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n,
                                        int dim, size_t xtda, double *d_phi, size_t lda)
{
  return sinterp_rbf_fill_ex(ctx, kind, eps, d_x, n, dim, xtda, d_phi, lda, 0);
}

int sinterp_rbf_fill_ex(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim, size_t xtda,
                        double *d_phi, size_t lda, int lower_only)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  REQUIRE(ctx, dim >= 1 && dim <= 3 && xtda >= (size_t)dim && lda >= n, ST_EINVAL);
  REQUIRE(ctx, known_kind(kind), ST_EINVAL);
  REQUIRE(ctx, n == 0 || (d_x && d_phi), ST_EFAULT);
  REQUIRE(ctx, (n + 15) / 16 <= 65535, ST_EINVAL);
  if (n == 0) return ST_SUCCESS;
  int st = ensure_tables(ctx);
  if (st) return st;
  const double coef = kernel_coef(kind, eps);
  if (kind == GSL_SINTERP_RBF_WENDLAND) return launch_fill<GSL_SINTERP_RBF_WENDLAND>(ctx, coef, d_x, n, dim, xtda, d_phi, lda, lower_only);
  return kind == GSL_SINTERP_RBF_GAUSSIAN ? launch_fill<GSL_SINTERP_RBF_GAUSSIAN>(ctx, coef, d_x, n, dim, xtda, d_phi, lda, lower_only)
                                          : launch_fill<GSL_SINTERP_RBF_TPS>(ctx, coef, d_x, n, dim, xtda, d_phi, lda, lower_only);
}

template <int KIND, int TPT>
static int launch_eval(gsl_sinterp_hip_ctx *ctx, double coef, const double *d_x, size_t n, int dim, size_t xtda,
                       const double *d_w, const double *d_y, size_t m, size_t ytda, double *d_s, const int *d_perm, const unsigned *d_omap)
{
  const size_t per_block = (size_t)EV_THREADS * TPT;
  dim3 grid((unsigned)((m + per_block - 1) / per_block));
  switch (dim) {
    case 1: hipLaunchKernelGGL((rbf_eval_kernel<KIND, 1, TPT>), grid, dim3(EV_THREADS), 0, ctx->stream, coef, d_x, n, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap); break;
    case 2: hipLaunchKernelGGL((rbf_eval_kernel<KIND, 2, TPT>), grid, dim3(EV_THREADS), 0, ctx->stream, coef, d_x, n, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap); break;
    default: hipLaunchKernelGGL((rbf_eval_kernel<KIND, 3, TPT>), grid, dim3(EV_THREADS), 0, ctx->stream, coef, d_x, n, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap); break;
  }
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_rbf_eval(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n,
                                        int dim, size_t xtda, const double *d_w, const double *d_y, size_t m,
                                        size_t ytda, double *d_s)
{
  return gsl_sinterp_hip_rbf_eval_model(ctx, kind, eps, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, 0ULL);
}

static int rbf_eval_dispatch(gsl_sinterp_hip_ctx *ctx, int kind, double coef, const double *d_x, size_t n, int dim, size_t xtda,
                             const double *d_w, const double *d_y, size_t m, size_t ytda, double *d_s, const int *d_perm,
                             const unsigned *d_omap, unsigned long long model_id);

extern "C" int gsl_sinterp_hip_rbf_eval_model(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n,
                                              int dim, size_t xtda, const double *d_w, const double *d_y, size_t m,
                                              size_t ytda, double *d_s, unsigned long long model_id)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  REQUIRE(ctx, dim >= 1 && dim <= 3 && xtda >= (size_t)dim && ytda >= (size_t)dim, ST_EINVAL);
  REQUIRE(ctx, known_kind(kind), ST_EINVAL);
  REQUIRE(ctx, m == 0 || (d_y && d_s && (n == 0 || (d_x && d_w))), ST_EFAULT);
  if (m == 0) return ST_SUCCESS;
  int st = ensure_tables(ctx);
  if (st) return st;
  const double coef = kernel_coef(kind, eps);
  /* Gaussian / Wendland: group the targets spatially so that whole waves skip negligible (Wendland: zero) terms
     together (TPS has no decay: nothing to skip, no sort) */
  const bool local = kind != GSL_SINTERP_RBF_TPS;
  int *d_perm = NULL;
  const unsigned *d_omap = NULL;
  if (local && m >= 4096 && !(getenv("GSL_SINTERP_NO_SORT") && getenv("GSL_SINTERP_NO_SORT")[0] == '1')) {
    /* large batches: the targets are physically put in cell order (sort.hip, two-level reorder), swept contiguously, each
       result stored through the order's map, and the values gathered back -- one random pass instead of the three of
       the permutation route (histogram atomics, perm scatter, gather + scatter inside the sweep) */
    if (sinterp_sort_reorder_is_two_level(m)) {
      sinterp_sorted srt;
      st = sinterp_sort_reorder(ctx, d_y, m, ytda, dim, 64, &srt, m, -1, (const unsigned long long *)NULL);
      if (st) return st;
      if (srt.two_level) {
        st = rbf_eval_dispatch(ctx, kind, coef, d_x, n, dim, xtda, d_w, (const double *)srt.ys, m, (size_t)dim, srt.res1, (const int *)NULL,
                               (const unsigned *)srt.inv, model_id);
        if (st) return st;
        return sinterp_unsort(ctx, &srt, m, d_s, (int *)NULL);
      }
    }
    st = sinterp_sort_targets(ctx, d_y, m, ytda, dim, 64, &d_perm);
    if (st) return st;
  }
  return rbf_eval_dispatch(ctx, kind, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap, model_id);
}

static int rbf_eval_dispatch(gsl_sinterp_hip_ctx *ctx, int kind, double coef, const double *d_x, size_t n, int dim, size_t xtda,
                             const double *d_w, const double *d_y, size_t m, size_t ytda, double *d_s, const int *d_perm,
                             const unsigned *d_omap, unsigned long long model_id)
{
  const bool local = kind != GSL_SINTERP_RBF_TPS;
  /* few targets: 1 per lane keeps more CUs busy; many: 2 per lane for ILP */
  const bool small = m < (size_t)EV_THREADS * 2 * 512;
  static const bool no_cull = getenv("GSL_SINTERP_NO_CULL") && getenv("GSL_SINTERP_NO_CULL")[0] == '1';
  /* The culled kernel sums in the Morton order of the centres, the plain one in input order.  Which of the two
     runs must not depend on the batch (a target's value is a function of the model and the target alone, so a
     batch split into shards -- or a single-point call -- returns the bits of the one-batch result): it is chosen
     by N only; small batches simply run the culled kernel without the target sort. */
  if (local && !no_cull && n >= 1024 && (n + 31) / 32 <= CULL_MAX_TILES) {
    /* 3-D: one target per lane also for large batches -- a workgroup's 256 targets span half the box of 512, and
       in three dimensions that removes more tested-and-rejected centres than the second accumulator chain gains
       (C3 sweep 1.70 -> 1.34 ms; 2-D C4: 1.74 vs 1.77 ms, unchanged) */
    const bool small = m < (size_t)CULL_THREADS * 2 * 512 || dim == 3;
    if (kind == GSL_SINTERP_RBF_WENDLAND)
      return small ? launch_eval_cull<GSL_SINTERP_RBF_WENDLAND, 1>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap, model_id)
                   : launch_eval_cull<GSL_SINTERP_RBF_WENDLAND, 2>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap, model_id);
    return small ? launch_eval_cull<GSL_SINTERP_RBF_GAUSSIAN, 1>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap, model_id)
                 : launch_eval_cull<GSL_SINTERP_RBF_GAUSSIAN, 2>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap, model_id);
  }
  if (kind == GSL_SINTERP_RBF_WENDLAND)
    return small ? launch_eval<GSL_SINTERP_RBF_WENDLAND, 1>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap)
                 : launch_eval<GSL_SINTERP_RBF_WENDLAND, 2>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap);
  if (kind == GSL_SINTERP_RBF_GAUSSIAN)
    return small ? launch_eval<GSL_SINTERP_RBF_GAUSSIAN, 1>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap)
                 : launch_eval<GSL_SINTERP_RBF_GAUSSIAN, 2>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap);
  return small ? launch_eval<GSL_SINTERP_RBF_TPS, 1>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap)
               : launch_eval<GSL_SINTERP_RBF_TPS, 2>(ctx, coef, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, d_perm, d_omap);
}
