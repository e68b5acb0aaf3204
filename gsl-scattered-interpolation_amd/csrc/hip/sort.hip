/*
 * sort.hip -- spatial binning of evaluation targets (counting sort by grid cell).
 *
 * Neither the reference (per-point API, interpolation/linear_simplex.c:331,678) nor the
 * oracle orders its targets; results per target are independent of the order.  On the GPU
 * the order decides whether the 64 lanes of a wave walk the same DAG nodes (barycentric) /
 * see the same negligible Gaussian terms (RBF sweep), so both sweeps process targets
 * through a permutation that groups them by cell of a uniform grid over their bounding box.
 * Outputs are written back at the original positions: the caller-visible layout is unchanged.
 *
 * Four small kernels: bounding box (atomic min/max on order-preserving integer keys),
 * histogram (one atomic per point, its return value is the point's slot in the cell),
 * exclusive scan (one workgroup), atomic-free scatter.
 */
#include "common.h"
#include <math.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned long long dkey(double v)   /* monotone double -> uint64 */
{
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ __forceinline__ double dunkey(unsigned long long k)
{
  unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
  return __longlong_as_double((long long)u);
}

__global__ void bbox_init_kernel(unsigned long long *__restrict__ box)
{
  if (threadIdx.x < 6) box[threadIdx.x] = (threadIdx.x & 1) ? 0ULL : ~0ULL;
}

/* box[2c] = min key, box[2c+1] = max key */
__global__ void __launch_bounds__(256)
bbox_kernel(const double *__restrict__ y, size_t m, size_t ytda, int dim, unsigned long long *__restrict__ box)
{
  unsigned long long lo[3] = {~0ULL, ~0ULL, ~0ULL}, hi[3] = {0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride)
    for (int c = 0; c < dim; c++) {
      const double v = y[k * ytda + c];
      if (v == v) { const unsigned long long key = dkey(v); lo[c] = key < lo[c] ? key : lo[c]; hi[c] = key > hi[c] ? key : hi[c]; }
    }
  /* wave reduce -> workgroup reduce in LDS -> one atomic pair per workgroup and coordinate */
  __shared__ unsigned long long s_lo[3][4], s_hi[3][4];
  for (int c = 0; c < dim; c++) {
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long ol = __shfl_xor(lo[c], off), oh = __shfl_xor(hi[c], off);
      lo[c] = ol < lo[c] ? ol : lo[c];
      hi[c] = oh > hi[c] ? oh : hi[c];
    }
    if ((threadIdx.x & 63) == 0) { s_lo[c][threadIdx.x >> 6] = lo[c]; s_hi[c][threadIdx.x >> 6] = hi[c]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < dim) {
    const int c = threadIdx.x;
    unsigned long long l = s_lo[c][0], h = s_hi[c][0];
    for (int w = 1; w < 4; w++) { l = s_lo[c][w] < l ? s_lo[c][w] : l; h = s_hi[c][w] > h ? s_hi[c][w] : h; }
    atomicMin(&box[2 * c], l);
    atomicMax(&box[2 * c + 1], h);
  }
}

/* g > 0: row-major cell index on a g^dim grid.  g < 0: Morton (bit-interleaved) index on a grid of
   2^bits = -g cells per axis, so that runs of consecutive cells are spatially compact. */
__device__ __forceinline__ unsigned cell_of(const double *__restrict__ y, size_t k, size_t ytda, int dim, int g,
                                            const unsigned long long *__restrict__ box)
{
  const int gg = g < 0 ? -g : g;
  unsigned cell = 0, ic[3] = {0, 0, 0};
  for (int c = dim - 1; c >= 0; c--) {
    const double lo = dunkey(box[2 * c]), hi = dunkey(box[2 * c + 1]);
    const double v = y[k * ytda + c];
    double f = (hi > lo) ? (v - lo) / (hi - lo) : 0.0;
    int i = (f == f) ? (int)(f * gg) : 0;          /* NaN coordinates go to cell 0 */
    i = i < 0 ? 0 : (i >= gg ? gg - 1 : i);
    ic[c] = (unsigned)i;
    cell = cell * (unsigned)gg + (unsigned)i;
  }
  if (g < 0) {
    cell = 0;
    for (int b = 0; (1 << b) < gg; b++)
      for (int c = 0; c < dim; c++) cell |= ((ic[c] >> b) & 1u) << (b * dim + c);
  }
  return cell;
}

/* one atomic per point: the returned old count is the point's slot inside its cell, so the scatter
   pass needs no atomics */
__global__ void __launch_bounds__(256)
cell_hist_kernel(const double *__restrict__ y, size_t m, size_t ytda, int dim, int g, const unsigned long long *__restrict__ box,
                 unsigned *__restrict__ cellid, unsigned *__restrict__ slot, unsigned *__restrict__ count)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
    const unsigned c = cell_of(y, k, ytda, dim, g, box);
    cellid[k] = c;
    slot[k] = atomicAdd(&count[c], 1u);
  }
}

/* in-place exclusive scan of count[0..ncell) by one workgroup, 32 consecutive entries per thread and
   pass; count[ncell] receives the total */
#define SCAN_PER 32
__global__ void __launch_bounds__(1024)
cell_scan_kernel(unsigned *__restrict__ count, unsigned ncell)
{
  __shared__ unsigned s_wave[16];
  __shared__ unsigned s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (unsigned base = 0; base < ncell; base += 1024 * SCAN_PER) {
    const unsigned i0 = base + threadIdx.x * SCAN_PER;
    unsigned v[SCAN_PER], sum = 0;
#pragma unroll
    for (int q = 0; q < SCAN_PER; q++) { v[q] = (i0 + q < ncell) ? count[i0 + q] : 0u; sum += v[q]; }
    unsigned incl = sum;
    for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wave; w++) woff += s_wave[w];
    const unsigned carry = s_carry;
    unsigned run = carry + woff + incl - sum;
#pragma unroll
    for (int q = 0; q < SCAN_PER; q++) { if (i0 + q < ncell) count[i0 + q] = run; run += v[q]; }
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) count[ncell] = s_carry;
}

/* large cell counts: wave-level scans of 1024-entry runs (coalesced, shuffle based), a single-workgroup
   scan of the run totals, and an add pass */
__global__ void __launch_bounds__(256)
cell_scan_runs_kernel(unsigned *__restrict__ count, unsigned ncell, unsigned *__restrict__ runsum)
{
  const int lane = threadIdx.x & 63;
  const unsigned run = blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned base = run * 1024u;
  unsigned carry = 0;
#pragma unroll 4
  for (int it = 0; it < 16; it++) {
    const unsigned i = base + it * 64 + lane;
    const unsigned v = i < ncell ? count[i] : 0u;
    unsigned incl = v;
    for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (i < ncell) count[i] = carry + incl - v;
    carry += __shfl(incl, 63);
  }
  if (lane == 0) runsum[run] = carry;
}

__global__ void __launch_bounds__(256)
cell_scan_add_kernel(unsigned *__restrict__ count, unsigned ncell, const unsigned *__restrict__ runsum, unsigned nruns)
{
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i < ncell) count[i] += runsum[i >> 10];
  if (i == 0) count[ncell] = runsum[nruns];
}

__global__ void __launch_bounds__(256)
cell_scatter_kernel(const unsigned *__restrict__ cellid, const unsigned *__restrict__ slot, size_t m,
                    const unsigned *__restrict__ offset, int *__restrict__ perm)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride)
    perm[offset[cellid[k]] + slot[k]] = (int)k;
}


/* exclusive scan of count[0..ncell), total in count[ncell]; runsum needs ncell/1024 + 2 entries */
static void launch_cell_scan(gsl_sinterp_hip_ctx *ctx, unsigned *count, size_t ncell, unsigned *runsum);
void sinterp_scan_u32(gsl_sinterp_hip_ctx *ctx, unsigned *count, size_t n, unsigned *runsum) { launch_cell_scan(ctx, count, n, runsum); }
static void launch_cell_scan(gsl_sinterp_hip_ctx *ctx, unsigned *count, size_t ncell, unsigned *runsum)
{
  if (ncell <= 32768) {
    hipLaunchKernelGGL(cell_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, count, (unsigned)ncell);
    return;
  }
  const unsigned nruns = (unsigned)((ncell + 1023) / 1024);
  hipLaunchKernelGGL(cell_scan_runs_kernel, dim3((nruns + 3) / 4), dim3(256), 0, ctx->stream, count, (unsigned)ncell, runsum);
  hipLaunchKernelGGL(cell_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, runsum, nruns);
  hipLaunchKernelGGL(cell_scan_add_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, ctx->stream, count, (unsigned)ncell,
                     (const unsigned *)runsum, nruns);
}

/* perm[i] = index of the i-th target in cell order.  Targets per cell ~ `per_cell`. */
int sinterp_sort_targets(gsl_sinterp_hip_ctx *ctx, const double *d_y, size_t m, size_t ytda, int dim, int per_cell,
                         int **d_perm_out)
{
  *d_perm_out = NULL;
  if (m == 0) return ST_SUCCESS;
  if (m > 0x7fffffffULL) return sinterp_fail(ctx, ST_EINVAL, "sort_targets: more than 2^31 targets", hipSuccess, __FILE__, __LINE__);
  double cells = (double)m / (double)(per_cell > 0 ? per_cell : 64);
  int g = (int)ceil(pow(cells < 1 ? 1.0 : cells, 1.0 / dim));
  const int gmax = dim == 1 ? (1 << 20) : (dim == 2 ? 1024 : 100);
  g = g < 1 ? 1 : (g > gmax ? gmax : g);
  size_t ncell = 1;
  for (int c = 0; c < dim; c++) ncell *= (size_t)g;
  void *buf = NULL;
  const size_t bytes = 64 + m * 4 /*perm*/ + m * 4 /*cellid*/ + m * 4 /*slot*/ + (ncell + 1) * 4 + (ncell / 1024 + 8) * 4;
  int st = sinterp_sortbuf(ctx, bytes, &buf);
  if (st) return st;
  unsigned long long *box = (unsigned long long *)buf;
  int *perm = (int *)((char *)buf + 64);
  unsigned *cellid = (unsigned *)(perm + m);
  unsigned *slot = cellid + m;
  unsigned *count = slot + m;
  hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, ctx->stream, box);   /* no host-sourced async copy */
  HIP_OK(ctx, hipMemsetAsync(count, 0, ncell * 4, ctx->stream));
  size_t blocks = (m + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, ctx->stream, d_y, m, ytda, dim, box);
  hipLaunchKernelGGL(cell_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_y, m, ytda, dim, g,
                     (const unsigned long long *)box, cellid, slot, count);
  launch_cell_scan(ctx, count, ncell, count + ncell + 1);
  hipLaunchKernelGGL(cell_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)cellid,
                     (const unsigned *)slot, m, (const unsigned *)count, perm);
  LAUNCH_CHECK(ctx);
  *d_perm_out = perm;
  return ST_SUCCESS;
}

/* cell c occupies perm_in[offset[c] .. offset[c+1]) in the (run-to-run varying) order of the atomic scatter; the
   sweep must see each cell's centres in ORIGINAL index order (fixed summation order).  One thread per centre k:
   its place inside the run is the number of run members with a smaller index -- a rank sort, no serial pass
   (the round-1 kernel sorted every run with one thread by insertion in global memory: 196 us at N = 16384 for a
   few KB of data, 9 % of C3's sweep).  Every cell is ranked, whatever its size: a heavily clustered cloud (or one far
   outlier stretching the bounding box) puts thousands of centres into one cell, and leaving such a cell in scatter
   order made the culled sum's order -- hence its last bits -- vary run to run and between the members of a device
   group.  The loop is O(cell size) per centre; one cell holding all N = 16384 centres costs ~1 ms once per model. */
__global__ void __launch_bounds__(256)
cell_rank_kernel(const unsigned *__restrict__ cellid, const unsigned *__restrict__ slot, const unsigned *__restrict__ offset,
                 const int *__restrict__ perm_in, int *__restrict__ perm_out, size_t n)
{
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const unsigned c = cellid[k], b = offset[c], e = offset[c + 1];
  unsigned rank = 0;
  for (unsigned j = b; j < e; j++) rank += (unsigned)(perm_in[j] < (int)k);
  perm_out[b + rank] = (int)k;
}

/* cell c occupies perm[offset[c] .. offset[c+1]): order every cell's run by original index (superseded by
   cell_rank_kernel; kept for GSL_SINTERP_SERIAL_CELL_ORDER=1) */
__global__ void __launch_bounds__(256)
cell_order_kernel(const unsigned *__restrict__ offset, unsigned ncell, int *__restrict__ perm)
{
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const unsigned b = offset[c], e = offset[c + 1];
  /* one thread per cell: cells hold ~8 centres.  A degenerate cloud (thousands of centres in one cell)
     would turn this insertion sort into seconds of single-thread work; such a cell keeps the order of
     the atomic scatter (the sum is then reproducible to rounding only) */
  if (e - b > 2048u) return;
  for (unsigned i = b + 1; i < e; i++) {
    const int v = perm[i];
    unsigned j = i;
    while (j > b && perm[j - 1] > v) { perm[j] = perm[j - 1]; j--; }
    perm[j] = v;
  }
}

int sinterp_sort_centres(gsl_sinterp_hip_ctx *ctx, const double *d_x, size_t n, size_t xtda, int dim, int per_cell,
                         int **d_perm_out)
{
  *d_perm_out = NULL;
  if (n == 0) return ST_SUCCESS;
  if (n > 0x7fffffffULL) return sinterp_fail(ctx, ST_EINVAL, "sort_centres: more than 2^31 centres", hipSuccess, __FILE__, __LINE__);
  const double cells = (double)n / (double)(per_cell > 0 ? per_cell : 8);
  int g = 1;
  const int gmax = dim == 1 ? (1 << 16) : (dim == 2 ? 512 : 64);
  while (g < gmax && pow((double)(2 * g), dim) <= cells) g *= 2;      /* power of two per axis */
  size_t ncell = 1;
  for (int c = 0; c < dim; c++) ncell *= (size_t)g;
  void *buf = NULL;
  const size_t bytes = 64 + n * 4 + n * 4 + n * 4 + n * 4 + (ncell + 1) * 4 + (ncell / 1024 + 8) * 4;
  int st = sinterp_sortbuf2(ctx, bytes, &buf);
  if (st) return st;
  unsigned long long *box = (unsigned long long *)buf;
  int *perm_sorted = (int *)((char *)buf + 64);
  int *perm = perm_sorted + n;
  unsigned *cellid = (unsigned *)(perm + n);
  unsigned *slot = cellid + n;
  unsigned *count = slot + n;
  hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, ctx->stream, box);
  HIP_OK(ctx, hipMemsetAsync(count, 0, ncell * 4, ctx->stream));
  size_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_x, n, xtda, dim, box);
  hipLaunchKernelGGL(cell_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_x, n, xtda, dim, -g,
                     (const unsigned long long *)box, cellid, slot, count);
  launch_cell_scan(ctx, count, ncell, count + ncell + 1);
  hipLaunchKernelGGL(cell_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)cellid,
                     (const unsigned *)slot, n, (const unsigned *)count, perm);
  static const bool serial_order = getenv("GSL_SINTERP_SERIAL_CELL_ORDER") && getenv("GSL_SINTERP_SERIAL_CELL_ORDER")[0] == '1';
  if (serial_order) {
    hipLaunchKernelGGL(cell_order_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, ctx->stream, (const unsigned *)count,
                       (unsigned)ncell, perm);
    LAUNCH_CHECK(ctx);
    *d_perm_out = perm;
    return ST_SUCCESS;
  }
  hipLaunchKernelGGL(cell_rank_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const unsigned *)cellid,
                     (const unsigned *)slot, (const unsigned *)count, (const int *)perm, perm_sorted, n);
  LAUNCH_CHECK(ctx);
  *d_perm_out = perm_sorted;
  return ST_SUCCESS;
}

/* ---- physical reorder ---------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
cell_scatter_points_kernel(const double *__restrict__ y, size_t m, size_t ytda, int dim, const unsigned *__restrict__ cellid,
                           const unsigned *__restrict__ slot, const unsigned *__restrict__ offset, double *__restrict__ ys)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
    const size_t pos = (size_t)offset[cellid[k]] + slot[k];
    if (dim == 2) {
      double2 v = make_double2(y[k * ytda], y[k * ytda + 1]);
      *reinterpret_cast<double2 *>(ys + pos * 2) = v;               /* one 16-byte store */
    } else {
      for (int c = 0; c < dim; c++) ys[pos * dim + c] = y[k * ytda + c];
    }
  }
}

__global__ void __launch_bounds__(256)
unsort_kernel(const unsigned *__restrict__ cellid, const unsigned *__restrict__ slot, const unsigned *__restrict__ offset, size_t m,
              const double *__restrict__ vs, double *__restrict__ values, const int *__restrict__ ls, int *__restrict__ leaf)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
    const size_t pos = (size_t)offset[cellid[k]] + slot[k];
    if (values) values[k] = vs[pos];
    if (leaf) leaf[k] = ls[pos];
  }
}


/* ---- two-level reorder (large batches) --------------------------------------------------------
   What a pass costs on this memory system is the number of distinct 64-byte segments its wave instructions touch, not
   its bytes: every pass of the one-atomic-per-point scheme above that touches one RANDOM segment per point costs
   0.2 - 0.4 ms per 10^7 points, whatever the operation -- the histogram's returning atomics (tools/atomics_study: 400 us
   with the table shared, private to the XCD of the issuing workgroup, or at any scope: it is the segment rate, not the
   atomic), the 16-byte scatter (285 us), the gather of the un-sort (216 us) -- and a first version of this scheme whose
   scatters were local (inside an L2-resident window) but still one segment per lane was no faster (2.10 vs 1.91 ms at
   C5).  A sequential pass over the same points costs 40 - 60 us.  So every scattered store here goes through LDS first:
   a workgroup orders its chunk in LDS and writes RUNS (consecutive lanes -> consecutive addresses).
       A  coarse histogram   cnt[bin][workgroup], bins = runs of 2^shift consecutive cells, <= TL_NB of them; LDS atomics only
          scan of cnt        -> first position of every (bin, workgroup) run in the coarse order
       B  coarse scatter     chunk ordered by bin in LDS, t_y[p1] written in runs of ~CH/NB points; pos1[k] = p1 (sequential)
       C1 fine histogram     per unit of P consecutive p1: LDS counts over a window of TL_W cells, one global atomicAdd per
                             occupied (unit, cell) reserves the unit's share of the cell
          scan of count      -> offset[cell]
       C2 fine scatter       unit ordered by cell in LDS, ys[p] written in runs (one per occupied cell), inv[p] = p1 beside it
       sweep                 the consumer stores the result of sorted target p at res1[inv[p]] (a store inside the window)
       un-sort               values[k] = res1[pos1[k]]: the ONE random pass that is left
   A point whose cell lies outside its unit's window (sparse regions: a unit spanning > TL_W cells holds < 3 points per
   cell) takes the one-level route for that point: slot by a global atomic in C1, placed at the tail of the LDS image in
   C2.  The cell of a point is recomputed from its coordinates in every pass (tl_cell: subtract, multiply by a
   per-axis factor computed once, truncate -- the same instructions on the same bits each time) instead of being carried
   along.  The order inside a cell is arbitrary (as before); results do not depend on it. */
#define TL_NB 256
#define TL_W 2048
#define TL_MIN_M (1u << 18)
#define TL_THREADS 512
#ifndef TL_CH2
#define TL_CH2 6144
#endif
#ifndef TL_CH3
#define TL_CH3 4096
#endif
#ifndef TL_PU
#define TL_PU 4096
#endif
template <int DIM> struct TlGeom {
  static constexpr int CH = DIM == 3 ? TL_CH3 : TL_CH2;   /* points per workgroup, coarse passes: CH * 8 DIM + 4 CH bytes of LDS */
  static constexpr int P = TL_PU;                         /* points per unit, fine passes: P * (8 DIM + 8) bytes + three windows */
};

/* cell = Morton (bit-interleaved) code of the per-axis indices i_c = min(g-1, max(0, (int)((y_c - lo_c) f_c))): runs of
   consecutive cells -- a workgroup's targets in the sweeps, a coarse bin here -- are compact blocks, not strips (the culled
   Gaussian sweep tests the centre tiles within the cut-off of its workgroup's bounding box: 16 cells as a 4 x 4 block instead
   of a 16 x 1 strip is ~20 % less area at C4's cut-off radius of 15 cells) */
struct TlGrid { double lo[3], f[3]; int g; };

__global__ void tl_grid_kernel(const unsigned long long *__restrict__ box, int dim, int g, TlGrid *__restrict__ out)
{
  if (threadIdx.x != 0) return;
  TlGrid t;
  t.g = g;
  for (int c = 0; c < 3; c++) {
    const double lo = c < dim ? dunkey(box[2 * c]) : 0.0, hi = c < dim ? dunkey(box[2 * c + 1]) : 0.0;
    t.lo[c] = lo;
    t.f[c] = (c < dim && hi > lo) ? (double)g / (hi - lo) : 0.0;
  }
  *out = t;
}

template <int DIM>
__device__ __forceinline__ unsigned tl_cell(const TlGrid &t, const double (&v)[DIM])
{
  unsigned ic[DIM];
#pragma unroll
  for (int c = 0; c < DIM; c++) {
    const double f = __dmul_rn(__dsub_rn(v[c], t.lo[c]), t.f[c]);
    int i = (f == f) ? (f >= 2147483647.0 ? t.g - 1 : (f <= 0.0 ? 0 : (int)f)) : 0;          /* NaN coordinates go to cell 0 */
    ic[c] = (unsigned)(i >= t.g ? t.g - 1 : i);
  }
  if (DIM == 1) return ic[0];
  if (DIM == 2) {
    unsigned cell = 0;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      unsigned x = ic[c] & 0xffffu;                                  /* spread the bits: abcd -> 0a0b0c0d */
      x = (x | (x << 8)) & 0x00ff00ffu; x = (x | (x << 4)) & 0x0f0f0f0fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
      cell |= x << c;
    }
    return cell;
  }
  unsigned cell = 0;
#pragma unroll
  for (int c = 0; c < DIM; c++) {
    unsigned x = ic[c] & 0x3ffu;                                     /* abcd -> 00a00b00c00d */
    x = (x | (x << 16)) & 0x030000ffu; x = (x | (x << 8)) & 0x0300f00fu; x = (x | (x << 4)) & 0x030c30c3u; x = (x | (x << 2)) & 0x09249249u;
    cell |= x << c;
  }
  return cell;
}

/* ALIGNED: the scheme's own dense [.][DIM] arrays (16-byte aligned); the caller's targets are only known to be doubles */
template <int DIM, bool ALIGNED = false>
__device__ __forceinline__ void tl_load(const double *__restrict__ y, size_t k, size_t ytda, double (&v)[DIM])
{
  if (ALIGNED && DIM == 2) { const double2 t = *reinterpret_cast<const double2 *>(y + k * 2); v[0] = t.x; v[1] = t.y; }
  else {
#pragma unroll
    for (int c = 0; c < DIM; c++) v[c] = y[k * ytda + c];
  }
}

/* exclusive scan of a[0..n) in LDS by the whole workgroup (n <= 4 * TL_THREADS); returns the total; s_w: 8 words of LDS */
__device__ __forceinline__ unsigned tl_block_scan(unsigned *a, int n, unsigned *s_w)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned v[4], sum = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) { const int i = tid * 4 + q; v[q] = i < n ? a[i] : 0u; sum += v[q]; }
  unsigned incl = sum;
  for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(incl, off); if (lane >= off) incl += t; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  unsigned woff = 0, total = 0;
  for (int w = 0; w < TL_THREADS / 64; w++) { if (w < wave) woff += s_w[w]; total += s_w[w]; }
  unsigned run = woff + incl - sum;
#pragma unroll
  for (int q = 0; q < 4; q++) { const int i = tid * 4 + q; if (i < n) a[i] = run; run += v[q]; }
  __syncthreads();
  return total;
}

template <int DIM>
__global__ void __launch_bounds__(TL_THREADS)
tl_coarse_hist_kernel(const double *__restrict__ y, size_t m, size_t ytda, const TlGrid *__restrict__ grid, int shift, unsigned nb,
                      unsigned nwg, unsigned *__restrict__ cnt)
{
  constexpr int CH = TlGeom<DIM>::CH;
  __shared__ unsigned h[TL_NB];
  const TlGrid t = *grid;
  for (int i = threadIdx.x; i < TL_NB; i += TL_THREADS) h[i] = 0;
  __syncthreads();
  const size_t k0 = (size_t)blockIdx.x * CH, k1 = k0 + CH < m ? k0 + CH : m;
  for (size_t k = k0 + threadIdx.x; k < k1; k += TL_THREADS) {
    double v[DIM];
    tl_load<DIM>(y, k, ytda, v);
    atomicAdd(&h[tl_cell<DIM>(t, v) >> shift], 1u);
  }
  __syncthreads();
  for (unsigned b = threadIdx.x; b < nb; b += TL_THREADS) cnt[(size_t)b * nwg + blockIdx.x] = h[b];
}

template <int DIM>
__global__ void __launch_bounds__(TL_THREADS)
tl_coarse_scatter_kernel(const double *__restrict__ y, size_t m, size_t ytda, const TlGrid *__restrict__ grid, int shift, unsigned nb,
                         unsigned nwg, const unsigned *__restrict__ cnt, double *__restrict__ t_y, unsigned *__restrict__ pos1)
{
  constexpr int CH = TlGeom<DIM>::CH, PT = CH / TL_THREADS;
  extern __shared__ __attribute__((aligned(16))) double tl_lds[];
  double *ly = tl_lds;                                              /* [CH][DIM] the chunk in bin order */
  unsigned *ldest = (unsigned *)(ly + (size_t)CH * DIM);            /* [CH] destination of every LDS slot */
  __shared__ unsigned lh[TL_NB], gbase[TL_NB], s_w[8];
  const TlGrid t = *grid;
  for (unsigned b = threadIdx.x; b < TL_NB; b += TL_THREADS) { lh[b] = 0; gbase[b] = b < nb ? cnt[(size_t)b * nwg + blockIdx.x] : 0u; }
  __syncthreads();
  const size_t k0 = (size_t)blockIdx.x * CH, k1 = k0 + CH < m ? k0 + CH : m;
  double v[PT][DIM];
  unsigned bb[PT], rr[PT];
#pragma unroll
  for (int i = 0; i < PT; i++) {
    const size_t k = k0 + (size_t)i * TL_THREADS + threadIdx.x;
    if (k < k1) {
      tl_load<DIM>(y, k, ytda, v[i]);
      bb[i] = tl_cell<DIM>(t, v[i]) >> shift;
      rr[i] = atomicAdd(&lh[bb[i]], 1u);
    }
  }
  __syncthreads();
  tl_block_scan(lh, TL_NB, s_w);                                    /* lh: first LDS slot of every bin */
#pragma unroll
  for (int i = 0; i < PT; i++) {
    const size_t k = k0 + (size_t)i * TL_THREADS + threadIdx.x;
    if (k < k1) {
      const unsigned lp = lh[bb[i]] + rr[i], dest = gbase[bb[i]] + rr[i];
#pragma unroll
      for (int c = 0; c < DIM; c++) ly[(size_t)lp * DIM + c] = v[i][c];
      ldest[lp] = dest;
      pos1[k] = dest;
    }
  }
  __syncthreads();
  const unsigned nloc = (unsigned)(k1 - k0);
  for (unsigned sl = threadIdx.x; sl < nloc; sl += TL_THREADS) {
    const size_t d = ldest[sl];
    if (DIM == 2) *reinterpret_cast<double2 *>(t_y + d * 2) = *reinterpret_cast<const double2 *>(ly + (size_t)sl * 2);
    else {
#pragma unroll
      for (int c = 0; c < DIM; c++) t_y[d * DIM + c] = ly[(size_t)sl * DIM + c];
    }
  }
}

template <int DIM>
__global__ void __launch_bounds__(TL_THREADS)
tl_fine_hist_kernel(const double *__restrict__ t_y, size_t m, const TlGrid *__restrict__ grid, int shift, unsigned *__restrict__ count,
                    unsigned *__restrict__ ubase, unsigned *__restrict__ fin)
{
  constexpr int P = TlGeom<DIM>::P;
  __shared__ unsigned h[TL_W];
  const TlGrid t = *grid;
  for (int i = threadIdx.x; i < TL_W; i += TL_THREADS) h[i] = 0;
  __syncthreads();
  const size_t i0 = (size_t)blockIdx.x * P, i1 = i0 + P < m ? i0 + P : m;
  double v0[DIM];
  tl_load<DIM, true>(t_y, i0, DIM, v0);
  const unsigned c_first = (tl_cell<DIM>(t, v0) >> shift) << shift;   /* bins ascend along p1: no cell of the unit is below its first point's bin */
  for (size_t i = i0 + threadIdx.x; i < i1; i += TL_THREADS) {
    double v[DIM];
    tl_load<DIM, true>(t_y, i, DIM, v);
    const unsigned c = tl_cell<DIM>(t, v), d = c - c_first;
    if (d < TL_W) atomicAdd(&h[d], 1u);
    else fin[i] = atomicAdd(&count[c], 1u);
  }
  __syncthreads();
  for (unsigned d = threadIdx.x; d < TL_W; d += TL_THREADS)
    if (h[d]) ubase[(size_t)blockIdx.x * TL_W + d] = atomicAdd(&count[c_first + d], h[d]);
}

template <int DIM>
__global__ void __launch_bounds__(TL_THREADS)
tl_fine_scatter_kernel(const double *__restrict__ t_y, size_t m, const TlGrid *__restrict__ grid, int shift, unsigned ncell,
                       const unsigned *__restrict__ offset, const unsigned *__restrict__ ubase, const unsigned *__restrict__ fin,
                       double *__restrict__ ys, unsigned *__restrict__ inv)
{
  constexpr int P = TlGeom<DIM>::P, PT = P / TL_THREADS;
  extern __shared__ __attribute__((aligned(16))) double tl_lds[];
  double *ly = tl_lds;                                              /* [P][DIM] the unit in cell order, the out-of-window points at the tail */
  unsigned *ldest = (unsigned *)(ly + (size_t)P * DIM);             /* [P] position in cell order */
  unsigned *lsrc = ldest + P;                                       /* [P] position in the coarse order */
  unsigned *lcnt = lsrc + P;                                        /* [TL_W] counts, then first LDS slot of every cell */
  unsigned *ub = lcnt + TL_W;                                       /* [TL_W] offset[cell] + the unit's share of the cell (C1) */
  __shared__ unsigned s_w[8], s_tail;
  const TlGrid t = *grid;
  const size_t i0 = (size_t)blockIdx.x * P, i1 = i0 + P < m ? i0 + P : m;
  double v0[DIM];
  tl_load<DIM, true>(t_y, i0, DIM, v0);
  const unsigned c_first = (tl_cell<DIM>(t, v0) >> shift) << shift;
  /* (entries of cells the unit does not hold are never read: ubase is undefined there, offset may lie past the table) */
  for (unsigned d = threadIdx.x; d < TL_W; d += TL_THREADS) {
    lcnt[d] = 0;
    ub[d] = (c_first + d < ncell ? offset[c_first + d] : 0u) + ubase[(size_t)blockIdx.x * TL_W + d];
  }
  if (threadIdx.x == 0) s_tail = 0;
  __syncthreads();
  double v[PT][DIM];
  unsigned cc[PT], rr[PT];
#pragma unroll
  for (int q = 0; q < PT; q++) {
    const size_t i = i0 + (size_t)q * TL_THREADS + threadIdx.x;
    if (i < i1) {
      tl_load<DIM, true>(t_y, i, DIM, v[q]);
      cc[q] = tl_cell<DIM>(t, v[q]);
      const unsigned d = cc[q] - c_first;
      rr[q] = d < TL_W ? atomicAdd(&lcnt[d], 1u) : atomicAdd(&s_tail, 1u);
    }
  }
  __syncthreads();
  const unsigned nwin = tl_block_scan(lcnt, TL_W, s_w);
#pragma unroll
  for (int q = 0; q < PT; q++) {
    const size_t i = i0 + (size_t)q * TL_THREADS + threadIdx.x;
    if (i < i1) {
      const unsigned d = cc[q] - c_first;
      const unsigned lp = d < TL_W ? lcnt[d] + rr[q] : nwin + rr[q];
      const unsigned dest = d < TL_W ? ub[d] + rr[q] : offset[cc[q]] + fin[i];
#pragma unroll
      for (int c = 0; c < DIM; c++) ly[(size_t)lp * DIM + c] = v[q][c];
      ldest[lp] = dest;
      lsrc[lp] = (unsigned)i;
    }
  }
  __syncthreads();
  const unsigned nloc = (unsigned)(i1 - i0);
  for (unsigned sl = threadIdx.x; sl < nloc; sl += TL_THREADS) {
    const size_t d = ldest[sl];
    if (DIM == 2) *reinterpret_cast<double2 *>(ys + d * 2) = *reinterpret_cast<const double2 *>(ly + (size_t)sl * 2);
    else {
#pragma unroll
      for (int c = 0; c < DIM; c++) ys[d * DIM + c] = ly[(size_t)sl * DIM + c];
    }
    inv[d] = lsrc[sl];
  }
}

/* the one random gather of the two-level scheme: 16 (8) bytes per target from the results in the coarse order */
template <int PACKED>
__global__ void __launch_bounds__(256)
tl_unsort_final_kernel(const unsigned *__restrict__ pos1, size_t m, const double *__restrict__ res, double *__restrict__ values,
                       int *__restrict__ leaf)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
    const size_t p = pos1[k];
    if (PACKED) {
      const double2 r = reinterpret_cast<const double2 *>(res)[p];
      if (values) values[k] = r.x;
      if (leaf) leaf[k] = (int)__double_as_longlong(r.y);
    } else values[k] = res[p];
  }
}

/* The same gather staged through LDS: workgroup w of the coarse scatter wrote its chunk as <= TL_NB runs, one per bin, at
   positions [run[b], run[b] + len[b]) of the coarse order (run = the scanned cnt[bin][workgroup] matrix, len = the
   difference to the next entry).  The runs are copied into LDS with consecutive lanes on consecutive positions, then
   target k reads its result from the LDS image (bin of pos1[k] by binary search in run[]): the global side of the un-sort is
   sequential.  ESZ: doubles per result (1: value, 2: {value, leaf}). */
template <int ESZ, int CH>
__global__ void __launch_bounds__(TL_THREADS)
tl_unsort_staged_kernel(const unsigned *__restrict__ pos1, size_t m, const unsigned *__restrict__ cnt, unsigned nb, unsigned nwg,
                        const double *__restrict__ res, double *__restrict__ values, int *__restrict__ leaf)
{
  extern __shared__ __attribute__((aligned(16))) double tl_lds[];
  double *lres = tl_lds;                                            /* [CH][ESZ] */
  __shared__ unsigned run[TL_NB], lstart[TL_NB + 1], s_w[8];
  for (unsigned b = threadIdx.x; b < TL_NB; b += TL_THREADS) {
    unsigned r = 0, len = 0;
    if (b < nb) { const size_t idx = (size_t)b * nwg + blockIdx.x; r = cnt[idx]; len = cnt[idx + 1] - r; }
    run[b] = r; lstart[b] = len;
  }
  __syncthreads();
  const unsigned nloc = tl_block_scan(lstart, TL_NB, s_w);          /* lstart: first LDS slot of every run */
  if (threadIdx.x == 0) lstart[TL_NB] = nloc;
  __syncthreads();
  for (unsigned sl = threadIdx.x; sl < nloc; sl += TL_THREADS) {
    unsigned lo = 0, hi = nb;                                       /* largest b with lstart[b] <= sl (empty runs share a start: take the last) */
    while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (lstart[mid] <= sl) lo = mid; else hi = mid; }
    const size_t src = (size_t)run[lo] + (sl - lstart[lo]);
    if (ESZ == 2) reinterpret_cast<double2 *>(lres)[sl] = reinterpret_cast<const double2 *>(res)[src];
    else lres[sl] = res[src];
  }
  __syncthreads();
  const size_t k0 = (size_t)blockIdx.x * CH, k1 = k0 + CH < m ? k0 + CH : m;
  for (size_t k = k0 + threadIdx.x; k < k1; k += TL_THREADS) {
    const unsigned p = pos1[k];
    unsigned lo = 0, hi = nb;                                       /* the run that holds p: the largest b with run[b] <= p (runs of later bins start past p) */
    while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (run[mid] <= p) lo = mid; else hi = mid; }
    const unsigned sl = lstart[lo] + (p - run[lo]);
    if (ESZ == 2) {
      const double2 r = reinterpret_cast<const double2 *>(lres)[sl];
      if (values) values[k] = r.x;
      if (leaf) leaf[k] = (int)__double_as_longlong(r.y);
    } else values[k] = lres[sl];
  }
}

static bool sort_two_level(size_t m)
{
  const char *e = getenv("GSL_SINTERP_SORT_LEVELS");         /* developer override "1" / "2"; read per call: the tests compare both routes */
  const bool two = e && (e[0] == '1' || e[0] == '2') ? e[0] == '2' : true;
  return two && m >= TL_MIN_M;
}

bool sinterp_sort_reorder_is_two_level(size_t m) { return sort_two_level(m); }

template <int DIM>
static int tl_launch(gsl_sinterp_hip_ctx *ctx, const double *d_y, size_t m, size_t ytda, size_t ncell, const TlGrid *grid, double *t_y,
                     unsigned *cnt, unsigned *ubase, sinterp_sorted *out)
{
  constexpr int CH = TlGeom<DIM>::CH, P = TlGeom<DIM>::P;
  int shift = 0;
  while (((ncell - 1) >> shift) >= TL_NB) shift++;
  const unsigned nb = (unsigned)((ncell - 1) >> shift) + 1u, nwg = (unsigned)((m + CH - 1) / CH), nu = (unsigned)((m + P - 1) / P);
  const size_t nc = (size_t)nb * nwg;
  const size_t lds_b = (size_t)CH * DIM * 8 + (size_t)CH * 4, lds_c = (size_t)P * DIM * 8 + (size_t)P * 8 + (size_t)TL_W * 8;
  { int ast = sinterp_func_lds(ctx, (const void *)tl_coarse_scatter_kernel<DIM>, (int)lds_b); if (ast) return ast; }
  { int ast = sinterp_func_lds(ctx, (const void *)tl_fine_scatter_kernel<DIM>, (int)lds_c); if (ast) return ast; }
  hipLaunchKernelGGL((tl_coarse_hist_kernel<DIM>), dim3(nwg), dim3(TL_THREADS), 0, ctx->stream, d_y, m, ytda, grid, shift, nb, nwg, cnt);
  launch_cell_scan(ctx, cnt, nc, cnt + nc + 1);
  hipLaunchKernelGGL((tl_coarse_scatter_kernel<DIM>), dim3(nwg), dim3(TL_THREADS), lds_b, ctx->stream, d_y, m, ytda, grid, shift, nb, nwg,
                     (const unsigned *)cnt, t_y, out->slot);
  hipLaunchKernelGGL((tl_fine_hist_kernel<DIM>), dim3(nu), dim3(TL_THREADS), 0, ctx->stream, (const double *)t_y, m, grid, shift, out->offset,
                     ubase, out->fin);
  launch_cell_scan(ctx, out->offset, ncell, out->offset + ncell + 1);
  hipLaunchKernelGGL((tl_fine_scatter_kernel<DIM>), dim3(nu), dim3(TL_THREADS), lds_c, ctx->stream, (const double *)t_y, m, grid, shift,
                     (unsigned)ncell, (const unsigned *)out->offset, (const unsigned *)ubase, (const unsigned *)out->fin, out->ys, out->inv);
  out->tl_cnt = cnt; out->tl_nb = nb; out->tl_nwg = nwg; out->tl_ch = (unsigned)CH;
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

/* un-sort of a two-level order: ESZ doubles per result */
template <int ESZ>
static int tl_unsort(gsl_sinterp_hip_ctx *ctx, const sinterp_sorted *s, size_t m, double *d_values, int *d_leaf)
{
  /* measured at M = 10^7: 8-byte results 85 us staged against ~100 us for the plain gather (three workgroups per CU);
     16-byte {value, leaf} pairs 200 us staged against 130 us (98 KB of LDS: one workgroup per CU, nothing hides the table
     loads and the searches) -- the pairs keep the plain gather.  GSL_SINTERP_UNSORT_GATHER=1 (developer): plain gather always */
  static const bool unstaged = getenv("GSL_SINTERP_UNSORT_GATHER") && getenv("GSL_SINTERP_UNSORT_GATHER")[0] == '1';
  if (unstaged || !s->tl_cnt || ESZ == 2) {
    size_t blocks = (m + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((tl_unsort_final_kernel<ESZ == 2>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)s->slot, m,
                       (const double *)s->res1, d_values, d_leaf);
  } else if (s->tl_ch == (unsigned)TlGeom<2>::CH) {
    constexpr int CH = TlGeom<2>::CH;
    const size_t lds = (size_t)CH * ESZ * 8;
    { int ast = sinterp_func_lds(ctx, (const void *)tl_unsort_staged_kernel<ESZ, CH>, (int)lds); if (ast) return ast; }
    hipLaunchKernelGGL((tl_unsort_staged_kernel<ESZ, CH>), dim3(s->tl_nwg), dim3(TL_THREADS), lds, ctx->stream, (const unsigned *)s->slot, m,
                       (const unsigned *)s->tl_cnt, s->tl_nb, s->tl_nwg, (const double *)s->res1, d_values, d_leaf);
  } else {
    constexpr int CH = TlGeom<3>::CH;
    const size_t lds = (size_t)CH * ESZ * 8;
    { int ast = sinterp_func_lds(ctx, (const void *)tl_unsort_staged_kernel<ESZ, CH>, (int)lds); if (ast) return ast; }
    hipLaunchKernelGGL((tl_unsort_staged_kernel<ESZ, CH>), dim3(s->tl_nwg), dim3(TL_THREADS), lds, ctx->stream, (const unsigned *)s->slot, m,
                       (const unsigned *)s->tl_cnt, s->tl_nb, s->tl_nwg, (const double *)s->res1, d_values, d_leaf);
  }
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

/* m_cap >= m sizes the buffer section (two sections -- `slot` 0 / 1 -- so that two chunks of one batch can be in
   flight on two streams); box_in != NULL: bounding-box keys to bin by (e.g. the data's box kept with the jump table)
   instead of a pass over the targets -- points outside it land in the border cells, which only costs locality */
int sinterp_sort_reorder(gsl_sinterp_hip_ctx *ctx, const double *d_y, size_t m, size_t ytda, int dim, int per_cell,
                         sinterp_sorted *out, size_t m_cap, int slot, const unsigned long long *box_in)
{
  memset(out, 0, sizeof *out);
  if (m == 0) return ST_SUCCESS;
  if (m_cap < m) m_cap = m;
  if (m_cap > 0x7fffffffULL) return sinterp_fail(ctx, ST_EINVAL, "sort_reorder: more than 2^31 targets", hipSuccess, __FILE__, __LINE__);
  const int gmax = dim == 1 ? (1 << 20) : (dim == 2 ? 1024 : 100);
  /* cells per axis for a batch of mm targets.  The two-level route numbers its cells by Morton code: a power-of-two grid
     makes the id space dense (runs of ids are compact blocks at every scale) -- C3 sweep 1.27 -> 1.14 ms, C4 1.19 -> 1.15,
     C5 1.57 -> 1.55 against the same scheme on the 25^3 / 395^2 grids; the nearer power of two in ratio, so a cell holds
     between per_cell / 2^(dim/2) and per_cell 2^(dim/2) targets.  Monotone in mm (the capacity below relies on it). */
  auto grid_of = [&](size_t mm) {
    const double cells = (double)mm / (double)(per_cell > 0 ? per_cell : 64);
    int gg = (int)ceil(pow(cells < 1 ? 1.0 : cells, 1.0 / dim));
    gg = gg < 1 ? 1 : (gg > gmax ? gmax : gg);
    if (dim >= 2 && sort_two_level(mm)) {
      int p2 = 1;
      while (p2 * 2 <= gg) p2 *= 2;                    /* p2 <= gg < 2 p2 */
      gg = ((double)gg / p2 > (double)(2 * p2) / gg) ? 2 * p2 : p2;
      const int gcap = dim == 2 ? 1024 : 128;
      if (gg > gcap) gg = gcap;
    }
    return gg;
  };
  const int g = grid_of(m), gc = grid_of(m_cap);
  /* the two-level route's id space is (2^bits)^dim, 2^bits >= g (= g^dim for its power-of-two grids) */
  auto id_space = [&](int gg, bool morton) {
    int side = gg;
    if (morton && dim >= 2) { side = 1; while (side < gg) side *= 2; }
    size_t c = 1;
    for (int q = 0; q < dim; q++) c *= (size_t)side;
    return c;
  };
  const size_t ncell = id_space(g, sort_two_level(m));
  size_t ncell_cap = id_space(gc, sort_two_level(m_cap));
  {
    /* a smaller batch in a section sized for m_cap may take the one-level route on a (non power-of-two) grid */
    const double cc = (double)m_cap / (double)(per_cell > 0 ? per_cell : 64);
    int gr = (int)ceil(pow(cc < 1 ? 1.0 : cc, 1.0 / dim));
    gr = gr < 1 ? 1 : (gr > gmax ? gmax : gr);
    const size_t raw = id_space(gr, false);
    if (ncell_cap < raw) ncell_cap = raw;
    if (ncell_cap < ncell) ncell_cap = ncell;
  }
  /* layout: box | ys | vs | ls | cellid | slot | count(+1) [| t_y (later res1) | inv | fin | cnt | ubase | grid : two-level] ; every section
     16-byte aligned */
  auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const bool two = sort_two_level(m_cap) && dim >= 1 && dim <= 3;
  const size_t tl_ch = dim == 3 ? TlGeom<3>::CH : TlGeom<2>::CH, tl_p = TlGeom<2>::P;
  const size_t nwg_cap = (m_cap + tl_ch - 1) / tl_ch, nu_cap = (m_cap + tl_p - 1) / tl_p;
  const size_t cnt_n = (size_t)TL_NB * nwg_cap;
  const size_t o_ys = 64, o_vs = o_ys + up(m_cap * dim * 8), o_ls = o_vs + up(m_cap * 16), o_cell = o_ls + up(m_cap * 4),
               o_slot = o_cell + up(m_cap * 4), o_cnt = o_slot + up(m_cap * 4),
               o_ty = o_cnt + up((ncell_cap + 1) * 4 + (ncell_cap / 1024 + 8) * 4),
               o_inv = o_ty + (two ? up(m_cap * (dim * 8 > 16 ? dim * 8 : 16)) : 0), o_fin = o_inv + (two ? up(m_cap * 4) : 0),
               o_ca = o_fin + (two ? up(m_cap * 4) : 0), o_ub = o_ca + (two ? up((cnt_n + 1) * 4 + (cnt_n / 1024 + 8) * 4) : 0),
               o_grid = o_ub + (two ? up(nu_cap * TL_W * 4) : 0), bytes = (o_grid + (two ? 64 : 0) + 255) & ~(size_t)255;
  void *buf = NULL;
  int st = sinterp_sortbuf(ctx, bytes * (slot >= 0 ? 2 : 1), &buf);
  if (st) return st;
  char *b = (char *)buf + (slot > 0 ? bytes : 0);
  out->box = (unsigned long long *)b;
  out->ys = (double *)(b + o_ys); out->vs = (double *)(b + o_vs); out->ls = (int *)(b + o_ls);
  out->cellid = (unsigned *)(b + o_cell); out->slot = (unsigned *)(b + o_slot); out->offset = (unsigned *)(b + o_cnt);
  out->two_level = two && m >= TL_MIN_M;
  out->fin = (unsigned *)(b + o_fin); out->res1 = (double *)(b + o_ty); out->inv = (unsigned *)(b + o_inv);
  HIP_OK(ctx, hipMemsetAsync(out->offset, 0, ncell * 4, ctx->stream));
  size_t blocks = (m + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const unsigned long long *box = box_in;
  if (!box) {
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, ctx->stream, out->box);
    hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, ctx->stream, d_y, m, ytda, dim, out->box);
    box = out->box;
  } else out->box = (unsigned long long *)box_in;
  if (out->two_level) {
    TlGrid *grid = (TlGrid *)(b + o_grid);
    double *t_y = (double *)(b + o_ty);
    unsigned *cnt = (unsigned *)(b + o_ca), *ubase = (unsigned *)(b + o_ub);
    hipLaunchKernelGGL(tl_grid_kernel, dim3(1), dim3(64), 0, ctx->stream, box, dim, g, grid);
    if (dim == 2) return tl_launch<2>(ctx, d_y, m, ytda, ncell, grid, t_y, cnt, ubase, out);
    if (dim == 3) return tl_launch<3>(ctx, d_y, m, ytda, ncell, grid, t_y, cnt, ubase, out);
    return tl_launch<1>(ctx, d_y, m, ytda, ncell, grid, t_y, cnt, ubase, out);
  }
  hipLaunchKernelGGL(cell_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_y, m, ytda, dim, g,
                     box, out->cellid, out->slot, out->offset);
  launch_cell_scan(ctx, out->offset, ncell, out->offset + ncell + 1);
  hipLaunchKernelGGL(cell_scatter_points_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_y, m, ytda, dim,
                     (const unsigned *)out->cellid, (const unsigned *)out->slot, (const unsigned *)out->offset, out->ys);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

__global__ void __launch_bounds__(256)
unsort_packed_kernel(const unsigned *__restrict__ cellid, const unsigned *__restrict__ slot, const unsigned *__restrict__ offset, size_t m,
                     const double2 *__restrict__ vl, double *__restrict__ values, int *__restrict__ leaf)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
    const size_t pos = (size_t)offset[cellid[k]] + slot[k];
    const double2 r = vl[pos];                                  /* one 16-byte gather */
    values[k] = r.x;
    leaf[k] = (int)__double_as_longlong(r.y);
  }
}

int sinterp_unsort_packed(gsl_sinterp_hip_ctx *ctx, const sinterp_sorted *s, size_t m, double *d_values, int *d_leaf)
{
  if (m == 0) return ST_SUCCESS;
  size_t blocks = (m + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (s->two_level) return tl_unsort<2>(ctx, s, m, d_values, d_leaf);   /* the sweep stored {value, leaf} pairs at res1[inv[p]] */
  hipLaunchKernelGGL(unsort_packed_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)s->cellid,
                     (const unsigned *)s->slot, (const unsigned *)s->offset, m, (const double2 *)s->vs, d_values, d_leaf);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

int sinterp_unsort(gsl_sinterp_hip_ctx *ctx, const sinterp_sorted *s, size_t m, double *d_values, int *d_leaf)
{
  if (m == 0 || (!d_values && !d_leaf)) return ST_SUCCESS;
  size_t blocks = (m + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (s->two_level) {                            /* the sweep stored plain values at res1[inv[p]] */
    if (d_leaf) return sinterp_fail(ctx, ST_EINVAL, "unsort: leaf output of a two-level reorder is packed", hipSuccess, __FILE__, __LINE__);
    return tl_unsort<1>(ctx, s, m, d_values, (int *)NULL);
  }
  hipLaunchKernelGGL(unsort_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)s->cellid,
                     (const unsigned *)s->slot, (const unsigned *)s->offset, m, (const double *)s->vs, d_values,
                     (const int *)s->ls, d_leaf);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

int sinterp_bbox_keys(gsl_sinterp_hip_ctx *ctx, const double *d_p, size_t n, size_t tda, int dim, unsigned long long *d_box)
{
  hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, ctx->stream, d_box);
  if (n) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_p, n, tda, dim, d_box);
  }
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}
