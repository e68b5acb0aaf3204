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
   Every pass of the one-atomic-per-point scheme above that touches a RANDOM line per point costs 0.2 - 0.4 ms per 10^7
   points on this memory system, whatever the operation: the histogram's returning atomics (tools/atomics_study:
   400 us with the table shared, private to the XCD, or at any scope -- it is the line rate, not the atomic), the
   16-byte scatter (285 us), the gather of the un-sort (216 us); a sequential pass over the same points costs 40 - 60 us.
   Here the points are first partitioned into <= 1024 coarse bins (runs of consecutive cells) with workgroup-private LDS
   histograms -- no global atomic, every write lands in a run of the workgroup's chunk -- and then ordered by cell inside
   windows of TL_W consecutive cells, again in LDS, with one global atomic per (unit, occupied cell) to reserve the unit's
   share of the cell.  Only the final gather of the un-sort remains a random pass:
       A  coarse histogram      cellid[k];  cnt[bin][workgroup]          (sequential read, LDS atomics)
          scan of cnt           -> first position of every (bin, workgroup) run
       B  coarse scatter        t_y / t_c[p1] = point / cell, pos1[k] = p1 (runs of ~CH/NB points)
       C1 fine histogram        per unit of TL_P consecutive p1: LDS counts of its cells, one atomicAdd per occupied cell
          scan of count         -> offset[cell]                            (as in the one-level scheme)
       C2 fine scatter          ys[p] = t_y[p1], fin[p1] = p               (p within the window: local writes)
       un-sort                  res1[p1] = vs[fin[p1]] (local gather), values[k] = res1[pos1[k]] (the random pass)
   A point whose cell lies outside its unit's window (sparse regions: a unit spanning > TL_W cells holds < 4 points per
   cell) takes the one-level route for that point: slot by a global atomic in C1, position in C2.  The order inside a
   cell is arbitrary (as before); results do not depend on it. */
#define TL_CH 16384
#define TL_P 8192
#define TL_W 2048
#define TL_NB 1024
#define TL_MIN_M (1u << 18)

__global__ void __launch_bounds__(256)
tl_coarse_hist_kernel(const double *__restrict__ y, size_t m, size_t ytda, int dim, int g, const unsigned long long *__restrict__ box,
                      int shift, unsigned nb, unsigned nwg, unsigned *__restrict__ cellid, unsigned *__restrict__ cnt)
{
  __shared__ unsigned h[TL_NB];
  for (int i = threadIdx.x; i < TL_NB; i += 256) h[i] = 0;
  __syncthreads();
  const size_t k0 = (size_t)blockIdx.x * TL_CH, k1 = k0 + TL_CH < m ? k0 + TL_CH : m;
  for (size_t k = k0 + threadIdx.x; k < k1; k += 256) {
    const unsigned c = cell_of(y, k, ytda, dim, g, box);
    cellid[k] = c;
    atomicAdd(&h[c >> shift], 1u);
  }
  __syncthreads();
  for (unsigned b = threadIdx.x; b < nb; b += 256) cnt[(size_t)b * nwg + blockIdx.x] = h[b];
}

template <int DIM>
__global__ void __launch_bounds__(256)
tl_coarse_scatter_kernel(const double *__restrict__ y, size_t m, size_t ytda, const unsigned *__restrict__ cellid, int shift, unsigned nb,
                         unsigned nwg, const unsigned *__restrict__ cnt, double *__restrict__ t_y, unsigned *__restrict__ t_c,
                         unsigned *__restrict__ pos1)
{
  __shared__ unsigned base[TL_NB], cur[TL_NB];
  for (unsigned b = threadIdx.x; b < TL_NB; b += 256) { base[b] = b < nb ? cnt[(size_t)b * nwg + blockIdx.x] : 0u; cur[b] = 0; }
  __syncthreads();
  const size_t k0 = (size_t)blockIdx.x * TL_CH, k1 = k0 + TL_CH < m ? k0 + TL_CH : m;
  for (size_t k = k0 + threadIdx.x; k < k1; k += 256) {
    const unsigned c = cellid[k], b = c >> shift;
    const unsigned p = base[b] + atomicAdd(&cur[b], 1u);
    if (DIM == 2) *reinterpret_cast<double2 *>(t_y + (size_t)p * 2) = make_double2(y[k * ytda], y[k * ytda + 1]);
    else
      for (int d = 0; d < DIM; d++) t_y[(size_t)p * DIM + d] = y[k * ytda + d];
    t_c[p] = c;
    pos1[k] = p;
  }
}

__global__ void __launch_bounds__(256)
tl_fine_hist_kernel(const unsigned *__restrict__ t_c, size_t m, int shift, unsigned *__restrict__ count, unsigned *__restrict__ ubase,
                    unsigned *__restrict__ fin)
{
  __shared__ unsigned h[TL_W];
  for (int i = threadIdx.x; i < TL_W; i += 256) h[i] = 0;
  __syncthreads();
  const size_t i0 = (size_t)blockIdx.x * TL_P, i1 = i0 + TL_P < m ? i0 + TL_P : m;
  const unsigned c_first = (t_c[i0] >> shift) << shift;   /* bins ascend along p1: no cell of the unit is below its first point's bin */
  for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const unsigned c = t_c[i], d = c - c_first;
    if (d < TL_W) atomicAdd(&h[d], 1u);
    else fin[i] = atomicAdd(&count[c], 1u);
  }
  __syncthreads();
  for (unsigned d = threadIdx.x; d < TL_W; d += 256)
    if (h[d]) ubase[(size_t)blockIdx.x * TL_W + d] = atomicAdd(&count[c_first + d], h[d]);
}

template <int DIM>
__global__ void __launch_bounds__(256)
tl_fine_scatter_kernel(const double *__restrict__ t_y, const unsigned *__restrict__ t_c, size_t m, int shift,
                       const unsigned *__restrict__ offset, const unsigned *__restrict__ ubase, unsigned *__restrict__ fin,
                       double *__restrict__ ys)
{
  __shared__ unsigned ub[TL_W], cur[TL_W];
  for (unsigned d = threadIdx.x; d < TL_W; d += 256) { ub[d] = ubase[(size_t)blockIdx.x * TL_W + d]; cur[d] = 0; }   /* unused entries: never read */
  __syncthreads();
  const size_t i0 = (size_t)blockIdx.x * TL_P, i1 = i0 + TL_P < m ? i0 + TL_P : m;
  const unsigned c_first = (t_c[i0] >> shift) << shift;
  for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const unsigned c = t_c[i], d = c - c_first;
    const unsigned p = offset[c] + (d < TL_W ? ub[d] + atomicAdd(&cur[d], 1u) : fin[i]);
    if (DIM == 2) *reinterpret_cast<double2 *>(ys + (size_t)p * 2) = *reinterpret_cast<const double2 *>(t_y + i * 2);
    else
      for (int q = 0; q < DIM; q++) ys[(size_t)p * DIM + q] = t_y[i * DIM + q];
    fin[i] = p;
  }
}

/* un-sort, first hop: results from cell order back to the coarse order (a gather inside the unit's window) */
template <int PACKED>
__global__ void __launch_bounds__(256)
tl_unsort_local_kernel(const unsigned *__restrict__ fin, size_t m, const double *__restrict__ vs, const int *__restrict__ ls,
                       double *__restrict__ res)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    const size_t p = fin[i];
    if (PACKED == 1) reinterpret_cast<double2 *>(res)[i] = reinterpret_cast<const double2 *>(vs)[p];
    else if (PACKED == 2) reinterpret_cast<double2 *>(res)[i] = make_double2(vs[p], __longlong_as_double((long long)ls[p]));
    else res[i] = vs[p];
  }
}

/* second hop: the one random gather, 16 (8) bytes per target */
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

static bool sort_two_level(size_t m)
{
  const char *e = getenv("GSL_SINTERP_SORT_TWO_LEVEL");      /* opt-in; read per call: the tests compare both routes in one process */
  return e && e[0] == '1' && m >= TL_MIN_M;
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
  double cells = (double)m / (double)(per_cell > 0 ? per_cell : 64);
  int g = (int)ceil(pow(cells < 1 ? 1.0 : cells, 1.0 / dim));
  const int gmax = dim == 1 ? (1 << 20) : (dim == 2 ? 1024 : 100);
  g = g < 1 ? 1 : (g > gmax ? gmax : g);
  size_t ncell = 1, ncell_cap = 1;
  for (int c = 0; c < dim; c++) ncell *= (size_t)g;
  {
    double cc = (double)m_cap / (double)(per_cell > 0 ? per_cell : 64);
    int gc = (int)ceil(pow(cc < 1 ? 1.0 : cc, 1.0 / dim));
    gc = gc < 1 ? 1 : (gc > gmax ? gmax : gc);
    for (int c = 0; c < dim; c++) ncell_cap *= (size_t)gc;
    if (ncell_cap < ncell) ncell_cap = ncell;
  }
  /* layout: box | ys | vs | ls | cellid | slot | count(+1) [| t_y | t_c | fin | cnt | ubase : two-level] ; every section
     16-byte aligned */
  auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const bool two = sort_two_level(m_cap) && dim >= 1 && dim <= 3;
  const size_t nwg_cap = (m_cap + TL_CH - 1) / TL_CH, nu_cap = (m_cap + TL_P - 1) / TL_P;
  const size_t cnt_n = (size_t)TL_NB * nwg_cap;
  const size_t o_ys = 64, o_vs = o_ys + up(m_cap * dim * 8), o_ls = o_vs + up(m_cap * 16), o_cell = o_ls + up(m_cap * 4),
               o_slot = o_cell + up(m_cap * 4), o_cnt = o_slot + up(m_cap * 4),
               o_ty = o_cnt + up((ncell_cap + 1) * 4 + (ncell_cap / 1024 + 8) * 4),
               o_tc = o_ty + (two ? up(m_cap * (dim * 8 > 16 ? dim * 8 : 16)) : 0), o_fin = o_tc + (two ? up(m_cap * 4) : 0),
               o_ca = o_fin + (two ? up(m_cap * 4) : 0), o_ub = o_ca + (two ? up((cnt_n + 1) * 4 + (cnt_n / 1024 + 8) * 4) : 0),
               bytes = (o_ub + (two ? up(nu_cap * TL_W * 4) : 0) + 255) & ~(size_t)255;
  void *buf = NULL;
  int st = sinterp_sortbuf(ctx, bytes * (slot >= 0 ? 2 : 1), &buf);
  if (st) return st;
  char *b = (char *)buf + (slot > 0 ? bytes : 0);
  out->box = (unsigned long long *)b;
  out->ys = (double *)(b + o_ys); out->vs = (double *)(b + o_vs); out->ls = (int *)(b + o_ls);
  out->cellid = (unsigned *)(b + o_cell); out->slot = (unsigned *)(b + o_slot); out->offset = (unsigned *)(b + o_cnt);
  out->two_level = two && m >= TL_MIN_M;
  out->fin = (unsigned *)(b + o_fin); out->res1 = (double *)(b + o_ty);
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
    int shift = 0;
    while (((ncell - 1) >> shift) >= TL_NB) shift++;
    const unsigned nb = (unsigned)((ncell - 1) >> shift) + 1u, nwg = (unsigned)((m + TL_CH - 1) / TL_CH), nu = (unsigned)((m + TL_P - 1) / TL_P);
    double *t_y = (double *)(b + o_ty);
    unsigned *t_c = (unsigned *)(b + o_tc), *cnt = (unsigned *)(b + o_ca), *ubase = (unsigned *)(b + o_ub);
    const size_t nc = (size_t)nb * nwg;
    hipLaunchKernelGGL(tl_coarse_hist_kernel, dim3(nwg), dim3(256), 0, ctx->stream, d_y, m, ytda, dim, g, box, shift, nb, nwg, out->cellid, cnt);
    launch_cell_scan(ctx, cnt, nc, cnt + nc + 1);
    if (dim == 2)
      hipLaunchKernelGGL(tl_coarse_scatter_kernel<2>, dim3(nwg), dim3(256), 0, ctx->stream, d_y, m, ytda, (const unsigned *)out->cellid, shift, nb,
                         nwg, (const unsigned *)cnt, t_y, t_c, out->slot);
    else if (dim == 3)
      hipLaunchKernelGGL(tl_coarse_scatter_kernel<3>, dim3(nwg), dim3(256), 0, ctx->stream, d_y, m, ytda, (const unsigned *)out->cellid, shift, nb,
                         nwg, (const unsigned *)cnt, t_y, t_c, out->slot);
    else
      hipLaunchKernelGGL(tl_coarse_scatter_kernel<1>, dim3(nwg), dim3(256), 0, ctx->stream, d_y, m, ytda, (const unsigned *)out->cellid, shift, nb,
                         nwg, (const unsigned *)cnt, t_y, t_c, out->slot);
    hipLaunchKernelGGL(tl_fine_hist_kernel, dim3(nu), dim3(256), 0, ctx->stream, (const unsigned *)t_c, m, shift, out->offset, ubase, out->fin);
    launch_cell_scan(ctx, out->offset, ncell, out->offset + ncell + 1);
    if (dim == 2)
      hipLaunchKernelGGL(tl_fine_scatter_kernel<2>, dim3(nu), dim3(256), 0, ctx->stream, (const double *)t_y, (const unsigned *)t_c, m, shift,
                         (const unsigned *)out->offset, (const unsigned *)ubase, out->fin, out->ys);
    else if (dim == 3)
      hipLaunchKernelGGL(tl_fine_scatter_kernel<3>, dim3(nu), dim3(256), 0, ctx->stream, (const double *)t_y, (const unsigned *)t_c, m, shift,
                         (const unsigned *)out->offset, (const unsigned *)ubase, out->fin, out->ys);
    else
      hipLaunchKernelGGL(tl_fine_scatter_kernel<1>, dim3(nu), dim3(256), 0, ctx->stream, (const double *)t_y, (const unsigned *)t_c, m, shift,
                         (const unsigned *)out->offset, (const unsigned *)ubase, out->fin, out->ys);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
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
  if (s->two_level) {
    hipLaunchKernelGGL(tl_unsort_local_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)s->fin, m,
                       (const double *)s->vs, (const int *)NULL, s->res1);
    hipLaunchKernelGGL(tl_unsort_final_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)s->slot, m,
                       (const double *)s->res1, d_values, d_leaf);
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
  }
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
  if (s->two_level) {
    if (d_leaf) {
      hipLaunchKernelGGL(tl_unsort_local_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)s->fin, m,
                         (const double *)s->vs, (const int *)s->ls, s->res1);
      hipLaunchKernelGGL(tl_unsort_final_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)s->slot, m,
                         (const double *)s->res1, d_values, d_leaf);
    } else {
      hipLaunchKernelGGL(tl_unsort_local_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)s->fin, m,
                         (const double *)s->vs, (const int *)NULL, s->res1);
      hipLaunchKernelGGL(tl_unsort_final_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const unsigned *)s->slot, m,
                         (const double *)s->res1, d_values, (int *)NULL);
    }
    LAUNCH_CHECK(ctx);
    return ST_SUCCESS;
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
