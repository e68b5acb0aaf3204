/*
 * check.hip -- device-side integrity checks of a Delaunay history DAG.
 * Compiled with -ffp-contract=off: the circumsphere arithmetic is the reference's, operation for
 * operation, so the verdict equals the CPU restatement's on every tree.
 *
 * Replaces, as O(leaves) and O(leaves x N) kernels (reference file:line):
 *   _check_leaf_nodes   interpolation/linear_simplex_integrity_check.c:62-119
 *        no vertex repeated in a leaf, a leaf is not its own neighbour, no neighbour repeated,
 *        and per neighbour: it is a leaf, the vertex opposite the shared face is not one of its
 *        vertices, exactly one reverse link exists, its opposite vertex is not one of ours
 *   _check_delaunay     interpolation/linear_simplex_integrity_check.c:134-160
 *        no data point strictly inside the circumsphere of any leaf, tolerance
 *        r^2 (1 - GSL_SQRT_DBL_EPSILON); circumsphere per calculate_hypersphere_points
 *        (interpolation/linear_simplex.c:555-605: 2x2 LU with partial pivoting in standardised
 *        coordinates).  A leaf's own vertices are not tested against its sphere and a leaf with a
 *        singular system is skipped (the conventions of the CPU restatement in oracle/).
 * The reference runs these as a recursive DFS with an O(n) linked-list membership test after EVERY
 * insertion (linear_simplex.c:489) -- O(N^3) in all, N = 200 takes 6 s.  Here every
 * leaf is checked independently: C5's 100 001 leaves x 50 000 points is one launch of a few ms.
 */
#include "common.h"
#include <math.h>

#define SQRT_DBL_EPSILON 1.4901161193847656e-08   /* gsl_machine.h:18 */

struct CheckGeom { double seed[6]; double shift[2]; double scale[2]; };

/* out[0] leaf violations, out[1] first (smallest) offending leaf + 1, out[2] delaunay violations,
   out[3] first offending leaf + 1, out[4] a violating point of that leaf + 1 (any) */
__global__ void check_init_kernel(unsigned long long *out)
{
  if (threadIdx.x < 8) out[threadIdx.x] = (threadIdx.x == 1 || threadIdx.x == 3) ? ~0ULL : 0ULL;
}

__device__ __forceinline__ bool has_vertex(const int *__restrict__ pidx, int node, int v)
{
  return pidx[3 * node] == v || pidx[3 * node + 1] == v || pidx[3 * node + 2] == v;
}

__global__ void __launch_bounds__(256)
check_leaf_kernel(int n_nodes, const int *__restrict__ type, const int *__restrict__ pidx, const int *__restrict__ links,
                  unsigned long long *__restrict__ out)
{
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node < 1 || node >= n_nodes || type[node] != 0) return;
  bool bad = false;
  int p[3], l[3];
  for (int i = 0; i < 3; i++) { p[i] = pidx[3 * node + i]; l[i] = links[3 * node + i]; }
  for (int j = 0; j < 3; j++)
    for (int k = j + 1; k < 3; k++) {
      bad |= p[k] == p[j];                                  /* :79-81 a point is repeated */
      bad |= l[k] == node || l[j] == node;                  /* :82-84 neighbour of itself */
      bad |= l[k] != 0 && l[k] == l[j];                     /* :85-88 repeated neighbour  */
    }
  for (int i = 0; i < 3 && !bad; i++) {
    const int nb = l[i];
    if (!nb) continue;
    if (nb < 0 || nb >= n_nodes || type[nb] != 0) { bad = true; break; }   /* :73 assert(LEAF(node)) on the neighbour */
    if (has_vertex(pidx, nb, p[i])) bad = true;             /* :97-101 the opposite vertex is not shared */
    int back = 0, jrev = -1;
    for (int j = 0; j < 3; j++) if (links[3 * nb + j] == node) { back++; jrev = j; }
    if (back != 1) { bad = true; break; }                   /* :102-103 reverse link (FIND) */
    if (has_vertex(pidx, node, pidx[3 * nb + jrev])) bad = true;          /* :104-108 */
    for (int j = 0; j < 3; j++)                             /* the other d vertices ARE shared (shared-face consistency) */
      if (j != i && !has_vertex(pidx, nb, p[j])) bad = true;
  }
  if (bad) {
    atomicAdd(&out[0], 1ULL);
    atomicMin(&out[1], (unsigned long long)node);
  }
}

__device__ __forceinline__ void vertex_of(int id, int n_points, const double *__restrict__ points, const CheckGeom &g,
                                          double &x, double &y)
{
  if (id < 0) { x = g.seed[2 * (-id - 1)]; y = g.seed[2 * (-id - 1) + 1]; }
  else if (id < n_points) { x = points[2 * id]; y = points[2 * id + 1]; }
  else { x = y = 0.0; }
}

/* one leaf per thread, the points staged through LDS in standardised coordinates */
#define CK_THREADS 256
#define CK_TILE 1024
__global__ void __launch_bounds__(CK_THREADS)
check_delaunay_kernel(int n_leaves, const int *__restrict__ leaf_ids, const int *__restrict__ pidx, int n_points,
                      const double *__restrict__ points, CheckGeom g, unsigned long long *__restrict__ out)
{
  __shared__ double s_p[CK_TILE][2];
  const int li = blockIdx.x * CK_THREADS + threadIdx.x;
  const int node = li < n_leaves ? leaf_ids[li] : -1;
  const double s0 = g.scale[0], s1 = g.scale[1], h0 = g.shift[0], h1 = g.shift[1];
  bool active = node >= 0;
  int v[3] = {0, 0, 0};
  double c0 = 0, c1 = 0, r2 = 0;
  if (active) {
    double px[3], py[3];
    for (int i = 0; i < 3; i++) { v[i] = pidx[3 * node + i]; vertex_of(v[i], n_points, points, g, px[i], py[i]); }
    /* linear_simplex.c:566-586 */
    double m[2][2], rhs[2];
    for (int i = 0; i < 2; i++) {
      double c = 0;
      const double pi0 = s0 * (px[i] - h0), pi10 = s0 * (px[i + 1] - h0);
      c = c + pi0 * pi0 - pi10 * pi10;
      m[i][0] = pi0 - pi10;
      const double pi1 = s1 * (py[i] - h1), pi11 = s1 * (py[i + 1] - h1);
      c = c + pi1 * pi1 - pi11 * pi11;
      m[i][1] = pi1 - pi11;
      rhs[i] = 0.5 * c;
    }
    /* lu.c:82-119 at N = 2, linear_simplex_util.h:14-26 */
    double m00 = m[0][0], m01 = m[0][1], m10 = m[1][0], m11 = m[1][1];
    const bool sw = fabs(m10) > fabs(m00);
    if (sw) { double t = m00; m00 = m10; m10 = t; t = m01; m01 = m11; m11 = t; }
    double l10 = m10, u11 = m11;
    if (m00 != 0.0) { l10 = m10 / m00; u11 = m11 - l10 * m01; }
    if (m00 == 0 || u11 == 0) active = false;               /* degenerate: no sphere (:590-591), leaf skipped */
    else {
      /* lu.c:189-197 */
      double t0 = sw ? rhs[1] : rhs[0], t1 = sw ? rhs[0] : rhs[1];
      t1 -= l10 * t0;
      t1 = t1 / u11;
      t0 -= m01 * t1;
      t0 = t0 / m00;
      c0 = t0; c1 = t1;
      /* linear_simplex.c:594-602 */
      double d0 = px[0]; d0 = d0 - h0; d0 = d0 * s0; d0 = d0 - c0;
      double d1 = py[0]; d1 = d1 - h1; d1 = d1 * s1; d1 = d1 - c1;
      double mag2 = 0;
      mag2 += d0 * d0;
      mag2 += d1 * d1;
      r2 = mag2;
    }
  }
  const double bound = r2 * (1 - SQRT_DBL_EPSILON);          /* linear_simplex_integrity_check.c:155 */
  unsigned long long nviol = 0;
  int witness = -1;
  for (int base = 0; base < n_points; base += CK_TILE) {
    const int cnt = n_points - base < CK_TILE ? n_points - base : CK_TILE;
    __syncthreads();
    for (int e = threadIdx.x; e < cnt; e += CK_THREADS) {
      s_p[e][0] = s0 * (points[2 * (base + e)] - h0);        /* :149-152 (p - shift) * scale */
      s_p[e][1] = s1 * (points[2 * (base + e) + 1] - h1);
    }
    __syncthreads();
    if (!active) continue;
    for (int e = 0; e < cnt; e++) {
      const int p = base + e;
      const double a = s_p[e][0] - c0, b = s_p[e][1] - c1;
      double d2 = 0;
      d2 += a * a;
      d2 += b * b;
      if (d2 < bound && p != v[0] && p != v[1] && p != v[2]) { nviol++; witness = p; }
    }
  }
  if (nviol) {
    atomicAdd(&out[2], nviol);
    const unsigned long long old = atomicMin(&out[3], (unsigned long long)node);
    if ((unsigned long long)node <= old) out[4] = (unsigned long long)witness + 1;   /* best effort witness */
  }
}

/* stream compaction of the leaf ids (order irrelevant): one atomic per leaf */
__global__ void __launch_bounds__(256)
collect_leaves_kernel(int n_nodes, const int *__restrict__ type, int *__restrict__ leaf_ids, unsigned *__restrict__ count)
{
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node < 1 || node >= n_nodes || type[node] != 0) return;
  leaf_ids[atomicAdd(count, 1u)] = node;
}

extern "C" int gsl_sinterp_hip_tree_check(gsl_sinterp_hip_ctx *ctx, int n_nodes, const int *d_type, const int *d_pidx,
                                          const int *d_links, int n_points, const double *d_points, const double *h_geom,
                                          int what, long long *h_leaf_violations, long long *h_delaunay_violations,
                                          int *h_first)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, n_nodes >= 1 && n_points >= 0 && h_geom != NULL, ST_EINVAL);
  REQUIRE(ctx, d_type && d_pidx && d_links && (n_points == 0 || d_points), ST_EFAULT);
  if (h_leaf_violations) *h_leaf_violations = 0;
  if (h_delaunay_violations) *h_delaunay_violations = 0;
  if (h_first) h_first[0] = h_first[1] = h_first[2] = -1;
  CheckGeom g;
  memcpy(g.seed, h_geom, 6 * sizeof(double));
  memcpy(g.shift, h_geom + 6, 2 * sizeof(double));
  memcpy(g.scale, h_geom + 8, 2 * sizeof(double));
  void *buf = NULL;
  int st = sinterp_sortbuf(ctx, 128 + (size_t)n_nodes * sizeof(int), &buf);   /* out[8] | leaf count | leaf ids */
  if (st) return st;
  unsigned long long *out = (unsigned long long *)buf;
  unsigned *count = (unsigned *)((char *)buf + 64);
  int *leaf_ids = (int *)((char *)buf + 128);
  const unsigned nblk = (unsigned)((n_nodes + 255) / 256);
  hipLaunchKernelGGL(check_init_kernel, dim3(1), dim3(64), 0, ctx->stream, out);
  HIP_OK(ctx, hipMemsetAsync(count, 0, sizeof(unsigned), ctx->stream));
  if (what & 1)
    hipLaunchKernelGGL(check_leaf_kernel, dim3(nblk), dim3(256), 0, ctx->stream, n_nodes, d_type, d_pidx, d_links, out);
  unsigned n_leaves = 0;
  if (what & 2) {
    hipLaunchKernelGGL(collect_leaves_kernel, dim3(nblk), dim3(256), 0, ctx->stream, n_nodes, d_type, leaf_ids, count);
    LAUNCH_CHECK(ctx);
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    HIP_OK(ctx, hipMemcpy(&n_leaves, count, sizeof n_leaves, hipMemcpyDeviceToHost));
    if (n_leaves && n_points)
      hipLaunchKernelGGL(check_delaunay_kernel, dim3((n_leaves + CK_THREADS - 1) / CK_THREADS), dim3(CK_THREADS), 0, ctx->stream,
                         (int)n_leaves, (const int *)leaf_ids, d_pidx, n_points, d_points, g, out);
  }
  LAUNCH_CHECK(ctx);
  unsigned long long h[8];
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  if (h_leaf_violations) *h_leaf_violations = (long long)h[0];
  if (h_delaunay_violations) *h_delaunay_violations = (long long)h[2];
  if (h_first) {
    h_first[0] = h[0] ? (int)h[1] : -1;
    h_first[1] = h[2] ? (int)h[3] : -1;
    h_first[2] = h[2] ? (int)h[4] - 1 : -1;
  }
  return ST_SUCCESS;
}
