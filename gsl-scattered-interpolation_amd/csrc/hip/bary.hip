/*
 * bary.hip -- barycentric evaluation over the host-built Delaunay history DAG.
 * Compiled with -ffp-contract=off: every fp64 *, -, / below is a separately
 * rounded IEEE operation in the reference's order, so located leaves and
 * values are bit-identical to the CPU path.
 *
 * Replaces, for M targets at once (reference file:line):
 *   find_leaf / _find_leaf        interpolation/linear_simplex.c:331-402
 *   contains_point                interpolation/linear_simplex.c:653-676
 *   calculate_bary_coords         interpolation/linear_simplex.c:607-651
 *     gsl_linalg_LU_decomp N=2    linalg/lu.c:59-124   (done once per node: tree_pack)
 *     gsl_linalg_LU_svx           linalg/lu.c:166-201, cblas/source_trsv_r.h:33-79
 *   interp_point                  interpolation/linear_simplex.c:678-711
 *
 * HBM layout: one 64-byte record per DAG node so that a containment test is a
 * single aligned 64-byte gather (the reference re-gathers three vertices and
 * refactors the 2x2 system on every visit):
 *     x0[2]   raw coordinates of the node's LAST vertex
 *     u00,u01,l10,u11   the pivoted 2x2 LU of the standardised edge matrix
 *     child[3]          links (children when internal, neighbours when leaf)
 *     meta              bits 0-1 node type, bit 2 rows swapped, bit 3 singular,
 *                       bits 4-5 number of children
 * plus a 32-byte per-node table {f(v0), f(v1), f(v2), seed mask} bound to one
 * response column.
 */
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

struct __attribute__((aligned(64))) NodeRec {
  double x0, x1;
  double u00, u01, l10, u11;
  int child[3];
  int meta;
};
static_assert(sizeof(NodeRec) == GSL_SINTERP_TREE_RECORD_BYTES, "record size");

struct __attribute__((aligned(32))) LeafRec {
  double f[3];
  int mask; /* bit i set: vertex i is a data point (not a cage seed) */
  int pad;
};
static_assert(sizeof(LeafRec) == GSL_SINTERP_TREE_LEAFTAB_BYTES, "leaf table size");

#define META_TYPE(m) ((m) & 3)
#define META_SWAPPED(m) (((m) >> 2) & 1)
#define META_SINGULAR(m) (((m) >> 3) & 1)
#define META_NCHILD(m) (((m) >> 4) & 3)

struct Geom { double seed[6]; double shift[2]; double scale[2]; };

/* ------------------------------------------------------------------------ */
__global__ void tree_pack_kernel(int n_nodes, const int *__restrict__ type, const int *__restrict__ pidx,
                                 const int *__restrict__ links, int n_points, const double *__restrict__ points,
                                 Geom g, NodeRec *__restrict__ rec)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_nodes) return;
  double v[3][2];
  for (int i = 0; i < 3; i++) {
    int id = pidx[3 * k + i];
    if (id < 0) { v[i][0] = g.seed[2 * (-id - 1)]; v[i][1] = g.seed[2 * (-id - 1) + 1]; }
    else if (id < n_points) { v[i][0] = points[2 * id]; v[i][1] = points[2 * id + 1]; }
    else { v[i][0] = v[i][1] = 0.0; }
  }
  const double s0 = g.scale[0], s1 = g.scale[1], h0 = g.shift[0], h1 = g.shift[1];
  /* linear_simplex.c:622-635 */
  const double xv0 = s0 * (v[2][0] - h0);
  const double xv1 = s1 * (v[2][1] - h1);
  double m00 = s0 * (v[0][0] - h0) - xv0;
  double m01 = s0 * (v[1][0] - h0) - xv0;
  double m10 = s1 * (v[0][1] - h1) - xv1;
  double m11 = s1 * (v[1][1] - h1) - xv1;
  /* lu.c:82-119 at N=2 */
  int swapped = fabs(m10) > fabs(m00);
  if (swapped) { double t = m00; m00 = m10; m10 = t; t = m01; m01 = m11; m11 = t; }
  double l10 = m10, u11 = m11;
  if (m00 != 0.0) { l10 = m10 / m00; u11 = m11 - l10 * m01; }
  int singular = (m00 == 0) || (u11 == 0);           /* linear_simplex_util.h:14-26 */

  const int t = type ? type[k] : 0;                   /* no type array: every node is a leaf (imported triangulation) */
  const int nchild = t == 1 ? 3 : (t == 0 ? 0 : 2);   /* linear_simplex.h:67-80 */
  NodeRec r;
  r.x0 = v[2][0]; r.x1 = v[2][1];
  r.u00 = m00; r.u01 = m01; r.l10 = l10; r.u11 = u11;
  r.child[0] = links[3 * k]; r.child[1] = links[3 * k + 1]; r.child[2] = links[3 * k + 2];
  r.meta = (t & 3) | (swapped << 2) | (singular << 3) | (nchild << 4);
  rec[k] = r;
}

__global__ void tree_bind_kernel(int n_nodes, const int *__restrict__ pidx, int n_points,
                                 const double *__restrict__ response, LeafRec *__restrict__ tab)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_nodes) return;
  LeafRec r;
  r.mask = 0; r.pad = 0;
  for (int i = 0; i < 3; i++) {
    int id = pidx[3 * k + i];
    if (id >= 0 && id < n_points) { r.f[i] = response[id]; r.mask |= 1 << i; }
    else r.f[i] = 0.0;
  }
  tab[k] = r;
}

/* ------------------------------------------------------------------------ */
/* coords <- U^-1 L^-1 P ((y - x0) * scale)   (linear_simplex.c:644-649)      */
__device__ __forceinline__ void solve_node(const NodeRec &r, double y0, double y1, double s0, double s1,
                                           double &c0, double &c1)
{
  double b0 = (y0 - r.x0) * s0;
  double b1 = (y1 - r.x1) * s1;
  const bool sw = META_SWAPPED(r.meta);
  double t0 = sw ? b1 : b0;
  double t1 = sw ? b0 : b1;
  t1 -= r.l10 * t0;
  t1 = t1 / r.u11;
  t0 -= r.u01 * t1;
  t0 = t0 / r.u00;
  c0 = t0; c1 = t1;
}

__device__ __forceinline__ bool inside_unit(double c0, double c1)
{
  double tot = 0;
  tot += c0;
  if ((c0 < 0) || (c0 > 1)) return false;
  tot += c1;
  if ((c1 < 0) || (c1 > 1)) return false;
  if ((tot < 0) || (tot > 1)) return false;
  return true;
}

__device__ __forceinline__ double violation(double c0, double c1)
{
  double worst = 0, tot = 0;
  tot += c0;
  if ((c0 < 0) && (-c0 > worst)) worst = -c0;
  else if ((c0 > 1) && (c0 - 1 > worst)) worst = c0 - 1;
  tot += c1;
  if ((c1 < 0) && (-c1 > worst)) worst = -c1;
  else if ((c1 > 1) && (c1 - 1 > worst)) worst = c1 - 1;
  if ((tot < 0) && (-tot > worst)) worst = -tot;
  else if ((tot > 1) && (tot - 1 > worst)) worst = tot - 1;
  return worst;
}

/* Division-free containment test with a certificate (round 2).  The walk spends most of its VALU time in
   the two IEEE fp64 divides of solve_node (~35 instructions each, two per test, ~60 tests per target).  A test only
   needs the DECISION inside_unit(c0, c1); the quotients themselves are needed in the final leaf alone.  Here
   the coordinates are formed with reciprocals (v_rcp_f64 + two Newton steps, the seed + refinement of the
   compiler's own fdiv expansion without its scaling / fix-up: relative error eps <= 2^-50 for any seed better than
   2^-13) from numerators computed by the very operations of the exact path, and the decision is accepted only
   when it provably equals the exact one (below 8e-15 stands for eps + a few roundings, generously):
       exact:   c1 = fl(n1 / u11),  n0 = fl(t0 - fl(u01 c1)),  c0 = fl(n0 / u00),  tot = fl(c0 + c1)
       approx:  c1a = n1 r11,       n0a = fl(t0 - fl(u01 c1a)), c0a = n0a r00,     ta  = fl(c0a + c1a)
       |c1a - c1| <= 8e-15 |c1a|,   |c0a - c0| <= 8e-15 (|u01 c1a| + |t0|) / |u00| + 8e-15 |c0a|  =: e0,
       |ta - tot| <= e0 + 8e-15 |c1a| + 3e-16 (|c0a| + |c1a|)
   so with  B = 1e-12 (1 + E0 + |c1a|),  E0 = (|u01 c1a| + |t0|) |r00|  (>= |c0a|): more than 100x those bounds,
       every value in [B, 1 - B]           => the exact test says inside,
       some value <= -B or >= 1 + B        => the exact test says outside,
   anything else (a target within ~1e-12 of an edge of this node, a sliver with a huge E0, inf / NaN from a
   denormal pivot: every comparison below is false on NaN) is UNDECIDED and the caller repeats the node with the
   exact arithmetic.  Returns +1 inside, -1 outside, 0 undecided. */
__device__ __forceinline__ double rcp_newton(double d)
{
  double y = __builtin_amdgcn_rcp(d);                /* v_rcp_f64: a ~2^-23-accurate seed */
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);                               /* second step: the bound below must not hinge on the seed's accuracy */
  return fma(y, e, y);
}

__device__ __forceinline__ int classify_fast(const NodeRec &r, double y0, double y1, double s0, double s1)
{
  const double b0 = (y0 - r.x0) * s0;
  const double b1 = (y1 - r.x1) * s1;
  const bool sw = META_SWAPPED(r.meta);
  const double t0 = sw ? b1 : b0;
  double t1 = sw ? b0 : b1;
  t1 -= r.l10 * t0;                                  /* n1: the exact path's numerator, same operations */
  const double r11 = rcp_newton(r.u11), r00 = rcp_newton(r.u00);
  const double c1a = t1 * r11;
  const double p = r.u01 * c1a;
  const double c0a = (t0 - p) * r00;
  const double E0 = (fabs(p) + fabs(t0)) * fabs(r00);
  const double B = 1e-12 * ((1.0 + E0) + fabs(c1a));
  const double ta = c0a + c1a;
  const double lo = fmin(fmin(c0a, c1a), ta), hi = fmax(fmax(c0a, c1a), ta);
  /* fmin / fmax drop NaNs: test the three values for NaN through B and E0 as well (NaN anywhere -> E0 or B NaN
     or the sums NaN); the explicit self-comparisons keep the certificate independent of that reasoning */
  const bool finite = (c0a == c0a) && (c1a == c1a) && (B == B) && (B < 1e300);
  if (!finite) return 0;
  if (lo >= B && hi <= 1.0 - B) return 1;
  if (lo <= -B || hi >= 1.0 + B) return -1;
  return 0;
}

__device__ __forceinline__ NodeRec load_rec(const NodeRec *__restrict__ rec, int k)
{
  /* four 16-byte loads of one aligned 64-byte line */
  const double2 *p = reinterpret_cast<const double2 *>(rec + k);
  double2 a = p[0], b = p[1], c = p[2];
  int4 d = *reinterpret_cast<const int4 *>(p + 3);
  NodeRec r;
  r.x0 = a.x; r.x1 = a.y; r.u00 = b.x; r.u01 = b.y; r.l10 = c.x; r.u11 = c.y;
  r.child[0] = d.x; r.child[1] = d.y; r.child[2] = d.z; r.meta = d.w;
  return r;
}

/* ------------------------------------------------------------------------ */
/* Jump table (index-exact grid locator).  The walk from the root spends its first
   levels in simplices that contain whole neighbourhoods of targets.  For every cell of a G x G grid
   over the targets' bounding box this kernel follows the walk for the CELL: it descends from the
   root while, in the reference's child order, every earlier valid child provably rejects every point
   of the (slightly enlarged) cell -- some barycentric coordinate <= -JUMP_DELTA at all four corners,
   hence (affine) everywhere in the cell -- and the next one provably accepts it (all coordinates in
   [JUMP_DELTA, 1 - JUMP_DELTA] at the corners).  By induction the reference walk of ANY target in the
   cell passes through the recorded node with the same persistent coordinates, so starting there
   changes no result.  JUMP_DELTA = 1e-7 is eight orders above the rounding of a well-conditioned 2x2 solve (and the cage, whose barycentric coordinates over the data are ~1e-4, still classifies). */
#define JUMP_CELLS_PER_NODE 90.0   /* resolution of the table built by tree_pack: see the comment there */
#define JUMP_DELTA 1e-7
#define JUMP_SLACK 1e-6

__device__ __forceinline__ double key_to_double(unsigned long long k)   /* inverse of sort.hip's dkey */
{
  unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
  return __longlong_as_double((long long)u);
}

/* +1: the enlarged cell is provably inside the node's simplex, -1: provably outside, 0: undecided.
   The certificates carry margins of 1e-7 (JUMP_DELTA) and the affinity check one of 1e-9, so the coordinates here are
   formed with the node's two reciprocals (1 ulp from solve_node's quotients) instead of ten IEEE divisions per node:
   the table only ever decides where a walk STARTS, every decision of the walk itself uses the exact arithmetic. */
__device__ __forceinline__ int classify_cell(const NodeRec &r, const double (&cx)[2], const double (&cy)[2], double s0, double s1)
{
  const bool sw = META_SWAPPED(r.meta);
  const double i11 = 1.0 / r.u11, i00 = 1.0 / r.u00;
  auto coords = [&](double y0, double y1, double &c0, double &c1) {
    const double b0 = (y0 - r.x0) * s0, b1 = (y1 - r.x1) * s1;
    double t0 = sw ? b1 : b0, t1 = sw ? b0 : b1;
    t1 = (t1 - r.l10 * t0) * i11;
    t0 = (t0 - r.u01 * t1) * i00;
    c0 = t0; c1 = t1;
  };
  bool in = true;
  double mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  double m0 = 0.0, m1 = 0.0;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      double c0, c1;
      coords(cx[a], cy[b], c0, c1);
      const double c2 = 1.0 - c0 - c1;
      in = in && (c0 >= JUMP_DELTA) && (c1 >= JUMP_DELTA) && (c2 >= JUMP_DELTA);     /* NaN -> false */
      mx[0] = fmax(mx[0], c0 == c0 ? c0 : INFINITY);
      mx[1] = fmax(mx[1], c1 == c1 ? c1 : INFINITY);
      mx[2] = fmax(mx[2], c2 == c2 ? c2 : INFINITY);
      m0 += 0.25 * c0; m1 += 0.25 * c1;
    }
  /* The argument needs the computed coordinates to be affine in the target up to an error far below
     JUMP_DELTA.  For a badly conditioned (sliver) node that is not a given: check it -- the coordinates
     of the cell centre must equal the mean of the corners' to 1e-9, else the node is left undecided. */
  {
    double c0, c1;
    coords(0.5 * (cx[0] + cx[1]), 0.5 * (cy[0] + cy[1]), c0, c1);
    if (!(fabs(c0 - m0) <= 1e-9 && fabs(c1 - m1) <= 1e-9)) return 0;
  }
  if (in) return 1;
  if (mx[0] <= -JUMP_DELTA || mx[1] <= -JUMP_DELTA || mx[2] <= -JUMP_DELTA) return -1;
  return 0;
}

__global__ void __launch_bounds__(256)
jump_build_kernel(int n_nodes, const NodeRec *__restrict__ rec, double s0, double s1, const unsigned long long *__restrict__ box,
                  int G, int *__restrict__ jump, const int *__restrict__ coarse, int Gc)
{
  const size_t cell = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= (size_t)G * G) return;
  const int ix = (int)(cell % G), iy = (int)(cell / G);
  const double lo0 = key_to_double(box[0]), hi0 = key_to_double(box[1]), lo1 = key_to_double(box[2]), hi1 = key_to_double(box[3]);
  const double w0 = (hi0 - lo0) / G, w1 = (hi1 - lo1) / G;
  const double cx[2] = {lo0 + w0 * ix - JUMP_SLACK * w0, lo0 + w0 * (ix + 1) + JUMP_SLACK * w0};
  const double cy[2] = {lo1 + w1 * iy - JUMP_SLACK * w1, lo1 + w1 * (iy + 1) + JUMP_SLACK * w1};
  int node = 0;
  if (coarse) {
    const int r = G / Gc;
    node = coarse[(size_t)(iy / r) * Gc + ix / r];
    if (!(node > 0 && node < n_nodes)) node = 0;
  }
  NodeRec cur = load_rec(rec, node);
  if (node == 0 && (!(w0 >= 0.0 && w1 >= 0.0) || META_SINGULAR(cur.meta) || classify_cell(cur, cx, cy, s0, s1) != 1)) { jump[cell] = 0; return; }
  for (int guard = 0; guard < 4096 && META_TYPE(cur.meta) != 0; guard++) {
    const int nc = META_NCHILD(cur.meta);
    int next = -1;
    NodeRec nrec = cur;
    for (int i = 0; i < nc; i++) {
      const int ch = i == 0 ? cur.child[0] : (i == 1 ? cur.child[1] : cur.child[2]);
      if (!(ch > 0 && ch < n_nodes)) continue;          /* the reference does not test these */
      const NodeRec cr = load_rec(rec, ch);
      if (META_SINGULAR(cr.meta)) continue;             /* tested without a solve: never a hit */
      const int cls = classify_cell(cr, cx, cy, s0, s1);
      if (cls == 1) { next = ch; nrec = cr; }
      if (cls != -1) break;                             /* accepted, or undecided: stop looking */
    }
    if (next < 0) break;
    node = next;
    cur = nrec;
  }
  jump[cell] = node;
}

/* ------------------------------------------------------------------------ */
/* Affine walk records (round 2).  Barycentric coordinates are affine in the target, so the DECISION of a
   containment test needs no triangular solve: per node the inverse of the standardised edge matrix (scale and row
   swap folded in) is formed once per batch from the node's LU record, and a test is
       d = y - x,   c0 ~ m00 d0 + m01 d1,   c1 ~ m10 d0 + m11 d1          (6 flops instead of ~90 instructions)
   The walk kernel below accepts such a decision only under a certificate that it equals the decision of the exact
   path (solve_node + inside_unit, the reference's arithmetic); a target that meets an uncertified test anywhere is
   queued and walked by bary_eval_kernel from the start.  Located leaves get their coordinates from solve_node, so
   the values and leaves of both kernels are the reference's, bit for bit.

   Certificate.  With u = 2^-53, (a, b) = the target components feeding (t0, t1) of solve_node, and
       Q1 = (s_b |d_b| + |l10| s_a |d_a|) / |u11|,     Q0 = (s_a |d_a| + |u01| Q1) / |u00|,    q = Q0 + Q1,
   a standard forward analysis of solve_node's ten operations (no underflow, see below) gives for its computed
   coordinates against the real-arithmetic ones c*:  |c1 - c1*| <= 4u Q1,  |c0 - c0*| <= 7u Q0,  |tot - tot*| <= 12u q;
   the matrix entries are formed with <= 5u relative error against the same absolute-value expressions, so the
   affine values obey |c1a - c1*| <= 8u Q1, |c0a - c0*| <= 8u Q0, |ta - tot*| <= 9u q.  Hence all three differ from
   the exact path's by less than 21u q, and q <= alpha (|d0| + |d1|) with the per-node constant alpha stored (as a
   float rounded up, times 2^-47 = 64u) in the record.  With B = bound (|d0| + |d1|) + 2^-800 >= 64u q:
       min(c0a, c1a, ta) >= B  and  max(...) <= 1 - B    =>  inside_unit(exact) is true,
       min(...) <= -B  or  max(...) >= 1 + B  (B < 2^400) =>  inside_unit(exact) is false,
   otherwise undecided.  Underflow / overflow: a node gets bound = +inf (never certified) when it is singular or
   when the product of max(1, |v|, 1/|v|) over its LU entries and the scales exceeds 2^200; below that, underflows
   in the exact path perturb its results by < 2^-860 (covered by the 2^-800 floor) and B < 2^400 keeps every
   intermediate below 2^650.  NaN / inf anywhere makes every comparison false: undecided. */
struct __attribute__((aligned(64))) WalkRec {
  double x0, x1;             /* NodeRec's origin */
  double m00, m01, m10, m11;
  float bound;               /* 64u alpha, rounded up; +inf: never certified */
  int child[3];              /* children the reference would test AND could hit (0: none); child[0] = -1: leaf */
};
static_assert(sizeof(WalkRec) == 64, "walk record size");
#define WALK_FLOOR 0x1p-800

__global__ void __launch_bounds__(256)
walk_pack_kernel(int n_nodes, const NodeRec *__restrict__ rec, double s0, double s1, WalkRec *__restrict__ wrec)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_nodes) return;
  const NodeRec r = rec[k];
  const bool sw = META_SWAPPED(r.meta);
  const double sa = sw ? s1 : s0, sb = sw ? s0 : s1;
  const double m1a = -(r.l10 * sa) / r.u11, m1b = sb / r.u11;
  const double m0a = (sa - r.u01 * m1a) / r.u00, m0b = -(r.u01 * m1b) / r.u00;
  const double A1a = fabs(r.l10) * sa / fabs(r.u11), A1b = sb / fabs(r.u11);
  const double A0a = (sa + fabs(r.u01) * A1a) / fabs(r.u00), A0b = fabs(r.u01) * A1b / fabs(r.u00);
  const double alpha = fmax(A0a + A1a, A0b + A1b) * (0x1p-47 * (1.0 + 0x1p-10));
  double g = 1.0;
  const double v[6] = {r.l10, r.u01, r.u11, r.u00, sa, sb};
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const double a = fabs(v[i]);
    g *= fmax(1.0, a);
    if (i >= 2) g *= fmax(1.0, 1.0 / a);
  }
  const bool ok = !META_SINGULAR(r.meta) && sa > 0.0 && sb > 0.0 && g <= 0x1p200 && alpha < 1e30 &&
                  m0a == m0a && m0b == m0b && m1a == m1a && m1b == m1b;             /* NaN in g or alpha: false */
  WalkRec w;
  w.x0 = r.x0; w.x1 = r.x1;
  w.m00 = sw ? m0b : m0a; w.m01 = sw ? m0a : m0b;       /* back to the (d0, d1) order */
  w.m10 = sw ? m1b : m1a; w.m11 = sw ? m1a : m1b;
  w.bound = ok ? fmaxf((float)alpha * (1.0f + 0x1p-20f), 1.1754944e-38f) : INFINITY;
  if (META_TYPE(r.meta) == 0) { w.child[0] = -1; w.child[1] = 0; w.child[2] = 0; }
  else {
    const int nc = META_NCHILD(r.meta);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int ch = r.child[i];
      w.child[i] = (i < nc && ch > 0 && ch < n_nodes) ? ch : 0;
    }
  }
  wrec[k] = w;
}

__device__ __forceinline__ WalkRec load_wrec(const WalkRec *__restrict__ wrec, int k)
{
  union { int4 q[4]; WalkRec r; } u;
  const int4 *p = (const int4 *)(wrec + k);
  u.q[0] = p[0]; u.q[1] = p[1]; u.q[2] = p[2]; u.q[3] = p[3];
  return u.r;
}

/* in: the exact test certainly says inside; out: certainly outside; neither: undecided.  Branch-free: the walk's
   step evaluates all three children and combines the flags as lane masks. */
__device__ __forceinline__ void classify_affine(const WalkRec &w, double y0, double y1, bool &in, bool &out)
{
  const double d0 = y0 - w.x0, d1 = y1 - w.x1;
  const double c0 = fma(w.m01, d1, w.m00 * d0);
  const double c1 = fma(w.m11, d1, w.m10 * d0);
  const double t = c0 + c1;
  const double B = (double)w.bound * (fabs(d0) + fabs(d1)) + WALK_FLOOR;
  const double lo = fmin(fmin(c0, c1), t), hi = fmax(fmax(c0, c1), t);
  in = (lo >= B) & (hi <= 1.0 - B);
  out = ((lo <= -B) | (hi >= 1.0 + B)) & (B < 0x1p400);
}

#ifdef SINTERP_DIAG_PROF
/* developer build (make prof, tools/walk_stats.py): wave-iterations and active lanes of bary_walk_kernel:
   [0] step iterations, [1] lanes stepping, [2] refills, [3] lanes refilled */
__device__ unsigned long long g_walk_stats[40];
extern "C" int gsl_sinterp_hip_debug_walk_stats(unsigned long long *out, int reset)
{
  int st = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_stats), sizeof(unsigned long long) * 40);
  if (reset) { unsigned long long z[40] = {0}; st |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_walk_stats), z, sizeof z); }
  return st;
}
#endif

/* The certified walk is three kernels over the cell-sorted targets.
   bary_start_kernel   one thread per target, streaming: jump table -> start node, certified containment in it; a
                       target whose start node is its leaf (C5: 40 %) is finished here (exact coordinates, store); the
                       others ("walkers") are packed, per workgroup of 256 targets, into that workgroup's 256-entry
                       slice of the walker list {y, node, children} (ballot ranks, no global atomic);
   bary_walk_kernel    persistent waves, a lane is a worker: the walk lengths of 64 neighbouring targets differ widely
                       (mean 6.5 steps, but the longest of 64 averages 24), so a wave that kept one target per lane
                       until all were done ran at 27 % lane utilisation.  When enough lanes are idle the wave draws
                       that many walkers (slices are handed out through one counter, one atomic per slice; one atomic
                       per refill serialised ~500k same-address atomics: 7 ms); their list entries are loaded in the
                       same burst as the step's gathers, so a refill costs no extra memory round trip.  A lane that
                       reaches its leaf records the node in the list and is idle again;
   bary_finish_kernel  one thread per list entry, streaming: exact coordinates in the located leaf (interp_point
                       recomputes them: the reference's arithmetic), values and leaves stored.
   A target that meets an uncertified test anywhere goes to the queue of the exact kernel. */
#define WALK_BATCH 16
#define WALK_SLICE 256
/* Gathers of the walk.  A lane that reads its own 64-byte record with four 16-byte loads costs the CU's L1 four
   cache-line look-ups per record, and the L1 serves about one line per clock: with 64 lanes on 64 different lines
   the step's twelve loads took ~770 clocks per wave and bounded the kernel (measured: the time did not move with
   occupancy 3..7 nor with half the VALU work).  Instead the four lanes of a quad fetch ONE record per instruction,
   16 bytes each (one line per quad), straight into LDS (global_load_lds_dwordx4, lane l -> base + 16 l), four
   instructions per child = the records of the quad's four lanes; a lane then reads its record back with four
   ds_read_b128.  The per-instruction LDS images are 1040 bytes apart so that the 16 lanes of a read pass hit 64
   different banks. */
#define WALK_IMG 1040
#define WALK_WAVE_LDS (8 * WALK_IMG)
/* Only the first two children are fetched per step (eight images per wave, one wave per workgroup: 19 waves per CU
   instead of 12 with all three -- the walk is bound by the latency of its gathers, and 85 % of the DAG's inner nodes
   are flips with two children).  A lane whose two tests both say "certainly not" at a three-child node takes the
   third child in a step of its own. */

template <int J>
__device__ __forceinline__ int quad_bcast(int v)          /* lane J of the quad -> all four */
{
  return __builtin_amdgcn_update_dpp(0, v, J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false);
}

__device__ __forceinline__ void dma_part(const WalkRec *wrec, int idx, int part, char *img)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)(wrec + idx) + part * 16),
                                   (__attribute__((address_space(3))) void *)img, 16, 0, 0);
}

__device__ __forceinline__ WalkRec lds_rec(const char *p)
{
  union { int4 q[4]; WalkRec r; } u;
  const int4 *q = (const int4 *)p;
  u.q[0] = q[0]; u.q[1] = q[1]; u.q[2] = q[2]; u.q[3] = q[3];
  return u.r;
}

/* packed & 1: values is an array of {value, leaf} pairs (16 bytes, one store here and ONE gather per target in the
   un-sort pass instead of two).  packed & 2 (two-level reorder, sort.hip): leaf_out is not a leaf array but the map
   from the position in cell order to the position in the coarse order, where the result is stored. */
__device__ __forceinline__ void store_result(double *__restrict__ values, int *__restrict__ leaf_out, size_t k, double v, int leaf,
                                             int packed)
{
  if (packed & 2) {
    k = reinterpret_cast<const unsigned *>(leaf_out)[k];
    leaf_out = NULL;
  }
  if (packed & 1) {
    *reinterpret_cast<double2 *>(values + 2 * k) = make_double2(v, __longlong_as_double((long long)leaf));
  } else {
    values[k] = v;
    if (leaf_out) leaf_out[k] = leaf;
  }
}

/* exact coordinates in the located leaf, value and leaf stored (linear_simplex.c:678-711) */
__device__ __forceinline__ void finish_target(const NodeRec *__restrict__ rec, const LeafRec *__restrict__ tab, int node, double y0,
                                              double y1, double s0, double s1, size_t k, double *__restrict__ values,
                                              int *__restrict__ leaf_out, int packed)
{
  const NodeRec cur = load_rec(rec, node);
  double c0, c1;
  solve_node(cur, y0, y1, s0, s1, c0, c1);
  const LeafRec lr = tab[node];
  double tot = 0, interp = 0;
  tot += c0;
  if (lr.mask & 1) interp += c0 * lr.f[0];
  tot += c1;
  if (lr.mask & 2) interp += c1 * lr.f[1];
  if (lr.mask & 4) interp += (1 - tot) * lr.f[2];
  store_result(values, leaf_out, k, interp, node, packed);
}

/* the walker list: slice b (entries [256 b, 256 b + count[b])) holds the walkers among targets [256 b, 256 b + 256) */
struct WalkList {
  double2 *y;        /* target */
  int4 *st;          /* in: {node, child0, child1, child2}; out: .x = located leaf, -1 if queued for the exact kernel */
  int *k;            /* position in the (sorted) target array */
  unsigned *count;   /* per slice */
};

__global__ void __launch_bounds__(WALK_SLICE)
bary_start_kernel(int n_nodes, const NodeRec *__restrict__ rec, const WalkRec *__restrict__ wrec, const LeafRec *__restrict__ tab,
                  double s0, double s1, const double *__restrict__ targets, size_t m, size_t ttda, double *__restrict__ values,
                  int *__restrict__ leaf_out, const int *__restrict__ jump, int G, const unsigned long long *__restrict__ box,
                  unsigned *__restrict__ todo_count, int *__restrict__ todo, WalkList wl, int packed)
{
  double jlo0 = 0, jlo1 = 0, jw0 = 0, jw1 = 0;
  if (jump) {
    jlo0 = key_to_double(box[0]); jlo1 = key_to_double(box[2]);
    jw0 = (key_to_double(box[1]) - jlo0) / G; jw1 = (key_to_double(box[3]) - jlo1) / G;
  }
  __shared__ unsigned s_wave[4];
  const size_t k = (size_t)blockIdx.x * WALK_SLICE + threadIdx.x;
  bool walker = false;
  double y0 = 0, y1 = 0;
  int node = 0;
  WalkRec cw;
  cw.child[0] = cw.child[1] = cw.child[2] = 0;
  if (k < m) {
    y0 = targets[k * ttda]; y1 = targets[k * ttda + 1];
    if (jump && jw0 > 0.0 && jw1 > 0.0 && y0 == y0 && y1 == y1) {       /* same start as bary_eval_kernel */
      int ix = (int)((y0 - jlo0) / jw0), iy = (int)((y1 - jlo1) / jw1);
      ix = ix < 0 ? 0 : (ix >= G ? G - 1 : ix);
      iy = iy < 0 ? 0 : (iy >= G ? G - 1 : iy);
      const bool in_cell = y0 >= jlo0 + jw0 * ix - JUMP_SLACK * jw0 && y0 <= jlo0 + jw0 * (ix + 1) + JUMP_SLACK * jw0 &&
                           y1 >= jlo1 + jw1 * iy - JUMP_SLACK * jw1 && y1 <= jlo1 + jw1 * (iy + 1) + JUMP_SLACK * jw1;
      const int start = in_cell ? jump[iy * G + ix] : 0;
      if (start > 0 && start < n_nodes) node = start;
    }
    cw = load_wrec(wrec, node);
    /* the start node (a jump-table node or the caging simplex) must certainly contain the target */
    bool in, out;
    classify_affine(cw, y0, y1, in, out);
    if (!in) todo[atomicAdd(todo_count, 1u)] = (int)k;                  /* the exact walk takes it */
    else if (cw.child[0] == -1) finish_target(rec, tab, node, y0, y1, s0, s1, k, values, leaf_out, packed);
    else walker = true;
  }
  const unsigned long long wmask = __ballot(walker);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = (unsigned)__popcll(wmask);
  __syncthreads();
  unsigned base = 0, total = 0;
  for (int w = 0; w < 4; w++) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
  if (walker) {
    const size_t p = (size_t)blockIdx.x * WALK_SLICE + base + (unsigned)__popcll(wmask & ((1ULL << lane) - 1ULL));
    wl.y[p] = make_double2(y0, y1);
    wl.st[p] = make_int4(node, cw.child[0], cw.child[1], cw.child[2]);
    wl.k[p] = (int)k;
  }
  if (threadIdx.x == 0) wl.count[blockIdx.x] = total;
}

__global__ void __launch_bounds__(64)
bary_walk_kernel(const WalkRec *__restrict__ wrec, WalkList wl, unsigned n_slices, unsigned *__restrict__ todo_count,
                 int *__restrict__ todo, unsigned long long *__restrict__ next_slice, int batch)
{
  __shared__ __attribute__((aligned(16))) char img[WALK_WAVE_LDS];
  const int lane = threadIdx.x;
  bool walking = false;
  int node = 0, ch0 = 0, ch1 = 0, ch2 = 0, guard = 0;
  unsigned p = 0;                                          /* the lane's list entry */
  double y0 = 0, y1 = 0;
  bool exhausted = false;                                  /* wave-uniform */
  unsigned cbeg = 0, cend = 0;                             /* wave-uniform: what is left of the wave's current slice */
#ifdef SINTERP_DIAG_PROF
  unsigned long long it_step = 0, it_start = 0, lanes_step = 0, lanes_start = 0;
#endif
  for (;;) {
    const unsigned long long wmask = __ballot(walking);
    if (!wmask && exhausted) break;
    /* ---- refill: list entries for the idle lanes, loaded together with the step's gathers */
    bool fresh = false;
    int4 nst = make_int4(0, 0, 0, 0);
    double2 ny = make_double2(0, 0);
    const unsigned nidle_all = 64u - (unsigned)__popcll(wmask);
    if (!exhausted && (nidle_all >= (unsigned)batch || !wmask)) {
      if (cbeg == cend) {
        unsigned long long got = 0;
        if (lane == 0) got = atomicAdd(next_slice, 1ULL);
        const unsigned slice = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)got);
        if (slice < n_slices) {
          cbeg = slice * WALK_SLICE;
          cend = cbeg + (unsigned)__builtin_amdgcn_readfirstlane((int)wl.count[slice]);
        } else {
          exhausted = true;
        }
      }
      unsigned take = cend - cbeg;
      if (take > nidle_all) take = nidle_all;
      const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(~wmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)~wmask, 0u));
      if (!walking && rank < take) {
        p = cbeg + rank;
        nst = wl.st[p];
        ny = wl.y[p];
        fresh = true;
      }
      cbeg += take;
#ifdef SINTERP_DIAG_PROF
      it_start++; lanes_start += take;
#endif
    }
    /* ---- step gathers (all 64 lanes): instruction (child c, j) fetches the child-c record of the quad's lane j */
    if (wmask) {
      const int e0 = walking ? ch0 : 0, e1 = walking ? ch1 : 0;
      const int part = lane & 3;
#define WALK_FETCH(C, E, J) { const int idx = quad_bcast<J>(E); if (idx > 0) dma_part(wrec, idx, part, img + ((C) * 4 + (J)) * WALK_IMG); }
      WALK_FETCH(0, e0, 0) WALK_FETCH(0, e0, 1) WALK_FETCH(0, e0, 2) WALK_FETCH(0, e0, 3)
      WALK_FETCH(1, e1, 0) WALK_FETCH(1, e1, 1) WALK_FETCH(1, e1, 2) WALK_FETCH(1, e1, 3)
#undef WALK_FETCH
#ifdef SINTERP_DIAG_PROF
      it_step++; lanes_step += __popcll(wmask);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    /* ---- step: the reference's order -- the first child that certainly contains the target, all earlier ones
       certainly not; anything else (also: no child certainly contains it, the reference's least-violation
       fallback) goes to the exact walk */
    if (walking) {
      const char *mine = img + (lane & 3) * WALK_IMG + (lane & ~3) * 16;
      const WalkRec w0 = lds_rec(mine), w1 = lds_rec(mine + 4 * WALK_IMG);
      bool in0, out0, in1, out1;
      classify_affine(w0, y0, y1, in0, out0);
      classify_affine(w1, y0, y1, in1, out1);
      const bool v0 = ch0 > 0, v1 = ch1 > 0;
      const bool hit0 = v0 & in0, pass0 = !v0 | out0;                 /* pass: certainly not a hit */
      const bool hit1 = pass0 & v1 & in1, pass1 = pass0 & (!v1 | out1);
      if (hit0 | hit1) {
        node = hit0 ? ch0 : ch1;
        const int n0 = hit0 ? w0.child[0] : w1.child[0];
        ch1 = hit0 ? w0.child[1] : w1.child[1];
        ch2 = hit0 ? w0.child[2] : w1.child[2];
        ch0 = n0;
        if (++guard >= 4096) { todo[atomicAdd(todo_count, 1u)] = wl.k[p]; wl.st[p].x = -1; walking = false; }
        else if (n0 == -1) { wl.st[p].x = node; walking = false; }     /* located: bary_finish_kernel takes over */
      } else if (pass1 & (ch2 > 0)) {
        ch0 = ch2; ch1 = 0; ch2 = 0;                                   /* both certainly not: the third child, alone */
      } else {
        todo[atomicAdd(todo_count, 1u)] = wl.k[p];
        wl.st[p].x = -1;
        walking = false;
      }
    }
    if (fresh) { node = nst.x; ch0 = nst.y; ch1 = nst.z; ch2 = nst.w; y0 = ny.x; y1 = ny.y; guard = 0; walking = true; }
  }
#ifdef SINTERP_DIAG_PROF
  if (lane == 0) {
    atomicAdd(&g_walk_stats[0], it_step); atomicAdd(&g_walk_stats[1], lanes_step);
    atomicAdd(&g_walk_stats[2], it_start); atomicAdd(&g_walk_stats[3], lanes_start);
  }
#endif
}

__global__ void __launch_bounds__(WALK_SLICE)
bary_finish_kernel(const NodeRec *__restrict__ rec, const LeafRec *__restrict__ tab, double s0, double s1, WalkList wl,
                   double *__restrict__ values, int *__restrict__ leaf_out, int packed)
{
  if (threadIdx.x >= wl.count[blockIdx.x]) return;
  const size_t p = (size_t)blockIdx.x * WALK_SLICE + threadIdx.x;
  const int node = wl.st[p].x;
  if (node < 0) return;                                     /* queued for the exact kernel */
  const double2 y = wl.y[p];
  finish_target(rec, tab, node, y.x, y.y, s0, s1, (size_t)wl.k[p], values, leaf_out, packed);
}

template <bool FAST>
__global__ void __launch_bounds__(256)
bary_eval_kernel(int n_nodes, const NodeRec *__restrict__ rec, const LeafRec *__restrict__ tab, double s0, double s1,
                 const double *__restrict__ targets, size_t m, size_t ttda, double *__restrict__ values,
                 int *__restrict__ leaf_out, unsigned long long *__restrict__ n_outside, const int *__restrict__ perm,
                 const int *__restrict__ jump, int G, const unsigned long long *__restrict__ box,
                 const unsigned *__restrict__ m_dev, int packed)
{
  if (m_dev) m = *m_dev;                       /* the queue bary_walk_kernel left (perm = its entries) */
  double jlo0 = 0, jlo1 = 0, jw0 = 0, jw1 = 0;
  if (jump) {
    jlo0 = key_to_double(box[0]); jlo1 = key_to_double(box[2]);
    jw0 = (key_to_double(box[1]) - jlo0) / G; jw1 = (key_to_double(box[3]) - jlo1) / G;
  }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x; slot < m; slot += stride) {
    /* cell-sorted order: the 64 lanes of a wave hold spatial neighbours and descend through
       (mostly) the same DAG nodes, so their 64-byte gathers coalesce */
    const size_t k = perm ? (size_t)perm[slot] : slot;
    const double y0 = targets[k * ttda], y1 = targets[k * ttda + 1];
    /* coords persist across tests exactly like accel->coords in the reference */
    double c0 = 0, c1 = 0;
    int node = 0;
    NodeRec cur;
    bool started = false;
    if (jump && jw0 > 0.0 && jw1 > 0.0 && y0 == y0 && y1 == y1) {
      /* the node the reference walk of every target of this grid cell passes through (jump_build_kernel);
         the target must lie in the enlarged cell that was classified */
      int ix = (int)((y0 - jlo0) / jw0), iy = (int)((y1 - jlo1) / jw1);
      ix = ix < 0 ? 0 : (ix >= G ? G - 1 : ix);
      iy = iy < 0 ? 0 : (iy >= G ? G - 1 : iy);
      const bool in_cell = y0 >= jlo0 + jw0 * ix - JUMP_SLACK * jw0 && y0 <= jlo0 + jw0 * (ix + 1) + JUMP_SLACK * jw0 &&
                           y1 >= jlo1 + jw1 * iy - JUMP_SLACK * jw1 && y1 <= jlo1 + jw1 * (iy + 1) + JUMP_SLACK * jw1;
      const int start = in_cell ? jump[iy * G + ix] : 0;
      if (start > 0 && start < n_nodes) {
        cur = load_rec(rec, start);
        solve_node(cur, y0, y1, s0, s1, c0, c1);        /* the reference's persistent coordinates at this node */
        if (inside_unit(c0, c1)) { node = start; started = true; }
      }
    }
    if (!started) {
      c0 = 0; c1 = 0;
      cur = load_rec(rec, 0);
      bool in_cage = false;
      if (!META_SINGULAR(cur.meta)) { solve_node(cur, y0, y1, s0, s1, c0, c1); in_cage = inside_unit(c0, c1); }
      if (!in_cage) {                                   /* linear_simplex.c:341-347 (q7: no abort) */
        store_result(values, leaf_out, k, __builtin_nan(""), -1, packed);
        atomicAdd(n_outside, 1ULL);
        continue;
      }
    }
    int guard = 0;
    /* (c0, c1) hold the reference's persistent coordinates (accel->coords) unless the last descent was decided by
       the division-free test; then they are the exact coordinates of `cur`, recomputed on demand */
    bool have_exact = true;
    while (META_TYPE(cur.meta) != 0 && guard++ < 4096) { /* depth is O(log N); bound the walk */
      const int nc = META_NCHILD(cur.meta);
      int best = 0, next = -1;
      double best_worst = -1;
      NodeRec nrec = cur;
      /* The walk is a chain of dependent gathers (latency bound, ~0.7 TB/s of 64-byte records): all
         children's records are requested at once, then TESTED in the reference's order -- the first
         containing child wins and later ones are not evaluated, so the persistent coordinates and
         the fallback see exactly the reference's sequence of operations. */
      const int ch0 = cur.child[0], ch1 = cur.child[1], ch2 = cur.child[2];
      const bool v0 = nc > 0 && ch0 > 0 && ch0 < n_nodes, v1 = nc > 1 && ch1 > 0 && ch1 < n_nodes,
                 v2 = nc > 2 && ch2 > 0 && ch2 < n_nodes;
      const NodeRec cr0 = load_rec(rec, v0 ? ch0 : 0), cr1 = load_rec(rec, v1 ? ch1 : 0), cr2 = load_rec(rec, v2 ? ch2 : 0);
      /* (named records, not an array: an indexed array of structs ends up in scratch memory) */
      if (FAST) {
        /* children in the reference's order; an invalid or singular child is never a hit (certainly "outside");
           the first certainly-inside child after only certainly-outside ones is the reference's choice */
        bool decided = true;
#define BARY_FAST_CHILD(I, CR, VALID, CH)                                                            \
        if ((I) < nc && next < 0 && decided) {                                                        \
          if ((VALID) && !META_SINGULAR((CR).meta)) {                                                 \
            const int cls = classify_fast(CR, y0, y1, s0, s1);                                        \
            if (cls > 0) { next = (CH); nrec = (CR); }                                                \
            else if (cls == 0) decided = false;                                                       \
          }                                                                                           \
        }
        BARY_FAST_CHILD(0, cr0, v0, ch0)
        BARY_FAST_CHILD(1, cr1, v1, ch1)
        BARY_FAST_CHILD(2, cr2, v2, ch2)
#undef BARY_FAST_CHILD
        if (decided && next >= 0) { node = next; cur = nrec; have_exact = false; continue; }
        /* undecided child, or no child certainly contains the target (the reference's least-violation fallback
           compares exact coordinates): repeat this node with the exact arithmetic */
        next = -1;
        nrec = cur;
        if (!have_exact) { solve_node(cur, y0, y1, s0, s1, c0, c1); have_exact = true; }   /* cur was a hit: not singular */
      }
#define BARY_TEST_CHILD(I, CR, VALID, CH)                                                            \
      if ((I) < nc && next < 0) {                                                                     \
        bool hit = false;                                                                             \
        if (VALID) {                                                                                  \
          if (!META_SINGULAR((CR).meta)) { solve_node(CR, y0, y1, s0, s1, c0, c1); hit = inside_unit(c0, c1); } \
          if (hit) { next = (CH); nrec = (CR); }                                                      \
        }                                                                                             \
        if (!hit) {                                                                                   \
          const double worst = violation(c0, c1);                                                     \
          if ((best_worst < 0) || (worst < best_worst)) { best_worst = worst; best = (I); }           \
        }                                                                                             \
      }
      BARY_TEST_CHILD(0, cr0, v0, ch0)
      BARY_TEST_CHILD(1, cr1, v1, ch1)
      BARY_TEST_CHILD(2, cr2, v2, ch2)
#undef BARY_TEST_CHILD
      if (next < 0) {                                   /* rounding fallback, linear_simplex.c:398-400 */
        next = best == 0 ? ch0 : (best == 1 ? ch1 : ch2);
        if (next <= 0 || next >= n_nodes) break;
        nrec = best == 0 ? cr0 : (best == 1 ? cr1 : cr2);
      }
      node = next;
      cur = nrec;
    }
    /* interp_point recomputes the coordinates in the final leaf */
    if (!META_SINGULAR(cur.meta)) solve_node(cur, y0, y1, s0, s1, c0, c1);
    const LeafRec lr = tab[node];
    double tot = 0, interp = 0;
    tot += c0;
    if (lr.mask & 1) interp += c0 * lr.f[0];
    tot += c1;
    if (lr.mask & 2) interp += c1 * lr.f[1];
    if (lr.mask & 4) interp += (1 - tot) * lr.f[2];
    store_result(values, leaf_out, k, interp, node, packed);
  }
}

/* builds the locator data of "Certified leaf walk" for the records just packed; leaves ctx->lw_rec NULL when the tree
   cannot be certified (a sub-DAG too deep / wide for the push kernel's stack, non-finite constants) */
static int lw_build(gsl_sinterp_hip_ctx *ctx, int n_nodes, const int *d_type, const int *d_pidx, const int *d_links, int n_points,
                    const double *d_points, const Geom &g, const NodeRec *d_records);

/* ------------------------------------------------------------------------ */
extern "C" int gsl_sinterp_hip_tree_pack(gsl_sinterp_hip_ctx *ctx, int n_nodes, const int *d_type, const int *d_pidx,
                                         const int *d_links, int n_points, const double *d_points,
                                         const double *h_geom, void *d_records)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  REQUIRE(ctx, n_nodes > 0 && n_points >= 0, ST_EINVAL);
  REQUIRE(ctx, d_type && d_pidx && d_links && h_geom && d_records, ST_EFAULT);
  REQUIRE(ctx, ((uintptr_t)d_records & 63) == 0, ST_EINVAL);
  Geom g;
  for (int i = 0; i < 6; i++) g.seed[i] = h_geom[i];
  g.shift[0] = h_geom[6]; g.shift[1] = h_geom[7];
  g.scale[0] = h_geom[8]; g.scale[1] = h_geom[9];
  hipLaunchKernelGGL(tree_pack_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, ctx->stream, n_nodes, d_type, d_pidx,
                     d_links, n_points, d_points, g, (NodeRec *)d_records);
  LAUNCH_CHECK(ctx);
  /* jump table over the bounding box of the data (see jump_build_kernel): a property of the packed DAG,
     built once here; evaluations of these records on this context start their walks from it */
  /* certified leaf walk: seed grid + per-leaf line lists + margin constants (see "Certified leaf walk") */
  ctx->lw_rec = NULL;
  static const bool no_lw = getenv("GSL_SINTERP_NO_LEAFWALK") && getenv("GSL_SINTERP_NO_LEAFWALK")[0] == '1';
  if (!no_lw && n_nodes >= 2048 && n_points >= 3 && d_points) {
    int st = lw_build(ctx, n_nodes, d_type, d_pidx, d_links, n_points, d_points, g, (const NodeRec *)d_records);
    if (st) return st;
  }
  ctx->jump_rec = NULL;
  static const bool no_jump = getenv("GSL_SINTERP_NO_JUMP") && getenv("GSL_SINTERP_NO_JUMP")[0] == '1';
  if (!no_jump && n_nodes >= 2048 && n_points >= 3 && d_points) {
    int G = 32;
    /* cells per node: what stops the descent of a cell is a HISTORIC edge crossing it (flips leave them all over the final
       triangles), so the share of targets that start at their leaf grows with the resolution -- measured at C5 (450 k nodes,
       per step): G = 2048: 1.94 ms, 4096 (40 cells per node, the round-2 rule): 1.67, 8192: 1.57, 16384: 1.61 (the table no
       longer stays in the Infinity Cache).  90 cells per node, at most 8192^2 (256 MB; built coarse-to-fine once per tree, 14.5 ms at C5).
       Developer knobs: GSL_SINTERP_JUMP_GMAX / GSL_SINTERP_JUMP_FACTOR */
    const int gcap = getenv("GSL_SINTERP_JUMP_GMAX") ? atoi(getenv("GSL_SINTERP_JUMP_GMAX")) : 8192;
    /* with the leaf walk in place the table only starts the exact kernel on the ~1 % of a batch the margin test leaves (and on
       batches below 4096 targets): a quarter of the cells builds in a third of the time (12.8 -> 4 ms at C5), same step time.
       (A sixteenth: 19.6 ms for the whole pack, step unchanged as well; the quarter keeps small batches closer to their leaf.)
       Also measured with the leaf walk: its steps decided with the affine walk records (6 flops a test, no divide) and only the
       final leaf solved exactly -- 1.217 vs 1.228 ms per step, bit-identical; the walk is not bound by the solves, not kept. */
    const bool lw_ok = ctx->lw_rec == d_records && ctx->d_lw_lines != NULL;
    const double gfac = getenv("GSL_SINTERP_JUMP_FACTOR") ? atof(getenv("GSL_SINTERP_JUMP_FACTOR")) : (lw_ok ? JUMP_CELLS_PER_NODE / 4.0 : (double)JUMP_CELLS_PER_NODE);
    while (G < gcap && G < 512 && (double)(2 * G) * (2 * G) <= gfac * (double)n_nodes) G *= 2;
    if (G >= 512) {                                      /* above 512: multiples of 512 (the coarser levels divide by 4) */
      const int want = (int)(sqrt(gfac * (double)n_nodes) / 512.0) * 512;
      G = want < 512 ? 512 : (want > gcap ? gcap : want);
    }
    /* levels of the hierarchical build: ..., G / 16, G / 4, G (the coarser tables live behind the final one) */
    static const bool flat = getenv("GSL_SINTERP_JUMP_FLAT") && getenv("GSL_SINTERP_JUMP_FLAT")[0] == '1';   /* developer: every cell from the root */
    int lev[8], nlev = 0;
    lev[nlev++] = G;
    if (!flat) while (nlev < 8 && lev[nlev - 1] >= 512) { lev[nlev] = lev[nlev - 1] / 4; nlev++; }
    size_t cells = 0;
    for (int i = 0; i < nlev; i++) cells += (size_t)lev[i] * lev[i];
    const size_t bytes = 64 + cells * sizeof(int);
    if (bytes > ctx->jumpt_bytes) {
      if (ctx->d_jumpt) { HIP_OK(ctx, hipStreamSynchronize(ctx->stream)); HIP_OK(ctx, hipFree(ctx->d_jumpt)); ctx->d_jumpt = NULL; ctx->jumpt_bytes = 0; }
      HIP_OK(ctx, hipMalloc(&ctx->d_jumpt, bytes));
      ctx->jumpt_bytes = bytes;
    }
    unsigned long long *d_box = (unsigned long long *)ctx->d_jumpt;
    int st = sinterp_bbox_keys(ctx, d_points, (size_t)n_points, 2, 2, d_box);
    if (st) return st;
    int *tab[8];
    tab[0] = (int *)((char *)ctx->d_jumpt + 64);
    for (int i = 1; i < nlev; i++) tab[i] = tab[i - 1] + (size_t)lev[i - 1] * lev[i - 1];
    for (int i = nlev - 1; i >= 0; i--) {                 /* coarsest first; level i starts from level i + 1 */
      const int Gi = lev[i];
      hipLaunchKernelGGL(jump_build_kernel, dim3((unsigned)(((size_t)Gi * Gi + 255) / 256)), dim3(256), 0, ctx->stream, n_nodes,
                         (const NodeRec *)d_records, g.scale[0], g.scale[1], (const unsigned long long *)d_box, Gi, tab[i],
                         i + 1 < nlev ? (const int *)tab[i + 1] : (const int *)NULL, i + 1 < nlev ? lev[i + 1] : 0);
    }
    LAUNCH_CHECK(ctx);
    ctx->jump_rec = d_records; ctx->jump_nodes = n_nodes; ctx->jump_G = G;
  }
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_tree_bind(gsl_sinterp_hip_ctx *ctx, int n_nodes, const int *d_pidx, int n_points,
                                         const double *d_response, void *d_leaftab)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  REQUIRE(ctx, n_nodes > 0 && n_points >= 0, ST_EINVAL);
  REQUIRE(ctx, d_pidx && d_leaftab && (d_response || n_points == 0), ST_EFAULT);
  REQUIRE(ctx, ((uintptr_t)d_leaftab & 31) == 0, ST_EINVAL);
  hipLaunchKernelGGL(tree_bind_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, ctx->stream, n_nodes, d_pidx,
                     n_points, d_response, (LeafRec *)d_leaftab);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

/* ======================================================================== */
/* Certified leaf walk (round 4; SURVEY.md 8(f) row 1 as written: grid seed + walk over the leaves' `links` adjacency,
   verified with the reference's containment arithmetic, the DAG walk inside an eps-band of any edge).

   The reference's find_leaf descends the history DAG with a floating-point closed test at every node
   (linear_simplex.c:352-402).  Let p be a target and L a FINAL leaf whose own test accepts p.  If every test the descent
   can meet agrees with geometry -- i.e. p keeps a distance from every edge of every DAG node that exceeds the rounding
   error of that node's test -- then at every node exactly one child contains p, the first-containing-child rule picks it,
   and the descent ends in the one leaf that contains p: L, with the persistent coordinates the test at L computes.
   Which edges can come near p?  A segment that does not meet the interior of L is at least dist(p, boundary of L) away
   from p, so only (i) L's own edges and (ii) the HISTORIC edges that cross L matter.  An edge disappears from a
   Delaunay history only through a flip (edge_flip.c:117-207): the flipped-away edge e of node A is the edge of A that
   neither child keeps, and it runs through both children -- and through whatever later replaces them.  tree_pack
   therefore pushes every such e down the sub-DAG of A's children (lw_push_kernel: a conservative segment / triangle
   clip test in barycentric coordinates with 1e-7 slack, depth-first with a small visited set) and leaves, per final
   leaf, the LINES of its three edges and of every historic edge that reaches it (Euclidean normal form in raw
   coordinates, CSR lists).  The rounding error of a node's test, in the same Euclidean terms, is at most
   alpha_N (|d0| + |d1|) h_N (alpha_N: the per-node constant of the certified walk below, 64u x the forward error
   constant; h_N the largest height of the triangle), so with K = max_N alpha_N h_N, R = max_N |x0_N - c|_1:
        | line(p) | > E(p) = 2 K (|p - c|_1 + R)   for every line of L's list     ==>     find_leaf(p) = L.
   A target that fails the margin, leaves the walk without a leaf (outside the cage, NaN, a step bound), or meets
   anything non-finite is queued for bary_eval_kernel, the reference's arithmetic at every step.  So leaves and values
   stay the reference's bit for bit; what changes is the cost: a grid seed + a few neighbour steps (one 64-byte record
   each, shared by the wave because the targets arrive in cell order) instead of a 6.5-step descent with two records per
   step.  GSL_SINTERP_NO_LEAFWALK=1 selects the certified DAG walk of round 2. */
__global__ void mesh_seed_init_kernel(int *__restrict__ seed, size_t cells);
__global__ void mesh_seed_fill_kernel(const int *__restrict__ seed_in, int *__restrict__ seed_out, int G);
#define LW_STACK 96
#define LW_SEEN 48
#define LW_DELTA 1e-7
#define LW_MAX_STEPS 192

__device__ __forceinline__ void lw_vertex(int id, const double *__restrict__ points, int n_points, const Geom &g, double &x, double &y)
{
  if (id < 0) { x = g.seed[2 * (-id - 1)]; y = g.seed[2 * (-id - 1) + 1]; }
  else if (id < n_points) { x = points[2 * id]; y = points[2 * id + 1]; }
  else { x = y = 0.0; }
}

__device__ __forceinline__ void atomic_max_nonneg(unsigned long long *dst, double v)
{
  if (!(v >= 0.0)) v = INFINITY;                        /* NaN: poison the maximum, the fast path is then never taken */
  atomicMax(dst, (unsigned long long)__double_as_longlong(v));
}

/* Per-node error constants of the floating-point containment test, in Euclidean (raw-coordinate) terms:
       | distance error of node N's test at p |  <=  kappa_N |p - x0_N|_1  <=  kappa_N |p - c|_1 + rho_N,
   kappa_N = alpha_N h_N (alpha: the constant of the certified DAG walk, 64u x the forward error constant of solve_node;
   h_N: the largest height), rho_N = kappa_N |x0_N - c|_1.  own[2 N], own[2 N + 1] = their bit patterns (non-negative
   doubles order like their bits); +inf for nodes whose test cannot be bounded; 0 for singular nodes (rejected without a solve). */
__global__ void __launch_bounds__(256)
lw_bound_kernel(int n_nodes, const NodeRec *__restrict__ rec, double s0, double s1, double c0, double c1,
                unsigned long long *__restrict__ own, unsigned long long *__restrict__ acc, double tau_k, double tau_r)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_nodes) return;
  const NodeRec r = rec[k];
  double kn = 0.0, rn = 0.0;
  if (!META_SINGULAR(r.meta)) {
    const bool sw = META_SWAPPED(r.meta);
    const double sa = sw ? s1 : s0, sb = sw ? s0 : s1;
    const double m1a = -(r.l10 * sa) / r.u11, m1b = sb / r.u11;
    const double m0a = (sa - r.u01 * m1a) / r.u00, m0b = -(r.u01 * m1b) / r.u00;
    const double A1a = fabs(r.l10) * sa / fabs(r.u11), A1b = sb / fabs(r.u11);
    const double A0a = (sa + fabs(r.u01) * A1a) / fabs(r.u00), A0b = fabs(r.u01) * A1b / fabs(r.u00);
    const double alpha = fmax(A0a + A1a, A0b + A1b) * (0x1p-47 * (1.0 + 0x1p-10));    /* as walk_pack_kernel */
    double g = 1.0;
    const double v[6] = {r.l10, r.u01, r.u11, r.u00, sa, sb};
#pragma unroll
    for (int i = 0; i < 6; i++) { const double a = fabs(v[i]); g *= fmax(1.0, a); if (i >= 2) g *= fmax(1.0, 1.0 / a); }
    const double g0 = sqrt(m0a * m0a + m0b * m0b), g1 = sqrt(m1a * m1a + m1b * m1b);
    const double g2 = sqrt((m0a + m1a) * (m0a + m1a) + (m0b + m1b) * (m0b + m1b));
    const double hmax = 1.0 / fmin(g0, fmin(g1, g2));
    const bool ok = sa > 0.0 && sb > 0.0 && g <= 0x1p200 && alpha < 1e30;
    /* the exact path's coordinates are within 12u q of the real-arithmetic ones (derivation above WalkRec); alpha carries 64u */
    kn = ok ? alpha * (13.0 / 64.0) * hmax * (1.0 + 0x1p-10) : INFINITY;
    if (!(kn >= 0.0)) kn = INFINITY;
    rn = kn * (fabs(r.x0 - c0) + fabs(r.x1 - c1)) * (1.0 + 0x1p-10);
    if (!(rn >= 0.0)) rn = INFINITY;
  }
  own[2 * (size_t)k] = (unsigned long long)__double_as_longlong(kn);
  own[2 * (size_t)k + 1] = (unsigned long long)__double_as_longlong(rn);
  /* a node's constants count for the targets that can reach its PARENT: lw_relax_kernel hands them down from there, except
     for nodes with LARGE constants under a SMALL parent (cage-connected slivers: a far vertex and two nearby hull points) --
     through flips the relaxation would carry those to almost every leaf; lw_region_kernel gives them to exactly the leaves
     that overlap the parent's triangle */
  (void)tau_k; (void)tau_r;
  acc[2 * (size_t)k] = k == 0 ? own[0] : 0ULL;          /* the root is tested for every target; everything else arrives from a parent */
  acc[2 * (size_t)k + 1] = k == 0 ? own[1] : 0ULL;
}

/* (P, child slot) pairs with large constants go by region unless the region turned out too large for the workgroup's
   visited set (relax[4 P + slot] set by lw_region_kernel: the early, long cage-connected triangles) */
__device__ __forceinline__ bool lw_by_region(const unsigned char *__restrict__ relax, int P, int slot, unsigned long long ok_,
                                             unsigned long long or_, unsigned long long tau_k, unsigned long long tau_r)
{
  if (ok_ <= tau_k && or_ <= tau_r) return false;
  return relax == NULL || relax[4 * (size_t)P + slot] == 0;
}

/* One relaxation sweep of  acc[N] = max(own[N], max over parents P of max(acc[P], own of P's children)):
   after depth-of-the-DAG sweeps acc[L] bounds the constants of every node the reference walk can TEST on any way down
   to L (the nodes on the way and their siblings).  changed: set when a value grew. */
/* In topological order (Kahn): npar[c] counts the parents of c that have not handed their constants down yet; a node whose
   count reaches zero in sweep `it` is final and hands its own down in sweep it + 1 (stamp[c] = it + 1).  Every node is
   processed once, the number of sweeps is the depth of the DAG (72 at C5) and a sweep costs little more than reading the
   stamps.  (The first version iterated "push to the children until nothing grows": 24 ms, the same nodes updated once per
   ancestor level.) */
__global__ void __launch_bounds__(256)
lw_npar_kernel(int n_nodes, const NodeRec *__restrict__ rec, int *__restrict__ npar)
{
  const int P = blockIdx.x * blockDim.x + threadIdx.x;
  if (P >= n_nodes) return;
  const NodeRec r = rec[P];
  if (META_TYPE(r.meta) == 0) return;
  const int nc = META_NCHILD(r.meta);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int c = i == 0 ? r.child[0] : (i == 1 ? r.child[1] : r.child[2]);
    if (i < nc && c > 0 && c < n_nodes) atomicAdd(&npar[c], 1);
  }
}
__global__ void __launch_bounds__(256)
lw_roots_kernel(int n_nodes, const int *__restrict__ npar, int *__restrict__ stamp)
{
  const int P = blockIdx.x * blockDim.x + threadIdx.x;
  if (P < n_nodes) stamp[P] = npar[P] == 0 ? 0 : -1;
}
__global__ void __launch_bounds__(256)
lw_relax_kernel(int n_nodes, const NodeRec *__restrict__ rec, const unsigned long long *__restrict__ own,
                unsigned long long *__restrict__ acc, unsigned *__restrict__ changed, unsigned long long tau_k, unsigned long long tau_r,
                const unsigned char *__restrict__ relax, int *__restrict__ stamp, int *__restrict__ npar, int it)
{
  const int P = blockIdx.x * blockDim.x + threadIdx.x;
  if (P >= n_nodes) return;
  if (stamp[P] != it) return;
  const NodeRec r = rec[P];
  if (META_TYPE(r.meta) == 0) return;
  const int nc = META_NCHILD(r.meta);
  unsigned long long v0 = acc[2 * (size_t)P], v1 = acc[2 * (size_t)P + 1];
  int ch[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int c = i == 0 ? r.child[0] : (i == 1 ? r.child[1] : r.child[2]);
    ch[i] = (i < nc && c > 0 && c < n_nodes) ? c : 0;
    if (ch[i]) {
      const unsigned long long o0 = own[2 * (size_t)ch[i]], o1 = own[2 * (size_t)ch[i] + 1];
      if (!lw_by_region(relax, P, i, o0, o1, tau_k, tau_r)) { v0 = max(v0, o0); v1 = max(v1, o1); }
    }
  }
  bool ready = false;
#pragma unroll
  for (int i = 0; i < 3; i++)
    if (ch[i]) {
      atomicMax(&acc[2 * (size_t)ch[i]], v0);
      atomicMax(&acc[2 * (size_t)ch[i] + 1], v1);
      /* no fence: the child reads acc in a later launch (a device-scope fence here writes the XCD's L2 back: 100 us a sweep) */
      if (atomicSub(&npar[ch[i]], 1) == 1) { stamp[ch[i]] = it + 1; ready = true; }
    }
  if (ready) atomicExch(changed, 1u);
}

struct LwGrid { double lo0, lo1, w0, w1; int G; };
__device__ __forceinline__ int lw_cell(const LwGrid &g, double y0, double y1)
{
  int ix = (int)((y0 - g.lo0) / g.w0), iy = (int)((y1 - g.lo1) / g.w1);
  ix = ix < 0 ? 0 : (ix >= g.G ? g.G - 1 : ix);
  iy = iy < 0 ? 0 : (iy >= g.G ? g.G - 1 : iy);
  return iy * g.G + ix;
}

__global__ void lw_seed_kernel(int n_nodes, const int *__restrict__ type, const int *__restrict__ pidx, int n_points,
                               const double *__restrict__ points, LwGrid g, int *__restrict__ seed)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_nodes || type[k] != 0) return;
  const int a = pidx[3 * k], b = pidx[3 * k + 1], c = pidx[3 * k + 2];
  if (a < 0 || b < 0 || c < 0 || a >= n_points || b >= n_points || c >= n_points) return;     /* cage leaves: reached by walking */
  const double cx = (points[2 * a] + points[2 * b] + points[2 * c]) / 3.0, cy = (points[2 * a + 1] + points[2 * b + 1] + points[2 * c + 1]) / 3.0;
  atomicMax(&seed[lw_cell(g, cx, cy)], k);
}

/* line through P, Q in Euclidean normal form (raw coordinates): a x + b y + c = signed distance */
__device__ __forceinline__ void lw_line(double px, double py, double qx, double qy, double *__restrict__ out)
{
  const double dx = qx - px, dy = qy - py, len = sqrt(dx * dx + dy * dy);
  const double a = dy / len, b = -dx / len;
  out[0] = a; out[1] = b; out[2] = -(a * px + b * py);
}

/* Can the segment e = (u, v) (vertex ids; P, Q their coordinates) meet the INTERIOR of the triangle of node y?  Conservative
   (in doubt: yes), except where the answer is combinatorial: e is an edge of y (no); e starts at a vertex of y and leaves
   through the outside of its angle there (no -- it touches y in that vertex only, and a segment that does not meet the
   interior of a leaf is no closer to a target than the leaf's boundary is). */
__device__ __forceinline__ bool lw_crosses(const NodeRec &r, const int *__restrict__ pid, int u, int v, double px, double py, double qx,
                                           double qy, double s0, double s1)
{
  if (META_SINGULAR(r.meta)) return true;
  double p0, p1, q0, q1;
  solve_node(r, px, py, s0, s1, p0, p1);
  solve_node(r, qx, qy, s0, s1, q0, q1);
  const double cp[3] = {p0, p1, 1.0 - p0 - p1}, cq[3] = {q0, q1, 1.0 - q0 - q1};
  int ju = -1, jv = -1;
#pragma unroll
  for (int i = 0; i < 3; i++) { if (pid[i] == u) ju = i; if (pid[i] == v) jv = i; }
  if (ju >= 0 && jv >= 0) return false;                                /* an edge of y */
  if (ju >= 0 || jv >= 0) {
    /* from vertex j towards the other endpoint o: interior iff both other coordinates of o are positive */
    const int j = ju >= 0 ? ju : jv;
    const double *co = ju >= 0 ? cq : cp;
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 3; i++) if (i != j) inside = inside && !(co[i] <= -LW_DELTA);      /* NaN: stays true */
    return inside;
  }
  double t0 = 0.0, t1 = 1.0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (!(cp[i] == cp[i] && cq[i] == cq[i])) return true;
    const double a = cp[i] + LW_DELTA, d = cq[i] - cp[i];               /* c(t) = cp + t (cq - cp) >= -delta */
    if (d == 0.0) { if (a < 0.0) return false; continue; }
    const double tc = -a / d;
    if (d > 0.0) t0 = fmax(t0, tc); else t1 = fmin(t1, tc);
  }
  return t0 <= t1 + 1e-12;
}

/* pass 0: cnt[leaf] += 1 for every (historic edge, leaf) incidence; pass 1: the lines themselves behind the leaf's header and
   own three (off[leaf] + 4 + slot).  One thread per internal node A; its flipped-away edges = edges of A that no child keeps.
   An edge whose sub-DAG is too large for the thread's stack / visited set (the long edges of the first insertions cross
   hundreds of later triangles) is deferred to lw_push_big_kernel: nothing is committed for it here. */
#define LW_LEAVES 24
__global__ void __launch_bounds__(128)
lw_push_kernel(int n_nodes, const int *__restrict__ type, const int *__restrict__ pidx, const int *__restrict__ links, int n_points,
               const double *__restrict__ points, Geom g, const NodeRec *__restrict__ rec, int pass, unsigned *__restrict__ cnt,
               const unsigned *__restrict__ off, double *__restrict__ lines, unsigned long long *__restrict__ consts,
               const unsigned long long *__restrict__ acc, const unsigned long long *__restrict__ lb, unsigned *__restrict__ big,
               unsigned big_cap)
{
  const int A = blockIdx.x * blockDim.x + threadIdx.x;
  if (A >= n_nodes) return;
  const int t = type[A];
  if (t == 0) {
    if (pass == 1) {                                    /* header {kappa, rho} + the leaf's own edges open its list */
      double vx[3], vy[3];
      for (int i = 0; i < 3; i++) lw_vertex(pidx[3 * A + i], points, n_points, g, vx[i], vy[i]);
      double *hdr = lines + 3 * (size_t)off[A];
      hdr[0] = __longlong_as_double((long long)max(acc[2 * (size_t)A], lb[2 * (size_t)A]));
      hdr[1] = __longlong_as_double((long long)max(acc[2 * (size_t)A + 1], lb[2 * (size_t)A + 1]));
      hdr[2] = 0.0;
      for (int i = 0; i < 3; i++) lw_line(vx[(i + 1) % 3], vy[(i + 1) % 3], vx[(i + 2) % 3], vy[(i + 2) % 3], lines + 3 * ((size_t)off[A] + 1 + i));
    }
    return;
  }
  const int nch = t == 1 ? 3 : 2;
  int ch[3];
  for (int i = 0; i < 3; i++) ch[i] = i < nch ? links[3 * A + i] : 0;
  /* a flip makes two nodes with the same children (linear_simplex / edge_flip: LINK(leaf, 2) = neighbor): one of them pushes */
  if (t != 1) {
    const int B = links[3 * A + 2];
    if (B > 0 && B < n_nodes && B < A && type[B] == t && links[3 * B] == ch[0] && links[3 * B + 1] == ch[1] && links[3 * B + 2] == A) return;
  }
  const int va[3] = {pidx[3 * A], pidx[3 * A + 1], pidx[3 * A + 2]};
  for (int e = 0; e < 3; e++) {
    const int u = va[(e + 1) % 3], v = va[(e + 2) % 3];
    bool kept = false;
    for (int i = 0; i < nch && !kept; i++) {
      const int c = ch[i];
      if (!(c > 0 && c < n_nodes)) continue;
      int hit = 0;
      for (int q = 0; q < 3; q++) { const int w = pidx[3 * c + q]; hit += (w == u) + (w == v); }
      kept = hit == 2;
    }
    if (kept) continue;
    double px, py, qx, qy, ln[3];
    lw_vertex(u, points, n_points, g, px, py);
    lw_vertex(v, points, n_points, g, qx, qy);
    lw_line(px, py, qx, qy, ln);
    int stack[LW_STACK], seen[LW_SEEN], found[LW_LEAVES], sp = 0, ns = 0, nf = 0;
    bool over = false;
    for (int i = 0; i < nch; i++) if (ch[i] > 0 && ch[i] < n_nodes) stack[sp++] = ch[i];
    while (sp > 0 && !over) {
      const int y = stack[--sp];
      bool dup = false;
      for (int i = 0; i < ns; i++) dup = dup || seen[i] == y;
      if (dup) continue;
      if (ns >= LW_SEEN) { over = true; break; }
      seen[ns++] = y;
      const NodeRec r = rec[y];
      if (!lw_crosses(r, pidx + 3 * (size_t)y, u, v, px, py, qx, qy, g.scale[0], g.scale[1])) continue;
      if (META_TYPE(r.meta) == 0) {
        if (nf >= LW_LEAVES) { over = true; break; }
        found[nf++] = y;
        continue;
      }
      const int nc = META_NCHILD(r.meta);
      for (int i = 0; i < nc; i++) {
        const int c = r.child[i];
        if (!(c > 0 && c < n_nodes)) continue;
        if (sp >= LW_STACK) { over = true; break; }
        stack[sp++] = c;
      }
    }
    if (over) {                                         /* the whole edge goes to the workgroup kernel */
      const unsigned slot = atomicAdd(&big[0], 1u);
      if (slot < big_cap) big[1 + slot] = (unsigned)(3 * A + e);
      else atomicExch(&consts[2], 1ULL);
      continue;
    }
    for (int i = 0; i < nf; i++) {
      const int y = found[i];
      const unsigned slot = atomicAdd(&cnt[y], 1u);
      if (pass == 1) { double *o = lines + 3 * ((size_t)off[y] + 4 + slot); o[0] = ln[0]; o[1] = ln[1]; o[2] = ln[2]; }
    }
  }
}

/* the deferred (long) edges: one workgroup per edge, level-synchronous traversal with the frontier and the visited set in LDS */
#define LWB_HASH 16384
#define LWB_Q 6144
__global__ void __launch_bounds__(256)
lw_push_big_kernel(int n_nodes, const int *__restrict__ pidx, const int *__restrict__ links, const int *__restrict__ type, int n_points,
                   const double *__restrict__ points, Geom g, const NodeRec *__restrict__ rec, int pass, unsigned *__restrict__ cnt,
                   const unsigned *__restrict__ off, double *__restrict__ lines, unsigned long long *__restrict__ consts,
                   const unsigned *__restrict__ big)
{
  __shared__ int s_hash[LWB_HASH];
  __shared__ int s_q[2][LWB_Q];
  __shared__ int s_n[2], s_fail;
  const unsigned nbig = big[0];
  for (unsigned w = blockIdx.x; w < nbig; w += gridDim.x) {
    const int A = (int)(big[1 + w] / 3u), e = (int)(big[1 + w] % 3u);
    const int u = pidx[3 * A + (e + 1) % 3], v = pidx[3 * A + (e + 2) % 3];
    double px, py, qx, qy, ln[3];
    lw_vertex(u, points, n_points, g, px, py);
    lw_vertex(v, points, n_points, g, qx, qy);
    lw_line(px, py, qx, qy, ln);
    for (int i = threadIdx.x; i < LWB_HASH; i += blockDim.x) s_hash[i] = -1;
    if (threadIdx.x == 0) {
      const int nch = type[A] == 1 ? 3 : 2;
      int n0 = 0;
      for (int i = 0; i < nch; i++) { const int c = links[3 * A + i]; if (c > 0 && c < n_nodes) s_q[0][n0++] = c; }
      s_n[0] = n0; s_n[1] = 0; s_fail = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) for (int i = 0; i < s_n[0]; i++) s_hash[((unsigned)s_q[0][i] * 2654435761u) % LWB_HASH] = s_q[0][i];   /* <= 3 distinct: collisions would only cost a revisit */
    __syncthreads();
    int cur = 0;
    while (s_n[cur] > 0 && !s_fail) {
      const int ncur = s_n[cur];
      for (int i = threadIdx.x; i < ncur; i += blockDim.x) {
        const int y = s_q[cur][i];
        const NodeRec r = rec[y];
        if (!lw_crosses(r, pidx + 3 * (size_t)y, u, v, px, py, qx, qy, g.scale[0], g.scale[1])) continue;
        if (META_TYPE(r.meta) == 0) {
          const unsigned slot = atomicAdd(&cnt[y], 1u);
          if (pass == 1) { double *o = lines + 3 * ((size_t)off[y] + 4 + slot); o[0] = ln[0]; o[1] = ln[1]; o[2] = ln[2]; }
          continue;
        }
        const int nc = META_NCHILD(r.meta);
        for (int q = 0; q < nc; q++) {
          const int c = q == 0 ? r.child[0] : (q == 1 ? r.child[1] : r.child[2]);
          if (!(c > 0 && c < n_nodes)) continue;
          /* visited set: open addressing; a node enters the next frontier exactly once */
          unsigned h = ((unsigned)c * 2654435761u) % LWB_HASH;
          bool fresh = false;
          for (int probe = 0; probe < 64; probe++) {
            const int old = atomicCAS(&s_hash[h], -1, c);
            if (old == -1) { fresh = true; break; }
            if (old == c) break;
            h = (h + 1) % LWB_HASH;
            if (probe == 63) s_fail = 1;
          }
          if (fresh) {
            const int pos = atomicAdd(&s_n[cur ^ 1], 1);
            if (pos < LWB_Q) s_q[cur ^ 1][pos] = c; else s_fail = 1;
          }
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) s_n[cur] = 0;
      cur ^= 1;
      __syncthreads();
    }
    if (s_fail && threadIdx.x == 0) atomicExch(&consts[2], 1ULL);
    __syncthreads();
  }
}

/* (P, child) pairs whose child has large constants: pairs[0] = count, then P * 4 + child slot */
__global__ void __launch_bounds__(256)
lw_badpairs_kernel(int n_nodes, const NodeRec *__restrict__ rec, const unsigned long long *__restrict__ own, unsigned long long tau_k,
                   unsigned long long tau_r, unsigned *__restrict__ pairs, unsigned cap, unsigned long long *__restrict__ consts)
{
  const int P = blockIdx.x * blockDim.x + threadIdx.x;
  if (P >= n_nodes) return;
  const NodeRec r = rec[P];
  if (META_TYPE(r.meta) == 0) return;
  const int nc = META_NCHILD(r.meta);
  for (int i = 0; i < nc; i++) {
    const int c = i == 0 ? r.child[0] : (i == 1 ? r.child[1] : r.child[2]);
    if (!(c > 0 && c < n_nodes)) continue;
    if (!lw_by_region((const unsigned char *)NULL, P, i, own[2 * (size_t)c], own[2 * (size_t)c + 1], tau_k, tau_r)) continue;
    const unsigned slot = atomicAdd(&pairs[0], 1u);
    if (slot < cap) pairs[1 + slot] = (unsigned)P * 4u + (unsigned)i;
    else atomicExch(&consts[2], 1ULL);
  }
}

/* may the closed triangles of P (vertices pv) and of node y (record r, vertex ids pid) share a point?  separating-edge test
   in barycentric coordinates, conservative */
__device__ __forceinline__ bool lw_tri_overlap(const NodeRec &rp, const double (&pvx)[3], const double (&pvy)[3], const NodeRec &ry,
                                               const double (&yvx)[3], const double (&yvy)[3], double s0, double s1)
{
  if (META_SINGULAR(rp.meta) || META_SINGULAR(ry.meta)) return true;
  for (int side = 0; side < 2; side++) {
    const NodeRec &r = side ? ry : rp;
    const double *vx = side ? pvx : yvx, *vy = side ? pvy : yvy;       /* the OTHER triangle's vertices in r's coordinates */
    double c[3][3];
    for (int k = 0; k < 3; k++) {
      double a, b;
      solve_node(r, vx[k], vy[k], s0, s1, a, b);
      if (!(a == a && b == b)) return true;
      c[k][0] = a; c[k][1] = b; c[k][2] = 1.0 - a - b;
    }
    for (int i = 0; i < 3; i++)
      if (c[0][i] < -LW_DELTA && c[1][i] < -LW_DELTA && c[2][i] < -LW_DELTA) return false;     /* edge i of r separates */
  }
  return true;
}

/* one workgroup per (P, bad child): every leaf whose triangle may overlap P's -- the region in which the reference can test
   the child -- takes the child's constants (atomicMax into lb[2 leaf], lb[2 leaf + 1]) */
__global__ void __launch_bounds__(256)
lw_region_kernel(int n_nodes, const int *__restrict__ pidx, int n_points, const double *__restrict__ points, Geom g,
                 const NodeRec *__restrict__ rec, const unsigned long long *__restrict__ own, const unsigned *__restrict__ pairs,
                 unsigned long long *__restrict__ lb, unsigned char *__restrict__ relax)
{
  __shared__ int s_hash[LWB_HASH];
  __shared__ int s_q[2][LWB_Q];
  __shared__ int s_n[2], s_fail;
  const unsigned npairs = pairs[0];
  for (unsigned w = blockIdx.x; w < npairs; w += gridDim.x) {
    const int P = (int)(pairs[1 + w] / 4u), ci = (int)(pairs[1 + w] % 4u);
    const NodeRec rp = rec[P];
    const int bad = ci == 0 ? rp.child[0] : (ci == 1 ? rp.child[1] : rp.child[2]);
    const unsigned long long bk = own[2 * (size_t)bad], br = own[2 * (size_t)bad + 1];
    double pvx[3], pvy[3];
    for (int i = 0; i < 3; i++) lw_vertex(pidx[3 * P + i], points, n_points, g, pvx[i], pvy[i]);
    for (int i = threadIdx.x; i < LWB_HASH; i += blockDim.x) s_hash[i] = -1;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int nc = META_NCHILD(rp.meta);
      int n0 = 0;
      for (int i = 0; i < nc; i++) {
        const int c = i == 0 ? rp.child[0] : (i == 1 ? rp.child[1] : rp.child[2]);
        if (c > 0 && c < n_nodes) { s_q[0][n0++] = c; s_hash[((unsigned)c * 2654435761u) % LWB_HASH] = c; }
      }
      s_n[0] = n0; s_n[1] = 0; s_fail = 0;
    }
    __syncthreads();
    int cur = 0;
    while (s_n[cur] > 0 && !s_fail) {
      const int ncur = s_n[cur];
      for (int i = threadIdx.x; i < ncur; i += blockDim.x) {
        const int y = s_q[cur][i];
        const NodeRec r = rec[y];
        double yvx[3], yvy[3];
        for (int k = 0; k < 3; k++) lw_vertex(pidx[3 * (size_t)y + k], points, n_points, g, yvx[k], yvy[k]);
        if (!lw_tri_overlap(rp, pvx, pvy, r, yvx, yvy, g.scale[0], g.scale[1])) continue;
        if (META_TYPE(r.meta) == 0) { atomicMax(&lb[2 * (size_t)y], bk); atomicMax(&lb[2 * (size_t)y + 1], br); continue; }
        const int nc = META_NCHILD(r.meta);
        for (int q = 0; q < nc; q++) {
          const int c = q == 0 ? r.child[0] : (q == 1 ? r.child[1] : r.child[2]);
          if (!(c > 0 && c < n_nodes)) continue;
          unsigned h = ((unsigned)c * 2654435761u) % LWB_HASH;
          bool fresh = false;
          for (int probe = 0; probe < 64; probe++) {
            const int old = atomicCAS(&s_hash[h], -1, c);
            if (old == -1) { fresh = true; break; }
            if (old == c) break;
            h = (h + 1) % LWB_HASH;
            if (probe == 63) s_fail = 1;
          }
          if (fresh) {
            const int pos = atomicAdd(&s_n[cur ^ 1], 1);
            if (pos < LWB_Q) s_q[cur ^ 1][pos] = c; else s_fail = 1;
          }
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) s_n[cur] = 0;
      cur ^= 1;
      __syncthreads();
    }
    if (s_fail && threadIdx.x == 0) relax[pairs[1 + w]] = 1;       /* too large a region: this pair is relaxed from P instead */
    __syncthreads();
  }
}

__global__ void lw_cnt_init_kernel(int n_nodes, const int *__restrict__ type, unsigned *__restrict__ cnt, unsigned base)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k <= n_nodes) cnt[k] = (k < n_nodes && type[k] == 0) ? base : 0u;
}
__global__ void lw_cnt_zero_kernel(int n, unsigned *__restrict__ cnt)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) cnt[k] = 0u;
}

__device__ __forceinline__ void leaf_finish(const NodeRec &cur, const LeafRec *__restrict__ tab, int t, double c0, double c1, size_t k,
                                            double *__restrict__ values, int *__restrict__ leaf_out, int packed)
{
  const LeafRec lr = tab[t];
  double tot = 0, interp = 0;                                  /* linear_simplex.c:678-711 */
  tot += c0;
  if (lr.mask & 1) interp += c0 * lr.f[0];
  tot += c1;
  if (lr.mask & 2) interp += c1 * lr.f[1];
  if (lr.mask & 4) interp += (1 - tot) * lr.f[2];
  store_result(values, leaf_out, k, interp, t, packed);
}

/* one target per lane and slice (cell order).  Returns true when the target was finished here. */
__device__ __forceinline__ bool leafwalk_target(int n_nodes, const NodeRec *__restrict__ rec, const LeafRec *__restrict__ tab,
                                                const int *__restrict__ seed, const LwGrid &g, const unsigned *__restrict__ off,
                                                const double *__restrict__ lines, double F, double c0c, double c1c, double s0, double s1,
                                                double y0, double y1, size_t k, double *__restrict__ values, int *__restrict__ leaf_out,
                                                int packed)
{
  int found = -1;
  double c0 = 0.0, c1 = 0.0;
  NodeRec cur;
  if (y0 == y0 && y1 == y1) {
    int t = seed[lw_cell(g, y0, y1)], prev = -1;
    for (int step = 0; step < LW_MAX_STEPS && t > 0 && t < n_nodes; step++) {
      cur = load_rec(rec, t);
      if (META_TYPE(cur.meta) != 0 || META_SINGULAR(cur.meta)) break;
      solve_node(cur, y0, y1, s0, s1, c0, c1);
      if (inside_unit(c0, c1)) { found = t; break; }
      const double c2 = 1.0 - (c0 + c1);
      if (!(c0 == c0 && c1 == c1 && c2 == c2)) break;
      /* cross the edge opposite the most negative coordinate that has a neighbour */
      const double v[3] = {c0, c1, c2};
      int next = -1;
      double best = 0.0;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int nb = i == 0 ? cur.child[0] : (i == 1 ? cur.child[1] : cur.child[2]);
        if (v[i] < best && nb > 0 && nb != prev) { best = v[i]; next = nb; }
      }
      if (next < 0) break;
      prev = t;
      t = next;
    }
  }
  /* The margin test: E(p) = F (kappa_L |p - c|_1 + rho_L); kappa_L, rho_L bound every test the reference can make on its way
     to this leaf.  Eight lines per trip, their loads in flight together with the header's: one line per trip is a chain of
     ~13 dependent L2 round trips (0.62 ms -> 0.56 ms at C5; the walk alone is 0.38 ms).  Past the end: the last line again.
     (Taking the wave's leaves one at a time with the list read through the scalar cache -- uniform addresses -- was slower at
     every trip count tried, 0.65-0.68 ms.) */
  bool safe = found >= 0;
  if (safe) {
    const unsigned b = off[found], e = off[found + 1];
    safe = e > b + 1;
    const size_t last = e > b ? e - 1 : b;
    double d[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const size_t ii = min((size_t)b + 1 + j, last);
      d[j] = lines[3 * ii] * y0 + lines[3 * ii + 1] * y1 + lines[3 * ii + 2];
    }
    const double E = F * (lines[3 * (size_t)b] * (fabs(y0 - c0c) + fabs(y1 - c1c)) + lines[3 * (size_t)b + 1]);
    safe = safe && E == E && E < INFINITY;
#pragma unroll
    for (int j = 0; j < 8; j++) safe = safe && fabs(d[j]) > E;                /* NaN -> false */
    for (unsigned i = b + 9; i < e && safe; i += 4) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const size_t ii = min((size_t)i + j, last);
        d[j] = lines[3 * ii] * y0 + lines[3 * ii + 1] * y1 + lines[3 * ii + 2];
      }
      safe = fabs(d[0]) > E && fabs(d[1]) > E && fabs(d[2]) > E && fabs(d[3]) > E;
    }
  }
  if (safe) leaf_finish(cur, tab, found, c0, c1, k, values, leaf_out, packed);
  return safe;
}

/* `slices` x 256 consecutive targets per workgroup; what the margin test leaves is gathered in LDS and appended to the exact
   kernel's queue (todo_count / todo) with ONE atomic per workgroup: at ~1 % queued, one atomic per queued target on the single
   counter cost more than the walk itself (0.92 ms against 0.4 ms at C5).  slices = 1 by default: with 8 slices (one atomic
   per 2048 targets) the grid is only 2.4 rounds of resident workgroups and its tail costs more than the atomics save --
   C5 step 1.231 / 1.185 / 1.158 / 1.148 ms for 8 / 4 / 2 / 1 slices (GSL_SINTERP_LW_SLICES, developer). */
#define LW_SLICES 8
__global__ void __launch_bounds__(256)
leafwalk_kernel(int n_nodes, const NodeRec *__restrict__ rec, const LeafRec *__restrict__ tab, const int *__restrict__ seed, LwGrid g,
                const unsigned *__restrict__ off, const double *__restrict__ lines, double F, double c0c, double c1c,
                double s0, double s1, const double *__restrict__ targets, size_t m, size_t ttda, double *__restrict__ values,
                int *__restrict__ leaf_out, unsigned *__restrict__ todo_count, int *__restrict__ todo, int packed, int slices)
{
  __shared__ int s_q[LW_SLICES * 256];
  __shared__ unsigned s_nq, s_base;
  if (threadIdx.x == 0) s_nq = 0;
  __syncthreads();
  for (int sl = 0; sl < slices; sl++) {
    const size_t k = ((size_t)blockIdx.x * slices + sl) * 256 + threadIdx.x;
    if (k >= m) break;
    const double y0 = targets[k * ttda], y1 = targets[k * ttda + 1];
    if (!leafwalk_target(n_nodes, rec, tab, seed, g, off, lines, F, c0c, c1c, s0, s1, y0, y1, k, values, leaf_out, packed))
      s_q[atomicAdd(&s_nq, 1u)] = (int)k;
  }
  __syncthreads();
  const unsigned nq = s_nq;
  if (nq == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(todo_count, nq);
  __syncthreads();
  for (unsigned i = threadIdx.x; i < nq; i += 256) todo[s_base + i] = s_q[i];
}

static int lw_build(gsl_sinterp_hip_ctx *ctx, int n_nodes, const int *d_type, const int *d_pidx, const int *d_links, int n_points,
                    const double *d_points, const Geom &g, const NodeRec *d_records)
{
  /* seed grid: about two leaves per cell (a triangulation of n points has ~2n leaves; n_nodes ~ 9n) */
  int Gs = 16;
  while (Gs < 2048 && (double)Gs * Gs < (double)n_nodes / 9.0) Gs *= 2;
  const size_t cells = (size_t)Gs * Gs, nrun = (size_t)n_nodes / 1024 + 4;
  const size_t a_bytes = 64 + (2 * cells) * sizeof(int) + ((size_t)n_nodes + 1 + nrun) * sizeof(unsigned);
  if (a_bytes > ctx->lw_a_bytes) {
    if (ctx->d_lw_a) { HIP_OK(ctx, hipStreamSynchronize(ctx->stream)); HIP_OK(ctx, hipFree(ctx->d_lw_a)); ctx->d_lw_a = NULL; ctx->lw_a_bytes = 0; }
    HIP_OK(ctx, hipMalloc(&ctx->d_lw_a, a_bytes));
    ctx->lw_a_bytes = a_bytes;
  }
  unsigned long long *consts = (unsigned long long *)ctx->d_lw_a;      /* [2] overflow flag, [3] low word: relaxation "changed" */
  int *seed = (int *)((char *)ctx->d_lw_a + 64), *seed_raw = seed + cells;
  unsigned *off = (unsigned *)(seed_raw + cells), *runsum = off + n_nodes + 1;
  /* scratch of the build (the context's second sort buffer): per-node constants own / acc (4 x 8 B) + fill counters */
  void *sb = NULL;
  int st = sinterp_sortbuf2(ctx, (size_t)n_nodes * 6 * sizeof(unsigned long long) + ((size_t)n_nodes + 1) * sizeof(unsigned) + 12 * (size_t)n_nodes + 64, &sb);
  if (st) return st;
  unsigned long long *own = (unsigned long long *)sb, *acc = own + 2 * (size_t)n_nodes, *lb = acc + 2 * (size_t)n_nodes;
  unsigned *fillcnt = (unsigned *)(lb + 2 * (size_t)n_nodes);
  unsigned char *relax = (unsigned char *)(fillcnt + n_nodes + 1);
  int *stamp = (int *)(relax + 4 * (size_t)n_nodes), *npar = stamp + n_nodes;     /* see lw_relax_kernel */
  const unsigned big_cap = (unsigned)n_nodes;            /* deferred (long) edges: [0] = count */
  void *bb = NULL;
  st = sinterp_walkbuf(ctx, ((size_t)big_cap + 1) * sizeof(unsigned), &bb);
  if (st) return st;
  unsigned *big = (unsigned *)bb;
  /* bounding box of the data */
  unsigned long long hbox[4];
  {
    unsigned long long *d_box = (unsigned long long *)((char *)ctx->d_lw_a + 32);     /* bytes 32..63 of the header */
    st = sinterp_bbox_keys(ctx, d_points, (size_t)n_points, 2, 2, d_box);
    if (st) return st;
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    HIP_OK(ctx, hipMemcpy(hbox, d_box, sizeof hbox, hipMemcpyDeviceToHost));
  }
  auto key = [](unsigned long long k) { unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k; double d; memcpy(&d, &u, sizeof d); return d; };
  const double lo0 = key(hbox[0]), hi0 = key(hbox[1]), lo1 = key(hbox[2]), hi1 = key(hbox[3]);
  if (!(hi0 > lo0 && hi1 > lo1)) return ST_SUCCESS;
  LwGrid lg;
  lg.lo0 = lo0; lg.lo1 = lo1; lg.w0 = (hi0 - lo0) / Gs; lg.w1 = (hi1 - lo1) / Gs; lg.G = Gs;
  const double c0 = 0.5 * (lo0 + hi0), c1 = 0.5 * (lo1 + hi1);
  HIP_OK(ctx, hipMemsetAsync(consts, 0, 32, ctx->stream));
  const unsigned nb = (unsigned)((n_nodes + 255) / 256);
  /* constants above tau (a margin of ~1e-7 of the data's extent) stay out of the relaxation and are pushed by region */
  const double ext = (hi0 - lo0) + (hi1 - lo1), tau_r = 5e-8 * ext, tau_k = 5e-8;
  unsigned long long tk_bits, tr_bits;
  memcpy(&tk_bits, &tau_k, 8); memcpy(&tr_bits, &tau_r, 8);
  hipLaunchKernelGGL(lw_bound_kernel, dim3(nb), dim3(256), 0, ctx->stream, n_nodes, d_records, g.scale[0], g.scale[1], c0, c1, own, acc, tau_k, tau_r);
  HIP_OK(ctx, hipMemsetAsync(lb, 0, (size_t)n_nodes * 2 * sizeof(unsigned long long), ctx->stream));
  HIP_OK(ctx, hipMemsetAsync(relax, 0, 4 * (size_t)n_nodes, ctx->stream));
  HIP_OK(ctx, hipMemsetAsync(npar, 0, sizeof(int) * (size_t)n_nodes, ctx->stream));
  hipLaunchKernelGGL(lw_npar_kernel, dim3(nb), dim3(256), 0, ctx->stream, n_nodes, d_records, npar);
  hipLaunchKernelGGL(lw_roots_kernel, dim3(nb), dim3(256), 0, ctx->stream, n_nodes, (const int *)npar, stamp);
  /* the large constants, by region */
  HIP_OK(ctx, hipMemsetAsync(big, 0, sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(lw_badpairs_kernel, dim3(nb), dim3(256), 0, ctx->stream, n_nodes, d_records, (const unsigned long long *)own, tk_bits, tr_bits,
                     big, big_cap, consts);
  hipLaunchKernelGGL(lw_region_kernel, dim3(512), dim3(256), 0, ctx->stream, n_nodes, d_pidx, n_points, d_points, g, d_records,
                     (const unsigned long long *)own, (const unsigned *)big, lb, relax);
  if (getenv("GSL_SINTERP_LW_DEBUG") && getenv("GSL_SINTERP_LW_DEBUG")[0] == '1') {
    unsigned np_ = 0;
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipMemcpy(&np_, big, sizeof np_, hipMemcpyDeviceToHost);
    fprintf(stderr, "leaf walk: %u (parent, child) pairs with large constants pushed by region\n", np_);
  }
  /* the constants handed down the DAG, one level of its depth per sweep (lw_relax_kernel) */
  unsigned *changed = (unsigned *)&consts[3];
  for (int it = 0; it < 4096; it += 8) {
    HIP_OK(ctx, hipMemsetAsync(changed, 0, sizeof(unsigned), ctx->stream));
    for (int q = 0; q < 8; q++)
      hipLaunchKernelGGL(lw_relax_kernel, dim3(nb), dim3(256), 0, ctx->stream, n_nodes, d_records, (const unsigned long long *)own, acc, changed, tk_bits, tr_bits, (const unsigned char *)relax, stamp, npar, it + q);
    unsigned hch = 0;
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    HIP_OK(ctx, hipMemcpy(&hch, changed, sizeof hch, hipMemcpyDeviceToHost));
    if (!hch) break;
    if (it + 8 >= 4096) return ST_SUCCESS;              /* did not settle: no fast path */
  }
  hipLaunchKernelGGL(mesh_seed_init_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream, seed_raw, cells);
  hipLaunchKernelGGL(lw_seed_kernel, dim3(nb), dim3(256), 0, ctx->stream, n_nodes, d_type, d_pidx, n_points, d_points, lg, seed_raw);
  hipLaunchKernelGGL(mesh_seed_fill_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream, (const int *)seed_raw, seed, Gs);
  /* pass 0: list lengths (header + 3 own edges per leaf + the historic edges that reach it), scan, pass 1: the lines */
  hipLaunchKernelGGL(lw_cnt_init_kernel, dim3((unsigned)((n_nodes + 256) / 256)), dim3(256), 0, ctx->stream, n_nodes, d_type, off, 4u);
  const unsigned pb = (unsigned)((n_nodes + 127) / 128);
  HIP_OK(ctx, hipMemsetAsync(big, 0, sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(lw_push_kernel, dim3(pb), dim3(128), 0, ctx->stream, n_nodes, d_type, d_pidx, d_links, n_points, d_points, g, d_records,
                     0, off, (const unsigned *)NULL, (double *)NULL, consts, (const unsigned long long *)acc, (const unsigned long long *)lb, big, big_cap);
  hipLaunchKernelGGL(lw_push_big_kernel, dim3(512), dim3(256), 0, ctx->stream, n_nodes, d_pidx, d_links, d_type, n_points, d_points, g, d_records,
                     0, off, (const unsigned *)NULL, (double *)NULL, consts, (const unsigned *)big);
  sinterp_scan_u32(ctx, off, (size_t)n_nodes, runsum);
  LAUNCH_CHECK(ctx);
  unsigned long long hc[4] = {0, 0, 0, 0};
  unsigned total = 0;
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(hc, consts, 32, hipMemcpyDeviceToHost));
  HIP_OK(ctx, hipMemcpy(&total, off + n_nodes, sizeof total, hipMemcpyDeviceToHost));
  static const bool dbg = getenv("GSL_SINTERP_LW_DEBUG") && getenv("GSL_SINTERP_LW_DEBUG")[0] == '1';
  if (dbg) {
    unsigned nbig = 0;
    (void)hipMemcpy(&nbig, big, sizeof nbig, hipMemcpyDeviceToHost);
    fprintf(stderr, "leaf walk: nodes %d, seed grid %d, overflow %llu, list entries %u, long edges %u\n", n_nodes, Gs, hc[2], total, nbig);
  }
  if (hc[2] != 0 || total == 0) return ST_SUCCESS;      /* a sub-DAG too large for the push kernel's stack: certified DAG walk */
  const size_t l_bytes = (size_t)total * 3 * sizeof(double);
  if (l_bytes > ctx->lw_lines_bytes) {
    if (ctx->d_lw_lines) { HIP_OK(ctx, hipFree(ctx->d_lw_lines)); ctx->d_lw_lines = NULL; ctx->lw_lines_bytes = 0; }
    HIP_OK(ctx, hipMalloc(&ctx->d_lw_lines, l_bytes));
    ctx->lw_lines_bytes = l_bytes;
  }
  hipLaunchKernelGGL(lw_cnt_zero_kernel, dim3((unsigned)((n_nodes + 256) / 256)), dim3(256), 0, ctx->stream, n_nodes + 1, fillcnt);
  HIP_OK(ctx, hipMemsetAsync(big, 0, sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(lw_push_kernel, dim3(pb), dim3(128), 0, ctx->stream, n_nodes, d_type, d_pidx, d_links, n_points, d_points, g, d_records,
                     1, fillcnt, (const unsigned *)off, (double *)ctx->d_lw_lines, consts, (const unsigned long long *)acc, (const unsigned long long *)lb, big, big_cap);
  hipLaunchKernelGGL(lw_push_big_kernel, dim3(512), dim3(256), 0, ctx->stream, n_nodes, d_pidx, d_links, d_type, n_points, d_points, g, d_records,
                     1, fillcnt, (const unsigned *)off, (double *)ctx->d_lw_lines, consts, (const unsigned *)big);
  LAUNCH_CHECK(ctx);
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(hc, consts, 32, hipMemcpyDeviceToHost));
  if (hc[2] != 0) return ST_SUCCESS;
  ctx->lw_rec = d_records; ctx->lw_nodes = n_nodes; ctx->lw_Gs = Gs;
  ctx->lw_c[0] = c0; ctx->lw_c[1] = c1;
  ctx->lw_lo[0] = lo0; ctx->lw_lo[1] = lo1; ctx->lw_w[0] = lg.w0; ctx->lw_w[1] = lg.w1;
  return ST_SUCCESS;
}

/* One batch (or one chunk of a batch) on ctx->stream.  wrec != NULL: the certified walk with these records (built by
   the caller), the walker lists / queue in the section `wslot` of the walk buffer; slot: section of the sort buffer
   (-1: the whole buffer); m_cap: the largest chunk of the batch (sizes the sections); inner_side: the finish kernel
   may run on the context's side stream (only when the caller does not use that stream for another chunk). */
static int bary_eval_part(gsl_sinterp_hip_ctx *ctx, int n_nodes, const void *d_records, const void *d_leaftab, const double *h_scale,
                          const double *d_targets, size_t m, size_t ttda, double *d_values, int *d_leaf, unsigned long long *d_count,
                          const WalkRec *wrec, char *wsec, size_t m_cap, int slot, bool inner_side)
{
  static const bool no_fast = getenv("GSL_SINTERP_NO_FASTDIV") && getenv("GSL_SINTERP_NO_FASTDIV")[0] == '1';
  const bool will_sort = m >= 4096 && !(getenv("GSL_SINTERP_NO_SORT") && getenv("GSL_SINTERP_NO_SORT")[0] == '1');
  const bool use_walk = wrec != NULL;
  ctx->lw_last = 0;
  const unsigned n_slices = (unsigned)((m + WALK_SLICE - 1) / WALK_SLICE);
  unsigned *todo_count = NULL;
  int *todo = NULL;
  WalkList wl;
  memset(&wl, 0, sizeof wl);
  if (use_walk) {
    const size_t mp = ((m_cap + WALK_SLICE - 1) / WALK_SLICE) * WALK_SLICE;
    auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t o_todo = 64, o_y = o_todo + up(m_cap * sizeof(int)), o_st = o_y + up(mp * sizeof(double2)),
                 o_k = o_st + up(mp * sizeof(int4)), o_sc = o_k + up(mp * sizeof(int));
    todo_count = (unsigned *)wsec;
    todo = (int *)(wsec + o_todo);
    wl.y = (double2 *)(wsec + o_y); wl.st = (int4 *)(wsec + o_st); wl.k = (int *)(wsec + o_k); wl.count = (unsigned *)(wsec + o_sc);
    HIP_OK(ctx, hipMemsetAsync(todo_count, 0, 64, ctx->stream));
  }
  static const bool no_jump = getenv("GSL_SINTERP_NO_JUMP") && getenv("GSL_SINTERP_NO_JUMP")[0] == '1';
  const bool have_table = !no_jump && ctx->jump_rec == d_records && ctx->jump_nodes == n_nodes && ctx->d_jumpt;
  sinterp_sorted srt;
  bool sorted = false;
  if (will_sort) {
    /* bin by the data's bounding box when tree_pack kept one for these records: saves the pass over the targets */
    const int per_cell = getenv("GSL_SINTERP_BARY_PER_CELL") ? atoi(getenv("GSL_SINTERP_BARY_PER_CELL")) : 64;   /* developer */
    int st = sinterp_sort_reorder(ctx, d_targets, m, ttda, 2, per_cell, &srt, m_cap, slot,
                                  have_table ? (const unsigned long long *)ctx->d_jumpt : (const unsigned long long *)NULL);
    if (st) return st;
    sorted = true;
  }
  int *d_jump = NULL;
  int G = 0;
  const unsigned long long *d_jbox = NULL;
  if (have_table) {
    /* the table tree_pack built for these records on this context */
    d_jump = (int *)((char *)ctx->d_jumpt + 64);
    d_jbox = (const unsigned long long *)ctx->d_jumpt;
    G = ctx->jump_G;
  } else if (sorted && !no_jump && n_nodes >= 2048 && slot < 0) {
    /* records packed elsewhere (e.g. received by broadcast): a table over THIS batch's bounding box */
    /* grid fine enough that a cell is about the size of the final triangles (~n_nodes/9 points) */
    G = 32;
    while (G < 1024 && (double)(2 * G) * (2 * G) <= (double)n_nodes) G *= 2;
    void *jb = NULL;
    int st = sinterp_sortbuf2(ctx, (size_t)G * G * sizeof(int), &jb);
    if (st) return st;
    d_jump = (int *)jb;
    hipLaunchKernelGGL(jump_build_kernel, dim3((unsigned)((G * G + 255) / 256)), dim3(256), 0, ctx->stream, n_nodes,
                       (const NodeRec *)d_records, h_scale[0], h_scale[1], (const unsigned long long *)srt.box, G, d_jump, (const int *)NULL, 0);
    d_jbox = srt.box;
  }
  size_t blocks = (m + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  const double *yt = sorted ? (const double *)srt.ys : d_targets;
  const size_t yl = sorted ? (size_t)2 : ttda;
  const bool via_map = sorted && srt.two_level;            /* results go to srt.res1 through srt.inv, see store_result */
  double *vt = sorted ? (via_map ? srt.res1 : srt.vs) : d_values;
  int *lt = sorted ? (via_map ? (int *)srt.inv : (int *)NULL) : d_leaf;
  const int packed = (sorted && d_leaf != NULL ? 1 : 0) | (via_map ? 2 : 0);   /* 1: {value, leaf} pairs */
  const int *perm = NULL;
  const unsigned *m_dev = NULL;
  bool side = false;
  if (use_walk) {
    hipLaunchKernelGGL(bary_start_kernel, dim3(n_slices), dim3(WALK_SLICE), 0, ctx->stream, n_nodes, (const NodeRec *)d_records,
                       (const WalkRec *)wrec, (const LeafRec *)d_leaftab, h_scale[0], h_scale[1], yt, m, yl, vt, lt,
                       (const int *)d_jump, G, d_jbox, todo_count, todo, wl, packed);
    /* persistent waves, as many as are resident at once, drawing slices of the walker list from a counter */
    static int s_cus[64], s_per_cu[64];   /* per device, queried once */
    const int dv = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    if (s_cus[dv] == 0) {
      int c = 0, w = 0;                        /* w: resident workgroups per CU (the LDS images bound it) */
      HIP_OK(ctx, hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, ctx->device));
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, bary_walk_kernel, 64, 0) != hipSuccess || w < 1) w = 16;
      s_per_cu[dv] = w;
      s_cus[dv] = c > 0 ? c : 256;
    }
    const int cus = s_cus[dv];
    int per_cu = s_per_cu[dv];
    int batch = WALK_BATCH;
    if (getenv("GSL_SINTERP_WALK_WGS_PER_CU")) per_cu = atoi(getenv("GSL_SINTERP_WALK_WGS_PER_CU")) > 0 ? atoi(getenv("GSL_SINTERP_WALK_WGS_PER_CU")) : per_cu;
    if (getenv("GSL_SINTERP_WALK_BATCH")) batch = atoi(getenv("GSL_SINTERP_WALK_BATCH")) > 0 ? atoi(getenv("GSL_SINTERP_WALK_BATCH")) : batch;
    size_t wblocks = (size_t)(cus > 0 ? cus : 256) * (size_t)per_cu;
    if (wblocks > n_slices) wblocks = n_slices;
    hipLaunchKernelGGL(bary_walk_kernel, dim3((unsigned)wblocks), dim3(64), 0, ctx->stream, (const WalkRec *)wrec, wl, n_slices,
                       todo_count, todo, (unsigned long long *)(todo_count + 2), batch);
    hipStream_t fs = ctx->stream;
    side = inner_side && ctx->side_stream != NULL;
    if (side) {                                  /* the finish kernel beside the exact kernel (disjoint targets) */
      HIP_OK(ctx, hipEventRecord(ctx->side_ev[2], ctx->stream));
      HIP_OK(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_ev[2], 0));
      fs = ctx->side_stream;
    }
    hipLaunchKernelGGL(bary_finish_kernel, dim3(n_slices), dim3(WALK_SLICE), 0, fs, (const NodeRec *)d_records,
                       (const LeafRec *)d_leaftab, h_scale[0], h_scale[1], wl, vt, lt, packed);
    if (side) HIP_OK(ctx, hipEventRecord(ctx->side_ev[3], ctx->side_stream));
    perm = todo;
    m_dev = todo_count;
    if (blocks > 2048) blocks = 2048;          /* the queue is normally (almost) empty */
  }
  if (no_fast)
    hipLaunchKernelGGL(bary_eval_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n_nodes,
                       (const NodeRec *)d_records, (const LeafRec *)d_leaftab, h_scale[0], h_scale[1], yt, m, yl, vt, lt, d_count,
                       perm, (const int *)d_jump, G, d_jbox, m_dev, packed);
  else
    hipLaunchKernelGGL(bary_eval_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n_nodes,
                       (const NodeRec *)d_records, (const LeafRec *)d_leaftab, h_scale[0], h_scale[1], yt, m, yl, vt, lt, d_count,
                       perm, (const int *)d_jump, G, d_jbox, m_dev, packed);
  LAUNCH_CHECK(ctx);
  if (side) HIP_OK(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_ev[3], 0));   /* join the finish kernel */
  if (sorted) {
    int st = (packed & 1) ? sinterp_unsort_packed(ctx, &srt, m, d_values, d_leaf) : sinterp_unsort(ctx, &srt, m, d_values, d_leaf);
    if (st) return st;
  }
  return ST_SUCCESS;
}

/* developer / test hook: how many targets of the LAST large batch on this context the fast locator (certified leaf walk, or the
   certified DAG walk) left to the exact kernel; *leafwalk = 1 when that batch went through the certified leaf walk */
extern "C" int gsl_sinterp_hip_bary_last_queue(gsl_sinterp_hip_ctx *ctx, unsigned *h_queued, int *h_leafwalk)
{
  REQUIRE(ctx, ctx != NULL && h_queued != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  *h_queued = 0;
  if (h_leafwalk) *h_leafwalk = ctx->lw_last;
  if (!ctx->d_walk) return ST_SUCCESS;
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(h_queued, ctx->lw_rec ? ctx->d_walk : (void *)((char *)ctx->d_walk + 0), sizeof(unsigned), hipMemcpyDeviceToHost));
  return ST_SUCCESS;
}

/* a batch of >= 4096 targets through the certified leaf walk (records packed by tree_pack on this context) */
static int bary_eval_leafwalk(gsl_sinterp_hip_ctx *ctx, int n_nodes, const void *d_records, const void *d_leaftab, const double *h_scale,
                              const double *d_targets, size_t m, size_t ttda, double *d_values, int *d_leaf, unsigned long long *d_count)
{
  static const bool no_jump = getenv("GSL_SINTERP_NO_JUMP") && getenv("GSL_SINTERP_NO_JUMP")[0] == '1';
  const bool have_table = !no_jump && ctx->jump_rec == d_records && ctx->jump_nodes == n_nodes && ctx->d_jumpt;
  sinterp_sorted srt;
  int st = sinterp_sort_reorder(ctx, d_targets, m, ttda, 2, 64, &srt, m, -1,
                                have_table ? (const unsigned long long *)ctx->d_jumpt : (const unsigned long long *)NULL);
  if (st) return st;
  void *wb = NULL;
  st = sinterp_walkbuf(ctx, 64 + m * sizeof(int), &wb);
  if (st) return st;
  unsigned *todo_count = (unsigned *)wb;
  int *todo = (int *)((char *)wb + 64);
  HIP_OK(ctx, hipMemsetAsync(todo_count, 0, 64, ctx->stream));
  const bool via_map = srt.two_level;
  double *vt = via_map ? srt.res1 : srt.vs;
  int *lt = via_map ? (int *)srt.inv : (int *)NULL;
  const int packed = (d_leaf != NULL ? 1 : 0) | (via_map ? 2 : 0);
  static const int lw_slices_env = getenv("GSL_SINTERP_LW_SLICES") ? atoi(getenv("GSL_SINTERP_LW_SLICES")) : 0;   /* developer: 1 .. 8 */
  const int lw_slices = lw_slices_env >= 1 && lw_slices_env <= LW_SLICES ? lw_slices_env : 1;
  static const double lwF = getenv("GSL_SINTERP_LW_F") ? atof(getenv("GSL_SINTERP_LW_F")) : 4.0;   /* developer */
  ctx->lw_last = 1;
  LwGrid lg;
  lg.lo0 = ctx->lw_lo[0]; lg.lo1 = ctx->lw_lo[1]; lg.w0 = ctx->lw_w[0]; lg.w1 = ctx->lw_w[1]; lg.G = ctx->lw_Gs;
  const size_t cells = (size_t)ctx->lw_Gs * ctx->lw_Gs;
  const int *seed = (const int *)((const char *)ctx->d_lw_a + 64);
  const unsigned *off = (const unsigned *)(seed + 2 * cells);
  hipLaunchKernelGGL(leafwalk_kernel, dim3((unsigned)((m + (size_t)lw_slices * 256 - 1) / ((size_t)lw_slices * 256))), dim3(256), 0, ctx->stream, n_nodes, (const NodeRec *)d_records,
                     (const LeafRec *)d_leaftab, seed, lg, off, (const double *)ctx->d_lw_lines, lwF, ctx->lw_c[0],
                     ctx->lw_c[1], h_scale[0], h_scale[1], (const double *)srt.ys, m, (size_t)2, vt, lt, todo_count, todo, packed, lw_slices);
  /* what the margin test (or the walk) left: the reference's arithmetic at every step of the DAG */
  const int *d_jump = have_table ? (const int *)((const char *)ctx->d_jumpt + 64) : (const int *)NULL;
  hipLaunchKernelGGL(bary_eval_kernel<true>, dim3(2048), dim3(256), 0, ctx->stream, n_nodes, (const NodeRec *)d_records,
                     (const LeafRec *)d_leaftab, h_scale[0], h_scale[1], (const double *)srt.ys, m, (size_t)2, vt, lt, d_count, (const int *)todo,
                     d_jump, have_table ? ctx->jump_G : 0, have_table ? (const unsigned long long *)ctx->d_jumpt : (const unsigned long long *)NULL,
                     (const unsigned *)todo_count, packed);
  LAUNCH_CHECK(ctx);
  return (packed & 1) ? sinterp_unsort_packed(ctx, &srt, m, d_values, d_leaf) : sinterp_unsort(ctx, &srt, m, d_values, d_leaf);
}

extern "C" int gsl_sinterp_hip_bary_eval(gsl_sinterp_hip_ctx *ctx, int n_nodes, const void *d_records,
                                         const void *d_leaftab, const double *h_scale, const double *d_targets,
                                         size_t m, size_t ttda, double *d_values, int *d_leaf,
                                         long long *h_n_outside)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));      /* one context per device: bind before any launch */
  REQUIRE(ctx, n_nodes > 0 && ttda >= 2, ST_EINVAL);
  REQUIRE(ctx, d_records && d_leaftab && h_scale && (m == 0 || (d_targets && d_values)), ST_EFAULT);
  if (h_n_outside) *h_n_outside = 0;
  if (m == 0) return ST_SUCCESS;
  unsigned long long *d_count = (unsigned long long *)ctx->d_scratch;
  HIP_OK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
  /* m >= 4096: the targets are gathered into grid-cell order, swept contiguously, and the results
     un-sorted afterwards (see sinterp_sort_reorder) */
  static const bool no_fast = getenv("GSL_SINTERP_NO_FASTDIV") && getenv("GSL_SINTERP_NO_FASTDIV")[0] == '1';
  static const bool no_affine = getenv("GSL_SINTERP_NO_AFFINE_WALK") && getenv("GSL_SINTERP_NO_AFFINE_WALK")[0] == '1';
  static const bool no_side = getenv("GSL_SINTERP_NO_SIDE_STREAM") && getenv("GSL_SINTERP_NO_SIDE_STREAM")[0] == '1';
  const bool will_sort = m >= 4096 && !(getenv("GSL_SINTERP_NO_SORT") && getenv("GSL_SINTERP_NO_SORT")[0] == '1');
  /* batches in cell order on records this context packed: the certified leaf walk (round 4) */
  static const bool no_lw = getenv("GSL_SINTERP_NO_LEAFWALK") && getenv("GSL_SINTERP_NO_LEAFWALK")[0] == '1';
  if (!no_lw && !no_fast && will_sort && m < 0x7fffffffULL && ctx->lw_rec == d_records && ctx->lw_nodes == n_nodes && ctx->d_lw_lines) {
    int st = bary_eval_leafwalk(ctx, n_nodes, d_records, d_leaftab, h_scale, d_targets, m, ttda, d_values, d_leaf, d_count);
    if (st) return st;
    if (h_n_outside) {
      unsigned long long cnt = 0;
      HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
      HIP_OK(ctx, hipMemcpy(&cnt, d_count, sizeof cnt, hipMemcpyDeviceToHost));
      HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
      *h_n_outside = (long long)cnt;
      if (cnt) return sinterp_fail(ctx, ST_EDOM, "bary_eval: target(s) outside the caging simplex", hipSuccess, __FILE__, __LINE__);
    }
    return ST_SUCCESS;
  }
  /* Large batches: per-batch affine walk records (a pass over the node records: ~n_nodes x 128 bytes) and the
     certified walk; what it could not certify is queued for the exact kernel.  The results do not depend on
     which kernel walked a target. */
  const bool use_walk = !no_fast && !no_affine && will_sort && m < 0x7fffffffULL && m >= (size_t)n_nodes / 8;
  if (!no_side && use_walk && !ctx->side_stream) {
    if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) ctx->side_stream = NULL;
    for (int i = 0; i < 4 && ctx->side_stream; i++)
      if (hipEventCreateWithFlags(&ctx->side_ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(ctx->side_stream); ctx->side_stream = NULL; }
  }
  const bool side = use_walk && !no_side && ctx->side_stream != NULL;
  /* (cutting a large batch into chunks on two streams was measured slower in round 3 -- DESIGN.md 6 -- and removed) */
  const int n_chunks = 1;
  const size_t m_cap = m;
  WalkRec *wrec = NULL;
  char *wsec[2] = {NULL, NULL};
  if (use_walk) {
    const size_t mp = ((m_cap + WALK_SLICE - 1) / WALK_SLICE) * WALK_SLICE;
    auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t o_rec = up((size_t)n_nodes * sizeof(WalkRec));
    const size_t sec = 64 + up(m_cap * sizeof(int)) + up(mp * sizeof(double2)) + up(mp * sizeof(int4)) + up(mp * sizeof(int)) +
                       up((mp / WALK_SLICE) * sizeof(unsigned));
    void *wb = NULL;
    int st = sinterp_walkbuf(ctx, o_rec + sec * (n_chunks > 1 ? 2 : 1), &wb);
    if (st) return st;
    wrec = (WalkRec *)wb;
    wsec[0] = (char *)wb + o_rec;
    wsec[1] = n_chunks > 1 ? wsec[0] + sec : wsec[0];
    hipStream_t ps = ctx->stream;
    if (side) {
      HIP_OK(ctx, hipEventRecord(ctx->side_ev[0], ctx->stream));          /* after whatever produced the records */
      HIP_OK(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_ev[0], 0));
      ps = ctx->side_stream;
    }
    hipLaunchKernelGGL(walk_pack_kernel, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, ps, n_nodes,
                       (const NodeRec *)d_records, h_scale[0], h_scale[1], wrec);
    LAUNCH_CHECK(ctx);
    if (side) HIP_OK(ctx, hipEventRecord(ctx->side_ev[1], ctx->side_stream));
  }
  int st = ST_SUCCESS;
  {
    /* the walk records are built on the side stream while the targets are sorted on the main one; the sort comes
       first inside bary_eval_part, so the join is placed in front of it only when there is no side stream */
    if (side) HIP_OK(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_ev[1], 0));
    st = bary_eval_part(ctx, n_nodes, d_records, d_leaftab, h_scale, d_targets, m, ttda, d_values, d_leaf, d_count, wrec, wsec[0], m_cap,
                        -1, side);
  }
  if (st) return st;
  if (h_n_outside) {
    unsigned long long cnt = 0;
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    HIP_OK(ctx, hipMemcpy(&cnt, d_count, sizeof cnt, hipMemcpyDeviceToHost));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    *h_n_outside = (long long)cnt;
    if (cnt) return sinterp_fail(ctx, ST_EDOM, "bary_eval: target(s) outside the caging simplex", hipSuccess, __FILE__, __LINE__);
  }
  return ST_SUCCESS;
}


/* ======================================================================== */
/* Imported triangulations (the reference's README:28-31 lists "import triangulations
   from QHull / CGAL" as future work).  Such a triangulation has no history DAG: only triangles and their edge
   neighbours.  Located by SEED + WALK OVER THE LEAF ADJACENCY (interpolation/linear_simplex.h:62-63: link i of a leaf =
   neighbour opposite vertex i):
     * records: the same 64-byte NodeRec as a DAG leaf (tree_pack_kernel with no type array), child[] = neighbours,
       -1 = hull edge; the same 32-byte response table (tree_bind_kernel);
     * seed: a G x G grid over the points' bounding box, cell -> a triangle whose centroid lies in it (largest index:
       atomicMax, deterministic), empty cells take the nearest filled cell of the surrounding rings;
     * walk: barycentric coordinates of the target in the current triangle by the reference's arithmetic
       (solve_node = calculate_bary_coords, linear_simplex.c:607-651); the closed containment rule of contains_point
       (:653-676) ends the walk, otherwise it crosses the edge opposite the most negative coordinate.  The walk is
       a function of (mesh, target) only -- the seed depends on the target alone -- so results do not depend on the
       batch or the shard.  A target ON an edge or vertex belongs to every triangle that contains it; the walk
       returns the first it reaches (with no reference walk to agree with, any containing triangle is the
       reference's answer under its tie rule; the interpolated value is the same to rounding).
     * hull edge in the crossing direction: outside -> index -1, value NaN (convex meshes: Delaunay output of QHull /
       CGAL).  For a mesh declared non-convex, and for a walk that does not terminate within its bound (possible in
       non-Delaunay input), the target goes to an exhaustive scan of all triangles (smallest containing index).
   Values: interp_point's arithmetic (:678-711) in the located triangle, so a mesh exported from a simplex_tree
   (same vertex order, same standardisation) returns the bits of the DAG path wherever the containing leaf is unique. */
__global__ void mesh_seed_init_kernel(int *__restrict__ seed, size_t cells)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < cells) seed[c] = -1;
}

struct MeshGrid { double lo0, lo1, w0, w1; int G; };

__device__ __forceinline__ int mesh_cell(const MeshGrid &g, double y0, double y1)
{
  int ix = (y0 == y0 && g.w0 > 0.0) ? (int)fmin(fmax((y0 - g.lo0) / g.w0, 0.0), (double)(g.G - 1)) : 0;
  int iy = (y1 == y1 && g.w1 > 0.0) ? (int)fmin(fmax((y1 - g.lo1) / g.w1, 0.0), (double)(g.G - 1)) : 0;
  return iy * g.G + ix;
}

__global__ void mesh_seed_kernel(int n_tri, const int *__restrict__ tri, int n_points, const double *__restrict__ points, MeshGrid g,
                                 int *__restrict__ seed)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tri) return;
  double c0 = 0.0, c1 = 0.0;
  for (int i = 0; i < 3; i++) {
    const int v = tri[3 * t + i];
    if (v < 0 || v >= n_points) return;
    c0 += points[2 * v]; c1 += points[2 * v + 1];
  }
  atomicMax(&seed[mesh_cell(g, c0 / 3.0, c1 / 3.0)], t);
}

#define MESH_SEED_RINGS 8
#define MESH_GAP 1e-9   /* a target within this much (standardised barycentric units) of a triangle that no triangle contains
                           under the floating-point closed test is given to the least violating triangle */
__global__ void mesh_seed_fill_kernel(const int *__restrict__ seed_in, int *__restrict__ seed_out, int G)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= G * G) return;
  int s = seed_in[c];
  const int cx = c % G, cy = c / G;
  for (int r = 1; r <= MESH_SEED_RINGS && s < 0; r++)          /* fixed scan order: deterministic */
    for (int dy = -r; dy <= r && s < 0; dy++)
      for (int dx = -r; dx <= r && s < 0; dx++) {
        if (abs(dx) != r && abs(dy) != r) continue;
        const int x = cx + dx, y = cy + dy;
        if (x < 0 || y < 0 || x >= G || y >= G) continue;
        s = seed_in[y * G + x];
      }
  seed_out[c] = s < 0 ? 0 : s;                                 /* nothing nearby: start at triangle 0 */
}

__device__ __forceinline__ void mesh_finish(const NodeRec &cur, const LeafRec *__restrict__ tab, int t, double y0, double y1, double s0,
                                            double s1, size_t k, double *__restrict__ values, int *__restrict__ tri_out, int packed)
{
  double c0, c1;
  solve_node(cur, y0, y1, s0, s1, c0, c1);
  const LeafRec lr = tab[t];
  double tot = 0, interp = 0;                                  /* linear_simplex.c:678-711 */
  tot += c0;
  if (lr.mask & 1) interp += c0 * lr.f[0];
  tot += c1;
  if (lr.mask & 2) interp += c1 * lr.f[1];
  if (lr.mask & 4) interp += (1 - tot) * lr.f[2];
  store_result(values, tri_out, k, interp, t, packed);
}

/* todo: [0] = count, [1..] = indices of the targets left to the exhaustive scan.  packed: store_result's modes (batches of
   >= 4096 targets arrive in grid-cell order like the DAG path's: neighbouring lanes then start from neighbouring seeds and
   walk through the same few triangles -- their 64-byte records are shared by the wave instead of one cache line per lane) */
__global__ void __launch_bounds__(256)
mesh_walk_kernel(int n_tri, const NodeRec *__restrict__ rec, const LeafRec *__restrict__ tab, const int *__restrict__ seed, MeshGrid g,
                 double s0, double s1, int convex, int max_steps, const double *__restrict__ targets, size_t m, size_t ttda,
                 double *__restrict__ values, int *__restrict__ tri_out, unsigned long long *__restrict__ n_outside,
                 unsigned *__restrict__ todo, int packed)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const double y0 = targets[k * ttda], y1 = targets[k * ttda + 1];
  int t = seed[mesh_cell(g, y0, y1)];
  int found = -2;                                              /* -2 walking, -1 outside, -3 exhaustive scan */
  if (!(y0 == y0 && y1 == y1)) found = -1;                     /* NaN target: outside, like the DAG path */
  NodeRec cur;
  int prev = -1;
  double prev_viol = 0.0;
  for (int step = 0; found == -2; step++) {
    if (step >= max_steps || t < 0 || t >= n_tri) { found = -3; break; }
    cur = load_rec(rec, t);
    double c0, c1;
    solve_node(cur, y0, y1, s0, s1, c0, c1);
    if (inside_unit(c0, c1)) { found = t; break; }
    const double c2 = 1.0 - (c0 + c1);
    if (!(c0 == c0 && c1 == c1 && c2 == c2) || META_SINGULAR(cur.meta)) { found = -3; break; }   /* degenerate triangle */
    /* the edge to cross: opposite the most negative coordinate; a hull edge there -> the next most negative one */
    double v[3] = {c0, c1, c2};
    int order[3] = {0, 1, 2};
    if (v[order[1]] < v[order[0]]) { const int q = order[0]; order[0] = order[1]; order[1] = q; }
    if (v[order[2]] < v[order[1]]) { const int q = order[1]; order[1] = order[2]; order[2] = q; }
    if (v[order[1]] < v[order[0]]) { const int q = order[0]; order[0] = order[1]; order[1] = q; }
    int next = -1;
    for (int a = 0; a < 3 && next < 0; a++)
      if (v[order[a]] < 0.0 && cur.child[order[a]] >= 0) next = cur.child[order[a]];
    const double viol = violation(c0, c1);
    if (next < 0) {                                            /* only hull edges in the way */
      if (viol <= MESH_GAP) found = t;                         /* ... and the target within rounding of one: it is this triangle's */
      else found = (v[order[0]] < 0.0 && convex) ? -1 : -3;
      break;
    }
    if (next == prev) {
      /* both triangles of an edge send the target across it: the target lies within rounding of the edge and the
         floating-point closed test fails on both sides (the gap the reference's walk closes with its "least violating
         child", linear_simplex.c:374-400).  Same rule: the less violating of the two, the smaller index on a tie;
         anything beyond rounding size goes to the exhaustive scan. */
      const bool take_prev = prev_viol < viol || (prev_viol == viol && prev < t);
      if ((take_prev ? prev_viol : viol) <= MESH_GAP) { found = take_prev ? prev : t; if (take_prev) cur = load_rec(rec, prev); }
      else found = -3;
      break;
    }
    prev = t; prev_viol = viol;
    t = next;
  }
  if (found >= 0) { mesh_finish(cur, tab, found, y0, y1, s0, s1, k, values, tri_out, packed); return; }
  if (found == -3) {
    const unsigned slot = atomicAdd(&todo[0], 1u);
    todo[1 + slot] = (unsigned)k;
    return;
  }
  store_result(values, tri_out, k, __longlong_as_double(0x7ff8000000000000LL), -1, packed);
  atomicAdd(n_outside, 1ULL);
}

/* exhaustive scan: one workgroup per queued target; the least violating triangle (0 = containing), smallest index on a
   tie, accepted up to MESH_GAP */
__global__ void __launch_bounds__(256)
mesh_scan_kernel(int n_tri, const NodeRec *__restrict__ rec, const LeafRec *__restrict__ tab, double s0, double s1,
                 const double *__restrict__ targets, size_t ttda, double *__restrict__ values, int *__restrict__ tri_out,
                 unsigned long long *__restrict__ n_outside, const unsigned *__restrict__ todo, int packed)
{
  __shared__ unsigned long long s_viol;
  __shared__ int s_best;
  const unsigned count = todo[0];
  for (unsigned q = blockIdx.x; q < count; q += gridDim.x) {
    const size_t k = todo[1 + q];
    const double y0 = targets[k * ttda], y1 = targets[k * ttda + 1];
    if (threadIdx.x == 0) { s_viol = ~0ULL; s_best = 0x7fffffff; }
    __syncthreads();
    double my_viol = INFINITY;
    int my_t = 0x7fffffff;
    for (int t = threadIdx.x; t < n_tri; t += blockDim.x) {
      const NodeRec r = load_rec(rec, t);
      double c0, c1;
      solve_node(r, y0, y1, s0, s1, c0, c1);
      if (META_SINGULAR(r.meta) || !(c0 == c0 && c1 == c1)) continue;
      const double viol = inside_unit(c0, c1) ? 0.0 : violation(c0, c1);
      if (viol < my_viol) { my_viol = viol; my_t = t; }          /* ascending t: the first of equal violations stays */
    }
    if (my_t != 0x7fffffff) atomicMin(&s_viol, (unsigned long long)__double_as_longlong(my_viol));   /* viol >= 0: bits ordered */
    __syncthreads();
    if (my_t != 0x7fffffff && (unsigned long long)__double_as_longlong(my_viol) == s_viol) atomicMin(&s_best, my_t);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = s_best;
      if (t != 0x7fffffff && __longlong_as_double((long long)s_viol) <= MESH_GAP) mesh_finish(load_rec(rec, t), tab, t, y0, y1, s0, s1, k, values, tri_out, packed);
      else {
        store_result(values, tri_out, k, __longlong_as_double(0x7ff8000000000000LL), -1, packed);
        atomicAdd(n_outside, 1ULL);
      }
    }
    __syncthreads();
  }
}

/* h_geom[8] = shift(2), scale(2), bounding box of the points lo0, lo1, hi0, hi1; d_seed: 2 G^2 ints */
extern "C" int gsl_sinterp_hip_mesh_pack(gsl_sinterp_hip_ctx *ctx, int n_tri, const int *d_tri, const int *d_nbr, int n_points,
                                         const double *d_points, const double *h_geom, int G, void *d_records, int *d_seed)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, n_tri > 0 && n_points >= 3 && G >= 1 && G <= 4096, ST_EINVAL);
  REQUIRE(ctx, d_tri && d_nbr && d_points && h_geom && d_records && d_seed, ST_EFAULT);
  REQUIRE(ctx, ((uintptr_t)d_records & 63) == 0, ST_EINVAL);
  Geom g;
  for (int i = 0; i < 6; i++) g.seed[i] = 0.0;
  g.shift[0] = h_geom[0]; g.shift[1] = h_geom[1];
  g.scale[0] = h_geom[2]; g.scale[1] = h_geom[3];
  hipLaunchKernelGGL(tree_pack_kernel, dim3((n_tri + 255) / 256), dim3(256), 0, ctx->stream, n_tri, (const int *)NULL, d_tri, d_nbr,
                     n_points, d_points, g, (NodeRec *)d_records);
  MeshGrid mg;
  mg.lo0 = h_geom[4]; mg.lo1 = h_geom[5]; mg.w0 = (h_geom[6] - h_geom[4]) / G; mg.w1 = (h_geom[7] - h_geom[5]) / G; mg.G = G;
  const size_t cells = (size_t)G * G;
  hipLaunchKernelGGL(mesh_seed_init_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream, d_seed + cells, cells);
  hipLaunchKernelGGL(mesh_seed_kernel, dim3((n_tri + 255) / 256), dim3(256), 0, ctx->stream, n_tri, d_tri, n_points, d_points, mg,
                     d_seed + cells);
  hipLaunchKernelGGL(mesh_seed_fill_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const int *)(d_seed + cells), d_seed, G);
  LAUNCH_CHECK(ctx);
  return ST_SUCCESS;
}

extern "C" int gsl_sinterp_hip_mesh_eval(gsl_sinterp_hip_ctx *ctx, int n_tri, const void *d_records, const void *d_leaftab,
                                         const int *d_seed, int G, const double *h_geom, int convex, const double *d_targets,
                                         size_t m, size_t ttda, double *d_values, int *d_tri, long long *h_n_outside)
{
  REQUIRE(ctx, ctx != NULL, ST_EFAULT);
  HIP_OK(ctx, hipSetDevice(ctx->device));
  REQUIRE(ctx, n_tri > 0 && ttda >= 2 && G >= 1, ST_EINVAL);
  REQUIRE(ctx, d_records && d_leaftab && d_seed && h_geom && (m == 0 || (d_targets && d_values)), ST_EFAULT);
  if (h_n_outside) *h_n_outside = 0;
  if (m == 0) return ST_SUCCESS;
  REQUIRE(ctx, m < 0xffffffffULL, ST_EINVAL);
  unsigned long long *d_count = (unsigned long long *)ctx->d_scratch;
  HIP_OK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
  void *buf = NULL;
  int st = sinterp_walkbuf(ctx, (m + 1) * sizeof(unsigned), &buf);      /* queue of the exhaustive scan */
  if (st) return st;
  unsigned *todo = (unsigned *)buf;
  HIP_OK(ctx, hipMemsetAsync(todo, 0, sizeof(unsigned), ctx->stream));
  MeshGrid mg;
  mg.lo0 = h_geom[4]; mg.lo1 = h_geom[5]; mg.w0 = (h_geom[6] - h_geom[4]) / G; mg.w1 = (h_geom[7] - h_geom[5]) / G; mg.G = G;
  /* batches of >= 4096 targets: the DAG path's reorder (cell order, two-level from 2^18 targets), results through the
     order's map, un-sorted afterwards.  A result depends on (mesh, target) only: same bits either way. */
  const bool will_sort = m >= 4096 && !(getenv("GSL_SINTERP_NO_SORT") && getenv("GSL_SINTERP_NO_SORT")[0] == '1');
  sinterp_sorted srt;
  if (will_sort) {
    st = sinterp_sort_reorder(ctx, d_targets, m, ttda, 2, 64, &srt, m, -1, (const unsigned long long *)NULL);
    if (st) return st;
  }
  const double *yt = will_sort ? (const double *)srt.ys : d_targets;
  const size_t yl = will_sort ? (size_t)2 : ttda;
  const bool via_map = will_sort && srt.two_level;
  double *vt = will_sort ? (via_map ? srt.res1 : srt.vs) : d_values;
  int *lt = will_sort ? (via_map ? (int *)srt.inv : (int *)NULL) : d_tri;
  const int packed = (will_sort && d_tri != NULL ? 1 : 0) | (via_map ? 2 : 0);
  /* a straight walk from a grid seed crosses a handful of triangles; the bound only guards against cycles (an imported
     non-Delaunay mesh can make the walk circle: such targets go to the exhaustive scan after 64 + 4 G steps at most) */
  const int max_steps = 64 + 4 * G;
  hipLaunchKernelGGL(mesh_walk_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, n_tri, (const NodeRec *)d_records,
                     (const LeafRec *)d_leaftab, d_seed, mg, h_geom[2], h_geom[3], convex, max_steps, yt, m, yl, vt,
                     lt, d_count, todo, packed);
  hipLaunchKernelGGL(mesh_scan_kernel, dim3(256), dim3(256), 0, ctx->stream, n_tri, (const NodeRec *)d_records,
                     (const LeafRec *)d_leaftab, h_geom[2], h_geom[3], yt, yl, vt, lt, d_count,
                     (const unsigned *)todo, packed);
  LAUNCH_CHECK(ctx);
  if (will_sort) {
    st = (packed & 1) ? sinterp_unsort_packed(ctx, &srt, m, d_values, d_tri) : sinterp_unsort(ctx, &srt, m, d_values, d_tri);
    if (st) return st;
  }
  if (h_n_outside) {
    unsigned long long c = 0;
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    HIP_OK(ctx, hipMemcpy(&c, d_count, sizeof c, hipMemcpyDeviceToHost));
    *h_n_outside = (long long)c;
    if (c) { snprintf(ctx->err, sizeof ctx->err, "mesh_eval: %llu target(s) outside the triangulation", c); return ST_EDOM; }
  }
  return ST_SUCCESS;
}
