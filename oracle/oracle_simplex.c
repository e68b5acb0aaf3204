/*
 * oracle_simplex.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Dimension-generic restatement of the reference's history-DAG Delaunay tree:
 * build (insert + flip), point location, barycentric coordinates and linear
 * interpolation.  Follows
 *   interpolation/linear_simplex.c   (alloc/init/find_leaf/insert/bary/interp)
 *   interpolation/edge_flip.c        (delaunay(), flippable(), link surgery)
 *   interpolation/linear_simplex_util.h (singular, dnrm22, orthonormalize)
 * Deliberate fixes: q1 seed matrix zero-initialised, q2 min/max
 * copied when all of data/min/max are given, q7 outside-cage target returns -1,
 * q9 debug integrity check / file dumps not run inside the build.
 * Everything else (q3 asymmetric matrix/rhs expressions, q4 closed inclusion
 * test + first-child-wins, q5 in-circle slack, q6 seed vertices contribute 0,
 * q12 strict pivot compare / exact-zero singularity) is reproduced.
 *
 * Node layout: the reference allocates (dim+1) pidx and (dim+1) links slots
 * for every node, in step (linear_simplex.c:31-46), so node k owns slots
 * [k*(dim+1), (k+1)*(dim+1)) of both arrays; that identity is used directly.
 */
#include "oracle.h"
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define D1(t) ((t)->dim + 1)
#define PT(t, node, i) ((t)->pidx[(size_t)(node) * D1(t) + (i)])
#define LK(t, node, i) ((t)->links[(size_t)(node) * D1(t) + (i)])
#define IS_LEAF(t, node) ((t)->type[node] == ORACLE_LEAF)

/* raw coordinates of vertex id v: v<0 -> cage seed -(v)-1, else data row
   shuffle[v]  (linear_simplex.h:83-93) */
static const double *vertex(const oracle_tree *t, const double *data, size_t tda, int v)
{
  if (v < 0) return t->seed + (size_t)(-v - 1) * t->dim;
  return data + t->shuffle[v] * tda;
}

static int n_children(const oracle_tree *t, int node)   /* linear_simplex.h:67-80 */
{
  switch (t->type[node]) {
    case ORACLE_SUB_DPLUS1: return t->dim + 1;
    case ORACLE_SUB_D: return t->dim;
    case ORACLE_SUB_2: return 2;
    default: return 0;
  }
}

static int node_alloc(oracle_tree *t)                   /* linear_simplex.c:17-51 */
{
  if (t->n_nodes + 1 >= t->cap_nodes) {
    int cap = t->cap_nodes * 2;
    if (cap < 16) cap = 16;
    t->type = (int *)realloc(t->type, (size_t)cap * sizeof(int));
    t->pidx = (int *)realloc(t->pidx, (size_t)cap * D1(t) * sizeof(int));
    t->links = (int *)realloc(t->links, (size_t)cap * D1(t) * sizeof(int));
    t->cap_nodes = cap;
  }
  int k = t->n_nodes++;
  t->type[k] = ORACLE_LEAF;
  for (int i = 0; i < D1(t); i++) { PT(t, k, i) = 0; LK(t, k, i) = 0; }
  return k;
}

oracle_tree *oracle_tree_alloc(int dim, int n_points)
{
  oracle_tree *t = (oracle_tree *)calloc(1, sizeof *t);
  t->dim = dim;
  t->max_points = n_points;
  t->cap_nodes = 9 * n_points;                          /* overhead = 9, :63,:74 */
  if (t->cap_nodes < 16) t->cap_nodes = 16;
  t->type = (int *)malloc((size_t)t->cap_nodes * sizeof(int));
  t->pidx = (int *)malloc((size_t)t->cap_nodes * D1(t) * sizeof(int));
  t->links = (int *)malloc((size_t)t->cap_nodes * D1(t) * sizeof(int));
  t->seed = (double *)calloc((size_t)D1(t) * dim, sizeof(double));   /* q1: zeroed */
  t->shift = (double *)calloc(dim, sizeof(double));
  t->scale = (double *)calloc(dim, sizeof(double));
  t->min = (double *)calloc(dim, sizeof(double));
  t->max = (double *)calloc(dim, sizeof(double));
  t->shuffle = (size_t *)malloc((size_t)(n_points > 0 ? n_points : 1) * sizeof(size_t));
  t->acc_mat = (double *)calloc((size_t)dim * dim, sizeof(double));
  t->acc_perm = (size_t *)calloc(dim, sizeof(size_t));
  t->acc_coords = (double *)calloc(dim, sizeof(double));
  t->acc_current = -1;
  node_alloc(t);                                        /* root, :79 */
  return t;
}

void oracle_tree_free(oracle_tree *t)
{
  if (!t) return;
  free(t->type); free(t->pidx); free(t->links); free(t->seed);
  free(t->shift); free(t->scale); free(t->min); free(t->max);
  free(t->shuffle); free(t->acc_mat); free(t->acc_perm); free(t->acc_coords);
  free(t);
}

/* ---------------------------------------------------------------------- */
/* barycentric coordinates with respect to the LAST vertex                 */
/* linear_simplex.c:607-651                                                */
int oracle_bary_coords(oracle_tree *t, int node, const double *data, size_t tda,
                       const double *point)
{
  const int dim = t->dim;
  const double *x0 = vertex(t, data, tda, PT(t, node, dim));

  if (node != t->acc_current) {
    t->acc_current = node;
    for (int i = 0; i < dim; i++) {
      const double *p = vertex(t, data, tda, PT(t, node, i));
      for (int j = 0; j < dim; j++) {
        double pv = t->scale[j] * (p[j] - t->shift[j]);
        double xv = t->scale[j] * (x0[j] - t->shift[j]);
        t->acc_mat[(size_t)j * dim + i] = pv - xv;       /* :633 */
      }
    }
    int signum;
    oracle_lu_decomp(dim, t->acc_mat, dim, t->acc_perm, &signum);
  }
  if (oracle_lu_singular(dim, t->acc_mat, dim)) return ORACLE_FAILURE;

  double pp[16];
  for (int j = 0; j < dim; j++) {                        /* :645-647: memcpy, sub, mul */
    double v = point[j];
    v = v - x0[j];
    v = v * t->scale[j];
    pp[j] = v;
  }
  oracle_lu_svx(dim, t->acc_mat, dim, t->acc_perm, pp);
  for (int j = 0; j < dim; j++) t->acc_coords[j] = pp[j];
  return ORACLE_SUCCESS;
}

/* linear_simplex.c:653-676 */
int oracle_contains_point(oracle_tree *t, int node, const double *data, size_t tda,
                          const double *point)
{
  t->stat_tests++;
  if (oracle_bary_coords(t, node, data, tda, point) != ORACLE_SUCCESS) return 0;
  double tot = 0;
  for (int i = 0; i < t->dim; i++) {
    double c = t->acc_coords[i];
    tot += c;
    if ((c < 0) || (c > 1)) return 0;
  }
  if ((tot < 0) || (tot > 1)) return 0;
  return 1;
}

/* linear_simplex.c:352-402 (tail recursion written as a loop) */
static int descend(oracle_tree *t, int node, const double *data, size_t tda, const double *point)
{
  const int dim = t->dim;
  long depth = 0;
  while (!IS_LEAF(t, node)) {
    int best_match = 0;
    double best_worst = -1;
    int next = -1;
    int nc = n_children(t, node);
    for (int i = 0; i < nc; i++) {
      double worst = 0;
      int child = LK(t, node, i);
      if (child && oracle_contains_point(t, child, data, tda, point)) { next = child; break; }
      double tot = 0;
      for (int j = 0; j < dim; j++) {
        double c = t->acc_coords[j];
        tot += c;
        if ((c < 0) && (-c > worst)) worst = -c;
        else if ((c > 1) && (c - 1 > worst)) worst = c - 1;
      }
      if ((tot < 0) && (-tot > worst)) worst = -tot;
      else if ((tot > 1) && (tot - 1 > worst)) worst = tot - 1;
      if ((best_worst < 0) || (worst < best_worst)) { best_worst = worst; best_match = i; }
    }
    if (next < 0) { next = LK(t, node, best_match); t->stat_fallbacks++; }   /* :398-400 */
    node = next;
    depth++;
  }
  t->stat_depth += depth;
  if (depth > t->stat_maxdepth) t->stat_maxdepth = depth;
  return node;
}

/* linear_simplex.c:331-350; q7: outside the cage -> -1 instead of assert(0) */
int oracle_find_leaf(oracle_tree *t, const double *data, size_t tda, const double *point)
{
  if (!oracle_contains_point(t, 0, data, tda, point)) return -1;
  return descend(t, 0, data, tda, point);
}

/* ---------------------------------------------------------------------- */
/* circumsphere in standardised coordinates, linear_simplex.c:555-605      */
static int hypersphere_points(oracle_tree *t, const int *points, const double *data, size_t tda,
                              double *x0, double *r2)
{
  const int dim = t->dim;
  t->acc_current = -1;
  for (int i = 0; i < dim; i++) {
    t->acc_coords[i] = 0;
    const double *vi = vertex(t, data, tda, points[i]);
    const double *vi1 = vertex(t, data, tda, points[i + 1]);
    for (int j = 0; j < dim; j++) {
      double pij = t->scale[j] * (vi[j] - t->shift[j]);
      double pij1 = t->scale[j] * (vi1[j] - t->shift[j]);
      t->acc_coords[i] = t->acc_coords[i] + pij * pij - pij1 * pij1;
      t->acc_mat[(size_t)i * dim + j] = pij - pij1;
    }
    t->acc_coords[i] = 0.5 * t->acc_coords[i];
  }
  int signum;
  oracle_lu_decomp(dim, t->acc_mat, dim, t->acc_perm, &signum);
  if (oracle_lu_singular(dim, t->acc_mat, dim)) return ORACLE_FAILURE;
  for (int j = 0; j < dim; j++) x0[j] = t->acc_coords[j];
  oracle_lu_svx(dim, t->acc_mat, dim, t->acc_perm, x0);

  const double *first = vertex(t, data, tda, points[0]);
  double mag2 = 0;
  for (int j = 0; j < dim; j++) {                        /* :596-602 */
    double v = first[j];
    v = v - t->shift[j];
    v = v * t->scale[j];
    v = v - x0[j];
    t->acc_coords[j] = v;
  }
  for (int j = 0; j < dim; j++) mag2 += t->acc_coords[j] * t->acc_coords[j];
  *r2 = mag2;
  return ORACLE_SUCCESS;
}

/* linear_simplex.c:507-537 */
static int in_hypersphere_points(oracle_tree *t, const int *points, const double *data, size_t tda, int idx)
{
  double x0[16], r2;
  const double *p = vertex(t, data, tda, idx);
  if (hypersphere_points(t, points, data, tda, x0, &r2) != ORACLE_SUCCESS) return 1;
  double dist2 = 0;
  for (int i = 0; i < t->dim; i++) {
    double comp = t->scale[i] * (p[i] - t->shift[i]);
    double val = comp - x0[i];
    dist2 += val * val;
  }
  return dist2 < (r2 * (1 - 10 * ORACLE_DBL_EPSILON));
}

int oracle_in_hypersphere(oracle_tree *t, int node, const double *data, size_t tda, int idx)
{
  int pts[17];
  for (int i = 0; i < D1(t); i++) pts[i] = PT(t, node, i);
  return in_hypersphere_points(t, pts, data, tda, idx);
}

static int point_in_simplex(const oracle_tree *t, int node, int point)   /* :318-329 */
{
  for (int i = 0; i < D1(t); i++)
    if (PT(t, node, i) == point) return 1;
  return 0;
}

/* ---------------------------------------------------------------------- */
/* Gram-Schmidt with the reference BLAS kernels, linear_simplex_util.h:43-70 */
static double blas_nrm2(int n, const double *x)          /* cblas/source_nrm2_r.h:20-50 */
{
  double scale = 0.0, ssq = 1.0;
  if (n <= 0) return 0;
  if (n == 1) return fabs(x[0]);
  for (int i = 0; i < n; i++) {
    double v = x[i];
    if (v != 0.0) {
      double ax = fabs(v);
      if (scale < ax) { ssq = 1.0 + ssq * (scale / ax) * (scale / ax); scale = ax; }
      else ssq += (ax / scale) * (ax / scale);
    }
  }
  return scale * sqrt(ssq);
}

static int orthonormalize(int n, double *m)
{
  double scale = -1;
  for (int i = 0; i < n; i++) {
    double *vi = m + (size_t)i * n;
    double mag = blas_nrm2(n, vi);
    if (scale < mag) scale = mag;
    if (mag < scale * 100 * ORACLE_DBL_EPSILON) return ORACLE_FAILURE;
    double a = 1 / mag;
    for (int k = 0; k < n; k++) vi[k] *= a;               /* dscal */
    for (int j = i + 1; j < n; j++) {
      double *vj = m + (size_t)j * n;
      double proj = 0.0;
      for (int k = 0; k < n; k++) proj += vi[k] * vj[k];  /* ddot */
      double alpha = -proj;
      if (alpha != 0.0)                                   /* daxpy early-out, source_axpy_r.h:23 */
        for (int k = 0; k < n; k++) vj[k] += alpha * vi[k];
    }
  }
  return ORACLE_SUCCESS;
}

/* edge_flip.c:17-35 : left_out[s] = s-th vertex slot that is not `face` */
static void set_left_out(int dim, int face, int *left_out)
{
  for (int s = 0; s < dim; s++)
    for (int i = 0; i < dim + 1; i++) {
      if (i == face) continue;
      int on_face = i > face ? i - 1 : i;
      if (on_face == s) left_out[s] = i;
    }
}

/* edge_flip.c:39-95 : convexity of the union of the two simplices, RAW coords */
static int flippable(oracle_tree *t, const double *data, size_t tda, int leaf, int face,
                     int neighbor, int far, const int *left_out)
{
  const int dim = t->dim;
  int ok = 1;
  double mat[16 * 16], v[16];
  const double *p_face = vertex(t, data, tda, PT(t, leaf, face));
  const double *p_far = vertex(t, data, tda, PT(t, neighbor, far));
  for (int s = 0; s < dim; s++) {
    for (int i = 0; i < dim + 1; i++) {
      if (i == face) continue;
      int on_face = i > face ? i - 1 : i;
      if (on_face == s) continue;
      if (on_face > s) on_face--;
      const double *p = vertex(t, data, tda, PT(t, leaf, i));
      for (int k = 0; k < dim; k++) mat[on_face * dim + k] = p[k] - p_face[k];
    }
    const double *p_lo = vertex(t, data, tda, PT(t, leaf, left_out[s]));
    for (int k = 0; k < dim; k++) mat[(dim - 1) * dim + k] = p_lo[k] - p_face[k];

    if (orthonormalize(dim, mat) != ORACLE_SUCCESS) return 1;

    for (int k = 0; k < dim; k++) v[k] = p_far[k] - p_face[k];
    double proj = 0.0;
    for (int k = 0; k < dim; k++) proj += mat[(dim - 1) * dim + k] * v[k];
    ok &= (proj > 0);
    if (!ok) break;
  }
  return ok;
}

/* edge_flip.c:98-114 */
static void save_neighbors(const oracle_tree *t, int leaf, int neighbor, int *old)
{
  int k = 0;
  for (int i = 0; i < D1(t); i++)
    if (LK(t, leaf, i) != neighbor) old[k++] = LK(t, leaf, i);
}

/* edge_flip.c:149-183 */
static void set_external_link(oracle_tree *t, const int *old, int replaced, int slot,
                              int fresh, int point_left_out)
{
  const int dim = t->dim;
  int j;
  for (j = 0; j < dim; j++) {
    if (!old[j]) continue;
    if (!point_in_simplex(t, old[j], point_left_out)) break;
  }
  int ext = (j < dim) ? old[j] : 0;
  LK(t, fresh, slot) = ext;
  if (ext) {
    for (int k = 0; k < dim + 1; k++)
      if (LK(t, ext, k) == replaced) { LK(t, ext, k) = fresh; break; }
  }
}

/* edge_flip.c:211-320 */
static int restore_delaunay(oracle_tree *t, int leaf, const double *data, size_t tda, int face)
{
  const int dim = t->dim;
  if (!LK(t, leaf, face)) return 0;
  int neighbor = LK(t, leaf, face);

  int far;
  for (far = 0; far < dim + 1; far++)
    if (LK(t, neighbor, far) == leaf) break;
  if (far == dim + 1) return 0;                          /* reference asserts */

  if (!oracle_in_hypersphere(t, leaf, data, tda, PT(t, neighbor, far))) return 0;

  int left_out[16];
  set_left_out(dim, face, left_out);
  if (!flippable(t, data, tda, leaf, face, neighbor, far, left_out)) return 0;

  /* the reference evaluates the reciprocal in-circle test inside an assert
     (edge_flip.c:255-256); it only touches accelerator scratch, kept for
     state fidelity */
  (void)oracle_in_hypersphere(t, neighbor, data, tda, PT(t, leaf, face));

  t->type[leaf] = ORACLE_SUB_D;
  t->type[neighbor] = ORACLE_SUB_D;

  int old1[16], old2[16], fresh[17];
  save_neighbors(t, leaf, neighbor, old1);
  save_neighbors(t, neighbor, leaf, old2);
  for (int s = 0; s < dim; s++) fresh[s] = node_alloc(t);

  for (int s = 0; s < dim; s++) {                        /* set_points, edge_flip.c:117-146 */
    PT(t, fresh[s], 0) = PT(t, leaf, face);
    PT(t, fresh[s], 1) = PT(t, neighbor, far);
    for (int j = 0; j < dim + 1; j++) {
      if (j == face) continue;
      int on_face = j > face ? j - 1 : j;
      if (on_face == s) continue;
      if (on_face > s) on_face--;
      PT(t, fresh[s], on_face + 2) = PT(t, leaf, j);
    }
  }
  for (int s = 0; s < dim; s++) {                        /* :283-289 */
    int lo = PT(t, leaf, left_out[s]);
    set_external_link(t, old2, neighbor, 0, fresh[s], lo);
    set_external_link(t, old1, leaf, 1, fresh[s], lo);
  }
  for (int s = 0; s < dim; s++)                          /* set_internal_links, :186-207 */
    for (int i = 2; i < dim + 1; i++) {
      int j;
      for (j = 0; j < dim; j++) {
        if (s == j) continue;
        if (!point_in_simplex(t, fresh[j], PT(t, fresh[s], i))) break;
      }
      if (j < dim) LK(t, fresh[s], i) = fresh[j];
    }
  for (int i = 0; i < dim; i++) { LK(t, leaf, i) = fresh[i]; LK(t, neighbor, i) = fresh[i]; }
  LK(t, leaf, dim) = neighbor;
  LK(t, neighbor, dim) = leaf;

  for (int s = 0; s < dim; s++)                          /* :307-316 */
    for (int i = 0; i < dim + 1; i++) {
      if (!IS_LEAF(t, LK(t, leaf, s))) break;
      if (!LK(t, LK(t, leaf, s), i)) continue;
      restore_delaunay(t, LK(t, leaf, s), data, tda, i);
    }
  return 1;
}

/* linear_simplex.c:404-492 */
int oracle_insert_point(oracle_tree *t, int leaf, const double *data, size_t tda)
{
  const int dim = t->dim;
  if (leaf < 0 || !IS_LEAF(t, leaf)) return ORACLE_FAILURE;
  t->type[leaf] = ORACLE_SUB_DPLUS1;

  int fresh[17];
  for (int s = 0; s < dim + 1; s++) fresh[s] = node_alloc(t);

  for (int i = 0; i < dim + 1; i++) {                    /* :425-434 */
    PT(t, fresh[i], 0) = t->n_points;
    int k = 1;
    for (int j = 0; j < dim + 1; j++) {
      if (j == i) continue;
      PT(t, fresh[i], k++) = PT(t, leaf, j);
    }
  }
  for (int i = 0; i < dim + 1; i++) {                    /* :437-455 */
    int nb = LK(t, leaf, i);
    LK(t, fresh[i], 0) = nb;
    if (nb) {
      for (int j = 0; j < dim + 1; j++)
        if (LK(t, nb, j) == leaf) { LK(t, nb, j) = fresh[i]; break; }
    }
  }
  for (int s = 0; s < dim + 1; s++)                      /* :458-475 */
    for (int i = 1; i < dim + 1; i++) {
      int j;
      for (j = 0; j < dim + 1; j++) {
        if (s == j) continue;
        if (!point_in_simplex(t, fresh[j], PT(t, fresh[s], i))) break;
      }
      if (j < dim + 1) LK(t, fresh[s], i) = fresh[j];
    }
  for (int i = 0; i < dim + 1; i++) LK(t, leaf, i) = fresh[i];
  t->n_points++;

  for (int i = 0; i < dim + 1; i++) {                    /* :483-488 */
    if (!IS_LEAF(t, LK(t, leaf, i))) continue;
    restore_delaunay(t, LK(t, leaf, i), data, tda, 0);
  }
  return ORACLE_SUCCESS;
}

/* linear_simplex.c:134-296 */
int oracle_tree_init(oracle_tree *t, const double *data, size_t n, size_t tda,
                     const double *min, const double *max, int flags, oracle_mt *rng)
{
  const int dim = t->dim;
  if (!(data || (min && max) || (flags & ORACLE_TREE_NOSTANDARDIZE))) return ORACLE_FAILURE;
  if (flags & ORACLE_TREE_NOSTANDARDIZE) {
    for (int i = 0; i < dim; i++) { t->min[i] = -0.5; t->max[i] = +0.5; }
  } else if (data && (!min || !max)) {
    for (int i = 0; i < dim; i++) {
      t->min[i] = min ? min[i] : data[i];
      t->max[i] = max ? max[i] : data[i];
    }
    for (size_t r = 1; r < n; r++)
      for (int j = 0; j < dim; j++) {
        double v = data[r * tda + j];
        if (!min && v < t->min[j]) t->min[j] = v;
        if (!max && v > t->max[j]) t->max[j] = v;
      }
  } else {                                               /* q2 fix: min && max given */
    for (int i = 0; i < dim; i++) { t->min[i] = min[i]; t->max[i] = max[i]; }
  }

  for (int i = 0; i < dim; i++) {                        /* :188-198 */
    double lo = t->min[i], hi = t->max[i];
    t->shift[i] = (lo + hi) / 2.0;
    t->scale[i] = (hi - lo <= 0) ? 1.0 : 1.0 / (hi - lo);
  }
  if (!(flags & ORACLE_TREE_NOSTANDARDIZE) && (flags & ORACLE_TREE_ISOSCALE)) {
    double mn = t->scale[0];
    for (int i = 1; i < dim; i++) if (mn > t->scale[i]) mn = t->scale[i];
    for (int i = 0; i < dim; i++) t->scale[i] = mn;
  }

  /* regular cage simplex, :217-232 */
  for (int i = 0; i < dim; i++) {
    double tot2 = 0;
    for (int j = 0; j < i; j++) { double c = t->seed[i * dim + j]; tot2 += c * c; }
    double chosen = sqrt(1 - tot2);
    t->seed[i * dim + i] = chosen;
    double others = -(1.0 / dim + tot2) / chosen;
    for (int j = i + 1; j < dim + 1; j++) t->seed[j * dim + i] = others;
  }
  double radius = (t->seed[0] - t->seed[dim]) / (dim + 1);           /* :241-243 */
  double grow = 1 / (ORACLE_ROOT5_DBL_EPSILON * radius);             /* :251 */
  for (int i = 0; i < (dim + 1) * dim; i++) t->seed[i] *= grow;
  for (int i = 0; i < dim + 1; i++)
    for (int j = 0; j < dim; j++) {                                   /* :255-260 */
      t->seed[i * dim + j] /= t->scale[j];
      t->seed[i * dim + j] += t->shift[j];
    }

  for (int i = 0; i < dim + 1; i++) { PT(t, 0, i) = -(i + 1); LK(t, 0, i) = 0; }
  for (int i = 0; i < t->max_points; i++) t->shuffle[i] = (size_t)i;

  int ret = ORACLE_SUCCESS;
  if (data) {
    if (t->n_points + (long)n > t->max_points) return ORACLE_FAILURE;
    if (rng) oracle_shuffle_sizet(rng, t->shuffle, n);
    for (size_t i = 0; i < n; i++) {
      const double *p = vertex(t, data, tda, (int)i);
      int leaf = oracle_find_leaf(t, data, tda, p);
      ret = oracle_insert_point(t, leaf, data, tda);
      if (ret != ORACLE_SUCCESS) break;
    }
  }
  return ret;
}

/* linear_simplex.c:678-711 */
double oracle_interp_point(oracle_tree *t, int leaf, const double *data, size_t tda,
                           const double *response, size_t rstride, const double *point)
{
  const int dim = t->dim;
  oracle_bary_coords(t, leaf, data, tda, point);
  double tot = 0, interp = 0;
  for (int i = 0; i < dim; i++) {
    double c = t->acc_coords[i];
    tot += c;
    int v = PT(t, leaf, i);
    if (v >= 0) interp += c * response[t->shuffle[v] * rstride];
  }
  int v = PT(t, leaf, dim);
  if (v >= 0) interp += (1 - tot) * response[t->shuffle[v] * rstride];
  return interp;
}

int oracle_bary_eval_many(oracle_tree *t, const double *data, size_t tda,
                          const double *response, size_t rstride,
                          const double *targets, size_t m, size_t ttda,
                          double *values, int *leaf)
{
  int status = ORACLE_SUCCESS;
  for (size_t k = 0; k < m; k++) {
    const double *y = targets + k * ttda;
    int lf = oracle_find_leaf(t, data, tda, y);
    if (leaf) leaf[k] = lf;
    if (lf < 0) { values[k] = NAN; status = ORACLE_EDOM; continue; }
    values[k] = oracle_interp_point(t, lf, data, tda, response, rstride, y);
  }
  return status;
}

/* ---------------------------------------------------------------------- */
/* structural predicates, linear_simplex_integrity_check.c:62-160           */
int oracle_check_leaf_nodes(oracle_tree *t)
{
  const int dim = t->dim;
  for (int node = 1; node < t->n_nodes; node++) {
    if (!IS_LEAF(t, node)) continue;
    for (int i = 0; i < dim + 1; i++)
      for (int j = i + 1; j < dim + 1; j++)
        if (PT(t, node, i) == PT(t, node, j)) return 0;          /* repeated vertex */
    for (int i = 0; i < dim + 1; i++) {
      int nb = LK(t, node, i);
      if (!nb) continue;
      if (!IS_LEAF(t, nb)) return 0;
      int back = 0;
      for (int j = 0; j < dim + 1; j++) if (LK(t, nb, j) == node) back++;
      if (back != 1) return 0;                                    /* symmetric links */
      if (point_in_simplex(t, nb, PT(t, node, i))) return 0;      /* opposite vertex not shared */
      for (int j = 0; j < dim + 1; j++)                           /* the other d vertices are shared */
        if (j != i && !point_in_simplex(t, nb, PT(t, node, j))) return 0;
    }
  }
  return 1;
}

int oracle_check_delaunay(oracle_tree *t, const double *data, size_t tda)
{
  const int dim = t->dim;
  double x0[16], r2;
  int pts[17];
  for (int node = 1; node < t->n_nodes; node++) {
    if (!IS_LEAF(t, node)) continue;
    for (int i = 0; i < dim + 1; i++) pts[i] = PT(t, node, i);
    if (hypersphere_points(t, pts, data, tda, x0, &r2) != ORACLE_SUCCESS) continue;
    for (int p = 0; p < t->n_points; p++) {
      if (point_in_simplex(t, node, p)) continue;
      const double *q = vertex(t, data, tda, p);
      double d2 = 0;
      for (int j = 0; j < dim; j++) {
        double c = t->scale[j] * (q[j] - t->shift[j]) - x0[j];
        d2 += c * c;
      }
      if (d2 < r2 * (1 - ORACLE_SQRT_DBL_EPSILON)) return 0;      /* :155 tolerance */
    }
  }
  return 1;
}

/* linear_simplex_integrity_check.c:170-284 */
struct dump_ctx { FILE *flines, *fcircles; const double *data; size_t tda; const double *response; size_t rstride; int std_out; unsigned char *seen; };

static void dump_leaf(oracle_tree *t, int node, struct dump_ctx *c)
{
  const int dim = t->dim;
  if (c->flines)
    for (int i = 0; i < dim + 1; i++)
      for (int j = i + 1; j < dim + 1; j++) {
        const int id[2] = {PT(t, node, i), PT(t, node, j)};
        if (id[0] < 0 || id[1] < 0) continue;
        for (int e = 0; e < 2; e++) {
          const double *p = vertex(t, c->data, c->tda, id[e]);
          const double r = c->response ? c->response[t->shuffle[id[e]] * c->rstride] : 0;
          for (int k = 0; k < dim; k++)
            fprintf(c->flines, "%g ", c->std_out ? t->scale[k] * (p[k] - t->shift[k]) : p[k]);
          fprintf(c->flines, e == 0 ? "%g\n" : "%g\n\n\n", r);
        }
      }
  if (c->fcircles) {
    double x0[16] = {0}, r2 = 0;
    int pts[17];
    for (int i = 0; i < dim + 1; i++) pts[i] = PT(t, node, i);
    (void)hypersphere_points(t, pts, c->data, c->tda, x0, &r2);
    fprintf(c->fcircles, "%g %g %g\n", x0[0], x0[1], sqrt(r2));
  }
}

static void dump_walk(oracle_tree *t, int node, struct dump_ctx *c)      /* :62-119, recursive like the reference */
{
  c->seen[node] = 1;
  dump_leaf(t, node, c);
  for (int i = 0; i < t->dim + 1; i++) {
    const int nb = LK(t, node, i);
    if (nb && !c->seen[nb]) dump_walk(t, nb, c);
  }
}

int oracle_output_triangulation(oracle_tree *t, const double *data, size_t tda, const double *response, size_t rstride,
                                int standardize_output, const char *lines_filename, const char *points_filename,
                                const char *circles_filename)
{
  struct dump_ctx c = {NULL, NULL, data, tda, response, rstride, standardize_output, NULL};
  if (lines_filename) c.flines = fopen(lines_filename, "w");
  if (circles_filename) c.fcircles = fopen(circles_filename, "w");
  if (points_filename) {
    FILE *fp = fopen(points_filename, "w");
    if (fp) {
      for (int i = 0; i < t->n_points; i++) {
        const double *q = data + t->shuffle[i] * tda;
        fprintf(fp, "%g %g\n", t->scale[0] * (q[0] - t->shift[0]), t->scale[1] * (q[1] - t->shift[1]));
      }
      fclose(fp);
    }
  }
  int leaf = 0;
  while (!IS_LEAF(t, leaf)) leaf = LK(t, leaf, 0);                    /* :124-128 */
  c.seen = (unsigned char *)calloc((size_t)t->n_nodes, 1);
  if (c.seen) dump_walk(t, leaf, &c);
  free(c.seen);
  if (c.flines) fclose(c.flines);
  if (c.fcircles) fclose(c.fcircles);
  return ORACLE_SUCCESS;
}

uint64_t oracle_tree_hash(const oracle_tree *t)
{
  uint64_t h = 1469598103934665603ULL;
  const size_t nd = (size_t)t->n_nodes, w = (size_t)D1(t);
#define MIX(val) do { uint32_t _v = (uint32_t)(val); for (int _b = 0; _b < 4; _b++) { \
    h ^= (_v >> (8 * _b)) & 0xffu; h *= 1099511628211ULL; } } while (0)
  for (size_t k = 0; k < nd; k++) MIX(t->type[k]);
  for (size_t k = 0; k < nd * w; k++) MIX(t->pidx[k]);
  for (size_t k = 0; k < nd * w; k++) MIX(t->links[k]);
#undef MIX
  return h;
}

/* ---------------------------------------------------------------------- */
/* Imported triangulations (README:28-31: future work in the reference, so there is no reference walk: PARITY
   UNPINNED).  What the reference fixes is the per-triangle arithmetic, restated here on explicit vertices:
   calculate_bary_coords (linear_simplex.c:607-651: matrix of standardised edge vectors wrt the LAST vertex,
   gsl_linalg_LU_decomp + _svx), contains_point's closed rule (:653-676) and interp_point (:678-711).          */
int oracle_mesh_coords(const double *data, size_t tda, const double *shift, const double *scale, const int *tri,
                       const double *point, double *coords)
{
  const int dim = 2;
  const double *x0 = data + (size_t)tri[dim] * tda;
  double mat[4];
  size_t perm[2];
  int signum;
  for (int i = 0; i < dim; i++) {
    const double *p = data + (size_t)tri[i] * tda;
    for (int j = 0; j < dim; j++) {
      double pv = scale[j] * (p[j] - shift[j]);
      double xv = scale[j] * (x0[j] - shift[j]);
      mat[(size_t)j * dim + i] = pv - xv;                /* :633 */
    }
  }
  oracle_lu_decomp(dim, mat, dim, perm, &signum);
  if (oracle_lu_singular(dim, mat, dim)) return ORACLE_FAILURE;
  double pp[2];
  for (int j = 0; j < dim; j++) {                        /* :645-647 */
    double v = point[j];
    v = v - x0[j];
    v = v * scale[j];
    pp[j] = v;
  }
  oracle_lu_svx(dim, mat, dim, perm, pp);
  coords[0] = pp[0]; coords[1] = pp[1];
  return ORACLE_SUCCESS;
}

int oracle_mesh_contains(const double *data, size_t tda, const double *shift, const double *scale, const int *tri,
                         const double *point)
{
  double c[2], tot = 0;
  if (oracle_mesh_coords(data, tda, shift, scale, tri, point, c) != ORACLE_SUCCESS) return 0;
  for (int i = 0; i < 2; i++) {
    tot += c[i];
    if ((c[i] < 0) || (c[i] > 1)) return 0;
  }
  if ((tot < 0) || (tot > 1)) return 0;
  return 1;
}

double oracle_mesh_interp(const double *data, size_t tda, const double *shift, const double *scale, const int *tri,
                          const double *response, size_t rstride, const double *point)
{
  double c[2], tot = 0, interp = 0;
  oracle_mesh_coords(data, tda, shift, scale, tri, point, c);
  for (int i = 0; i < 2; i++) { tot += c[i]; interp += c[i] * response[(size_t)tri[i] * rstride]; }
  interp += (1 - tot) * response[(size_t)tri[2] * rstride];
  return interp;
}

/* exhaustive search: smallest index of a containing triangle (-1: none); *n_containing counts them */
int oracle_mesh_locate(const double *data, size_t tda, const double *shift, const double *scale, const int *tris, size_t nt,
                       const double *point, int *n_containing)
{
  int first = -1, cnt = 0;
  for (size_t t = 0; t < nt; t++)
    if (oracle_mesh_contains(data, tda, shift, scale, tris + 3 * t, point)) { if (first < 0) first = (int)t; cnt++; }
  if (n_containing) *n_containing = cnt;
  return first;
}
