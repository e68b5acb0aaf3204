/*
 * simplex_tree.c -- host side of the barycentric path: the reference's
 * simplex_tree_* / find_leaf / interp_point symbols (2-D), i.e. the one-off
 * randomized incremental Delaunay build whose history DAG the GPU kernel walks
 * ("host-built").
 *
 * Written for d = 2 as straight-line 2x2 arithmetic.  Every floating-point
 * expression keeps the reference's operation order so that node numbering,
 * located leaves and interpolated values are bit-identical (compiled with
 * -ffp-contract=off):
 *   interpolation/linear_simplex.c:17-51   node allocation (arrays double on overflow)
 *   interpolation/linear_simplex.c:134-296 init: min/max, shift/scale, cage, shuffled insertion
 *   interpolation/linear_simplex.c:331-402 find_leaf / _find_leaf
 *   interpolation/linear_simplex.c:404-492 insert_point
 *   interpolation/linear_simplex.c:495-605 in-circle test in standardised coordinates
 *   interpolation/linear_simplex.c:607-711 barycentric coords, containment, interpolation
 *   interpolation/edge_flip.c:17-320       flippable() + 2->2 flip + recursive restoration
 *   linalg/lu.c:59-201 at N=2, cblas/source_trsv_r.h:33-79, cblas/source_nrm2_r.h:20-50
 * Deviations: q1 zeroed cage matrix, q2 min&&max honoured,
 * q7 target outside the cage -> -1 + GSL_EDOM instead of assert, q8 accel cache
 * not trusted, q9 no per-insertion O(N^2) debug check / file dumps.
 */
#include "gsl_sinterp.h"
#include <math.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define NV 3 /* vertices per simplex in 2-D */

/* ---------------------------------------------------------------------- */
/* 2x2 LU with partial pivoting, exactly gsl_linalg_LU_decomp at N=2        */
typedef struct {
  double u00, u01, l10, u11;
  int swapped;
} lu2;

static inline void lu2_factor(lu2 *f, double m00, double m01, double m10, double m11)
{
  f->swapped = fabs(m10) > fabs(m00);          /* strict: ties keep row 0 (lu.c:82-93) */
  if (f->swapped) {
    double t;
    t = m00; m00 = m10; m10 = t;
    t = m01; m01 = m11; m11 = t;
  }
  f->u00 = m00; f->u01 = m01; f->l10 = m10; f->u11 = m11;
  if (m00 != 0.0) {                            /* lu.c:105-119 */
    double l = m10 / m00;
    f->l10 = l;
    f->u11 = m11 - l * m01;
  }
}

static inline int lu2_singular(const lu2 *f) { return f->u00 == 0 || f->u11 == 0; }

/* x <- U^-1 L^-1 P b  (lu.c:189-197; source_trsv_r.h:56-79 then :33-55) */
static inline void lu2_solve(const lu2 *f, double b0, double b1, double *x0, double *x1)
{
  double c0 = f->swapped ? b1 : b0;
  double c1 = f->swapped ? b0 : b1;
  c1 -= f->l10 * c0;
  c1 = c1 / f->u11;
  c0 -= f->u01 * c1;
  c0 = c0 / f->u00;
  *x0 = c0; *x1 = c1;
}

/* ---------------------------------------------------------------------- */
static inline const double *vertex_xy(const simplex_tree *tree, const gsl_matrix *data, int v)
{
  if (v < 0) return tree->seed_points->data + (size_t)(-v - 1) * tree->seed_points->tda;
  return data->data + tree->shuffle->data[v] * data->tda;
}

simplex_index simplex_tree_node_alloc(simplex_tree *tree)
{
  const int dim = tree->dim;
  /* grow by doubling (linear_simplex.c:318-329); the old block stays valid and owned by the tree when realloc fails */
  if (tree->n_simplexes + 1 >= tree->max_simplexes) {
    simplex_tree_node *grown = (simplex_tree_node *)realloc(tree->simplexes, 2 * (size_t)tree->max_simplexes * sizeof(simplex_tree_node));
    if (!grown) { gsl_error("out of memory growing simplex tree", __FILE__, __LINE__, GSL_ENOMEM); return -1; }
    tree->simplexes = grown;
    tree->max_simplexes *= 2;
  }
  if (tree->n_pidx + dim + 1 >= tree->max_pidx) {
    int *grown = (int *)realloc(tree->pidx, 2 * (size_t)tree->max_pidx * sizeof(int));
    if (!grown) { gsl_error("out of memory growing simplex tree", __FILE__, __LINE__, GSL_ENOMEM); return -1; }
    tree->pidx = grown;
    tree->max_pidx *= 2;
  }
  if (tree->n_links + dim + 1 >= tree->max_links) {
    simplex_index *grown = (simplex_index *)realloc(tree->links, 2 * (size_t)tree->max_links * sizeof(simplex_index));
    if (!grown) { gsl_error("out of memory growing simplex tree", __FILE__, __LINE__, GSL_ENOMEM); return -1; }
    tree->links = grown;
    tree->max_links *= 2;
  }
  simplex_tree_node *node = &tree->simplexes[tree->n_simplexes];
  node->points = tree->n_pidx;
  node->links = tree->n_links;
  node->type = leaf_type;
  for (int i = 0; i < dim + 1; i++) {
    tree->pidx[tree->n_pidx + i] = 0;
    tree->links[tree->n_links + i] = 0;
  }
  tree->n_pidx += dim + 1;
  tree->n_links += dim + 1;
  return tree->n_simplexes++;
}

simplex_tree_accel *simplex_tree_accel_alloc(int dim)
{
  simplex_tree_accel *accel = (simplex_tree_accel *)malloc(sizeof *accel);
  if (!accel) GSL_ERROR_NULL("failed to allocate simplex_tree_accel", GSL_ENOMEM);
  accel->simplex_matrix = gsl_matrix_calloc(dim, dim);
  accel->perm = gsl_permutation_alloc(dim);
  accel->coords = gsl_vector_calloc(dim);
  gsl_permutation_init(accel->perm);
  accel->current_simplex = -1;
  return accel;
}

void simplex_tree_accel_free(simplex_tree_accel *accel)
{
  if (!accel) return;
  gsl_matrix_free(accel->simplex_matrix);
  gsl_permutation_free(accel->perm);
  gsl_vector_free(accel->coords);
  free(accel);
}

simplex_tree *simplex_tree_alloc(int dim, int n_points)
{
  if (dim != 2)
    GSL_ERROR_NULL("simplex_tree: only dim = 2 is supported (the reference's flip logic is 2-D only)", GSL_EUNIMPL);
  if (n_points < 0) GSL_ERROR_NULL("simplex_tree: negative capacity", GSL_EINVAL);
  simplex_tree *tree = (simplex_tree *)calloc(1, sizeof *tree);
  if (!tree) GSL_ERROR_NULL("failed to allocate simplex_tree", GSL_ENOMEM);
  tree->dim = dim;
  tree->seed_points = gsl_matrix_calloc(dim + 1, dim);
  tree->n_points = 0;
  tree->max_points = n_points;

  const int overhead = 9;                       /* linear_simplex.c:63 */
  if (n_points > (INT_MAX / (dim + 1)) / overhead) {          /* 9 * n_points * (dim + 1) must stay an int */
    simplex_tree_free(tree);
    GSL_ERROR_NULL("simplex_tree: capacity too large", GSL_EINVAL);
  }
  int cap = overhead * n_points;
  if (cap < 8) cap = 8;
  tree->max_pidx = cap * (dim + 1);
  tree->pidx = (int *)malloc((size_t)tree->max_pidx * sizeof(int));
  tree->max_links = cap * (dim + 1);
  tree->links = (simplex_index *)malloc((size_t)tree->max_links * sizeof(simplex_index));
  tree->max_simplexes = cap;
  tree->simplexes = (simplex_tree_node *)malloc((size_t)tree->max_simplexes * sizeof(simplex_tree_node));
  /* the reference leaves these unchecked (linear_simplex.c:56-103; SURVEY.md quirk q10, marked "fix") */
  if (!tree->seed_points || !tree->pidx || !tree->links || !tree->simplexes) {
    simplex_tree_free(tree);
    GSL_ERROR_NULL("failed to allocate simplex_tree arrays", GSL_ENOMEM);
  }

  simplex_tree_node_alloc(tree);                /* root */
  tree->accel = simplex_tree_accel_alloc(dim);
  tree->new_simplexes = (simplex_index *)calloc(dim + 1, sizeof(simplex_index));
  tree->old_neighbors1 = (simplex_index *)calloc(dim, sizeof(simplex_index));
  tree->old_neighbors2 = (simplex_index *)calloc(dim, sizeof(simplex_index));
  tree->left_out = (int *)calloc(dim, sizeof(int));
  tree->shift = gsl_vector_calloc(dim);
  tree->scale = gsl_vector_calloc(dim);
  tree->min = gsl_vector_calloc(dim);
  tree->max = gsl_vector_calloc(dim);
  tree->shuffle = gsl_permutation_alloc(n_points);
  tree->tmp_vec1 = gsl_vector_calloc(dim);
  tree->tmp_vec2 = gsl_vector_calloc(dim);
  tree->tmp_mat = gsl_matrix_calloc(dim, dim);
  tree->tmp_points1 = (int *)calloc(dim + 1, sizeof(int));
  if (!tree->accel || !tree->new_simplexes || !tree->old_neighbors1 || !tree->old_neighbors2 || !tree->left_out ||
      !tree->shift || !tree->scale || !tree->min || !tree->max || !tree->shuffle || !tree->tmp_vec1 || !tree->tmp_vec2 ||
      !tree->tmp_mat || !tree->tmp_points1) {
    simplex_tree_free(tree);
    GSL_ERROR_NULL("failed to allocate simplex_tree scratch", GSL_ENOMEM);
  }
  gsl_permutation_init(tree->shuffle);
  return tree;
}

void simplex_tree_free(simplex_tree *tree)
{
  if (!tree) return;
  gsl_matrix_free(tree->seed_points);
  simplex_tree_accel_free(tree->accel);
  free(tree->simplexes); free(tree->pidx); free(tree->links);
  free(tree->new_simplexes); free(tree->old_neighbors1); free(tree->old_neighbors2);
  free(tree->left_out);
  gsl_vector_free(tree->shift); gsl_vector_free(tree->scale);
  gsl_vector_free(tree->min); gsl_vector_free(tree->max);
  gsl_permutation_free(tree->shuffle);
  gsl_vector_free(tree->tmp_vec1); gsl_vector_free(tree->tmp_vec2);
  gsl_matrix_free(tree->tmp_mat);
  free(tree->tmp_points1);
  free(tree);
}

int point_in_simplex(simplex_tree *tree, simplex_index node, int point)
{
  for (int i = 0; i < NV; i++)
    if (POINT(node, i) == point) return 1;
  return 0;
}

/* ---------------------------------------------------------------------- */
/* barycentric coordinates with respect to the last vertex                 */
static int bary2(simplex_tree *tree, simplex_index node, const gsl_matrix *data,
                 double y0, double y1, simplex_tree_accel *accel)
{
  const double s0 = tree->scale->data[0], s1 = tree->scale->data[tree->scale->stride];
  const double h0 = tree->shift->data[0], h1 = tree->shift->data[tree->shift->stride];
  const double *p0 = vertex_xy(tree, data, POINT(node, 0));
  const double *p1 = vertex_xy(tree, data, POINT(node, 1));
  const double *x0 = vertex_xy(tree, data, POINT(node, 2));

  const double xv0 = s0 * (x0[0] - h0);
  const double xv1 = s1 * (x0[1] - h1);
  lu2 f;
  lu2_factor(&f,
             s0 * (p0[0] - h0) - xv0, s0 * (p1[0] - h0) - xv0,
             s1 * (p0[1] - h1) - xv1, s1 * (p1[1] - h1) - xv1);
  accel->current_simplex = node;
  if (lu2_singular(&f)) return GSL_FAILURE;

  double b0 = (y0 - x0[0]) * s0;               /* note: not the matrix expression (q3) */
  double b1 = (y1 - x0[1]) * s1;
  double c0, c1;
  lu2_solve(&f, b0, b1, &c0, &c1);
  accel->coords->data[0] = c0;
  accel->coords->data[accel->coords->stride] = c1;
  return GSL_SUCCESS;
}

static inline int inside_unit(double c0, double c1)
{
  double tot = 0;
  tot += c0;
  if ((c0 < 0) || (c0 > 1)) return 0;
  tot += c1;
  if ((c1 < 0) || (c1 > 1)) return 0;
  if ((tot < 0) || (tot > 1)) return 0;
  return 1;
}

static int contains2(simplex_tree *tree, simplex_index node, const gsl_matrix *data,
                     double y0, double y1, simplex_tree_accel *accel)
{
  if (bary2(tree, node, data, y0, y1, accel) != GSL_SUCCESS) return 0;
  return inside_unit(accel->coords->data[0], accel->coords->data[accel->coords->stride]);
}

int calculate_bary_coords(simplex_tree *tree, simplex_index node, gsl_matrix *data,
                          gsl_vector *point, simplex_tree_accel *accel)
{
  if (!accel) accel = tree->accel;
  return bary2(tree, node, data, point->data[0], point->data[point->stride], accel);
}

int contains_point(simplex_tree *tree, simplex_index node, gsl_matrix *data, gsl_vector *point,
                   simplex_tree_accel *accel)
{
  if (!accel) accel = tree->accel;
  return contains2(tree, node, data, point->data[0], point->data[point->stride], accel);
}

static inline int children_of(const simplex_tree *tree, simplex_index node)
{
  switch (tree->simplexes[node].type) {
    case sub_dplus1_type: return 3;
    case sub_d_type: return 2;
    case sub_2_type: return 2;
    default: return 0;
  }
}

static simplex_index descend2(simplex_tree *tree, simplex_index node, const gsl_matrix *data,
                              double y0, double y1, simplex_tree_accel *accel)
{
  while (!LEAF(node)) {
    int best = 0;
    double best_worst = -1;
    simplex_index next = -1;
    const int nc = children_of(tree, node);
    for (int i = 0; i < nc; i++) {
      simplex_index child = LINK(node, i);
      if (child && contains2(tree, child, data, y0, y1, accel)) { next = child; break; }
      /* violation of the child just tested (or stale coords if it was singular) */
      double worst = 0, tot = 0;
      for (int j = 0; j < 2; j++) {
        double c = accel->coords->data[j * accel->coords->stride];
        tot += c;
        if ((c < 0) && (-c > worst)) worst = -c;
        else if ((c > 1) && (c - 1 > worst)) worst = c - 1;
      }
      if ((tot < 0) && (-tot > worst)) worst = -tot;
      else if ((tot > 1) && (tot - 1 > worst)) worst = tot - 1;
      if ((best_worst < 0) || (worst < best_worst)) { best_worst = worst; best = i; }
    }
    node = next >= 0 ? next : LINK(node, best);
  }
  return node;
}

simplex_index _find_leaf(simplex_tree *tree, simplex_index node, gsl_matrix *data,
                         gsl_vector *point, simplex_tree_accel *accel)
{
  if (!accel) accel = tree->accel;
  return descend2(tree, node, data, point->data[0], point->data[point->stride], accel);
}

simplex_index find_leaf(simplex_tree *tree, gsl_matrix *data, gsl_vector *point,
                        simplex_tree_accel *accel)
{
  if (!accel) accel = tree->accel;
  const double y0 = point->data[0], y1 = point->data[point->stride];
  if (!contains2(tree, 0, data, y0, y1, accel))
    GSL_ERROR_VAL("find_leaf: point outside the caging simplex", GSL_EDOM, -1);
  return descend2(tree, 0, data, y0, y1, accel);
}

/* ---------------------------------------------------------------------- */
/* circumcircle in standardised coordinates                                 */
int calculate_hypersphere_points(simplex_tree *tree, int *points, gsl_matrix *data,
                                 gsl_vector *x0, double *r2, simplex_tree_accel *accel)
{
  if (!accel) accel = tree->accel;
  const double s[2] = {tree->scale->data[0], tree->scale->data[tree->scale->stride]};
  const double h[2] = {tree->shift->data[0], tree->shift->data[tree->shift->stride]};
  accel->current_simplex = -1;
  double m[2][2], rhs[2];
  for (int i = 0; i < 2; i++) {
    const double *vi = vertex_xy(tree, data, points[i]);
    const double *vi1 = vertex_xy(tree, data, points[i + 1]);
    double c = 0;
    for (int j = 0; j < 2; j++) {
      double pij = s[j] * (vi[j] - h[j]);
      double pij1 = s[j] * (vi1[j] - h[j]);
      c = c + pij * pij - pij1 * pij1;
      m[i][j] = pij - pij1;
    }
    rhs[i] = 0.5 * c;
  }
  lu2 f;
  lu2_factor(&f, m[0][0], m[0][1], m[1][0], m[1][1]);
  if (lu2_singular(&f)) {
    accel->coords->data[0] = rhs[0];
    accel->coords->data[accel->coords->stride] = rhs[1];
    return GSL_FAILURE;
  }
  double c0, c1;
  lu2_solve(&f, rhs[0], rhs[1], &c0, &c1);
  x0->data[0] = c0;
  x0->data[x0->stride] = c1;

  const double *first = vertex_xy(tree, data, points[0]);
  double d0 = first[0]; d0 = d0 - h[0]; d0 = d0 * s[0]; d0 = d0 - c0;
  double d1 = first[1]; d1 = d1 - h[1]; d1 = d1 * s[1]; d1 = d1 - c1;
  accel->coords->data[0] = d0;
  accel->coords->data[accel->coords->stride] = d1;
  double mag2 = 0;
  mag2 += d0 * d0;
  mag2 += d1 * d1;
  *r2 = mag2;
  return GSL_SUCCESS;
}

int calculate_hypersphere(simplex_tree *tree, simplex_index node, gsl_matrix *data,
                          gsl_vector *x0, double *r2, simplex_tree_accel *accel)
{
  int *pts = tree->tmp_points1;
  for (int i = 0; i < NV; i++) pts[i] = POINT(node, i);
  return calculate_hypersphere_points(tree, pts, data, x0, r2, accel);
}

int in_hypersphere_points(simplex_tree *tree, int *points, gsl_matrix *data, int idx,
                          simplex_tree_accel *accel)
{
  gsl_vector *x0 = tree->tmp_vec1;
  double r2;
  const double *p = vertex_xy(tree, data, idx);
  /* degenerate (collinear) vertices: treat as "inside" like the reference */
  if (calculate_hypersphere_points(tree, points, data, x0, &r2, accel) != GSL_SUCCESS) return 1;
  double dist2 = 0;
  for (int i = 0; i < 2; i++) {
    double comp = tree->scale->data[i * tree->scale->stride] * (p[i] - tree->shift->data[i * tree->shift->stride]);
    double val = comp - x0->data[i * x0->stride];
    dist2 += val * val;
  }
  return dist2 < (r2 * (1 - 10 * GSL_DBL_EPSILON));
}

int in_hypersphere(simplex_tree *tree, simplex_index node, gsl_matrix *data, int idx,
                   simplex_tree_accel *accel)
{
  int pts[NV];
  for (int i = 0; i < NV; i++) pts[i] = POINT(node, i);
  return in_hypersphere_points(tree, pts, data, idx, accel);
}

/* ---------------------------------------------------------------------- */
/* dnrm2 of a 2-vector with the reference BLAS scaling recurrence            */
static inline double nrm2_2(double a, double b)
{
  double scale = 0.0, ssq = 1.0;
  const double v[2] = {a, b};
  for (int i = 0; i < 2; i++) {
    if (v[i] != 0.0) {
      double ax = fabs(v[i]);
      if (scale < ax) { ssq = 1.0 + ssq * (scale / ax) * (scale / ax); scale = ax; }
      else ssq += (ax / scale) * (ax / scale);
    }
  }
  return scale * sqrt(ssq);
}

/* Is the quadrilateral leaf U neighbor strictly convex at both shared-edge
   ends?  (edge_flip.c:39-95, raw coordinates; Gram-Schmidt per
   linear_simplex_util.h:43-70) */
static int flippable2(simplex_tree *tree, const gsl_matrix *data, simplex_index leaf, int face,
                      simplex_index neighbor, int far)
{
  const double *pf = vertex_xy(tree, data, POINT(leaf, face));
  const double *pq = vertex_xy(tree, data, POINT(neighbor, far));
  const int e[2] = {face == 0 ? 1 : 0, face == 2 ? 1 : 2};   /* the two edge vertices */
  for (int s = 0; s < 2; s++) {
    const double *along = vertex_xy(tree, data, POINT(leaf, e[1 - s]));
    const double *normal = vertex_xy(tree, data, POINT(leaf, e[s]));
    double r0x = along[0] - pf[0], r0y = along[1] - pf[1];
    double r1x = normal[0] - pf[0], r1y = normal[1] - pf[1];

    double span = -1;
    double mag = nrm2_2(r0x, r0y);
    if (span < mag) span = mag;
    if (mag < span * 100 * GSL_DBL_EPSILON) return 1;
    double inv = 1 / mag;
    r0x *= inv; r0y *= inv;
    double proj = 0.0;
    proj += r0x * r1x;
    proj += r0y * r1y;
    double alpha = -proj;
    if (alpha != 0.0) { r1x += alpha * r0x; r1y += alpha * r0y; }
    mag = nrm2_2(r1x, r1y);
    if (span < mag) span = mag;
    if (mag < span * 100 * GSL_DBL_EPSILON) return 1;
    inv = 1 / mag;
    r1x *= inv; r1y *= inv;

    double vx = pq[0] - pf[0], vy = pq[1] - pf[1];
    double side = 0.0;
    side += r1x * vx;
    side += r1y * vy;
    if (!(side > 0)) return 0;
  }
  return 1;
}

static void relink(simplex_tree *tree, simplex_index node, simplex_index from, simplex_index to)
{
  for (int j = 0; j < NV; j++)
    if (LINK(node, j) == from) { LINK(node, j) = to; return; }
  gsl_error("simplex tree inconsistency: reverse link not found", __FILE__, __LINE__, GSL_ESANITY);
}

/* neighbour of the flipped pair that does not contain `apex` (edge_flip.c:149-183) */
static void attach_outer(simplex_tree *tree, const simplex_index *old, simplex_index replaced,
                         int slot, simplex_index fresh, int apex)
{
  int j;
  for (j = 0; j < 2; j++) {
    if (!old[j]) continue;
    if (!point_in_simplex(tree, old[j], apex)) break;
  }
  simplex_index ext = j < 2 ? old[j] : 0;
  LINK(fresh, slot) = ext;
  if (ext) relink(tree, ext, replaced, fresh);
}

/* The reference's flip cascade (edge_flip.c:211-320) recurses without bound; on exactly degenerate input
   (a regular lattice: collinear and co-circular points everywhere) it never settles and overruns the stack.
   Deliberate difference: the depth is bounded and the build fails with GSL_EFAILED instead of crashing. */
#define DELAUNAY_MAX_DEPTH 4000
static __thread int delaunay_depth = 0;
static __thread int delaunay_runaway = 0;   /* the current cascade hit the depth bound: unwind without flipping */
static __thread int delaunay_failed = 0;    /* sticky record of a runaway for simplex_tree_init (reset per insertion) */

static int delaunay_body(simplex_tree *tree, simplex_index leaf, gsl_matrix *data, int face, simplex_tree_accel *accel);

int delaunay(simplex_tree *tree, simplex_index leaf, gsl_matrix *data, int face,
             simplex_tree_accel *accel)
{
  if (delaunay_depth == 0) delaunay_runaway = 0;   /* a fresh cascade: do not inherit a failed build's flag */
  if (delaunay_runaway) return 0;
  if (delaunay_depth >= DELAUNAY_MAX_DEPTH) {
    delaunay_runaway = 1;
    delaunay_failed = 1;
    GSL_ERROR_VAL("delaunay: the flip cascade does not terminate (degenerate point set?)", GSL_EFAILED, 0);
  }
  delaunay_depth++;
  const int r = delaunay_body(tree, leaf, data, face, accel);
  delaunay_depth--;
  return r;
}

static int delaunay_body(simplex_tree *tree, simplex_index leaf, gsl_matrix *data, int face,
                         simplex_tree_accel *accel)
{
  if (!accel) accel = tree->accel;
  if (leaf <= 0 || !LEAF(leaf))
    GSL_ERROR_VAL("delaunay: flip test needs a non-root leaf", GSL_EINVAL, 0);
  simplex_index neighbor = LINK(leaf, face);
  if (!neighbor) return 0;

  int far;
  for (far = 0; far < NV; far++)
    if (LINK(neighbor, far) == leaf) break;
  if (far == NV) GSL_ERROR_VAL("simplex tree inconsistency: reverse link not found", GSL_ESANITY, 0);

  if (!in_hypersphere(tree, leaf, data, POINT(neighbor, far), accel)) return 0;
  if (!flippable2(tree, data, leaf, face, neighbor, far)) return 0;

  SIMP(leaf)->type = sub_d_type;
  SIMP(neighbor)->type = sub_d_type;

  /* outer neighbours before surgery (edge_flip.c:98-114) */
  simplex_index out_leaf[2], out_nb[2];
  for (int i = 0, k = 0; i < NV; i++) if (LINK(leaf, i) != neighbor) out_leaf[k++] = LINK(leaf, i);
  for (int i = 0, k = 0; i < NV; i++) if (LINK(neighbor, i) != leaf) out_nb[k++] = LINK(neighbor, i);

  simplex_index fresh[2];
  fresh[0] = simplex_tree_node_alloc(tree);
  fresh[1] = simplex_tree_node_alloc(tree);

  /* slots of the shared edge's vertices in `leaf`, in on-face order */
  const int e[2] = {face == 0 ? 1 : 0, face == 2 ? 1 : 2};
  for (int s = 0; s < 2; s++) {                          /* edge_flip.c:117-146 */
    POINT(fresh[s], 0) = POINT(leaf, face);
    POINT(fresh[s], 1) = POINT(neighbor, far);
    POINT(fresh[s], 2) = POINT(leaf, e[1 - s]);          /* keeps the edge vertex != e[s] */
  }
  for (int s = 0; s < 2; s++) {                          /* edge_flip.c:283-289 */
    int apex = POINT(leaf, e[s]);
    attach_outer(tree, out_nb, neighbor, 0, fresh[s], apex);
    attach_outer(tree, out_leaf, leaf, 1, fresh[s], apex);
  }
  LINK(fresh[0], 2) = fresh[1];                          /* edge_flip.c:186-207 at d=2 */
  LINK(fresh[1], 2) = fresh[0];

  for (int i = 0; i < 2; i++) { LINK(leaf, i) = fresh[i]; LINK(neighbor, i) = fresh[i]; }
  LINK(leaf, 2) = neighbor;
  LINK(neighbor, 2) = leaf;

  for (int s = 0; s < 2; s++)                            /* edge_flip.c:307-316 */
    for (int i = 0; i < NV; i++) {
      if (!LEAF(LINK(leaf, s))) break;
      if (!LINK(LINK(leaf, s), i)) continue;
      delaunay(tree, LINK(leaf, s), data, i, accel);
    }
  return 1;
}

int insert_point(simplex_tree *tree, simplex_index leaf, gsl_matrix *data, gsl_vector *point,
                 simplex_tree_accel *accel)
{
  (void)point; /* like the reference, the new vertex is data row shuffle[n_points] */
  if (!accel) accel = tree->accel;
  if (leaf < 0 || leaf >= tree->n_simplexes || !LEAF(leaf))
    GSL_ERROR("insert_point: a point can only be inserted into a leaf", GSL_EINVAL);
  if (tree->n_points >= tree->max_points)
    GSL_ERROR("insert_point: tree is full", GSL_FAILURE);
  SIMP(leaf)->type = sub_dplus1_type;

  simplex_index fresh[NV];
  for (int s = 0; s < NV; s++) fresh[s] = simplex_tree_node_alloc(tree);

  for (int i = 0; i < NV; i++) {
    POINT(fresh[i], 0) = tree->n_points;
    int k = 1;
    for (int j = 0; j < NV; j++)
      if (j != i) POINT(fresh[i], k++) = POINT(leaf, j);
  }
  for (int i = 0; i < NV; i++) {
    simplex_index nb = LINK(leaf, i);
    LINK(fresh[i], 0) = nb;
    if (nb) relink(tree, nb, leaf, fresh[i]);
  }
  for (int s = 0; s < NV; s++)
    for (int i = 1; i < NV; i++) {
      int j;
      for (j = 0; j < NV; j++) {
        if (s == j) continue;
        if (!point_in_simplex(tree, fresh[j], POINT(fresh[s], i))) break;
      }
      if (j < NV) LINK(fresh[s], i) = fresh[j];
    }
  for (int i = 0; i < NV; i++) LINK(leaf, i) = fresh[i];
  tree->n_points++;

  for (int i = 0; i < NV; i++) {
    if (!LEAF(LINK(leaf, i))) continue;
    delaunay(tree, LINK(leaf, i), data, 0, accel);
  }
  return GSL_SUCCESS;
}

int simplex_tree_init(simplex_tree *tree, gsl_matrix *data, gsl_vector *min, gsl_vector *max,
                      int init_flags, gsl_rng *rng)
{
  const int dim = tree->dim;
  if (!(data || (min && max) || (init_flags & SIMPLEX_TREE_NOSTANDARDIZE))) return GSL_FAILURE;
  if (data && data->size2 < (size_t)dim) GSL_ERROR("simplex_tree_init: data has fewer than dim columns", GSL_EBADLEN);

  if (init_flags & SIMPLEX_TREE_NOSTANDARDIZE) {
    for (int i = 0; i < dim; i++) { gsl_vector_set(tree->min, i, -0.5); gsl_vector_set(tree->max, i, +0.5); }
  } else if (data && (!min || !max)) {
    for (int i = 0; i < dim; i++) {
      gsl_vector_set(tree->min, i, min ? gsl_vector_get(min, i) : gsl_matrix_get(data, 0, i));
      gsl_vector_set(tree->max, i, max ? gsl_vector_get(max, i) : gsl_matrix_get(data, 0, i));
    }
    for (size_t r = 1; r < data->size1; r++)
      for (int j = 0; j < dim; j++) {
        double v = gsl_matrix_get(data, r, j);
        if (!min && v < gsl_vector_get(tree->min, j)) gsl_vector_set(tree->min, j, v);
        if (!max && v > gsl_vector_get(tree->max, j)) gsl_vector_set(tree->max, j, v);
      }
  } else {
    for (int i = 0; i < dim; i++) {
      gsl_vector_set(tree->min, i, gsl_vector_get(min, i));
      gsl_vector_set(tree->max, i, gsl_vector_get(max, i));
    }
  }

  for (int i = 0; i < dim; i++) {
    double lo = gsl_vector_get(tree->min, i), hi = gsl_vector_get(tree->max, i);
    gsl_vector_set(tree->shift, i, (lo + hi) / 2.0);
    gsl_vector_set(tree->scale, i, (hi - lo <= 0) ? 1.0 : 1.0 / (hi - lo));
  }
  if (!(init_flags & SIMPLEX_TREE_NOSTANDARDIZE) && (init_flags & SIMPLEX_TREE_ISOSCALE)) {
    double mn = gsl_vector_get(tree->scale, 0);
    for (int i = 1; i < dim; i++) if (mn > gsl_vector_get(tree->scale, i)) mn = gsl_vector_get(tree->scale, i);
    for (int i = 0; i < dim; i++) gsl_vector_set(tree->scale, i, mn);
  }

  /* regular caging simplex blown up by 1/(eps^(1/5) * inradius), then mapped back
     to raw coordinates */
  gsl_matrix *sp = tree->seed_points;
  for (size_t k = 0; k < sp->size1 * sp->tda; k++) sp->data[k] = 0.0;
  for (int i = 0; i < dim; i++) {
    double tot2 = 0;
    for (int j = 0; j < i; j++) { double c = gsl_matrix_get(sp, i, j); tot2 += c * c; }
    double chosen = sqrt(1 - tot2);
    gsl_matrix_set(sp, i, i, chosen);
    double others = -(1.0 / dim + tot2) / chosen;
    for (int j = i + 1; j < dim + 1; j++) gsl_matrix_set(sp, j, i, others);
  }
  double radius = (gsl_matrix_get(sp, 0, 0) - gsl_matrix_get(sp, 1, 0)) / (dim + 1);
  double grow = 1 / (GSL_ROOT5_DBL_EPSILON * radius);
  for (int i = 0; i < dim + 1; i++)
    for (int j = 0; j < dim; j++) {
      double v = gsl_matrix_get(sp, i, j) * grow;
      v /= gsl_vector_get(tree->scale, j);
      v += gsl_vector_get(tree->shift, j);
      gsl_matrix_set(sp, i, j, v);
    }

  for (int i = 0; i < dim + 1; i++) { POINT(0, i) = -(i + 1); LINK(0, i) = 0; }
  gsl_permutation_init(tree->shuffle);

  int ret = GSL_SUCCESS;
  if (data) {
    if (tree->n_points + (long)data->size1 > tree->max_points) return GSL_FAILURE;
    if (rng) gsl_ran_shuffle(rng, tree->shuffle->data, data->size1, sizeof(size_t));
    gsl_error_handler_t *saved = gsl_set_error_handler_off();   /* report, do not abort mid-build */
    for (size_t i = 0; i < data->size1; i++) {
      const double *p = vertex_xy(tree, data, (int)i);
      simplex_index leaf = -1;
      if (contains2(tree, 0, data, p[0], p[1], tree->accel))
        leaf = descend2(tree, 0, data, p[0], p[1], tree->accel);
      delaunay_failed = 0;
      ret = leaf < 0 ? GSL_EDOM : insert_point(tree, leaf, data, NULL, tree->accel);
      if (ret == GSL_SUCCESS && delaunay_failed) ret = GSL_EFAILED;
      delaunay_failed = 0;
      if (ret != GSL_SUCCESS) break;
    }
    gsl_set_error_handler(saved);
    if (ret != GSL_SUCCESS) GSL_ERROR("simplex_tree_init: insertion failed", ret);
  }
  return ret;
}

double interp_point(simplex_tree *tree, simplex_index leaf, gsl_matrix *data,
                    gsl_vector *response, gsl_vector *point, simplex_tree_accel *accel)
{
  if (!accel) accel = tree->accel;
  if (leaf < 0 || leaf >= tree->n_simplexes || !LEAF(leaf))
    GSL_ERROR_VAL("interp_point: interpolation must be on a leaf node", GSL_EINVAL, GSL_NAN);
  bary2(tree, leaf, data, point->data[0], point->data[point->stride], accel);
  double tot = 0, interp = 0;
  for (int i = 0; i < 2; i++) {
    double c = accel->coords->data[i * accel->coords->stride];
    tot += c;
    int v = POINT(leaf, i);
    if (v >= 0) interp += c * gsl_vector_get(response, tree->shuffle->data[v]);
  }
  int v = POINT(leaf, 2);
  if (v >= 0) interp += (1 - tot) * gsl_vector_get(response, tree->shuffle->data[v]);
  return interp;
}
