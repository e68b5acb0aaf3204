/*
 * simplex_mesh.c -- imported triangulations: QHull / CGAL style arrays (points, triangles, neighbours) instead of
 * the history DAG that simplex_tree_init builds.  The reference lists this as future work (README:28-31); what it
 * fixes is the per-triangle arithmetic -- calculate_bary_coords / contains_point / interp_point
 * (interpolation/linear_simplex.c:607-711) -- and the link convention of a leaf (linear_simplex.h:62-63: link i =
 * neighbour opposite vertex i), both kept here.  All evaluation runs on the GPU (csrc/hip/bary.hip, "Imported
 * triangulations"); this file validates, derives neighbour links when they are not given, and mirrors the arrays.
 */
#include "gsl_sinterp.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct simplex_mesh {
  size_t n_tri, n_points;
  int *tri, *nbr;        /* [3 n_tri] */
  int *node;             /* [n_tri] DAG node (from_tree) or NULL */
  double *points;        /* [2 n_points], row order, packed */
  double shift[2], scale[2], lo[2], hi[2];
  int convex;
};

void simplex_mesh_free(simplex_mesh *mesh)
{
  if (!mesh) return;
  free(mesh->tri); free(mesh->nbr); free(mesh->node); free(mesh->points);
  free(mesh);
}

size_t simplex_mesh_n_triangles(const simplex_mesh *mesh) { return mesh ? mesh->n_tri : 0; }
size_t simplex_mesh_n_points(const simplex_mesh *mesh) { return mesh ? mesh->n_points : 0; }
const int *simplex_mesh_triangles(const simplex_mesh *mesh) { return mesh ? mesh->tri : NULL; }
const int *simplex_mesh_neighbours(const simplex_mesh *mesh) { return mesh ? mesh->nbr : NULL; }
const int *simplex_mesh_tree_nodes(const simplex_mesh *mesh) { return mesh ? mesh->node : NULL; }
void simplex_mesh_set_convex(simplex_mesh *mesh, int convex) { if (mesh) mesh->convex = convex != 0; }
int simplex_mesh_convex(const simplex_mesh *mesh) { return mesh ? mesh->convex : 0; }
void simplex_mesh_geometry(const simplex_mesh *mesh, double shift[2], double scale[2])
{
  for (int j = 0; j < 2; j++) { shift[j] = mesh->shift[j]; scale[j] = mesh->scale[j]; }
}

static simplex_mesh *mesh_alloc(size_t nt, size_t np, int with_nodes)
{
  simplex_mesh *m = (simplex_mesh *)calloc(1, sizeof *m);
  if (!m) return NULL;
  m->n_tri = nt; m->n_points = np; m->convex = 1;
  m->tri = (int *)malloc(3 * nt * sizeof(int));
  m->nbr = (int *)malloc(3 * nt * sizeof(int));
  m->points = (double *)malloc(2 * np * sizeof(double));
  if (with_nodes) m->node = (int *)malloc(nt * sizeof(int));
  if (!m->tri || !m->nbr || !m->points || (with_nodes && !m->node)) { simplex_mesh_free(m); return NULL; }
  return m;
}

static void mesh_bbox(simplex_mesh *m)
{
  for (int j = 0; j < 2; j++) { m->lo[j] = m->points[j]; m->hi[j] = m->points[j]; }
  for (size_t r = 1; r < m->n_points; r++)
    for (int j = 0; j < 2; j++) {
      const double v = m->points[2 * r + j];
      if (v < m->lo[j]) m->lo[j] = v;
      if (v > m->hi[j]) m->hi[j] = v;
    }
}

/* neighbour links by edge matching: every edge (a, b), a < b, with the triangle and the slot opposite to it, sorted */
typedef struct { int a, b, t, slot; } mesh_edge;
static int edge_cmp(const void *x, const void *y)
{
  const mesh_edge *p = (const mesh_edge *)x, *q = (const mesh_edge *)y;
  if (p->a != q->a) return p->a < q->a ? -1 : 1;
  if (p->b != q->b) return p->b < q->b ? -1 : 1;
  return p->t < q->t ? -1 : (p->t > q->t);
}

static int derive_neighbours(simplex_mesh *m)
{
  const size_t ne = 3 * m->n_tri;
  mesh_edge *e = (mesh_edge *)malloc(ne * sizeof *e);
  if (!e) return GSL_ENOMEM;
  for (size_t t = 0; t < m->n_tri; t++)
    for (int k = 0; k < 3; k++) {
      const int u = m->tri[3 * t + (k + 1) % 3], v = m->tri[3 * t + (k + 2) % 3];   /* the edge opposite vertex k */
      mesh_edge *x = &e[3 * t + k];
      x->a = u < v ? u : v; x->b = u < v ? v : u; x->t = (int)t; x->slot = k;
    }
  qsort(e, ne, sizeof *e, edge_cmp);
  for (size_t i = 0; i < 3 * m->n_tri; i++) m->nbr[i] = -1;
  int status = GSL_SUCCESS;
  for (size_t i = 0; i < ne;) {
    size_t j = i + 1;
    while (j < ne && e[j].a == e[i].a && e[j].b == e[i].b) j++;
    if (j - i == 2) {
      m->nbr[3 * e[i].t + e[i].slot] = e[i + 1].t;
      m->nbr[3 * e[i + 1].t + e[i + 1].slot] = e[i].t;
    } else if (j - i > 2) status = GSL_EINVAL;          /* an edge shared by three triangles: not a triangulation */
    i = j;
  }
  free(e);
  return status;
}

/* Convexity of an imported mesh, decided from its boundary (the edges without a neighbour): convex = ONE closed loop
   whose turns all have the sign of the loop's orientation (collinear boundary points allowed).  A hole, a concave
   outline, several components or a non-manifold boundary vertex all answer 0, and the locate step then never takes a
   boundary edge in the walking direction as proof that the target lies outside (exhaustive scan instead). */
static int mesh_detect_convex(const simplex_mesh *m)
{
  const size_t np = m->n_points, nt = m->n_tri;
  int *next = (int *)malloc(np * sizeof(int));
  if (!next) return 0;
  for (size_t i = 0; i < np; i++) next[i] = -1;
  size_t n_edges = 0;
  int first = -1, ok = 1;
  for (size_t t = 0; t < nt && ok; t++) {
    const int *v = m->tri + 3 * t;
    const double *p0 = m->points + 2 * v[0], *p1 = m->points + 2 * v[1], *p2 = m->points + 2 * v[2];
    const double orient = (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p1[1] - p0[1]) * (p2[0] - p0[0]);
    for (int k = 0; k < 3; k++) {
      if (m->nbr[3 * t + k] >= 0) continue;
      int a = v[(k + 1) % 3], b = v[(k + 2) % 3];               /* the edge opposite vertex k, in the triangle's order */
      if (orient < 0) { const int tmp = a; a = b; b = tmp; }    /* walk every boundary edge counter-clockwise */
      if (next[a] >= 0) { ok = 0; break; }                      /* two boundary edges leave one vertex: not a simple loop */
      next[a] = b;
      if (first < 0) first = a;
      n_edges++;
    }
  }
  if (ok && (first < 0 || n_edges < 3)) ok = 0;
  if (ok) {
    size_t seen = 0;
    int a = first;
    do {
      const int b = next[a];
      if (b < 0 || next[b] < 0) { ok = 0; break; }
      const int c = next[b];
      const double *pa = m->points + 2 * a, *pb = m->points + 2 * b, *pc = m->points + 2 * c;
      const double ux = pb[0] - pa[0], uy = pb[1] - pa[1], wx = pc[0] - pb[0], wy = pc[1] - pb[1];
      const double cross = ux * wy - uy * wx, tol = 1e-12 * sqrt((ux * ux + uy * uy) * (wx * wx + wy * wy));
      if (cross < -tol) { ok = 0; break; }                      /* a right turn on a counter-clockwise loop: concave */
      a = b;
      seen++;
    } while (a != first && seen <= n_edges);
    if (ok && (a != first || seen != n_edges)) ok = 0;          /* more boundary edges than this loop: holes / components */
  }
  free(next);
  return ok;
}

simplex_mesh *simplex_mesh_import(const gsl_matrix *points, const int *triangles, const int *neighbours, size_t n_triangles)
{
  if (!points || !triangles) GSL_ERROR_NULL("simplex_mesh_import: null argument", GSL_EFAULT);
  if (points->size2 < 2 || points->size1 < 3 || n_triangles < 1 || n_triangles > (size_t)INT_MAX / 3 || points->size1 > (size_t)INT_MAX)
    GSL_ERROR_NULL("simplex_mesh_import: need >= 3 points with 2 coordinates and >= 1 triangle", GSL_EINVAL);
  const size_t np = points->size1;
  for (size_t i = 0; i < 3 * n_triangles; i++) {
    if (triangles[i] < 0 || (size_t)triangles[i] >= np) GSL_ERROR_NULL("simplex_mesh_import: vertex id out of range", GSL_EINVAL);
    if (neighbours && (neighbours[i] < -1 || neighbours[i] >= (int)n_triangles))
      GSL_ERROR_NULL("simplex_mesh_import: neighbour id out of range", GSL_EINVAL);
  }
  for (size_t t = 0; t < n_triangles; t++)
    if (triangles[3 * t] == triangles[3 * t + 1] || triangles[3 * t] == triangles[3 * t + 2] || triangles[3 * t + 1] == triangles[3 * t + 2])
      GSL_ERROR_NULL("simplex_mesh_import: triangle with a repeated vertex", GSL_EINVAL);
  simplex_mesh *m = mesh_alloc(n_triangles, np, 0);
  if (!m) GSL_ERROR_NULL("simplex_mesh_import: out of memory", GSL_ENOMEM);
  memcpy(m->tri, triangles, 3 * n_triangles * sizeof(int));
  for (size_t r = 0; r < np; r++) { m->points[2 * r] = points->data[r * points->tda]; m->points[2 * r + 1] = points->data[r * points->tda + 1]; }
  mesh_bbox(m);
  /* the standardisation simplex_tree_init would use for these points (linear_simplex.c:226-247 as restated in
     simplex_tree.c): centre of the bounding box, 1 / extent */
  for (int j = 0; j < 2; j++) {
    m->shift[j] = (m->lo[j] + m->hi[j]) / 2.0;
    m->scale[j] = (m->hi[j] - m->lo[j] <= 0) ? 1.0 : 1.0 / (m->hi[j] - m->lo[j]);
  }
  if (neighbours) {
    memcpy(m->nbr, neighbours, 3 * n_triangles * sizeof(int));
    /* every link must be answered by the neighbour, across the same edge */
    for (size_t t = 0; t < n_triangles; t++)
      for (int k = 0; k < 3; k++) {
        const int nb = m->nbr[3 * t + k];
        if (nb < 0) continue;
        const int u = m->tri[3 * t + (k + 1) % 3], v = m->tri[3 * t + (k + 2) % 3];
        int ok = 0;
        for (int q = 0; q < 3 && !ok; q++)
          if (m->nbr[3 * nb + q] == (int)t) {
            const int a = m->tri[3 * nb + (q + 1) % 3], b = m->tri[3 * nb + (q + 2) % 3];
            ok = (a == u && b == v) || (a == v && b == u);
          }
        if (!ok) { simplex_mesh_free(m); GSL_ERROR_NULL("simplex_mesh_import: neighbour links are not mutual", GSL_EINVAL); }
      }
  } else {
    const int st = derive_neighbours(m);
    if (st != GSL_SUCCESS) { simplex_mesh_free(m); GSL_ERROR_NULL("simplex_mesh_import: cannot derive neighbour links", st); }
  }
  m->convex = mesh_detect_convex(m);                    /* simplex_mesh_set_convex overrides */
  return m;
}

simplex_mesh *simplex_mesh_from_tree(simplex_tree *tree, gsl_matrix *data)
{
  if (!tree || !data) GSL_ERROR_NULL("simplex_mesh_from_tree: null argument", GSL_EFAULT);
  if (tree->dim != 2) GSL_ERROR_NULL("simplex_mesh_from_tree: 2-D trees only", GSL_EUNIMPL);
  const int n = tree->n_simplexes, np = tree->n_points;
  int *index = (int *)malloc((size_t)n * sizeof(int));          /* DAG node -> triangle, -1 = not exported */
  if (!index) GSL_ERROR_NULL("simplex_mesh_from_tree: out of memory", GSL_ENOMEM);
  size_t nt = 0;
  for (int k = 0; k < n; k++) {
    index[k] = -1;
    if (!LEAF(k)) continue;
    if (POINT(k, 0) < 0 || POINT(k, 1) < 0 || POINT(k, 2) < 0) continue;      /* touches the cage: outside the hull */
    index[k] = (int)nt++;
  }
  if (nt == 0 || np < 3) { free(index); GSL_ERROR_NULL("simplex_mesh_from_tree: the tree has no triangle of data points", GSL_EINVAL); }
  simplex_mesh *m = mesh_alloc(nt, (size_t)np > data->size1 ? (size_t)np : data->size1, 1);
  if (!m) { free(index); GSL_ERROR_NULL("simplex_mesh_from_tree: out of memory", GSL_ENOMEM); }
  for (size_t r = 0; r < m->n_points; r++) {
    m->points[2 * r] = r < data->size1 ? data->data[r * data->tda] : 0.0;
    m->points[2 * r + 1] = r < data->size1 ? data->data[r * data->tda + 1] : 0.0;
  }
  for (int k = 0; k < n; k++) {
    const int t = index[k];
    if (t < 0) continue;
    m->node[t] = k;
    for (int i = 0; i < 3; i++) {
      m->tri[3 * t + i] = (int)gsl_permutation_get(tree->shuffle, (size_t)POINT(k, i));   /* insertion index -> data row */
      const simplex_index nb = LINK(k, i);
      m->nbr[3 * t + i] = nb > 0 ? index[nb] : -1;              /* a leaf's link 0 = none; cage neighbours -> hull */
    }
  }
  free(index);
  mesh_bbox(m);
  for (int j = 0; j < 2; j++) { m->shift[j] = gsl_vector_get(tree->shift, j); m->scale[j] = gsl_vector_get(tree->scale, j); }
  return m;
}

/* ------------------------------------------------------------------------ */
struct simplex_mesh_device {
  gsl_sinterp_hip_ctx *ctx;
  int n_tri, n_points, G, convex;
  double geom[8];
  void *d_records, *d_leaftab;
  int *d_tri, *d_seed;
  int response_bound;
};

gsl_sinterp_hip_ctx *simplex_mesh_device_ctx(simplex_mesh_device *dev) { return dev ? dev->ctx : NULL; }

void simplex_mesh_device_free(simplex_mesh_device *dev)
{
  if (!dev) return;
  if (dev->ctx) {
    gsl_sinterp_hip_free(dev->ctx, dev->d_records); gsl_sinterp_hip_free(dev->ctx, dev->d_leaftab);
    gsl_sinterp_hip_free(dev->ctx, dev->d_tri); gsl_sinterp_hip_free(dev->ctx, dev->d_seed);
    gsl_sinterp_hip_ctx_destroy(dev->ctx);
  }
  free(dev);
}

simplex_mesh_device *simplex_mesh_device_alloc(const simplex_mesh *mesh, int device)
{
  if (!mesh) GSL_ERROR_NULL("simplex_mesh_device_alloc: null mesh", GSL_EFAULT);
  simplex_mesh_device *dev = (simplex_mesh_device *)calloc(1, sizeof *dev);
  if (!dev) GSL_ERROR_NULL("simplex_mesh_device_alloc: out of memory", GSL_ENOMEM);
  dev->n_tri = (int)mesh->n_tri; dev->n_points = (int)mesh->n_points; dev->convex = mesh->convex;
  /* about two triangles per seed cell */
  int G = (int)ceil(sqrt((double)mesh->n_tri / 2.0));
  dev->G = G < 1 ? 1 : (G > 2048 ? 2048 : G);
  dev->geom[0] = mesh->shift[0]; dev->geom[1] = mesh->shift[1]; dev->geom[2] = mesh->scale[0]; dev->geom[3] = mesh->scale[1];
  dev->geom[4] = mesh->lo[0]; dev->geom[5] = mesh->lo[1]; dev->geom[6] = mesh->hi[0]; dev->geom[7] = mesh->hi[1];
  if (gsl_sinterp_hip_ctx_create(&dev->ctx, device, NULL) != GSL_SUCCESS) {
    free(dev);
    GSL_ERROR_NULL("simplex_mesh_device_alloc: no usable HIP device (GPU path has no CPU fallback)", GSL_EFAILED);
  }
  gsl_sinterp_hip_ctx *c = dev->ctx;
  const size_t tb = 3 * mesh->n_tri * sizeof(int), pb = 2 * mesh->n_points * sizeof(double);
  int *d_nbr = NULL;
  double *d_pts = NULL;
  int st = gsl_sinterp_hip_malloc(c, (void **)&dev->d_tri, tb);
  if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_nbr, tb);
  if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_pts, pb);
  if (!st) st = gsl_sinterp_hip_malloc(c, &dev->d_records, mesh->n_tri * GSL_SINTERP_TREE_RECORD_BYTES);
  if (!st) st = gsl_sinterp_hip_malloc(c, &dev->d_leaftab, mesh->n_tri * GSL_SINTERP_TREE_LEAFTAB_BYTES);
  if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&dev->d_seed, 2 * (size_t)dev->G * dev->G * sizeof(int));
  if (!st) st = gsl_sinterp_hip_h2d(c, dev->d_tri, mesh->tri, tb);
  if (!st) st = gsl_sinterp_hip_h2d(c, d_nbr, mesh->nbr, tb);
  if (!st) st = gsl_sinterp_hip_h2d(c, d_pts, mesh->points, pb);
  if (!st) st = gsl_sinterp_hip_mesh_pack(c, dev->n_tri, dev->d_tri, d_nbr, dev->n_points, d_pts, dev->geom, dev->G, dev->d_records, dev->d_seed);
  if (!st) st = gsl_sinterp_hip_sync(c);
  gsl_sinterp_hip_free(c, d_nbr); gsl_sinterp_hip_free(c, d_pts);
  if (st != GSL_SUCCESS) {
    gsl_error(gsl_sinterp_hip_last_error(c), __FILE__, __LINE__, st);
    simplex_mesh_device_free(dev);
    return NULL;
  }
  return dev;
}

int simplex_mesh_device_set_response(simplex_mesh_device *dev, const gsl_vector *response)
{
  if (!dev || !response) GSL_ERROR("simplex_mesh_device_set_response: null argument", GSL_EFAULT);
  if (response->size < (size_t)dev->n_points) GSL_ERROR("simplex_mesh_device_set_response: response shorter than the point set", GSL_EBADLEN);
  const size_t np = (size_t)dev->n_points;
  double *h = (double *)malloc(np * sizeof(double)), *d_resp = NULL;
  if (!h) GSL_ERROR("simplex_mesh_device_set_response: out of memory", GSL_ENOMEM);
  for (size_t i = 0; i < np; i++) h[i] = response->data[i * response->stride];
  int st = gsl_sinterp_hip_malloc(dev->ctx, (void **)&d_resp, np * sizeof(double));
  if (!st) st = gsl_sinterp_hip_h2d(dev->ctx, d_resp, h, np * sizeof(double));
  if (!st) st = gsl_sinterp_hip_tree_bind(dev->ctx, dev->n_tri, dev->d_tri, dev->n_points, d_resp, dev->d_leaftab);
  if (!st) st = gsl_sinterp_hip_sync(dev->ctx);
  gsl_sinterp_hip_free(dev->ctx, d_resp);
  free(h);
  if (st) GSL_ERROR(gsl_sinterp_hip_last_error(dev->ctx), st);
  dev->response_bound = 1;
  return GSL_SUCCESS;
}

int simplex_mesh_device_eval_resident(simplex_mesh_device *dev, const double *d_targets, size_t m, size_t ttda,
                                      double *d_values, int *d_triangle)
{
  if (!dev) GSL_ERROR("simplex_mesh_device_eval_resident: null device mirror", GSL_EFAULT);
  if (!dev->response_bound) GSL_ERROR("simplex_mesh_device_eval_resident: no response bound", GSL_EINVAL);
  int st = gsl_sinterp_hip_mesh_eval(dev->ctx, dev->n_tri, dev->d_records, dev->d_leaftab, dev->d_seed, dev->G, dev->geom, dev->convex,
                                     d_targets, m, ttda, d_values, d_triangle, NULL);
  if (st) GSL_ERROR(gsl_sinterp_hip_last_error(dev->ctx), st);
  return GSL_SUCCESS;
}

int simplex_mesh_device_eval_many(simplex_mesh_device *dev, const gsl_matrix *targets, gsl_vector *values, int *triangle)
{
  if (!dev || !targets || !values) GSL_ERROR("simplex_mesh_device_eval_many: null argument", GSL_EFAULT);
  if (!dev->response_bound) GSL_ERROR("simplex_mesh_device_eval_many: no response bound", GSL_EINVAL);
  if (targets->size2 != 2) GSL_ERROR("simplex_mesh_device_eval_many: targets must be M x 2", GSL_EBADLEN);
  const size_t m = targets->size1;
  if (values->size != m) GSL_ERROR("simplex_mesh_device_eval_many: values length must equal target rows", GSL_EBADLEN);
  if (m == 0) return GSL_SUCCESS;
  gsl_sinterp_hip_ctx *c = dev->ctx;
  double *h_y = (double *)malloc(2 * m * sizeof(double)), *h_s = (double *)malloc(m * sizeof(double));
  double *d_y = NULL, *d_s = NULL;
  int *d_t = NULL;
  long long n_out = 0;
  int st = (h_y && h_s) ? GSL_SUCCESS : GSL_ENOMEM;
  if (!st) for (size_t k = 0; k < m; k++) { h_y[2 * k] = targets->data[k * targets->tda]; h_y[2 * k + 1] = targets->data[k * targets->tda + 1]; }
  if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_y, 2 * m * sizeof(double));
  if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_s, m * sizeof(double));
  if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_t, m * sizeof(int));
  if (!st) st = gsl_sinterp_hip_h2d(c, d_y, h_y, 2 * m * sizeof(double));
  int st_eval = GSL_SUCCESS;
  if (!st) {
    st_eval = gsl_sinterp_hip_mesh_eval(c, dev->n_tri, dev->d_records, dev->d_leaftab, dev->d_seed, dev->G, dev->geom, dev->convex, d_y, m, 2,
                                        d_s, d_t, &n_out);
    if (st_eval != GSL_SUCCESS && st_eval != GSL_EDOM) st = st_eval;
  }
  if (!st) st = gsl_sinterp_hip_d2h(c, h_s, d_s, m * sizeof(double));
  if (!st && triangle) st = gsl_sinterp_hip_d2h(c, triangle, d_t, m * sizeof(int));
  if (!st) for (size_t k = 0; k < m; k++) values->data[k * values->stride] = h_s[k];
  gsl_sinterp_hip_free(c, d_y); gsl_sinterp_hip_free(c, d_s); gsl_sinterp_hip_free(c, d_t);
  free(h_y); free(h_s);
  if (st) GSL_ERROR(gsl_sinterp_hip_last_error(c), st);
  if (st_eval == GSL_EDOM) GSL_ERROR("simplex_mesh_device_eval_many: target(s) outside the triangulation", GSL_EDOM);
  return GSL_SUCCESS;
}
