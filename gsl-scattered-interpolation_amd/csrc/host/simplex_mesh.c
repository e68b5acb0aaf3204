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
#include <stdint.h>
#include <stdio.h>
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
const double *simplex_mesh_points(const simplex_mesh *mesh) { return mesh ? mesh->points : NULL; }
void simplex_mesh_bbox(const simplex_mesh *mesh, double lo[2], double hi[2])
{
  for (int j = 0; j < 2; j++) { lo[j] = mesh->lo[j]; hi[j] = mesh->hi[j]; }
}
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
        if (nb == (int)t) { simplex_mesh_free(m); GSL_ERROR_NULL("simplex_mesh_import: a triangle is its own neighbour", GSL_EINVAL); }
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
/* Binary checkpoint of a mesh (gsl_matrix_fwrite conventions: native byte order, GSL_EFAILED on a short transfer):
     magic "GSLSMSH1" | n_tri, n_points, has_nodes, convex (int64) | triangles | neighbours | [tree nodes] | points |
     shift, scale, lo, hi.
   fread re-checks everything the import checks (ids in range, no repeated vertex, mutual links across the same edge):
   the locate kernels index with these arrays. */
static const char MESH_MAGIC[8] = {'G', 'S', 'L', 'S', 'M', 'S', 'H', '1'};

int simplex_mesh_fwrite(FILE *stream, const simplex_mesh *m)
{
  if (!stream || !m) GSL_ERROR("simplex_mesh_fwrite: null argument", GSL_EFAULT);
  const int64_t head[4] = {(int64_t)m->n_tri, (int64_t)m->n_points, m->node ? 1 : 0, m->convex};
  int ok = fwrite(MESH_MAGIC, 1, 8, stream) == 8 && fwrite(head, sizeof head[0], 4, stream) == 4;
  ok = ok && fwrite(m->tri, sizeof(int), 3 * m->n_tri, stream) == 3 * m->n_tri;
  ok = ok && fwrite(m->nbr, sizeof(int), 3 * m->n_tri, stream) == 3 * m->n_tri;
  if (m->node) ok = ok && fwrite(m->node, sizeof(int), m->n_tri, stream) == m->n_tri;
  ok = ok && fwrite(m->points, sizeof(double), 2 * m->n_points, stream) == 2 * m->n_points;
  double geo[8] = {m->shift[0], m->shift[1], m->scale[0], m->scale[1], m->lo[0], m->lo[1], m->hi[0], m->hi[1]};
  ok = ok && fwrite(geo, sizeof(double), 8, stream) == 8;
  if (!ok) GSL_ERROR("simplex_mesh_fwrite: fwrite failed", GSL_EFAILED);
  return GSL_SUCCESS;
}

simplex_mesh *simplex_mesh_fread(FILE *stream)
{
  if (!stream) GSL_ERROR_NULL("simplex_mesh_fread: null stream", GSL_EFAULT);
  char magic[8];
  int64_t head[4];
  if (fread(magic, 1, 8, stream) != 8 || memcmp(magic, MESH_MAGIC, 8) != 0)
    GSL_ERROR_NULL("simplex_mesh_fread: not a simplex_mesh checkpoint", GSL_EFAILED);
  if (fread(head, sizeof head[0], 4, stream) != 4) GSL_ERROR_NULL("simplex_mesh_fread: short header", GSL_EFAILED);
  if (head[0] < 1 || head[0] > INT_MAX / 3 || head[1] < 3 || head[1] > INT_MAX || (head[2] != 0 && head[2] != 1))
    GSL_ERROR_NULL("simplex_mesh_fread: corrupt header", GSL_EFAILED);
  const size_t nt = (size_t)head[0], np = (size_t)head[1];
  {
    /* seekable stream: the counts must fit what is left of the file before anything is allocated for them */
    const long here = ftell(stream);
    if (here >= 0 && fseek(stream, 0L, SEEK_END) == 0) {
      const long end = ftell(stream);
      const long long need = 4LL * (6 + head[2]) * (long long)nt + 16LL * (long long)np + 64;
      if (fseek(stream, here, SEEK_SET) != 0) GSL_ERROR_NULL("simplex_mesh_fread: stream error", GSL_EFAILED);
      if (end >= here && (long long)(end - here) < need) GSL_ERROR_NULL("simplex_mesh_fread: short or corrupt checkpoint", GSL_EFAILED);
    }
  }
  simplex_mesh *m = mesh_alloc(nt, np, (int)head[2]);
  if (!m) GSL_ERROR_NULL("simplex_mesh_fread: out of memory", GSL_ENOMEM);
  double geo[8];
  int ok = fread(m->tri, sizeof(int), 3 * nt, stream) == 3 * nt && fread(m->nbr, sizeof(int), 3 * nt, stream) == 3 * nt;
  if (m->node) ok = ok && fread(m->node, sizeof(int), nt, stream) == nt;
  ok = ok && fread(m->points, sizeof(double), 2 * np, stream) == 2 * np && fread(geo, sizeof(double), 8, stream) == 8;
  for (size_t i = 0; ok && i < 3 * nt; i++)
    ok = m->tri[i] >= 0 && (size_t)m->tri[i] < np && m->nbr[i] >= -1 && m->nbr[i] < (int)nt;
  for (size_t t = 0; ok && t < nt; t++) {
    const int *v = m->tri + 3 * t;
    ok = v[0] != v[1] && v[0] != v[2] && v[1] != v[2];
    for (int k = 0; ok && k < 3; k++) {                          /* links mutual, across the same edge */
      const int nb = m->nbr[3 * t + k];
      if (nb < 0) continue;
      if (nb == (int)t) { ok = 0; break; }                       /* a triangle is not its own neighbour */
      const int u = v[(k + 1) % 3], w = v[(k + 2) % 3];
      int found = 0;
      for (int q = 0; q < 3 && !found; q++)
        if (m->nbr[3 * nb + q] == (int)t) {
          const int a = m->tri[3 * nb + (q + 1) % 3], b = m->tri[3 * nb + (q + 2) % 3];
          found = (a == u && b == w) || (a == w && b == u);
        }
      ok = found;
    }
  }
  if (!ok) { simplex_mesh_free(m); GSL_ERROR_NULL("simplex_mesh_fread: short or corrupt checkpoint", GSL_EFAILED); }
  for (int j = 0; j < 2; j++) { m->shift[j] = geo[j]; m->scale[j] = geo[2 + j]; m->lo[j] = geo[4 + j]; m->hi[j] = geo[6 + j]; }
  m->convex = head[3] != 0;
  return m;
}
