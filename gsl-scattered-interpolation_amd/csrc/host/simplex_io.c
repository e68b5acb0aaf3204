/*
 * simplex_io.c -- traversal, dump and (de)serialisation of a built simplex_tree (host C).
 *
 *   check_leaf_nodes      interpolation/linear_simplex_integrity_check.c:121-132 -- the depth-first walk over
 *                         the leaf adjacency that applies a callback to every leaf, in the reference's order
 *                         (neighbours by link index, pre-order).  The reference recurses and keeps the visited
 *                         set in a linked list (O(n) membership test, :52-60, quadratic overall); here it is an
 *                         explicit stack and a bitmap, so N = 50 000 walks in milliseconds and cannot overrun
 *                         the C stack.  The structural asserts of _check_leaf_nodes (:62-119) live on the GPU
 *                         (csrc/hip/check.hip); this walk only reports a neighbour that is not a leaf.
 *   check_delaunay        :162-168 -- same name and return convention (1 = passed); runs the device-side check
 *                         (simplex_tree_check_device) on the default GPU instead of the reference's O(N^3) pass.
 *   output_triangulation  :170-284 -- the gnuplot dumps lines.dat / points.dat / circles.dat, byte for byte the
 *                         reference's text (same "%g" formats, same leaf order, same edge order).
 *   simplex_tree_fwrite / simplex_tree_fread -- binary checkpoint of a built tree (the reference has none;
 *                         SURVEY.md section 5 "checkpoint / resume", 8(f) row 2), following the
 *                         gsl_matrix_fwrite / _fread conventions (block/fwrite_source.c:21-52: native byte
 *                         order, GSL_EFAILED on a short read/write).
 */
#include "gsl_sinterp.h"
#include <math.h>
#include <stdint.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define NV 3

/* ---------------------------------------------------------------------- */
void check_leaf_nodes(simplex_tree *tree, void (*fn)(simplex_tree *, simplex_index))
{
  if (!tree || tree->n_simplexes < 1) return;
  simplex_index leaf = 0;
  while (!LEAF(leaf)) leaf = LINK(leaf, 0);                     /* :124-128 */
  const int n = tree->n_simplexes;
  unsigned char *seen = (unsigned char *)calloc((size_t)n, 1);
  /* frame = (node, next link to look at); depth <= number of leaves */
  int *stack_node = (int *)malloc((size_t)n * sizeof(int));
  unsigned char *stack_next = (unsigned char *)malloc((size_t)n);
  if (!seen || !stack_node || !stack_next) {
    free(seen); free(stack_node); free(stack_next);
    gsl_error("check_leaf_nodes: out of memory", __FILE__, __LINE__, GSL_ENOMEM);
    return;
  }
  int top = 0;
  stack_node[0] = leaf; stack_next[0] = 0;
  seen[leaf] = 1;
  if (fn) fn(tree, leaf);
  while (top >= 0) {
    const int node = stack_node[top];
    if (stack_next[top] >= NV) { top--; continue; }
    const int i = stack_next[top]++;
    const simplex_index nb = LINK(node, i);
    if (!nb || nb < 0 || nb >= n || seen[nb]) continue;         /* :113 recurse only into unseen neighbours */
    if (!LEAF(nb)) {                                            /* :73 assert(LEAF(node)) */
      gsl_error("check_leaf_nodes: a leaf's neighbour is not a leaf", __FILE__, __LINE__, GSL_ESANITY);
      continue;
    }
    seen[nb] = 1;
    if (fn) fn(tree, nb);
    top++;
    stack_node[top] = nb; stack_next[top] = 0;
  }
  free(seen); free(stack_node); free(stack_next);
}

int check_delaunay(simplex_tree *tree, gsl_matrix *data)
{
  const char *s = getenv("GSL_SINTERP_DEVICE");
  const int ok = simplex_tree_check_device(tree, data, s ? atoi(s) : 0, NULL, NULL);
  return ok == 1;
}

/* ---------------------------------------------------------------------- */
/* output_triangulation: file-scope state like the reference's (:170 flines, fcircles, gdata, ...) */
static __thread FILE *t_flines, *t_fcircles;
static __thread gsl_matrix *t_data;
static __thread gsl_vector *t_response;
static __thread int t_standardize;

static void output_leaf(simplex_tree *tree, simplex_index node)
{
  if (t_flines) {
    for (int i = 0; i < NV; i++)
      for (int j = i + 1; j < NV; j++) {
        const int i1 = POINT(node, i), i2 = POINT(node, j);
        if (i1 < 0 || i2 < 0) continue;                         /* :184 edges to the cage are not drawn */
        const int id[2] = {i1, i2};
        for (int e = 0; e < 2; e++) {
          gsl_vector_view p = DATA_POINT(t_data, id[e]);
          const double r = t_response ? gsl_vector_get(t_response, gsl_permutation_get(tree->shuffle, (size_t)id[e])) : 0;
          for (int k = 0; k < tree->dim; k++) {
            if (t_standardize)
              fprintf(t_flines, "%g ", gsl_vector_get(tree->scale, k) * (gsl_vector_get(&p.vector, k) - gsl_vector_get(tree->shift, k)));
            else
              fprintf(t_flines, "%g ", gsl_vector_get(&p.vector, k));
          }
          fprintf(t_flines, e == 0 ? "%g\n" : "%g\n\n\n", r);   /* :218, :230 */
        }
      }
  }
  if (t_fcircles) {
    double c[2] = {0, 0}, r2 = 0;
    gsl_vector_view x0 = gsl_vector_view_array(c, 2);
    int points[NV];
    for (int i = 0; i < NV; i++) points[i] = POINT(node, i);
    /* the reference ignores the status (:240): a degenerate leaf prints whatever x0 / r2 hold -- here zeros */
    (void)calculate_hypersphere_points(tree, points, t_data, &x0.vector, &r2, tree->accel);
    fprintf(t_fcircles, "%g %g %g\n", c[0], c[1], sqrt(r2));
  }
}

void output_triangulation(simplex_tree *tree, gsl_matrix *data, gsl_vector *response, int standardize_output,
                          char lines_filename[], char points_filename[], char circles_filename[])
{
  if (!tree) return;
  t_flines = lines_filename ? fopen(lines_filename, "w") : NULL;
  t_fcircles = circles_filename ? fopen(circles_filename, "w") : NULL;
  t_standardize = standardize_output;
  if (points_filename) {
    FILE *fpoints = fopen(points_filename, "w");
    if (fpoints) {
      for (int i = 0; i < tree->n_points; i++) {
        const size_t row = gsl_permutation_get(tree->shuffle, (size_t)i);
        fprintf(fpoints, "%g %g\n",                              /* :262-271: always standardised */
                gsl_vector_get(tree->scale, 0) * (gsl_matrix_get(data, row, 0) - gsl_vector_get(tree->shift, 0)),
                gsl_vector_get(tree->scale, 1) * (gsl_matrix_get(data, row, 1) - gsl_vector_get(tree->shift, 1)));
      }
      fclose(fpoints);
    }
  }
  t_data = data;
  t_response = response;
  gsl_error_handler_t *saved = gsl_set_error_handler_off();     /* degenerate circles are reported by value, not aborted */
  check_leaf_nodes(tree, output_leaf);
  gsl_set_error_handler(saved);
  if (t_flines) fclose(t_flines);
  if (t_fcircles) fclose(t_fcircles);
  t_flines = t_fcircles = NULL;
}

/* ---------------------------------------------------------------------- */
/* binary checkpoint:  magic | version | dim | n_simplexes | n_points | max_points |
                       type[n] (int32) | pidx[3n] | links[3n] | seed_points (3x2) | shift | scale | min | max |
                       shuffle[max_points] (uint64)                                                          */
static const char TREE_MAGIC[8] = {'G', 'S', 'L', 'S', 'T', 'R', 'E', '1'};

#define IO_TRY(cond, what)                                                  \
  do {                                                                      \
    if (!(cond)) { GSL_ERROR(what, GSL_EFAILED); }                          \
  } while (0)

int simplex_tree_fwrite(FILE *stream, const simplex_tree *tree)
{
  if (!stream || !tree) GSL_ERROR("simplex_tree_fwrite: null argument", GSL_EFAULT);
  const int n = tree->n_simplexes, dim = tree->dim;
  for (int k = 0; k < n; k++)
    if (tree->simplexes[k].points != (dim + 1) * k || tree->simplexes[k].links != (dim + 1) * k)
      GSL_ERROR("simplex_tree_fwrite: unexpected node slot layout", GSL_ESANITY);
  int32_t head[6] = {1, dim, n, tree->n_points, tree->max_points, 0};
  IO_TRY(fwrite(TREE_MAGIC, 1, 8, stream) == 8, "fwrite failed");
  IO_TRY(fwrite(head, sizeof head[0], 6, stream) == 6, "fwrite failed");
  int32_t *type = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  if (!type) GSL_ERROR("simplex_tree_fwrite: out of memory", GSL_ENOMEM);
  for (int k = 0; k < n; k++) type[k] = (int32_t)tree->simplexes[k].type;
  int ok = fwrite(type, sizeof(int32_t), (size_t)n, stream) == (size_t)n;
  free(type);
  IO_TRY(ok, "fwrite failed");
  const size_t w = (size_t)(dim + 1) * (size_t)n;
  IO_TRY(fwrite(tree->pidx, sizeof(int), w, stream) == w, "fwrite failed");
  IO_TRY(fwrite(tree->links, sizeof(simplex_index), w, stream) == w, "fwrite failed");
  double geo[6 + 8];
  for (int i = 0; i < dim + 1; i++)
    for (int j = 0; j < dim; j++) geo[i * dim + j] = gsl_matrix_get(tree->seed_points, i, j);
  for (int j = 0; j < dim; j++) {
    geo[6 + j] = gsl_vector_get(tree->shift, j); geo[8 + j] = gsl_vector_get(tree->scale, j);
    geo[10 + j] = gsl_vector_get(tree->min, j); geo[12 + j] = gsl_vector_get(tree->max, j);
  }
  IO_TRY(fwrite(geo, sizeof(double), 14, stream) == 14, "fwrite failed");
  const size_t np = (size_t)(tree->max_points > 0 ? tree->max_points : 0);
  for (size_t i = 0; i < np; i++) {
    const uint64_t v = (uint64_t)tree->shuffle->data[i];
    IO_TRY(fwrite(&v, sizeof v, 1, stream) == 1, "fwrite failed");
  }
  return GSL_SUCCESS;
}

simplex_tree *simplex_tree_fread(FILE *stream)
{
  if (!stream) GSL_ERROR_NULL("simplex_tree_fread: null stream", GSL_EFAULT);
  char magic[8];
  int32_t head[6];
  if (fread(magic, 1, 8, stream) != 8 || memcmp(magic, TREE_MAGIC, 8) != 0)
    GSL_ERROR_NULL("simplex_tree_fread: not a simplex_tree checkpoint", GSL_EFAILED);
  if (fread(head, sizeof head[0], 6, stream) != 6 || head[0] != 1)
    GSL_ERROR_NULL("simplex_tree_fread: unsupported checkpoint version", GSL_EFAILED);
  const int dim = head[1], n = head[2], n_points = head[3], max_points = head[4];
  /* No bound of n in terms of the point count: 9 nodes per point is alloc's average-case preallocation
     (linear_simplex.c:63, doubled on demand), and un-shuffled or sorted inputs build far longer histories (a sorted
     parabola of 200 points: 40 287 nodes).  What guards against a corrupt 8-byte header is the stream itself: the
     node types are read first, in bounded chunks, so memory only grows with what the file really holds, and the
     node arrays are allocated once n types have arrived. */
  if (dim != 2 || n < 1 || n_points < 0 || max_points < n_points || max_points > INT_MAX / 27 || n > INT_MAX / (4 * (dim + 1)))
    GSL_ERROR_NULL("simplex_tree_fread: corrupt header", GSL_EFAILED);
  {
    /* seekable stream: the counts must fit what is left of the file (types + vertex ids + links + geometry + shuffle) */
    const long here = ftell(stream);
    if (here >= 0 && fseek(stream, 0L, SEEK_END) == 0) {
      const long end = ftell(stream);
      const long long need = 4LL * n + 8LL * (dim + 1) * n + 14 * 8 + 8LL * max_points;
      if (fseek(stream, here, SEEK_SET) != 0) GSL_ERROR_NULL("simplex_tree_fread: stream error", GSL_EFAILED);
      if (end >= here && (long long)(end - here) < need)
        GSL_ERROR_NULL("simplex_tree_fread: short or corrupt checkpoint", GSL_EFAILED);
    }
  }
  int ok = 1;
  int32_t *type = NULL;
  {
    size_t have = 0, cap = 0;
    while (ok && have < (size_t)n) {
      size_t want = (size_t)n - have;
      if (want > ((size_t)1 << 20)) want = (size_t)1 << 20;
      if (have + want > cap) {
        cap = cap ? 2 * cap : want;
        if (cap > (size_t)n) cap = (size_t)n;
        if (cap < have + want) cap = have + want;
        int32_t *grown = (int32_t *)realloc(type, cap * sizeof(int32_t));
        if (!grown) { ok = 0; break; }
        type = grown;
      }
      ok = fread(type + have, sizeof(int32_t), want, stream) == want;
      have += want;
    }
  }
  if (!ok) {
    free(type);
    GSL_ERROR_NULL("simplex_tree_fread: short or corrupt checkpoint", GSL_EFAILED);
  }
  simplex_tree *tree = simplex_tree_alloc(dim, max_points);
  if (!tree) { free(type); return NULL; }
  /* make room for n nodes (alloc preallocates 9 per point and doubles on demand, linear_simplex.c:23-46) */
  while (tree->n_simplexes < n)
    if (simplex_tree_node_alloc(tree) < 0) { free(type); simplex_tree_free(tree); return NULL; }
  const size_t w = (size_t)(dim + 1) * (size_t)n;
  ok = ok && fread(tree->pidx, sizeof(int), w, stream) == w;
  ok = ok && fread(tree->links, sizeof(simplex_index), w, stream) == w;
  double geo[14];
  ok = ok && fread(geo, sizeof(double), 14, stream) == 14;
  /* the shuffle must be a permutation (every row index exactly once), not merely in range */
  unsigned char *seen = (unsigned char *)calloc((size_t)(max_points > 0 ? max_points : 1), 1);
  ok = ok && seen != NULL;
  for (size_t i = 0; ok && i < (size_t)max_points; i++) {
    uint64_t v = 0;
    ok = fread(&v, sizeof v, 1, stream) == 1 && v < (uint64_t)max_points && !seen[v];
    if (ok) { seen[v] = 1; tree->shuffle->data[i] = (size_t)v; }
  }
  free(seen);
  if (ok) {
    for (int k = 0; k < n && ok; k++) {
      ok = type[k] >= 0 && type[k] <= 3;
      tree->simplexes[k].type = (node_type)type[k];
      for (int i = 0; i < dim + 1 && ok; i++) {
        const int v = tree->pidx[(dim + 1) * k + i], l = tree->links[(dim + 1) * k + i];
        ok = v >= -(dim + 1) && v < n_points && l >= 0 && l < n;     /* indices stay inside the arrays */
        /* children are allocated after their parent (simplex_tree_node_alloc appends), so a child link of an inner
           node points forward: this makes the DAG acyclic and every find_leaf walk finite */
        const int nch = type[k] == sub_dplus1_type ? dim + 1 : type[k] == sub_d_type ? dim : type[k] == sub_2_type ? 2 : 0;
        if (ok && i < nch) ok = l > k;
      }
    }
  }
  free(type);
  if (!ok) {
    simplex_tree_free(tree);
    GSL_ERROR_NULL("simplex_tree_fread: short or corrupt checkpoint", GSL_EFAILED);
  }
  for (int i = 0; i < dim + 1; i++)
    for (int j = 0; j < dim; j++) gsl_matrix_set(tree->seed_points, i, j, geo[i * dim + j]);
  for (int j = 0; j < dim; j++) {
    gsl_vector_set(tree->shift, j, geo[6 + j]); gsl_vector_set(tree->scale, j, geo[8 + j]);
    gsl_vector_set(tree->min, j, geo[10 + j]); gsl_vector_set(tree->max, j, geo[12 + j]);
  }
  tree->n_points = n_points;
  return tree;
}
