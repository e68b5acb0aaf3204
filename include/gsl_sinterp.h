/*
 * gsl_sinterp.h -- host-side C API of the MI355X-native scattered-data
 * interpolation library (libgsl_sinterp.so).
 *
 * Part 1 keeps the reference's existing scattered-interpolation symbols
 * (interpolation/linear_simplex.h:105-179, interpolation/edge_flip.h:42-47)
 * with their signatures, flags, `simplex_tree` field order and accessor macros,
 * so a program written against the reference recompiles against this header.
 * They run on the host, exactly like the reference: the Delaunay history DAG is
 * built once on the CPU (2-D).
 *
 * Part 2 is the batched, GPU-only entry the reference lacks
 * (simplex_tree_device_*): the DAG is mirrored into HBM and M targets are
 * located + interpolated by one kernel.
 *
 * Part 3 is the gsl_sinterp facade shaped like gsl_interp
 * (interpolation/gsl_interp.h:49-71, interpolation/interp.c:30-138):
 * alloc(type, dim, n) / init / eval_e / eval / eval_many / free with three
 * types: Gaussian RBF, thin-plate-spline RBF, linear simplex (barycentric).
 *
 * There is no CPU fallback for parts 2 and 3: without a usable gfx950 device
 * they fail with GSL_EFAILED through the GSL error handler.
 */
#ifndef GSL_SINTERP_H
#define GSL_SINTERP_H

#include "gsl_sinterp_compat.h"
#include "gsl_sinterp_hip.h"
#include <assert.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================== */
/* Part 1: reference simplex_tree API (host)                                 */
/* ======================================================================== */
typedef int simplex_index;

typedef enum { leaf_type = 0, sub_dplus1_type, sub_d_type, sub_2_type } node_type;

typedef struct simplex_tree_node_struct {
  int points;          /* offset of this node's vertex ids in pidx[]  */
  simplex_index links; /* offset of this node's links in links[]      */
  node_type type : 2;
} simplex_tree_node;

typedef struct {
  gsl_matrix *simplex_matrix;
  gsl_permutation *perm;
  gsl_vector *coords;
  simplex_index current_simplex;
} simplex_tree_accel;

typedef struct simplex_tree_struct {
  int n_simplexes, max_simplexes;
  simplex_tree_node *simplexes;
  int n_pidx, max_pidx;
  int *pidx;
  int n_links, max_links;
  simplex_index *links;
  gsl_matrix *seed_points;
  int n_points;
  int max_points;
  int dim;
  gsl_vector *shift;
  gsl_vector *scale;
  gsl_vector *min;
  gsl_vector *max;
  gsl_permutation *shuffle;
  simplex_tree_accel *accel;
  simplex_index *new_simplexes;
  simplex_index *old_neighbors1;
  simplex_index *old_neighbors2;
  int *left_out;
  int *tmp_points1;
  gsl_vector *tmp_vec1, *tmp_vec2;
  gsl_matrix *tmp_mat;
} simplex_tree;

/* accessor macros with the reference's names; they expect a variable `tree` */
#define SIMP(I) (&(tree->simplexes[(I)]))
#define LINK(NODE, I) (tree->links[(I) + SIMP(NODE)->links])
#define SLINK(NODE, I) (SIMP(LINK(NODE, I)))
#define POINT(NODE, I) (tree->pidx[(I) + SIMP(NODE)->points])
#define LEAF(NODE) (leaf_type == SIMP(NODE)->type)

/* N_CHILDREN / DATA_POINT / FIND of the reference header (linear_simplex.h:67-102): number of
   children by node type, the vertex behind a vertex id (cage seed row when negative, data row
   shuffle[id] otherwise), and the "first index satisfying a predicate" loop. */
#define N_CHILDREN(NODE) _n_children(tree, NODE)
#define DATA_POINT(DATA, POINT) _data_point(tree, (DATA), (POINT))
#define FIND(VAR, PRED, ...)                                                   \
  for (VAR = 0; VAR < tree->dim + 1; VAR++) {                                  \
    if (PRED) break;                                                           \
  }                                                                            \
  assert(("Couldn't satisfy predicate: ", PRED, "" __VA_ARGS__ "", VAR < tree->dim + 1));

static inline int _n_children(simplex_tree *tree, simplex_index node)
{
  const node_type t = SIMP(node)->type;
  return t == sub_dplus1_type ? tree->dim + 1 : t == sub_d_type ? tree->dim : t == sub_2_type ? 2 : 0;
}

static inline gsl_vector_view _data_point(simplex_tree *tree, gsl_matrix *data, int point)
{
  return point < 0 ? gsl_matrix_row(tree->seed_points, (size_t)(-point - 1))
                   : gsl_matrix_row(data, gsl_permutation_get(tree->shuffle, (size_t)point));
}

#define SIMPLEX_TREE_DEFAULT 0
#define SIMPLEX_TREE_NOSTANDARDIZE (1 << 0)
#define SIMPLEX_TREE_ISOSCALE (1 << 1)

simplex_index simplex_tree_node_alloc(simplex_tree *tree);
simplex_tree *simplex_tree_alloc(int dim, int n_points);
int simplex_tree_init(simplex_tree *tree, gsl_matrix *data, gsl_vector *min, gsl_vector *max,
                      int init_flags, gsl_rng *rng);
void simplex_tree_free(simplex_tree *tree);
simplex_tree_accel *simplex_tree_accel_alloc(int dim);
void simplex_tree_accel_free(simplex_tree_accel *accel);
int point_in_simplex(simplex_tree *tree, simplex_index node, int point);
simplex_index find_leaf(simplex_tree *tree, gsl_matrix *data, gsl_vector *point,
                        simplex_tree_accel *accel);
simplex_index _find_leaf(simplex_tree *tree, simplex_index node, gsl_matrix *data,
                         gsl_vector *point, simplex_tree_accel *accel);
int insert_point(simplex_tree *tree, simplex_index leaf, gsl_matrix *data, gsl_vector *point,
                 simplex_tree_accel *accel);
int in_hypersphere(simplex_tree *tree, simplex_index node, gsl_matrix *data, int idx,
                   simplex_tree_accel *accel);
int in_hypersphere_points(simplex_tree *tree, int *points, gsl_matrix *data, int idx,
                          simplex_tree_accel *accel);
int calculate_hypersphere(simplex_tree *tree, simplex_index node, gsl_matrix *data,
                          gsl_vector *x0, double *r2, simplex_tree_accel *accel);
int calculate_hypersphere_points(simplex_tree *tree, int *points, gsl_matrix *data,
                                 gsl_vector *x0, double *r2, simplex_tree_accel *accel);
int calculate_bary_coords(simplex_tree *tree, simplex_index node, gsl_matrix *data,
                          gsl_vector *point, simplex_tree_accel *accel);
int contains_point(simplex_tree *tree, simplex_index node, gsl_matrix *data, gsl_vector *point,
                   simplex_tree_accel *accel);
double interp_point(simplex_tree *tree, simplex_index leaf, gsl_matrix *data,
                    gsl_vector *response, gsl_vector *point, simplex_tree_accel *accel);
int delaunay(simplex_tree *tree, simplex_index leaf, gsl_matrix *data, int face,
             simplex_tree_accel *accel);

/* ======================================================================== */
/* Part 2: batched GPU evaluation over a built tree                          */
/* ======================================================================== */
typedef struct simplex_tree_device simplex_tree_device;

/* Mirror `tree` (built over `data`) into the HBM of `device` (ordinal). */
simplex_tree_device *simplex_tree_device_alloc(simplex_tree *tree, gsl_matrix *data, int device);
void simplex_tree_device_free(simplex_tree_device *dev);
/* Bind the response column used by the following eval calls. */
int simplex_tree_device_set_response(simplex_tree_device *dev, const gsl_vector *response);
/* find_leaf + interp_point for every row of `targets` (M x 2, tda honoured).
   `leaf` may be NULL.  Rows outside the cage give leaf -1 / value NaN and the
   call returns GSL_EDOM (no abort). */
int simplex_tree_device_eval_many(simplex_tree_device *dev, const gsl_matrix *targets,
                                  gsl_vector *values, simplex_index *leaf);
/* Same, with targets / outputs already resident in HBM. */
int simplex_tree_device_eval_resident(simplex_tree_device *dev, const double *d_targets, size_t m,
                                      size_t ttda, double *d_values, simplex_index *d_leaf);
gsl_sinterp_hip_ctx *simplex_tree_device_ctx(simplex_tree_device *dev);
/* Mirror on several GPUs: the raw DAG arrays are replicated with one broadcast per array, every member
   packs its own records; eval_many shards the targets (eval_resident uses the first device). */
simplex_tree_device *simplex_tree_device_alloc_multi(simplex_tree *tree, gsl_matrix *data, const int *devices,
                                                     int n_devices);
int simplex_tree_device_n_devices(const simplex_tree_device *dev);
const char *simplex_tree_device_transport(const simplex_tree_device *dev);   /* "rccl" | "peer-copy" | "none" */

/* The reference's traversal / dump entries (interpolation/linear_simplex_integrity_check.h:11-21), same signatures:
   check_leaf_nodes walks the leaf adjacency depth first from the first leaf and applies fn to every leaf (the
   reference's order); check_delaunay returns 1 when the device-side checks pass; output_triangulation writes
   the gnuplot files of :246-284 (any name may be NULL). */
void check_leaf_nodes(simplex_tree *tree, void (*fn)(simplex_tree *, simplex_index));
int check_delaunay(simplex_tree *tree, gsl_matrix *data);
void output_triangulation(simplex_tree *tree, gsl_matrix *data, gsl_vector *response, int standardize_output,
                          char lines_filename[], char points_filename[], char circles_filename[]);
/* binary checkpoint of a built tree (native byte order; GSL_EFAILED on a short or corrupt stream) */
int simplex_tree_fwrite(FILE *stream, const simplex_tree *tree);
simplex_tree *simplex_tree_fread(FILE *stream);

/* check_leaf_nodes + check_delaunay of the reference (interpolation/linear_simplex_integrity_check.c:121-168;
   there an O(N^3) debug pass after every insertion) as two GPU kernels over the finished tree: O(leaves) and
   O(leaves x N).  Returns 1 when every predicate holds -- check_delaunay's own return convention (:162-168) --
   0 when a violation was found (counts in *leaf_violations / *delaunay_violations, either may be NULL), and a
   negative value (-GSL status, raised through the handler) when the check could not run. */
int simplex_tree_check_device(simplex_tree *tree, gsl_matrix *data, int device, long long *leaf_violations,
                              long long *delaunay_violations);

/* ======================================================================== */
/* Part 2b: imported triangulations (no history DAG)                          */
/* ======================================================================== */
/* The reference's README:28-31 lists "import triangulations from QHull / CGAL" as future work.  Such a triangulation
   comes as arrays: `triangles` [3 n] vertex ids = rows of `points`, `neighbours` [3 n] = triangle across the edge
   OPPOSITE vertex k of the triangle, -1 on the hull (the convention of QHull's neighbours and of the leaves'
   links in interpolation/linear_simplex.h:62-63).  neighbours == NULL: derived from the triangles (edge matching).
   Evaluation is the barycentric interpolation of interp_point (linear_simplex.c:678-711) in a triangle that contains
   the target under the closed rule of contains_point (:653-676), found on the GPU by a grid seed + a walk over the
   neighbour links.  A target outside the triangulation gives index -1, value NaN and GSL_EDOM. */
typedef struct simplex_mesh simplex_mesh;
typedef struct simplex_mesh_device simplex_mesh_device;
simplex_mesh *simplex_mesh_import(const gsl_matrix *points, const int *triangles, const int *neighbours, size_t n_triangles);
/* the final triangulation of a built tree: its leaves without cage vertices, vertex order, neighbour links and
   standardisation kept, so that evaluation returns the bits of the DAG path wherever the containing leaf is unique */
simplex_mesh *simplex_mesh_from_tree(simplex_tree *tree, gsl_matrix *data);
void simplex_mesh_free(simplex_mesh *mesh);
size_t simplex_mesh_n_triangles(const simplex_mesh *mesh);
size_t simplex_mesh_n_points(const simplex_mesh *mesh);
const int *simplex_mesh_triangles(const simplex_mesh *mesh);   /* [3 n] */
const int *simplex_mesh_neighbours(const simplex_mesh *mesh);  /* [3 n] */
const int *simplex_mesh_tree_nodes(const simplex_mesh *mesh);  /* [n] DAG node of every triangle (from_tree), else NULL */
void simplex_mesh_geometry(const simplex_mesh *mesh, double shift[2], double scale[2]);
/* convex = 1: a boundary edge in the walking direction proves the target outside (index -1, NaN, GSL_EDOM); 0: such walks
   are resolved by an exhaustive scan (meshes with holes / concave outlines).  simplex_mesh_import decides it from the
   boundary (one closed loop without a reflex turn = convex), simplex_mesh_from_tree exports convex = 1 (the hull of a
   Delaunay triangulation); simplex_mesh_set_convex overrides, simplex_mesh_convex reads the current setting. */
void simplex_mesh_set_convex(simplex_mesh *mesh, int convex);
int simplex_mesh_convex(const simplex_mesh *mesh);
const double *simplex_mesh_points(const simplex_mesh *mesh);    /* [2 n_points], packed rows */
void simplex_mesh_bbox(const simplex_mesh *mesh, double lo[2], double hi[2]);
/* binary checkpoint of a mesh (gsl_matrix_fwrite conventions); fread re-validates ids and neighbour links */
int simplex_mesh_fwrite(FILE *stream, const simplex_mesh *mesh);
simplex_mesh *simplex_mesh_fread(FILE *stream);
simplex_mesh_device *simplex_mesh_device_alloc(const simplex_mesh *mesh, int device);
/* the mirror replicated over a device group (one broadcast of the raw arrays, every member packs its own records and
   seed grid); eval_many shards its targets like simplex_tree_device_alloc_multi's mirror */
simplex_mesh_device *simplex_mesh_device_alloc_multi(const simplex_mesh *mesh, const int *devices, int n_devices);
int simplex_mesh_device_n_devices(const simplex_mesh_device *dev);
void simplex_mesh_device_free(simplex_mesh_device *dev);
int simplex_mesh_device_set_response(simplex_mesh_device *dev, const gsl_vector *response);
int simplex_mesh_device_eval_many(simplex_mesh_device *dev, const gsl_matrix *targets, gsl_vector *values, int *triangle);
int simplex_mesh_device_eval_resident(simplex_mesh_device *dev, const double *d_targets, size_t m, size_t ttda,
                                      double *d_values, int *d_triangle);
gsl_sinterp_hip_ctx *simplex_mesh_device_ctx(simplex_mesh_device *dev);

/* ======================================================================== */
/* Part 3: gsl_sinterp facade                                                */
/* ======================================================================== */
typedef struct gsl_sinterp_struct gsl_sinterp;
#define GSL_SINTERP_MAX_DEVICES 64

typedef struct {
  const char *name;
  unsigned int min_size;
  void *(*alloc)(size_t dim, size_t size);
  int (*init)(gsl_sinterp *interp, const gsl_matrix *x, const gsl_vector *f);
  int (*eval_many)(const gsl_sinterp *interp, const gsl_matrix *y, gsl_vector *s, int *leaf);
  int (*eval_resident)(const gsl_sinterp *interp, const double *d_y, size_t m, size_t ytda,
                       double *d_s, int *d_leaf);
  void (*free)(void *state);
} gsl_sinterp_type;

struct gsl_sinterp_struct {
  const gsl_sinterp_type *type;
  size_t dim;
  size_t size;
  int device;        /* GPU ordinal; default 0 or $GSL_SINTERP_DEVICE */
  double shape;      /* Gaussian shape parameter eps; <= 0 -> 2 * size^(1/dim) */
  int init_flags;    /* SIMPLEX_TREE_* flags (linear simplex type)            */
  gsl_rng *rng;      /* insertion-order rng (linear simplex type), may be NULL */
  void *state;
  int n_devices;     /* > 1: the model is replicated over devices[] and eval_many shards its targets */
  int devices[GSL_SINTERP_MAX_DEVICES];
  int solver;        /* GSL_SINTERP_SOLVER_* (RBF types); default = by kernel class          */
  int want_rcond;    /* estimate the reciprocal condition number at init (Cholesky solvers)  */
  double rcond;      /* the estimate of the last init, NaN when none was made                */
  int route;         /* solver route the last init took (gsl_sinterp_hip_rbf_solve_ex)       */
  double nugget;     /* kriging: added to the diagonal of the covariance matrix (>= 0, default 0) */
};

extern const gsl_sinterp_type *gsl_sinterp_rbf_gaussian;
extern const gsl_sinterp_type *gsl_sinterp_rbf_tps;
/* the thin-plate spline with its affine tail: s(y) = sum_j w_j phi(|y - x_j|) + c_0 + sum_a c_a y_a with P^T w = 0
   (the (N + d + 1) saddle system; reproduces linear data exactly).  gsl_sinterp_poly reads c back. */
extern const gsl_sinterp_type *gsl_sinterp_rbf_tps_affine;
extern const gsl_sinterp_type *gsl_sinterp_rbf_wendland;    /* compactly supported C2 kernel (README:18-26 future list) */
extern const gsl_sinterp_type *gsl_sinterp_linear_simplex;
/* piecewise-linear interpolation over an IMPORTED triangulation (QHull / CGAL arrays, README:28-31 future list): set the
   triangles with gsl_sinterp_set_triangulation before gsl_sinterp_init(x, f); `leaf` of eval_many = triangle index */
extern const gsl_sinterp_type *gsl_sinterp_linear_mesh;
/* ordinary kriging with a Gaussian covariance exp(-(eps h)^2) and an optional nugget (README:24 future list):
   s(y) = mu + sum_j w_j C(|y - x_j|) with [C + nugget I, 1; 1^T, 0] [w; mu] = [f; 0].  nugget = 0 interpolates the data,
   nugget > 0 smooths (s(x_i) = f_i - nugget w_i); far from the data s -> mu.  gsl_sinterp_set_shape sets eps. */
extern const gsl_sinterp_type *gsl_sinterp_kriging;

gsl_sinterp *gsl_sinterp_alloc(const gsl_sinterp_type *T, size_t dim, size_t size);
int gsl_sinterp_set_device(gsl_sinterp *interp, int device);
/* Multi-GPU: solve on the first device, ONE broadcast of the model (RCCL over xGMI),
   targets of gsl_sinterp_eval_many sharded contiguously over the devices, each shard copied back by its
   own GPU.  set_devices(n) = ordinals 0..n-1; the environment variable GSL_SINTERP_DEVICES ("4" or
   "0,2,5") sets the default at alloc time (GSL_SINTERP_DEVICE the single-device default). */
int gsl_sinterp_set_devices(gsl_sinterp *interp, int n_devices);
int gsl_sinterp_set_device_list(gsl_sinterp *interp, const int *devices, int n_devices);
int gsl_sinterp_n_devices(const gsl_sinterp *interp);
int gsl_sinterp_set_shape(gsl_sinterp *interp, double eps);
/* Solver breadth (linalg/cholesky.c:392-537, linalg/pcholesky.c, linalg/lu.c:204): GSL_SINTERP_SOLVER_DEFAULT picks
   by kernel class (Gaussian: Cholesky; thin-plate spline: shifted-SPD Cholesky, LU as fall-back); _CHOLESKY2 the
   diagonally scaled Cholesky; _PCHOLESKY the pivoted LDL^T for semi-definite / nuggeted kernel matrices;
   _LU_REFINE pivoted LU plus one refinement step.  gsl_sinterp_set_rcond(interp, 1) makes the next init estimate
   the reciprocal condition number of the kernel matrix (Cholesky solvers; gsl_linalg_cholesky_rcond), read back
   with gsl_sinterp_rcond (GSL_EINVAL when none is available). */
int gsl_sinterp_set_nugget(gsl_sinterp *interp, double nugget);     /* kriging type only (GSL_EINVAL otherwise) */
int gsl_sinterp_mean(const gsl_sinterp *interp, double *mean);     /* the estimated mean mu of an initialised kriging interpolant */
int gsl_sinterp_poly(const gsl_sinterp *interp, gsl_vector *c);    /* c_0 .. c_dim of an initialised gsl_sinterp_rbf_tps_affine interpolant */
int gsl_sinterp_set_solver(gsl_sinterp *interp, int solver);
int gsl_sinterp_set_rcond(gsl_sinterp *interp, int want);
int gsl_sinterp_rcond(const gsl_sinterp *interp, double *rcond);
int gsl_sinterp_route(const gsl_sinterp *interp);
int gsl_sinterp_set_tree_options(gsl_sinterp *interp, int init_flags, gsl_rng *rng);
/* gsl_sinterp_linear_mesh: triangles [3 n] (vertex = row of x), neighbours [3 n] (opposite vertex k, -1 = boundary) or NULL
   (derived by edge matching); copied, validated by the next gsl_sinterp_init */
int gsl_sinterp_set_triangulation(gsl_sinterp *interp, const int *triangles, const int *neighbours, size_t n_triangles);
int gsl_sinterp_init(gsl_sinterp *interp, const gsl_matrix *x, const gsl_vector *f);
const char *gsl_sinterp_name(const gsl_sinterp *interp);
unsigned int gsl_sinterp_min_size(const gsl_sinterp *interp);
int gsl_sinterp_eval_e(const gsl_sinterp *interp, const gsl_vector *y, double *s);
double gsl_sinterp_eval(const gsl_sinterp *interp, const gsl_vector *y);
int gsl_sinterp_eval_many(const gsl_sinterp *interp, const gsl_matrix *y, gsl_vector *s, int *leaf);
int gsl_sinterp_eval_resident(const gsl_sinterp *interp, const double *d_y, size_t m, size_t ytda,
                              double *d_s, int *d_leaf);
/* Gridded front-end (interpolation/scattered_interp_example.c:175-217): evaluate on the regular grid
   x_i = min[0] + i (max[0]-min[0])/n0, y_j = min[1] + j (max[1]-min[1])/n1 (the reference's steps: range / n_grid,
   the upper bounds excluded) with n0 = grid->size1, n1 = grid->size2; grid(i, j) receives the value.  The
   targets are generated on the device; dim = 2 interpolants only.  gsl_sinterp_fprintf_grid writes the grid in
   the reference's /tmp/plot.dat form ("%g %g %g" per node, a blank line after each i). */
int gsl_sinterp_eval_grid(const gsl_sinterp *interp, const gsl_vector *min, const gsl_vector *max, gsl_matrix *grid);
int gsl_sinterp_fprintf_grid(FILE *stream, const gsl_vector *min, const gsl_vector *max, const gsl_matrix *grid);
/* Binary checkpoint of an INITIALISED interpolant (RBF: centres + solved weights; linear simplex: the built
   history DAG + data + response), gsl_matrix_fwrite / _fread conventions (native byte order, GSL_EFAILED on a
   short transfer).  fread needs an interpolant allocated with the same type, dim and size (GSL_EBADLEN
   otherwise) and leaves it ready to evaluate: nothing is solved or triangulated again. */
int gsl_sinterp_fwrite(FILE *stream, const gsl_sinterp *interp);
int gsl_sinterp_fread(FILE *stream, gsl_sinterp *interp);
/* RBF types: copy the solved weights (length size) to the host. */
int gsl_sinterp_get_weights(const gsl_sinterp *interp, gsl_vector *w);
void gsl_sinterp_free(gsl_sinterp *interp);

#ifdef __cplusplus
}
#endif
#endif
