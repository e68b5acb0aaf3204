/*
 * sinterp.c -- (a) the batched GPU entry over a host-built simplex_tree
 * (simplex_tree_device_*) and (b) the gsl_sinterp facade.
 *
 * The facade follows the alloc / init / eval_e / eval / free convention of
 * gsl_interp (interpolation/gsl_interp.h:49-71; interpolation/interp.c:30-138):
 * alloc checks min_size (GSL_EINVAL) and allocation (GSL_ENOMEM), init
 * validates sizes, eval_e writes *s and returns a status (out of domain ->
 * NaN + GSL_EDOM, interp.c:131-135), eval raises through the GSL handler,
 * free is NULL-safe (interp.c:114-122).
 *
 * All numerical work is done by the HIP kernels behind include/gsl_sinterp_hip.h.
 * There is deliberately no CPU evaluation path here.
 */
#include "gsl_sinterp.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define HIP_TRY(call, ctx)                                                         \
  do {                                                                             \
    int _st = (call);                                                              \
    if (_st != GSL_SUCCESS) {                                                      \
      gsl_error(gsl_sinterp_hip_last_error(ctx), __FILE__, __LINE__, _st);         \
      return _st;                                                                  \
    }                                                                              \
  } while (0)

static int default_device(void)
{
  const char *s = getenv("GSL_SINTERP_DEVICE");
  return s ? atoi(s) : 0;
}

/* ======================================================================== */
/* simplex_tree_device                                                       */
/* ======================================================================== */
struct simplex_tree_device {
  gsl_sinterp_hip_ctx *ctx;
  simplex_tree *tree; /* borrowed */
  int n_nodes, n_points;
  void *d_records, *d_leaftab;
  int *d_pidx;
  double scale[2];
  int response_bound;
};

gsl_sinterp_hip_ctx *simplex_tree_device_ctx(simplex_tree_device *dev) { return dev ? dev->ctx : NULL; }

void simplex_tree_device_free(simplex_tree_device *dev)
{
  if (!dev) return;
  if (dev->ctx) {
    gsl_sinterp_hip_free(dev->ctx, dev->d_records);
    gsl_sinterp_hip_free(dev->ctx, dev->d_leaftab);
    gsl_sinterp_hip_free(dev->ctx, dev->d_pidx);
    gsl_sinterp_hip_ctx_destroy(dev->ctx);
  }
  free(dev);
}

simplex_tree_device *simplex_tree_device_alloc(simplex_tree *tree, gsl_matrix *data, int device)
{
  if (!tree || tree->dim != 2) GSL_ERROR_NULL("simplex_tree_device_alloc: need a 2-D tree", GSL_EINVAL);
  if (tree->n_points > 0 && !data) GSL_ERROR_NULL("simplex_tree_device_alloc: data matrix required", GSL_EINVAL);
  const int n = tree->n_simplexes, np = tree->n_points;
  for (int k = 0; k < n; k++)
    if (tree->simplexes[k].points != 3 * k || tree->simplexes[k].links != 3 * k)
      GSL_ERROR_NULL("simplex_tree_device_alloc: unexpected node slot layout", GSL_ESANITY);

  simplex_tree_device *dev = (simplex_tree_device *)calloc(1, sizeof *dev);
  if (!dev) GSL_ERROR_NULL("simplex_tree_device_alloc: out of memory", GSL_ENOMEM);
  dev->tree = tree; dev->n_nodes = n; dev->n_points = np;
  dev->scale[0] = gsl_vector_get(tree->scale, 0);
  dev->scale[1] = gsl_vector_get(tree->scale, 1);

  int st = gsl_sinterp_hip_ctx_create(&dev->ctx, device, NULL);
  if (st != GSL_SUCCESS) {
    free(dev);
    GSL_ERROR_NULL("simplex_tree_device_alloc: no usable HIP device (GPU path has no CPU fallback)", GSL_EFAILED);
  }

  int *h_type = (int *)malloc((size_t)n * sizeof(int));
  double *h_pts = (double *)malloc((size_t)(np > 0 ? np : 1) * 2 * sizeof(double));
  double geom[10];
  int *d_type = NULL, *d_links = NULL;
  double *d_pts = NULL;
  st = GSL_ENOMEM;
  if (h_type && h_pts) {
    for (int k = 0; k < n; k++) h_type[k] = (int)tree->simplexes[k].type;
    for (int i = 0; i < np; i++) {
      const double *row = data->data + tree->shuffle->data[i] * data->tda;
      h_pts[2 * i] = row[0]; h_pts[2 * i + 1] = row[1];
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 2; j++) geom[2 * i + j] = gsl_matrix_get(tree->seed_points, i, j);
    geom[6] = gsl_vector_get(tree->shift, 0); geom[7] = gsl_vector_get(tree->shift, 1);
    geom[8] = dev->scale[0]; geom[9] = dev->scale[1];

    gsl_sinterp_hip_ctx *c = dev->ctx;
    const size_t nb = (size_t)n * sizeof(int);
    st = gsl_sinterp_hip_malloc(c, (void **)&d_type, nb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&dev->d_pidx, 3 * nb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_links, 3 * nb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_pts, (size_t)(np > 0 ? np : 1) * 2 * sizeof(double));
    if (!st) st = gsl_sinterp_hip_malloc(c, &dev->d_records, (size_t)n * GSL_SINTERP_TREE_RECORD_BYTES);
    if (!st) st = gsl_sinterp_hip_malloc(c, &dev->d_leaftab, (size_t)n * GSL_SINTERP_TREE_LEAFTAB_BYTES);
    if (!st) st = gsl_sinterp_hip_h2d(c, d_type, h_type, nb);
    if (!st) st = gsl_sinterp_hip_h2d(c, dev->d_pidx, tree->pidx, 3 * nb);
    if (!st) st = gsl_sinterp_hip_h2d(c, d_links, tree->links, 3 * nb);
    if (!st && np > 0) st = gsl_sinterp_hip_h2d(c, d_pts, h_pts, (size_t)np * 2 * sizeof(double));
    if (!st) st = gsl_sinterp_hip_tree_pack(c, n, d_type, dev->d_pidx, d_links, np, d_pts, geom, dev->d_records);
    if (!st) st = gsl_sinterp_hip_sync(c);
    gsl_sinterp_hip_free(c, d_type);
    gsl_sinterp_hip_free(c, d_links);
    gsl_sinterp_hip_free(c, d_pts);
  }
  free(h_type); free(h_pts);
  if (st != GSL_SUCCESS) {
    gsl_error(dev->ctx ? gsl_sinterp_hip_last_error(dev->ctx) : "out of memory", __FILE__, __LINE__, st);
    simplex_tree_device_free(dev);
    return NULL;
  }
  return dev;
}

int simplex_tree_device_set_response(simplex_tree_device *dev, const gsl_vector *response)
{
  if (!dev) GSL_ERROR("simplex_tree_device_set_response: null device tree", GSL_EFAULT);
  const int np = dev->n_points;
  if (np > 0 && (!response || response->size < (size_t)np))
    GSL_ERROR("simplex_tree_device_set_response: response shorter than the point set", GSL_EBADLEN);
  double *h = (double *)malloc((size_t)(np > 0 ? np : 1) * sizeof(double));
  if (!h) GSL_ERROR("simplex_tree_device_set_response: out of memory", GSL_ENOMEM);
  for (int i = 0; i < np; i++) h[i] = gsl_vector_get(response, dev->tree->shuffle->data[i]);
  double *d_resp = NULL;
  int st = gsl_sinterp_hip_malloc(dev->ctx, (void **)&d_resp, (size_t)(np > 0 ? np : 1) * sizeof(double));
  if (!st && np > 0) st = gsl_sinterp_hip_h2d(dev->ctx, d_resp, h, (size_t)np * sizeof(double));
  if (!st) st = gsl_sinterp_hip_tree_bind(dev->ctx, dev->n_nodes, dev->d_pidx, np, d_resp, dev->d_leaftab);
  if (!st) st = gsl_sinterp_hip_sync(dev->ctx);
  gsl_sinterp_hip_free(dev->ctx, d_resp);
  free(h);
  HIP_TRY(st, dev->ctx);
  dev->response_bound = 1;
  return GSL_SUCCESS;
}

int simplex_tree_device_eval_resident(simplex_tree_device *dev, const double *d_targets, size_t m,
                                      size_t ttda, double *d_values, simplex_index *d_leaf)
{
  if (!dev) GSL_ERROR("simplex_tree_device_eval: null device tree", GSL_EFAULT);
  if (!dev->response_bound) GSL_ERROR("simplex_tree_device_eval: no response bound", GSL_EINVAL);
  HIP_TRY(gsl_sinterp_hip_bary_eval(dev->ctx, dev->n_nodes, dev->d_records, dev->d_leaftab, dev->scale,
                                    d_targets, m, ttda, d_values, d_leaf, NULL), dev->ctx);
  return GSL_SUCCESS;
}

int simplex_tree_device_eval_many(simplex_tree_device *dev, const gsl_matrix *targets,
                                  gsl_vector *values, simplex_index *leaf)
{
  if (!dev) GSL_ERROR("simplex_tree_device_eval_many: null device tree", GSL_EFAULT);
  if (!dev->response_bound) GSL_ERROR("simplex_tree_device_eval_many: no response bound", GSL_EINVAL);
  if (!targets || !values) GSL_ERROR("simplex_tree_device_eval_many: null argument", GSL_EFAULT);
  if (targets->size2 != 2) GSL_ERROR("simplex_tree_device_eval_many: targets must be M x 2", GSL_EBADLEN);
  const size_t m = targets->size1;
  if (values->size != m) GSL_ERROR("simplex_tree_device_eval_many: values length must equal target rows", GSL_EBADLEN);
  if (m == 0) return GSL_SUCCESS;

  gsl_sinterp_hip_ctx *c = dev->ctx;
  double *h_y = (double *)malloc(m * 2 * sizeof(double));
  double *h_s = (double *)malloc(m * sizeof(double));
  if (!h_y || !h_s) { free(h_y); free(h_s); GSL_ERROR("simplex_tree_device_eval_many: out of memory", GSL_ENOMEM); }
  for (size_t k = 0; k < m; k++) {
    h_y[2 * k] = targets->data[k * targets->tda];
    h_y[2 * k + 1] = targets->data[k * targets->tda + 1];
  }
  double *d_y = NULL, *d_s = NULL;
  int *d_leaf = NULL;
  long long outside = 0;
  int st = gsl_sinterp_hip_malloc(c, (void **)&d_y, m * 2 * sizeof(double));
  if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_s, m * sizeof(double));
  if (!st && leaf) st = gsl_sinterp_hip_malloc(c, (void **)&d_leaf, m * sizeof(int));
  if (!st) st = gsl_sinterp_hip_h2d(c, d_y, h_y, m * 2 * sizeof(double));
  if (!st) st = gsl_sinterp_hip_bary_eval(c, dev->n_nodes, dev->d_records, dev->d_leaftab, dev->scale,
                                          d_y, m, 2, d_s, d_leaf, &outside);
  int eval_st = st;
  if (st == GSL_EDOM) st = GSL_SUCCESS;
  if (!st) st = gsl_sinterp_hip_d2h(c, h_s, d_s, m * sizeof(double));
  if (!st && leaf) st = gsl_sinterp_hip_d2h(c, leaf, d_leaf, m * sizeof(int));
  if (!st) for (size_t k = 0; k < m; k++) gsl_vector_set(values, k, h_s[k]);
  gsl_sinterp_hip_free(c, d_y); gsl_sinterp_hip_free(c, d_s); gsl_sinterp_hip_free(c, d_leaf);
  free(h_y); free(h_s);
  HIP_TRY(st, c);
  if (eval_st == GSL_EDOM) GSL_ERROR("simplex_tree_device_eval_many: target(s) outside the caging simplex", GSL_EDOM);
  return GSL_SUCCESS;
}

/* ======================================================================== */
/* RBF types                                                                 */
/* ======================================================================== */
typedef struct {
  int kind;
  gsl_sinterp_hip_ctx *ctx;
  size_t n, dim;
  double eps;
  double *d_x; /* n x dim, packed */
  double *d_w; /* n */
} rbf_state;

static void *rbf_alloc_kind(int kind, size_t dim, size_t size)
{
  rbf_state *st = (rbf_state *)calloc(1, sizeof *st);
  if (!st) return NULL;
  st->kind = kind; st->n = size; st->dim = dim;
  return st;
}
static void *rbf_gauss_alloc(size_t dim, size_t size) { return rbf_alloc_kind(GSL_SINTERP_RBF_GAUSSIAN, dim, size); }
static void *rbf_tps_alloc(size_t dim, size_t size) { return rbf_alloc_kind(GSL_SINTERP_RBF_TPS, dim, size); }

static void rbf_free(void *vstate)
{
  rbf_state *st = (rbf_state *)vstate;
  if (!st) return;
  if (st->ctx) {
    gsl_sinterp_hip_free(st->ctx, st->d_x);
    gsl_sinterp_hip_free(st->ctx, st->d_w);
    gsl_sinterp_hip_ctx_destroy(st->ctx);
  }
  free(st);
}

static int rbf_init(gsl_sinterp *interp, const gsl_matrix *x, const gsl_vector *f)
{
  rbf_state *st = (rbf_state *)interp->state;
  const size_t n = st->n, dim = st->dim;
  if (!st->ctx) {
    if (gsl_sinterp_hip_ctx_create(&st->ctx, interp->device, NULL) != GSL_SUCCESS)
      GSL_ERROR("gsl_sinterp_init: no usable HIP device (GPU path has no CPU fallback)", GSL_EFAILED);
  }
  gsl_sinterp_hip_ctx *c = st->ctx;
  st->eps = interp->shape > 0 ? interp->shape : 2.0 * pow((double)n, 1.0 / (double)dim);

  double *h_x = (double *)malloc(n * dim * sizeof(double));
  double *h_f = (double *)malloc(n * sizeof(double));
  if (!h_x || !h_f) { free(h_x); free(h_f); GSL_ERROR("gsl_sinterp_init: out of memory", GSL_ENOMEM); }
  for (size_t i = 0; i < n; i++) {
    for (size_t cdim = 0; cdim < dim; cdim++) h_x[i * dim + cdim] = x->data[i * x->tda + cdim];
    h_f[i] = gsl_vector_get(f, i);
  }
  gsl_sinterp_hip_free(c, st->d_x); gsl_sinterp_hip_free(c, st->d_w);
  st->d_x = st->d_w = NULL;
  double *d_phi = NULL;
  int route = 0;
  int s = gsl_sinterp_hip_malloc(c, (void **)&st->d_x, n * dim * sizeof(double));
  if (!s) s = gsl_sinterp_hip_malloc(c, (void **)&st->d_w, n * sizeof(double));
  if (!s) s = gsl_sinterp_hip_malloc(c, (void **)&d_phi, n * n * sizeof(double));
  if (!s) s = gsl_sinterp_hip_h2d(c, st->d_x, h_x, n * dim * sizeof(double));
  if (!s) s = gsl_sinterp_hip_h2d(c, st->d_w, h_f, n * sizeof(double));
  /* fill + dense solve on the device: Cholesky (Gaussian), shifted-SPD Cholesky with a
     Woodbury correction or pivoted LU (thin-plate spline) -- csrc/hip/solve.hip */
  if (!s) s = gsl_sinterp_hip_rbf_solve(c, st->kind, st->eps, st->d_x, n, (int)dim, dim, d_phi, n, st->d_w, &route);
  if (!s) s = gsl_sinterp_hip_sync(c);
  gsl_sinterp_hip_free(c, d_phi);
  free(h_x); free(h_f);
  if (s == GSL_EDOM) GSL_ERROR("gsl_sinterp_init: kernel matrix is not positive definite", GSL_EDOM);
  HIP_TRY(s, c);
  return GSL_SUCCESS;
}

static int rbf_eval_resident(const gsl_sinterp *interp, const double *d_y, size_t m, size_t ytda,
                             double *d_s, int *d_leaf)
{
  (void)d_leaf;
  const rbf_state *st = (const rbf_state *)interp->state;
  if (!st->d_w) GSL_ERROR("gsl_sinterp_eval: interpolant not initialised", GSL_EINVAL);
  HIP_TRY(gsl_sinterp_hip_rbf_eval(st->ctx, st->kind, st->eps, st->d_x, st->n, (int)st->dim, st->dim,
                                   st->d_w, d_y, m, ytda, d_s), st->ctx);
  return GSL_SUCCESS;
}

static int rbf_eval_many(const gsl_sinterp *interp, const gsl_matrix *y, gsl_vector *sv, int *leaf)
{
  const rbf_state *st = (const rbf_state *)interp->state;
  if (!st->d_w) GSL_ERROR("gsl_sinterp_eval_many: interpolant not initialised", GSL_EINVAL);
  const size_t m = y->size1, dim = st->dim;
  if (m == 0) return GSL_SUCCESS;
  gsl_sinterp_hip_ctx *c = st->ctx;
  double *h_y = (double *)malloc(m * dim * sizeof(double));
  double *h_s = (double *)malloc(m * sizeof(double));
  if (!h_y || !h_s) { free(h_y); free(h_s); GSL_ERROR("gsl_sinterp_eval_many: out of memory", GSL_ENOMEM); }
  for (size_t k = 0; k < m; k++)
    for (size_t cdim = 0; cdim < dim; cdim++) h_y[k * dim + cdim] = y->data[k * y->tda + cdim];
  double *d_y = NULL, *d_s = NULL;
  int s = gsl_sinterp_hip_malloc(c, (void **)&d_y, m * dim * sizeof(double));
  if (!s) s = gsl_sinterp_hip_malloc(c, (void **)&d_s, m * sizeof(double));
  if (!s) s = gsl_sinterp_hip_h2d(c, d_y, h_y, m * dim * sizeof(double));
  if (!s) s = gsl_sinterp_hip_rbf_eval(c, st->kind, st->eps, st->d_x, st->n, (int)dim, dim, st->d_w, d_y, m, dim, d_s);
  if (!s) s = gsl_sinterp_hip_d2h(c, h_s, d_s, m * sizeof(double));
  if (!s) for (size_t k = 0; k < m; k++) gsl_vector_set(sv, k, h_s[k]);
  if (!s && leaf) for (size_t k = 0; k < m; k++) leaf[k] = -1;
  gsl_sinterp_hip_free(c, d_y); gsl_sinterp_hip_free(c, d_s);
  free(h_y); free(h_s);
  HIP_TRY(s, c);
  return GSL_SUCCESS;
}

/* ======================================================================== */
/* linear simplex (barycentric) type                                         */
/* ======================================================================== */
typedef struct {
  size_t n;
  simplex_tree *tree;
  simplex_tree_device *dev;
  gsl_matrix *x; /* private copy of the centres: the tree keeps pointers into it */
} simplex_state;

static void *simplex_alloc(size_t dim, size_t size)
{
  if (dim != 2) return NULL;
  simplex_state *st = (simplex_state *)calloc(1, sizeof *st);
  if (st) st->n = size;
  return st;
}

static void simplex_free(void *vstate)
{
  simplex_state *st = (simplex_state *)vstate;
  if (!st) return;
  simplex_tree_device_free(st->dev);
  simplex_tree_free(st->tree);
  gsl_matrix_free(st->x);
  free(st);
}

static int simplex_init(gsl_sinterp *interp, const gsl_matrix *x, const gsl_vector *f)
{
  simplex_state *st = (simplex_state *)interp->state;
  simplex_tree_device_free(st->dev); st->dev = NULL;
  simplex_tree_free(st->tree); st->tree = NULL;
  gsl_matrix_free(st->x);
  st->x = gsl_matrix_alloc(st->n, 2);
  if (!st->x) return GSL_ENOMEM;
  for (size_t i = 0; i < st->n; i++) {
    gsl_matrix_set(st->x, i, 0, x->data[i * x->tda]);
    gsl_matrix_set(st->x, i, 1, x->data[i * x->tda + 1]);
  }
  st->tree = simplex_tree_alloc(2, (int)st->n);
  if (!st->tree) return GSL_ENOMEM;
  int s = simplex_tree_init(st->tree, st->x, NULL, NULL, interp->init_flags, interp->rng);
  if (s != GSL_SUCCESS) return s;
  st->dev = simplex_tree_device_alloc(st->tree, st->x, interp->device);
  if (!st->dev) return GSL_EFAILED;
  return simplex_tree_device_set_response(st->dev, f);
}

static int simplex_eval_many(const gsl_sinterp *interp, const gsl_matrix *y, gsl_vector *s, int *leaf)
{
  const simplex_state *st = (const simplex_state *)interp->state;
  if (!st->dev) GSL_ERROR("gsl_sinterp_eval_many: interpolant not initialised", GSL_EINVAL);
  return simplex_tree_device_eval_many(st->dev, y, s, leaf);
}

static int simplex_eval_resident(const gsl_sinterp *interp, const double *d_y, size_t m, size_t ytda,
                                 double *d_s, int *d_leaf)
{
  const simplex_state *st = (const simplex_state *)interp->state;
  if (!st->dev) GSL_ERROR("gsl_sinterp_eval_resident: interpolant not initialised", GSL_EINVAL);
  return simplex_tree_device_eval_resident(st->dev, d_y, m, ytda, d_s, d_leaf);
}

/* ======================================================================== */
/* type table + generic entry points                                         */
/* ======================================================================== */
static const gsl_sinterp_type gauss_type = {"rbf-gaussian", 1, &rbf_gauss_alloc, &rbf_init, &rbf_eval_many, &rbf_eval_resident, &rbf_free};
static const gsl_sinterp_type tps_type = {"rbf-thin-plate-spline", 1, &rbf_tps_alloc, &rbf_init, &rbf_eval_many, &rbf_eval_resident, &rbf_free};
static const gsl_sinterp_type simplex_type = {"linear-simplex", 3, &simplex_alloc, &simplex_init, &simplex_eval_many, &simplex_eval_resident, &simplex_free};
const gsl_sinterp_type *gsl_sinterp_rbf_gaussian = &gauss_type;
const gsl_sinterp_type *gsl_sinterp_rbf_tps = &tps_type;
const gsl_sinterp_type *gsl_sinterp_linear_simplex = &simplex_type;

gsl_sinterp *gsl_sinterp_alloc(const gsl_sinterp_type *T, size_t dim, size_t size)
{
  if (!T) GSL_ERROR_NULL("gsl_sinterp_alloc: null type", GSL_EFAULT);
  if (size < T->min_size)
    GSL_ERROR_NULL("insufficient number of points for interpolation type", GSL_EINVAL);
  if (dim < 1 || dim > 3) GSL_ERROR_NULL("gsl_sinterp_alloc: dim must be 1, 2 or 3", GSL_EINVAL);
  if (T == &simplex_type && dim != 2)
    GSL_ERROR_NULL("gsl_sinterp_alloc: linear-simplex supports dim = 2 only", GSL_EUNIMPL);
  gsl_sinterp *interp = (gsl_sinterp *)calloc(1, sizeof *interp);
  if (!interp) GSL_ERROR_NULL("failed to allocate space for sinterp struct", GSL_ENOMEM);
  interp->type = T; interp->dim = dim; interp->size = size;
  interp->device = default_device();
  interp->shape = 0.0; interp->init_flags = SIMPLEX_TREE_DEFAULT; interp->rng = NULL;
  interp->state = T->alloc(dim, size);
  if (!interp->state) {
    free(interp);
    GSL_ERROR_NULL("failed to allocate space for sinterp state", GSL_ENOMEM);
  }
  return interp;
}

int gsl_sinterp_set_device(gsl_sinterp *interp, int device)
{
  if (!interp) GSL_ERROR("gsl_sinterp_set_device: null interpolant", GSL_EFAULT);
  if (device < 0) GSL_ERROR("gsl_sinterp_set_device: negative ordinal", GSL_EINVAL);
  interp->device = device;
  return GSL_SUCCESS;
}

int gsl_sinterp_set_shape(gsl_sinterp *interp, double eps)
{
  if (!interp) GSL_ERROR("gsl_sinterp_set_shape: null interpolant", GSL_EFAULT);
  interp->shape = eps;
  return GSL_SUCCESS;
}

int gsl_sinterp_set_tree_options(gsl_sinterp *interp, int init_flags, gsl_rng *rng)
{
  if (!interp) GSL_ERROR("gsl_sinterp_set_tree_options: null interpolant", GSL_EFAULT);
  interp->init_flags = init_flags; interp->rng = rng;
  return GSL_SUCCESS;
}

int gsl_sinterp_init(gsl_sinterp *interp, const gsl_matrix *x, const gsl_vector *f)
{
  if (!interp || !x || !f) GSL_ERROR("gsl_sinterp_init: null argument", GSL_EFAULT);
  if (x->size1 != interp->size || f->size != interp->size)
    GSL_ERROR("data must match size of interpolation object", GSL_EINVAL);
  if (x->size2 != interp->dim)
    GSL_ERROR("centre matrix must have dim columns", GSL_EINVAL);
  return interp->type->init(interp, x, f);
}

const char *gsl_sinterp_name(const gsl_sinterp *interp) { return interp->type->name; }
unsigned int gsl_sinterp_min_size(const gsl_sinterp *interp) { return interp->type->min_size; }

int gsl_sinterp_eval_many(const gsl_sinterp *interp, const gsl_matrix *y, gsl_vector *s, int *leaf)
{
  if (!interp || !y || !s) GSL_ERROR("gsl_sinterp_eval_many: null argument", GSL_EFAULT);
  if (y->size2 != interp->dim) GSL_ERROR("target matrix must have dim columns", GSL_EBADLEN);
  if (s->size != y->size1) GSL_ERROR("output length must equal the number of targets", GSL_EBADLEN);
  return interp->type->eval_many(interp, y, s, leaf);
}

int gsl_sinterp_eval_resident(const gsl_sinterp *interp, const double *d_y, size_t m, size_t ytda,
                              double *d_s, int *d_leaf)
{
  if (!interp) GSL_ERROR("gsl_sinterp_eval_resident: null interpolant", GSL_EFAULT);
  return interp->type->eval_resident(interp, d_y, m, ytda, d_s, d_leaf);
}

int gsl_sinterp_eval_e(const gsl_sinterp *interp, const gsl_vector *y, double *s)
{
  if (!interp || !y || !s) GSL_ERROR("gsl_sinterp_eval_e: null argument", GSL_EFAULT);
  if (y->size != interp->dim) { *s = GSL_NAN; GSL_ERROR("target must have dim components", GSL_EBADLEN); }
  double yy[3], out = GSL_NAN;
  for (size_t c = 0; c < interp->dim; c++) yy[c] = gsl_vector_get(y, c);
  gsl_matrix_view Y = gsl_matrix_view_array(yy, 1, interp->dim);
  gsl_vector_view S = gsl_vector_view_array(&out, 1);
  /* like gsl_interp_eval_e: report EDOM by status + NaN, never through the handler */
  gsl_error_handler_t *saved = gsl_set_error_handler_off();
  int st = interp->type->eval_many(interp, &Y.matrix, &S.vector, NULL);
  gsl_set_error_handler(saved);
  *s = (st == GSL_SUCCESS) ? out : GSL_NAN;
  if (st != GSL_SUCCESS && st != GSL_EDOM) GSL_ERROR("gsl_sinterp_eval_e: evaluation failed", st);
  return st;
}

double gsl_sinterp_eval(const gsl_sinterp *interp, const gsl_vector *y)
{
  double s;
  int st = gsl_sinterp_eval_e(interp, y, &s);
  if (st != GSL_SUCCESS) GSL_ERROR_VAL("interpolation error", st, GSL_NAN);
  return s;
}

int gsl_sinterp_get_weights(const gsl_sinterp *interp, gsl_vector *w)
{
  if (!interp || !w) GSL_ERROR("gsl_sinterp_get_weights: null argument", GSL_EFAULT);
  if (interp->type == &simplex_type) GSL_ERROR("gsl_sinterp_get_weights: not an RBF interpolant", GSL_EINVAL);
  const rbf_state *st = (const rbf_state *)interp->state;
  if (!st->d_w) GSL_ERROR("gsl_sinterp_get_weights: interpolant not initialised", GSL_EINVAL);
  if (w->size != st->n) GSL_ERROR("gsl_sinterp_get_weights: wrong length", GSL_EBADLEN);
  double *h = (double *)malloc(st->n * sizeof(double));
  if (!h) GSL_ERROR("gsl_sinterp_get_weights: out of memory", GSL_ENOMEM);
  int s = gsl_sinterp_hip_d2h(st->ctx, h, st->d_w, st->n * sizeof(double));
  if (!s) for (size_t i = 0; i < st->n; i++) gsl_vector_set(w, i, h[i]);
  free(h);
  HIP_TRY(s, st->ctx);
  return GSL_SUCCESS;
}

void gsl_sinterp_free(gsl_sinterp *interp)
{
  if (!interp) return;
  if (interp->type->free) interp->type->free(interp->state);
  free(interp);
}
