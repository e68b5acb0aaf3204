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
#include <limits.h>
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

/* GSL_SINTERP_DEVICES: "4" = devices 0..3, or an explicit list "0,2,5" (SURVEY.md section 5, config row).
   Returns the number of entries written to list[], 0 when the variable is unset / unusable. */
#define SINTERP_MAX_DEVICES 64
static int env_device_list(int *list)
{
  const char *s = getenv("GSL_SINTERP_DEVICES");
  if (!s || !*s) return 0;
  if (!strchr(s, ',')) {
    int n = atoi(s);
    if (n < 1) return 0;
    if (n > SINTERP_MAX_DEVICES) n = SINTERP_MAX_DEVICES;
    for (int i = 0; i < n; i++) list[i] = i;
    return n;
  }
  int n = 0;
  while (*s && n < SINTERP_MAX_DEVICES) {
    char *end = NULL;
    long v = strtol(s, &end, 10);
    if (end == s) break;
    list[n++] = (int)v;
    s = (*end == ',') ? end + 1 : end;
  }
  return n;
}

/* ======================================================================== */
/* target shards over a device group (one host thread drives every member)   */
/* ======================================================================== */
typedef struct {
  gsl_sinterp_hip_group *grp;
  int n;
  /* grow-only per-member device buffers of a shard: targets, values, leaf indices */
  double *d_y[SINTERP_MAX_DEVICES], *d_s[SINTERP_MAX_DEVICES];
  int *d_leaf[SINTERP_MAX_DEVICES];
  size_t cap[SINTERP_MAX_DEVICES], cap_dim;
  /* pinned host staging for the asynchronous copies */
  void *h_stage;
  size_t h_bytes;
} shard_set;

typedef int (*shard_eval_fn)(void *state, int member, const double *d_y, size_t m, double *d_s, int *d_leaf);

static void shard_set_release(shard_set *ss)
{
  for (int i = 0; i < ss->n; i++) {
    gsl_sinterp_hip_ctx *c = gsl_sinterp_hip_group_ctx(ss->grp, i);
    gsl_sinterp_hip_free(c, ss->d_y[i]); gsl_sinterp_hip_free(c, ss->d_s[i]); gsl_sinterp_hip_free(c, ss->d_leaf[i]);
    ss->d_y[i] = ss->d_s[i] = NULL; ss->d_leaf[i] = NULL; ss->cap[i] = 0;
  }
  gsl_sinterp_hip_host_free(ss->h_stage);
  ss->h_stage = NULL; ss->h_bytes = 0;
  gsl_sinterp_hip_group_destroy(ss->grp);
  ss->grp = NULL; ss->n = 0;
}

/* Evaluate every row of y: member r takes the contiguous shard gsl_sinterp_hip_shard_bounds gives it.
   All H2D copies, sweeps and D2H copies are enqueued before the first synchronisation, so the members
   run concurrently; results come back per shard (no gather collective: each GPU copies its own shard to
   the host).  Returns the number of targets whose leaf came back negative in *n_neg. */
static int shard_eval_many(shard_set *ss, size_t dim, const gsl_matrix *y, gsl_vector *sv, int *leaf,
                           shard_eval_fn fn, void *state, int want_leaf, size_t *n_neg)
{
  const size_t m = y->size1;
  if (n_neg) *n_neg = 0;
  if (m == 0) return GSL_SUCCESS;
  const size_t o_s = m * dim * sizeof(double), o_l = o_s + m * sizeof(double), need = o_l + (want_leaf ? m * sizeof(int) : 0);
  if (need > ss->h_bytes) {
    gsl_sinterp_hip_host_free(ss->h_stage);
    ss->h_stage = NULL; ss->h_bytes = 0;
    if (gsl_sinterp_hip_host_alloc(&ss->h_stage, need) != GSL_SUCCESS) return GSL_ENOMEM;
    ss->h_bytes = need;
  }
  double *h_y = (double *)ss->h_stage, *h_s = (double *)((char *)ss->h_stage + o_s);
  int *h_l = want_leaf ? (int *)((char *)ss->h_stage + o_l) : NULL;
  const int packed = y->tda == dim;

  /* Staging is per shard: member r's rows are repacked (tda -> dim; one memcpy when the caller's matrix is already
     dense) and its H2D -> sweep -> D2H chain is enqueued before shard r+1 is touched, so the host-side repack of the
     later shards runs while the earlier members copy and compute (round 2 repacked all M rows element-wise before
     the first copy was enqueued: tens of ms of serial host time at C4's 160 MB in front of a ~0.2 ms/GPU sweep). */
  int st = GSL_SUCCESS;
  for (int r = 0; r < ss->n && !st; r++) {
    size_t first, cnt;
    gsl_sinterp_hip_shard_bounds(m, ss->n, r, &first, &cnt);
    if (!cnt) continue;
    gsl_sinterp_hip_ctx *c = gsl_sinterp_hip_group_ctx(ss->grp, r);
    if (cnt > ss->cap[r] || dim > ss->cap_dim) {
      gsl_sinterp_hip_free(c, ss->d_y[r]); gsl_sinterp_hip_free(c, ss->d_s[r]); gsl_sinterp_hip_free(c, ss->d_leaf[r]);
      ss->d_y[r] = ss->d_s[r] = NULL; ss->d_leaf[r] = NULL; ss->cap[r] = 0;
      st = gsl_sinterp_hip_malloc(c, (void **)&ss->d_y[r], cnt * 3 * sizeof(double));   /* room for any dim <= 3 */
      if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&ss->d_s[r], cnt * sizeof(double));
      if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&ss->d_leaf[r], cnt * sizeof(int));
      if (st) break;
      ss->cap[r] = cnt; ss->cap_dim = 3;
    }
    if (packed) memcpy(h_y + first * dim, y->data + first * dim, cnt * dim * sizeof(double));
    else
      for (size_t k = first; k < first + cnt; k++)
        for (size_t cc = 0; cc < dim; cc++) h_y[k * dim + cc] = y->data[k * y->tda + cc];
    st = gsl_sinterp_hip_h2d_async(c, ss->d_y[r], h_y + first * dim, cnt * dim * sizeof(double));
    if (!st) st = fn(state, r, ss->d_y[r], cnt, ss->d_s[r], want_leaf ? ss->d_leaf[r] : NULL);
    if (!st) st = gsl_sinterp_hip_d2h_async(c, h_s + first, ss->d_s[r], cnt * sizeof(double));
    if (!st && want_leaf) st = gsl_sinterp_hip_d2h_async(c, h_l + first, ss->d_leaf[r], cnt * sizeof(int));
    if (st) gsl_error(gsl_sinterp_hip_last_error(c), __FILE__, __LINE__, st);
  }
  size_t neg = 0;
  for (int r = 0; r < ss->n; r++) {                      /* always drain every member, also after a failure */
    int s2 = gsl_sinterp_hip_sync(gsl_sinterp_hip_group_ctx(ss->grp, r));
    if (!st && s2) { st = s2; gsl_error(gsl_sinterp_hip_last_error(gsl_sinterp_hip_group_ctx(ss->grp, r)), __FILE__, __LINE__, st); }
    if (st) continue;
    /* member r's shard goes back to the caller while the later members are still running */
    size_t first, cnt;
    gsl_sinterp_hip_shard_bounds(m, ss->n, r, &first, &cnt);
    if (sv->stride == 1) memcpy(sv->data + first, h_s + first, cnt * sizeof(double));
    else for (size_t k = first; k < first + cnt; k++) sv->data[k * sv->stride] = h_s[k];
    if (want_leaf)
      for (size_t k = first; k < first + cnt; k++) { neg += h_l[k] < 0; if (leaf) leaf[k] = h_l[k]; }
  }
  if (st) return st;
  if (n_neg) *n_neg = neg;
  return GSL_SUCCESS;
}

/* ======================================================================== */
/* host batches on ONE device: chunks pipelined over the copy pipe            */
/* ======================================================================== */
/* The facade's host-matrix entries (gsl_sinterp_eval_many, simplex_tree_device_eval_many) used to run
   repack -> hipMalloc -> H2D -> sweep -> D2H -> copy-out strictly in sequence, with pageable staging allocated per call.
   Here a batch is cut into chunks; chunk i is repacked into pinned staging and its H2D -> sweep -> D2H chain enqueued on
   the context's copy pipe (upload stream | context stream | download stream) while the host already repacks chunk i+1,
   so the PCIe directions and the sweep overlap and the per-call allocations are gone (grow-only buffers in the state).
   Results are bit-identical to the one-shot path: a value depends on (model, target) only. */
#define CHUNK_MIN ((size_t)1 << 19)      /* keeps every chunk on the large-batch kernels (two-level reorder from 2^18 targets) */
#define CHUNK_MAX_N 8
typedef struct {
  gsl_sinterp_hip_pipe *pipe;
  void *h_stage;                        /* pinned: targets | values | leaf */
  size_t h_bytes;
  double *d_y, *d_s;
  int *d_leaf;
  size_t cap;                           /* targets the device buffers hold (any dim <= 3) */
} chunk_set;

typedef int (*chunk_eval_fn)(void *state, const double *d_y, size_t m, double *d_s, int *d_leaf);

static void chunk_set_release(chunk_set *cs, gsl_sinterp_hip_ctx *c)
{
  if (cs->pipe) gsl_sinterp_hip_pipe_destroy(cs->pipe);
  cs->pipe = NULL;
  if (c) { gsl_sinterp_hip_free(c, cs->d_y); gsl_sinterp_hip_free(c, cs->d_s); gsl_sinterp_hip_free(c, cs->d_leaf); }
  cs->d_y = cs->d_s = NULL; cs->d_leaf = NULL; cs->cap = 0;
  gsl_sinterp_hip_host_free(cs->h_stage);
  cs->h_stage = NULL; cs->h_bytes = 0;
}

static int chunk_eval_many(chunk_set *cs, gsl_sinterp_hip_ctx *c, size_t dim, const gsl_matrix *y, gsl_vector *sv, int *leaf,
                           chunk_eval_fn fn, void *state, int want_leaf, size_t *n_neg)
{
  const size_t m = y->size1;
  if (n_neg) *n_neg = 0;
  if (m == 0) return GSL_SUCCESS;
  /* Dense caller buffers go over PCIe as they are: measured on the MI355X box (tools/pcie_probe), the runtime moves
     pageable memory at 38 GB/s (H2D) / 56 GB/s (D2H) against 30 / 22 GB/s for a single-threaded memcpy into / out of pinned
     staging -- which would then still have to be copied.  Strided matrices / vectors are repacked through pinned staging. */
  const int y_direct = y->tda == dim, s_direct = sv->stride == 1, l_direct = want_leaf && leaf != NULL;
  const int l_down = want_leaf && leaf != NULL;          /* the indices only travel when the caller asked for them */
  int st = GSL_SUCCESS;
  if (!cs->pipe) st = gsl_sinterp_hip_pipe_create(c, &cs->pipe);
  const size_t b_y = y_direct ? 0 : m * dim * sizeof(double), b_s = s_direct ? 0 : m * sizeof(double),
               b_l = 0, need = b_y + b_s + b_l;
  if (!st && need > cs->h_bytes) {
    gsl_sinterp_hip_host_free(cs->h_stage);
    cs->h_stage = NULL; cs->h_bytes = 0;
    st = gsl_sinterp_hip_host_alloc(&cs->h_stage, need);
    if (!st) cs->h_bytes = need;
  }
  if (!st && m > cs->cap) {
    gsl_sinterp_hip_free(c, cs->d_y); gsl_sinterp_hip_free(c, cs->d_s); gsl_sinterp_hip_free(c, cs->d_leaf);
    cs->d_y = cs->d_s = NULL; cs->d_leaf = NULL; cs->cap = 0;
    st = gsl_sinterp_hip_malloc(c, (void **)&cs->d_y, m * 3 * sizeof(double));
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&cs->d_s, m * sizeof(double));
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&cs->d_leaf, m * sizeof(int));
    if (!st) cs->cap = m;
  }
  if (st) { gsl_error(gsl_sinterp_hip_last_error(c), __FILE__, __LINE__, st); return st; }
  double *h_y = y_direct ? y->data : (double *)cs->h_stage;
  double *h_s = s_direct ? sv->data : (double *)((char *)cs->h_stage + b_y);
  int *h_l = l_direct ? leaf : NULL;
  size_t nch = m / CHUNK_MIN;
  if (nch < 1) nch = 1;
  if (nch > CHUNK_MAX_N) nch = CHUNK_MAX_N;
  /* chunk i: [repack] -> H2D -> sweep enqueued; the download of chunk i-1 is enqueued AFTER the sweep of chunk i, so a
     host-blocking copy (pageable memory) waits while the GPU is busy with the next chunk, not in front of it */
  size_t pf = 0, pc = 0;
  int pmark = -1;
  for (size_t ch = 0; ch <= nch && !st; ch++) {
    size_t first = 0, cnt = 0;
    if (ch < nch) gsl_sinterp_hip_shard_bounds(m, (int)nch, (int)ch, &first, &cnt);
    if (cnt) {
      if (!y_direct)
        for (size_t k = first; k < first + cnt; k++)
          for (size_t cc = 0; cc < dim; cc++) h_y[k * dim + cc] = y->data[k * y->tda + cc];
      st = gsl_sinterp_hip_pipe_upload(cs->pipe, cs->d_y + first * dim, h_y + first * dim, cnt * dim * sizeof(double));
      if (!st) st = fn(state, cs->d_y + first * dim, cnt, cs->d_s + first, want_leaf ? cs->d_leaf + first : NULL);
    }
    int mark = -1;
    if (!st && cnt) st = gsl_sinterp_hip_pipe_mark(cs->pipe, &mark);         /* "sweep of this chunk done" */
    if (!st && pc) {
      st = gsl_sinterp_hip_pipe_download(cs->pipe, pmark, h_s + pf, cs->d_s + pf, pc * sizeof(double));
      if (!st && l_down) st = gsl_sinterp_hip_pipe_download(cs->pipe, pmark, h_l + pf, cs->d_leaf + pf, pc * sizeof(int));
    }
    pf = first; pc = cnt; pmark = mark;
  }
  /* the outside-the-cage verdict: counted on the device (one 8-byte read-back, not a host pass over m indices) */
  long long neg = 0;
  if (!st && want_leaf) st = gsl_sinterp_hip_count_negative(c, cs->d_leaf, m, &neg);
  int s2 = gsl_sinterp_hip_pipe_sync(cs->pipe);            /* always drain, also after a failure */
  if (!st) st = s2;
  if (st) { gsl_error(gsl_sinterp_hip_last_error(c), __FILE__, __LINE__, st); return st; }
  if (!s_direct) for (size_t k = 0; k < m; k++) sv->data[k * sv->stride] = h_s[k];
  if (n_neg) *n_neg = (size_t)neg;
  return GSL_SUCCESS;
}

/* ======================================================================== */
/* simplex_tree_device                                                       */
/* ======================================================================== */
struct simplex_tree_device {
  gsl_sinterp_hip_ctx *ctx;        /* member 0 (the only one for a single-device mirror) */
  simplex_tree *tree; /* borrowed */
  int n_nodes, n_points;
  void *d_records, *d_leaftab;     /* member 0 */
  int *d_pidx;
  double scale[2];
  int response_bound;
  /* multi-GPU mirror (simplex_tree_device_alloc_multi): every member holds its own packed records,
     leaf table and vertex ids; ss.grp owns the contexts (ctx above aliases member 0's) */
  shard_set ss;
  void *m_records[SINTERP_MAX_DEVICES], *m_leaftab[SINTERP_MAX_DEVICES];
  int *m_pidx[SINTERP_MAX_DEVICES];
  chunk_set cs;                    /* single-device mirror: pipelined host batches */
};

gsl_sinterp_hip_ctx *simplex_tree_device_ctx(simplex_tree_device *dev) { return dev ? dev->ctx : NULL; }

void simplex_tree_device_free(simplex_tree_device *dev)
{
  if (!dev) return;
  if (dev->ss.grp) {
    for (int i = 0; i < dev->ss.n; i++) {
      gsl_sinterp_hip_ctx *c = gsl_sinterp_hip_group_ctx(dev->ss.grp, i);
      gsl_sinterp_hip_free(c, dev->m_records[i]); gsl_sinterp_hip_free(c, dev->m_leaftab[i]); gsl_sinterp_hip_free(c, dev->m_pidx[i]);
    }
    shard_set_release(&dev->ss);                   /* destroys the members' contexts, dev->ctx among them */
    free(dev);
    return;
  }
  if (dev->ctx) {
    chunk_set_release(&dev->cs, dev->ctx);
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

/* Mirror `tree` on every device of the list.  The raw DAG arrays (node types, vertex ids, links, points:
   SURVEY.md 8(e)'s ~17 MB at N = 50 000) go to member 0 over PCIe ONCE and are replicated with one
   broadcast per array (RCCL over xGMI); every member then packs its own node records and jump table,
   so all members evaluate from an identical, locally built mirror. */
simplex_tree_device *simplex_tree_device_alloc_multi(simplex_tree *tree, gsl_matrix *data, const int *devices, int n_devices)
{
  if (n_devices == 1 && devices) return simplex_tree_device_alloc(tree, data, devices[0]);
  if (!devices || n_devices < 1 || n_devices > SINTERP_MAX_DEVICES)
    GSL_ERROR_NULL("simplex_tree_device_alloc_multi: bad device list", GSL_EINVAL);
  if (!tree || tree->dim != 2) GSL_ERROR_NULL("simplex_tree_device_alloc_multi: need a 2-D tree", GSL_EINVAL);
  if (tree->n_points > 0 && !data) GSL_ERROR_NULL("simplex_tree_device_alloc_multi: data matrix required", GSL_EINVAL);
  const int n = tree->n_simplexes, np = tree->n_points;
  for (int k = 0; k < n; k++)
    if (tree->simplexes[k].points != 3 * k || tree->simplexes[k].links != 3 * k)
      GSL_ERROR_NULL("simplex_tree_device_alloc_multi: unexpected node slot layout", GSL_ESANITY);
  simplex_tree_device *dev = (simplex_tree_device *)calloc(1, sizeof *dev);
  if (!dev) GSL_ERROR_NULL("simplex_tree_device_alloc_multi: out of memory", GSL_ENOMEM);
  dev->tree = tree; dev->n_nodes = n; dev->n_points = np;
  dev->scale[0] = gsl_vector_get(tree->scale, 0);
  dev->scale[1] = gsl_vector_get(tree->scale, 1);
  if (gsl_sinterp_hip_group_create(&dev->ss.grp, devices, n_devices) != GSL_SUCCESS) {
    free(dev);
    GSL_ERROR_NULL("simplex_tree_device_alloc_multi: cannot create the device group (GPU path has no CPU fallback)", GSL_EFAILED);
  }
  dev->ss.n = n_devices;
  dev->ctx = gsl_sinterp_hip_group_ctx(dev->ss.grp, 0);

  const size_t nb = (size_t)n * sizeof(int), pb = (size_t)(np > 0 ? np : 1) * 2 * sizeof(double);
  int *h_type = (int *)malloc(nb);
  double *h_pts = (double *)malloc(pb);
  double geom[10];
  int *d_type[SINTERP_MAX_DEVICES] = {0}, *d_links[SINTERP_MAX_DEVICES] = {0};
  double *d_pts[SINTERP_MAX_DEVICES] = {0};
  int st = (h_type && h_pts) ? GSL_SUCCESS : GSL_ENOMEM;
  if (!st) {
    for (int k = 0; k < n; k++) h_type[k] = (int)tree->simplexes[k].type;
    for (int i = 0; i < np; i++) {
      const double *row = data->data + tree->shuffle->data[i] * data->tda;
      h_pts[2 * i] = row[0]; h_pts[2 * i + 1] = row[1];
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 2; j++) geom[2 * i + j] = gsl_matrix_get(tree->seed_points, i, j);
    geom[6] = gsl_vector_get(tree->shift, 0); geom[7] = gsl_vector_get(tree->shift, 1);
    geom[8] = dev->scale[0]; geom[9] = dev->scale[1];
  }
  for (int r = 0; r < n_devices && !st; r++) {
    gsl_sinterp_hip_ctx *c = gsl_sinterp_hip_group_ctx(dev->ss.grp, r);
    st = gsl_sinterp_hip_malloc(c, (void **)&d_type[r], nb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&dev->m_pidx[r], 3 * nb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_links[r], 3 * nb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_pts[r], pb);
    if (!st) st = gsl_sinterp_hip_malloc(c, &dev->m_records[r], (size_t)n * GSL_SINTERP_TREE_RECORD_BYTES);
    if (!st) st = gsl_sinterp_hip_malloc(c, &dev->m_leaftab[r], (size_t)n * GSL_SINTERP_TREE_LEAFTAB_BYTES);
  }
  if (!st) {
    gsl_sinterp_hip_ctx *c0 = dev->ctx;
    st = gsl_sinterp_hip_h2d(c0, d_type[0], h_type, nb);
    if (!st) st = gsl_sinterp_hip_h2d(c0, dev->m_pidx[0], tree->pidx, 3 * nb);
    if (!st) st = gsl_sinterp_hip_h2d(c0, d_links[0], tree->links, 3 * nb);
    if (!st && np > 0) st = gsl_sinterp_hip_h2d(c0, d_pts[0], h_pts, (size_t)np * 2 * sizeof(double));
    /* model replication: the only collective of the path */
    if (!st) st = gsl_sinterp_hip_group_broadcast(dev->ss.grp, (void *const *)d_type, nb);
    if (!st) st = gsl_sinterp_hip_group_broadcast(dev->ss.grp, (void *const *)dev->m_pidx, 3 * nb);
    if (!st) st = gsl_sinterp_hip_group_broadcast(dev->ss.grp, (void *const *)d_links, 3 * nb);
    if (!st) st = gsl_sinterp_hip_group_broadcast(dev->ss.grp, (void *const *)d_pts, pb);
  }
  for (int r = 0; r < n_devices && !st; r++)
    st = gsl_sinterp_hip_tree_pack(gsl_sinterp_hip_group_ctx(dev->ss.grp, r), n, d_type[r], dev->m_pidx[r], d_links[r], np, d_pts[r],
                                   geom, dev->m_records[r]);
  for (int r = 0; r < n_devices; r++) {
    gsl_sinterp_hip_ctx *c = gsl_sinterp_hip_group_ctx(dev->ss.grp, r);
    int s2 = gsl_sinterp_hip_sync(c);
    if (!st) st = s2;
    gsl_sinterp_hip_free(c, d_type[r]); gsl_sinterp_hip_free(c, d_links[r]); gsl_sinterp_hip_free(c, d_pts[r]);
  }
  free(h_type); free(h_pts);
  if (st != GSL_SUCCESS) {
    gsl_error(gsl_sinterp_hip_last_error(dev->ctx), __FILE__, __LINE__, st);
    simplex_tree_device_free(dev);
    return NULL;
  }
  dev->d_records = dev->m_records[0]; dev->d_leaftab = dev->m_leaftab[0]; dev->d_pidx = dev->m_pidx[0];
  return dev;
}

int simplex_tree_device_n_devices(const simplex_tree_device *dev) { return dev ? (dev->ss.grp ? dev->ss.n : 1) : 0; }
const char *simplex_tree_device_transport(const simplex_tree_device *dev)
{
  return (dev && dev->ss.grp) ? gsl_sinterp_hip_group_transport(dev->ss.grp) : "none";
}

static int simplex_shard_eval(void *state, int member, const double *d_y, size_t m, double *d_s, int *d_leaf)
{
  simplex_tree_device *dev = (simplex_tree_device *)state;
  int st = gsl_sinterp_hip_bary_eval(gsl_sinterp_hip_group_ctx(dev->ss.grp, member), dev->n_nodes, dev->m_records[member],
                                     dev->m_leaftab[member], dev->scale, d_y, m, 2, d_s, d_leaf, NULL);
  return st == GSL_EDOM ? GSL_SUCCESS : st;
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
  if (dev->ss.grp) {
    /* response (insertion order) -> member 0 -> one broadcast -> every member binds its own leaf table */
    double *d_r[SINTERP_MAX_DEVICES] = {0};
    const size_t rb = (size_t)(np > 0 ? np : 1) * sizeof(double);
    int st = GSL_SUCCESS;
    for (int r = 0; r < dev->ss.n && !st; r++) st = gsl_sinterp_hip_malloc(gsl_sinterp_hip_group_ctx(dev->ss.grp, r), (void **)&d_r[r], rb);
    if (!st && np > 0) st = gsl_sinterp_hip_h2d(dev->ctx, d_r[0], h, (size_t)np * sizeof(double));
    if (!st) st = gsl_sinterp_hip_group_broadcast(dev->ss.grp, (void *const *)d_r, rb);
    for (int r = 0; r < dev->ss.n && !st; r++)
      st = gsl_sinterp_hip_tree_bind(gsl_sinterp_hip_group_ctx(dev->ss.grp, r), dev->n_nodes, dev->m_pidx[r], np, d_r[r], dev->m_leaftab[r]);
    for (int r = 0; r < dev->ss.n; r++) {
      gsl_sinterp_hip_ctx *c = gsl_sinterp_hip_group_ctx(dev->ss.grp, r);
      int s2 = gsl_sinterp_hip_sync(c);
      if (!st) st = s2;
      gsl_sinterp_hip_free(c, d_r[r]);
    }
    free(h);
    HIP_TRY(st, dev->ctx);
    dev->response_bound = 1;
    return GSL_SUCCESS;
  }
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

static int simplex_chunk_eval(void *state, const double *d_y, size_t m, double *d_s, int *d_leaf)
{
  simplex_tree_device *dev = (simplex_tree_device *)state;
  return gsl_sinterp_hip_bary_eval(dev->ctx, dev->n_nodes, dev->d_records, dev->d_leaftab, dev->scale, d_y, m, 2, d_s, d_leaf,
                                   (long long *)NULL);     /* NULL: no host read-back, the call stays asynchronous */
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
  if (dev->ss.grp) {
    size_t outside_n = 0;
    int sst = shard_eval_many(&dev->ss, 2, targets, values, leaf, &simplex_shard_eval, dev, 1, &outside_n);
    if (sst != GSL_SUCCESS) GSL_ERROR("simplex_tree_device_eval_many: sharded evaluation failed", sst);
    if (outside_n) GSL_ERROR("simplex_tree_device_eval_many: target(s) outside the caging simplex", GSL_EDOM);
    return GSL_SUCCESS;
  }

  /* single device: chunks over the copy pipe (H2D | sweep | D2H overlap); the leaf indices always come back, they
     carry the outside-the-cage verdict (index -1, value NaN) */
  size_t outside_n = 0;
  int st = chunk_eval_many(&dev->cs, dev->ctx, 2, targets, values, leaf, &simplex_chunk_eval, dev, 1, &outside_n);
  if (st != GSL_SUCCESS) GSL_ERROR("simplex_tree_device_eval_many: evaluation failed", st);
  const int eval_st = outside_n ? GSL_EDOM : GSL_SUCCESS;
  if (eval_st == GSL_EDOM) GSL_ERROR("simplex_tree_device_eval_many: target(s) outside the caging simplex", GSL_EDOM);
  return GSL_SUCCESS;
}

/* reference: check_leaf_nodes / check_delaunay (linear_simplex_integrity_check.c:121-168) */
int simplex_tree_check_device(simplex_tree *tree, gsl_matrix *data, int device, long long *leaf_violations,
                              long long *delaunay_violations)
{
  if (leaf_violations) *leaf_violations = 0;
  if (delaunay_violations) *delaunay_violations = 0;
  if (!tree || tree->dim != 2) GSL_ERROR_VAL("simplex_tree_check_device: need a 2-D tree", GSL_EINVAL, -GSL_EINVAL);
  if (tree->n_points > 0 && !data) GSL_ERROR_VAL("simplex_tree_check_device: data matrix required", GSL_EINVAL, -GSL_EINVAL);
  const int n = tree->n_simplexes, np = tree->n_points;
  for (int k = 0; k < n; k++)
    if (tree->simplexes[k].points != 3 * k || tree->simplexes[k].links != 3 * k)
      GSL_ERROR_VAL("simplex_tree_check_device: unexpected node slot layout", GSL_ESANITY, -GSL_ESANITY);
  gsl_sinterp_hip_ctx *c = NULL;
  if (gsl_sinterp_hip_ctx_create(&c, device, NULL) != GSL_SUCCESS)
    GSL_ERROR_VAL("simplex_tree_check_device: no usable HIP device (GPU path has no CPU fallback)", GSL_EFAILED, -GSL_EFAILED);
  const size_t nb = (size_t)n * sizeof(int), pb = (size_t)(np > 0 ? np : 1) * 2 * sizeof(double);
  int *h_type = (int *)malloc(nb);
  double *h_pts = (double *)malloc(pb);
  int *d_type = NULL, *d_pidx = NULL, *d_links = NULL;
  double *d_pts = NULL, geom[10];
  long long lv = 0, dv = 0;
  int st = (h_type && h_pts) ? GSL_SUCCESS : GSL_ENOMEM;
  if (!st) {
    for (int k = 0; k < n; k++) h_type[k] = (int)tree->simplexes[k].type;
    for (int i = 0; i < np; i++) {
      const double *row = data->data + tree->shuffle->data[i] * data->tda;
      h_pts[2 * i] = row[0]; h_pts[2 * i + 1] = row[1];
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 2; j++) geom[2 * i + j] = gsl_matrix_get(tree->seed_points, i, j);
    geom[6] = gsl_vector_get(tree->shift, 0); geom[7] = gsl_vector_get(tree->shift, 1);
    geom[8] = gsl_vector_get(tree->scale, 0); geom[9] = gsl_vector_get(tree->scale, 1);
    st = gsl_sinterp_hip_malloc(c, (void **)&d_type, nb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_pidx, 3 * nb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_links, 3 * nb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_pts, pb);
    if (!st) st = gsl_sinterp_hip_h2d(c, d_type, h_type, nb);
    if (!st) st = gsl_sinterp_hip_h2d(c, d_pidx, tree->pidx, 3 * nb);
    if (!st) st = gsl_sinterp_hip_h2d(c, d_links, tree->links, 3 * nb);
    if (!st && np > 0) st = gsl_sinterp_hip_h2d(c, d_pts, h_pts, (size_t)np * 2 * sizeof(double));
    if (!st) st = gsl_sinterp_hip_tree_check(c, n, d_type, d_pidx, d_links, np, d_pts, geom, 3, &lv, &dv, NULL);
  }
  if (st) gsl_error(gsl_sinterp_hip_last_error(c), __FILE__, __LINE__, st);
  gsl_sinterp_hip_free(c, d_type); gsl_sinterp_hip_free(c, d_pidx); gsl_sinterp_hip_free(c, d_links); gsl_sinterp_hip_free(c, d_pts);
  gsl_sinterp_hip_ctx_destroy(c);
  free(h_type); free(h_pts);
  if (st) return -st;
  if (leaf_violations) *leaf_violations = lv;
  if (delaunay_violations) *delaunay_violations = dv;
  return (lv == 0 && dv == 0) ? 1 : 0;
}

/* ======================================================================== */
/* simplex_mesh_device: imported triangulations on one device or a group     */
/* ======================================================================== */
struct simplex_mesh_device {
  gsl_sinterp_hip_ctx *ctx;        /* member 0 */
  int n_tri, n_points, G, convex, n_members;
  double geom[8];                  /* shift, scale, bounding box */
  void *m_records[SINTERP_MAX_DEVICES], *m_leaftab[SINTERP_MAX_DEVICES];
  int *m_tri[SINTERP_MAX_DEVICES], *m_seed[SINTERP_MAX_DEVICES];
  shard_set ss;                    /* n_members > 1: ss.grp owns the contexts */
  chunk_set cs;                    /* n_members == 1: pipelined host batches */
  int response_bound;
};

gsl_sinterp_hip_ctx *simplex_mesh_device_ctx(simplex_mesh_device *dev) { return dev ? dev->ctx : NULL; }
int simplex_mesh_device_n_devices(const simplex_mesh_device *dev) { return dev ? dev->n_members : 0; }

static gsl_sinterp_hip_ctx *mesh_member_ctx(const simplex_mesh_device *dev, int r)
{
  return dev->ss.grp ? gsl_sinterp_hip_group_ctx(dev->ss.grp, r) : dev->ctx;
}

void simplex_mesh_device_free(simplex_mesh_device *dev)
{
  if (!dev) return;
  for (int r = 0; r < dev->n_members; r++) {
    gsl_sinterp_hip_ctx *c = mesh_member_ctx(dev, r);
    if (!c) continue;
    gsl_sinterp_hip_free(c, dev->m_records[r]); gsl_sinterp_hip_free(c, dev->m_leaftab[r]);
    gsl_sinterp_hip_free(c, dev->m_tri[r]); gsl_sinterp_hip_free(c, dev->m_seed[r]);
  }
  if (dev->ss.grp) shard_set_release(&dev->ss);      /* destroys the members' contexts */
  else if (dev->ctx) { chunk_set_release(&dev->cs, dev->ctx); gsl_sinterp_hip_ctx_destroy(dev->ctx); }
  free(dev);
}

simplex_mesh_device *simplex_mesh_device_alloc_multi(const simplex_mesh *mesh, const int *devices, int n_devices)
{
  if (!mesh || !devices) GSL_ERROR_NULL("simplex_mesh_device_alloc: null argument", GSL_EFAULT);
  if (n_devices < 1 || n_devices > SINTERP_MAX_DEVICES) GSL_ERROR_NULL("simplex_mesh_device_alloc: bad device list", GSL_EINVAL);
  simplex_mesh_device *dev = (simplex_mesh_device *)calloc(1, sizeof *dev);
  if (!dev) GSL_ERROR_NULL("simplex_mesh_device_alloc: out of memory", GSL_ENOMEM);
  const size_t nt = simplex_mesh_n_triangles(mesh), np = simplex_mesh_n_points(mesh);
  dev->n_tri = (int)nt; dev->n_points = (int)np; dev->convex = simplex_mesh_convex(mesh); dev->n_members = n_devices;
  /* about two triangles per seed cell */
  int G = (int)ceil(sqrt((double)nt / 2.0));
  dev->G = G < 1 ? 1 : (G > 2048 ? 2048 : G);
  simplex_mesh_geometry(mesh, dev->geom, dev->geom + 2);
  simplex_mesh_bbox(mesh, dev->geom + 4, dev->geom + 6);
  if (n_devices > 1) {
    if (gsl_sinterp_hip_group_create(&dev->ss.grp, devices, n_devices) != GSL_SUCCESS) {
      free(dev);
      GSL_ERROR_NULL("simplex_mesh_device_alloc: cannot create the device group (GPU path has no CPU fallback)", GSL_EFAILED);
    }
    dev->ss.n = n_devices;
    dev->ctx = gsl_sinterp_hip_group_ctx(dev->ss.grp, 0);
  } else if (gsl_sinterp_hip_ctx_create(&dev->ctx, devices[0], NULL) != GSL_SUCCESS) {
    free(dev);
    GSL_ERROR_NULL("simplex_mesh_device_alloc: no usable HIP device (GPU path has no CPU fallback)", GSL_EFAILED);
  }
  const size_t tb = 3 * nt * sizeof(int), pb = 2 * np * sizeof(double);
  int *d_nbr[SINTERP_MAX_DEVICES] = {0};
  double *d_pts[SINTERP_MAX_DEVICES] = {0};
  int st = GSL_SUCCESS;
  for (int r = 0; r < n_devices && !st; r++) {
    gsl_sinterp_hip_ctx *c = mesh_member_ctx(dev, r);
    st = gsl_sinterp_hip_malloc(c, (void **)&dev->m_tri[r], tb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_nbr[r], tb);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_pts[r], pb);
    if (!st) st = gsl_sinterp_hip_malloc(c, &dev->m_records[r], nt * GSL_SINTERP_TREE_RECORD_BYTES);
    if (!st) st = gsl_sinterp_hip_malloc(c, &dev->m_leaftab[r], nt * GSL_SINTERP_TREE_LEAFTAB_BYTES);
    if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&dev->m_seed[r], 2 * (size_t)dev->G * dev->G * sizeof(int));
  }
  if (!st) st = gsl_sinterp_hip_h2d(dev->ctx, dev->m_tri[0], simplex_mesh_triangles(mesh), tb);
  if (!st) st = gsl_sinterp_hip_h2d(dev->ctx, d_nbr[0], simplex_mesh_neighbours(mesh), tb);
  if (!st) st = gsl_sinterp_hip_h2d(dev->ctx, d_pts[0], simplex_mesh_points(mesh), pb);
  if (!st && dev->ss.grp) {                             /* model replication: the only collective of the path */
    st = gsl_sinterp_hip_group_broadcast(dev->ss.grp, (void *const *)dev->m_tri, tb);
    if (!st) st = gsl_sinterp_hip_group_broadcast(dev->ss.grp, (void *const *)d_nbr, tb);
    if (!st) st = gsl_sinterp_hip_group_broadcast(dev->ss.grp, (void *const *)d_pts, pb);
  }
  for (int r = 0; r < n_devices && !st; r++)            /* every member packs its own records and seed grid */
    st = gsl_sinterp_hip_mesh_pack(mesh_member_ctx(dev, r), dev->n_tri, dev->m_tri[r], d_nbr[r], dev->n_points, d_pts[r], dev->geom, dev->G,
                                   dev->m_records[r], dev->m_seed[r]);
  for (int r = 0; r < n_devices; r++) {
    gsl_sinterp_hip_ctx *c = mesh_member_ctx(dev, r);
    int s2 = gsl_sinterp_hip_sync(c);
    if (!st) st = s2;
    gsl_sinterp_hip_free(c, d_nbr[r]); gsl_sinterp_hip_free(c, d_pts[r]);
  }
  if (st != GSL_SUCCESS) {
    gsl_error(gsl_sinterp_hip_last_error(dev->ctx), __FILE__, __LINE__, st);
    simplex_mesh_device_free(dev);
    return NULL;
  }
  return dev;
}

simplex_mesh_device *simplex_mesh_device_alloc(const simplex_mesh *mesh, int device)
{
  return simplex_mesh_device_alloc_multi(mesh, &device, 1);
}

int simplex_mesh_device_set_response(simplex_mesh_device *dev, const gsl_vector *response)
{
  if (!dev || !response) GSL_ERROR("simplex_mesh_device_set_response: null argument", GSL_EFAULT);
  if (response->size < (size_t)dev->n_points) GSL_ERROR("simplex_mesh_device_set_response: response shorter than the point set", GSL_EBADLEN);
  const size_t np = (size_t)dev->n_points, rb = np * sizeof(double);
  double *h = (double *)malloc(rb), *d_r[SINTERP_MAX_DEVICES] = {0};
  if (!h) GSL_ERROR("simplex_mesh_device_set_response: out of memory", GSL_ENOMEM);
  for (size_t i = 0; i < np; i++) h[i] = response->data[i * response->stride];
  int st = GSL_SUCCESS;
  for (int r = 0; r < dev->n_members && !st; r++) st = gsl_sinterp_hip_malloc(mesh_member_ctx(dev, r), (void **)&d_r[r], rb);
  if (!st) st = gsl_sinterp_hip_h2d(dev->ctx, d_r[0], h, rb);
  if (!st && dev->ss.grp) st = gsl_sinterp_hip_group_broadcast(dev->ss.grp, (void *const *)d_r, rb);
  for (int r = 0; r < dev->n_members && !st; r++)
    st = gsl_sinterp_hip_tree_bind(mesh_member_ctx(dev, r), dev->n_tri, dev->m_tri[r], dev->n_points, d_r[r], dev->m_leaftab[r]);
  for (int r = 0; r < dev->n_members; r++) {
    gsl_sinterp_hip_ctx *c = mesh_member_ctx(dev, r);
    int s2 = gsl_sinterp_hip_sync(c);
    if (!st) st = s2;
    gsl_sinterp_hip_free(c, d_r[r]);
  }
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
  /* resident buffers live on ONE device: member 0 evaluates them */
  int st = gsl_sinterp_hip_mesh_eval(dev->ctx, dev->n_tri, dev->m_records[0], dev->m_leaftab[0], dev->m_seed[0], dev->G, dev->geom, dev->convex,
                                     d_targets, m, ttda, d_values, d_triangle, NULL);
  if (st) GSL_ERROR(gsl_sinterp_hip_last_error(dev->ctx), st);
  return GSL_SUCCESS;
}

static int mesh_shard_eval(void *state, int member, const double *d_y, size_t m, double *d_s, int *d_leaf)
{
  simplex_mesh_device *dev = (simplex_mesh_device *)state;
  return gsl_sinterp_hip_mesh_eval(mesh_member_ctx(dev, member), dev->n_tri, dev->m_records[member], dev->m_leaftab[member], dev->m_seed[member],
                                   dev->G, dev->geom, dev->convex, d_y, m, 2, d_s, d_leaf, (long long *)NULL);
}
static int mesh_chunk_eval(void *state, const double *d_y, size_t m, double *d_s, int *d_leaf) { return mesh_shard_eval(state, 0, d_y, m, d_s, d_leaf); }

int simplex_mesh_device_eval_many(simplex_mesh_device *dev, const gsl_matrix *targets, gsl_vector *values, int *triangle)
{
  if (!dev || !targets || !values) GSL_ERROR("simplex_mesh_device_eval_many: null argument", GSL_EFAULT);
  if (!dev->response_bound) GSL_ERROR("simplex_mesh_device_eval_many: no response bound", GSL_EINVAL);
  if (targets->size2 != 2) GSL_ERROR("simplex_mesh_device_eval_many: targets must be M x 2", GSL_EBADLEN);
  const size_t m = targets->size1;
  if (values->size != m) GSL_ERROR("simplex_mesh_device_eval_many: values length must equal target rows", GSL_EBADLEN);
  if (m == 0) return GSL_SUCCESS;
  size_t outside_n = 0;
  int st = dev->ss.grp ? shard_eval_many(&dev->ss, 2, targets, values, triangle, &mesh_shard_eval, dev, 1, &outside_n)
                       : chunk_eval_many(&dev->cs, dev->ctx, 2, targets, values, triangle, &mesh_chunk_eval, dev, 1, &outside_n);
  if (st != GSL_SUCCESS) GSL_ERROR("simplex_mesh_device_eval_many: evaluation failed", st);
  if (outside_n) GSL_ERROR("simplex_mesh_device_eval_many: target(s) outside the triangulation", GSL_EDOM);
  return GSL_SUCCESS;
}

/* ======================================================================== */
/* RBF types                                                                 */
/* ======================================================================== */
typedef struct {
  int kind;
  gsl_sinterp_hip_ctx *ctx;   /* member 0: fill + factorisation + solves run here ("replicas only" for the solve) */
  size_t n, dim;
  double eps;
  double *d_x; /* n x dim, packed  (member 0: start of the model buffer) */
  double *d_w; /* n                (member 0: model buffer + n*dim)      */
  /* device group (n_devices > 1): the model buffer [x | w] of every member; ss.grp owns the contexts */
  shard_set ss;
  double *m_model[SINTERP_MAX_DEVICES];
  /* id of the model the buffers hold (fresh after every init / fread): lets the sweep keep its per-model
     preprocessing between evaluations (gsl_sinterp_hip_rbf_eval_model) */
  unsigned long long model_id;
  /* ordinary kriging (gsl_sinterp_kriging): the kernel is the covariance, `mean` the estimated mean added by every sweep */
  int krige;
  double mean;
  /* thin-plate spline with its affine tail (gsl_sinterp_rbf_tps_affine): c_0 + sum_a c_a y_a added by every sweep */
  int affine;
  double poly[4];
  chunk_set cs;                /* single-device: pipelined host batches */
} rbf_state;

static unsigned long long next_model_id(void)
{
  static unsigned long long counter = 0;               /* the facade is single-threaded like the reference */
  return ++counter;
}

static void *rbf_alloc_kind(int kind, size_t dim, size_t size)
{
  rbf_state *st = (rbf_state *)calloc(1, sizeof *st);
  if (!st) return NULL;
  st->kind = kind; st->n = size; st->dim = dim;
  return st;
}
static void *rbf_gauss_alloc(size_t dim, size_t size) { return rbf_alloc_kind(GSL_SINTERP_RBF_GAUSSIAN, dim, size); }
static void *rbf_tps_alloc(size_t dim, size_t size) { return rbf_alloc_kind(GSL_SINTERP_RBF_TPS, dim, size); }
static void *rbf_wendland_alloc(size_t dim, size_t size) { return rbf_alloc_kind(GSL_SINTERP_RBF_WENDLAND, dim, size); }
static void *rbf_tps_affine_alloc(size_t dim, size_t size)
{
  rbf_state *st = (rbf_state *)rbf_alloc_kind(GSL_SINTERP_RBF_TPS, dim, size);
  if (st) st->affine = 1;
  return st;
}
static void *krige_alloc(size_t dim, size_t size)
{
  rbf_state *st = (rbf_state *)rbf_alloc_kind(GSL_SINTERP_RBF_GAUSSIAN, dim, size);      /* Gaussian covariance */
  if (st) st->krige = 1;
  return st;
}

static void rbf_release_devices(rbf_state *st)
{
  if (st->ss.grp) {
    for (int r = 0; r < st->ss.n; r++) {
      gsl_sinterp_hip_free(gsl_sinterp_hip_group_ctx(st->ss.grp, r), st->m_model[r]);
      st->m_model[r] = NULL;
    }
    shard_set_release(&st->ss);
  } else if (st->ctx) {
    chunk_set_release(&st->cs, st->ctx);
    gsl_sinterp_hip_free(st->ctx, st->d_x);       /* one buffer: d_w points into it */
    gsl_sinterp_hip_ctx_destroy(st->ctx);
  }
  st->ctx = NULL; st->d_x = st->d_w = NULL;
}

static void rbf_free(void *vstate)
{
  rbf_state *st = (rbf_state *)vstate;
  if (!st) return;
  rbf_release_devices(st);
  free(st);
}

static int rbf_prepare_devices(gsl_sinterp *interp, rbf_state *st);

static int rbf_init(gsl_sinterp *interp, const gsl_matrix *x, const gsl_vector *f)
{
  rbf_state *st = (rbf_state *)interp->state;
  const size_t n = st->n, dim = st->dim;
  const int nd = interp->n_devices > 1 ? interp->n_devices : 1;
  int s = rbf_prepare_devices(interp, st);
  if (s) return s;
  gsl_sinterp_hip_ctx *c = st->ctx;
  /* default shape: Gaussian eps = 2 N^(1/d) (the C-configurations); Wendland: support radius of eight mean
     spacings of a unit box, eps = N^(1/d) / 8 */
  st->model_id = next_model_id();                       /* the buffers are about to change */
  st->eps = interp->shape > 0 ? interp->shape
            : (st->kind == GSL_SINTERP_RBF_WENDLAND ? 0.125 : 2.0) * pow((double)n, 1.0 / (double)dim);

  double *h_x = (double *)malloc(n * dim * sizeof(double));
  double *h_f = (double *)malloc(n * sizeof(double));
  if (!h_x || !h_f) { free(h_x); free(h_f); GSL_ERROR("gsl_sinterp_init: out of memory", GSL_ENOMEM); }
  for (size_t i = 0; i < n; i++) {
    for (size_t cdim = 0; cdim < dim; cdim++) h_x[i * dim + cdim] = x->data[i * x->tda + cdim];
    h_f[i] = gsl_vector_get(f, i);
  }
  const size_t model_bytes = n * (dim + 1) * sizeof(double);
  double *d_phi = NULL;
  int route = 0;
  /* affine thin-plate spline: room for the (n + d + 1) augmented matrix of the pivoted-LU route (lda even) */
  const size_t lda = st->affine ? ((n + dim + 2) & ~(size_t)1) : n, phi_rows = st->affine ? n + dim + 1 : n;
  s = gsl_sinterp_hip_malloc(c, (void **)&d_phi, phi_rows * lda * sizeof(double));
  if (!s) s = gsl_sinterp_hip_h2d(c, st->d_x, h_x, n * dim * sizeof(double));
  if (!s) s = gsl_sinterp_hip_h2d(c, st->d_w, h_f, n * sizeof(double));
  /* fill + dense solve on the device: Cholesky (Gaussian), shifted-SPD Cholesky with a
     Woodbury correction or pivoted LU (thin-plate spline) -- csrc/hip/solve.hip */
  double rcond = GSL_NAN;
  if (!s && st->krige)
    s = gsl_sinterp_hip_krige_solve(c, st->kind, st->eps, interp->nugget, st->d_x, n, (int)dim, dim, d_phi, lda, st->d_w, &st->mean, &route);
  else if (!s && st->affine)
    s = gsl_sinterp_hip_rbf_solve_affine(c, st->kind, st->eps, st->d_x, n, (int)dim, dim, d_phi, lda, st->d_w, st->poly, &route);
  else if (!s)
    s = gsl_sinterp_hip_rbf_solve_ex(c, st->kind, st->eps, st->d_x, n, (int)dim, dim, d_phi, lda, st->d_w, interp->solver,
                                     interp->want_rcond ? &rcond : NULL, &route);
  interp->rcond = rcond; interp->route = route;
  /* replicate the solved model: ONE broadcast of the weight vector (+ centres) */
  if (!s && st->ss.grp) s = gsl_sinterp_hip_group_broadcast(st->ss.grp, (void *const *)st->m_model, model_bytes);
  if (!s) s = gsl_sinterp_hip_sync(c);
  if (st->ss.grp)
    for (int r = 1; r < nd; r++) { int s2 = gsl_sinterp_hip_sync(gsl_sinterp_hip_group_ctx(st->ss.grp, r)); if (!s) s = s2; }
  gsl_sinterp_hip_free(c, d_phi);
  free(h_x); free(h_f);
  if (s == GSL_EDOM) GSL_ERROR("gsl_sinterp_init: kernel matrix is not positive definite", GSL_EDOM);
  HIP_TRY(s, c);
  return GSL_SUCCESS;
}

/* contexts (one, or a device group) + the model buffer [centres | weights] of every member */
static int rbf_prepare_devices(gsl_sinterp *interp, rbf_state *st)
{
  const size_t n = st->n, dim = st->dim;
  const int nd = interp->n_devices > 1 ? interp->n_devices : 1;
  /* (re)build the device side when the requested device set changed */
  int same = st->ctx != NULL && ((nd == 1 && !st->ss.grp && gsl_sinterp_hip_ctx_device(st->ctx) == interp->device) ||
                                 (st->ss.grp && st->ss.n == nd));
  if (same && st->ss.grp)
    for (int r = 0; r < nd; r++) same = same && gsl_sinterp_hip_group_device(st->ss.grp, r) == interp->devices[r];
  if (!same) {
    rbf_release_devices(st);
    if (nd > 1) {
      if (gsl_sinterp_hip_group_create(&st->ss.grp, interp->devices, nd) != GSL_SUCCESS)
        GSL_ERROR("gsl_sinterp_init: cannot create the device group (GPU path has no CPU fallback)", GSL_EFAILED);
      st->ss.n = nd;
      st->ctx = gsl_sinterp_hip_group_ctx(st->ss.grp, 0);
    } else if (gsl_sinterp_hip_ctx_create(&st->ctx, interp->device, NULL) != GSL_SUCCESS) {
      st->ctx = NULL;
      GSL_ERROR("gsl_sinterp_init: no usable HIP device (GPU path has no CPU fallback)", GSL_EFAILED);
    }
  }
  /* the model = [centres | weights], one buffer per member: N (d+1) 8 bytes, the broadcast payload */
  const size_t model_bytes = n * (dim + 1) * sizeof(double);
  int s = GSL_SUCCESS;
  if (st->ss.grp) {
    for (int r = 0; r < nd && !s; r++)
      if (!st->m_model[r]) s = gsl_sinterp_hip_malloc(gsl_sinterp_hip_group_ctx(st->ss.grp, r), (void **)&st->m_model[r], model_bytes);
    st->d_x = st->m_model[0];
  } else if (!st->d_x) {
    s = gsl_sinterp_hip_malloc(st->ctx, (void **)&st->d_x, model_bytes);
  }
  st->d_w = st->d_x ? st->d_x + n * dim : NULL;
  if (s) { gsl_error(gsl_sinterp_hip_last_error(st->ctx), __FILE__, __LINE__, s); return s; }
  return GSL_SUCCESS;
}

/* the sweep of an RBF-type model on context c (plain, + kriging mean, + affine tail) */
static int rbf_sweep(const rbf_state *st, gsl_sinterp_hip_ctx *c, const double *d_x, const double *d_w, const double *d_y, size_t m,
                     size_t ytda, double *d_s)
{
  if (st->krige)
    return gsl_sinterp_hip_krige_eval(c, st->kind, st->eps, st->mean, d_x, st->n, (int)st->dim, st->dim, d_w, d_y, m, ytda, d_s, st->model_id);
  if (st->affine)
    return gsl_sinterp_hip_rbf_eval_affine(c, st->kind, st->eps, st->poly, d_x, st->n, (int)st->dim, st->dim, d_w, d_y, m, ytda, d_s,
                                           st->model_id);
  return gsl_sinterp_hip_rbf_eval_model(c, st->kind, st->eps, d_x, st->n, (int)st->dim, st->dim, d_w, d_y, m, ytda, d_s, st->model_id);
}

static int rbf_eval_resident(const gsl_sinterp *interp, const double *d_y, size_t m, size_t ytda,
                             double *d_s, int *d_leaf)
{
  (void)d_leaf;
  const rbf_state *st = (const rbf_state *)interp->state;
  if (!st->d_w) GSL_ERROR("gsl_sinterp_eval: interpolant not initialised", GSL_EINVAL);
  /* resident buffers live on ONE device: member 0 evaluates them (shard resident targets yourself with
     gsl_sinterp_hip_shard_bounds + one interpolant per device, as bench.py does per process) */
  HIP_TRY(rbf_sweep(st, st->ctx, st->d_x, st->d_w, d_y, m, ytda, d_s), st->ctx);
  return GSL_SUCCESS;
}

static int rbf_shard_eval(void *state, int member, const double *d_y, size_t m, double *d_s, int *d_leaf)
{
  (void)d_leaf;
  rbf_state *st = (rbf_state *)state;
  const double *model = st->m_model[member];
  return rbf_sweep(st, gsl_sinterp_hip_group_ctx(st->ss.grp, member), model, model + st->n * st->dim, d_y, m, st->dim, d_s);
}

static int rbf_chunk_eval(void *state, const double *d_y, size_t m, double *d_s, int *d_leaf)
{
  (void)d_leaf;
  const rbf_state *st = (const rbf_state *)state;
  return rbf_sweep(st, st->ctx, st->d_x, st->d_w, d_y, m, st->dim, d_s);
}

static int rbf_eval_many(const gsl_sinterp *interp, const gsl_matrix *y, gsl_vector *sv, int *leaf)
{
  const rbf_state *st = (const rbf_state *)interp->state;
  if (!st->d_w) GSL_ERROR("gsl_sinterp_eval_many: interpolant not initialised", GSL_EINVAL);
  const size_t m = y->size1, dim = st->dim;
  if (m == 0) return GSL_SUCCESS;
  if (st->ss.grp) {
    rbf_state *mst = (rbf_state *)interp->state;         /* staging buffers are grow-only caches inside the state */
    int sst = shard_eval_many(&mst->ss, dim, y, sv, NULL, &rbf_shard_eval, mst, 0, NULL);
    if (sst != GSL_SUCCESS) GSL_ERROR("gsl_sinterp_eval_many: sharded evaluation failed", sst);
    if (leaf) for (size_t k = 0; k < m; k++) leaf[k] = -1;
    return GSL_SUCCESS;
  }
  {
    rbf_state *mst = (rbf_state *)interp->state;           /* staging buffers are grow-only caches inside the state */
    int s = chunk_eval_many(&mst->cs, st->ctx, dim, y, sv, NULL, &rbf_chunk_eval, mst, 0, NULL);
    if (s != GSL_SUCCESS) GSL_ERROR("gsl_sinterp_eval_many: evaluation failed", s);
    if (leaf) for (size_t k = 0; k < m; k++) leaf[k] = -1;
  }
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
  gsl_vector *f; /* private copy of the response (checkpoints) */
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
  gsl_vector_free(st->f);
  free(st);
}

/* mirror st->tree (built over st->x) on the interpolant's device(s) and bind st->f */
static int simplex_mirror(gsl_sinterp *interp, simplex_state *st)
{
  simplex_tree_device_free(st->dev);
  st->dev = interp->n_devices > 1 ? simplex_tree_device_alloc_multi(st->tree, st->x, interp->devices, interp->n_devices)
                                  : simplex_tree_device_alloc(st->tree, st->x, interp->device);
  if (!st->dev) return GSL_EFAILED;
  return simplex_tree_device_set_response(st->dev, st->f);
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
  gsl_vector_free(st->f);
  st->f = gsl_vector_alloc(st->n);
  if (!st->f) return GSL_ENOMEM;
  for (size_t i = 0; i < st->n; i++) gsl_vector_set(st->f, i, gsl_vector_get(f, i));
  return simplex_mirror(interp, st);
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
/* imported triangulation type (README:28-31: QHull / CGAL meshes)           */
/* ======================================================================== */
typedef struct {
  size_t n;
  int *tri, *nbr;            /* the caller's triangulation (gsl_sinterp_set_triangulation), copied */
  size_t n_tri;
  simplex_mesh *mesh;
  simplex_mesh_device *dev;
  gsl_matrix *x;             /* private copies (checkpoints) */
  gsl_vector *f;
} mesh_state;

static void *mesh_type_alloc(size_t dim, size_t size)
{
  if (dim != 2) return NULL;
  mesh_state *st = (mesh_state *)calloc(1, sizeof *st);
  if (st) st->n = size;
  return st;
}

static void mesh_type_free(void *vstate)
{
  mesh_state *st = (mesh_state *)vstate;
  if (!st) return;
  simplex_mesh_device_free(st->dev);
  simplex_mesh_free(st->mesh);
  gsl_matrix_free(st->x);
  gsl_vector_free(st->f);
  free(st->tri); free(st->nbr);
  free(st);
}

static int mesh_type_mirror(gsl_sinterp *interp, mesh_state *st)
{
  simplex_mesh_device_free(st->dev);
  st->dev = interp->n_devices > 1 ? simplex_mesh_device_alloc_multi(st->mesh, interp->devices, interp->n_devices)
                                  : simplex_mesh_device_alloc(st->mesh, interp->device);
  if (!st->dev) return GSL_EFAILED;
  return simplex_mesh_device_set_response(st->dev, st->f);
}

static int mesh_type_init(gsl_sinterp *interp, const gsl_matrix *x, const gsl_vector *f)
{
  mesh_state *st = (mesh_state *)interp->state;
  if (!st->tri) GSL_ERROR("gsl_sinterp_init: no triangulation set (gsl_sinterp_set_triangulation)", GSL_EINVAL);
  simplex_mesh_device_free(st->dev); st->dev = NULL;
  simplex_mesh_free(st->mesh); st->mesh = NULL;
  gsl_matrix_free(st->x); gsl_vector_free(st->f);
  st->x = gsl_matrix_alloc(st->n, 2);
  st->f = gsl_vector_alloc(st->n);
  if (!st->x || !st->f) return GSL_ENOMEM;
  for (size_t i = 0; i < st->n; i++) {
    gsl_matrix_set(st->x, i, 0, x->data[i * x->tda]);
    gsl_matrix_set(st->x, i, 1, x->data[i * x->tda + 1]);
    gsl_vector_set(st->f, i, gsl_vector_get(f, i));
  }
  st->mesh = simplex_mesh_import(st->x, st->tri, st->nbr, st->n_tri);
  if (!st->mesh) return GSL_EINVAL;
  return mesh_type_mirror(interp, st);
}

static int mesh_type_eval_many(const gsl_sinterp *interp, const gsl_matrix *y, gsl_vector *s, int *leaf)
{
  const mesh_state *st = (const mesh_state *)interp->state;
  if (!st->dev) GSL_ERROR("gsl_sinterp_eval_many: interpolant not initialised", GSL_EINVAL);
  return simplex_mesh_device_eval_many(st->dev, y, s, leaf);
}

static int mesh_type_eval_resident(const gsl_sinterp *interp, const double *d_y, size_t m, size_t ytda, double *d_s, int *d_leaf)
{
  const mesh_state *st = (const mesh_state *)interp->state;
  if (!st->dev) GSL_ERROR("gsl_sinterp_eval_resident: interpolant not initialised", GSL_EINVAL);
  return simplex_mesh_device_eval_resident(st->dev, d_y, m, ytda, d_s, d_leaf);
}

/* ======================================================================== */
/* type table + generic entry points                                         */
/* ======================================================================== */
static const gsl_sinterp_type gauss_type = {"rbf-gaussian", 1, &rbf_gauss_alloc, &rbf_init, &rbf_eval_many, &rbf_eval_resident, &rbf_free};
static const gsl_sinterp_type tps_type = {"rbf-thin-plate-spline", 1, &rbf_tps_alloc, &rbf_init, &rbf_eval_many, &rbf_eval_resident, &rbf_free};
static const gsl_sinterp_type simplex_type = {"linear-simplex", 3, &simplex_alloc, &simplex_init, &simplex_eval_many, &simplex_eval_resident, &simplex_free};
static const gsl_sinterp_type wendland_type = {"rbf-wendland-c2", 1, &rbf_wendland_alloc, &rbf_init, &rbf_eval_many, &rbf_eval_resident, &rbf_free};
static const gsl_sinterp_type mesh_type = {"linear-imported-triangulation", 3, &mesh_type_alloc, &mesh_type_init, &mesh_type_eval_many, &mesh_type_eval_resident, &mesh_type_free};
const gsl_sinterp_type *gsl_sinterp_linear_mesh = &mesh_type;
static const gsl_sinterp_type tps_affine_type = {"rbf-thin-plate-spline-affine", 3, &rbf_tps_affine_alloc, &rbf_init, &rbf_eval_many, &rbf_eval_resident, &rbf_free};
const gsl_sinterp_type *gsl_sinterp_rbf_tps_affine = &tps_affine_type;
static const gsl_sinterp_type krige_type = {"ordinary-kriging-gaussian", 1, &krige_alloc, &rbf_init, &rbf_eval_many, &rbf_eval_resident, &rbf_free};
const gsl_sinterp_type *gsl_sinterp_kriging = &krige_type;
const gsl_sinterp_type *gsl_sinterp_rbf_wendland = &wendland_type;
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
  interp->n_devices = env_device_list(interp->devices);   /* GSL_SINTERP_DEVICES: count or list; 0 = single device */
  /* GSL_SINTERP_DEVICES = "1" (a COUNT of one) selects no ordinal: GSL_SINTERP_DEVICE keeps naming the device */
  if (interp->n_devices == 1 && getenv("GSL_SINTERP_DEVICES") && !strchr(getenv("GSL_SINTERP_DEVICES"), ',')) interp->n_devices = 0;
  if (interp->n_devices >= 1) interp->device = interp->devices[0];
  else { interp->n_devices = 1; interp->devices[0] = interp->device; }
  interp->shape = 0.0; interp->init_flags = SIMPLEX_TREE_DEFAULT; interp->rng = NULL;
  interp->solver = GSL_SINTERP_SOLVER_DEFAULT; interp->want_rcond = 0; interp->rcond = GSL_NAN; interp->route = 0;
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
  interp->n_devices = 1; interp->devices[0] = device;
  return GSL_SUCCESS;
}

/* Evaluate on `n_devices` GPUs (ordinals 0 .. n_devices-1): the next gsl_sinterp_init solves on the first,
   replicates the model with one broadcast and gsl_sinterp_eval_many shards its targets. */
int gsl_sinterp_set_devices(gsl_sinterp *interp, int n_devices)
{
  if (!interp) GSL_ERROR("gsl_sinterp_set_devices: null interpolant", GSL_EFAULT);
  if (n_devices < 1 || n_devices > GSL_SINTERP_MAX_DEVICES) GSL_ERROR("gsl_sinterp_set_devices: bad device count", GSL_EINVAL);
  for (int i = 0; i < n_devices; i++) interp->devices[i] = i;
  interp->n_devices = n_devices; interp->device = 0;
  return GSL_SUCCESS;
}

int gsl_sinterp_set_device_list(gsl_sinterp *interp, const int *devices, int n_devices)
{
  if (!interp || !devices) GSL_ERROR("gsl_sinterp_set_device_list: null argument", GSL_EFAULT);
  if (n_devices < 1 || n_devices > GSL_SINTERP_MAX_DEVICES) GSL_ERROR("gsl_sinterp_set_device_list: bad device count", GSL_EINVAL);
  for (int i = 0; i < n_devices; i++) {
    if (devices[i] < 0) GSL_ERROR("gsl_sinterp_set_device_list: negative ordinal", GSL_EINVAL);
    interp->devices[i] = devices[i];
  }
  interp->n_devices = n_devices; interp->device = devices[0];
  return GSL_SUCCESS;
}

int gsl_sinterp_n_devices(const gsl_sinterp *interp) { return interp ? interp->n_devices : 0; }

int gsl_sinterp_set_shape(gsl_sinterp *interp, double eps)
{
  if (!interp) GSL_ERROR("gsl_sinterp_set_shape: null interpolant", GSL_EFAULT);
  interp->shape = eps;
  return GSL_SUCCESS;
}

int gsl_sinterp_set_nugget(gsl_sinterp *interp, double nugget)
{
  if (!interp) GSL_ERROR("gsl_sinterp_set_nugget: null interpolant", GSL_EFAULT);
  if (interp->type != &krige_type) GSL_ERROR("gsl_sinterp_set_nugget: kriging interpolants only", GSL_EINVAL);
  if (!(nugget >= 0.0)) GSL_ERROR("gsl_sinterp_set_nugget: the nugget must be >= 0", GSL_EDOM);
  interp->nugget = nugget;
  return GSL_SUCCESS;
}

int gsl_sinterp_poly(const gsl_sinterp *interp, gsl_vector *c)
{
  if (!interp || !c) GSL_ERROR("gsl_sinterp_poly: null argument", GSL_EFAULT);
  if (interp->type != &tps_affine_type) GSL_ERROR("gsl_sinterp_poly: affine thin-plate-spline interpolants only", GSL_EINVAL);
  const rbf_state *st = (const rbf_state *)interp->state;
  if (!st->d_w) GSL_ERROR("gsl_sinterp_poly: interpolant not initialised", GSL_EINVAL);
  if (c->size != st->dim + 1) GSL_ERROR("gsl_sinterp_poly: vector length must be dim + 1", GSL_EBADLEN);
  for (size_t a = 0; a <= st->dim; a++) gsl_vector_set(c, a, st->poly[a]);
  return GSL_SUCCESS;
}

int gsl_sinterp_mean(const gsl_sinterp *interp, double *mean)
{
  if (!interp || !mean) GSL_ERROR("gsl_sinterp_mean: null argument", GSL_EFAULT);
  if (interp->type != &krige_type) GSL_ERROR("gsl_sinterp_mean: kriging interpolants only", GSL_EINVAL);
  const rbf_state *st = (const rbf_state *)interp->state;
  if (!st->d_w) GSL_ERROR("gsl_sinterp_mean: interpolant not initialised", GSL_EINVAL);
  *mean = st->mean;
  return GSL_SUCCESS;
}

int gsl_sinterp_set_solver(gsl_sinterp *interp, int solver)
{
  if (!interp) GSL_ERROR("gsl_sinterp_set_solver: null interpolant", GSL_EFAULT);
  if (solver < GSL_SINTERP_SOLVER_DEFAULT || solver > GSL_SINTERP_SOLVER_LU_REFINE)
    GSL_ERROR("gsl_sinterp_set_solver: unknown solver", GSL_EINVAL);
  if (interp->type == &simplex_type) GSL_ERROR("gsl_sinterp_set_solver: not an RBF interpolant", GSL_EINVAL);
  /* kriging and the affine thin-plate spline solve saddle systems on routes of their own (7 / 8, 9 / 10): accepting a
     solver here and ignoring it at init would be a silent no-op */
  if ((interp->type == &krige_type || interp->type == &tps_affine_type) && solver != GSL_SINTERP_SOLVER_DEFAULT)
    GSL_ERROR("gsl_sinterp_set_solver: kriging / affine thin-plate-spline interpolants choose their own route", GSL_EINVAL);
  if (interp->type == &tps_type && (solver == GSL_SINTERP_SOLVER_CHOLESKY2 || solver == GSL_SINTERP_SOLVER_PCHOLESKY))
    GSL_ERROR("gsl_sinterp_set_solver: the thin-plate-spline matrix is indefinite (zero diagonal): no Cholesky-type solver", GSL_EINVAL);
  interp->solver = solver;
  return GSL_SUCCESS;
}

int gsl_sinterp_set_rcond(gsl_sinterp *interp, int want)
{
  if (!interp) GSL_ERROR("gsl_sinterp_set_rcond: null interpolant", GSL_EFAULT);
  if (want && (interp->type == &krige_type || interp->type == &tps_affine_type))
    GSL_ERROR("gsl_sinterp_set_rcond: no condition estimate on the kriging / affine thin-plate-spline routes", GSL_EINVAL);
  interp->want_rcond = want != 0;
  return GSL_SUCCESS;
}

int gsl_sinterp_rcond(const gsl_sinterp *interp, double *rcond)
{
  if (!interp || !rcond) GSL_ERROR("gsl_sinterp_rcond: null argument", GSL_EFAULT);
  *rcond = interp->rcond;
  if (interp->rcond != interp->rcond) GSL_ERROR("gsl_sinterp_rcond: no estimate available (set_rcond + a Cholesky solver + init)", GSL_EINVAL);
  return GSL_SUCCESS;
}

int gsl_sinterp_route(const gsl_sinterp *interp) { return interp ? interp->route : 0; }

int gsl_sinterp_set_triangulation(gsl_sinterp *interp, const int *triangles, const int *neighbours, size_t n_triangles)
{
  if (!interp || !triangles) GSL_ERROR("gsl_sinterp_set_triangulation: null argument", GSL_EFAULT);
  if (interp->type != &mesh_type) GSL_ERROR("gsl_sinterp_set_triangulation: imported-triangulation interpolants only", GSL_EINVAL);
  if (n_triangles < 1 || n_triangles > (size_t)INT_MAX / 3) GSL_ERROR("gsl_sinterp_set_triangulation: bad triangle count", GSL_EINVAL);
  mesh_state *st = (mesh_state *)interp->state;
  int *t = (int *)malloc(3 * n_triangles * sizeof(int)), *nb = neighbours ? (int *)malloc(3 * n_triangles * sizeof(int)) : NULL;
  if (!t || (neighbours && !nb)) { free(t); free(nb); GSL_ERROR("gsl_sinterp_set_triangulation: out of memory", GSL_ENOMEM); }
  memcpy(t, triangles, 3 * n_triangles * sizeof(int));
  if (nb) memcpy(nb, neighbours, 3 * n_triangles * sizeof(int));
  free(st->tri); free(st->nbr);
  st->tri = t; st->nbr = nb; st->n_tri = n_triangles;
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

/* ======================================================================== */
/* gridded front-end (interpolation/scattered_interp_example.c:175-217)      */
/* ======================================================================== */
static gsl_sinterp_hip_ctx *interp_ctx0(const gsl_sinterp *interp)
{
  if (interp->type == &simplex_type) {
    const simplex_state *st = (const simplex_state *)interp->state;
    return st->dev ? st->dev->ctx : NULL;
  }
  return ((const rbf_state *)interp->state)->ctx;
}

int gsl_sinterp_eval_grid(const gsl_sinterp *interp, const gsl_vector *min, const gsl_vector *max, gsl_matrix *grid)
{
  if (!interp || !min || !max || !grid) GSL_ERROR("gsl_sinterp_eval_grid: null argument", GSL_EFAULT);
  if (interp->dim != 2) GSL_ERROR("gsl_sinterp_eval_grid: 2-D interpolants only", GSL_EINVAL);
  if (min->size != 2 || max->size != 2) GSL_ERROR("gsl_sinterp_eval_grid: min / max must have 2 components", GSL_EBADLEN);
  gsl_sinterp_hip_ctx *c = interp_ctx0(interp);
  if (!c) GSL_ERROR("gsl_sinterp_eval_grid: interpolant not initialised", GSL_EINVAL);
  const size_t n0 = grid->size1, n1 = grid->size2, m = n0 * n1;
  if (m == 0) return GSL_SUCCESS;
  /* scattered_interp_example.c:179-183: step = range / n_grid */
  const double min0 = gsl_vector_get(min, 0), min1 = gsl_vector_get(min, 1);
  const double xrange = (gsl_vector_get(max, 0) - min0), xstep = xrange / (double)n0;
  const double yrange = (gsl_vector_get(max, 1) - min1), ystep = yrange / (double)n1;
  double *d_y = NULL, *d_s = NULL, *h_s = (double *)malloc(m * sizeof(double));
  if (!h_s) GSL_ERROR("gsl_sinterp_eval_grid: out of memory", GSL_ENOMEM);
  int st = gsl_sinterp_hip_malloc(c, (void **)&d_y, m * 2 * sizeof(double));
  if (!st) st = gsl_sinterp_hip_malloc(c, (void **)&d_s, m * sizeof(double));
  if (!st) st = gsl_sinterp_hip_grid_targets(c, min0, xstep, n0, min1, ystep, n1, d_y);
  int est = GSL_SUCCESS;
  if (!st) {
    gsl_error_handler_t *saved = gsl_set_error_handler_off();       /* EDOM (node outside the cage) is reported below */
    est = interp->type->eval_resident(interp, d_y, m, 2, d_s, NULL);
    gsl_set_error_handler(saved);
    if (est != GSL_SUCCESS && est != GSL_EDOM) st = est;
  }
  if (!st) st = gsl_sinterp_hip_d2h(c, h_s, d_s, m * sizeof(double));
  size_t n_nan = 0;
  if (!st)
    for (size_t i = 0; i < n0; i++)
      for (size_t j = 0; j < n1; j++) { const double v = h_s[i * n1 + j]; n_nan += v != v; gsl_matrix_set(grid, i, j, v); }
  gsl_sinterp_hip_free(c, d_y); gsl_sinterp_hip_free(c, d_s);
  free(h_s);
  HIP_TRY(st, c);
  if (interp->type == &simplex_type && n_nan) GSL_ERROR("gsl_sinterp_eval_grid: grid node(s) outside the caging simplex", GSL_EDOM);
  return GSL_SUCCESS;
}

int gsl_sinterp_fprintf_grid(FILE *stream, const gsl_vector *min, const gsl_vector *max, const gsl_matrix *grid)
{
  if (!stream || !min || !max || !grid) GSL_ERROR("gsl_sinterp_fprintf_grid: null argument", GSL_EFAULT);
  if (min->size != 2 || max->size != 2) GSL_ERROR("gsl_sinterp_fprintf_grid: min / max must have 2 components", GSL_EBADLEN);
  const size_t n0 = grid->size1, n1 = grid->size2;
  const double min0 = gsl_vector_get(min, 0), min1 = gsl_vector_get(min, 1);
  const double xstep = (gsl_vector_get(max, 0) - min0) / (double)n0, ystep = (gsl_vector_get(max, 1) - min1) / (double)n1;
  for (size_t i = 0; i < n0; i++) {                                  /* scattered_interp_example.c:203-215 */
    for (size_t j = 0; j < n1; j++)
      if (fprintf(stream, "%g %g %g\n", min0 + xstep * (double)i, min1 + ystep * (double)j, gsl_matrix_get(grid, i, j)) < 0)
        GSL_ERROR("fprintf failed", GSL_EFAILED);
    if (fprintf(stream, "\n") < 0) GSL_ERROR("fprintf failed", GSL_EFAILED);
  }
  return GSL_SUCCESS;
}

/* ======================================================================== */
/* binary checkpoint of an initialised interpolant                           */
/*   magic "GSLSINT1" | type id | dim | size | eps | init_flags | payload    */
/*   RBF payload: centres (size x dim) | weights (size)                      */
/*   linear simplex payload: tree (simplex_tree_fwrite) | centres | response */
/* ======================================================================== */
static const char INTERP_MAGIC[8] = {'G', 'S', 'L', 'S', 'I', 'N', 'T', '1'};

static int type_id(const gsl_sinterp_type *T)
{
  return T == &gauss_type ? 0 : (T == &tps_type ? 1 : (T == &wendland_type ? 3 : (T == &krige_type ? 4 : (T == &tps_affine_type ? 5 : (T == &mesh_type ? 6 : 2)))));
}

int gsl_sinterp_fwrite(FILE *stream, const gsl_sinterp *interp)
{
  if (!stream || !interp) GSL_ERROR("gsl_sinterp_fwrite: null argument", GSL_EFAULT);
  const uint64_t head[3] = {(uint64_t)type_id(interp->type), (uint64_t)interp->dim, (uint64_t)interp->size};
  if (interp->type == &simplex_type) {
    const simplex_state *st = (const simplex_state *)interp->state;
    if (!st->dev || !st->tree || !st->x || !st->f) GSL_ERROR("gsl_sinterp_fwrite: interpolant not initialised", GSL_EINVAL);
    const double eps = 0.0;
    const int64_t flags = interp->init_flags;
    if (fwrite(INTERP_MAGIC, 1, 8, stream) != 8 || fwrite(head, sizeof head[0], 3, stream) != 3 ||
        fwrite(&eps, sizeof eps, 1, stream) != 1 || fwrite(&flags, sizeof flags, 1, stream) != 1)
      GSL_ERROR("fwrite failed", GSL_EFAILED);
    int s = simplex_tree_fwrite(stream, st->tree);
    if (s) return s;
    for (size_t i = 0; i < st->n; i++) {
      const double row[3] = {gsl_matrix_get(st->x, i, 0), gsl_matrix_get(st->x, i, 1), gsl_vector_get(st->f, i)};
      if (fwrite(row, sizeof(double), 3, stream) != 3) GSL_ERROR("fwrite failed", GSL_EFAILED);
    }
    return GSL_SUCCESS;
  }
  if (interp->type == &mesh_type) {
    const mesh_state *st = (const mesh_state *)interp->state;
    if (!st->dev || !st->mesh || !st->x || !st->f) GSL_ERROR("gsl_sinterp_fwrite: interpolant not initialised", GSL_EINVAL);
    const double eps = 0.0;
    const int64_t flags = 0;
    if (fwrite(INTERP_MAGIC, 1, 8, stream) != 8 || fwrite(head, sizeof head[0], 3, stream) != 3 ||
        fwrite(&eps, sizeof eps, 1, stream) != 1 || fwrite(&flags, sizeof flags, 1, stream) != 1)
      GSL_ERROR("fwrite failed", GSL_EFAILED);
    int s = simplex_mesh_fwrite(stream, st->mesh);
    if (s) return s;
    for (size_t i = 0; i < st->n; i++) {
      const double fi = gsl_vector_get(st->f, i);
      if (fwrite(&fi, sizeof fi, 1, stream) != 1) GSL_ERROR("fwrite failed", GSL_EFAILED);
    }
    return GSL_SUCCESS;
  }
  const rbf_state *st = (const rbf_state *)interp->state;
  if (!st->d_w) GSL_ERROR("gsl_sinterp_fwrite: interpolant not initialised", GSL_EINVAL);
  const size_t cnt = st->n * (st->dim + 1);
  double *h = (double *)malloc(cnt * sizeof(double));
  if (!h) GSL_ERROR("gsl_sinterp_fwrite: out of memory", GSL_ENOMEM);
  int s = gsl_sinterp_hip_d2h(st->ctx, h, st->d_x, cnt * sizeof(double));      /* the model buffer: [centres | weights] */
  int64_t flags = 0;
  if (st->krige) memcpy(&flags, &st->mean, sizeof flags);                       /* kriging: the word carries the mean's bits */
  if (!s && (fwrite(INTERP_MAGIC, 1, 8, stream) != 8 || fwrite(head, sizeof head[0], 3, stream) != 3 ||
             fwrite(&st->eps, sizeof st->eps, 1, stream) != 1 || fwrite(&flags, sizeof flags, 1, stream) != 1 ||
             fwrite(h, sizeof(double), cnt, stream) != cnt ||
             (st->affine && fwrite(st->poly, sizeof(double), 4, stream) != 4))) {      /* affine tail: c_0 .. c_3 behind the weights */
    free(h);
    GSL_ERROR("fwrite failed", GSL_EFAILED);
  }
  free(h);
  HIP_TRY(s, st->ctx);
  return GSL_SUCCESS;
}

int gsl_sinterp_fread(FILE *stream, gsl_sinterp *interp)
{
  if (!stream || !interp) GSL_ERROR("gsl_sinterp_fread: null argument", GSL_EFAULT);
  char magic[8];
  uint64_t head[3];
  double eps;
  int64_t flags;
  if (fread(magic, 1, 8, stream) != 8 || memcmp(magic, INTERP_MAGIC, 8) != 0)
    GSL_ERROR("gsl_sinterp_fread: not a gsl_sinterp checkpoint", GSL_EFAILED);
  if (fread(head, sizeof head[0], 3, stream) != 3 || fread(&eps, sizeof eps, 1, stream) != 1 ||
      fread(&flags, sizeof flags, 1, stream) != 1)
    GSL_ERROR("fread failed", GSL_EFAILED);
  if (head[0] != (uint64_t)type_id(interp->type) || head[1] != interp->dim || head[2] != interp->size)
    GSL_ERROR("gsl_sinterp_fread: checkpoint does not match the type / dim / size of the interpolant", GSL_EBADLEN);
  if (interp->type == &simplex_type) {
    simplex_state *st = (simplex_state *)interp->state;
    simplex_tree *tree = simplex_tree_fread(stream);
    if (!tree) return GSL_EFAILED;
    if ((size_t)tree->n_points != st->n) { simplex_tree_free(tree); GSL_ERROR("gsl_sinterp_fread: tree / size mismatch", GSL_EBADLEN); }
    gsl_matrix *x = gsl_matrix_alloc(st->n, 2);
    gsl_vector *f = gsl_vector_alloc(st->n);
    int ok = x && f;
    for (size_t i = 0; ok && i < st->n; i++) {
      double row[3];
      ok = fread(row, sizeof(double), 3, stream) == 3;
      if (ok) { gsl_matrix_set(x, i, 0, row[0]); gsl_matrix_set(x, i, 1, row[1]); gsl_vector_set(f, i, row[2]); }
    }
    if (!ok) { simplex_tree_free(tree); gsl_matrix_free(x); gsl_vector_free(f); GSL_ERROR("fread failed", GSL_EFAILED); }
    simplex_tree_device_free(st->dev); st->dev = NULL;
    simplex_tree_free(st->tree); gsl_matrix_free(st->x); gsl_vector_free(st->f);
    st->tree = tree; st->x = x; st->f = f;
    interp->init_flags = (int)flags;
    return simplex_mirror(interp, st);                  /* upload + pack + bind: no triangulation */
  }
  if (interp->type == &mesh_type) {
    mesh_state *st = (mesh_state *)interp->state;
    simplex_mesh *mesh = simplex_mesh_fread(stream);
    if (!mesh) return GSL_EFAILED;
    if (simplex_mesh_n_points(mesh) != st->n) { simplex_mesh_free(mesh); GSL_ERROR("gsl_sinterp_fread: mesh / size mismatch", GSL_EBADLEN); }
    gsl_matrix *x = gsl_matrix_alloc(st->n, 2);
    gsl_vector *f = gsl_vector_alloc(st->n);
    int ok = x && f;
    for (size_t i = 0; ok && i < st->n; i++) {
      double fi;
      ok = fread(&fi, sizeof fi, 1, stream) == 1;
      if (ok) {
        gsl_vector_set(f, i, fi);
        gsl_matrix_set(x, i, 0, simplex_mesh_points(mesh)[2 * i]); gsl_matrix_set(x, i, 1, simplex_mesh_points(mesh)[2 * i + 1]);
      }
    }
    if (!ok) { simplex_mesh_free(mesh); gsl_matrix_free(x); gsl_vector_free(f); GSL_ERROR("fread failed", GSL_EFAILED); }
    simplex_mesh_device_free(st->dev); st->dev = NULL;
    simplex_mesh_free(st->mesh); gsl_matrix_free(st->x); gsl_vector_free(st->f);
    st->mesh = mesh; st->x = x; st->f = f;
    return mesh_type_mirror(interp, st);                /* upload + pack + bind: nothing is re-imported */
  }
  rbf_state *st = (rbf_state *)interp->state;
  const size_t cnt = st->n * (st->dim + 1);
  double *h = (double *)malloc(cnt * sizeof(double));
  if (!h) GSL_ERROR("gsl_sinterp_fread: out of memory", GSL_ENOMEM);
  if (fread(h, sizeof(double), cnt, stream) != cnt) { free(h); GSL_ERROR("fread failed", GSL_EFAILED); }
  double poly[4] = {0.0, 0.0, 0.0, 0.0};
  if (st->affine && fread(poly, sizeof(double), 4, stream) != 4) { free(h); GSL_ERROR("fread failed", GSL_EFAILED); }
  int s = rbf_prepare_devices(interp, st);
  if (s) { free(h); return s; }
  st->eps = eps;
  if (st->krige) memcpy(&st->mean, &flags, sizeof st->mean);
  if (st->affine) memcpy(st->poly, poly, sizeof poly);
  st->model_id = next_model_id();
  s = gsl_sinterp_hip_h2d(st->ctx, st->d_x, h, cnt * sizeof(double));
  if (!s && st->ss.grp) s = gsl_sinterp_hip_group_broadcast(st->ss.grp, (void *const *)st->m_model, cnt * sizeof(double));
  if (!s) s = gsl_sinterp_hip_sync(st->ctx);
  if (st->ss.grp)
    for (int r = 1; r < st->ss.n; r++) { int s2 = gsl_sinterp_hip_sync(gsl_sinterp_hip_group_ctx(st->ss.grp, r)); if (!s) s = s2; }
  free(h);
  HIP_TRY(s, st->ctx);
  return GSL_SUCCESS;                                   /* nothing was filled, factorised or solved */
}
