/*
 * gsl_compat.c -- the handful of GSL container / error / rng functions that the
 * scattered-interpolation boundary needs, for images without libgsl.
 * Left out of the link under -DGSL_SINTERP_SYSTEM_GSL (see
 * include/gsl_sinterp_compat.h).  Behaviour mirrored (reference file:line):
 *   error handler        err/error.c:32-65 (default: print + abort; _off: no-op)
 *   vector/matrix views  vector/view_source.c, matrix/view_source.c, matrix/rowcol_source.c
 *   mt19937              rng/mt.c:79-152 ; uniform_int rng/gsl_rng.h:190-212
 *   shuffle              randist/shuffle.c:69-79
 */
#ifndef GSL_SINTERP_SYSTEM_GSL
#include "gsl_sinterp_compat.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------- error handling ---------------- */
static gsl_error_handler_t *g_handler = NULL;

static void silent_handler(const char *reason, const char *file, int line, int gsl_errno)
{
  (void)reason; (void)file; (void)line; (void)gsl_errno;
}

void gsl_error(const char *reason, const char *file, int line, int gsl_errno)
{
  if (g_handler) {
    (*g_handler)(reason, file, line, gsl_errno);
    return;
  }
  fprintf(stderr, "gsl: %s:%d: ERROR: %s\n", file, line, reason);
  fflush(stdout);
  fprintf(stderr, "Default GSL error handler invoked.\n");
  fflush(stderr);
  abort();
}

gsl_error_handler_t *gsl_set_error_handler(gsl_error_handler_t *new_handler)
{
  gsl_error_handler_t *prev = g_handler;
  g_handler = new_handler;
  return prev;
}

gsl_error_handler_t *gsl_set_error_handler_off(void)
{
  gsl_error_handler_t *prev = g_handler;
  g_handler = silent_handler;
  return prev;
}

const char *gsl_strerror(const int e)
{
  switch (e) {
    case GSL_SUCCESS: return "success";
    case GSL_FAILURE: return "failure";
    case GSL_EDOM: return "input domain error";
    case GSL_EINVAL: return "invalid argument supplied by user";
    case GSL_EFAILED: return "generic failure";
    case GSL_ENOMEM: return "malloc failed";
    case GSL_EBADLEN: return "matrix/vector sizes are not conformant";
    case GSL_ENOTSQR: return "matrix not square";
    case GSL_ESING: return "apparent singularity detected";
    case GSL_EUNSUP: return "requested feature is not supported by the hardware";
    case GSL_EUNIMPL: return "requested feature not (yet) implemented";
    default: return "unknown error code";
  }
}

/* ---------------- vectors ---------------- */
static gsl_block *block_alloc(size_t n, int zero)
{
  gsl_block *b = (gsl_block *)malloc(sizeof *b);
  if (!b) return NULL;
  b->size = n;
  b->data = (double *)(zero ? calloc(n ? n : 1, sizeof(double)) : malloc((n ? n : 1) * sizeof(double)));
  if (!b->data) { free(b); return NULL; }
  return b;
}

static gsl_vector *vector_new(size_t n, int zero)
{
  gsl_vector *v = (gsl_vector *)malloc(sizeof *v);
  if (!v) GSL_ERROR_NULL("failed to allocate space for vector struct", GSL_ENOMEM);
  gsl_block *b = block_alloc(n, zero);
  if (!b) { free(v); GSL_ERROR_NULL("failed to allocate space for block", GSL_ENOMEM); }
  v->size = n; v->stride = 1; v->data = b->data; v->block = b; v->owner = 1;
  return v;
}
gsl_vector *gsl_vector_alloc(const size_t n) { return vector_new(n, 0); }
gsl_vector *gsl_vector_calloc(const size_t n) { return vector_new(n, 1); }

void gsl_vector_free(gsl_vector *v)
{
  if (!v) return;
  if (v->owner && v->block) { free(v->block->data); free(v->block); }
  free(v);
}

gsl_vector_view gsl_vector_view_array_with_stride(double *base, size_t stride, size_t n)
{
  gsl_vector_view view = {{0, 0, 0, 0, 0}};
  view.vector.size = n; view.vector.stride = stride; view.vector.data = base;
  return view;
}
gsl_vector_view gsl_vector_view_array(double *base, size_t n)
{
  return gsl_vector_view_array_with_stride(base, 1, n);
}

/* ---------------- matrices ---------------- */
static gsl_matrix *matrix_new(size_t n1, size_t n2, int zero)
{
  gsl_matrix *m = (gsl_matrix *)malloc(sizeof *m);
  if (!m) GSL_ERROR_NULL("failed to allocate space for matrix struct", GSL_ENOMEM);
  gsl_block *b = block_alloc(n1 * n2, zero);
  if (!b) { free(m); GSL_ERROR_NULL("failed to allocate space for block", GSL_ENOMEM); }
  m->size1 = n1; m->size2 = n2; m->tda = n2; m->data = b->data; m->block = b; m->owner = 1;
  return m;
}
gsl_matrix *gsl_matrix_alloc(const size_t n1, const size_t n2) { return matrix_new(n1, n2, 0); }
gsl_matrix *gsl_matrix_calloc(const size_t n1, const size_t n2) { return matrix_new(n1, n2, 1); }

void gsl_matrix_free(gsl_matrix *m)
{
  if (!m) return;
  if (m->owner && m->block) { free(m->block->data); free(m->block); }
  free(m);
}

gsl_matrix_view gsl_matrix_view_array_with_tda(double *base, const size_t n1, const size_t n2, const size_t tda)
{
  gsl_matrix_view view = {{0, 0, 0, 0, 0, 0}};
  if (n2 > tda) GSL_ERROR_VAL("matrix dimension n2 must not exceed tda", GSL_EINVAL, view);
  view.matrix.size1 = n1; view.matrix.size2 = n2; view.matrix.tda = tda; view.matrix.data = base;
  return view;
}
gsl_matrix_view gsl_matrix_view_array(double *base, const size_t n1, const size_t n2)
{
  return gsl_matrix_view_array_with_tda(base, n1, n2, n2);
}

gsl_matrix_view gsl_matrix_submatrix(gsl_matrix *m, const size_t i, const size_t j, const size_t n1, const size_t n2)
{
  gsl_matrix_view view = {{0, 0, 0, 0, 0, 0}};
  if (i + n1 > m->size1 || j + n2 > m->size2)
    GSL_ERROR_VAL("submatrix overflows the matrix", GSL_EINVAL, view);
  view.matrix.size1 = n1; view.matrix.size2 = n2; view.matrix.tda = m->tda;
  view.matrix.data = m->data + (i * m->tda + j);
  view.matrix.block = m->block;
  return view;
}

gsl_vector_view gsl_matrix_row(gsl_matrix *m, const size_t i)
{
  gsl_vector_view view = {{0, 0, 0, 0, 0}};
  if (i >= m->size1) GSL_ERROR_VAL("row index is out of range", GSL_EINVAL, view);
  view.vector.size = m->size2; view.vector.stride = 1; view.vector.data = m->data + i * m->tda;
  view.vector.block = m->block;
  return view;
}

gsl_vector_view gsl_matrix_column(gsl_matrix *m, const size_t j)
{
  gsl_vector_view view = {{0, 0, 0, 0, 0}};
  if (j >= m->size2) GSL_ERROR_VAL("column index is out of range", GSL_EINVAL, view);
  view.vector.size = m->size1; view.vector.stride = m->tda; view.vector.data = m->data + j;
  view.vector.block = m->block;
  return view;
}

/* ---------------- permutations ---------------- */
gsl_permutation *gsl_permutation_alloc(const size_t n)
{
  gsl_permutation *p = (gsl_permutation *)malloc(sizeof *p);
  if (!p) GSL_ERROR_NULL("failed to allocate space for permutation struct", GSL_ENOMEM);
  p->data = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
  if (!p->data) { free(p); GSL_ERROR_NULL("failed to allocate space for permutation data", GSL_ENOMEM); }
  p->size = n;
  return p;
}
void gsl_permutation_init(gsl_permutation *p) { for (size_t i = 0; i < p->size; i++) p->data[i] = i; }
void gsl_permutation_free(gsl_permutation *p) { if (p) { free(p->data); free(p); } }

/* ---------------- mt19937 ---------------- */
#define MTN 624
#define MTM 397
typedef struct { unsigned long mt[MTN]; int mti; } mt_state;

static void mt_seed(void *vs, unsigned long s)
{
  mt_state *st = (mt_state *)vs;
  if (s == 0) s = 4357;
  st->mt[0] = s & 0xffffffffUL;
  int i;
  for (i = 1; i < MTN; i++) {
    unsigned long p = st->mt[i - 1];
    st->mt[i] = (1812433253UL * (p ^ (p >> 30)) + (unsigned long)i) & 0xffffffffUL;
  }
  st->mti = i;
}

static unsigned long mt_next(void *vs)
{
  mt_state *st = (mt_state *)vs;
  unsigned long *mt = st->mt;
  if (st->mti >= MTN) {
    for (int k = 0; k < MTN; k++) {
      unsigned long y = (mt[k] & 0x80000000UL) | (mt[(k + 1) % MTN] & 0x7fffffffUL);
      mt[k] = mt[(k + MTM) % MTN] ^ (y >> 1) ^ ((y & 1UL) ? 0x9908b0dfUL : 0UL);
    }
    st->mti = 0;
  }
  unsigned long k = mt[st->mti++];
  k ^= (k >> 11);
  k ^= (k << 7) & 0x9d2c5680UL;
  k ^= (k << 15) & 0xefc60000UL;
  k ^= (k >> 18);
  return k;
}

static double mt_next_double(void *vs) { return mt_next(vs) / 4294967296.0; }

static const gsl_rng_type mt_type = {"mt19937", 0xffffffffUL, 0, sizeof(mt_state), &mt_seed, &mt_next, &mt_next_double};
const gsl_rng_type *gsl_rng_mt19937 = &mt_type;
const gsl_rng_type *gsl_rng_default = &mt_type;
unsigned long int gsl_rng_default_seed = 0;

const gsl_rng_type *gsl_rng_env_setup(void)
{
  const char *s = getenv("GSL_RNG_SEED");            /* rng/default.c:82 */
  if (s) gsl_rng_default_seed = strtoul(s, 0, 0);
  gsl_rng_default = &mt_type;                        /* only generator provided */
  return gsl_rng_default;
}

gsl_rng *gsl_rng_alloc(const gsl_rng_type *T)
{
  gsl_rng *r = (gsl_rng *)malloc(sizeof *r);
  if (!r) GSL_ERROR_NULL("failed to allocate space for rng struct", GSL_ENOMEM);
  r->state = calloc(1, T->size);
  if (!r->state) { free(r); GSL_ERROR_NULL("failed to allocate space for rng state", GSL_ENOMEM); }
  r->type = T;
  gsl_rng_set(r, gsl_rng_default_seed);
  return r;
}
void gsl_rng_set(const gsl_rng *r, unsigned long int seed) { (r->type->set)(r->state, seed); }
void gsl_rng_free(gsl_rng *r) { if (r) { free(r->state); free(r); } }
unsigned long int gsl_rng_get(const gsl_rng *r) { return (r->type->get)(r->state); }
double gsl_rng_uniform(const gsl_rng *r) { return (r->type->get_double)(r->state); }

unsigned long int gsl_rng_uniform_int(const gsl_rng *r, unsigned long int n)
{
  unsigned long offset = r->type->min;
  unsigned long range = r->type->max - offset;
  if (n > range || n == 0)
    GSL_ERROR_VAL("invalid n, either 0 or exceeds maximum value of generator", GSL_EINVAL, 0);
  unsigned long scale = range / n, k;
  do { k = ((r->type->get)(r->state) - offset) / scale; } while (k >= n);
  return k;
}

void gsl_ran_shuffle(const gsl_rng *r, void *base, size_t n, size_t size)
{
  if (n < 2) return;
  char *b = (char *)base;
  for (size_t i = n - 1; i > 0; i--) {
    size_t j = gsl_rng_uniform_int(r, i + 1);
    if (i == j) continue;
    char *pa = b + size * i, *pb = b + size * j;
    for (size_t s = 0; s < size; s++) { char t = pa[s]; pa[s] = pb[s]; pb[s] = t; }
  }
}
#endif /* !GSL_SINTERP_SYSTEM_GSL */
