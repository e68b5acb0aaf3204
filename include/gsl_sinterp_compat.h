/*
 * gsl_sinterp_compat.h -- the slice of the GSL container / error / rng ABI that
 * crosses the scattered-interpolation boundary.
 *
 * When a real libgsl is installed, build with -DGSL_SINTERP_SYSTEM_GSL: the
 * GSL headers are used as they are and csrc/host/gsl_compat.c is left out of
 * the link.  This image has no libgsl, so the same LP64 layouts are declared
 * here (own code) and the dozen functions the path needs are provided by
 * csrc/host/gsl_compat.c.  Layouts replaced (reference file:line):
 *   gsl_block        block/gsl_block_double.h:38-42
 *   gsl_vector       vector/gsl_vector_double.h:42-50     (+ _view :52-57)
 *   gsl_matrix       matrix/gsl_matrix_double.h:42-50     (+ _view :52-57), element (i,j) = data[i*tda+j]
 *   gsl_permutation  permutation/gsl_permutation.h:41-45
 *   gsl_rng(_type)   rng/gsl_rng.h:41-59
 *   error codes      err/gsl_errno.h:40-74 ; handler err/error.c:32-65
 */
#ifndef GSL_SINTERP_COMPAT_H
#define GSL_SINTERP_COMPAT_H

#ifdef GSL_SINTERP_SYSTEM_GSL
#include <gsl/gsl_errno.h>
#include <gsl/gsl_math.h>
#include <gsl/gsl_matrix.h>
#include <gsl/gsl_permutation.h>
#include <gsl/gsl_rng.h>
#include <gsl/gsl_randist.h>
#include <gsl/gsl_vector.h>
#else

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---- */
enum {
  GSL_SUCCESS = 0, GSL_FAILURE = -1, GSL_CONTINUE = -2,
  GSL_EDOM = 1, GSL_ERANGE = 2, GSL_EFAULT = 3, GSL_EINVAL = 4, GSL_EFAILED = 5,
  GSL_EFACTOR = 6, GSL_ESANITY = 7, GSL_ENOMEM = 8, GSL_EBADFUNC = 9, GSL_ERUNAWAY = 10,
  GSL_EMAXITER = 11, GSL_EZERODIV = 12, GSL_EBADTOL = 13, GSL_ETOL = 14, GSL_EUNDRFLW = 15,
  GSL_EOVRFLW = 16, GSL_ELOSS = 17, GSL_EROUND = 18, GSL_EBADLEN = 19, GSL_ENOTSQR = 20,
  GSL_ESING = 21, GSL_EDIVERGE = 22, GSL_EUNSUP = 23, GSL_EUNIMPL = 24, GSL_ECACHE = 25,
  GSL_ETABLE = 26, GSL_ENOPROG = 27, GSL_ENOPROGJ = 28, GSL_ETOLF = 29, GSL_ETOLX = 30,
  GSL_ETOLG = 31, GSL_EOF = 32
};

typedef void gsl_error_handler_t(const char *reason, const char *file, int line, int gsl_errno);
void gsl_error(const char *reason, const char *file, int line, int gsl_errno);
gsl_error_handler_t *gsl_set_error_handler(gsl_error_handler_t *new_handler);
gsl_error_handler_t *gsl_set_error_handler_off(void);
const char *gsl_strerror(const int gsl_errno);

#define GSL_ERROR(reason, gsl_errno)                      \
  do {                                                    \
    gsl_error(reason, __FILE__, __LINE__, gsl_errno);     \
    return gsl_errno;                                     \
  } while (0)
#define GSL_ERROR_VAL(reason, gsl_errno, value)           \
  do {                                                    \
    gsl_error(reason, __FILE__, __LINE__, gsl_errno);     \
    return value;                                         \
  } while (0)
#define GSL_ERROR_NULL(reason, gsl_errno) GSL_ERROR_VAL(reason, gsl_errno, 0)

/* ---- machine constants (gsl_machine.h:17-21) ---- */
#define GSL_DBL_EPSILON 2.2204460492503131e-16
#define GSL_SQRT_DBL_EPSILON 1.4901161193847656e-08
#define GSL_ROOT5_DBL_EPSILON 7.4009597974140505e-04
#define GSL_NAN (__builtin_nan(""))

/* ---- containers ---- */
typedef struct { size_t size; double *data; } gsl_block;

typedef struct {
  size_t size;
  size_t stride;
  double *data;
  gsl_block *block;
  int owner;
} gsl_vector;
typedef struct { gsl_vector vector; } _gsl_vector_view;
typedef _gsl_vector_view gsl_vector_view;

typedef struct {
  size_t size1;
  size_t size2;
  size_t tda;
  double *data;
  gsl_block *block;
  int owner;
} gsl_matrix;
typedef struct { gsl_matrix matrix; } _gsl_matrix_view;
typedef _gsl_matrix_view gsl_matrix_view;

typedef struct { size_t size; size_t *data; } gsl_permutation;

gsl_vector *gsl_vector_alloc(const size_t n);
gsl_vector *gsl_vector_calloc(const size_t n);
void gsl_vector_free(gsl_vector *v);
gsl_vector_view gsl_vector_view_array(double *base, size_t n);
gsl_vector_view gsl_vector_view_array_with_stride(double *base, size_t stride, size_t n);
static inline double gsl_vector_get(const gsl_vector *v, const size_t i) { return v->data[i * v->stride]; }
static inline void gsl_vector_set(gsl_vector *v, const size_t i, double x) { v->data[i * v->stride] = x; }

gsl_matrix *gsl_matrix_alloc(const size_t n1, const size_t n2);
gsl_matrix *gsl_matrix_calloc(const size_t n1, const size_t n2);
void gsl_matrix_free(gsl_matrix *m);
gsl_matrix_view gsl_matrix_view_array(double *base, const size_t n1, const size_t n2);
gsl_matrix_view gsl_matrix_view_array_with_tda(double *base, const size_t n1, const size_t n2, const size_t tda);
gsl_matrix_view gsl_matrix_submatrix(gsl_matrix *m, const size_t i, const size_t j, const size_t n1, const size_t n2);
gsl_vector_view gsl_matrix_row(gsl_matrix *m, const size_t i);
gsl_vector_view gsl_matrix_column(gsl_matrix *m, const size_t j);
static inline double gsl_matrix_get(const gsl_matrix *m, const size_t i, const size_t j) { return m->data[i * m->tda + j]; }
static inline void gsl_matrix_set(gsl_matrix *m, const size_t i, const size_t j, double x) { m->data[i * m->tda + j] = x; }

gsl_permutation *gsl_permutation_alloc(const size_t n);
void gsl_permutation_init(gsl_permutation *p);
void gsl_permutation_free(gsl_permutation *p);
static inline size_t gsl_permutation_get(const gsl_permutation *p, const size_t i) { return p->data[i]; }

/* ---- rng ---- */
typedef struct {
  const char *name;
  unsigned long int max;
  unsigned long int min;
  size_t size;
  void (*set)(void *state, unsigned long int seed);
  unsigned long int (*get)(void *state);
  double (*get_double)(void *state);
} gsl_rng_type;

typedef struct {
  const gsl_rng_type *type;
  void *state;
} gsl_rng;

extern const gsl_rng_type *gsl_rng_mt19937;
extern const gsl_rng_type *gsl_rng_default;
extern unsigned long int gsl_rng_default_seed;
const gsl_rng_type *gsl_rng_env_setup(void);
gsl_rng *gsl_rng_alloc(const gsl_rng_type *T);
void gsl_rng_set(const gsl_rng *r, unsigned long int seed);
void gsl_rng_free(gsl_rng *r);
unsigned long int gsl_rng_get(const gsl_rng *r);
double gsl_rng_uniform(const gsl_rng *r);
unsigned long int gsl_rng_uniform_int(const gsl_rng *r, unsigned long int n);
void gsl_ran_shuffle(const gsl_rng *r, void *base, size_t nmembm, size_t size);

#ifdef __cplusplus
}
#endif
#endif /* GSL_SINTERP_SYSTEM_GSL */
#endif
