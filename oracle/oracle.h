/*
 * oracle.h -- CPU oracle for the scattered-interpolation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and there only as the checker / timed CPU baseline.
 * The product library (gsl-scattered-interpolation_amd/) never links it.
 *
 * It is a plain-C restatement of the reference's algorithm (GSL branch
 * smithzvk/gsl-scattered-interpolation); every function cites the reference
 * file:line it follows.  Compile with -O2 -ffp-contract=off (x86-64 baseline,
 * no FMA) so each *, -, / is a separately rounded fp64 operation like the
 * reference build.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - barycentric / Delaunay-DAG path: pinned by the reference's own asserted
 *     known answers (interpolation/scattered_interp_example.c:51-77) and by the
 *     outputs of the reference captured at survey time (SURVEY.md section 4:
 *     leaf indices, vertex lists, %.17g values, node counts for 3 tree
 *     configurations; BASELINE.md: 449 445 nodes / 100 001 leaves at N=50 000).
 *   - dense solvers: pinned by the reference's Hilbert / Vandermonde exact
 *     solutions and tolerances (linalg/test.c:378-404,411-494,3328-3402) and the
 *     random-SPD reconstruction test (linalg/test_cholesky.c:59-169).
 *   - RBF fill / eval: the reference holds no RBF code and no RBF test, so the
 *     RBF kernels themselves are "parity unpinned" (composition of libm + the
 *     pinned solvers + naive j-ascending summation).
 *   The reference cannot be compiled here without hand-writing its
 *   autoconf-generated config.h / gsl_version.h, so no oracle/_ref exists.
 */
#ifndef SINTERP_ORACLE_H
#define SINTERP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes mirrored from err/gsl_errno.h:40-74 ---- */
#define ORACLE_SUCCESS 0
#define ORACLE_FAILURE (-1)
#define ORACLE_EDOM 1
#define ORACLE_EINVAL 4
#define ORACLE_ENOTSQR 20

/* ---- machine constants, gsl_machine.h:17-21 ---- */
#define ORACLE_DBL_EPSILON 2.2204460492503131e-16
#define ORACLE_SQRT_DBL_EPSILON 1.4901161193847656e-08
#define ORACLE_ROOT5_DBL_EPSILON 7.4009597974140505e-04

/* ======================= rng (oracle_rng.c) ========================== */
typedef struct oracle_mt oracle_mt;
oracle_mt *oracle_mt_alloc(unsigned long seed);          /* rng/mt.c:131 */
void oracle_mt_free(oracle_mt *r);
unsigned long oracle_mt_get(oracle_mt *r);               /* rng/mt.c:79 */
unsigned long oracle_mt_uniform_int(oracle_mt *r, unsigned long n); /* rng/gsl_rng.h:190 */
void oracle_shuffle_sizet(oracle_mt *r, size_t *base, size_t n);    /* randist/shuffle.c:69 */

/* ======================= dense linalg (oracle_linalg.c) ============== */
/* all matrices row-major with leading dimension lda (gsl_matrix tda) */
int oracle_lu_decomp(size_t n, double *a, size_t lda, size_t *perm, int *signum);   /* linalg/lu.c:59 */
int oracle_lu_singular(size_t n, const double *lu, size_t lda);                      /* linear_simplex_util.h:14 */
int oracle_lu_svx(size_t n, const double *lu, size_t lda, const size_t *perm, double *x); /* linalg/lu.c:166 */
int oracle_cholesky_decomp1(size_t n, double *a, size_t lda);                        /* linalg/cholesky.c:88 */
int oracle_cholesky_svx(size_t n, const double *llt, size_t lda, double *x);         /* linalg/cholesky.c:163 */

/* solver breadth */
int oracle_cholesky_decomp2(size_t n, double *a, size_t lda, double *s);                        /* linalg/cholesky.c:392 */
int oracle_cholesky_svx2(size_t n, const double *llt, size_t lda, const double *s, double *x); /* linalg/cholesky.c:431 */
int oracle_cholesky_rcond(size_t n, const double *llt, size_t lda, double *rcond, double *work /* 3n */); /* :499, condest.c:95 */
int oracle_lu_refine(size_t n, const double *a, size_t lda, const double *lu, size_t ldlu, const size_t *perm,
                     const double *b, double *x, double *work);                                  /* linalg/lu.c:204 */
int oracle_pcholesky_decomp(size_t n, double *a, size_t lda, size_t *perm);                     /* linalg/pcholesky.c:71 */
int oracle_pcholesky_svx(size_t n, const double *ldlt, size_t lda, const size_t *perm, double *x); /* linalg/pcholesky.c:190 */
int oracle_pcholesky_decomp2(size_t n, double *a, size_t lda, size_t *perm, double *s);            /* linalg/pcholesky.c:231 */
int oracle_pcholesky_svx2(size_t n, const double *ldlt, size_t lda, const size_t *perm, const double *s, double *x); /* :314 */
int oracle_pcholesky_rcond(size_t n, const double *ldlt, size_t lda, const size_t *perm, double *rcond, double *work); /* :472; work 3n */

/* ======================= simplex tree (oracle_simplex.c) ============= */
#define ORACLE_TREE_DEFAULT 0
#define ORACLE_TREE_NOSTANDARDIZE 1   /* linear_simplex.h:111 */
#define ORACLE_TREE_ISOSCALE 2        /* linear_simplex.h:112 */

enum { ORACLE_LEAF = 0, ORACLE_SUB_DPLUS1 = 1, ORACLE_SUB_D = 2, ORACLE_SUB_2 = 3 };

typedef struct oracle_tree {
  int dim;
  int n_nodes, cap_nodes;
  int *type;    /* node_type per node                       (linear_simplex.h:8-21)  */
  int *pidx;    /* (dim+1) vertex ids per node; <0 = seed   (linear_simplex.h:64)    */
  int *links;   /* (dim+1) per node: children / neighbours  (linear_simplex.h:62)    */
  double *seed; /* (dim+1) x dim cage vertices, row-major   (linear_simplex.c:217-260) */
  int n_points, max_points;
  double *shift, *scale, *min, *max;
  size_t *shuffle;                     /* insertion index -> data row */
  /* persistent accelerator state, linear_simplex.h:23-29 */
  double *acc_mat;   /* dim x dim LU */
  size_t *acc_perm;
  double *acc_coords;
  int acc_current;
  /* statistics (not in the reference) */
  long stat_tests, stat_depth, stat_maxdepth, stat_fallbacks;
} oracle_tree;

oracle_tree *oracle_tree_alloc(int dim, int n_points);                       /* linear_simplex.c:53 */
void oracle_tree_free(oracle_tree *t);                                       /* linear_simplex.c:108 */
int oracle_tree_init(oracle_tree *t, const double *data, size_t n, size_t tda,
                     const double *min, const double *max, int flags, oracle_mt *rng); /* :134 */
int oracle_find_leaf(oracle_tree *t, const double *data, size_t tda, const double *point); /* :331; -1 = outside cage */
int oracle_insert_point(oracle_tree *t, int leaf, const double *data, size_t tda);      /* :404 */
int oracle_contains_point(oracle_tree *t, int node, const double *data, size_t tda, const double *point); /* :653 */
int oracle_bary_coords(oracle_tree *t, int node, const double *data, size_t tda, const double *point);    /* :607 */
int oracle_in_hypersphere(oracle_tree *t, int node, const double *data, size_t tda, int idx);             /* :495 */
double oracle_interp_point(oracle_tree *t, int leaf, const double *data, size_t tda,
                           const double *response, size_t rstride, const double *point);                  /* :678 */
/* batch driver (loops the per-point reference API; not in the reference) */
int oracle_bary_eval_many(oracle_tree *t, const double *data, size_t tda,
                          const double *response, size_t rstride,
                          const double *targets, size_t m, size_t ttda,
                          double *values, int *leaf);
/* structural self checks restated from linear_simplex_integrity_check.c:62-160 */
int oracle_check_leaf_nodes(oracle_tree *t);
int oracle_check_delaunay(oracle_tree *t, const double *data, size_t tda);
uint64_t oracle_tree_hash(const oracle_tree *t);   /* FNV-1a over type/pidx/links */
/* the gnuplot dumps of linear_simplex_integrity_check.c:170-284, written by the reference's RECURSIVE walk over the
   leaf adjacency (:62-119: pre-order, neighbours by link index); any of the three file names may be NULL */
int oracle_output_triangulation(oracle_tree *t, const double *data, size_t tda, const double *response, size_t rstride,
                                int standardize_output, const char *lines_filename, const char *points_filename,
                                const char *circles_filename);

/* imported triangulations (no reference walk exists: parity unpinned); the reference's per-triangle arithmetic on
   explicit vertex rows `tri[3]` (the LAST one is the origin), linear_simplex.c:607-711 */
int oracle_mesh_coords(const double *data, size_t tda, const double *shift, const double *scale, const int *tri,
                       const double *point, double *coords);
int oracle_mesh_contains(const double *data, size_t tda, const double *shift, const double *scale, const int *tri,
                         const double *point);
double oracle_mesh_interp(const double *data, size_t tda, const double *shift, const double *scale, const int *tri,
                          const double *response, size_t rstride, const double *point);
int oracle_mesh_locate(const double *data, size_t tda, const double *shift, const double *scale, const int *tris, size_t nt,
                       const double *point, int *n_containing);

/* ======================= RBF harness (oracle_rbf.c) ================== */
#define ORACLE_RBF_GAUSSIAN 0   /* phi = exp(-(eps r)^2)                    */
#define ORACLE_RBF_TPS 1        /* phi = r^2 ln r = 0.5 r^2 ln r^2, phi(0)=0 */
#define ORACLE_RBF_WENDLAND 2   /* phi = (1 - eps r)_+^4 (4 eps r + 1)      */
double oracle_rbf_phi(int kind, double eps, double r2);
void oracle_rbf_fill(int kind, double eps, const double *x, size_t n, int dim, size_t tda,
                     double *phi, size_t lda);
/* fill + reference solver route (Cholesky for Gaussian, LU for TPS) -> weights */
int oracle_rbf_solve(int kind, double eps, const double *x, size_t n, int dim, size_t tda,
                     const double *f, double *w);
void oracle_rbf_eval(int kind, double eps, const double *x, size_t n, int dim, size_t tda,
                     const double *w, const double *y, size_t m, size_t ytda, double *s);

/* thin-plate spline with its affine tail: pivoted LU of the (n + dim + 1) saddle system; c[0 .. dim] in raw coordinates */
int oracle_rbf_solve_affine(int kind, double eps, const double *x, size_t n, int dim, size_t tda,
                            const double *f, double *w, double *c);
void oracle_rbf_eval_affine(int kind, double eps, const double *c, const double *x, size_t n, int dim, size_t tda,
                            const double *w, const double *y, size_t m, size_t ytda, double *s);
/* ordinary kriging (dual form) on the same kernels: K = Phi + nugget I, Cholesky (pivoted LDL^T when K is only
   semi-definite); no reference code (README:24), parity unpinned */
int oracle_krige_solve(int kind, double eps, double nugget, const double *x, size_t n, int dim, size_t tda,
                       const double *f, double *w, double *mean);
void oracle_krige_eval(int kind, double eps, double mean, const double *x, size_t n, int dim, size_t tda,
                       const double *w, const double *y, size_t m, size_t ytda, double *s);

/* ======================= synthetic inputs (oracle_synth.c) =========== */
/* u(k) = (splitmix64(seed ^ k) >> 11) * 2^-53 */
uint64_t oracle_splitmix64(uint64_t z);
void oracle_synth_centres(double *x, size_t n, int dim);            /* seed 0xC0FFEE01 */
void oracle_synth_targets(double *y, size_t first, size_t m, int dim); /* seed 0xC0FFEE02, 0.02+0.96u */
void oracle_synth_response(const double *x, size_t n, int dim, double *f);

#ifdef __cplusplus
}
#endif
#endif
