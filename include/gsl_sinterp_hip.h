/*
 * gsl_sinterp_hip.h -- C-ABI of the MI355X (gfx950) kernels behind the
 * scattered-interpolation hot path.  Plain pointers and sizes only; every
 * `d_*` argument is a DEVICE pointer (hipMalloc / torch data_ptr), every
 * `h_*` argument a host pointer.  All entry points return a GSL status code
 * (include/gsl_sinterp_compat.h) and never abort; the text of the last failure
 * is kept per context (gsl_sinterp_hip_last_error).
 *
 * Work is enqueued on the context's stream and is asynchronous unless the
 * entry point has a host-side output (`info`, `signum`), in which case it
 * synchronises the stream before returning.
 *
 * Reference routines each entry point replaces (paths under the reference
 * tree, smithzvk/gsl-scattered-interpolation):
 *   tree_pack + bary_eval   interpolation/linear_simplex.c:331-402 (find_leaf/_find_leaf),
 *                           :607-651 (calculate_bary_coords), :653-676 (contains_point),
 *                           :678-711 (interp_point); linalg/lu.c:59-201 at d=2
 *   rbf_fill                (no reference code; README:18-26) Phi_ij = phi(|x_i-x_j|)
 *   cholesky_decomp1        linalg/cholesky.c:88-131  (gsl_linalg_cholesky_decomp1)
 *   cholesky_svx            linalg/cholesky.c:163-185 (gsl_linalg_cholesky_svx)
 *   lu_decomp / lu_svx      linalg/lu.c:59-124, :166-201 (gsl_linalg_LU_decomp / _svx)
 *   rbf_eval                (no reference code) s(y) = sum_j w_j phi(|y-x_j|)
 *   tree_check              interpolation/linear_simplex_integrity_check.c:62-119 (_check_leaf_nodes),
 *                           :134-160 (_check_delaunay)
 */
#ifndef GSL_SINTERP_HIP_H
#define GSL_SINTERP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsl_sinterp_hip_ctx gsl_sinterp_hip_ctx;

/* radial kernels */
#define GSL_SINTERP_RBF_GAUSSIAN 0 /* phi(r) = exp(-(eps r)^2)                      */
#define GSL_SINTERP_RBF_TPS 1      /* phi(r) = r^2 ln r = 0.5 r^2 ln r^2, phi(0)=0  */
#define GSL_SINTERP_RBF_WENDLAND 2 /* phi(r) = (1 - eps r)_+^4 (4 eps r + 1): Wendland's C2 function, compact support of
                                      radius 1/eps, positive definite for dim <= 3 (the reference's README:18-26 lists
                                      compactly supported kernels as future work); Cholesky route, sweep with EXACT culling */

/* ---- context / memory --------------------------------------------------- */
int gsl_sinterp_hip_device_count(void); /* 0 when no GPU is visible */
/* stream: a hipStream_t owned by the caller (e.g. torch's current stream); NULL is
   the device's default stream, i.e. ordered with the caller's other default-stream
   work.  ctx_own_stream switches the context to a private non-blocking stream
   (for overlap with other streams; the caller then orders work explicitly). */
int gsl_sinterp_hip_ctx_create(gsl_sinterp_hip_ctx **ctx, int device, void *stream);
int gsl_sinterp_hip_ctx_own_stream(gsl_sinterp_hip_ctx *ctx);
void gsl_sinterp_hip_ctx_destroy(gsl_sinterp_hip_ctx *ctx);
int gsl_sinterp_hip_ctx_device(const gsl_sinterp_hip_ctx *ctx);   /* the ordinal the context was created on, -1 for NULL */
int gsl_sinterp_hip_sync(gsl_sinterp_hip_ctx *ctx);
const char *gsl_sinterp_hip_last_error(const gsl_sinterp_hip_ctx *ctx);
int gsl_sinterp_hip_malloc(gsl_sinterp_hip_ctx *ctx, void **d_ptr, size_t bytes);
int gsl_sinterp_hip_free(gsl_sinterp_hip_ctx *ctx, void *d_ptr);
int gsl_sinterp_hip_h2d(gsl_sinterp_hip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int gsl_sinterp_hip_d2h(gsl_sinterp_hip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* hipEvent pair on the context's stream (bench / roofline timing) */
int gsl_sinterp_hip_timer_start(gsl_sinterp_hip_ctx *ctx);
int gsl_sinterp_hip_timer_stop(gsl_sinterp_hip_ctx *ctx, float *h_ms);

/* ---- device groups: target shards over several GPUs of one node -------- */
/* One context per listed device (each on a private stream, so one host thread keeps all devices
   busy) and ONE collective: gsl_sinterp_hip_group_broadcast replicates a model buffer from member 0
   to every member -- ncclBroadcast over xGMI on a communicator made by ncclCommInitAll (RCCL is
   bound with dlopen when the first multi-device group is created).  A list that names one ordinal
   twice (one-GPU test boxes) or GSL_SINTERP_NO_RCCL=1 replicates with hipMemcpyPeerAsync instead;
   gsl_sinterp_hip_group_transport says which ("rccl" / "peer-copy" / "none").  There is no
   reduction and no all-to-all on the path; the factorisation runs on member 0 only. */
typedef struct gsl_sinterp_hip_group gsl_sinterp_hip_group;
int gsl_sinterp_hip_group_create(gsl_sinterp_hip_group **grp, const int *devices, int n_devices);
void gsl_sinterp_hip_group_destroy(gsl_sinterp_hip_group *grp);
int gsl_sinterp_hip_group_size(const gsl_sinterp_hip_group *grp);
int gsl_sinterp_hip_group_device(const gsl_sinterp_hip_group *grp, int member);
gsl_sinterp_hip_ctx *gsl_sinterp_hip_group_ctx(gsl_sinterp_hip_group *grp, int member);
const char *gsl_sinterp_hip_group_transport(const gsl_sinterp_hip_group *grp);
const char *gsl_sinterp_hip_group_last_error(const gsl_sinterp_hip_group *grp);
/* d_bufs[i]: `bytes` bytes on member i; member 0's content goes to all (stream ordered, asynchronous) */
int gsl_sinterp_hip_group_broadcast(gsl_sinterp_hip_group *grp, void *const *d_bufs, size_t bytes);
/* the shard rule: contiguous ceil(m/world)-sized shards, every target exactly once */
void gsl_sinterp_hip_shard_bounds(size_t m_total, int world, int rank, size_t *first, size_t *count);
/* asynchronous copies on the context's stream + pinned host staging for them */
int gsl_sinterp_hip_h2d_async(gsl_sinterp_hip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int gsl_sinterp_hip_d2h_async(gsl_sinterp_hip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int gsl_sinterp_hip_host_alloc(void **h_ptr, size_t bytes);
void gsl_sinterp_hip_host_free(void *h_ptr);
/* copy pipe of one context: an upload and a download stream beside the context's stream, so that the chunks of a host
   batch overlap H2D | sweep | D2H.  upload: later work on the context's stream waits for the copy; download: the copy
   waits for the work enqueued on the context's stream up to a mark (or so far).  Pinned host buffers, untouched until pipe_sync (which
   also synchronises the context's stream).  At most 64 copies between two syncs. */
typedef struct gsl_sinterp_hip_pipe gsl_sinterp_hip_pipe;
int gsl_sinterp_hip_pipe_create(gsl_sinterp_hip_ctx *ctx, gsl_sinterp_hip_pipe **out);
void gsl_sinterp_hip_pipe_destroy(gsl_sinterp_hip_pipe *pipe);
int gsl_sinterp_hip_pipe_upload(gsl_sinterp_hip_pipe *pipe, void *d_dst, const void *h_src, size_t bytes);
int gsl_sinterp_hip_pipe_mark(gsl_sinterp_hip_pipe *pipe, int *mark);      /* "the context's stream up to here" */
int gsl_sinterp_hip_pipe_download(gsl_sinterp_hip_pipe *pipe, int mark, void *h_dst, const void *d_src, size_t bytes);   /* mark < 0: now */
int gsl_sinterp_hip_pipe_sync(gsl_sinterp_hip_pipe *pipe);
/* number of negative entries of d_v[0 .. m) (-1 = outside the cage / mesh); synchronises the context's stream */
int gsl_sinterp_hip_count_negative(gsl_sinterp_hip_ctx *ctx, const int *d_v, size_t m, long long *h_count);

/* ---- barycentric evaluation over a host-built Delaunay history DAG ------- */
/* One 64-byte record per DAG node (see DESIGN.md "HBM layout"). */
#define GSL_SINTERP_TREE_RECORD_BYTES 64
#define GSL_SINTERP_TREE_LEAFTAB_BYTES 32

/* Build the node records on the device.  d_type[n_nodes] (0 leaf, 1 sub_{d+1},
   2 sub_d), d_pidx/d_links [3*n_nodes] exactly as the reference keeps them
   (linear_simplex.h:31-59), d_points [2*n_points] in INSERTION order
   (row shuffle[i] of the data matrix), h_geom[10] = seed_points(3x2, row-major),
   shift(2), scale(2).  The per-node 2x2 LU is computed with the reference's
   operation sequence (no FMA contraction) so it is bit-identical to the host. */
int gsl_sinterp_hip_tree_pack(gsl_sinterp_hip_ctx *ctx, int n_nodes, const int *d_type,
                              const int *d_pidx, const int *d_links, int n_points,
                              const double *d_points, const double *h_geom, void *d_records);
/* Per-node table of the three vertex responses (+ seed mask); d_response is in
   insertion order (response[shuffle[i]]). */
int gsl_sinterp_hip_tree_bind(gsl_sinterp_hip_ctx *ctx, int n_nodes, const int *d_pidx,
                              int n_points, const double *d_response, void *d_leaftab);
/* Locate + interpolate m targets (row k at d_targets + k*ttda).  d_leaf may be
   NULL.  Target outside the cage: leaf = -1, value = NaN, and *h_n_outside
   (may be NULL; forces a sync when given) counts them -> GSL_EDOM. */
int gsl_sinterp_hip_bary_eval(gsl_sinterp_hip_ctx *ctx, int n_nodes, const void *d_records,
                              const void *d_leaftab, const double *h_scale,
                              const double *d_targets, size_t m, size_t ttda, double *d_values,
                              int *d_leaf, long long *h_n_outside);

/* Imported triangulation (no DAG): records + seed grid, then locate + interpolate.  d_tri / d_nbr [3 n_tri] (vertex ids =
   rows of d_points; neighbour across the edge opposite vertex k, -1 = hull), d_points [2 n_points] in ROW order,
   h_geom[8] = shift(2), scale(2), bounding box lo0, lo1, hi0, hi1 of the points; G: cells per axis of the seed grid,
   d_seed: 2 G^2 ints.  Leaf table: gsl_sinterp_hip_tree_bind with d_pidx = d_tri.  convex = 0: a walk stopped by a
   hull edge is resolved by an exhaustive scan instead of "outside". */
int gsl_sinterp_hip_mesh_pack(gsl_sinterp_hip_ctx *ctx, int n_tri, const int *d_tri, const int *d_nbr, int n_points,
                              const double *d_points, const double *h_geom, int G, void *d_records, int *d_seed);
int gsl_sinterp_hip_mesh_eval(gsl_sinterp_hip_ctx *ctx, int n_tri, const void *d_records, const void *d_leaftab,
                              const int *d_seed, int G, const double *h_geom, int convex, const double *d_targets,
                              size_t m, size_t ttda, double *d_values, int *d_tri, long long *h_n_outside);

/* Device-side integrity checks of a DAG given as raw arrays (same arguments as tree_pack):
   what & 1: _check_leaf_nodes  (interpolation/linear_simplex_integrity_check.c:62-119), one thread per leaf;
   what & 2: _check_delaunay    (:134-160; circumsphere per linear_simplex.c:555-605), leaves x points.
   Counts of violating leaves / (leaf, point) pairs come back in *h_leaf_violations / *h_delaunay_violations
   (0 = the reference's asserts would all hold); h_first[3] (may be NULL) = first bad leaf of either check
   and one witness point, -1 when clean.  Synchronises the stream. */
int gsl_sinterp_hip_tree_check(gsl_sinterp_hip_ctx *ctx, int n_nodes, const int *d_type, const int *d_pidx,
                               const int *d_links, int n_points, const double *d_points, const double *h_geom,
                               int what, long long *h_leaf_violations, long long *h_delaunay_violations,
                               int *h_first);

/* ---- RBF: fill, dense solve, evaluation sweep ----------------------------- */
int gsl_sinterp_hip_rbf_fill(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x,
                             size_t n, int dim, size_t xtda, double *d_phi, size_t lda);
/* In-place A = L L^T, L in the lower triangle, original A kept in the strict
   upper triangle; *h_info = 0, or j+1 when pivot j <= 0 (-> GSL_EDOM). */
int gsl_sinterp_hip_cholesky_decomp1(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda,
                                     int *h_info);
int gsl_sinterp_hip_cholesky_svx(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_llt,
                                 size_t lda, double *d_x);
/* PA = LU with partial pivoting; d_perm[n] (int32) as gsl_permutation content. */
int gsl_sinterp_hip_lu_decomp(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda,
                              int *d_perm, int *h_signum);
int gsl_sinterp_hip_lu_svx(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_lu, size_t lda,
                           const int *d_perm, double *d_x);
/* ---- solver breadth ---------------------------------- */
/* gsl_linalg_cholesky_decomp2 (linalg/cholesky.c:392-429): S_i = 1/sqrt(A_ii) -> d_s, A <- diag(S) A diag(S),
   then decomp1; svx2 (:431-462): x *= S, two sweeps, x *= S. */
int gsl_sinterp_hip_cholesky_decomp2(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, double *d_s,
                                     int *h_info);
int gsl_sinterp_hip_cholesky_svx2(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_llt, size_t lda,
                                  const double *d_s, double *d_x);
/* gsl_linalg_cholesky_rcond (linalg/cholesky.c:499-537, linalg/condest.c:95-188): reciprocal 1-norm condition
   number of the matrix whose factor (with the original kept in the strict upper triangle) is d_llt. */
int gsl_sinterp_hip_cholesky_rcond(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_llt, size_t lda,
                                   double *h_rcond);
/* gsl_linalg_LU_refine (linalg/lu.c:204-252): one step of iterative refinement of d_x; d_work: n doubles. */
int gsl_sinterp_hip_lu_refine(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_a, size_t lda, const double *d_lu,
                              size_t ldlu, const int *d_perm, const double *d_b, double *d_x, double *d_work);
/* gsl_linalg_pcholesky_decomp / _svx (linalg/pcholesky.c:71-229): P A P^T = L D L^T with diagonal pivoting, for
   symmetric positive SEMI-definite matrices; L below the diagonal, D on it, the original in the strict upper
   triangle; d_perm[n] (int32) as gsl_permutation content.  Bit-identical to the reference-order CPU algorithm. */
int gsl_sinterp_hip_pcholesky_decomp(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *d_perm);
int gsl_sinterp_hip_pcholesky_svx(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_ldlt, size_t lda,
                                  const int *d_perm, double *d_x);
/* gsl_linalg_pcholesky_decomp2 / _svx2 (linalg/pcholesky.c:231-353): the matrix is kept UNSCALED in the strict upper
   triangle, S_i = 1/sqrt(A_ii) -> d_s, pivoted LDL^T of diag(S) A diag(S); svx2: x *= S, svx, x *= S.
   gsl_linalg_pcholesky_rcond (:472-580): reciprocal 1-norm condition number of the matrix in the upper triangle (its
   diagonal rebuilt from L D L^T), for the UNSCALED decomposition like the reference's own test (test_cholesky.c:675-687). */
int gsl_sinterp_hip_pcholesky_decomp2(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *d_perm, double *d_s);
int gsl_sinterp_hip_pcholesky_svx2(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_ldlt, size_t lda, const int *d_perm,
                                   const double *d_s, double *d_x);
int gsl_sinterp_hip_pcholesky_rcond(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_ldlt, size_t lda, const int *d_perm,
                                    double *h_rcond);

int gsl_sinterp_hip_rbf_eval(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x,
                             size_t n, int dim, size_t xtda, const double *d_w,
                             const double *d_y, size_t m, size_t ytda, double *d_s);

/* The same sweep for a model the caller promises not to change while it uses `model_id` (!= 0; a fresh id after
   every init): the Gaussian / Wendland sweep's per-model preprocessing -- Morton cell sort of the centres, packed
   {x, w} records, tile boxes -- is then done once per (model_id, d_x, d_w, n, dim, kind) and reused by later calls
   on this context (the single-point call behind gsl_sinterp_eval_e pays 1 launch instead of 9).  model_id = 0 is
   gsl_sinterp_hip_rbf_eval: nothing is assumed about the buffers, nothing cached.  Same bits either way.
   CONTRACT: the id vouches for the CONTENT of d_x / d_w, not just their addresses -- whoever rewrites either buffer (a
   new init into the same allocation, a checkpoint load, a broadcast) must pass a fresh id afterwards; reusing an id
   across re-filled buffers evaluates the OLD centres / weights silently.  The facade draws a new id in every init and
   fread (csrc/host/sinterp.c: next_model_id); ids are compared per context, so two contexts may use the same values. */
int gsl_sinterp_hip_rbf_eval_model(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x,
                                   size_t n, int dim, size_t xtda, const double *d_w,
                                   const double *d_y, size_t m, size_t ytda, double *d_s, unsigned long long model_id);

/* "init" of an RBF interpolant in one call: fill d_phi (n x n scratch, lda), solve Phi w = f
   with d_w holding f on entry and w on exit.  *h_route reports the solver used:
   1 Cholesky (Gaussian, SPD) -- 2 shifted-SPD Cholesky + rank-(d+1) Woodbury correction
   (thin-plate spline; values agree with the LU route to ~1e-13) -- 3 pivoted LU (the
   reference's route, taken when the shifted matrix is not SPD or GSL_SINTERP_FORCE_LU=1). */
int gsl_sinterp_hip_rbf_solve(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n,
                              int dim, size_t xtda, double *d_phi, size_t lda, double *d_w, int *h_route);

/* The same with an explicit solver (GSL_SINTERP_SOLVER_*) and an optional condition estimate:
     DEFAULT    the routes above;
     CHOLESKY2  scaled Cholesky, decomp2 + svx2 (route 4; SPD kernels);
     PCHOLESKY  pivoted LDL^T (route 5; semi-definite / nuggeted kernel matrices);
     LU_REFINE  pivoted LU + one gsl_linalg_LU_refine step (route 6; any kernel; needs a second n x n buffer,
                allocated inside).
   h_rcond (may be NULL): reciprocal condition number of the factored matrix for the Cholesky routes 1 and 4
   (of the SCALED matrix for 4), NaN for the others. */
#define GSL_SINTERP_SOLVER_DEFAULT 0
#define GSL_SINTERP_SOLVER_CHOLESKY2 1
#define GSL_SINTERP_SOLVER_PCHOLESKY 2
#define GSL_SINTERP_SOLVER_LU_REFINE 3
int gsl_sinterp_hip_rbf_solve_ex(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim,
                                 size_t xtda, double *d_phi, size_t lda, double *d_w, int solver, double *h_rcond,
                                 int *h_route);

/* Thin-plate spline WITH its affine tail (the "N + d + 1" system):
       [Phi P; P^T 0] [w; c] = [f; 0],  P = [1, x],     s(y) = sum_j w_j phi(|y - x_j|) + c_0 + sum_a c_a y_a.
   d_w holds f on entry and w on exit, h_poly[0 .. dim] receives c (raw coordinates).  Route 9: block elimination on the
   shifted SPD matrix of route 2 (one MFMA Cholesky, d + 2 right-hand sides, a (d+1) x (d+1) system on the host);
   route 10: pivoted LU of the augmented matrix (the reference route, linalg/lu.c:59-201; taken when the shifted
   matrix is not SPD or GSL_SINTERP_FORCE_LU=1) -- it needs d_phi with n + dim + 1 rows and lda >= n + dim + 1.
   kind must be GSL_SINTERP_RBF_TPS.  gsl_sinterp_hip_rbf_eval_affine = the RBF sweep + the polynomial. */
int gsl_sinterp_hip_rbf_solve_affine(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim,
                                     size_t xtda, double *d_phi, size_t lda, double *d_w, double *h_poly, int *h_route);
int gsl_sinterp_hip_rbf_eval_affine(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *h_poly, const double *d_x,
                                    size_t n, int dim, size_t xtda, const double *d_w, const double *d_y, size_t m,
                                    size_t ytda, double *d_s, unsigned long long model_id);

/* Ordinary kriging on a positive definite kernel used as covariance (GAUSSIAN or WENDLAND) with a nugget >= 0 (the
   reference's README:24 future list): d_w holds f on entry and the dual weights w on exit, *h_mean the estimated
   mean mu; s(y) = mu + sum_j w_j phi(|y - x_j|) (gsl_sinterp_hip_krige_eval = the RBF sweep + mu).  Route 7: Cholesky
   of K = Phi + nugget I with two right-hand sides; route 8: pivoted LDL^T when K is only semi-definite. */
int gsl_sinterp_hip_krige_solve(gsl_sinterp_hip_ctx *ctx, int kind, double eps, double nugget, const double *d_x, size_t n,
                                int dim, size_t xtda, double *d_phi, size_t lda, double *d_w, double *h_mean, int *h_route);
int gsl_sinterp_hip_krige_eval(gsl_sinterp_hip_ctx *ctx, int kind, double eps, double mean, const double *d_x, size_t n,
                               int dim, size_t xtda, const double *d_w, const double *d_y, size_t m, size_t ytda,
                               double *d_s, unsigned long long model_id);

/* Level-3 building block of both factorisations, exposed for tests and roofline
   measurement (role of gsl_blas_dgemm / dsyrk, blas/blas.c:1334,1649):
     C[m x n] -= A[m x k] * B^T  (b_is_kn = 0, B stored [n][k])
     C[m x n] -= A[m x k] * B    (b_is_kn = 1, B stored [k][n])
   lower_only != 0: C is square and only its lower triangle is updated. */
int gsl_sinterp_hip_gemm_minus(gsl_sinterp_hip_ctx *ctx, size_t m, size_t n, size_t k, const double *d_a,
                               size_t lda, const double *d_b, size_t ldb, int b_is_kn, double *d_c,
                               size_t ldc, int lower_only);

/* developer / test hook: targets of the last large barycentric batch on this context that the certified leaf walk left to the
   exact DAG kernel (meaningful when *h_leafwalk = 1: tree_pack built the locator data for the records in use) */
int gsl_sinterp_hip_bary_last_queue(gsl_sinterp_hip_ctx *ctx, unsigned *h_queued, int *h_leafwalk);

/* ---- gridded front-end: the targets of an n0 x n1 grid generated in HBM ---- */
/* row (i*n1 + j) of d_y (packed M x 2) = (min0 + step0*i, min1 + step1*j): the loop of
   interpolation/scattered_interp_example.c:183-197, same operations, so the same bits */
int gsl_sinterp_hip_grid_targets(gsl_sinterp_hip_ctx *ctx, double min0, double step0, size_t n0, double min1,
                                 double step1, size_t n1, double *d_y);

/* ---- synthetic clouds generated in HBM (bench / tests) ------- */
int gsl_sinterp_hip_synth_unit(gsl_sinterp_hip_ctx *ctx, uint64_t seed, uint64_t first,
                               double offset, double span, double *d_out, size_t count);

#ifdef __cplusplus
}
#endif
#endif
