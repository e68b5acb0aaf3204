/*
 * common.h -- shared by the HIP translation units behind include/gsl_sinterp_hip.h.
 */
#ifndef SINTERP_HIP_COMMON_H
#define SINTERP_HIP_COMMON_H

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "gsl_sinterp_hip.h"

/* GSL status codes used here (include/gsl_sinterp_compat.h, err/gsl_errno.h:40-74) */
enum { ST_SUCCESS = 0, ST_FAILURE = -1, ST_EDOM = 1, ST_EFAULT = 3, ST_EINVAL = 4, ST_EFAILED = 5, ST_ENOMEM = 8,
       ST_EBADLEN = 19 };

struct gsl_sinterp_hip_ctx {
  int device;
  hipStream_t stream;
  int owns_stream;
  hipEvent_t ev0, ev1;
  void *d_scratch;          /* small persistent scratch: counters / info words */
  size_t scratch_bytes;
  void *d_work;             /* growable workspace (factorisations) */
  size_t work_bytes;
  void *d_aux;              /* growable buffer of the solver route (right-hand sides, polynomial block) */
  size_t aux_bytes;
  void *d_inv;              /* inverted 64x64 diagonal blocks of the triangular sweeps */
  size_t inv_bytes;
  void *d_sort;             /* target permutation + cell counters (sort.hip) */
  size_t sort_bytes;
  void *d_sort2;            /* the same for the centres of the Gaussian sweep */
  size_t sort2_bytes;
  void *d_cent;             /* cell-ordered packed centres {x, w} + tile boxes of the Gaussian sweep */
  size_t cent_bytes;
  /* key of the packed centres currently in d_cent (gsl_sinterp_hip_rbf_eval_model); id 0 = nothing cached */
  struct { unsigned long long id; const void *x, *w; size_t n, xtda; int dim, kind; } cent_key;
  void *d_walk;             /* affine walk records + queue of the barycentric walk (bary.hip), rebuilt per batch */
  size_t walk_bytes;
  hipStream_t side_stream;  /* bary.hip: independent kernels of one evaluation run beside the main stream */
  hipEvent_t side_ev[4];    /* fork / join events of the side stream (created with the stream) */
  /* hipGraph cache: the recursive factorisation drivers issue ~1-2k small, fully static
     launches; they are captured once per (routine, n, lda, pointers) and replayed */
  hipStream_t cap_stream;
  struct GraphSlot { hipGraphExec_t exec; size_t n, lda; const void *p0, *p1, *work; } graph[4];
  int use_graphs;
  /* stream-K GEMM (gemm.hip): one partial-tile slot + one flag per persistent workgroup */
  double *d_sk_partial;
  unsigned *d_sk_flags;
  int sk_wgs;               /* persistent workgroups = CUs of the device; 0 = not prepared */
  /* dataflow sweeps (chol.hip): [0] = epoch of the last completed sweep, [1 + J] = epoch in which
     block J was last published.  Never reset (no memset node in the captured graphs): a sweep
     publishes with epoch + 1 and its last block advances the epoch. */
  unsigned *d_tf;
  size_t tf_count;
  unsigned long long *d_xq; /* hand-off buffer of the sweeps: per entry {epoch|lo32}, {epoch|hi32} */
  void *d_lu_coop;          /* cooperative LU panel: [generation, abort] + exchange slots (lu.hip) */
  /* jump table of the barycentric walk built by tree_pack over the data's bounding box (bary.hip):
     [64 B box keys][G*G node indices]; valid for records == jump_rec with jump_nodes nodes */
  void *d_jumpt;
  size_t jumpt_bytes;
  /* leaf-adjacency locator of the barycentric sweep (bary.hip, "Certified leaf walk"): seed grid + per-leaf line lists of the
     final edges and of the historic (flipped-away) edges that cross the leaf, valid for records == lw_rec */
  void *d_lw_a;             /* [64 B consts][seed Gs^2][off n_nodes + 1 + scan scratch] */
  void *d_lw_lines;         /* 3 doubles per line */
  size_t lw_a_bytes, lw_lines_bytes;
  const void *lw_rec;
  int lw_nodes, lw_Gs, lw_last;   /* lw_last: the last large batch went through the leaf walk */
  double lw_c[2], lw_lo[2], lw_w[2];
  const void *jump_rec;
  int jump_nodes, jump_G;
  int excl_depth;           /* nesting of sinterp_exclusive_begin/end */
  char err[512];
};

/* Graph helpers (shim.hip).  sinterp_graph_lookup returns 1 and launches the cached
   graph on the context's stream when slot `which` matches the key; otherwise 0.
   Between sinterp_capture_begin / _end every launch on ctx->stream is recorded. */
int sinterp_graph_try_launch(gsl_sinterp_hip_ctx *ctx, int which, size_t n, size_t lda, const void *p0, const void *p1,
                             int *launched);
int sinterp_capture_begin(gsl_sinterp_hip_ctx *ctx, hipStream_t *saved);
int sinterp_capture_end(gsl_sinterp_hip_ctx *ctx, hipStream_t saved, int which, size_t n, size_t lda, const void *p0,
                        const void *p1);

static inline int sinterp_fail(gsl_sinterp_hip_ctx *ctx, int status, const char *what, hipError_t e,
                               const char *file, int line)
{
  if (ctx)
    snprintf(ctx->err, sizeof ctx->err, "%s: %s (%s:%d)", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e),
             file, line);
  return status;
}

#define HIP_OK(ctx, call)                                                               \
  do {                                                                                  \
    hipError_t _e = (call);                                                             \
    if (_e != hipSuccess)                                                               \
      return sinterp_fail(ctx, _e == hipErrorOutOfMemory ? ST_ENOMEM : ST_EFAILED, #call, _e, __FILE__, __LINE__); \
  } while (0)

#define REQUIRE(ctx, cond, status)                                                      \
  do {                                                                                  \
    if (!(cond)) return sinterp_fail(ctx, status, "requirement failed: " #cond, hipSuccess, __FILE__, __LINE__); \
  } while (0)

#define LAUNCH_CHECK(ctx) HIP_OK(ctx, hipGetLastError())

/* hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: set it once per
   (kernel, device) pair (a process-wide flag would leave the second device of a process without it) */
int sinterp_func_lds(gsl_sinterp_hip_ctx *ctx, const void *func, int bytes);

/* Kernels that spin on other workgroups of their own launch (stream-K fix-up, dataflow sweeps, the
   cooperative LU panel) need all their workgroups co-resident; two of them running at once on one
   device (two contexts / streams) could each hold CUs while waiting for workgroups that cannot be
   scheduled.  Every entry point that launches such kernels brackets its launches with these: a
   process-wide per-device event chain orders the exclusive sections of ALL contexts of that device
   one after another on the GPU (nestable; the outermost pair does the work). */
int sinterp_exclusive_begin(gsl_sinterp_hip_ctx *ctx);
int sinterp_exclusive_end(gsl_sinterp_hip_ctx *ctx);

struct SinterpExclusive {      /* RAII form: every return path of an entry point leaves the section */
  gsl_sinterp_hip_ctx *ctx;
  int st;
  explicit SinterpExclusive(gsl_sinterp_hip_ctx *c) : ctx(c), st(sinterp_exclusive_begin(c)) {}
  ~SinterpExclusive() { if (st == ST_SUCCESS) (void)sinterp_exclusive_end(ctx); }
  SinterpExclusive(const SinterpExclusive &) = delete;
  SinterpExclusive &operator=(const SinterpExclusive &) = delete;
};
#define EXCLUSIVE_SECTION(ctx) SinterpExclusive _excl(ctx); if (_excl.st) return _excl.st

/* rbf.hip: the fill with the option of writing the tiles on / below the diagonal only */
int sinterp_rbf_fill_ex(gsl_sinterp_hip_ctx *ctx, int kind, double eps, const double *d_x, size_t n, int dim, size_t xtda,
                        double *d_phi, size_t lda, int lower_only);
int sinterp_tps_fill_shifted(gsl_sinterp_hip_ctx *ctx, const double *d_x, size_t n, int dim, size_t xtda, double *d_phi, size_t lda,
                             const double *d_Pk, int k, double cmul, unsigned long long *d_norm);

/* grow-only workspace owned by the context */
int sinterp_workspace(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out);

/* dense building blocks shared by the Cholesky and LU drivers (gemm.hip) */
/* C[m x n] -= A[m x k] * B^T   with B stored [n][k] (ldb)         -> b_is_kn = 0
   C[m x n] -= A[m x k] * B     with B stored [k][n] (ldb)         -> b_is_kn = 1
   lower_only: C is square on the diagonal, tiles strictly above it are skipped */
int sinterp_gemm_minus(gsl_sinterp_hip_ctx *ctx, size_t m, size_t n, size_t k, const double *A, size_t lda,
                       const double *B, size_t ldb, int b_is_kn, double *C, size_t ldc, int lower_only);

/* allocates the stream-K buffers of the context; must be called OUTSIDE stream capture (the
   factorisation drivers call it before they start capturing).  Without it the GEMM falls back to
   the one-tile-per-workgroup kernels. */
int sinterp_streamk_prepare(gsl_sinterp_hip_ctx *ctx);

/* blocked triangular sweeps (chol.hip): the block being solved is read from b and
   written to xout (b != xout), the remaining right-hand side is updated in b.
   mode 0 Lower/NoTrans fwd, 1 Lower/Trans bwd, 2 Upper/NoTrans bwd */
int sinterp_trsv(gsl_sinterp_hip_ctx *ctx, size_t n, const double *T, size_t ldt, double *b, double *xout, int mode,
                 int unit);
/* the same for nrhs <= 5 right-hand sides stored at b + r*ldb / xout + r*ldb */
int sinterp_trsv_multi(gsl_sinterp_hip_ctx *ctx, size_t n, const double *T, size_t ldt, double *b, double *xout, size_t ldb,
                       int nrhs, int mode, int unit);
int sinterp_cholesky_svx_multi(gsl_sinterp_hip_ctx *ctx, size_t n, const double *d_llt, size_t lda, double *d_x, size_t ldx,
                               int nrhs);
/* gsl_sinterp_hip_cholesky_decomp1 for an input that is stored symmetrically (both triangles valid) */
int sinterp_cholesky_decomp1_sym(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *h_info);
/* the same followed by the solve of nrhs right-hand sides in place (forward substitution folded into the factorisation when every panel is 128 wide) */
int sinterp_cholesky_factor_solve_sym(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *h_info, double *d_x, size_t ldx,
                                      int nrhs);
/* second grow-only buffer for vectors that must outlive factorisation workspaces */
int sinterp_aux(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out);
int sinterp_invbuf(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out);
int sinterp_sortbuf(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out);
/* sort.hip: permutation that groups the targets by cell of a uniform grid (~per_cell each) */
int sinterp_sort_targets(gsl_sinterp_hip_ctx *ctx, const double *d_y, size_t m, size_t ytda, int dim, int per_cell,
                         int **d_perm_out);
/* The same sort, but the targets are physically gathered into cell order (dense [m][dim]) and the
   results come back through sinterp_unsort: the sweep kernels then read and write contiguously, the
   only scattered accesses are one 8*dim-byte write per target here and one 8(+4)-byte READ per target in
   the un-sort (a permutation-indirect kernel pays a scattered read AND partial-sector scattered writes). */
struct sinterp_sorted {
  double *ys;                   /* [m][dim] targets in cell order */
  double *vs;                   /* [m] values in cell order (filled by the sweep); room for [m] {value, leaf} pairs */
  int *ls;                      /* [m] leaf indices in cell order (barycentric sweep) */
  unsigned *cellid, *slot, *offset;
  unsigned long long *box;      /* bounding-box keys, box[2c] = min, box[2c+1] = max */
  bool two_level;               /* large batches (sort.hip, "two-level reorder"): slot[k] = position of target k in the coarse order; */
  unsigned *inv;                /*   inv[p] = coarse position of the target at cell-order position p: the sweep stores the result of */
  double *res1;                 /*   sorted target p at res1[inv[p]] (8 bytes, or a {value, leaf} pair), NOT at vs[p]; */
  unsigned *fin;                /*   (internal: slots of the out-of-window points between the two fine passes) */
  unsigned *tl_cnt; unsigned tl_nb, tl_nwg, tl_ch;   /* (internal: first coarse position of every (bin, workgroup) run, for the staged un-sort) */
};
int sinterp_sort_reorder(gsl_sinterp_hip_ctx *ctx, const double *d_y, size_t m, size_t ytda, int dim, int per_cell,
                         sinterp_sorted *out, size_t m_cap, int slot, const unsigned long long *box_in);
/* whether a batch of m targets takes the two-level route (results through inv / res1) */
bool sinterp_sort_reorder_is_two_level(size_t m);
int sinterp_unsort(gsl_sinterp_hip_ctx *ctx, const sinterp_sorted *s, size_t m, double *d_values, int *d_leaf);
/* vs holds {value, leaf-as-integer-bits} pairs (16 bytes per target; the vs region is sized for it) */
int sinterp_unsort_packed(gsl_sinterp_hip_ctx *ctx, const sinterp_sorted *s, size_t m, double *d_values, int *d_leaf);

/* the centres: cells visited in Morton order (consecutive runs are spatially compact), original
   index order inside a cell (deterministic summation order); uses its own buffer */
int sinterp_sort_centres(gsl_sinterp_hip_ctx *ctx, const double *d_x, size_t n, size_t xtda, int dim, int per_cell,
                         int **d_perm_out);
int sinterp_sortbuf2(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out);
/* exclusive scan of count[0..n) in place, total in count[n]; runsum: n/1024 + 2 entries of scratch (sort.hip) */
void sinterp_scan_u32(gsl_sinterp_hip_ctx *ctx, unsigned *count, size_t n, unsigned *runsum);
int sinterp_walkbuf(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out);
/* sort.hip: bounding box of n points as order-preserving keys, box[2c] = min, box[2c+1] = max (device, 48 bytes) */
int sinterp_bbox_keys(gsl_sinterp_hip_ctx *ctx, const double *d_p, size_t n, size_t tda, int dim, unsigned long long *d_box);
int sinterp_centbuf(gsl_sinterp_hip_ctx *ctx, size_t bytes, void **out);

#endif
