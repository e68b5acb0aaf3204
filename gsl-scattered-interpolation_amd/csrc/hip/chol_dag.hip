/*
 * chol_dag.hip -- the SPD factorisation as ONE persistent launch: a task DAG over 128 x 128 blocks.
 *
 * Same contract as chol.hip (gsl_linalg_cholesky_decomp1, linalg/cholesky.c:88-131: L in the lower triangle, the strict
 * upper triangle untouched, first failing pivot reported).  Replaces the ~500 launches of the recursive driver -- whose
 * per-128-column chain  diag128 -> row solve -> update  runs with the chip idle -- when n is a multiple of 128.
 *
 *   chain workgroup (block 0), for j = 0 .. T-1, everything in LDS / registers:
 *       potrf128 of block (j, j)                           -> L(j,j), inverses of its 32 x 32 diagonal blocks published
 *       X = block (j+1, j) L(j,j)^-T                       (16-row strips per wave, TRANSPOSED in the MFMA accumulators:
 *                                                           X^T = L^-1 Y^T, so a strip's accumulators are the B operands of
 *                                                           the next step and never pass through LDS)
 *       block (j+1, j+1) -= X X^T                          -> stays in LDS for the next potrf128
 *   every other workgroup: tile tasks from a statically ordered list (chol_dag_sched.h), claimed with one atomic each:
 *       UPD   blocks (i0 .. i0+nr-1, j) -= L(rows, k0..k1) L(j, k0..k1)^T     the DMA-ring MFMA pipeline of gemm.hip, the
 *                                                                              accumulators start as -C: ONE pass over C
 *       FUSED block (i, j): that update with everything pending, then X = Y L(j,j)^-T as above
 *
 * Left-looking and lazy: a block takes the columns that became final since its last update in one task (K = 128 right
 * behind the chain, whole chunks of 1024 far from it).  Every task continues the accumulation of the block where the
 * previous one stopped (the accumulators are initialised from the block), always in ascending column order, so an entry
 * of L is one left-looking fused-multiply-add chain whatever the grouping: the result does not depend on the schedule.
 *
 * Hand-offs (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility", form R1): a finished task drains its
 * stores (s_waitcnt vmcnt(0) in every wave, workgroup barrier), one lane issues an agent-scope release fence and then
 * publishes progress counters with relaxed agent-scope stores; a task polls its counters with one lane (relaxed agent
 * loads + s_sleep), issues an agent-scope acquire fence, and only then loads.  Counters are monotone ints:
 *       rowdone[i]   blocks (i, 0 .. rowdone-1) of row i are final L        applied[i T + j]  columns [0, applied) are in block (i, j)
 *       diagdone[j]  L(j,j) and its inverted diagonal blocks are published
 * Deadlock freedom: the list is in the order of a simulated execution in which every task starts after its inputs exist;
 * workers claim in list order and wait only for earlier tasks (or for the chain, which waits only for list tasks that
 * precede it in that order), so the earliest unfinished item always runs.  Every spin is bounded: a wait that exceeds
 * its budget raises the abort word, every workgroup leaves, and the call fails (ST_EFAILED: the matrix is half-factored).
 *
 * STATUS: OPT-IN (GSL_SINTERP_CHOL_DAG=1), NOT the default -- measured round 3 on MI355X, factorisation alone, ms:
 *       n        this file     recursive driver (chol.hip)
 *       4096       2.43            2.06
 *       8192       6.87            6.70
 *      16384      31.2            31.3
 * Correct at every size (tests/test_gpu_chol_dag.py) but not faster, and the simulator that builds the list
 * (chol_dag_sched.h, tools/chol_dag_study) reproduces these times to 3 % once its costs are set to the measured ones
 * (GSL_SINTERP_DAG_PROF=1 prints them), so the reasons are known:
 *   - the chain step is 68 us (potrf128 22, store + release 8, row solve 15 + 6, syrk 15): the row solve and the syrk are
 *     128^3-flop products on ONE CU, 7.7 us each at the MFMA peak of a CU -- the launch-per-panel driver spreads the same
 *     two products over the whole chip (11.6 + 19 us including its launches);
 *   - every block row is its own dependent chain: block (i, j) cannot be solved before (i, j-1) is, and one column of
 *     that chain on one CU costs 13 (hand-off, C tile) + 19 (K = 128 update) + 16..31 (solve) us -- as long as a chain
 *     step.  The cycle  diag(j) -> solve (j+2, j) -> update (j+2, j+1) -> chain  is 84 us against the chain's own 49..68,
 *     so the step is bound at (84 + chain) / 2 = 66 us even with unlimited workers (simulated: 65 us at 2048 workers);
 *   - at n = 16384 the workers are 100 % busy for 25 ms, then the chain (which the list let fall behind: 250 us per
 *     step while the bulk was being served) needs 7 ms for the last ~60 columns at chain speed.
 * Neither the pool sizes, the chunk depth, the priority weights nor 2- / 4-way row splits of the near tasks move the
 * simulated times by more than 3 %.  What would: near-chain tasks spread over many CUs again (which is what the
 * launch-per-panel driver does), i.e. a different algorithm, not a tuning of this one.
 */
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include "chol_potrf.h"
#include <vector>
#include "chol_dag_sched.h"

#define DG_BK 16
#define DG_NT 512                      /* threads per workgroup: 8 waves, 2 per SIMD */

struct DagArgs {
  double *A; size_t lda; int T;
  const DagTask *tasks; unsigned n_tasks;
  const DagTask *express; unsigned n_express_tasks; unsigned n_express_wgs;
  int *rowdone, *applied, *diagdone;
  unsigned *head;                       /* [0] ordinary cursor, [1] express cursor */
  int *info;                            /* first failing pivot (1-based), 0 = none */
  int *abort_flag;
  double *Dinvg;                        /* [T][4][32 x 32] */
  unsigned spin_budget;
  long long *prof;                      /* developer: chain time stamps [T][8] (GSL_SINTERP_DAG_PROF=1), else NULL */
  long long *wprof;                     /* developer: per worker {claim+wait, update, solve+release, tasks}, else NULL */
};

/* ---------------------------------------------------------------------- hand-offs */
__device__ __forceinline__ int ld_cnt(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_cnt(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* one lane: wait until *p >= v; false when the budget ran out or somebody aborted */
__device__ __forceinline__ bool wait_ge(const DagArgs &a, const int *p, int v)
{
  unsigned n = 0;
  while (ld_cnt(p) < v) {
    __builtin_amdgcn_s_sleep(4);
    if ((++n & 255u) == 0) {
      if (ld_cnt(a.abort_flag) != 0) return false;
      if (n > a.spin_budget) { st_cnt(a.abort_flag, 1); return false; }
    }
  }
  return true;
}

/* all threads: `ok` (decided by thread 0) to everybody, behind an acquire of the other workgroups' data */
__device__ __forceinline__ bool acquire_all(bool ok, int *s_flag)
{
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *s_flag = ok ? 1 : 0;
  }
  __syncthreads();
  const bool r = *s_flag != 0;
  __syncthreads();                                       /* s_flag may be rewritten by the next wait */
  return r;
}

/* all threads: this workgroup's global stores are visible device-wide when this returns in thread 0 */
__device__ __forceinline__ void release_all()
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

/* ---------------------------------------------------------------------- UPD: C -= A B^T on the DMA ring */
__device__ __forceinline__ void dg_dma16(const double *gsrc, double *ldst)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
}

/* BM x 128 tile of C (row-major, ldc) -= Aop[BM x K] Bop[128 x K]^T, K = 16 nsteps; the accumulators start as -C, the
   tile is stored once.  diag_mask: only entries with col <= row (within the tile: the block is on the diagonal).
   The K loop is gemm.hip's software-pipelined three-stage ring (see gemm_minus_streamk_kernel, PIPE). */
template <int BM>
__device__ __noinline__ void dg_gemm_tile(double *smem, const double *__restrict__ Aop, const double *__restrict__ Bop, size_t lda,
                                          double *__restrict__ C, size_t ldc, unsigned nsteps, int diag_mask)
{
  constexpr int BN = 128, NW = 8, WN = 64, WM = BM / 4;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_TILE = BM * DG_BK, B_TILE = BN * DG_BK;
  constexpr int A_CH = BM / 8 / NW, B_CH = (BN / 8) / NW;
  constexpr int PER_GROUP = A_CH + B_CH;                 /* 6 (BM = 256) or 4 (BM = 128) */
  double *sA = smem;
  double *sB = smem + 3 * A_TILE;
  double4_t acc[FM][FN];
  {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    /* the accumulators start as -C: in flight together with the first DMA groups */
#pragma unroll
    for (int i = 0; i < FM; i++)
#pragma unroll
      for (int j = 0; j < FN; j++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
          acc[i][j][rg] = -C[(size_t)(wr * WM + i * 16 + fq + 4 * rg) * ldc + wc * WN + j * 16 + fr];
    const int sw = (fr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) koff[kk] = (((kk * 2 + (fq >> 1)) ^ sw) << 1) + (fq & 1);
    const int arow = (wr * WM + fr) * DG_BK, brow = (wc * WN + fr) * DG_BK;
    const double *srcA[A_CH], *srcB[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; i++) {
      const int rr = (wave * A_CH + i) * 8 + (lane >> 3);
      srcA[i] = Aop + (size_t)rr * lda + (((lane & 7) ^ ((rr >> 1) & 7)) << 1);
    }
#pragma unroll
    for (int i = 0; i < B_CH; i++) {
      const int rr = (wave * B_CH + i) * 8 + (lane >> 3);
      srcB[i] = Bop + (size_t)rr * lda + (((lane & 7) ^ ((rr >> 1) & 7)) << 1);
    }
    auto issue = [&](int stage, size_t k0) {
#pragma unroll
      for (int i = 0; i < A_CH; i++) dg_dma16(srcA[i] + k0, sA + stage * A_TILE + (wave * A_CH + i) * 128);
#pragma unroll
      for (int i = 0; i < B_CH; i++) dg_dma16(srcB[i] + k0, sB + stage * B_TILE + (wave * B_CH + i) * 128);
    };
    auto wait_keep = [&](int groups) {
      if (groups >= 2) {
        if constexpr (PER_GROUP == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (groups == 1) {
        if constexpr (PER_GROUP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto load_frag = [&](int stage, int kk, double (&af)[FM], double (&bf)[FN]) {
      const double *a_base = sA + stage * A_TILE + arow;
      const double *b_base = sB + stage * B_TILE + brow;
#pragma unroll
      for (int i = 0; i < FM; i++) af[i] = a_base[i * 16 * DG_BK + koff[kk]];
#pragma unroll
      for (int j = 0; j < FN; j++) bf[j] = b_base[j * 16 * DG_BK + koff[kk]];
    };
    auto mma = [&](const double (&af)[FM], const double (&bf)[FN]) {
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    const unsigned nst = nsteps;
    __syncthreads();                                     /* the ring is free */
    issue(0, 0);
    if (nst > 1) issue(1, DG_BK);
    if (nst > 2) issue(2, 2 * DG_BK);
    wait_keep(nst > 2 ? 2 : (int)nst - 1);               /* also covers the (older) loads of C */
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    double af0[FM], bf0[FN], af1[FM], bf1[FN];
    load_frag(0, 0, af0, bf0);
    for (unsigned rel = 0; rel + 1 < nst; rel++) {
      const int stage = (int)(rel % 3);
      load_frag(stage, 1, af1, bf1);
      mma(af0, bf0);
      load_frag(stage, 2, af0, bf0);
      mma(af1, bf1);
      load_frag(stage, 3, af1, bf1);
      mma(af0, bf0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < FM; i++) asm volatile("" : "+v"(af1[i]));
#pragma unroll
      for (int j = 0; j < FN; j++) asm volatile("" : "+v"(bf1[j]));
      wait_keep(rel + 2 < nst ? 1 : 0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (rel + 3 < nst) issue(stage, (size_t)(rel + 3) * DG_BK);
      load_frag((int)((rel + 1) % 3), 0, af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      mma(af1, bf1);
    }
    {
      const int stage = (int)((nst - 1) % 3);
      load_frag(stage, 1, af1, bf1);
      mma(af0, bf0);
      load_frag(stage, 2, af0, bf0);
      mma(af1, bf1);
      load_frag(stage, 3, af1, bf1);
      mma(af0, bf0);
      mma(af1, bf1);
    }
  }
  /* epilogue: its own laundered copies */
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < FM; i++) {
#pragma unroll
    for (int j = 0; j < FN; j++) {
      const int col = wc * WN + j * 16 + fr;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int row = wr * WM + i * 16 + fq + 4 * rg;
        if (!diag_mask || col <= row) C[(size_t)row * ldc + col] = -acc[i][j][rg];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

/* ---------------------------------------------------------------------- row solve of one 128 x 128 block */
/* X = Y L^-T for the 16 rows of a strip, transposed in the accumulators: Xt[f][r] holds X[row = lane & 15][col = 16 f + (lane >> 4) + 4 r].
   Block substitution over the four 32-column blocks c:  Y_c -= sum_{p < c} X_p L_cp^T, then X_c = Y_c Dinv_c^T; in the
   transposed form the operand from LDS is the A operand (L_cp resp. Dinv_c, rows = output columns) and the strip's own
   accumulators are the B operand: register r of fragment f IS the B fragment of k-group 4 (f & 1) + r of block f >> 1. */
__device__ __forceinline__ void dg_trsm_strip(const double *S, const double *Dv, double4_t (&X)[8], int lane)
{
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int c = 0; c < 4; c++) {
#pragma unroll
    for (int p = 0; p < c; p++) {
      const double *Lb = S + pblk(c, p);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const double *ap = Lb + (16 * h + fr) * PQ + fq;
#pragma unroll
        for (int g = 0; g < 8; g++)
          X[2 * c + h] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ap[4 * g], X[2 * p + (g >> 2)][g & 3], X[2 * c + h], 0, 0, 0);
      }
    }
    const double *Db = Dv + c * PBLK;
    double4_t n0 = (double4_t){0.0, 0.0, 0.0, 0.0}, n1 = n0;
#pragma unroll
    for (int g = 0; g < 4; g++) n0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Db[fr * PQ + fq + 4 * g], X[2 * c][g], n0, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 8; g++)
      n1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Db[(16 + fr) * PQ + fq + 4 * g], X[2 * c + (g >> 2)][g & 3], n1, 0, 0, 0);
    X[2 * c] = n0; X[2 * c + 1] = n1;
  }
}

/* the 128 x 128 block at Ct (row-major, lda): X = Y L^-T in place, 16 rows per wave; S / Dv in LDS.  Xs != NULL: X is
   also left in LDS as [128][130] (must not overlap S / Dv: the caller synchronises before reusing them). */
__device__ __forceinline__ void dg_trsm_block(const double *S, const double *Dv, double *__restrict__ Ct, size_t lda, int tid, double4_t (&X)[8])
{
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  double *rowp = Ct + (size_t)(wave * 16 + fr) * lda + fq;
#pragma unroll
  for (int f = 0; f < 8; f++)
#pragma unroll
    for (int r = 0; r < 4; r++) X[f][r] = rowp[16 * f + 4 * r];
  dg_trsm_strip(S, Dv, X, lane);
#pragma unroll
  for (int f = 0; f < 8; f++)
#pragma unroll
    for (int r = 0; r < 4; r++) rowp[16 * f + 4 * r] = X[f][r];
}

/* L(j,j) (its six off-diagonal 32 x 32 blocks) and the four inverted diagonal blocks from global memory into S / Dv */
__device__ __forceinline__ void dg_load_factor(const DagArgs &a, int j, double *S, double *Dv, int tid)
{
  const double *Ab = a.A + (size_t)j * PB * a.lda + (size_t)j * PB;
  const double *Dg = a.Dinvg + (size_t)j * (4 * 1024);
  for (int e = tid; e < 6 * 1024; e += DG_NT) {
    constexpr int BI[6] = {1, 2, 2, 3, 3, 3}, BJ[6] = {0, 0, 1, 0, 1, 2};
    const int b = e >> 10, r = (e >> 5) & 31, k = e & 31;
    S[pblk(BI[b], BJ[b]) + r * PQ + k] = Ab[(size_t)(BI[b] * 32 + r) * a.lda + BJ[b] * 32 + k];
  }
  for (int e = tid; e < 4 * 1024; e += DG_NT) {
    const int b = e >> 10, r = (e >> 5) & 31, k = e & 31;
    Dv[b * PBLK + r * PQ + k] = Dg[e];
  }
}

/* ---------------------------------------------------------------------- workers */
__device__ __noinline__ void dg_task_upd(const DagArgs &a, const DagTask &t, double *smem)
{
  const size_t lda = a.lda;
  const double *Aop = a.A + (size_t)t.i0 * PB * lda + (size_t)t.k0 * PB;
  const double *Bop = a.A + (size_t)t.j * PB * lda + (size_t)t.k0 * PB;
  double *C = a.A + (size_t)t.i0 * PB * lda + (size_t)t.j * PB;
  const unsigned nsteps = (unsigned)(t.k1 - t.k0) * (PB / DG_BK);
  if (t.nr == 2) dg_gemm_tile<256>(smem, Aop, Bop, lda, C, lda, nsteps, 0);
  else dg_gemm_tile<128>(smem, Aop, Bop, lda, C, lda, nsteps, t.i0 == t.j);
}

__device__ __noinline__ void dg_task_solve(const DagArgs &a, int i, int j, double *smem)
{
  double *S = smem, *Dv = smem + 10 * PBLK;
  const int tid = threadIdx.x;
  __syncthreads();                                       /* LDS free (the ring of a preceding update) */
  dg_load_factor(a, j, S, Dv, tid);
  __syncthreads();
  double4_t X[8];
  dg_trsm_block(S, Dv, a.A + (size_t)i * PB * a.lda + (size_t)j * PB, a.lda, tid, X);
}

/* wait for the inputs of a list task (thread 0 polls) */
__device__ __forceinline__ bool dg_wait_task(const DagArgs &a, const DagTask &t, int *s_flag)
{
  bool ok = true;
  if (threadIdx.x == 0) {
    for (int r = 0; r < (int)t.nr && ok; r++) {
      const int i = t.i0 + r;
      ok = wait_ge(a, a.applied + (size_t)i * a.T + t.j, t.k0);
      if (ok) ok = wait_ge(a, a.rowdone + i, t.k1 < i ? t.k1 : i);
    }
    if (ok) ok = wait_ge(a, a.rowdone + t.j, t.k1);
    if (ok && t.type == DAG_FUSED) ok = wait_ge(a, a.diagdone + t.j, 1);
  }
  return acquire_all(ok, s_flag);
}

__device__ __forceinline__ void dg_worker(const DagArgs &a, double *smem, int *s_flag, unsigned *s_task)
{
  const bool express = blockIdx.x <= a.n_express_wgs;    /* blocks 1 .. n_express_wgs (block 0 is the chain) */
  /* developer profile: [0] claim+wait, [1..3] 256-row updates {ticks, K steps, tasks}, [4..6] 128-row updates, [7..8] solve+release {ticks, tasks},
     [9] release of plain updates, [10] first stamp, [11] last stamp */
  long long w_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, w_t0 = a.wprof ? (long long)wall_clock64() : 0;
  w_acc[10] = w_t0;
#define DG_WSTAMP(acc) do { if (a.wprof && threadIdx.x == 0) { const long long now_ = (long long)wall_clock64(); acc += now_ - w_t0; w_t0 = now_; } } while (0)
#define DG_WDONE() do { if (a.wprof && threadIdx.x == 0) { long long *q_ = a.wprof + (size_t)blockIdx.x * 12; w_acc[11] = (long long)wall_clock64(); for (int z_ = 0; z_ < 12; z_++) q_[z_] = w_acc[z_]; } } while (0)
  for (;;) {
    /* claim: express workers from the express list; the others from the main list, then (when it is used up) from
       the express list -- the rule dag_check_schedule replays */
    if (threadIdx.x == 0) {
      unsigned idx = 0xffffffffu, which = 0;
      if (!express) {
        const unsigned k = atomicAdd(&a.head[0], 1u);
        if (k < a.n_tasks) { idx = k; which = 0; }
      }
      if (idx == 0xffffffffu) {
        const unsigned k = atomicAdd(&a.head[1], 1u);
        if (k < a.n_express_tasks) { idx = k; which = 1; }
      }
      s_task[0] = idx; s_task[1] = which;
    }
    __syncthreads();
    const unsigned idx = s_task[0], which = s_task[1];
    __syncthreads();
    if (idx == 0xffffffffu) { DG_WDONE(); return; }
    const DagTask t = (which ? a.express : a.tasks)[idx];
    if (!dg_wait_task(a, t, s_flag)) return;
    DG_WSTAMP(w_acc[0]);
    if (t.k1 > t.k0) dg_task_upd(a, t, smem);
    if (a.wprof && threadIdx.x == 0 && t.k1 > t.k0) { const int o_ = t.nr == 2 ? 1 : 4; w_acc[o_ + 1] += (t.k1 - t.k0) * 8; w_acc[o_ + 2]++; }
    DG_WSTAMP(w_acc[t.nr == 2 ? 1 : 4]);
    if (t.type == DAG_FUSED) {
      for (int r = 0; r < (int)t.nr; r++) {
        /* the update's stores of this block are re-read below by the same workgroup: drained, workgroup-scope fence,
           barrier (the first thing dg_task_solve does) */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        dg_task_solve(a, t.i0 + r, t.j, smem);
      }
    }
    release_all();
    if (threadIdx.x == 0) {
      for (int r = 0; r < (int)t.nr; r++) {
        st_cnt(a.applied + (size_t)(t.i0 + r) * a.T + t.j, (int)t.k1);
        if (t.type == DAG_FUSED) st_cnt(a.rowdone + t.i0 + r, (int)t.j + 1);
      }
    }
    if (t.type == DAG_FUSED) w_acc[8]++;
    DG_WSTAMP(w_acc[t.type == DAG_FUSED ? 7 : 9]);
  }
}

/* ---------------------------------------------------------------------- chain */
__device__ __forceinline__ void dg_chain(const DagArgs &a, double *smem, int *s_flag)
{
  double *S = smem, *Dv = smem + 10 * PBLK;
  double *Xs = smem;                                     /* [128][130], reuses S / Dv after the row solve */
  constexpr int XP = 130;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const size_t lda = a.lda;
  const int T = a.T;
  /* block (0, 0) into S (whole 32 x 32 blocks of the lower triangle) */
  for (int e = tid; e < 10 * 1024; e += DG_NT) {
    constexpr int BI[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3}, BJ[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
    const int b = e >> 10, r = (e >> 5) & 31, k = e & 31;
    S[b * PBLK + r * PQ + k] = a.A[(size_t)(BI[b] * 32 + r) * lda + BJ[b] * 32 + k];
  }
  __syncthreads();
  for (int j = 0; j < T; j++) {
    double *Ab = a.A + (size_t)j * PB * lda + (size_t)j * PB;
#define DG_STAMP(k) do { if (a.prof && tid == 0) a.prof[(size_t)j * 8 + (k)] = (long long)wall_clock64(); } while (0)
    DG_STAMP(0);
    potrf128_lds<8>(S, Dv, tid, a.info, (size_t)j * PB);
    DG_STAMP(1);
    /* L(j,j) -> A (lower part), inverted diagonal blocks -> Dinvg[j]: compile-time block indices, the LDS reads of a
       thread issued together (a run-time indexed loop here cost 17 us per column) */
    {
      const int r16 = tid >> 5, k = tid & 31;
      double v[20], dv[8];
#pragma unroll
      for (int t = 0; t < 20; t++) v[t] = S[(t >> 1) * PBLK + ((t & 1) * 16 + r16) * PQ + k];
#pragma unroll
      for (int t = 0; t < 8; t++) dv[t] = Dv[(t >> 1) * PBLK + ((t & 1) * 16 + r16) * PQ + k];
#pragma unroll
      for (int t = 0; t < 20; t++) {
        constexpr int BI[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3}, BJ[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
        const int b = t >> 1, r = (t & 1) * 16 + r16, bi = BI[b], bj = BJ[b];
        if (bi != bj || k <= r) Ab[(size_t)(bi * 32 + r) * lda + bj * 32 + k] = v[t];
      }
      double *Dg = a.Dinvg + (size_t)j * (4 * 1024);
#pragma unroll
      for (int t = 0; t < 8; t++) Dg[(t >> 1) * 1024 + ((t & 1) * 16 + r16) * 32 + k] = dv[t];
    }
    release_all();
    if (tid == 0) { st_cnt(a.diagdone + j, 1); st_cnt(a.rowdone + j, j + 1); }
    DG_STAMP(2);
    if (j + 1 == T) break;
    /* block (j+1, j): the list has applied the columns < j */
    bool ok = true;
    if (tid == 0) ok = wait_ge(a, a.applied + (size_t)(j + 1) * T + j, j) && wait_ge(a, a.applied + (size_t)(j + 1) * T + j + 1, j);
    if (!acquire_all(ok, s_flag)) return;
    DG_STAMP(3);
    double4_t X[8];
    double *Ct = a.A + (size_t)(j + 1) * PB * lda + (size_t)j * PB;
    dg_trsm_block(S, Dv, Ct, lda, tid, X);
    DG_STAMP(4);
    __syncthreads();                                     /* everybody is done with S / Dv */
#pragma unroll
    for (int f = 0; f < 8; f++)
#pragma unroll
      for (int r = 0; r < 4; r++) Xs[(wave * 16 + fr) * XP + 16 * f + fq + 4 * r] = X[f][r];
    /* block (j+1, j) is final */
    release_all();
    if (tid == 0) st_cnt(a.rowdone + j + 1, j + 1);
    DG_STAMP(5);
    /* (the barrier inside release_all also published Xs to the workgroup) */
    /* block (j+1, j+1) -= X X^T: the 36 lower 16 x 16 fragments, up to five per wave */
    double *Cd = a.A + (size_t)(j + 1) * PB * lda + (size_t)(j + 1) * PB;
    double4_t acc[5];
    int fa[5], fb[5];
#pragma unroll
    for (int s = 0; s < 5; s++) {
      const int e = wave + 8 * s;                        /* fragment id in row-major lower order */
      int aa = 0;
      while ((aa + 1) * (aa + 2) / 2 <= e) aa++;
      fa[s] = aa; fb[s] = e - aa * (aa + 1) / 2;
      if (e < 36) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) acc[s][rg] = -Cd[(size_t)(16 * fa[s] + fq + 4 * rg) * lda + 16 * fb[s] + fr];
      } else acc[s] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int s = 0; s < 5; s++) {
      if (wave + 8 * s < 36) {
        const double *ap = Xs + (16 * fa[s] + fr) * XP + fq, *bp = Xs + (16 * fb[s] + fr) * XP + fq;
#pragma unroll 8
        for (int g = 0; g < 32; g++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * g], bp[4 * g], acc[s], 0, 0, 0);
      }
    }
    DG_STAMP(6);
    __syncthreads();                                     /* Xs read by everybody: S may be rebuilt */
#pragma unroll
    for (int s = 0; s < 5; s++) {
      if (wave + 8 * s < 36) {
        double *Sb = S + pblk(fa[s] >> 1, fb[s] >> 1);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) Sb[(16 * (fa[s] & 1) + fq + 4 * rg) * PQ + 16 * (fb[s] & 1) + fr] = -acc[s][rg];
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(DG_NT, 2)
chol_dag_kernel(DagArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int s_flag;
  __shared__ unsigned s_task[2];
  if (blockIdx.x == 0) dg_chain(a, smem, &s_flag);
  else dg_worker(a, smem, &s_flag, s_task);
}

__global__ void chol_dag_reset_kernel(int *counters, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) counters[i] = 0;
}

/* ---------------------------------------------------------------------- host */
struct DagPlan {                                         /* one per (context, T): the lists live on the device */
  int T, workers, n_express;
  DagTask *d_tasks, *d_express;
  unsigned n_tasks, n_express_tasks;
  int *d_counters;                                       /* rowdone[T] | diagdone[T] | applied[T*T] | head[2] | info | abort */
  double *d_dinv;
  size_t n_counters;
};

static DagPlan g_plans[64][4];                           /* per device: a few sizes (round robin) */
static int g_plan_next[64];

static int dag_plan_get(gsl_sinterp_hip_ctx *ctx, int T, DagPlan **out)
{
  const int dv = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
  for (int i = 0; i < 4; i++) if (g_plans[dv][i].T == T) { *out = &g_plans[dv][i]; return ST_SUCCESS; }
  int cus = 0;
  HIP_OK(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
  if (cus < 8) return sinterp_fail(ctx, ST_EFAILED, "chol_dag: too few CUs", hipSuccess, __FILE__, __LINE__);
  DagPlan &p = g_plans[dv][g_plan_next[dv]];
  g_plan_next[dv] = (g_plan_next[dv] + 1) % 4;
  if (p.T) {
    HIP_OK(ctx, hipDeviceSynchronize());
    (void)hipFree(p.d_tasks); (void)hipFree(p.d_express); (void)hipFree(p.d_counters); (void)hipFree(p.d_dinv);
    memset(&p, 0, sizeof p);
  }
  DagSchedule s;
  DagCost cm = dag_default_cost(T);
  const int workers = cus - 1;
  if (cm.express > workers / 4) cm.express = workers / 4;
  dag_build_schedule(T, workers, cm, &s);
  if (dag_check_schedule(T, workers, s.n_express, s.tasks, s.express) != 0)
    return sinterp_fail(ctx, ST_EFAILED, "chol_dag: the task list failed its replay check", hipSuccess, __FILE__, __LINE__);
  p.workers = workers; p.n_express = s.n_express;
  p.n_tasks = (unsigned)s.tasks.size(); p.n_express_tasks = (unsigned)s.express.size();
  HIP_OK(ctx, hipMalloc((void **)&p.d_tasks, (s.tasks.size() + 1) * sizeof(DagTask)));
  HIP_OK(ctx, hipMalloc((void **)&p.d_express, (s.express.size() + 1) * sizeof(DagTask)));
  if (!s.tasks.empty()) HIP_OK(ctx, hipMemcpy(p.d_tasks, s.tasks.data(), s.tasks.size() * sizeof(DagTask), hipMemcpyHostToDevice));
  if (!s.express.empty()) HIP_OK(ctx, hipMemcpy(p.d_express, s.express.data(), s.express.size() * sizeof(DagTask), hipMemcpyHostToDevice));
  p.n_counters = (size_t)2 * T + (size_t)T * T + 8;
  HIP_OK(ctx, hipMalloc((void **)&p.d_counters, p.n_counters * sizeof(int)));
  HIP_OK(ctx, hipMalloc((void **)&p.d_dinv, (size_t)T * 4 * 1024 * sizeof(double)));
  p.T = T;
  *out = &p;
  return ST_SUCCESS;
}

/* returns ST_SUCCESS with *h_done = 1 when the factorisation ran (and *h_info holds the failing pivot or 0);
   *h_done = 0: not applicable / aborted, the caller uses the launch-per-panel driver (the matrix is untouched only in
   the not-applicable case: an aborted run has modified it, the caller must restore it -- it keeps no copy, so an abort
   is reported as an error instead) */
bool sinterp_cholesky_dag_applicable(size_t n, const double *d_a, size_t lda)
{
  const char *e = getenv("GSL_SINTERP_CHOL_DAG");         /* opt-in (see the header comment); read per call */
  const bool on = e && e[0] == '1';
  return on && n % PB == 0 && n >= 2 * PB && n / PB <= 1024 && (lda & 1) == 0 && ((((uintptr_t)d_a) & 15) == 0);
}

int sinterp_cholesky_dag(gsl_sinterp_hip_ctx *ctx, size_t n, double *d_a, size_t lda, int *h_info, int *h_done)
{
  *h_done = 0;
  if (!sinterp_cholesky_dag_applicable(n, d_a, lda)) return ST_SUCCESS;
  const int T = (int)(n / PB);
  DagPlan *p = NULL;
  int st = dag_plan_get(ctx, T, &p);
  if (st) return st;
  DagArgs a;
  a.A = d_a; a.lda = lda; a.T = T;
  a.tasks = p->d_tasks; a.n_tasks = p->n_tasks; a.express = p->d_express; a.n_express_tasks = p->n_express_tasks;
  a.n_express_wgs = (unsigned)p->n_express;
  a.rowdone = p->d_counters; a.diagdone = a.rowdone + T; a.applied = a.diagdone + T;
  a.head = (unsigned *)(a.applied + (size_t)T * T); a.info = (int *)(a.head + 2); a.abort_flag = a.info + 1;
  a.Dinvg = p->d_dinv;
  static const bool prof_on = getenv("GSL_SINTERP_DAG_PROF") && getenv("GSL_SINTERP_DAG_PROF")[0] == '1';
  a.prof = NULL;
  a.wprof = NULL;
  if (prof_on) {
    HIP_OK(ctx, hipMalloc((void **)&a.prof, (size_t)T * 8 * sizeof(long long))); HIP_OK(ctx, hipMemset(a.prof, 0, (size_t)T * 8 * sizeof(long long)));
    HIP_OK(ctx, hipMalloc((void **)&a.wprof, (size_t)(p->workers + 1) * 12 * sizeof(long long)));
    HIP_OK(ctx, hipMemset(a.wprof, 0, (size_t)(p->workers + 1) * 12 * sizeof(long long)));
  }
  a.spin_budget = 1u << 22;                               /* ~4 M polls of >= 64 cycles: seconds, then abort */
  const size_t lds = (size_t)3 * (256 + 128) * DG_BK * sizeof(double);     /* 144 KiB: the ring of the 256-row update */
  { int ast = sinterp_func_lds(ctx, (const void *)chol_dag_kernel, (int)lds); if (ast) return ast; }
  hipLaunchKernelGGL(chol_dag_reset_kernel, dim3((unsigned)((p->n_counters + 255) / 256)), dim3(256), 0, ctx->stream, p->d_counters, p->n_counters);
  hipLaunchKernelGGL(chol_dag_kernel, dim3((unsigned)(p->workers + 1)), dim3(DG_NT), lds, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  int res[2] = {0, 0};
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_OK(ctx, hipMemcpy(res, a.info, sizeof res, hipMemcpyDeviceToHost));
  if (a.prof) {
    std::vector<long long> h((size_t)T * 8);
    (void)hipMemcpy(h.data(), a.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    (void)hipFree(a.prof);
    double sum[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j + 1 < T; j++) {
      for (int k = 0; k < 6; k++) sum[k] += (double)(h[(size_t)j * 8 + k + 1] - h[(size_t)j * 8 + k]);
      sum[6] += (double)(h[(size_t)(j + 1) * 8] - h[(size_t)j * 8 + 6]);
    }
    const double u = 0.01 / (T - 1);                      /* 100 MHz ticks -> us per step */
    fprintf(stderr, "chol_dag T=%d chain us/step: potrf %.1f  store+release %.1f  wait %.1f  trsm %.1f  Xs+release %.1f  syrk %.1f  rebuild %.1f  | total %.1f\n",
            T, sum[0] * u, sum[1] * u, sum[2] * u, sum[3] * u, sum[4] * u, sum[5] * u, sum[6] * u, (double)(h[(size_t)(T - 1) * 8] - h[0]) * u);
    {
      std::vector<long long> w((size_t)(p->workers + 1) * 12);
      (void)hipMemcpy(w.data(), a.wprof, w.size() * sizeof(long long), hipMemcpyDeviceToHost);
      (void)hipFree(a.wprof);
      double t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, span = 0, et[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int b = 1; b <= p->workers; b++) {
        for (int z = 0; z < 10; z++) (b <= p->n_express ? et : t)[z] += (double)w[(size_t)b * 12 + z];
        span += (double)(w[(size_t)b * 12 + 11] - w[(size_t)b * 12 + 10]);
      }
      for (int pool = 0; pool < 2; pool++) {
        const double *q = pool ? et : t;
        const int nw = pool ? p->n_express : p->workers - p->n_express;
        fprintf(stderr, "chol_dag T=%d %s workers (%d): wait %.0f us each | 256-row updates: %.0f tasks, %.2f us/Kstep + fixed? mean %.1f us/task (%.1f steps) | 128-row: %.0f tasks, mean %.1f us/task (%.1f steps) | solves %.0f, mean %.1f us | plain release %.1f us/task\n",
                T, pool ? "express" : "list", nw, q[0] * 0.01 / (nw ? nw : 1), q[3], q[2] ? q[1] * 0.01 / q[2] : 0.0, q[3] ? q[1] * 0.01 / q[3] : 0.0, q[3] ? q[2] / q[3] : 0.0,
                q[6], q[6] ? q[4] * 0.01 / q[6] : 0.0, q[6] ? q[5] / q[6] : 0.0, q[8], q[8] ? q[7] * 0.01 / q[8] : 0.0, (q[3] + q[6] - q[8]) > 0 ? q[9] * 0.01 / (q[3] + q[6] - q[8]) : 0.0);
      }
      fprintf(stderr, "chol_dag T=%d kernel span %.0f us, workers alive %.0f us (mean); lists %u + %u\n", T, (double)(h[(size_t)(T - 1) * 8 + 2] - h[0]) * 0.01,
              span * 0.01 / p->workers, p->n_tasks, p->n_express_tasks);
    }
    if (getenv("GSL_SINTERP_DAG_PROF_STEPS"))
    {
      fprintf(stderr, "  chain wait per step (us):");
      for (int j = 0; j + 1 < T; j++) fprintf(stderr, " %.0f", (double)(h[(size_t)j * 8 + 3] - h[(size_t)j * 8 + 2]) * 0.01);
      fprintf(stderr, "\n");
    }
  }
  if (res[1]) return sinterp_fail(ctx, ST_EFAILED, "chol_dag: a dependency wait ran out of its budget (aborted)", hipSuccess, __FILE__, __LINE__);
  *h_info = res[0];
  *h_done = 1;
  return ST_SUCCESS;
}
