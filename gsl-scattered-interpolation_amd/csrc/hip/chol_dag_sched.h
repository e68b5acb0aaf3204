/*
 * chol_dag_sched.h -- host-side list scheduler of the task-DAG Cholesky (chol_dag.hip).
 *
 * The factorisation of an N x N matrix (N = 128 T) is cut into tile tasks on 128 x 128 blocks (i, j), j <= i < T:
 *
 *   chain (one dedicated workgroup, not in the list), for j = 0 .. T-1:
 *       potrf of block (j, j)  ->  publish L(j,j) and the inverses of its 32 x 32 diagonal blocks
 *       X = block (j+1, j) L(j,j)^-T                (needs the list's updates of that block)
 *       block (j+1, j+1) -= X X^T                   (needs the list's updates with the columns < j)
 *   list tasks (every other workgroup takes them in list order from one atomic cursor):
 *       UPD  (i0, nr, j, k0, k1):  blocks (i0 .. i0+nr-1, j) -= L(rows, k0..k1-1) L(j, k0..k1-1)^T     (nr = 1 | 2)
 *       FUSED(i, j, k0):           block (i, j): the update with the columns k0 .. j-1, then X = Y L(j,j)^-T
 *
 * Left-looking and LAZY: a block accumulates the columns that became available since its last update and takes
 * them in one task -- K = 128 right behind the chain, K = 1024 (KCB blocks) far away from it, where the updates
 * wait for whole chunks.  Every task starts its accumulators from the block itself and adds the columns in
 * ascending order, so the bits of the factor do not depend on how the columns were grouped into tasks
 * (a pure left-looking sum per entry), hence not on this schedule either.
 *
 * The list is produced by SIMULATING the execution on W workers with a cost model (classic list scheduling:
 * whenever a worker is free it takes the ready task of highest priority) and emitting the tasks in the order of
 * their simulated start times.  On the device a worker that gets a task before its inputs exist spins on the
 * progress counters; because every dependency points backwards in the simulated order and the cursor hands the
 * tasks out in that order, the earliest unfinished task always has its inputs: no deadlock, whatever the real
 * durations are (tests/test_dag_schedule.py checks the order property on the CPU for many sizes).
 */
#ifndef SINTERP_CHOL_DAG_SCHED_H
#define SINTERP_CHOL_DAG_SCHED_H

#include <algorithm>
#include <queue>
#include <stdint.h>
#include <vector>

enum { DAG_UPD = 0, DAG_FUSED = 1 };

struct DagTask {            /* 16 bytes, read by the device */
  uint16_t type, nr, i0, j, k0, k1;
  uint16_t pad0, pad1;
};

struct DagCost {            /* microseconds; calibrated on MI355X (tools/dag_calibrate.py) */
  double step256, step128;  /* one 16-wide K-step of a 256 x 128 / 128 x 128 update */
  double upd_fixed256, upd_fixed128;   /* poll + acquire + C tile load + ring fill + store + release */
  double fused_step, fused_fixed, fused_trsm;
  double potrf, chain_trsm, chain_syrk, chain_pub;
  int kcb;                  /* far updates wait for chunks of kcb column blocks */
  int near_rows;            /* rows this close to the chain take every new column at once */
  int express, urgent_rows; /* workgroups reserved for the tasks of the rows within urgent_rows of the chain */
};

static inline DagCost dag_default_cost(int T)
{
  DagCost c;
  c.step256 = 4.0; c.step128 = 2.4; c.upd_fixed256 = 9.0; c.upd_fixed128 = 9.0;
  c.fused_step = 2.6; c.fused_fixed = 11.0; c.fused_trsm = 9.0;
  c.potrf = 26.0; c.chain_trsm = 9.0; c.chain_syrk = 10.0; c.chain_pub = 2.0;
  c.kcb = T >= 96 ? 8 : (T >= 48 ? 4 : 2);
  c.near_rows = 4;
  c.express = 16; c.urgent_rows = 8;
  return c;
}

struct DagSchedule {
  std::vector<DagTask> tasks;     /* taken by the ordinary workers */
  std::vector<DagTask> express;   /* taken by the express workers */
  double makespan_us;       /* simulated */
  double busy_frac;         /* simulated worker utilisation */
  int n_express;            /* express workers the lists were made for */
  double chain_done_us;     /* simulated time at which the chain published its last block */
  std::vector<double> diag_us;   /* simulated publication time of every diagonal block */
};

/* workers: workgroups that take list tasks (the chain workgroup is extra) */
static inline void dag_build_schedule(int T, int workers, const DagCost &cm, DagSchedule *out)
{
  const int W = workers < 1 ? 1 : workers;
  const int KCB = cm.kcb;
  const int NEAR = cm.near_rows;
  std::vector<int> rd(T, 0);                       /* final tiles of row i (frontier) */
  std::vector<int> applied((size_t)T * T, 0);      /* completed updates: columns [0, applied) are in the block */
  std::vector<char> busy((size_t)T * T, 0);
  std::vector<char> diag(T, 0);
  std::vector<int> far_c(T, 0), far_j(T, 0);       /* per-row-pair cursor of the far (chunk) updates */
  out->tasks.clear(); out->express.clear(); out->diag_us.clear(); out->chain_done_us = 0;
  auto AP = [&](int i, int j) -> int & { return applied[(size_t)i * T + j]; };
  auto BUSY = [&](int i, int j) -> char & { return busy[(size_t)i * T + j]; };

  struct Ev { double t; int kind; int a, b, c, d; };   /* kind 0: list task done (a = type, b = i0|nr<<16, c = j, d = k1); 1: chain */
  auto cmp = [](const Ev &x, const Ev &y) { return x.t > y.t; };
  std::priority_queue<Ev, std::vector<Ev>, decltype(cmp)> evq(cmp);

  /* chain state machine: phase 0 potrf running (ends -> diag published), 1 waiting for block (j+1, j), 2 trsm running,
     3 waiting for block (j+1, j+1), 4 syrk running */
  int cj = 0, cphase = 0;
  double now = 0.0, busy_time = 0.0;
  const int NE = std::min(cm.express, W > 1 ? W - 1 : 0), URG = cm.urgent_rows;
  int free_workers = W - NE, free_express = NE;
  bool chain_done = false;
  evq.push(Ev{cm.potrf + cm.chain_pub, 1, 0, 0, 0, 0});

  auto chain_poll = [&]() {                        /* start the next chain phase when its input is there */
    if (chain_done) return;
    if (cphase == 1 && AP(cj + 1, cj) >= cj && !BUSY(cj + 1, cj)) { cphase = 2; evq.push(Ev{now + cm.chain_trsm, 1, 0, 0, 0, 0}); }
    else if (cphase == 3 && AP(cj + 1, cj + 1) >= cj && !BUSY(cj + 1, cj + 1)) { cphase = 4; evq.push(Ev{now + cm.chain_syrk, 1, 0, 0, 0, 0}); }
  };

  /* candidate of row i at the current state; returns false when the row has nothing ready */
  /* priority = slack of the row in chain steps: the chain reaches row i in (i - front) steps, the row needs about
     0.75 steps for each of its (i - f) remaining columns; a smaller score is served first */
  auto slack = [&](int i, int f) -> double { return (double)(i - cj) - 0.75 * (double)(i - f); };
  struct Cand { int type, i0, nr, j, k0, k1; double score; int pair; };   /* pair >= 0: from the far cursor of that row pair */
  auto row_candidate = [&](int i, Cand *c) -> bool {
    const int f = rd[i];
    const int front = cj;
    bool have = false;
    /* (A) the frontier block (i, f), f < i: its last update (+ the solve when the chain does not own it) */
    if (f < i && !BUSY(i, f) && rd[f] >= f) {
      const int a = AP(i, f);
      if (i == f + 1) {                            /* chain solves it; the list brings it up to date */
        if (a < f) { *c = Cand{DAG_UPD, i, 1, f, a, f, slack(i, f), -1}; have = true; }
      } else if (diag[f]) {
        *c = Cand{DAG_FUSED, i, 1, f, a, f, slack(i, f), -1}; have = true;
      }
    }
    if (have) return true;
    /* (D) rows right behind the chain (within NEAR block rows of it): every block of the row takes a new column as
       soon as it exists (K = 128), so that the frontier update on the chain's critical cycle is never more than one
       column deep.  At most NEAR^2 / 2 small tasks per chain step. */
    if (i - front <= NEAR) {
      for (int j = f + 1; j <= i; j++) {
        if (BUSY(i, j)) continue;
        const int tgt = j == i ? i - 1 : j;
        const int avail = std::min(std::min(f, rd[j]), tgt), a = AP(i, j);
        if (avail > a) { *c = Cand{DAG_UPD, i, 1, j, a, avail, slack(i, f) + 0.25 + 0.01 * (j - f), -1}; return true; }
      }
    }
    /* (B) the diagonal block (i, i): the list applies the columns < i-1, the chain adds column i-1 */
    if (!BUSY(i, i) && i >= 2) {
      const int a = AP(i, i), avail = std::min(f, i - 1);
      if (avail > a && (avail == i - 1 || avail - a >= KCB)) {
        const int k1 = avail == i - 1 ? avail : a + ((avail - a) / KCB) * KCB;
        *c = Cand{DAG_UPD, i, 1, i, a, k1, slack(i, f) + (avail == i - 1 ? 0.1 : 3.0), -1};
        return true;
      }
    }
    return false;
  };

  /* (C) far blocks: whole chunks of KCB columns, chunk by chunk, column by column, per PAIR of rows (2m, 2m+1) --
     the 256 x 128 tile is the efficient shape of the update.  Columns j <= 2m-1 as pairs, then block (2m+1, 2m)
     alone; a block that a frontier update has already carried past the chunk is skipped. */
  auto pair_candidate = [&](int m, Cand *c) -> bool {
    const int r0 = 2 * m, r1 = std::min(2 * m + 1, T - 1);
    for (int guard = 0; guard < 4 * T; guard++) {
      const int c0 = far_c[m];
      int j = far_j[m];
      const int kend = (c0 + 1) * KCB, k0 = c0 * KCB;
      if (kend > std::min(rd[r0], rd[r1])) return false;            /* the rows' L is not final through the chunk yet */
      if (j < kend) j = kend;
      if (j > r1 - 1 || (r1 == r0 && j > r0 - 1)) { far_c[m] = c0 + 1; far_j[m] = 0; continue; }
      far_j[m] = j;
      const bool two = j <= r0 - 1 && r1 != r0;
      const int lo = two ? r0 : r1;                                  /* j == r0: only block (r1, r0) */
      const bool need0 = two && j > rd[r0] && AP(r0, j) <= k0, need1 = j > rd[r1] && AP(r1, j) <= k0;
      if (!need0 && !need1) { far_j[m] = j + 1; continue; }
      if (rd[j] < kend) return false;
      if ((need0 && (BUSY(r0, j) || AP(r0, j) != k0)) || (need1 && (BUSY(r1, j) || AP(r1, j) != k0))) return false;
      const double sc = slack(lo, std::min(rd[r0], rd[r1])) + 3.0 + 0.02 * (j - rd[r0]);
      if (need0 && need1) *c = Cand{DAG_UPD, r0, 2, j, k0, kend, sc, m};
      else *c = Cand{DAG_UPD, need0 ? r0 : r1, 1, j, k0, kend, sc, m};
      return true;
    }
    return false;
  };

  auto duration = [&](const Cand &c) -> double {
    const int kb = c.k1 - c.k0;
    if (c.type == DAG_FUSED) return cm.fused_fixed + kb * 8 * cm.fused_step + cm.fused_trsm;
    return c.nr == 2 ? cm.upd_fixed256 + kb * 8 * cm.step256 : cm.upd_fixed128 + kb * 8 * cm.step128;
  };

  /* Two worker pools, two lists.  Express workers only take the tasks of the rows right behind the chain
     (i - front <= URG): everything on the chain's critical cycle.  Without them such a task, once ready, waits for
     the next worker that finishes a far update -- and far updates start and end in waves of hundreds (a chunk
     opens for every row at once), so the wait is a large fraction of their 0.27 ms, per hop, four hops per column. */
  auto assign_pool = [&](int pool) {
    int &freew = pool == 0 ? free_express : free_workers;
    while (freew > 0) {
      Cand best; bool found = false;
      for (int i = 1; i < T; i++) {
        const bool urgent = NE > 0 && i - cj <= URG;
        if (urgent != (pool == 0)) continue;
        Cand c;
        if (!row_candidate(i, &c)) continue;
        if (!found || c.score < best.score) { best = c; found = true; }
      }
      if (pool == 1)
        for (int m = 0; 2 * m < T; m++) {
          Cand c;
          if (!pair_candidate(m, &c)) continue;
          if (!found || c.score < best.score) { best = c; found = true; }
        }
      if (!found) break;
      DagTask t;
      t.type = (uint16_t)best.type; t.nr = (uint16_t)best.nr; t.i0 = (uint16_t)best.i0; t.j = (uint16_t)best.j;
      t.k0 = (uint16_t)best.k0; t.k1 = (uint16_t)best.k1; t.pad0 = t.pad1 = 0;
      (pool == 0 ? out->express : out->tasks).push_back(t);
      for (int r = 0; r < best.nr; r++) BUSY(best.i0 + r, best.j) = 1;
      if (best.pair >= 0) far_j[best.pair] = best.j + 1;   /* the cursor moves on: later columns do not wait for this task */
      const double d = duration(best);
      busy_time += d;
      evq.push(Ev{now + d, 0, best.type | (pool << 8), best.i0 | (best.nr << 16), best.j, best.k1});
      freew--;
    }
  };
  auto assign = [&]() { assign_pool(0); assign_pool(1); };

  assign();
  while (!evq.empty()) {
    const Ev e = evq.top(); evq.pop();
    now = e.t;
    if (e.kind == 0) {
      const int i0 = e.b & 0xffff, nr = e.b >> 16, j = e.c, k1 = e.d;
      for (int r = 0; r < nr; r++) {
        BUSY(i0 + r, j) = 0;
        AP(i0 + r, j) = k1;
        if ((e.a & 0xff) == DAG_FUSED) rd[i0 + r] = j + 1;
      }
      if (e.a >> 8) free_workers++; else free_express++;
    } else {
      if (cphase == 0) {                           /* potrf(cj) done: L(cj, cj) published */
        diag[cj] = 1; rd[cj] = cj + 1;
        out->diag_us.push_back(now);
        if (cj + 1 == T) { chain_done = true; out->chain_done_us = now; } else cphase = 1;
      } else if (cphase == 2) {                    /* block (cj+1, cj) solved */
        rd[cj + 1] = cj + 1;
        cphase = 3;
      } else if (cphase == 4) {                    /* block (cj+1, cj+1) has column cj */
        AP(cj + 1, cj + 1) = cj + 1;
        cj++; cphase = 0;
        evq.push(Ev{now + cm.potrf + cm.chain_pub, 1, 0, 0, 0, 0});
      }
    }
    chain_poll();
    assign();
  }
  out->makespan_us = now;
  out->busy_frac = W > 0 && now > 0 ? busy_time / (now * W) : 0.0;
  out->n_express = NE;
}

/* CPU check of a task list: replays it the way the device does (workers claim in list order and wait for their
   inputs, the chain runs beside them) with unit durations and verifies that nobody waits forever, that every block
   ends fully updated and solved, and that updates of one block never overlap.  Returns 0 when the list is valid. */
static inline int dag_check_schedule(int T, int workers, int n_express, const std::vector<DagTask> &tasks,
                                     const std::vector<DagTask> &express)
{
  std::vector<int> rd(T, 0), applied((size_t)T * T, 0);
  std::vector<char> diag(T, 0);
  auto AP = [&](int i, int j) -> int & { return applied[(size_t)i * T + j]; };
  size_t head = 0, ehead = 0;
  if (workers < 1) return 10;
  if (n_express >= workers) n_express = workers - 1;
  if (n_express == 0 && !express.empty()) return 11;
  std::vector<const DagTask *> held(workers, (const DagTask *)0);
  int cj = 0, cphase = 0;                           /* as above; phase 0 completes immediately in this replay */
  bool chain_done = false;
  auto ready = [&](const DagTask &t) -> bool {
    for (int r = 0; r < t.nr; r++) {
      const int i = t.i0 + r;
      if (AP(i, t.j) != t.k0) return false;
      if (rd[i] < std::min<int>(t.k1, i)) return false;
    }
    if (rd[t.j] < t.k1) return false;
    if (t.type == DAG_FUSED && !diag[t.j]) return false;
    return true;
  };
  for (long iter = 0; iter < 100000000L; iter++) {
    bool progress = false;
    for (int w = 0; w < workers; w++) {             /* claim: the device's rule (chol_dag.hip, worker loop) */
      if (held[w]) continue;
      if (w < n_express) { if (ehead < express.size()) { held[w] = &express[ehead++]; progress = true; } }
      else if (head < tasks.size()) { held[w] = &tasks[head++]; progress = true; }
      else if (ehead < express.size()) { held[w] = &express[ehead++]; progress = true; }
    }
    for (int w = 0; w < workers; w++) {             /* run whatever is ready */
      if (!held[w]) continue;
      const DagTask &t = *held[w];
      if (t.type > DAG_FUSED || (t.nr != 1 && t.nr != 2) || t.i0 + t.nr > T || t.j > t.i0 || t.k1 > t.j || t.k0 > t.k1) return 2;
      if (t.type == DAG_FUSED && (t.nr != 1 || t.k1 != t.j || t.i0 < t.j + 2)) return 3;
      if (t.type == DAG_UPD && t.k0 == t.k1) return 4;
      if (t.type == DAG_UPD && t.i0 == t.j && (t.nr != 1 || t.k1 > t.j - 1)) return 5;     /* column j-1 of (j, j) is the chain's */
      if (!ready(t)) continue;
      for (int r = 0; r < t.nr; r++) {
        AP(t.i0 + r, t.j) = t.k1;
        if (t.type == DAG_FUSED) { if (rd[t.i0 + r] != t.j) return 6; rd[t.i0 + r] = t.j + 1; }
      }
      held[w] = 0; progress = true;
    }
    if (!chain_done) {                              /* chain */
      if (cphase == 0) { if (AP(cj, cj) != cj) return 7; diag[cj] = 1; rd[cj] = cj + 1; if (cj + 1 == T) chain_done = true; else cphase = 1; progress = true; }
      else if (cphase == 1 && AP(cj + 1, cj) == cj) { rd[cj + 1] = cj + 1; cphase = 3; progress = true; }
      else if (cphase == 3 && AP(cj + 1, cj + 1) == cj) { AP(cj + 1, cj + 1) = cj + 1; cj++; cphase = 0; progress = true; }
    }
    bool all_idle = head >= tasks.size() && ehead >= express.size();
    for (int w = 0; w < workers && all_idle; w++) all_idle = held[w] == 0;
    if (all_idle && chain_done) break;
    if (!progress) return 1;                        /* deadlock */
  }
  for (int i = 0; i < T; i++) {
    if (rd[i] != i + 1) return 8;
    for (int j = 0; j <= i; j++) if (AP(i, j) != j) return 9;
  }
  return 0;
}

#endif
