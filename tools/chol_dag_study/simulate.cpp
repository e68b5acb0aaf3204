#include "chol_dag_sched.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv) {
  int T = atoi(argv[1]);
  DagCost cm = dag_default_cost(T);
  if (argc > 2) cm.kcb = atoi(argv[2]);
  if (argc > 3) cm.near_rows = atoi(argv[3]);
  if (argc > 4) cm.express = atoi(argv[4]);
  if (argc > 5) cm.urgent_rows = atoi(argv[5]);
  DagSchedule s;
  dag_build_schedule(T, 255, cm, &s);
  double tp = 0, t1c = 0, t1s = 0, tf = 0; size_t np = 0, n1c = 0, n1s = 0, nf = 0;
  auto acc = [&](const std::vector<DagTask> &v) {
    for (auto &t : v) {
      int kb = t.k1 - t.k0;
      if (t.type == DAG_FUSED) { tf += cm.fused_fixed + kb * 8 * cm.fused_step + cm.fused_trsm; nf++; }
      else if (t.nr == 2) { tp += cm.upd_fixed + kb * 8 * cm.step256; np++; }
      else if (kb >= cm.kcb) { t1c += cm.upd_fixed + kb * 8 * cm.step128; n1c++; }
      else { t1s += cm.upd_fixed + kb * 8 * cm.step128; n1s++; }
    }
  };
  acc(s.tasks); acc(s.express);
  printf("T=%d kcb=%d near=%d express=%d urg=%d: makespan %.0f chain %.0f busy %.3f | worker-ms: pairs %.1f (%zu) single-chunk %.1f (%zu) single-small %.1f (%zu) fused %.1f (%zu)  total/255 = %.2f ms\n",
         T, cm.kcb, cm.near_rows, cm.express, cm.urgent_rows, s.makespan_us, s.chain_done_us, s.busy_frac, tp / 1e3, np, t1c / 1e3, n1c, t1s / 1e3, n1s, tf / 1e3, nf, (tp + t1c + t1s + tf) / 255e3);
}
