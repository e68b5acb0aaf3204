// Developer tool: replays the task-DAG list scheduler of csrc/hip/chol_dag_sched.h with its cost model.
//   g++ -O2 -std=c++17 -I../../gsl-scattered-interpolation_amd/csrc/hip simulate.cpp -o /tmp/simulate && /tmp/simulate 64 potrf=31 kcb=4
#include "chol_dag_sched.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char **argv) {
  int T = atoi(argv[1]);
  DagCost cm = dag_default_cost(T);
  for (int i = 2; i < argc; i++) {
    char *eq = strchr(argv[i], '=');
    if (!eq) continue;
    *eq = 0;
    const double v = atof(eq + 1);
    const char *k = argv[i];
#define F(name) if (!strcmp(k, #name)) cm.name = v
#define I(name) if (!strcmp(k, #name)) cm.name = (int)v
    F(step256); F(step128); F(upd_fixed256); F(upd_fixed128); F(fused_step); F(fused_fixed); F(fused_trsm);
    F(potrf); F(chain_trsm); F(chain_syrk); F(chain_pub); I(kcb); I(near_rows); I(express); I(urgent_rows);
  }
  DagSchedule s;
  dag_build_schedule(T, 255, cm, &s);
  double work = 0;
  auto acc = [&](const std::vector<DagTask> &v) {
    for (auto &t : v) {
      int kb = t.k1 - t.k0;
      if (t.type == DAG_FUSED) work += cm.fused_fixed + kb * 8 * cm.fused_step + cm.fused_trsm;
      else work += t.nr == 2 ? cm.upd_fixed256 + kb * 8 * cm.step256 : cm.upd_fixed128 + kb * 8 * cm.step128;
    }
  };
  acc(s.tasks); acc(s.express);
  printf("T=%d kcb=%d near=%d express=%d urg=%d: makespan %.0f us, chain done %.0f (%.1f us/step), busy %.3f, lists %zu + %zu, work/255 = %.2f ms, check %d\n",
         T, cm.kcb, cm.near_rows, cm.express, cm.urgent_rows, s.makespan_us, s.chain_done_us, s.chain_done_us / T, s.busy_frac, s.tasks.size(), s.express.size(),
         work / 255e3, dag_check_schedule(T, 255, s.n_express, s.tasks, s.express));
}
