"""Developer check: the 3-D culled sweep over batch sizes whose ideal grid is not a power of two, one-level
(GSL_SINTERP_SORT_LEVELS=1) against the two-level reorder (which rounds the grid to a power of two)."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
n, dim = 16384, 3
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.random((n, dim))).cuda()
w = torch.from_numpy(rng.standard_normal(n)).cuda()
eps = 2.0 * n ** (1.0 / dim)
for m in (1_000_000, 1_300_000, 2_400_000, 3_000_000, 5_000_000):
    y = torch.rand((m, dim), dtype=torch.float64, device="cuda")
    s = torch.empty(m, dtype=torch.float64, device="cuda")
    out = []
    for lv in ("1", "2"):
        os.environ["GSL_SINTERP_SORT_LEVELS"] = lv
        for rep in range(3):
            ctx.timer_start()
            ctx.rbf_eval(0, eps, x.data_ptr(), n, dim, dim, w.data_ptr(), y.data_ptr(), m, dim, s.data_ptr())
            t = ctx.timer_stop()
        out.append(t)
    G = int(np.ceil((m / 64.0) ** (1 / 3)))
    print(f"m={m}: ideal G={G}  one-level {out[0]:.3f} ms  two-level {out[1]:.3f} ms  ({out[1]/m*1e3:.3f} us/target)", flush=True)
