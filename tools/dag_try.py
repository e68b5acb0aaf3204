"""Developer tool: the opt-in task-DAG Cholesky (chol_dag.hip) against numpy at a few sizes, with timing; set
GSL_SINTERP_CHOL_DAG=0 to time the default driver instead, GSL_SINTERP_DAG_PROF=1 for the chain / worker time stamps.
usage: python tools/dag_try.py [n ...]"""
import os, sys, time
os.environ.setdefault("GSL_SINTERP_CHOL_DAG", "1")
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
sizes = [int(v) for v in sys.argv[1:]] or [256, 384, 512, 1024, 2048, 4096]
for n in sizes:
    rng = np.random.default_rng(n)
    m = rng.random((n, n))
    a = np.tril(m) + np.tril(m, -1).T + 10.0 * n * np.eye(n)
    d0 = torch.from_numpy(a).cuda()
    d = d0.clone()
    t0 = time.time()
    st, info = ctx.cholesky_decomp1(n, d.data_ptr(), n)
    torch.cuda.synchronize()
    first = time.time() - t0
    L = np.tril(d.cpu().numpy())
    ref = np.linalg.cholesky(a)
    err = np.abs(L - ref).max() / np.abs(ref).max()
    up = np.array_equal(np.triu(d.cpu().numpy(), 1), np.triu(a, 1))
    ms = []
    for rep in range(3):
        d.copy_(d0)
        ctx.timer_start()
        ctx.cholesky_decomp1(n, d.data_ptr(), n)
        ms.append(ctx.timer_stop())
    print(f"n={n}: st={st} info={info} rel err {err:.2e} upper kept {up}  first call {first*1e3:.1f} ms, then {min(ms):.3f} ms", flush=True)
