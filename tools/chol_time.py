"""Developer tool: the SPD factorisation alone (gsl_sinterp_hip_cholesky_decomp1) against numpy, with timings.
usage: python tools/chol_time.py [n ...]        (GSL_SINTERP_NO_FUSED_POTRF=1, GSL_SINTERP_CHOL_DAG=1, ... select variants)"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
for n in [int(v) for v in sys.argv[1:]] or [384, 1024, 4096, 8192, 16384]:
    rng = np.random.default_rng(n)
    m = rng.random((n, n))
    a = np.tril(m) + np.tril(m, -1).T + 10.0 * n * np.eye(n)
    d0 = torch.from_numpy(a).cuda()
    d = d0.clone()
    st, info = ctx.cholesky_decomp1(n, d.data_ptr(), n)
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    ref = np.linalg.cholesky(a)
    err = np.abs(np.tril(got) - ref).max() / np.abs(ref).max()
    up = np.array_equal(np.triu(got, 1), np.triu(a, 1))
    ms = []
    for rep in range(4):
        d.copy_(d0)
        ctx.timer_start()
        ctx.cholesky_decomp1(n, d.data_ptr(), n)
        ms.append(ctx.timer_stop())
    print(f"n={n}: st={st} info={info} rel err {err:.2e} upper = original {up}  {min(ms):.3f} ms", flush=True)
