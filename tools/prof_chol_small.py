"""Runs small Cholesky factorisations (n = 128: one chol_diag128_kernel launch, n = 4096) so that
rocprofv3 --kernel-trace --stats shows the per-kernel durations of the panel kernels in isolation."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
for n, reps in ((128, 20), (4096, 5)):
    rng = np.random.default_rng(n)
    m = rng.random((n, n))
    a = np.tril(m) + np.tril(m, -1).T + 10.0 * n * np.eye(n)
    d0 = torch.from_numpy(a).cuda()
    for _ in range(reps):
        d = d0.clone()
        st, info = ctx.cholesky_decomp1(n, d.data_ptr(), n)
        assert st == 0
    torch.cuda.synchronize()
    L = np.tril(d.cpu().numpy())
    print(n, 'rec err', np.abs(L @ L.T - a).max() / np.abs(a).max(), flush=True)
