"""-m gpu: the opt-in persistent task-DAG Cholesky (csrc/hip/chol_dag.hip, GSL_SINTERP_CHOL_DAG=1) keeps the contract of
gsl_linalg_cholesky_decomp1 (linalg/cholesky.c:88-131): only the lower triangle is read, L in the lower triangle, the
ORIGINAL matrix in the strict upper triangle (cholesky.c:103), GSL_EDOM with the first failing pivot; run to run bit-identical (every entry is one ascending fused-multiply-add chain
whatever the task grouping).  Not the default route (measured no faster: see the file's header)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def spd(n, seed):
    rng = np.random.default_rng(seed)
    m = rng.random((n, n))
    return np.tril(m) + np.tril(m, -1).T + 10.0 * n * np.eye(n)


@pytest.mark.parametrize("n", [256, 384, 1024, 2176])
def test_dag_factor_matches_lapack_and_saves_the_original_in_the_upper_triangle(pkg, monkeypatch, n):
    monkeypatch.setenv("GSL_SINTERP_CHOL_DAG", "1")
    ctx = pkg.HipContext.on_torch_stream(0)
    a = spd(n, n)
    a[np.triu_indices(n, 1)] = 7.25                      # decomp1 reads the lower triangle only ...
    sym = np.tril(a) + np.tril(a, -1).T
    ref = np.linalg.cholesky(sym)
    outs = []
    for rep in range(2):
        d = torch.from_numpy(a).cuda()
        st, info = ctx.cholesky_decomp1(n, d.data_ptr(), n)
        torch.cuda.synchronize()
        assert st == 0 and info == 0
        outs.append(d.cpu().numpy())
    got = outs[0]
    assert np.abs(np.tril(got) - ref).max() / np.abs(ref).max() < 1e-13
    assert np.array_equal(np.triu(got, 1), np.triu(sym, 1))   # ... and saves the original there (cholesky.c:103)
    assert np.array_equal(outs[0].view(np.uint64), outs[1].view(np.uint64))
    # the default route on the same input: same factor to rounding
    monkeypatch.delenv("GSL_SINTERP_CHOL_DAG")
    d = torch.from_numpy(a).cuda()
    st, info = ctx.cholesky_decomp1(n, d.data_ptr(), n)
    torch.cuda.synchronize()
    assert st == 0
    assert np.abs(np.tril(d.cpu().numpy()) - np.tril(got)).max() / np.abs(ref).max() < 1e-13


def test_dag_reports_the_first_failing_pivot(pkg, monkeypatch):
    n = 640
    a = spd(n, 5)
    a[300, 300] = -1.0                                   # column 301 (1-based) is the first non-positive pivot
    ctx = pkg.HipContext.on_torch_stream(0)
    res = []
    for env in ("1", None):
        if env:
            monkeypatch.setenv("GSL_SINTERP_CHOL_DAG", env)
        else:
            monkeypatch.delenv("GSL_SINTERP_CHOL_DAG")
        d = torch.from_numpy(a).cuda()
        res.append(ctx.cholesky_decomp1(n, d.data_ptr(), n))
        torch.cuda.synchronize()
    assert res[0][0] == res[1][0] == pkg.capi.GSL_EDOM
    assert res[0][1] == res[1][1] == 301


def test_dag_route_through_the_rbf_solve(pkg, orc, monkeypatch):
    """N = 1024 Gaussian system through gsl_sinterp_hip_rbf_solve with the DAG factorisation: weights against the oracle."""
    from gpu_util import dev, ptr
    n, dim = 1024, 2
    eps = orc.gaussian_eps(n, dim)
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    want = orc.rbf_solve(0, eps, x, f)
    monkeypatch.setenv("GSL_SINTERP_CHOL_DAG", "1")
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x, d_w = dev(x), dev(f)
    d_phi = torch.empty((n, n), dtype=torch.float64, device="cuda")
    st, route = ctx.rbf_solve(0, eps, ptr(d_x), n, dim, dim, ptr(d_phi), n, ptr(d_w))
    ctx.sync()
    assert st == 0 and route == 1
    assert np.abs(d_w.cpu().numpy() - want).max() / np.abs(want).max() < 1e-10
