"""-m gpu: solver breadth on the HIP path vs the oracle's restatements of
gsl_linalg_cholesky_decomp2 / solve2 / rcond (linalg/cholesky.c:392-537, condest.c:95-188),
gsl_linalg_LU_refine (linalg/lu.c:204-252) and gsl_linalg_pcholesky_decomp / solve (linalg/pcholesky.c:71-229).
Reference-held goldens: the Hilbert rcond table of linalg/test_cholesky.c:54-57 (1e-6 relative) and the
reconstruction / solve tolerances of the same file (tests/golden/reference_linalg_known_answers.json)."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import dev, ptr

pytestmark = pytest.mark.gpu
EPS = 2.2204460492503131e-16
LINALG = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_linalg_known_answers.json")))


def hilbert(n):
    i = np.arange(n)
    return 1.0 / (i[:, None] + i[None, :] + 1.0)


def posdef(n, seed):
    rng = np.random.default_rng(seed)
    m = rng.random((n, n))
    return np.tril(m) + np.tril(m, -1).T + 10.0 * n * np.eye(n)


def test_rcond_hilbert_table(pkg):
    ctx = pkg.HipContext.on_torch_stream(0)
    spec = LINALG["hilbert_rcond"]
    for n, want in enumerate(spec["values"], start=1):
        if want <= spec["min_checked"]:
            continue
        d_a = dev(hilbert(n))
        st, info = ctx.cholesky_decomp1(n, ptr(d_a), n)
        assert st == 0
        got = ctx.cholesky_rcond(n, ptr(d_a), n)
        assert abs(got - want) <= spec["rel_tol"] * want, (n, got, want)


@pytest.mark.parametrize("n", [5, 64, 200, 1000, 2560])
def test_decomp2_solve2_rcond_match_oracle(pkg, orc, n):
    ctx = pkg.HipContext.on_torch_stream(0)
    if n >= 1000:
        x = orc.synth_centres(n, 2)
        a = orc.rbf_fill(0, orc.gaussian_eps(n, 2), x)          # the kernel matrix this path is for
    else:
        a = posdef(n, n)
    a = a * np.outer(np.linspace(1.0, 30.0, n), np.linspace(1.0, 30.0, n))   # badly scaled: what decomp2 is for
    b = np.arange(1.0, n + 1.0)
    st_o, v_o, s_o = orc.cholesky_decomp2(a)
    assert st_o == 0
    d_a, d_s = dev(a), torch.empty(n, dtype=torch.float64, device="cuda")
    st, info = ctx.cholesky_decomp2(n, ptr(d_a), n, ptr(d_s))
    assert st == 0 and info == 0
    ctx.sync()
    got = d_a.cpu().numpy()
    assert np.array_equal(d_s.cpu().numpy(), s_o)                               # 1/sqrt(A_ii): correctly rounded on both sides
    assert np.abs(np.tril(got) - np.tril(v_o)).max() <= 1e-12                   # factor of the scaled matrix (entries O(1))
    assert np.array_equal(np.triu(got, 1), np.triu(v_o, 1))                     # scaled original kept above the diagonal
    d_x = dev(b)
    ctx.cholesky_svx2(n, ptr(d_a), n, ptr(d_s), ptr(d_x))
    ctx.sync()
    x_o = orc.cholesky_solve2(v_o, s_o, b)
    assert np.abs(d_x.cpu().numpy() - x_o).max() <= 1e-10 * np.abs(x_o).max()
    r_o = orc.cholesky_rcond(v_o)
    r = ctx.cholesky_rcond(n, ptr(d_a), n)
    assert 0 < r <= 1 and abs(r - r_o) <= 1e-6 * r_o, (r, r_o)
    # the true 1-norm condition number brackets the estimate from above (it is a lower bound of |A^-1|)
    scaled = a * np.outer(s_o, s_o)
    true_rcond = 1.0 / (np.abs(scaled).sum(axis=0).max() * np.abs(np.linalg.inv(scaled)).sum(axis=0).max())
    assert r >= true_rcond * (1 - 1e-8) and r <= 3.0 * true_rcond


@pytest.mark.parametrize("n", [1, 2, 3, 7, 12, 33, 50, 300, 1500])
def test_pcholesky_bitexact_vs_oracle(pkg, orc, n):
    ctx = pkg.HipContext.on_torch_stream(0)
    tol = LINALG["solver_breadth_tolerances"]
    mats = [posdef(n, 100 + n)]
    if n <= 12:
        mats.append(hilbert(n))
    if n >= 33:
        # positive SEMI-definite, rank n/2, plus a small nugget: what the pivoted form is for (plain Cholesky fails on the rank-deficient one)
        rng = np.random.default_rng(n)
        g = rng.standard_normal((n, n // 2))
        mats.append(g @ g.T + 1e-9 * np.eye(n))
    for mi, a in enumerate(mats):
        st_o, ldlt_o, perm_o = orc.pcholesky_decomp(a)
        d_a, d_p = dev(a), torch.empty(n, dtype=torch.int32, device="cuda")
        ctx.pcholesky_decomp(n, ptr(d_a), n, ptr(d_p))
        ctx.sync()
        assert np.array_equal(d_p.cpu().numpy().astype(np.uintp), perm_o)        # same pivots ...
        assert np.array_equal(d_a.cpu().numpy(), ldlt_o)                         # ... same bits (no FMA contraction on either side)
        L = np.tril(ldlt_o, -1) + np.eye(n)
        rec = L @ np.diag(np.diag(ldlt_o)) @ L.T
        assert np.abs(rec - a[np.ix_(perm_o, perm_o)]).max() <= tol["pcholesky_reconstruct_eps_mult_per_n"] * n * EPS * np.abs(a).max()
        rng = np.random.default_rng(7 * n)
        sol = rng.random(n)
        rhs = a @ sol
        d_x = dev(rhs)
        ctx.pcholesky_svx(n, ptr(d_a), n, ptr(d_p), ptr(d_x))
        ctx.sync()
        x_o = orc.pcholesky_solve(ldlt_o, perm_o, rhs)
        got = d_x.cpu().numpy()
        if mi == 0:                                        # well conditioned: the two sweeps agree to rounding
            assert np.abs(got - x_o).max() <= 1e-10 * max(1.0, np.abs(x_o).max())
        else:                                              # Hilbert / near-singular: compare backward errors (cond up to 1e16)
            assert np.abs(a @ got - rhs).max() <= 1e-11 * np.abs(a).max() * max(1.0, np.abs(got).max()) * n
            assert np.abs(a @ x_o - rhs).max() <= 1e-11 * np.abs(a).max() * max(1.0, np.abs(x_o).max()) * n
    # reference tolerance on the well-conditioned one (test_cholesky.c:794)
    a = posdef(n, 100 + n)
    sol = np.random.default_rng(n).random(n)
    d_a, d_p, d_x = dev(a), torch.empty(n, dtype=torch.int32, device="cuda"), dev(a @ sol)
    ctx.pcholesky_decomp(n, ptr(d_a), n, ptr(d_p))
    ctx.pcholesky_svx(n, ptr(d_a), n, ptr(d_p), ptr(d_x))
    ctx.sync()
    assert np.abs(d_x.cpu().numpy() - sol).max() <= tol["pcholesky_solve_eps_mult_per_n"] * n * EPS * 4


@pytest.mark.parametrize("n", [4, 12, 100, 700])
def test_lu_refine_matches_oracle_and_improves(pkg, orc, n):
    ctx = pkg.HipContext.on_torch_stream(0)
    if n == 12:
        a = hilbert(12)
    else:
        rng = np.random.default_rng(n)
        a = rng.standard_normal((n, n)) + np.diag(np.linspace(1, 1e6, n))
    b = np.arange(1.0, n + 1.0)
    lu_o, perm_o, _ = orc.lu_decomp(a)
    st, x_o = orc.lu_solve(lu_o, perm_o, b)
    x0 = x_o * (1 + 1e-7 * np.cos(np.arange(n)))                              # a perturbed solution to refine
    st, xr_o = orc.lu_refine(a, lu_o, perm_o, b, x0)
    d_a, d_lu = dev(a), dev(a)
    d_perm = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.lu_decomp(n, ptr(d_lu), n, ptr(d_perm))
    d_x, d_b, d_work = dev(x0), dev(b), torch.empty(n, dtype=torch.float64, device="cuda")
    assert ctx.lu_refine(n, ptr(d_a), n, ptr(d_lu), n, ptr(d_perm), ptr(d_b), ptr(d_x), ptr(d_work)) == 0
    ctx.sync()
    got = d_x.cpu().numpy()
    scale = np.abs(xr_o).max()
    if n == 12:                                         # Hilbert(12), cond ~ 1e16: the reference's own bar (linalg/test.c:3384-3402: 0.5 relative)
        sol = np.array(LINALG["hilbert"]["12"]["solution"])
        assert (np.abs(got - sol) <= 0.5 * np.abs(sol)).all() and (np.abs(xr_o - sol) <= 0.5 * np.abs(sol)).all()
    else:
        assert np.abs(got - xr_o).max() <= 1e-10 * scale
    if n != 12:
        assert np.abs(a @ got - b).max() < 1e-3 * np.abs(a @ x0 - b).max()        # the step reduces the residual
    # singular LU: GSL_EDOM like lu.c:231-234
    d_sing = dev(np.zeros((n, n)))
    d_p2 = torch.arange(n, dtype=torch.int32, device="cuda")
    assert ctx.lu_refine(n, ptr(d_a), n, ptr(d_sing), n, ptr(d_p2), ptr(d_b), ptr(d_x), ptr(d_work)) == pkg.capi.GSL_EDOM


@pytest.mark.parametrize("solver", ["cholesky2", "pcholesky", "lu_refine"])
def test_facade_solver_choice_and_rcond(pkg, orc, solver):
    n, dim, m = 1500, 2, 4000
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    y = orc.synth_targets(0, m, dim)
    eps = orc.gaussian_eps(n, dim)
    want = orc.rbf_eval(0, eps, x, orc.rbf_solve(0, eps, x, f), y)
    s = pkg.Sinterp("gaussian", dim, n, 0)
    sid = {"cholesky2": pkg.capi.SOLVER_CHOLESKY2, "pcholesky": pkg.capi.SOLVER_PCHOLESKY, "lu_refine": pkg.capi.SOLVER_LU_REFINE}[solver]
    assert s.set_solver(sid) == 0 and s.set_rcond(True) == 0
    assert s.init(x, f) == 0
    assert s.route() == {"cholesky2": 4, "pcholesky": 5, "lu_refine": 6}[solver]
    st, got, _ = s.eval_many(y)
    assert st == 0 and np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    st, rc = s.rcond()
    if solver == "cholesky2":
        phi = orc.rbf_fill(0, eps, x)
        _, v_o, _ = orc.cholesky_decomp2(phi)
        assert st == 0 and abs(rc - orc.cholesky_rcond(v_o)) <= 1e-6 * rc
    else:
        assert st == pkg.capi.GSL_EINVAL and np.isnan(rc)
    # default solver + rcond: estimate of the unscaled Gaussian matrix (kappa ~ 1e4 at eps = 2/h)
    d = pkg.Sinterp("gaussian", dim, n, 0)
    assert d.set_rcond(True) == 0 and d.init(x, f) == 0 and d.route() == 1
    st, rc = d.rcond()
    st_o, llt_o = orc.cholesky_decomp1(orc.rbf_fill(0, eps, x))
    assert st == 0 and abs(rc - orc.cholesky_rcond(llt_o)) <= 1e-6 * rc and 1e-7 < rc < 1e-2
    # thin-plate spline: Cholesky-type solvers are refused, LU + refinement works
    t = pkg.Sinterp("tps", dim, n, 0)
    assert t.set_solver(pkg.capi.SOLVER_PCHOLESKY) == pkg.capi.GSL_EINVAL
    assert t.set_solver(pkg.capi.SOLVER_LU_REFINE) == 0 and t.init(x, f) == 0 and t.route() == 6
    st, got, _ = t.eval_many(y)
    want = orc.rbf_eval(1, 0.0, x, orc.rbf_solve(1, 0.0, x, f), y)
    assert st == 0 and np.abs(got - want).max() <= 1e-10 * np.abs(want).max()


@pytest.mark.parametrize("n", [3, 12, 50, 400])
def test_pcholesky_decomp2_svx2_bitexact_and_rcond_table(pkg, orc, n):
    """gsl_linalg_pcholesky_decomp2 / _svx2 / _rcond (linalg/pcholesky.c:231-353, 472-580).  The scaled, pivoted LDL^T has the
    pivots and the BITS of the oracle's restatement (no FMA contraction on either side); the solve agrees to rounding;
    rcond reproduces the reference-held Hilbert table (linalg/test_cholesky.c:54-57, 675-687: 1e-6) and the oracle."""
    ctx = pkg.HipContext.on_torch_stream(0)
    a = posdef(n, 500 + n) * np.outer(np.linspace(1.0, 20.0, n), np.linspace(1.0, 20.0, n))
    st_o, ldlt_o, perm_o, s_o = orc.pcholesky_decomp2(a)
    d_a, d_p, d_s = dev(a), torch.empty(n, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")
    ctx.pcholesky_decomp2(n, ptr(d_a), n, ptr(d_p), ptr(d_s))
    ctx.sync()
    assert np.array_equal(d_s.cpu().numpy(), s_o)
    assert np.array_equal(d_p.cpu().numpy().astype(np.uintp), perm_o)
    assert np.array_equal(d_a.cpu().numpy(), ldlt_o)
    assert np.array_equal(np.triu(ldlt_o, 1), np.triu(a, 1))                   # the UNSCALED matrix above the diagonal
    sol = np.random.default_rng(n).random(n)
    d_x = dev(a @ sol)
    ctx.pcholesky_svx2(n, ptr(d_a), n, ptr(d_p), ptr(d_s), ptr(d_x))
    ctx.sync()
    x_o = orc.pcholesky_solve2(ldlt_o, perm_o, s_o, a @ sol)
    got = d_x.cpu().numpy()
    assert np.abs(got - x_o).max() <= 1e-10 * np.abs(x_o).max() and np.abs(got - sol).max() <= 64.0 * n * EPS * 20
    # rcond: the unscaled decomposition (what the reference's test calls it on)
    b = posdef(n, 900 + n)
    st_b, ldlt_b, perm_b = orc.pcholesky_decomp(b)
    d_b, d_pb = dev(b), torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.pcholesky_decomp(n, ptr(d_b), n, ptr(d_pb))
    r, r_o = ctx.pcholesky_rcond(n, ptr(d_b), n, ptr(d_pb)), orc.pcholesky_rcond(ldlt_b, perm_b)
    true_rcond = 1.0 / (np.abs(b).sum(axis=0).max() * np.abs(np.linalg.inv(b)).sum(axis=0).max())
    assert abs(r - r_o) <= 1e-6 * r_o and true_rcond * (1 - 1e-8) <= r <= 3.0 * true_rcond
    if n == 3:
        spec = LINALG["hilbert_rcond"]
        for m, want in enumerate(spec["values"], start=1):
            if want <= 1.0e-12:
                continue
            d_h, d_ph = dev(hilbert(m)), torch.empty(m, dtype=torch.int32, device="cuda")
            ctx.pcholesky_decomp(m, ptr(d_h), m, ptr(d_ph))
            got_r = ctx.pcholesky_rcond(m, ptr(d_h), m, ptr(d_ph))
            assert abs(got_r - want) <= 1.0e-6 * want, (m, got_r, want)
