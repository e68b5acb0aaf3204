"""-m gpu: ordinary kriging (README:24 of the reference lists it as future work: no reference code, PARITY UNPINNED).
Checked against the oracle's composition -- libm fill + nugget + the pinned gsl_linalg_cholesky_decomp1 / _svx
restatement with two right-hand sides (pivoted LDL^T when the covariance matrix is only semi-definite) -- at the
1e-10 tolerance of the RBF path, and by the properties that define the method."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("dim,n,m,nugget", [(2, 700, 3000, 0.0), (2, 1500, 5000, 1e-3), (3, 1200, 4000, 1e-2), (1, 300, 1000, 0.0)])
def test_kriging_matches_oracle(pkg, orc, dim, n, m, nugget):
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x) + 3.0
    y = orc.synth_targets(0, m, dim)
    eps = orc.gaussian_eps(n, dim)
    s = pkg.Sinterp("kriging", dim, n, 0)
    assert s.name() == "ordinary-kriging-gaussian"
    assert s.set_nugget(nugget) == 0
    assert s.init(x, f) == 0 and s.route() == 7
    st, got, _ = s.eval_many(y)
    w, mu = orc.krige_solve(0, eps, nugget, x, f)
    want = orc.krige_eval(0, eps, mu, x, w, y)
    stm, mean = s.mean()
    assert st == 0 and stm == 0
    assert abs(mean - mu) <= TOL * abs(mu)
    assert relerr(s.weights()[1], w) < 1e-8 and relerr(got, want) < TOL
    # the constraint of the dual system and the behaviour at the data: s(x_i) = f_i - nugget w_i
    wg = s.weights()[1]
    assert abs(wg.sum()) <= 1e-9 * np.abs(wg).sum()
    st, at_data, _ = s.eval_many(x)
    assert np.abs(at_data - (f - nugget * wg)).max() <= 1e-9 * np.abs(f).max()
    # far from every site the predictor reverts to the mean (every covariance term is below the sweep's cut-off)
    st, far, _ = s.eval_many(np.full((3, dim), 50.0))
    assert (far == mean).all()


def test_kriging_constant_data_and_errors(pkg, orc):
    n, dim = 400, 2
    x = orc.synth_centres(n, dim)
    s = pkg.Sinterp("kriging", dim, n, 0)
    assert s.init(x, np.full(n, 2.5)) == 0
    st, mean = s.mean()
    assert abs(mean - 2.5) < 1e-12 and np.abs(s.weights()[1]).max() < 1e-9        # a constant field is its own mean
    st, v, _ = s.eval_many(orc.synth_targets(0, 500, dim))
    assert np.abs(v - 2.5).max() < 1e-10
    g = pkg.Sinterp("gaussian", dim, n, 0)
    assert g.set_nugget(0.1) == pkg.GSL_EINVAL and g.mean()[0] == pkg.GSL_EINVAL   # kriging interpolants only
    assert s.set_nugget(-1.0) == pkg.GSL_EDOM


def test_kriging_duplicate_sites_and_the_pivoted_route(pkg, orc):
    """Two coincident sites make the covariance matrix exactly singular: with nugget 0 the system has no unique
    solution (the Cholesky pivot is exactly 0, and so is the last D of the pivoted LDL^T) -> GSL_EDOM; a nugget makes
    it SPD again.  A numerically semi-definite matrix (flat kernel, cond ~ 1/eps_machine: a negative Cholesky pivot
    from rounding) is what the pivoted LDL^T route of linalg/pcholesky.c is for: init still succeeds there."""
    n, dim = 300, 2
    x = orc.synth_centres(n, dim)
    x[-1] = x[0]
    f = orc.synth_response(x)
    s = pkg.Sinterp("kriging", dim, n, 0)
    assert s.init(x, f) == pkg.GSL_EDOM
    assert s.set_nugget(1e-4) == 0
    assert s.init(x, f) == 0 and s.route() == 7
    st, at, _ = s.eval_many(x[[0, n - 1]])
    assert at[0] == at[1]                                                   # one site, one prediction
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    routes = []
    for factor in (0.15, 0.05, 0.02, 0.01):
        flat = pkg.Sinterp("kriging", dim, n, 0)
        assert flat.set_shape(factor * orc.gaussian_eps(n, dim)) == 0
        assert flat.init(x, f) == 0
        routes.append(flat.route())
        st, v, _ = flat.eval_many(orc.synth_targets(0, 500, dim))
        assert st == 0 and np.isfinite(v).all()
    print("flat covariances: routes", routes)
    assert set(routes) <= {7, 8} and 8 in routes                            # the pivoted route was exercised


def test_kriging_checkpoint_and_sharded_eval(pkg, orc, tmp_path):
    n, dim, m = 600, 2, 5000
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x) - 1.0
    y = orc.synth_targets(0, m, dim)
    s = pkg.Sinterp("kriging", dim, n, 0)
    assert s.set_nugget(1e-3) == 0 and s.init(x, f) == 0
    st, want, _ = s.eval_many(y)
    path = tmp_path / "krige.bin"
    assert s.fwrite(str(path)) == 0
    r = pkg.Sinterp("kriging", dim, n, 0)
    assert r.fread(str(path)) == 0
    st, got, _ = r.eval_many(y)
    assert np.array_equal(got, want) and r.mean()[1] == s.mean()[1]
    wrong = pkg.Sinterp("gaussian", dim, n, 0)
    assert wrong.fread(str(path)) == pkg.capi.GSL_EBADLEN                     # the type is part of the checkpoint
    g = pkg.Sinterp("kriging", dim, n, 0)
    assert g.set_device_list([0, 0, 0]) == 0 and g.set_nugget(1e-3) == 0 and g.init(x, f) == 0
    st, sharded, _ = g.eval_many(y)
    assert np.array_equal(sharded, want)                                 # shards = single device, bit for bit
