"""-m gpu: the thin-plate spline WITH its affine tail (gsl_sinterp_rbf_tps_affine; SURVEY.md 8 rows a8 / a10 / (d): the
"N + d + 1" augmented system).  The reference has no RBF code (README:18-26): PARITY UNPINNED.  Checked against the oracle's
composition -- libm fill + the pinned gsl_linalg_LU_decomp / _svx restatement on the full saddle matrix + naive sums -- at the
1e-10 tolerance of the RBF path, and by the properties that define the interpolant."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("dim,n,m", [(2, 512, 4000), (2, 1000, 3000), (3, 640, 2000), (1, 200, 500)])
def test_affine_tps_matches_oracle(pkg, orc, dim, n, m):
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x) + 2.0
    y = orc.synth_targets(0, m, dim)
    s = pkg.Sinterp("tps_affine", dim, n, 0)
    assert s.name() == "rbf-thin-plate-spline-affine"
    assert s.init(x, f) == 0 and s.route() == 9            # block elimination on the shifted SPD Cholesky
    st, got, _ = s.eval_many(y)
    w, c = orc.rbf_solve_affine(1, 0.0, x, f)
    want = orc.rbf_eval_affine(1, 0.0, c, x, w, y)
    assert st == 0 and relerr(got, want) < TOL, relerr(got, want)
    stp, cg = s.poly()
    assert stp == 0 and np.abs(cg - c).max() <= 1e-7 * max(1.0, np.abs(c).max())     # reported, like the weights (cond(Phi) level)
    stw, wg = s.weights()
    P = np.hstack([np.ones((n, 1)), x])
    assert stw == 0 and np.abs(P.T @ wg).max() < 1e-8 * max(1.0, np.abs(wg).max())   # side condition P^T w = 0
    # interpolation at the centres, single-point entry
    st, at, _ = s.eval_many(x[:300])
    assert st == 0 and np.abs(at - f[:300]).max() < 1e-9
    st1, v1 = s.eval_e(y[7])
    assert st1 == 0 and abs(v1 - got[7]) <= 1e-12 * max(1.0, abs(got[7]))


def test_affine_tps_reproduces_linear_data_with_zero_weights(pkg, orc):
    n, m = 512, 5000
    x = orc.synth_centres(n, 2)
    y = orc.synth_targets(0, m, 2) * 3.0 - 1.0                 # also outside the hull: the tail extrapolates linearly
    f = 0.75 - 2.0 * x[:, 0] + 0.5 * x[:, 1]
    s = pkg.Sinterp("tps_affine", 2, n, 0)
    assert s.init(x, f) == 0
    st, got, _ = s.eval_many(y)
    want = 0.75 - 2.0 * y[:, 0] + 0.5 * y[:, 1]
    assert st == 0 and np.abs(got - want).max() < 1e-12 * np.abs(want).max() * 10
    _, w = s.weights()
    _, c = s.poly()
    assert np.abs(w).max() < 1e-9 and np.abs(c - [0.75, -2.0, 0.5]).max() < 1e-11
    # the plain thin-plate spline does NOT reproduce it outside the data: the two types differ there
    p = pkg.Sinterp("tps", 2, n, 0)
    assert p.init(x, f) == 0
    _, plain, _ = p.eval_many(y)
    assert np.abs(plain - want).max() > 1e-3


def test_affine_tps_reference_route_and_checkpoint(pkg, orc, tmp_path):
    """GSL_SINTERP_FORCE_LU=1: pivoted LU of the augmented (n + d + 1) matrix on the device (route 10) = the oracle's route;
    checkpoint round trip carries the tail; device groups shard like the other RBF types; solver knobs are refused."""
    import os
    import subprocess
    import sys
    n, m = 600, 2500
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    y = orc.synth_targets(0, m, 2)
    s = pkg.Sinterp("tps_affine", 2, n, 0)
    assert s.set_solver(1) == pkg.GSL_EINVAL and s.set_rcond(True) == pkg.GSL_EINVAL and s.set_solver(0) == 0
    assert s.init(x, f) == 0
    st, got, _ = s.eval_many(y)
    path = tmp_path / "tps_affine.bin"
    assert s.fwrite(path) == 0
    t = pkg.Sinterp("tps_affine", 2, n, 0)
    assert t.fread(path) == 0
    st2, again, _ = t.eval_many(y)
    assert st == 0 and st2 == 0 and np.array_equal(got.view(np.uint64), again.view(np.uint64))
    assert pkg.Sinterp("tps", 2, n, 0).fread(path) == pkg.capi.GSL_EBADLEN
    g = pkg.Sinterp("tps_affine", 2, n, 0)
    assert g.set_device_list([0, 0, 0]) == 0 and g.init(x, f) == 0
    st3, shard, _ = g.eval_many(y)
    assert st3 == 0 and np.array_equal(got.view(np.uint64), shard.view(np.uint64))
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as g, oracle_lib as orc\n"
        "pkg = g.load_package()\n"
        "x = orc.synth_centres(%d, 2); f = orc.synth_response(x); y = orc.synth_targets(0, %d, 2)\n"
        "s = pkg.Sinterp('tps_affine', 2, %d, 0)\n"
        "assert s.init(x, f) == 0 and s.route() == 10, s.route()\n"
        "st, got, _ = s.eval_many(y)\n"
        "w, c = orc.rbf_solve_affine(1, 0.0, x, f)\n"
        "want = orc.rbf_eval_affine(1, 0.0, c, x, w, y)\n"
        "err = np.abs(got - want).max() / np.abs(want).max()\n"
        "assert st == 0 and err < 1e-10, err\n"
        "print('ok', err)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), n, m, n)
    env = dict(os.environ, GSL_SINTERP_FORCE_LU="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:]
