"""Pin the CPU oracle (oracle/*.c) against the reference's own known answers.

Sources of truth (no reference code runs here):
  * tests/golden/survey_known_answers.json  -- outputs of the reference captured at
    survey time: leaf index, leaf vertices, %.17g values.
  * interpolation/scattered_interp_example.c:51-77 -- asserted known answers.
  * tests/golden/reference_linalg_known_answers.json -- linalg/test.c fixtures.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SURVEY = json.load(open(os.path.join(HERE, "golden", "survey_known_answers.json")))
LINALG = json.load(open(os.path.join(HERE, "golden", "reference_linalg_known_answers.json")))
EPS = 2.2204460492503131e-16


def build_cfg(orc, weather, cfg):
    t = orc.Tree(2, 50)
    if cfg == "cfg0":
        assert t.init(None, flags=1) == 0
        for i in range(50):
            leaf = t.find_leaf(weather, weather[i, :2])
            assert t.insert_point(leaf, weather) == 0
    elif cfg == "cfg2":
        assert t.init(weather, flags=0) == 0
    else:
        assert t.init(weather, flags=0, seed=0) == 0
    return t


@pytest.mark.parametrize("cfg", ["cfg0", "cfg2", "cfg1"])
def test_survey_known_answers(orc, weather, cfg):
    spec = SURVEY["configs"][cfg]
    t = build_cfg(orc, weather, cfg)
    assert t.c.n_nodes == spec["n_nodes"]
    if "shift" in spec:
        for i in range(2):
            assert t.c.shift[i] == float(spec["shift"][i])
            assert t.c.scale[i] == float(spec["scale"][i])
    resp = weather[:, 2]
    for q in spec["queries"]:
        leaf = t.find_leaf(weather, q["point"])
        assert leaf == q["leaf"]
        assert t.data_rows(leaf) == q["rows"]
        v = t.interp_point(leaf, weather, resp, q["point"])
        assert v == float(q["value"]), (v, q["value"])          # bit-exact: %.17g round-trips
    assert t.check_leaf_nodes() == 1
    assert t.check_delaunay(weather) == 1


def test_reference_trivial_test_asserts(orc):
    """scattered_interp_example.c:38-77, the reference's only asserted scattered-interp test."""
    t = orc.Tree(2, 50)
    assert t.init(None, flags=1) == 0
    data = np.array([[-88.0, 41.0], [-89.0, 41.0]])
    leaf = t.find_leaf(None, data[0])
    assert leaf == 0
    assert t.interp_point(leaf, data, None, data[0]) == 0.0            # :51-52 empty cage -> exactly 0
    assert t.insert_point(leaf, data) == 0
    ty, pidx, links = t.arrays()
    assert ty[leaf] != 0                                                 # :59 !LEAF(leaf)
    kids = links[0:3]
    assert [list(pidx[3 * k:3 * k + 3]) for k in kids] == [[0, -2, -3], [0, -1, -3], [0, -1, -2]]   # :60-68
    assert t.in_hypersphere(0, data, 0) == 1                             # :70
    leaf2 = t.find_leaf(data, data[1])
    assert t.vertices(leaf2) == [0, -2, -3]                              # :74-77


def test_mt19937_known_answer(orc):
    """rng/test.c:145 -- 1000th output of mt19937 seeded with 4357; seed 0 aliases 4357 (rng/mt.c:137)."""
    ka = LINALG["mt19937"]
    for seed in (ka["seed"], 0):
        r = orc.lib().oracle_mt_alloc(seed)
        k = 0
        for _ in range(ka["n"]):
            k = orc.lib().oracle_mt_get(r)
        orc.lib().oracle_mt_free(r)
        assert k == ka["value"]


def rel_ok(x, actual, eps):
    """linalg/test.c:118-133 check()"""
    if x == actual:
        return True
    if actual == 0:
        return abs(x) <= eps
    return abs(x - actual) / abs(actual) <= eps


def hilbert(n):
    i = np.arange(n)
    return 1.0 / (i[:, None] + i[None, :] + 1.0)


def vandermonde(n):
    return np.array([[float(i + 1.0) ** (n - j - 1.0) for j in range(n)] for i in range(n)])


@pytest.mark.parametrize("n", [2, 3, 4, 12])
def test_lu_solve_hilbert_vandermonde(orc, n):
    rhs = np.arange(1, n + 1, dtype=np.float64)
    for name, mat in (("hilbert", hilbert(n)), ("vandermonde", vandermonde(n))):
        spec = LINALG[name][str(n)]
        lu, perm, _ = orc.lu_decomp(mat)
        st, x = orc.lu_solve(lu, perm, rhs)
        assert st == 0
        tol = spec["lu_eps_mult"] * EPS if "lu_eps_mult" in spec else spec["lu_abs_tol"]
        assert all(rel_ok(x[i], spec["solution"][i], tol) for i in range(n)), (name, n, x)


@pytest.mark.parametrize("n", [2, 3, 4, 12])
def test_cholesky_solve_hilbert(orc, n):
    spec = LINALG["hilbert"][str(n)]
    st, llt = orc.cholesky_decomp1(hilbert(n))
    assert st == 0
    x = orc.cholesky_solve(llt, np.arange(1, n + 1, dtype=np.float64))
    tol = spec["chol_eps_mult"] * EPS if "chol_eps_mult" in spec else spec["chol_abs_tol"]
    assert all(rel_ok(x[i], spec["solution"][i], tol) for i in range(n)), (n, x)


def test_cholesky_decomp_random_spd(orc):
    """linalg/test_cholesky.c:137-169 with create_posdef_matrix (test_common.c:68-88): L L^T == A to 100 N eps,
    and the original matrix survives in the strict upper triangle (cholesky.c:103)."""
    rng = np.random.default_rng(7)
    for n in range(1, 51):
        m = rng.random((n, n))
        m = np.tril(m) + np.tril(m, -1).T + 10.0 * n * np.eye(n)
        st, v = orc.cholesky_decomp1(m)
        assert st == 0
        L = np.tril(v)
        rec = L @ L.T
        assert np.all(np.abs(rec - m) <= 100.0 * n * EPS * np.abs(m))
        assert np.array_equal(np.triu(v, 1), np.triu(m, 1))


def test_cholesky_rejects_indefinite(orc):
    st, _ = orc.cholesky_decomp1(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert st == 1                                                      # GSL_EDOM, cholesky.c:120-123


def test_linear_reproduction_and_structure(orc):
    """linear functions are reproduced to ~1e-15 inside the hull;
    N = 50 000 always gives 2N+1 leaves and N sub_{d+1} nodes."""
    n, m = 50000, 20000
    x = orc.synth_centres(n, 2)
    t = orc.Tree(2, n)
    assert t.init(x, flags=0, seed=0) == 0
    ty, _, _ = t.arrays()
    assert (ty == 0).sum() == 2 * n + 1 and (ty == 1).sum() == n
    assert abs(t.c.n_nodes / n - 9.0) < 0.1                             # "overhead = 9" (linear_simplex.c:63)
    y = orc.synth_targets(0, m, 2)
    g = 2 * x[:, 0] - 3 * x[:, 1] + 0.5
    t.reset_stats()
    vals, leaf = t.eval_many(x, g, y)
    assert np.abs(vals - (2 * y[:, 0] - 3 * y[:, 1] + 0.5)).max() < 5e-15
    assert t.c.stat_fallbacks == 0
    assert 55 < t.c.stat_tests / m < 75                                 # survey: ~65 containment tests / target


def test_small_trees_pass_reference_integrity_predicates(orc):
    for n in (60, 150, 400):
        x = orc.synth_centres(n, 2)
        t = orc.Tree(2, n)
        assert t.init(x, flags=0, seed=0) == 0
        assert t.check_leaf_nodes() == 1
        assert t.check_delaunay(x) == 1


# ---- solver breadth: pins for the oracle's decomp2 / rcond / LU_refine / pcholesky restatements
EPS = 2.2204460492503131e-16


def posdef(n, seed):
    """create_posdef_matrix of linalg/test_common.c:68-88: symmetric U(0,1) entries + 10 N on the diagonal"""
    rng = np.random.default_rng(seed)
    m = rng.random((n, n))
    a = np.tril(m) + np.tril(m, -1).T
    return a + 10.0 * n * np.eye(n)


def test_cholesky_rcond_hilbert_table(orc):
    spec = LINALG["hilbert_rcond"]
    for n, want in enumerate(spec["values"], start=1):
        if want <= spec["min_checked"]:
            continue
        st, llt = orc.cholesky_decomp1(hilbert(n))
        assert st == 0
        got = orc.cholesky_rcond(llt)
        assert abs(got - want) <= spec["rel_tol"] * want, (n, got, want)


@pytest.mark.parametrize("n", list(range(1, 13)) + [30, 50])
def test_cholesky_decomp2_reconstructs(orc, n):
    for a in ([hilbert(n)] if n <= 12 else []) + [posdef(n, n)]:
        st, v, s = orc.cholesky_decomp2(a)
        assert st == 0
        assert np.array_equal(s, 1.0 / np.sqrt(np.diag(a)))
        L = np.tril(v) / s[:, None]                              # L <- S^-1 L  (test_cholesky.c:86-98)
        rec = L @ L.T
        assert np.abs(rec - a).max() <= max(n, 4) * EPS * 100 * np.abs(a).max()     # :161-162 (N eps, entrywise relative)
        # the scaled matrix survives in the strict upper triangle (decomp1's transpose copy)
        scaled = a * np.outer(s, s)
        assert np.allclose(np.triu(v, 1), np.triu(scaled, 1), rtol=4 * EPS, atol=0)
        b = np.arange(1.0, n + 1.0)
        x = orc.cholesky_solve2(v, s, b)
        assert np.abs(a @ x - b).max() <= 1e-6 * max(1.0, np.abs(x).max()) * np.abs(a).max()


@pytest.mark.parametrize("n", [1, 2, 3, 7, 12, 33, 50])
def test_pcholesky_reconstructs_and_solves(orc, n):
    tol = LINALG["solver_breadth_tolerances"]
    mats = [posdef(n, 100 + n)] + ([hilbert(n)] if n <= 12 else [])
    for a in mats:
        st, ldlt, perm = orc.pcholesky_decomp(a)
        assert st == 0 and sorted(perm.tolist()) == list(range(n))
        L = np.tril(ldlt, -1) + np.eye(n)
        D = np.diag(np.diag(ldlt))
        pap = a[np.ix_(perm, perm)]                              # P A P^T
        assert np.abs(L @ D @ L.T - pap).max() <= tol["pcholesky_reconstruct_eps_mult_per_n"] * n * EPS * np.abs(a).max()
        d = np.diag(ldlt)
        assert (np.diff(d[: max(1, n)]) <= 1e-12 * np.abs(d).max()).all() or n > 12      # pivoting: D non-increasing (exactly so in exact arithmetic)
        assert np.array_equal(np.triu(ldlt, 1), np.triu(a, 1))   # original kept in the strict upper triangle
    a = posdef(n, 100 + n)
    rng = np.random.default_rng(n)
    sol = rng.random(n)
    rhs = a @ sol
    st, ldlt, perm = orc.pcholesky_decomp(a)
    x = orc.pcholesky_solve(ldlt, perm, rhs)
    assert np.abs(x - sol).max() <= tol["pcholesky_solve_eps_mult_per_n"] * n * EPS * max(1.0, np.abs(sol).max()) * 4
    if n <= 3:
        h = hilbert(n)
        rhs = h @ sol
        st, ldlt, perm = orc.pcholesky_decomp(h)
        x = orc.pcholesky_solve(ldlt, perm, rhs)
        assert np.abs(x - sol).max() <= tol["pcholesky_solve_hilbert_eps_mult_per_n"] * n * EPS * 4


def test_pcholesky_rcond_hilbert_table_and_decomp2(orc):
    """gsl_linalg_pcholesky_rcond is pinned by the same reference-held table as cholesky_rcond: the reference's own test
    (linalg/test_cholesky.c:675-687, 716-725) compares it with hilb_rcond to 1e-6 for the UNSCALED decomposition.
    decomp2 / svx2 (pcholesky.c:231-353): the unscaled matrix in the strict upper triangle, L D L^T = P S A S P^T to the
    reference's 1024 N eps, solutions to its 64 N eps (random) / 2048 N eps (Hilbert) (test_cholesky.c:712-713, 794-802)."""
    spec = LINALG["hilbert_rcond"]
    for n, want in enumerate(spec["values"], start=1):
        if want <= 1.0e-12:                                          # test_cholesky.c:719-720
            continue
        st, ldlt, perm = orc.pcholesky_decomp(hilbert(n))
        assert st == 0
        got = orc.pcholesky_rcond(ldlt, perm)
        assert abs(got - want) <= 1.0e-6 * want, (n, got, want)
    for n in (1, 2, 5, 12, 40):
        for a in [posdef(n, 300 + n)] + ([hilbert(n)] if n <= 12 else []):
            st, ldlt, perm, sc = orc.pcholesky_decomp2(a)
            assert st == 0 and np.array_equal(sc, 1.0 / np.sqrt(np.diag(a)))
            assert np.array_equal(np.triu(ldlt, 1), np.triu(a, 1))   # the UNSCALED original above the diagonal
            L = np.tril(ldlt, -1) + np.eye(n)
            D = np.diag(np.diag(ldlt))
            scaled = a * np.outer(sc, sc)
            assert np.abs(L @ D @ L.T - scaled[np.ix_(perm, perm)]).max() <= 1024.0 * n * EPS * np.abs(scaled).max()
            sol = np.random.default_rng(n).random(n)
            x = orc.pcholesky_solve2(ldlt, perm, sc, a @ sol)
            mult = 2048.0 if (n <= 12 and a is not None and np.allclose(a, hilbert(n))) else 64.0
            if n <= 5 or mult == 64.0:
                assert np.abs(x - sol).max() <= mult * n * EPS * 8 * max(1.0, np.linalg.cond(a) * 1e-3)


@pytest.mark.parametrize("n", [2, 3, 4, 12])
def test_lu_refine_keeps_the_known_answers(orc, n):
    """linalg/test.c:411-494: after LU_solve, LU_refine must still meet the Hilbert / Vandermonde tolerances."""
    b = np.arange(1.0, n + 1.0)
    for name, mat in (("hilbert", hilbert(n)), ("vandermonde", vandermonde(n))):
        spec = LINALG[name][str(n)]
        lu, perm, _ = orc.lu_decomp(mat)
        st, x = orc.lu_solve(lu, perm, b)
        st2, xr = orc.lu_refine(mat, lu, perm, b, x)
        assert st == 0 and st2 == 0
        tol = spec["lu_eps_mult"] * EPS if "lu_eps_mult" in spec else spec["lu_abs_tol"]
        assert all(rel_ok(xr[i], spec["solution"][i], tol) for i in range(n)), (name, n, xr)


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_wendland_kernel_properties(orc, dim):
    """The compactly supported kernel of the oracle (parity unpinned by any reference file: README:18-26 lists such
    kernels as future work): phi(0) = 1, exactly 0 from the support radius on, the closed form inside, and a kernel
    matrix the reference's Cholesky accepts (positive definite for dim <= 3) whose solve reproduces the data."""
    n = 150
    x = orc.synth_centres(n, dim)
    eps = 0.25 * n ** (1.0 / dim)
    phi = orc.rbf_fill(2, eps, x)
    r = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(axis=2))
    t = eps * r
    want = np.where(t < 1.0, (1.0 - t) ** 4 * (4.0 * t + 1.0), 0.0)
    assert np.allclose(phi, want, rtol=0, atol=1e-14) and (np.diag(phi) == 1.0).all()
    assert (phi[t >= 1.0 + 1e-12] == 0.0).all() and np.array_equal(phi, phi.T)
    assert np.linalg.eigvalsh(phi).min() > 0.0
    f = orc.synth_response(x)
    w = orc.rbf_solve(2, eps, x, f)
    assert np.abs(orc.rbf_eval(2, eps, x, w, x) - f).max() < 1e-10 * max(1.0, np.abs(f).max())


@pytest.mark.parametrize("dim,n", [(1, 40), (2, 150), (3, 120)])
def test_affine_thin_plate_spline_properties(orc, dim, n):
    """The affine-augmented thin-plate spline of the oracle (pivoted LU of the (n + d + 1) saddle system; no reference
    code: parity unpinned) by the properties that define it: it interpolates the data, its weights are orthogonal to
    the affine polynomials (P^T w = 0), and affine data are reproduced by the tail alone (w = 0)."""
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    w, c = orc.rbf_solve_affine(1, 0.0, x, f)
    assert np.abs(orc.rbf_eval_affine(1, 0.0, c, x, w, x) - f).max() < 1e-10
    P = np.hstack([np.ones((n, 1)), x])
    assert np.abs(P.T @ w).max() < 1e-9 * max(1.0, np.abs(w).max())
    coef = np.arange(1, dim + 2) * 0.5
    lin = P @ coef
    w2, c2 = orc.rbf_solve_affine(1, 0.0, x, lin)
    assert np.abs(w2).max() < 1e-9 and np.abs(c2 - coef).max() < 1e-10
    y = orc.synth_targets(0, 200, dim)
    assert np.abs(orc.rbf_eval_affine(1, 0.0, c2, x, w2, y) - (np.hstack([np.ones((200, 1)), y]) @ coef)).max() < 1e-10
