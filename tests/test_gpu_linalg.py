"""-m gpu: blocked fp64-MFMA Cholesky / LU and the triangular sweeps vs the reference-order
unblocked CPU oracle, plus the reference's own Hilbert / Vandermonde known answers."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import dev, ptr

pytestmark = pytest.mark.gpu
EPS = 2.2204460492503131e-16
LINALG = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_linalg_known_answers.json")))


def spd(n, seed):
    rng = np.random.default_rng(seed)
    m = rng.random((n, n))
    return np.tril(m) + np.tril(m, -1).T + 10.0 * n * np.eye(n)      # linalg/test_common.c:68-88


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 64, 100, 128, 160, 256, 257, 384, 1000, 1152, 2048, 3000])
def test_cholesky_decomp_and_solve(pkg, orc, n):
    a = spd(n, n)
    b = np.arange(1, n + 1, dtype=np.float64)
    ctx = pkg.HipContext.on_torch_stream(0)
    lda = n + (3 if n % 2 else 2)                                      # tda > size2 honoured
    d_a = torch.zeros((n, lda), dtype=torch.float64, device="cuda")
    d_a[:, :n] = dev(a)
    st, info = ctx.cholesky_decomp1(n, ptr(d_a), lda)
    assert st == 0 and info == 0
    got = d_a.cpu().numpy()[:, :n]
    st_o, want = orc.cholesky_decomp1(a)
    L, Lo = np.tril(got), np.tril(want)
    assert np.abs(L - Lo).max() <= 1e-12 * np.abs(Lo).max()
    assert np.array_equal(np.triu(got, 1), np.triu(a, 1))             # original kept above the diagonal
    rec = L @ L.T
    if n <= 64:
        assert np.all(np.abs(rec - a) <= 100.0 * n * EPS * np.abs(a))  # linalg/test_cholesky.c:59-135 (N <= 50 there)
    assert np.abs(rec - a).max() <= (10.0 + n / 20.0) * EPS * np.abs(a).max()   # norm-wise backward error (bound ~ n eps |A|)
    d_x = dev(b)
    ctx.cholesky_svx(n, ptr(d_a), lda, ptr(d_x))
    ctx.sync()
    x = d_x.cpu().numpy()
    xo = orc.cholesky_solve(want, b)
    assert np.abs(x - xo).max() <= 1e-11 * np.abs(xo).max()


def test_cholesky_rejects_indefinite(pkg):
    ctx = pkg.HipContext.on_torch_stream(0)
    n = 200
    a = spd(n, 1)
    a[150, 150] = -1.0
    d_a = dev(a)
    st, info = ctx.cholesky_decomp1(n, ptr(d_a), n)
    assert st == pkg.capi.GSL_EDOM and info == 151
    # inside a 128-wide panel (chol_diag128_kernel), second panel, third 32-step
    n = 512
    a = spd(n, 2)
    a[200, 200] = -1.0
    d_a = dev(a)
    st, info = ctx.cholesky_decomp1(n, ptr(d_a), n)
    assert st == pkg.capi.GSL_EDOM and info == 201


@pytest.mark.parametrize("n", [2, 3, 4, 12])
def test_hilbert_known_answers(pkg, n):
    """linalg/test.c:378-398 exact solutions with the reference's tolerances, on the GPU path."""
    i = np.arange(n)
    h = 1.0 / (i[:, None] + i[None, :] + 1.0)
    rhs = np.arange(1, n + 1, dtype=np.float64)
    spec = LINALG["hilbert"][str(n)]
    ctx = pkg.HipContext.on_torch_stream(0)

    def ok(x, tol):
        return all(x[k] == spec["solution"][k] or abs(x[k] - spec["solution"][k]) / abs(spec["solution"][k]) <= tol
                   for k in range(n))
    d_a, d_x = dev(h), dev(rhs)
    st, info = ctx.cholesky_decomp1(n, ptr(d_a), n)
    assert st == 0
    ctx.cholesky_svx(n, ptr(d_a), n, ptr(d_x)); ctx.sync()
    assert ok(d_x.cpu().numpy(), spec["chol_eps_mult"] * EPS if "chol_eps_mult" in spec else spec["chol_abs_tol"])
    d_a, d_x = dev(h), dev(rhs)
    d_p = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.lu_decomp(n, ptr(d_a), n, ptr(d_p))
    assert ctx.lu_svx(n, ptr(d_a), n, ptr(d_p), ptr(d_x)) == 0
    ctx.sync()
    assert ok(d_x.cpu().numpy(), spec["lu_eps_mult"] * EPS if "lu_eps_mult" in spec else spec["lu_abs_tol"])


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 33, 100, 255, 1024, 1100, 2000, 2048, 3000])    # > 1024 rows: cooperative 64-wide panels
def test_lu_decomp_and_solve(pkg, orc, n):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    b = rng.standard_normal(n)
    ctx = pkg.HipContext.on_torch_stream(0)
    lda = n + 2
    d_a = torch.zeros((n, lda), dtype=torch.float64, device="cuda")
    d_a[:, :n] = dev(a)
    d_p = torch.zeros(n, dtype=torch.int32, device="cuda")
    signum = ctx.lu_decomp(n, ptr(d_a), lda, ptr(d_p))
    lu, perm = d_a.cpu().numpy()[:, :n], d_p.cpu().numpy()
    lu_o, perm_o, sg_o = orc.lu_decomp(a)
    assert np.array_equal(perm, perm_o.astype(np.int64)) and signum == sg_o      # same pivot sequence
    Lm, U = np.tril(lu, -1) + np.eye(n), np.triu(lu)
    assert np.abs(Lm @ U - a[perm]).max() <= 1e-13 * n * np.abs(a).max()
    assert np.abs(lu - lu_o).max() <= 1e-9 * np.abs(lu_o).max()
    d_x = dev(b)
    assert ctx.lu_svx(n, ptr(d_a), lda, ptr(d_p), ptr(d_x)) == 0
    ctx.sync()
    st, xo = orc.lu_solve(lu_o, perm_o, b)
    x = d_x.cpu().numpy()
    assert np.abs(a @ x - b).max() <= 1e-10 * max(1.0, np.abs(x).max()) * n
    assert np.abs(x - xo).max() <= 1e-7 * np.abs(xo).max()


@pytest.mark.parametrize("case", ["ties", "zero_column", "nan_below", "nan_diagonal", "odd_lda"])
def test_lu_pivot_rule_edge_cases_on_tall_panels(pkg, orc, case):
    """The pivot rule of lu.c:82-105 on panels tall enough for the cooperative kernel (> 1024 rows): strict '>' (the FIRST row
    attaining the maximum wins a tie), a NaN below the diagonal never becomes the pivot, a NaN diagonal keeps itself, a zero
    pivot leaves its column untouched; an odd row stride takes the single-workgroup kernels.  Same permutation as the oracle,
    same finite entries."""
    n = 1600
    rng = np.random.default_rng(7)
    if case == "ties":
        a = rng.integers(-3, 4, size=(n, n)).astype(np.float64)          # many equal |a| per column
        a[np.arange(n), np.arange(n)] += 0.0
    else:
        a = rng.standard_normal((n, n))
    if case == "zero_column":
        a[:, 70] = 0.0; a[:, 1300] = 0.0
    if case == "nan_below":
        a[900, 5] = np.nan; a[1500, 1100] = np.nan
    if case == "nan_diagonal":
        a[0, 0] = np.nan
    lda = n + (1 if case == "odd_lda" else 0)
    ctx = pkg.HipContext.on_torch_stream(0)
    d_a = torch.zeros((n, lda), dtype=torch.float64, device="cuda")
    d_a[:, :n] = dev(a)
    d_p = torch.zeros(n, dtype=torch.int32, device="cuda")
    signum = ctx.lu_decomp(n, ptr(d_a), lda, ptr(d_p))
    lu, perm = d_a.cpu().numpy()[:, :n], d_p.cpu().numpy()
    lu_o, perm_o, sg_o = orc.lu_decomp(a)
    assert np.array_equal(perm, perm_o.astype(np.int64)) and signum == sg_o
    fin = np.isfinite(lu_o)
    assert np.array_equal(np.isfinite(lu), fin)
    if case in ("ties", "zero_column", "odd_lda"):
        scale = np.abs(lu_o[fin]).max()
        assert np.abs(lu[fin] - lu_o[fin]).max() <= 1e-8 * scale


def test_lu_more_than_sixteen_cooperating_workgroups(pkg):
    """n = 4400: the first panels are factored by 18 workgroups, so a wave polls its exchange slots in more than one batch.
    Checked against LAPACK's partial pivoting (scipy: same pivot rule away from ties) and by the residual of P A = L U;
    the oracle's unblocked sweep would take minutes at this size."""
    import scipy.linalg
    n = 4400
    rng = np.random.default_rng(4400)
    a = rng.standard_normal((n, n))
    ctx = pkg.HipContext.on_torch_stream(0)
    d_a = dev(a)
    d_p = torch.zeros(n, dtype=torch.int32, device="cuda")
    signum = ctx.lu_decomp(n, ptr(d_a), n, ptr(d_p))
    lu, perm = d_a.cpu().numpy(), d_p.cpu().numpy()
    lu_ref, piv = scipy.linalg.lu_factor(a)
    perm_ref = np.arange(n)
    for k, p in enumerate(piv):
        perm_ref[[k, p]] = perm_ref[[p, k]]
    assert np.array_equal(perm, perm_ref)
    assert signum == (-1) ** int(np.count_nonzero(piv != np.arange(n)))
    assert np.abs(lu - lu_ref).max() <= 1e-9 * np.abs(lu_ref).max()
    Lm, U = np.tril(lu, -1) + np.eye(n), np.triu(lu)
    assert np.abs(Lm @ U - a[perm]).max() <= 1e-13 * n * np.abs(a).max()


def test_lu_singular_reports_edom(pkg):
    ctx = pkg.HipContext.on_torch_stream(0)
    a = np.ones((4, 4))
    d_a, d_p, d_x = dev(a), torch.zeros(4, dtype=torch.int32, device="cuda"), dev(np.ones(4))
    ctx.lu_decomp(4, ptr(d_a), 4, ptr(d_p))
    assert ctx.lu_svx(4, ptr(d_a), 4, ptr(d_p), ptr(d_x)) == pkg.capi.GSL_EDOM     # lu.c:181-184


@pytest.mark.parametrize("m,n,k,kn,lower", [
    (4096, 4096, 64, 0, 0), (4096, 4096, 64, 0, 1),      # 1024 tiles: 256x128 8-wave direct-to-LDS kernel
    (6144, 2048, 80, 0, 1),                                # lower trapezoid (rows below the square part)
    (1024, 1024, 128, 0, 1), (1024, 768, 128, 0, 0),      # 128x128 4-wave direct-to-LDS kernel
    (4000, 64, 64, 0, 1), (777, 32, 32, 0, 1),            # single-shot small-K kernel, ragged rows
    (300, 200, 50, 0, 0), (300, 200, 50, 1, 0), (515, 515, 33, 0, 1),   # guarded register-staged kernel
    (1024, 1024, 512, 0, 1), (2304, 384, 1024, 0, 0),    # stream-K, 4-wave: tiles cut in 2 / 4 K-ranges
    (2048, 2048, 2048, 0, 1), (2048, 1024, 1536, 0, 0),  # stream-K, 8-wave: several contributors per tile
    (4224, 2048, 256, 0, 0), (4224, 2048, 192, 0, 1),    # stream-K, 128x128 tiles (>= 512 tiles, rows not a multiple of 256)
    (3072, 256, 256, 0, 1), (1920, 128, 128, 0, 1),      # stream-K, 64x64 tiles: panel updates of the recursion's low levels
    (2048, 2048, 2048, 1, 0), (1024, 1024, 512, 1, 0), (2304, 384, 1024, 1, 0), (4224, 2048, 256, 1, 0),   # B stored [k][n] (LU's
    (1152, 128, 64, 1, 0), (256, 256, 4096, 1, 0),                                                         # N.N updates): stream-K, [k][n] image
])
def test_gemm_building_block(pkg, m, n, k, kn, lower):
    """C -= A op(B) on fp64 MFMA vs numpy, for every kernel variant behind gsl_sinterp_hip_gemm_minus."""
    rng = np.random.default_rng(m + n + k)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n) if kn else (n, k))
    Cm = rng.standard_normal((m, n))
    ctx = pkg.HipContext.on_torch_stream(0)
    dA, dB, dC = dev(A), dev(B), dev(Cm)
    ctx.gemm_minus(m, n, k, ptr(dA), k, ptr(dB), B.shape[1], kn, ptr(dC), n, lower)
    ctx.sync()
    got = dC.cpu().numpy()
    want = Cm - (A @ B if kn else A @ B.T)
    if lower:
        rows, cols = np.indices((m, n))
        mask = cols <= rows
        assert np.abs(got - want)[mask].max() <= 1e-12 * k
        assert np.array_equal(got[~mask], Cm[~mask])          # strict upper part untouched
    else:
        assert np.abs(got - want).max() <= 1e-12 * k


@pytest.mark.gpu
def test_gemm_stream_k_is_reproducible(pkg):
    """The stream-K split points and the order partial tiles are added in are functions of the shape
    only: repeated launches (which also re-use the flag / partial buffers) give identical bits."""
    m, n, k = 2048, 2048, 1024
    rng = np.random.default_rng(5)
    A = rng.standard_normal((m, k)); Cm = rng.standard_normal((m, n))
    ctx = pkg.HipContext.on_torch_stream(0)
    dA = dev(A)
    outs = []
    for _ in range(3):
        dC = dev(Cm)
        ctx.gemm_minus(m, n, k, ptr(dA), k, ptr(dA), k, 0, ptr(dC), n, 1)
        ctx.sync()
        outs.append(dC.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    want = Cm - A @ A.T
    rows, cols = np.indices((m, n))
    assert np.abs(outs[0] - want)[cols <= rows].max() <= 1e-12 * k


@pytest.mark.gpu
@pytest.mark.parametrize("n", [16640, 4100])
def test_single_launch_sweeps_grid_stride_beyond_the_cu_count(pkg, n):
    """The dataflow sweep kernel is launched with min(blocks, #CUs) workgroups -- one per CU, so every workgroup
    of the launch is co-resident BY CONSTRUCTION (the launch never exceeds CUs x resident workgroups; that is the
    guarantee the spin-waits rest on) -- and takes the 64-row blocks in grid-stride order: with n = 16640 there are
    260 blocks for 256 workgroups, so four workgroups own two.  Property check at full size (no oracle run):
    (L L^T) x = b with a synthetic well-conditioned factor, residual computed with torch on the GPU."""
    g = torch.Generator(device="cuda").manual_seed(n)
    L = torch.rand((n, n), dtype=torch.float64, device="cuda", generator=g)
    L = torch.tril(L, -1) / n + torch.diag(1.0 + torch.rand(n, dtype=torch.float64, device="cuda", generator=g))
    xs = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    b = L @ (L.T @ xs)
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x = b.clone()
    ctx.cholesky_svx(n, ptr(L), n, ptr(d_x))            # raises on a non-zero status
    ctx.sync()
    assert float((d_x - xs).abs().max()) <= 1e-11 * float(xs.abs().max())


@pytest.mark.gpu
def test_graph_replays_are_correct(pkg, orc):
    """The factorisation and sweep launch sequences are captured into hipGraphs on first use and
    REPLAYED afterwards: repeated calls on one context (same buffers) must give the same answers --
    including the failure flag, which a replay has to reset -- bit for bit."""
    n = 1024
    a = spd(n, 3)
    bad = a.copy(); bad[700, 700] = -1.0
    b = np.cos(np.arange(n))
    ctx = pkg.HipContext.on_torch_stream(0)
    d_a = torch.empty((n, n), dtype=torch.float64, device="cuda")
    d_x = torch.empty(n, dtype=torch.float64, device="cuda")
    results = []
    for rep, mat in enumerate([a, bad, a, a, bad, a]):
        d_a.copy_(dev(mat))
        st, info = ctx.cholesky_decomp1(n, ptr(d_a), n)
        if mat is bad:
            assert st == pkg.capi.GSL_EDOM and info == 701
            continue
        assert st == 0 and info == 0
        d_x.copy_(dev(b))
        ctx.cholesky_svx(n, ptr(d_a), n, ptr(d_x))
        ctx.sync()
        results.append(d_x.cpu().numpy())
    xo = orc.cholesky_solve(orc.cholesky_decomp1(a)[1], b)
    assert np.abs(results[0] - xo).max() <= 1e-11 * np.abs(xo).max()
    for r in results[1:]:
        assert np.array_equal(r, results[0])
