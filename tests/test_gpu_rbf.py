"""-m gpu: RBF fill / solve / eval on the HIP path vs the CPU oracle (libm + reference-order
unblocked solvers + naive j-ascending sums).  fp64 tolerance: 1e-10 relative (north star);
the RBF kernels themselves are 'parity unpinned' by any reference test (README:18-26)."""
import numpy as np
import pytest
import torch

from gpu_util import bits, dev, ptr

pytestmark = pytest.mark.gpu
TOL = 1e-10


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("kind,dim,n", [(0, 2, 512), (0, 3, 700), (1, 2, 512), (0, 1, 130), (1, 3, 333)])
def test_fill_matches_oracle(pkg, orc, kind, dim, n):
    x = orc.synth_centres(n, dim)
    eps = orc.gaussian_eps(n, dim)
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x = dev(x)
    phi = torch.empty((n, n), dtype=torch.float64, device="cuda")
    ctx.rbf_fill(kind, eps, ptr(d_x), n, dim, dim, ptr(phi), n)
    got = phi.cpu().numpy()
    want = orc.rbf_fill(kind, eps, x)
    # O(1) entries; r^2 is accumulated with FMA on the device, so a few ulp (not bit-equal to libm)
    assert np.abs(got - want).max() <= 2e-15 * max(1.0, np.abs(want).max())
    assert np.array_equal(got, got.T)                                              # bitwise symmetric
    if kind == 1:
        assert (np.diag(got) == 0).all()                                           # phi(0) = 0


@pytest.mark.parametrize("kind,dim,n,m", [(0, 2, 512, 10000), (1, 2, 512, 10000), (0, 3, 1000, 5000), (0, 2, 2048, 3000),
                                          (1, 2, 1500, 3000)])
def test_facade_rbf_matches_oracle(pkg, orc, kind, dim, n, m):
    """C1 (N=512 Gaussian, M=10^4) and friends through gsl_sinterp alloc/init/eval_many."""
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    y = orc.synth_targets(0, m, dim)
    eps = orc.gaussian_eps(n, dim)
    s = pkg.Sinterp("gaussian" if kind == 0 else "tps", dim, n, 0)
    assert s.init(x, f) == 0
    st, got, _ = s.eval_many(y)
    assert st == 0
    w = orc.rbf_solve(kind, eps, x, f)
    want = orc.rbf_eval(kind, eps, x, w, y)
    assert relerr(got, want) < TOL
    st, gw = s.weights()
    assert st == 0
    if kind == 0:
        assert relerr(gw, w) < TOL                      # Gaussian weights too (kappa <~ 2e4)
    # the interpolant reproduces the data at the centres
    st, at_centres, _ = s.eval_many(x)
    assert relerr(at_centres, f) < 1e-9
    # single-point entry agrees with the batch
    st1, v1 = s.eval_e(y[0])
    assert st1 == 0 and abs(v1 - got[0]) <= 1e-13 * max(1.0, abs(got[0]))


def test_eval_linearity_and_layout(pkg, orc):
    """size-independent properties at a larger M: linear in the weights; ragged strides honoured."""
    n, m, dim = 1024, 300000, 2
    x = orc.synth_centres(n, dim)
    eps = orc.gaussian_eps(n, dim)
    rng = np.random.default_rng(3)
    w1, w2 = rng.standard_normal(n), rng.standard_normal(n)
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x = dev(x)
    d_y = torch.empty((m, 3), dtype=torch.float64, device="cuda")          # ytda = 3 > dim
    ctx.synth_unit(0xC0FFEE02, 0, 0.02, 0.96, ptr(d_y), 3 * m)
    outs = []
    for w in (w1, w2, 2.0 * w1 - 0.5 * w2):
        d_w = dev(w)
        d_s = torch.empty(m, dtype=torch.float64, device="cuda")
        ctx.rbf_eval(0, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_y), m, 3, ptr(d_s))
        ctx.sync()
        outs.append(d_s.cpu().numpy())
    assert np.abs(outs[2] - (2.0 * outs[0] - 0.5 * outs[1])).max() < 1e-11 * np.abs(outs[2]).max()
    yh = d_y.cpu().numpy()[:2000, :2]
    want = orc.rbf_eval(0, eps, x, w1, np.ascontiguousarray(yh))
    assert relerr(outs[0][:2000], want) < TOL


def test_facade_errors(pkg, orc):
    n = 64
    x = orc.synth_centres(n, 2)
    x[5] = x[4]                                          # duplicate centre: Gaussian matrix singular
    s = pkg.Sinterp("gaussian", 2, n, 0)
    st = s.init(x, orc.synth_response(x))
    assert st == pkg.capi.GSL_EDOM                       # cholesky.c:120-123 semantics
    s2 = pkg.Sinterp("gaussian", 2, n, 0)
    x = orc.synth_centres(n, 2)
    assert s2.init(x, orc.synth_response(x)) == 0
    st, vals, _ = s2.eval_many(np.zeros((0, 2)))
    assert st == 0 and len(vals) == 0


@pytest.mark.parametrize("dim,n", [(2, 900), (3, 700), (1, 300), (2, 2500)])
def test_tps_solver_routes_agree(pkg, orc, dim, n, monkeypatch):
    """TPS init: the shifted-SPD Cholesky + Woodbury route (2) and the pivoted-LU reference
    route (3, forced) give the same interpolant: values within 1e-10 of each other and of the
    CPU oracle (reference-order LU)."""
    x = orc.synth_centres(n, dim) * 2.5 - 7.0              # non-unit box: exercises the centring of P
    f = orc.synth_response(orc.synth_centres(n, dim))
    y = orc.synth_targets(0, 4000, dim) * 2.5 - 7.0
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x, d_y = dev(x), dev(y)
    res = {}
    for force in ("0", "1"):
        monkeypatch.setenv("GSL_SINTERP_FORCE_LU", force)
        d_w = dev(f)
        d_phi = torch.empty((n, n), dtype=torch.float64, device="cuda")
        st, route = ctx.rbf_solve(1, 0.0, ptr(d_x), n, dim, dim, ptr(d_phi), n, ptr(d_w))
        assert st == 0 and route == (3 if force == "1" else 2)
        d_s = torch.empty(len(y), dtype=torch.float64, device="cuda")
        ctx.rbf_eval(1, 0.0, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_y), len(y), dim, ptr(d_s))
        ctx.sync()
        res[force] = (d_s.cpu().numpy(), d_w.cpu().numpy())
    want = orc.rbf_eval(1, 0.0, x, orc.rbf_solve(1, 0.0, x, f), y)
    assert relerr(res["0"][0], res["1"][0]) < TOL
    assert relerr(res["0"][0], want) < TOL and relerr(res["1"][0], want) < TOL
    # residual of the SPD-route weights against the true (unshifted) system
    phi = orc.rbf_fill(1, 0.0, x)
    assert np.abs(phi @ res["0"][1] - f).max() < 1e-9 * max(1.0, np.abs(f).max())


@pytest.mark.parametrize("dim,n,m,eps_scale", [(3, 3000, 6000, 1.0), (2, 4096, 5000, 1.0), (1, 1500, 4100, 1.0),
                                               (3, 1100, 4500, 0.05), (2, 2000, 4200, 4.0)])
def test_gaussian_tile_culling_matches_oracle(pkg, orc, dim, n, m, eps_scale):
    """The Gaussian sweep skips whole tiles of (Morton-ordered) centres whose bounding box is beyond
    the 2^-72 cut-off of a workgroup's targets: same values as the oracle's full j-ascending sums
    (<= 1e-10), for ragged tile counts, for wide kernels (nothing culled, eps_scale << 1), narrow ones,
    and for targets far outside the cloud (everything culled -> exactly 0)."""
    x = orc.synth_centres(n, dim)
    eps = orc.gaussian_eps(n, dim) * eps_scale
    rng = np.random.default_rng(n + m)
    w = rng.standard_normal(n)
    y = orc.synth_targets(0, m, dim)
    y[-50:] += 40.0                                      # far away: every term < 2^-72
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x, d_w, d_y = dev(x), dev(w), dev(y)
    d_s = torch.full((m,), 7.0, dtype=torch.float64, device="cuda")
    ctx.rbf_eval(0, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_y), m, dim, ptr(d_s))
    ctx.sync()
    got = d_s.cpu().numpy()
    want = orc.rbf_eval(0, eps, x, w, y)
    assert relerr(got, want) < TOL
    assert (got[-50:] == 0.0).all()
    # run to run: the centre (summation) order is fixed and a target takes exactly the terms above the
    # 2^-72 cut-off of ITS OWN distance (not its wave-mates'), so the result does not depend on the
    # grouping of the targets, which comes from an atomic scatter and differs between runs: bit-identical
    d_s2 = torch.empty(m, dtype=torch.float64, device="cuda")
    ctx.rbf_eval(0, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_y), m, dim, ptr(d_s2))
    ctx.sync()
    assert np.array_equal(got, d_s2.cpu().numpy())
    # and independent of the ORDER of the targets: a shuffled batch gives the same value per target
    p = rng.permutation(m)
    d_yp = dev(y[p])
    ctx.rbf_eval(0, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_yp), m, dim, ptr(d_s2))
    ctx.sync()
    assert np.array_equal(got[p], d_s2.cpu().numpy())


@pytest.mark.parametrize("m", [300, 9000])
def test_gaussian_nan_and_inf_targets_propagate(pkg, orc, m):
    """A NaN coordinate makes every r^2 NaN: the naive sum (oracle) is NaN, and so must the sweeps be
    (plain kernel for small batches, culled kernel for sorted ones); an infinite coordinate gives 0."""
    n, dim = 2048, 2
    x = orc.synth_centres(n, dim)
    eps = orc.gaussian_eps(n, dim)
    w = np.random.default_rng(1).standard_normal(n)
    y = orc.synth_targets(0, m, dim)
    y[5, 0] = np.nan; y[77, 1] = np.nan; y[100] = np.nan
    y[200, 0] = np.inf; y[201] = [-np.inf, np.inf]
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x, d_w, d_y = dev(x), dev(w), dev(y)
    d_s = torch.full((m,), 7.0, dtype=torch.float64, device="cuda")
    ctx.rbf_eval(0, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_y), m, dim, ptr(d_s))
    ctx.sync()
    got = d_s.cpu().numpy()
    want = orc.rbf_eval(0, eps, x, w, y)
    assert np.isnan(want[[5, 77, 100]]).all() and (want[[200, 201]] == 0).all()      # what the oracle does
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert relerr(got[ok], want[ok]) < TOL and (got[[200, 201]] == 0).all()


@pytest.mark.parametrize("cfg", ["C2", "C3"])
def test_full_size_properties(pkg, orc, cfg):
    """BASELINE.json's full sizes, checked through size-independent properties (an oracle run at these
    sizes takes minutes to hours): the interpolant reproduces the data at its centres; init is linear
    in the response; a sample of rows of L L^T matches Phi (C3, Cholesky route)."""
    kind, dim, n = (1, 2, 4096) if cfg == "C2" else (0, 3, 16384)
    eps = 2.0 * n ** (1.0 / dim)
    ctx = pkg.HipContext.on_torch_stream(0)
    f64 = torch.float64
    d_x = torch.empty((n, dim), dtype=f64, device="cuda")
    ctx.synth_unit(0xC0FFEE01, 0, 0.0, 1.0, ptr(d_x), n * dim)
    d_f1 = sum(torch.sin(3.0 * (c + 1) * d_x[:, c]) for c in range(dim))
    d_f2 = torch.cos(2.0 * d_x[:, 0]) + 0.5
    d_phi = torch.empty((n, n), dtype=f64, device="cuda")
    ws = []
    for d_f in (d_f1, d_f2, 2.0 * d_f1 - 0.25 * d_f2):
        d_w = d_f.clone()
        st, route = ctx.rbf_solve(kind, eps, ptr(d_x), n, dim, dim, ptr(d_phi), n, ptr(d_w))
        assert st == 0 and route == (2 if kind == 1 else 1)
        d_s = torch.empty(n, dtype=f64, device="cuda")
        ctx.rbf_eval(kind, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_x), n, dim, ptr(d_s))
        ctx.sync()
        err = float((d_s - d_f).abs().max()) / float(d_f.abs().max())
        print(cfg, "reproduction error", err, "max|w|", float(d_w.abs().max()))
        assert err <= 1e-6                                 # s(x_i) = f_i up to cond(Phi) eps (|w| reaches 1e5..1e7 here)
        ws.append(d_w)
    lin = 2.0 * ws[0] - 0.25 * ws[1]
    assert float((ws[2] - lin).abs().max()) <= 1e-9 * float(lin.abs().max())           # init is linear in f
    if kind == 0:
        # d_phi now holds L (lower, diagonal included) and the original matrix strictly above the diagonal
        rows = torch.tensor([0, 1, 127, 128, 129, 4095, 8191, 8192, 12345, n - 1], device="cuda")
        L = torch.tril(d_phi)
        rec = L[rows] @ L.T                                                              # rows of L L^T
        phi_rows = torch.empty((len(rows), n), dtype=f64, device="cuda")
        d2 = ((d_x[rows][:, None, :] - d_x[None, :, :]) ** 2).sum(-1)
        phi_rows = torch.exp(-(eps * eps) * d2)
        assert float((rec - phi_rows).abs().max()) <= 1e-12


@pytest.mark.parametrize("kind,dim,n", [(1, 2, 1500), (0, 3, 1300)])
def test_repeated_init_on_one_context(pkg, orc, kind, dim, n):
    """bench.py's loop: init (fill + factorisation + solves, replayed hipGraphs from the second call
    on) several times on one context and one set of buffers; every repetition must return the first
    call's weights bit for bit, for alternating right-hand sides too."""
    ctx = pkg.HipContext.on_torch_stream(0)
    x = orc.synth_centres(n, dim)
    f1 = orc.synth_response(x)
    f2 = np.cos(2.0 * x[:, 0]) + 0.5
    eps = orc.gaussian_eps(n, dim)
    d_x = dev(x)
    d_phi = torch.empty((n, n), dtype=torch.float64, device="cuda")
    d_w = torch.empty(n, dtype=torch.float64, device="cuda")
    first = {}
    for rep, f in enumerate([f1, f2, f1, f2, f2, f1]):
        d_w.copy_(dev(f))
        st, route = ctx.rbf_solve(kind, eps, ptr(d_x), n, dim, dim, ptr(d_phi), n, ptr(d_w))
        assert st == 0
        ctx.sync()
        w = d_w.cpu().numpy()
        key = id(f)
        if key in first:
            assert np.array_equal(w, first[key]), rep
        else:
            first[key] = w
            want = orc.rbf_solve(kind, eps, x, f)
            y = orc.synth_targets(0, 500, dim)
            d_s = torch.empty(500, dtype=torch.float64, device="cuda")
            ctx.rbf_eval(kind, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(dev(y)), 500, dim, ptr(d_s))
            ctx.sync()
            assert relerr(d_s.cpu().numpy(), orc.rbf_eval(kind, eps, x, want, y)) < TOL


# ---- compactly supported kernel (Wendland C2; the reference's README:18-26 future list) ------------------------
@pytest.mark.parametrize("dim,n", [(2, 600), (3, 700), (1, 130)])
def test_wendland_fill_matches_oracle(pkg, orc, dim, n):
    x = orc.synth_centres(n, dim)
    eps = 0.125 * n ** (1.0 / dim)
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x = dev(x)
    phi = torch.empty((n, n), dtype=torch.float64, device="cuda")
    ctx.rbf_fill(pkg.capi.RBF_WENDLAND, eps, ptr(d_x), n, dim, dim, ptr(phi), n)
    got = phi.cpu().numpy()
    want = orc.rbf_fill(2, eps, x)
    assert np.abs(got - want).max() <= 4e-15
    assert np.array_equal(got, got.T) and (np.diag(got) == 1.0).all()
    assert np.array_equal(got == 0.0, want == 0.0)                   # the support is cut at exactly the same pairs


@pytest.mark.parametrize("dim,n,m", [(2, 900, 6000), (3, 1500, 5000), (2, 4096, 20000), (1, 300, 700)])
def test_wendland_facade_matches_oracle(pkg, orc, dim, n, m):
    """alloc / init (Cholesky route) / eval through the facade; n >= 1024 takes the culled sweep, whose culling is
    exact for a compactly supported kernel (dropped terms are 0), the others the plain one."""
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    y = np.ascontiguousarray(np.concatenate([orc.synth_targets(0, m - 4, dim), x[:2], 3.0 + orc.synth_targets(1, 2, dim)]))
    eps = 0.125 * n ** (1.0 / dim)                                   # the facade's default shape for this kernel
    s = pkg.Sinterp("wendland", dim, n, 0)
    assert s.init(x, f) == 0 and s.route() == 1
    st, got, _ = s.eval_many(y)
    assert st == 0
    w = orc.rbf_solve(2, eps, x, f)
    want = orc.rbf_eval(2, eps, x, w, y)
    assert relerr(got, want) < TOL
    assert (got[-2:] == 0.0).all()                                   # targets outside every support: exactly 0
    st, gw = s.weights()
    assert st == 0 and relerr(gw, w) < 1e-8
    st, at_centres, _ = s.eval_many(x)
    assert relerr(at_centres, f) < 1e-9
    # bit-reproducible, and independent of how the batch is split
    st, again, _ = s.eval_many(y)
    assert np.array_equal(got, again)
    st, part, _ = s.eval_many(np.ascontiguousarray(y[: m // 3]))
    assert np.array_equal(part, got[: m // 3])
    # NaN targets propagate
    st, nanv, _ = s.eval_many(np.full((1, dim), np.nan))
    assert np.isnan(nanv[0])


def test_wendland_checkpoint_round_trip(pkg, orc, tmp_path):
    n, dim = 500, 2
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    y = orc.synth_targets(2, 1000, dim)
    a = pkg.Sinterp("wendland", dim, n, 0)
    assert a.init(x, f) == 0
    st, va, _ = a.eval_many(y)
    path = str(tmp_path / "wendland.bin")
    assert a.fwrite(path) == 0
    b = pkg.Sinterp("wendland", dim, n, 0)
    assert b.fread(path) == 0
    st, vb, _ = b.eval_many(y)
    assert st == 0 and np.array_equal(va, vb)
    c = pkg.Sinterp("gaussian", dim, n, 0)                           # a checkpoint of another kernel is refused
    assert c.fread(path) != 0


@pytest.mark.parametrize("dim", [2, 3])
def test_clustered_centres_with_outlier_are_bit_reproducible(pkg, orc, dim):
    """a far outlier stretches the bounding box of the centre sort, so thousands of centres share one
    Morton cell.  The sweep must still sum every cell in ORIGINAL index order (cell_rank_kernel ranks a cell of any
    size): two runs, two contexts (each sorts its own centres, like the members of a device group) and a shuffled
    target batch all give the same bits, and the values match the oracle's naive sums."""
    n, m = 6000, 20000
    rng = np.random.default_rng(77 + dim)
    x = 0.5 + 0.01 * rng.random((n, dim))                  # one tight cluster ...
    x[-1] = 400.0                                          # ... and one outlier: > 4000 centres per cell
    w = rng.standard_normal(n)
    y = 0.5 + 0.01 * rng.random((m, dim))
    eps = 2.0 * (n ** (1.0 / dim)) / 0.01 * 0.25           # a few hundred centres inside the cut-off radius
    d_x, d_w, d_y = dev(x), dev(w), dev(y)
    outs = []
    for ctx in (pkg.HipContext.on_torch_stream(0), pkg.HipContext.on_torch_stream(0)):
        for rep in range(2):
            d_s = torch.empty(m, dtype=torch.float64, device="cuda")
            ctx.rbf_eval(0, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_y), m, dim, ptr(d_s))
            ctx.sync()
            outs.append(d_s.cpu().numpy())
    for o in outs[1:]:
        assert np.array_equal(bits(outs[0]), bits(o))
    want = orc.rbf_eval(0, eps, x, w, y)
    assert relerr(outs[0], want) < TOL


def test_model_cached_sweep_equals_uncached(pkg, orc):
    """gsl_sinterp_hip_rbf_eval_model: with a model id the packed centres are kept between calls; the values are
    the bits of the uncached sweep, a new id (or new pointers) repacks, id 0 never caches."""
    n, dim, m = 5000, 2, 30000
    x = orc.synth_centres(n, dim)
    eps = orc.gaussian_eps(n, dim)
    rng = np.random.default_rng(5)
    w1, w2 = rng.standard_normal(n), rng.standard_normal(n)
    y = orc.synth_targets(0, m, dim)
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x, d_w, d_y = dev(x), dev(w1), dev(y)

    def sweep(mid, mm=m):
        d_s = torch.empty(mm, dtype=torch.float64, device="cuda")
        ctx.rbf_eval(0, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_y), mm, dim, ptr(d_s), model_id=mid)
        ctx.sync()
        return d_s.cpu().numpy()
    base = sweep(0)
    assert np.array_equal(bits(base), bits(sweep(11))) and np.array_equal(bits(base), bits(sweep(11)))
    assert np.array_equal(bits(base[:1]), bits(sweep(11, 1)))            # the single-point call reuses the cache
    d_w.copy_(dev(w2))                                                   # same pointers, new content -> new id
    want2 = orc.rbf_eval(0, eps, x, w2, y)
    assert relerr(sweep(12), want2) < TOL
    assert relerr(sweep(0), want2) < TOL


@pytest.mark.parametrize("kind,dim,n", [(0, 2, 3000), (0, 3, 2500), (2, 2, 3000), (0, 1, 1500)])
def test_large_batch_reorder_route_returns_the_bits_of_the_permutation_route(pkg, orc, monkeypatch, kind, dim, n):
    """Round 3: batches of >= 2^18 targets of the local kernels are physically put in cell order by the two-level reorder
    (sort.hip), swept contiguously, stored through the order's map and gathered back; smaller batches (and
    GSL_SINTERP_SORT_LEVELS=1) keep the permutation route.  A target's value depends on the model and the target only:
    both routes must agree bit for bit, strided targets (ytda > dim), NaN and far targets included; a sample against
    the oracle's naive sums."""
    m, ytda = 300_000, dim + 2
    x = orc.synth_centres(n, dim)
    eps = orc.gaussian_eps(n, dim) if kind == 0 else 0.125 * n ** (1.0 / dim)   # Wendland: support radius 1 / eps
    rng = np.random.default_rng(n + dim)
    w = rng.standard_normal(n)
    wide = np.full((m, ytda), 1e30)
    wide[:, :dim] = rng.random((m, dim))
    wide[-40:, :dim] += 40.0
    wide[1234, 0] = np.nan
    ctx = pkg.HipContext.on_torch_stream(0)
    d_x, d_w, d_y = dev(x), dev(w), dev(wide)
    out = {}
    for levels in ("1", "2"):
        monkeypatch.setenv("GSL_SINTERP_SORT_LEVELS", levels)
        d_s = torch.full((m,), 7.0, dtype=torch.float64, device="cuda")
        ctx.rbf_eval(kind, eps, ptr(d_x), n, dim, dim, ptr(d_w), ptr(d_y), m, ytda, ptr(d_s))
        ctx.sync()
        out[levels] = d_s.cpu().numpy()
    monkeypatch.delenv("GSL_SINTERP_SORT_LEVELS")
    assert np.array_equal(bits(out["1"]), bits(out["2"]))
    assert np.isnan(out["2"][1234]) and (out["2"][-40:] == 0.0).all()
    idx = np.setdiff1d(np.arange(0, m, 131), [1234])
    want = orc.rbf_eval(kind, eps, x, w, np.ascontiguousarray(wide[idx, :dim]))
    assert relerr(out["2"][idx], want) < TOL
