"""-m gpu: gridded front-end and checkpoints of the gsl_sinterp facade.

eval_grid reproduces the 100 x 100 sweep of interpolation/scattered_interp_example.c:175-217 on the
reference's 50-station dataset with the example's own min / max: node coordinates are generated on the
device with the host loop's operations, so every grid value equals the per-point host API
(find_leaf + interp_point, bit-identical to the oracle) bit for bit; fprintf_grid writes the plot.dat
text.  fwrite / fread: round-trip bit equality of evaluations, nothing re-solved."""
import time

import numpy as np
import pytest

from gpu_util import bits

pytestmark = pytest.mark.gpu

EX_MIN, EX_MAX = [-89.6763, 40.9479], [-86.303, 43.20]          # scattered_interp_example.c:166-167


def test_eval_grid_reproduces_the_example_sweep(pkg, orc, weather, tmp_path):
    data = np.ascontiguousarray(weather[:, :2])
    resp = np.ascontiguousarray(weather[:, 2])
    s = pkg.Sinterp("linear_simplex", 2, 50, 0)
    assert s.set_tree_options(0, pkg.capi.Rng(0)) == 0               # example.c:169: init(data, NULL, NULL, 0, rng)
    assert s.init(data, resp) == 0
    n_grid = 100
    st, grid = s.eval_grid(EX_MIN, EX_MAX, n_grid, n_grid)
    assert st == 0
    o = orc.Tree(2, 50)
    assert o.init(data, flags=0, seed=0) == 0
    xstep = (EX_MAX[0] - EX_MIN[0]) / n_grid
    ystep = (EX_MAX[1] - EX_MIN[1]) / n_grid
    nodes = np.array([[EX_MIN[0] + xstep * i, EX_MIN[1] + ystep * j] for i in range(n_grid) for j in range(n_grid)])
    want, _ = o.eval_many(data, resp, nodes)
    assert np.array_equal(bits(grid.reshape(-1)), bits(want))
    # inside the hull the interpolant is bounded by the data; outside it extrapolates along the hull edges (quirk q6)
    assert np.isfinite(grid).all() and resp.min() - 15.0 < grid.min() and grid.max() < resp.max() + 15.0
    # plot.dat text: "%g %g %g" per node, blank line per i (example.c:203-215)
    import ctypes as C
    path = tmp_path / "plot.dat"
    with pkg.capi.CFile(path, "w") as fp:
        assert pkg.lib().gsl_sinterp_fprintf_grid(fp, C.byref(pkg.capi.as_vector(np.array(EX_MIN))),
                                                  C.byref(pkg.capi.as_vector(np.array(EX_MAX))), C.byref(pkg.capi.as_matrix(grid))) == 0
    text = path.read_text()
    expect = "".join("".join("%g %g %g\n" % (EX_MIN[0] + xstep * i, EX_MIN[1] + ystep * j, grid[i, j]) for j in range(n_grid)) + "\n"
                     for i in range(n_grid))
    assert text == expect
    # ragged grid into a view of a wider matrix; RBF type through the same entry
    r = pkg.Sinterp("gaussian", 2, 50, 0)
    assert r.set_shape(1.5) == 0 and r.init(data, resp) == 0
    st, g2 = r.eval_grid(EX_MIN, EX_MAX, 37, 11)
    st2, direct, _ = r.eval_many(np.array([[EX_MIN[0] + (EX_MAX[0] - EX_MIN[0]) / 37 * i, EX_MIN[1] + (EX_MAX[1] - EX_MIN[1]) / 11 * j]
                                           for i in range(37) for j in range(11)]))
    assert st == 0 and st2 == 0 and np.array_equal(bits(g2.reshape(-1)), bits(direct))
    # a grid that leaves the cage: reported (EDOM), NaN at those nodes
    st, g3 = s.eval_grid([-1e9, 40.0], [1e9, 44.0], 8, 8)
    assert st == pkg.capi.GSL_EDOM and np.isnan(g3).any() and not np.isnan(g3).all()


@pytest.mark.parametrize("kind,dim,n", [("gaussian", 2, 1200), ("tps", 2, 700), ("gaussian", 3, 900), ("linear_simplex", 2, 4000)])
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_checkpoint_round_trip_bit_equal(pkg, orc, tmp_path, kind, dim, n, devices):
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    y = orc.synth_targets(0, 9001, dim)
    a = pkg.Sinterp(kind, dim, n, 0)
    if kind == "linear_simplex":
        assert a.set_tree_options(0, pkg.capi.Rng(0)) == 0
    assert a.init(x, f) == 0
    st, want, wl = a.eval_many(y, want_leaf=True)
    assert st == 0
    path = tmp_path / "interp.bin"
    assert a.fwrite(path) == 0
    b = pkg.Sinterp(kind, dim, n, 0)
    assert b.set_device_list(devices) == 0
    t0 = time.time()
    assert b.fread(path) == 0                                     # no solve, no triangulation
    print(f"{kind} N={n}: restored in {time.time() - t0:.3f} s")
    st, got, gl = b.eval_many(y, want_leaf=True)
    assert st == 0 and np.array_equal(bits(got), bits(want)) and np.array_equal(gl, wl)
    if kind != "linear_simplex":
        assert np.array_equal(a.weights()[1], b.weights()[1])
    # writing the restored object gives the same bytes
    path2 = tmp_path / "interp2.bin"
    assert b.fwrite(path2) == 0 and path.read_bytes() == path2.read_bytes()
    # mismatched size / type: refused
    c = pkg.Sinterp(kind, dim, n + 1, 0)
    assert c.fread(path) == pkg.capi.GSL_EBADLEN
    other = pkg.Sinterp("tps" if kind != "tps" else "gaussian", dim, n, 0)
    assert other.fread(path) == pkg.capi.GSL_EBADLEN
    (tmp_path / "short.bin").write_bytes(path.read_bytes()[:200])
    assert pkg.Sinterp(kind, dim, n, 0).fread(tmp_path / "short.bin") == pkg.capi.GSL_EFAILED
    # an uninitialised interpolant cannot be written
    assert pkg.Sinterp(kind, dim, n, 0).fwrite(tmp_path / "none.bin") == pkg.capi.GSL_EINVAL


def test_check_delaunay_reference_entry_runs_on_device(pkg, orc):
    import ctypes as C
    x = orc.synth_centres(2000, 2)
    t = pkg.SimplexTree(2, 2000)
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    assert pkg.lib().check_delaunay(t._t, t._m()) == 1


def test_host_batches_pipelined_over_the_copy_pipe_are_bit_identical(pkg, orc):
    """Round 4: a host batch on ONE device is cut into chunks whose H2D | sweep | D2H overlap (copy pipe; dense caller
    arrays travel as they are, strided ones through pinned staging).  A value depends on (model, target) only, so the
    result must equal the resident one-shot sweep bit for bit -- for the barycentric and the RBF types, dense and
    strided arguments, with and without the leaf output, and the EDOM verdict must survive the chunking."""
    import torch
    n, m = 3000, 1_200_000                                        # >= 2 chunks of 2^19
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    y = orc.synth_targets(0, m, 2)
    bits = lambda a: a.view(np.uint64)
    # --- barycentric
    t = pkg.SimplexTree(2, n)
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    d = t.device_alloc(0)
    assert d.set_response(f) == 0
    ty = torch.from_numpy(y).cuda()
    tv = torch.empty(m, dtype=torch.float64, device="cuda")
    tl = torch.empty(m, dtype=torch.int32, device="cuda")
    assert d.eval_resident(ty.data_ptr(), m, 2, tv.data_ptr(), tl.data_ptr()) == 0
    torch.cuda.synchronize()
    rv, rl = tv.cpu().numpy(), tl.cpu().numpy()
    st, v, l = d.eval_many(y)
    assert st == 0 and np.array_equal(bits(v), bits(rv)) and np.array_equal(l, rl)
    wide = np.zeros((m, 3)); wide[:, :2] = y                     # tda = 3: the staged route
    vs = np.zeros(2 * m)
    st, v2, l2 = d.eval_many(wide[:, :2], out=(vs[::2], np.empty(m, dtype=np.int32)))
    assert st == 0 and np.array_equal(bits(np.ascontiguousarray(v2)), bits(rv)) and np.array_equal(l2, rl)
    st, v3, _ = d.eval_many(y, want_leaf=False)
    assert st == 0 and np.array_equal(bits(v3), bits(rv))
    yo = y.copy(); yo[m - 5] = [1e9, 1e9]                         # one target outside the cage, in the last chunk
    st, v4, l4 = d.eval_many(yo)
    assert st == pkg.GSL_EDOM and l4[m - 5] == -1 and np.isnan(v4[m - 5]) and np.array_equal(l4[:m - 5], rl[:m - 5])
    st, v5, _ = d.eval_many(yo, want_leaf=False)                  # the verdict does not need the indices on the host
    assert st == pkg.GSL_EDOM
    # --- Gaussian RBF through the facade
    s = pkg.Sinterp("gaussian", 2, n, 0)
    assert s.init(x, f) == 0
    ts = torch.empty(m, dtype=torch.float64, device="cuda")
    assert s.eval_resident(ty.data_ptr(), m, 2, ts.data_ptr()) == 0
    torch.cuda.synchronize()
    rs = ts.cpu().numpy()
    st, g, _ = s.eval_many(y)
    assert st == 0 and np.array_equal(bits(g), bits(rs))
    st, g2, _ = s.eval_many(wide[:, :2], out=vs[::2])
    assert st == 0 and np.array_equal(bits(np.ascontiguousarray(g2)), bits(rs))
