"""-m gpu: barycentric HIP path vs the CPU oracle, through the C-ABI.

Bar: leaf indices and values BIT-EXACT (int32 / fp64 bit patterns)."""
import os

import numpy as np
import pytest

from gpu_util import bits, dev, ptr

pytestmark = pytest.mark.gpu


def build_pair(pkg, orc, x, flags=0, seeded=True):
    n = len(x)
    t = pkg.SimplexTree(2, n)
    assert t.init(x, flags=flags, rng=pkg.capi.Rng(0) if seeded else None) == 0
    o = orc.Tree(2, n)
    assert o.init(x, flags=flags, seed=0 if seeded else None) == 0
    return t, o


@pytest.mark.parametrize("n,m,flags", [(8, 500, 0), (512, 20000, 0), (4096, 100000, 0), (4096, 50000, 1), (4096, 50000, 2)])
def test_bary_matches_oracle_bitexact(pkg, orc, n, m, flags):
    scale, off = np.array([3.0, 0.5]), np.array([-1.0, 10.0])
    x = orc.synth_centres(n, 2) * scale + off
    f = orc.synth_response(x)
    y = orc.synth_targets(0, m, 2) * scale + off
    t, o = build_pair(pkg, orc, x, flags)
    d = t.device_alloc(0)
    assert d.set_response(f) == 0
    st, vals, leaf = d.eval_many(y)
    ovals, oleaf = o.eval_many(x, f, y)
    assert st == 0
    assert np.array_equal(leaf, oleaf)
    assert np.array_equal(bits(vals), bits(ovals))


def test_bary_weather_golden(pkg, weather):
    """survey-captured outputs of the reference (tests/golden/survey_known_answers.json), cfg1."""
    import json, os
    spec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_known_answers.json")))["configs"]["cfg1"]
    data = weather[:, :2]
    t = pkg.SimplexTree(2, 50)
    assert t.init(data, flags=0, rng=pkg.capi.Rng(0)) == 0
    d = t.device_alloc(0)
    assert d.set_response(weather[:, 2]) == 0            # stride-3 response column
    y = np.array([q["point"] for q in spec["queries"]], dtype=np.float64)
    st, vals, leaf = d.eval_many(y)
    assert st == 0
    assert list(leaf) == [q["leaf"] for q in spec["queries"]]
    assert [float(v) for v in vals] == [float(q["value"]) for q in spec["queries"]]


def test_bary_edge_cases(pkg, orc):
    n = 1000
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    t, o = build_pair(pkg, orc, x)
    d = t.device_alloc(0)
    assert d.set_response(f) == 0
    # targets ON data points, on edge midpoints, outside the hull (cage region) and outside the cage
    mids = 0.5 * (x[:-1] + x[1:])
    hull_out = np.array([[-5.0, 0.5], [0.5, 7.0], [30.0, 30.0]])
    y = np.concatenate([x[:200], mids[:200], hull_out])
    st, vals, leaf = d.eval_many(y)
    ovals, oleaf = o.eval_many(x, f, y)
    assert st == 0 and np.array_equal(leaf, oleaf) and np.array_equal(bits(vals), bits(ovals))
    far = np.array([[0.5, 0.5], [1e9, -1e9], [0.25, 0.75]])
    st, vals, leaf = d.eval_many(far)
    assert st == pkg.capi.GSL_EDOM and leaf[1] == -1 and np.isnan(vals[1])     # q7: reported, not aborted
    assert leaf[0] >= 0 and leaf[2] >= 0 and not np.isnan(vals[[0, 2]]).any()
    # empty batch
    st, vals, leaf = d.eval_many(np.zeros((0, 2)))
    assert st == 0 and len(vals) == 0
    # ragged target layout: rows of a wider matrix (tda = 5)
    wide = np.zeros((300, 5)); wide[:, :2] = orc.synth_targets(7, 300, 2)
    st, vals, leaf = d.eval_many(wide[:, :2])
    ovals, oleaf = o.eval_many(x, f, np.ascontiguousarray(wide[:, :2]))
    assert st == 0 and np.array_equal(leaf, oleaf) and np.array_equal(bits(vals), bits(ovals))


def test_bary_empty_cage_interpolates_zero(pkg):
    """scattered_interp_example.c:51-52 on the device path."""
    t = pkg.SimplexTree(2, 4)
    assert t.init(None, flags=pkg.capi.TREE_NOSTANDARDIZE) == 0
    d = t.device_alloc(0)
    assert d.set_response(np.zeros(0)) == 0
    st, vals, leaf = d.eval_many(np.array([[-88.0, 41.0], [0.0, 0.0]]))
    assert st == 0 and list(leaf) == [0, 0] and list(vals) == [0.0, 0.0]


def test_bary_resident_full_size_properties(pkg, orc):
    """C5 shape (N = 50 000, M = 10^6 here, 10^7 in bench): size-independent properties --
    linear reproduction, value bounds, sampled bit-exactness vs the oracle."""
    import torch
    n, m = 50000, 1_000_000
    x = orc.synth_centres(n, 2)
    g = 2 * x[:, 0] - 3 * x[:, 1] + 0.5
    t, o = build_pair(pkg, orc, x)
    d = t.device_alloc(0)
    assert d.set_response(g) == 0
    ctx = pkg.HipContext.on_torch_stream(0)
    ty = torch.empty((m, 2), dtype=torch.float64, device="cuda")
    ctx.synth_unit(0xC0FFEE02, 0, 0.02, 0.96, ptr(ty), 2 * m)
    ctx.sync()
    yh = ty.cpu().numpy()
    assert np.array_equal(bits(yh[:1000]), bits(orc.synth_targets(0, 1000, 2)))   # same cloud on both sides
    tv = torch.empty(m, dtype=torch.float64, device="cuda")
    tl = torch.empty(m, dtype=torch.int32, device="cuda")
    assert d.eval_resident(ptr(ty), m, 2, ptr(tv), ptr(tl)) == 0
    torch.cuda.synchronize()
    lib_ctx = pkg.lib().simplex_tree_device_ctx(d._h)
    assert pkg.lib().gsl_sinterp_hip_sync(lib_ctx) == 0
    vals, leaf = tv.cpu().numpy(), tl.cpu().numpy()
    assert np.abs(vals - (2 * yh[:, 0] - 3 * yh[:, 1] + 0.5)).max() < 5e-15
    ty_, _, _ = t.arrays()
    assert (ty_[leaf] == 0).all()                                       # every result is a leaf
    idx = np.arange(0, m, 97)
    ovals, oleaf = o.eval_many(x, g, np.ascontiguousarray(yh[idx]))
    assert np.array_equal(leaf[idx], oleaf) and np.array_equal(bits(vals[idx]), bits(ovals))


def test_lowlevel_pack_matches_host_records(pkg, orc):
    """tree_pack on raw device arrays (the multi-GPU broadcast path) == records built via the host API."""
    import torch
    n, m = 3000, 30000
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    t, o = build_pair(pkg, orc, x)
    types, pidx, links = t.arrays()
    sh = t.shuffle()
    ctx = pkg.HipContext.on_torch_stream(0)
    nn = t.n_nodes
    d_type, d_pidx, d_links = dev(types), dev(pidx), dev(links)
    d_pts, d_resp = dev(x[sh]), dev(f[sh])
    rec = torch.empty(nn * 64, dtype=torch.uint8, device="cuda")
    tab = torch.empty(nn * 32, dtype=torch.uint8, device="cuda")
    geom = t.geom()
    ctx.tree_pack(nn, ptr(d_type), ptr(d_pidx), ptr(d_links), n, ptr(d_pts), geom, ptr(rec))
    ctx.tree_bind(nn, ptr(d_pidx), n, ptr(d_resp), ptr(tab))
    y = orc.synth_targets(0, m, 2)
    d_y = dev(y)
    d_v = torch.empty(m, dtype=torch.float64, device="cuda")
    d_l = torch.empty(m, dtype=torch.int32, device="cuda")
    outside = ctx.bary_eval(nn, ptr(rec), ptr(tab), geom[8:10], ptr(d_y), m, 2, ptr(d_v), ptr(d_l), count_outside=True)
    ovals, oleaf = o.eval_many(x, f, y)
    assert outside == 0
    assert np.array_equal(d_l.cpu().numpy(), oleaf) and np.array_equal(bits(d_v.cpu().numpy()), bits(ovals))


def test_facade_linear_simplex(pkg, orc):
    """gsl_sinterp alloc/init/eval_many/eval_e with the linear-simplex type == oracle, bit for bit."""
    n, m = 3000, 20000
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    y = orc.synth_targets(0, m, 2)
    s = pkg.Sinterp("linear_simplex", 2, n, 0)
    assert s.name() == "linear-simplex"
    assert s.set_tree_options(0, pkg.capi.Rng(0)) == 0
    assert s.init(x, f) == 0
    st, vals, leaf = s.eval_many(y, want_leaf=True)
    o = orc.Tree(2, n)
    assert o.init(x, flags=0, seed=0) == 0
    ovals, oleaf = o.eval_many(x, f, y)
    assert st == 0 and np.array_equal(leaf, oleaf) and np.array_equal(bits(vals), bits(ovals))
    st, v = s.eval_e(y[7])
    assert st == 0 and np.float64(v).view(np.uint64) == bits(ovals[7:8])[0]
    st, v = s.eval_e([1e12, 0.0])
    assert st == pkg.capi.GSL_EDOM and np.isnan(v)       # like gsl_interp_eval_e outside [xmin, xmax]


@pytest.mark.parametrize("packer_is_evaluator", [True, False])
def test_jump_table_is_exact_on_adversarial_targets(pkg, orc, packer_is_evaluator):
    """The walk starts from a per-grid-cell node (jump table, bary.hip) instead of the root.  Targets that
    stress exactly that: on the data points, on midpoints of point pairs, ON the lines of the jump grid
    (computed with the kernel's own expressions), just outside the data's bounding box, plus a random
    cloud -- leaf indices and values must stay bit-identical to the oracle's walk from the root.  Both table
    flavours: built by tree_pack over the data's box (packer == evaluator) and built per batch over the
    targets' box (records packed on another context, the multi-GPU broadcast situation)."""
    import torch
    n = 20000
    x = orc.synth_centres(n, 2) * np.array([2.0, 0.7]) + np.array([5.0, -3.0])
    f = orc.synth_response(x)
    t, o = build_pair(pkg, orc, x)
    nn = t.n_nodes
    G = 32
    while G < 4096 and G * G < 40.0 * nn:               # the rule of gsl_sinterp_hip_tree_pack
        G *= 2
    lo, hi = x.min(axis=0), x.max(axis=0)
    w = (hi - lo) / G
    rng = np.random.default_rng(11)
    lines = []
    for i in range(G + 1):
        lines.append(np.column_stack([np.full(8, lo[0] + w[0] * i), lo[1] + (hi[1] - lo[1]) * rng.random(8)]))
        lines.append(np.column_stack([lo[0] + (hi[0] - lo[0]) * rng.random(8), np.full(8, lo[1] + w[1] * i)]))
    corners = np.array([[lo[0] + w[0] * i, lo[1] + w[1] * j] for i in range(0, G + 1, 7) for j in range(0, G + 1, 5)])
    outside_box = np.column_stack([hi[0] + 0.01 * (1 + rng.random(50)), lo[1] + (hi[1] - lo[1]) * rng.random(50)])
    pairs = rng.integers(0, n, size=(20000, 2))
    y = np.concatenate([x, 0.5 * (x[pairs[:, 0]] + x[pairs[:, 1]]), np.concatenate(lines), corners, outside_box,
                        orc.synth_targets(3, 30000, 2) * np.array([2.0, 0.7]) + np.array([5.0, -3.0])])
    m = len(y)
    types, pidx, links = t.arrays()
    sh = t.shuffle()
    ctx_pack = pkg.HipContext.on_torch_stream(0)
    ctx_eval = ctx_pack if packer_is_evaluator else pkg.HipContext.on_torch_stream(0)
    d_type, d_pidx, d_links = dev(types), dev(pidx), dev(links)
    d_pts, d_resp = dev(x[sh]), dev(f[sh])
    rec = torch.empty(nn * 64, dtype=torch.uint8, device="cuda")
    tab = torch.empty(nn * 32, dtype=torch.uint8, device="cuda")
    geom = t.geom()
    ctx_pack.tree_pack(nn, ptr(d_type), ptr(d_pidx), ptr(d_links), n, ptr(d_pts), geom, ptr(rec))
    ctx_pack.tree_bind(nn, ptr(d_pidx), n, ptr(d_resp), ptr(tab))
    ctx_pack.sync()
    d_y = dev(y)
    d_v = torch.empty(m, dtype=torch.float64, device="cuda")
    d_l = torch.empty(m, dtype=torch.int32, device="cuda")
    outside = ctx_eval.bary_eval(nn, ptr(rec), ptr(tab), geom[8:10], ptr(d_y), m, 2, ptr(d_v), ptr(d_l), count_outside=True)
    ovals, oleaf = o.eval_many(x, f, y)
    assert outside == 0
    assert np.array_equal(d_l.cpu().numpy(), oleaf)
    assert np.array_equal(bits(d_v.cpu().numpy()), bits(ovals))
    # a small batch takes the unsorted path (and, on the packing context, still the table)
    outside = ctx_eval.bary_eval(nn, ptr(rec), ptr(tab), geom[8:10], ptr(d_y), 3000, 2, ptr(d_v), ptr(d_l), count_outside=True)
    assert np.array_equal(d_l.cpu().numpy()[:3000], oleaf[:3000]) and np.array_equal(bits(d_v.cpu().numpy()[:3000]), bits(ovals[:3000]))


@pytest.mark.parametrize("cloud", ["jittered_lattice", "clusters_and_near_collinear"])
def test_sliver_geometry_stays_bitexact(pkg, orc, cloud):
    """Point sets that produce sliver triangles and badly conditioned 2x2 systems (a lattice with 1e-7
    jitter; a 1e-3-wide cluster plus 1500 points within 1e-9 of a line): the jump table must leave the
    nodes it cannot classify alone, and the batched result must stay bit-identical to the oracle."""
    rng = np.random.default_rng(5)
    if cloud == "jittered_lattice":
        n1 = 48
        gx, gy = np.meshgrid(np.arange(n1) / (n1 - 1.0), np.arange(n1) / (n1 - 1.0))
        nodes = np.column_stack([gx.ravel(), gy.ravel()])
        x = np.ascontiguousarray(nodes + 1e-7 * rng.standard_normal(nodes.shape))
        extra = nodes                                          # the exact lattice nodes as targets
    else:
        x = np.ascontiguousarray(np.concatenate([rng.random((2000, 2)) * 1e-3 + 0.5,
                                                 np.column_stack([np.linspace(0, 1, 1500), 0.3 + 1e-9 * rng.standard_normal(1500)]),
                                                 rng.random((1500, 2))]))
        extra = np.column_stack([np.linspace(0, 1, 3000), np.full(3000, 0.3)])   # along the line itself
    f = orc.synth_response(x)
    t, o = build_pair(pkg, orc, x)
    d = t.device_alloc(0)
    assert d.set_response(f) == 0
    y = np.ascontiguousarray(np.concatenate([x, extra, rng.random((20000, 2)), 0.5 + 1e-3 * rng.random((5000, 2))]))
    st, vals, leaf = d.eval_many(y)
    ovals, oleaf = o.eval_many(x, f, y)
    assert st == 0
    assert np.array_equal(leaf, oleaf)
    assert np.array_equal(bits(vals), bits(ovals))


def test_division_free_containment_margin(pkg, orc):
    """The walk decides containment with reciprocals and an error certificate and repeats a node with the exact
    IEEE arithmetic when the certificate fails (bary.hip: classify_fast).  Targets placed ON edges and vertices
    of the triangulation and then moved off them by 1e-16 ... 1e-9 of the data's extent straddle exactly that
    margin: every one must still land in the oracle's leaf with the oracle's bits."""
    n = 6000
    rng = np.random.default_rng(23)
    x = orc.synth_centres(n, 2) * np.array([3.0, 0.5]) + np.array([-7.0, 11.0])
    f = orc.synth_response(x)
    t, o = build_pair(pkg, orc, x)
    d = t.device_alloc(0)
    assert d.set_response(f) == 0
    pairs = rng.integers(0, n, size=(4000, 2))
    lam = rng.random((4000, 1))
    on_segments = lam * x[pairs[:, 0]] + (1.0 - lam) * x[pairs[:, 1]]
    base = np.concatenate([x, on_segments])
    ys = [base]
    for mag in (1e-16, 1e-15, 1e-14, 1e-13, 3e-13, 1e-12, 3e-12, 1e-11, 1e-10, 1e-9):
        ys.append(base + mag * 3.0 * rng.standard_normal(base.shape))
    y = np.ascontiguousarray(np.concatenate(ys))
    st, vals, leaf = d.eval_many(y)
    ovals, oleaf = o.eval_many(x, f, y)
    assert st == 0
    assert np.array_equal(leaf, oleaf)
    assert np.array_equal(bits(vals), bits(ovals))


@pytest.mark.parametrize("shape", ["uniform", "one_point", "horizontal_line", "two_clusters", "with_nan_and_outside"])
def test_degenerate_large_batches_stay_bitexact(pkg, orc, shape):
    """Large batches go through the cell sort, the start / walk / finish kernels and the un-sort gather.  Degenerate
    batches stress the bookkeeping: every target in one cell (one counter takes every atomic), one populated grid
    row, two far clusters (empty cells in between), NaN / outside targets (cell 0, exact kernel's queue).  Every
    target must come back at its own position with the oracle's leaf and bits."""
    n = 3000
    rng = np.random.default_rng(31)
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    t, o = build_pair(pkg, orc, x)
    d = t.device_alloc(0)
    assert d.set_response(f) == 0
    m = 300_000
    lo, hi = x.min(axis=0), x.max(axis=0)
    if shape == "uniform":
        y = lo + (hi - lo) * rng.random((m, 2))
    elif shape == "one_point":
        y = np.tile(0.5 * (lo + hi), (m, 1))
    elif shape == "horizontal_line":
        y = np.column_stack([lo[0] + (hi[0] - lo[0]) * rng.random(m), np.full(m, 0.5 * (lo[1] + hi[1]))])
    elif shape == "two_clusters":
        y = np.concatenate([lo + 0.01 * (hi - lo) * rng.random((m // 2, 2)), hi - 0.01 * (hi - lo) * rng.random((m - m // 2, 2))])
    else:
        y = lo + (hi - lo) * rng.random((m, 2))
        y[::1000] = np.nan
        y[7::5000] = [1e9, -1e9]
    y = np.ascontiguousarray(y)
    st, vals, leaf = d.eval_many(y)
    sub = np.unique(np.concatenate([np.arange(0, m, 37), np.arange(0, 2000), np.arange(m - 2000, m)]))
    if shape == "with_nan_and_outside":                       # the oracle is asked about ordinary targets only
        sub = sub[(sub % 1000 != 0) & (sub % 5000 != 7)]
        assert np.all(np.isnan(vals[::1000]))
    ovals, oleaf = o.eval_many(x, f, y[sub])
    assert np.array_equal(leaf[sub], oleaf)
    assert np.array_equal(bits(vals[sub]), bits(ovals))
    if shape == "one_point":
        assert np.all(leaf == leaf[0]) and np.all(bits(vals) == bits(vals[:1])[0])
    if shape == "with_nan_and_outside":
        assert st == pkg.capi.GSL_EDOM and np.all(leaf[7::5000] == -1) and np.all(np.isnan(vals[7::5000]))
    else:
        assert st == 0


def test_large_batch_values_only_and_strided_targets(pkg, orc):
    """The sorted path with no leaf output (un-sort of plain values, no {value, leaf} pairs) and with targets that
    are the first two columns of a wider matrix (ttda = 5) must give the bits of the leaf-returning call."""
    import torch
    n = 4000
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    t, o = build_pair(pkg, orc, x)
    nn = t.n_nodes
    types, pidx, links = t.arrays()
    sh = t.shuffle()
    ctx = pkg.HipContext.on_torch_stream(0)
    d_type, d_pidx, d_links = dev(types), dev(pidx), dev(links)
    d_pts, d_resp = dev(x[sh]), dev(f[sh])
    rec = torch.empty(nn * 64, dtype=torch.uint8, device="cuda")
    tab = torch.empty(nn * 32, dtype=torch.uint8, device="cuda")
    geom = t.geom()
    ctx.tree_pack(nn, ptr(d_type), ptr(d_pidx), ptr(d_links), n, ptr(d_pts), geom, ptr(rec))
    ctx.tree_bind(nn, ptr(d_pidx), n, ptr(d_resp), ptr(tab))
    m = 50_000
    wide = np.zeros((m, 5))
    wide[:, :2] = orc.synth_targets(5, m, 2)
    wide[:, 2:] = 1e30                                            # must never be read as coordinates
    d_wide = dev(wide)
    d_v = torch.empty(m, dtype=torch.float64, device="cuda")
    d_l = torch.empty(m, dtype=torch.int32, device="cuda")
    ctx.bary_eval(nn, ptr(rec), ptr(tab), geom[8:10], ptr(d_wide), m, 5, ptr(d_v), ptr(d_l))
    ovals, oleaf = o.eval_many(x, f, np.ascontiguousarray(wide[:, :2]))
    assert np.array_equal(d_l.cpu().numpy(), oleaf) and np.array_equal(bits(d_v.cpu().numpy()), bits(ovals))
    d_v2 = torch.full((m,), -7.0, dtype=torch.float64, device="cuda")
    ctx.bary_eval(nn, ptr(rec), ptr(tab), geom[8:10], ptr(d_wide), m, 5, ptr(d_v2), None)
    assert np.array_equal(bits(d_v2.cpu().numpy()), bits(ovals))


@pytest.mark.parametrize("shape", ["uniform", "cluster_plus_sparse_background"])
def test_two_level_reorder_matches_one_level_and_oracle(pkg, orc, monkeypatch, shape):
    """Round 3: batches of >= 2^18 targets are ordered by the two-level reorder of sort.hip (coarse bins by LDS
    histograms, cells inside LDS windows, one global atomic per occupied (unit, cell), every scattered store staged through
    LDS into runs); GSL_SINTERP_SORT_LEVELS=1 / 2 forces the one-atomic-per-point route / this one.  A value depends on (records, target) only, so both routes must return the
    oracle's bits at every position.  `cluster_plus_sparse_background`: 95 % of the targets in a small disc stretch the
    grid so that the remaining 5 % leave < 4 points per cell -- units that span more cells than their LDS window and take
    the per-point fallback for the cells beyond it.  Also the values-only un-sort (8-byte results, no leaf)."""
    import torch
    n, m = 5000, 400_000
    rng = np.random.default_rng(77)
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    lo, hi = x.min(axis=0), x.max(axis=0)
    if shape == "uniform":
        y = lo + (hi - lo) * rng.random((m, 2))
    else:
        nc = int(0.95 * m)
        c = lo + 0.3 * (hi - lo) + 0.002 * (hi - lo) * rng.standard_normal((nc, 2))
        y = np.concatenate([c, lo + (hi - lo) * rng.random((m - nc, 2))])
        y = y[rng.permutation(m)]
    y = np.ascontiguousarray(y)
    t, o = build_pair(pkg, orc, x)
    d = t.device_alloc(0)
    assert d.set_response(f) == 0
    monkeypatch.setenv("GSL_SINTERP_SORT_LEVELS", "1")
    st1, v1, l1 = d.eval_many(y)
    monkeypatch.setenv("GSL_SINTERP_SORT_LEVELS", "2")
    st2, v2, l2 = d.eval_many(y)
    ty = dev(y)
    tv = torch.full((m,), -3.0, dtype=torch.float64, device="cuda")
    assert d.eval_resident(ptr(ty), m, 2, ptr(tv), None) == st2          # values only
    torch.cuda.synchronize()
    assert np.array_equal(bits(tv.cpu().numpy()), bits(v2))
    monkeypatch.delenv("GSL_SINTERP_SORT_LEVELS")
    assert st1 == st2
    assert np.array_equal(l1, l2) and np.array_equal(bits(v1), bits(v2))
    idx = np.unique(np.concatenate([np.arange(0, m, 41), np.arange(m - 3000, m)]))
    ov, ol = o.eval_many(x, f, np.ascontiguousarray(y[idx]))
    ok = ol >= 0
    assert np.array_equal(l2[idx], ol) and np.array_equal(bits(v2[idx][ok]), bits(ov[ok]))


def test_certified_leaf_walk_takes_the_bulk_and_queues_the_edges(pkg, orc):
    """Round 4: batches of >= 4096 targets on records packed by tree_pack are located by a walk over the LEAVES' adjacency from a
    grid seed and accepted only with a margin from every edge that can influence the reference's DAG descent (the leaf's own
    edges + the flipped-away edges that cross it); everything else goes to the exact DAG kernel.  Leaf and value bit-exact;
    the queue is tiny for targets in general position and takes every target placed on an edge or a vertex."""
    import ctypes
    n, m = 20000, 200_000
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    t, o = build_pair(pkg, orc, x)
    d = t.device_alloc(0)
    assert d.set_response(f) == 0
    y = orc.synth_targets(0, m, 2) * 1.2 - 0.1                  # some outside the hull, in the cage leaves
    st, v, l = d.eval_many(y)
    q, lw = ctypes.c_uint(0), ctypes.c_int(0)
    assert pkg.lib().gsl_sinterp_hip_bary_last_queue(pkg.lib().simplex_tree_device_ctx(d._h), ctypes.byref(q), ctypes.byref(lw)) == 0
    plain = not any(os.environ.get(k) == "1" for k in ("GSL_SINTERP_NO_SORT", "GSL_SINTERP_NO_LEAFWALK", "GSL_SINTERP_NO_FASTDIV"))
    assert lw.value == (1 if plain else 0)
    if plain:
        assert q.value < m // 100, q.value                      # ~0.6 %: margin bands of the early, long edges + the cage leaves
    idx = np.arange(0, m, 41)
    ov, ol = o.eval_many(x, f, np.ascontiguousarray(y[idx]))
    assert st == 0 and np.array_equal(l[idx], ol) and np.array_equal(bits(v[idx]), bits(ov))
    # targets ON data points, on edge midpoints of the final triangulation, and nudged off them by 1e-16 .. 1e-9
    types, pidx, links = t.arrays()
    sh = t.shuffle()
    leaves = np.nonzero((types == 0) & (pidx.reshape(-1, 3) >= 0).all(axis=1))[0][:3000]
    tri = sh[pidx.reshape(-1, 3)[leaves]]
    mid = 0.5 * (x[tri[:, 0]] + x[tri[:, 1]])
    hard = np.vstack([x[:3000], mid] + [mid + s * np.array([1.0, -0.7]) for s in (1e-16, 1e-14, 1e-12, 1e-10, 1e-9)])
    hard = np.ascontiguousarray(np.vstack([hard, y[: 4096]]))
    st2, v2, l2 = d.eval_many(hard)
    assert pkg.lib().gsl_sinterp_hip_bary_last_queue(pkg.lib().simplex_tree_device_ctx(d._h), ctypes.byref(q), ctypes.byref(lw)) == 0
    assert not plain or q.value >= 6000                          # every on-vertex / on-edge target was left to the exact kernel
    ov2, ol2 = o.eval_many(x, f, hard)
    assert np.array_equal(l2, ol2) and np.array_equal(bits(v2), bits(ov2))
