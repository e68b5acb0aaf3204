"""Product host code (csrc/host/simplex_tree.c, 2-D closed forms) vs the oracle
(dimension-generic LU restatement): identical DAGs, leaves and values; plus the
reference's own asserted known answers through the product's reference-named API."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SURVEY = json.load(open(os.path.join(HERE, "golden", "survey_known_answers.json")))


def product_tree(pkg, weather, cfg):
    data = weather[:, :2]            # 50 x 2 view with tda = 3, like scattered_interp_example.c:138
    t = pkg.SimplexTree(2, 50)
    if cfg == "cfg0":
        assert t.init(None, flags=pkg.capi.TREE_NOSTANDARDIZE) == 0
        t.set_data(data)
        for i in range(50):
            leaf = t.find_leaf(data[i])
            assert t.insert_point(leaf) == 0
    elif cfg == "cfg2":
        assert t.init(data, flags=0) == 0
    else:
        assert t.init(data, flags=0, rng=pkg.capi.Rng(0)) == 0
    return t, data


@pytest.mark.parametrize("cfg", ["cfg0", "cfg2", "cfg1"])
def test_product_matches_survey_known_answers(pkg, weather, cfg):
    spec = SURVEY["configs"][cfg]
    t, data = product_tree(pkg, weather, cfg)
    assert t.n_nodes == spec["n_nodes"]
    resp = weather[:, 2]             # stride-3 column, scattered_interp_example.c:137
    _, pidx, _ = t.arrays()
    shuffle = t.shuffle()
    for q in spec["queries"]:
        leaf = t.find_leaf(q["point"])
        assert leaf == q["leaf"]
        rows = [int(shuffle[v]) if v >= 0 else int(v) for v in pidx[3 * leaf:3 * leaf + 3]]
        assert rows == q["rows"]
        assert t.interp_point(leaf, resp, q["point"]) == float(q["value"])


def test_product_trivial_test(pkg):
    """scattered_interp_example.c:38-77 against the product symbols."""
    t = pkg.SimplexTree(2, 50)
    assert t.init(None, flags=pkg.capi.TREE_NOSTANDARDIZE) == 0
    data = np.array([[-88.0, 41.0], [-89.0, 41.0]])
    leaf = t.find_leaf(data[0])
    assert leaf == 0
    t.set_data(data)
    assert t.interp_point(leaf, None, data[0]) == 0.0
    assert t.insert_point(leaf) == 0
    ty, pidx, links = t.arrays()
    assert ty[0] != 0
    assert [list(pidx[3 * k:3 * k + 3]) for k in links[0:3]] == [[0, -2, -3], [0, -1, -3], [0, -1, -2]]
    assert t.in_hypersphere(0, 0) == 1
    leaf2 = t.find_leaf(data[1])
    assert list(pidx[3 * leaf2:3 * leaf2 + 3]) == [0, -2, -3]


@pytest.mark.parametrize("n,flags,seeded", [(60, 0, True), (150, 0, True), (1000, 0, False), (1000, 2, True),
                                             (5000, 1, True), (20000, 0, True)])
def test_product_dag_bit_identical_to_oracle(pkg, orc, n, flags, seeded):
    x = orc.synth_centres(n, 2) * np.array([3.0, 0.5]) + np.array([-1.0, 10.0])   # non-unit box: shift/scale non-trivial
    t = pkg.SimplexTree(2, n)
    assert t.init(x, flags=flags, rng=pkg.capi.Rng(0) if seeded else None) == 0
    o = orc.Tree(2, n)
    assert o.init(x, flags=flags, seed=0 if seeded else None) == 0
    for a, b in zip(t.arrays(), o.arrays()):
        assert np.array_equal(a, b)
    assert np.array_equal(t.shuffle(), o.shuffle())
    assert np.array_equal(t.geom(), o.geom())
    # located leaves and values for a few hundred targets, bit for bit
    f = orc.synth_response(x)
    y = orc.synth_targets(0, 300, 2) * np.array([3.0, 0.5]) + np.array([-1.0, 10.0])
    ovals, oleaf = o.eval_many(x, f, y)
    for k in range(len(y)):
        leaf = t.find_leaf(y[k])
        assert leaf == oleaf[k]
        v = t.interp_point(leaf, f, y[k])
        assert np.float64(v).view(np.uint64) == ovals[k:k + 1].view(np.uint64)[0]


def test_degenerate_inputs(pkg, orc):
    # collinear + duplicated points exercise the singular / not-flippable branches
    x = np.array([[0.0, 0.0], [1.0, 1.0], [2.0, 2.0], [3.0, 3.0], [1.0, 1.0], [0.5, 2.0], [2.0, 0.5], [4.0, 4.0]])
    t = pkg.SimplexTree(2, len(x))
    o = orc.Tree(2, len(x))
    assert t.init(x, flags=0) == 0 and o.init(x, flags=0) == 0
    for a, b in zip(t.arrays(), o.arrays()):
        assert np.array_equal(a, b)


def test_exact_lattice_fails_cleanly_and_jittered_lattice_matches_oracle(pkg, orc):
    """On an exactly regular lattice (collinear + co-circular points everywhere) the reference's recursive
    flip cascade never settles and overruns the stack (the oracle, a faithful restatement, does too -- it is
    not called on that input).  The product bounds the recursion and reports GSL_EFAILED.  The same lattice
    with 1e-7 jitter builds, bit-identical to the oracle."""
    n1 = 8
    gx, gy = np.meshgrid(np.arange(n1) / (n1 - 1.0), np.arange(n1) / (n1 - 1.0))
    lattice = np.ascontiguousarray(np.column_stack([gx.ravel(), gy.ravel()]))
    t = pkg.SimplexTree(2, len(lattice))
    assert t.init(lattice, flags=0, rng=pkg.capi.Rng(0)) == pkg.capi.GSL_EFAILED
    n1 = 24
    gx, gy = np.meshgrid(np.arange(n1) / (n1 - 1.0), np.arange(n1) / (n1 - 1.0))
    x = np.ascontiguousarray(np.column_stack([gx.ravel(), gy.ravel()]) + 1e-7 * np.random.default_rng(1).standard_normal((n1 * n1, 2)))
    t = pkg.SimplexTree(2, len(x))
    o = orc.Tree(2, len(x))
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0 and o.init(x, flags=0, seed=0) == 0
    for a, b in zip(t.arrays(), o.arrays()):
        assert np.array_equal(a, b)


def test_outside_cage_reports_edom(pkg):
    x = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    t = pkg.SimplexTree(2, 4)
    assert t.init(x, flags=0) == 0
    assert t.find_leaf([1e9, 1e9]) == -1          # q7: no abort, -1 (+ GSL_EDOM through the handler)
    assert t.find_leaf([0.3, 0.3]) > 0


def test_unsupported_dimension_and_capacity(pkg):
    with pytest.raises(pkg.capi.GslError):
        pkg.SimplexTree(3, 10)                     # reference flip logic is 2-D only
    t = pkg.SimplexTree(2, 2)
    x = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    assert t.init(x, flags=0) == pkg.capi.GSL_FAILURE   # linear_simplex.c:274-278


def test_capi_exports_every_declared_symbol(pkg):
    """The C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
    import re
    L = pkg.lib()
    root = os.path.dirname(HERE)
    declared = set()
    for hdr in ("gsl_sinterp.h", "gsl_sinterp_hip.h"):
        txt = open(os.path.join(root, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        declared |= set(re.findall(r"\b(gsl_sinterp_\w+|simplex_tree_\w+|find_leaf|_find_leaf|insert_point|"
                                   r"in_hypersphere\w*|calculate_\w+|contains_point|interp_point|delaunay|"
                                   r"point_in_simplex)\s*\(", txt))
    declared -= {"gsl_sinterp_struct", "simplex_tree_struct", "simplex_tree_node_struct"}
    assert len(declared) > 50
    for name in sorted(declared):
        assert hasattr(L, name), name
    for name in pkg.capi.DATA_SYMBOLS:
        assert pkg.capi._ptr(name)
    assert set(pkg.capi.SIGNATURES) >= {d for d in declared}
    assert L.gsl_sinterp_hip_device_count() >= 0


def test_facade_argument_errors_without_gpu(pkg):
    with pytest.raises(pkg.capi.GslError):
        pkg.Sinterp("linear_simplex", 2, 2)        # min_size 3 -> GSL_EINVAL like gsl_interp_alloc
    s = pkg.Sinterp("gaussian", 2, 8)
    assert s.name() == "rbf-gaussian"
    assert s.init(np.zeros((7, 2)), np.zeros(7)) == pkg.capi.GSL_EINVAL   # size mismatch, interp.c:73-76
    assert s.init(np.zeros((8, 3)), np.zeros(8)) == pkg.capi.GSL_EINVAL
    st, val = s.eval_e([0.0, 0.0])
    assert st != 0 and np.isnan(val)               # not initialised -> status + NaN, never a CPU answer


def test_docs_are_sane_text_files():
    """DESIGN.md was once corrupted into a 13 MB file by a doc-update slip: keep the docs small, line-structured text."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("DESIGN.md", "INTEGRATION.md", "README.md"):
        path = os.path.join(root, name)
        size = os.path.getsize(path)
        text = open(path, encoding="utf-8").read()
        assert 1000 < size < 200_000, (name, size)
        assert text.count("\n") < 2000 and text.startswith("#"), name
    design = open(os.path.join(root, "DESIGN.md"), encoding="utf-8").read()
    for sec in ("## 1.", "## 2.", "## 3.", "## 4.", "## 5.", "## 6.", "## 7."):
        assert design.count("\n" + sec) == 1, sec
