"""-m gpu: device-side check_leaf_nodes / check_delaunay (csrc/hip/check.hip) vs the CPU restatement of
interpolation/linear_simplex_integrity_check.c:62-160 in oracle/.  Parity = the SAME VERDICT on valid
trees and on trees corrupted in one place (a neighbour link, a vertex id, a moved data point)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def build(pkg, orc, x, flags=0):
    n = len(x)
    t = pkg.SimplexTree(2, n)
    assert t.init(x, flags=flags, rng=pkg.capi.Rng(0)) == 0
    o = orc.Tree(2, n)
    assert o.init(x, flags=flags, seed=0) == 0
    return t, o


def leaves_of(t):
    types, pidx, links = t.arrays()
    return np.flatnonzero(types == 0), types, pidx.reshape(-1, 3), links.reshape(-1, 3)


@pytest.mark.parametrize("n,flags", [(3, 0), (60, 0), (2000, 0), (2000, 1), (20000, 0)])
def test_valid_trees_pass_like_the_oracle(pkg, orc, n, flags):
    x = orc.synth_centres(n, 2) * np.array([3.0, 0.5]) + np.array([-1.0, 10.0])
    t, o = build(pkg, orc, x, flags)
    assert o.check_leaf_nodes() == 1 and o.check_delaunay(x) == 1
    ok, lv, dv = t.check_device(0)
    assert (ok, lv, dv) == (1, 0, 0)


def test_weather_dataset_tree_passes(pkg, orc, weather):
    data = np.ascontiguousarray(weather[:, :2])
    t, o = build(pkg, orc, data)
    assert o.check_leaf_nodes() == 1 and o.check_delaunay(data) == 1
    assert t.check_device(0) == (1, 0, 0)


def test_c5_size_tree_passes(pkg, orc):
    """N = 50 000: 100 001 leaves x 50 000 points in one launch (the reference's check: O(N^3), unusable)."""
    import time
    x = orc.synth_centres(50_000, 2)
    t = pkg.SimplexTree(2, len(x))
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    t0 = time.time()
    ok, lv, dv = t.check_device(0)
    print("device check of the C5 tree: %.3f s incl. uploads" % (time.time() - t0))
    assert (ok, lv, dv) == (1, 0, 0)


@pytest.mark.parametrize("kind", ["link_to_other_leaf", "drop_reverse_link", "repeat_vertex", "swap_vertex"])
def test_corrupted_structure_is_flagged_like_the_oracle(pkg, orc, kind):
    n = 3000
    x = orc.synth_centres(n, 2)
    t, o = build(pkg, orc, x)
    lv_ids, types, pidx, links = leaves_of(t)
    rng = np.random.default_rng(7)
    # a leaf with three real neighbours, away from the cage
    cand = [k for k in lv_ids if (links[k] > 0).all() and (pidx[k] >= 0).all()]
    k = int(cand[rng.integers(len(cand))])
    for tree in (t.c, o.c):                               # the same surgery on the product tree and on the oracle tree
        L, P = tree.links, tree.pidx
        if kind == "link_to_other_leaf":
            other = int([c for c in cand if c != k and c not in list(links[k])][0])
            L[3 * k + 1] = other
        elif kind == "drop_reverse_link":
            nb = int(links[k][0])
            for j in range(3):
                if L[3 * nb + j] == k:
                    L[3 * nb + j] = 0
        elif kind == "repeat_vertex":
            P[3 * k + 2] = P[3 * k + 0]
        else:                                             # a vertex id replaced by an unrelated point
            P[3 * k + 1] = int((pidx[k][1] + n // 2) % n)
    assert o.check_leaf_nodes() == 0
    ok, lv, dv = t.check_device(0)
    assert ok == 0 and lv >= 1


def test_moved_point_violates_delaunay_like_the_oracle(pkg, orc):
    """The tree is built, then one data row is moved into the middle of a far-away triangle: the structure is
    intact (leaf check clean) but that triangle's circumcircle now holds a point -- what a missed flip leaves."""
    n = 4000
    x = orc.synth_centres(n, 2)
    t, o = build(pkg, orc, x)
    lv_ids, types, pidx, links = leaves_of(t)
    sh = t.shuffle()
    k = int([k for k in lv_ids if (pidx[k] >= 0).all()][100])
    tri_rows = sh[pidx[k]]
    centroid = x[tri_rows].mean(axis=0)
    far = int(np.argmax(((x - centroid) ** 2).sum(axis=1)))      # a point far from that triangle
    assert far not in tri_rows
    x2 = x.copy()
    x2[far] = centroid
    t.set_data(x2)
    assert o.check_leaf_nodes() == 1 and o.check_delaunay(x2) == 0
    ok, lv, dv = t.check_device(0)
    assert ok == 0 and lv == 0 and dv >= 1
    t.set_data(x)
    assert t.check_device(0) == (1, 0, 0)
