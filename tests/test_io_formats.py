"""CPU tests of the on-disk formats: the gnuplot dumps of output_triangulation
(interpolation/linear_simplex_integrity_check.c:246-284), byte for byte against the oracle's recursive
restatement, the order of check_leaf_nodes, and the binary round trip of a built tree."""
import numpy as np


def build(pkg, orc, x):
    t = pkg.SimplexTree(2, len(x))
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    o = orc.Tree(2, len(x))
    assert o.init(x, flags=0, seed=0) == 0
    return t, o


def test_output_triangulation_matches_the_recursive_restatement(pkg, orc, weather, tmp_path):
    data = np.ascontiguousarray(weather[:, :2])
    resp = np.ascontiguousarray(weather[:, 2])
    for name, x, f in (("weather", data, resp), ("cloud", orc.synth_centres(700, 2) * [3.0, 0.5] + [-1.0, 10.0], None)):
        if f is None:
            f = orc.synth_response(x)
        t, o = build(pkg, orc, x)
        for std in (0, 1):
            got = [tmp_path / f"{name}_{std}_g_{k}.dat" for k in ("lines", "points", "circles")]
            want = [tmp_path / f"{name}_{std}_w_{k}.dat" for k in ("lines", "points", "circles")]
            t.output_triangulation(f, std, *got)
            o.output_triangulation(x, f, std, *want)
            for g, w in zip(got, want):
                gb, wb = g.read_bytes(), w.read_bytes()
                assert len(gb) > 100 and gb == wb, (name, std, g.name)
        # lines.dat: two "%g %g %g" rows + two blank lines per drawn edge; every data-data edge of every leaf is drawn
        types, pidx, _ = t.arrays()
        pidx = pidx.reshape(-1, 3)
        edges = sum(int((pidx[k][[a, b]] >= 0).all()) for k in np.flatnonzero(types == 0) for a, b in ((0, 1), (0, 2), (1, 2)))
        assert (tmp_path / f"{name}_0_g_lines.dat").read_text().count("\n\n\n") == edges
        # NULL file names are skipped
        t.output_triangulation(None, 0, None, tmp_path / "only_points.dat", None)
        assert (tmp_path / "only_points.dat").read_bytes() == (tmp_path / f"{name}_0_w_points.dat").read_bytes()


def test_check_leaf_nodes_visits_every_leaf_once_in_dfs_order(pkg, orc):
    x = orc.synth_centres(3000, 2)
    t, _ = build(pkg, orc, x)
    order = t.leaf_walk_order()
    types, _, links = t.arrays()
    links = links.reshape(-1, 3)
    leaves = set(np.flatnonzero(types == 0).tolist())
    assert len(order) == len(leaves) == 2 * 3000 + 1 and set(order) == leaves
    # the reference's recursion (linear_simplex_integrity_check.c:62-119) restated here
    import sys
    sys.setrecursionlimit(20000)
    leaf = 0
    while types[leaf] != 0:
        leaf = links[leaf][0]
    seen, want = set(), []

    def walk(node):
        seen.add(node); want.append(node)
        for nb in links[node]:
            if nb and nb not in seen:
                walk(int(nb))
    walk(int(leaf))
    assert order == want


def test_tree_checkpoint_round_trip(pkg, orc, tmp_path):
    x = orc.synth_centres(5000, 2) * [2.0, 0.7] + [5.0, -3.0]
    f = orc.synth_response(x)
    t, _ = build(pkg, orc, x)
    path = tmp_path / "tree.bin"
    assert t.fwrite(path) == 0
    t2 = pkg.SimplexTree.fread(path, data=x)
    assert t2 is not None and t2.n_nodes == t.n_nodes
    for a, b in zip(t.arrays(), t2.arrays()):
        assert np.array_equal(a, b)
    assert np.array_equal(t.shuffle(), t2.shuffle()) and np.array_equal(t.geom().view(np.uint64), t2.geom().view(np.uint64))
    y = orc.synth_targets(0, 300, 2) * [2.0, 0.7] + [5.0, -3.0]
    for p in y:
        la, lb = t.find_leaf(p), t2.find_leaf(p)
        assert la == lb and t.interp_point(la, f, p) == t2.interp_point(lb, f, p)
    # the restored tree can keep growing: arrays were re-created with room (node_alloc doubling)
    # truncated / foreign files are refused, not crashed on
    blob = path.read_bytes()
    (tmp_path / "short.bin").write_bytes(blob[: len(blob) // 2])
    (tmp_path / "junk.bin").write_bytes(b"not a tree" * 10)
    assert pkg.SimplexTree.fread(tmp_path / "short.bin") is None
    assert pkg.SimplexTree.fread(tmp_path / "junk.bin") is None
    corrupt = bytearray(blob)
    off = 8 + 24 + 4 * t.n_nodes + 4 * 10                       # a vertex id far out of range
    corrupt[off:off + 4] = (2 ** 30).to_bytes(4, "little")
    (tmp_path / "corrupt.bin").write_bytes(bytes(corrupt))
    assert pkg.SimplexTree.fread(tmp_path / "corrupt.bin") is None


def test_tree_checkpoint_round_trips_long_histories(pkg, orc, tmp_path):
    """9 nodes per point is alloc's average-case preallocation (linear_simplex.c:63), not a bound.  Points on a
    parabola inserted in sorted order without a shuffle (rng = NULL, the facade's default) flip almost every earlier edge:
    the history DAG has far more than 9 * n + 8 nodes, and simplex_tree_fread must still read back what fwrite wrote."""
    n = 200
    u = np.linspace(1.0, 0.0, n)
    x = np.ascontiguousarray(np.stack([u, u * u], axis=1))
    t = pkg.SimplexTree(2, n)
    assert t.init(x, flags=0, rng=None) == 0
    assert t.n_nodes > 9 * n + 8, t.n_nodes
    path = tmp_path / "long.bin"
    assert t.fwrite(path) == 0
    t2 = pkg.SimplexTree.fread(path, data=x)
    assert t2 is not None and t2.n_nodes == t.n_nodes
    for a, b in zip(t.arrays(), t2.arrays()):
        assert np.array_equal(a, b)
    f = orc.synth_response(x)
    for p in ([0.5, 0.3], [0.2, 0.1], [0.9, 0.85]):
        la, lb = t.find_leaf(np.array(p)), t2.find_leaf(np.array(p))
        assert la == lb and t.interp_point(la, f, np.array(p)) == t2.interp_point(lb, f, np.array(p))


def test_tree_checkpoint_rejects_corrupt_headers_and_links(pkg, orc, tmp_path):
    """simplex_tree_fread bounds the header counts (no multi-GB allocation / int overflow from 8 corrupt
    bytes), requires the shuffle to be a permutation and child links to point forward (acyclic DAG)."""
    import struct
    n = 40
    x = orc.synth_centres(n, 2)
    t = pkg.SimplexTree(2, n)
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    good = tmp_path / "tree.bin"
    assert t.fwrite(str(good)) == 0
    raw = bytearray(good.read_bytes())
    head = list(struct.unpack_from("<6i", raw, 8))                       # version, dim, n_nodes, n_points, max_points, 0
    nn = head[2]

    def load(buf):
        p = tmp_path / "bad.bin"
        p.write_bytes(bytes(buf))
        try:
            return pkg.SimplexTree.fread(str(p))
        except pkg.capi.GslError:
            return None
    assert load(raw) is not None
    for field, value in ((2, 2**31 - 1), (4, 2**31 - 1), (4, 2**31 // 9 + 5), (4, 2**31 // 27 - 5), (2, 9 * head[4] + 9), (2, 10 * nn)):
        bad = bytearray(raw)
        h = list(head); h[field] = value
        struct.pack_into("<6i", bad, 8, *h)
        assert load(bad) is None
    # shuffle with a repeated entry: last 8 bytes := first shuffle entry
    off_shuffle = len(raw) - 8 * head[4]
    bad = bytearray(raw)
    bad[-8:] = bad[off_shuffle:off_shuffle + 8]
    assert load(bad) is None
    # an inner node whose first child link points back at the root: a cycle for find_leaf
    off_links = 8 + 24 + 4 * nn + 12 * nn
    types = struct.unpack_from(f"<{nn}i", raw, 32)
    inner = next(k for k in range(1, nn) if types[k] != 0)
    bad = bytearray(raw)
    struct.pack_into("<i", bad, off_links + 12 * inner, 0)
    assert load(bad) is None
