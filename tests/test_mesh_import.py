"""Imported triangulations (the tail of row 4 + the leaf-adjacency walk of row 1; README:28-31 of the
reference lists the import as future work, so there is no reference walk: PARITY UNPINNED for the locate step --
the per-triangle arithmetic is the reference's and is checked against its restatement on explicit vertices,
oracle_mesh_*).  CPU part: validation and neighbour derivation; -m gpu part: tests/test_gpu_mesh.py."""
import numpy as np
import pytest


def qhull(points):
    from scipy.spatial import Delaunay
    d = Delaunay(points)
    return d, d.simplices.astype(np.int32), d.neighbors.astype(np.int32)


def test_import_derives_qhull_neighbours(pkg, orc):
    x = orc.synth_centres(500, 2)
    _, tri, nbr = qhull(x)
    given = pkg.SimplexMesh.from_arrays(x, tri, nbr)
    derived = pkg.SimplexMesh.from_arrays(x, tri)               # neighbours == NULL: edge matching
    assert given.n_triangles == len(tri)
    assert np.array_equal(given.neighbours(), nbr) and np.array_equal(derived.neighbours(), nbr)
    assert np.array_equal(derived.triangles(), tri) and derived.tree_nodes() is None


def test_import_decides_convexity_from_the_boundary(pkg, orc):
    """an imported mesh used to default to convex = 1, so targets inside a mesh with a hole or a concave
    outline were reported outside unless the caller knew to call simplex_mesh_set_convex(mesh, 0).  The import now looks
    at the boundary loop: QHull's Delaunay output is convex; a punched hole, an L-shaped outline and two separate
    components are not."""
    x = orc.synth_centres(800, 2)
    d, tri, nbr = qhull(x)
    assert pkg.SimplexMesh.from_arrays(x, tri, nbr).convex() and pkg.SimplexMesh.from_arrays(x, tri).convex()
    cen = x[tri].mean(axis=1)
    hole = ~((np.abs(cen[:, 0] - 0.5) < 0.2) & (np.abs(cen[:, 1] - 0.5) < 0.2))
    assert not pkg.SimplexMesh.from_arrays(x, np.ascontiguousarray(tri[hole])).convex()
    ell = ~((cen[:, 0] > 0.5) & (cen[:, 1] > 0.5))                                     # L-shaped outline
    assert not pkg.SimplexMesh.from_arrays(x, np.ascontiguousarray(tri[ell])).convex()
    two = (cen[:, 0] < 0.3) | (cen[:, 0] > 0.7)                                        # two components
    assert not pkg.SimplexMesh.from_arrays(x, np.ascontiguousarray(tri[two])).convex()
    m = pkg.SimplexMesh.from_arrays(x, tri)
    m.set_convex(False)
    assert not m.convex()
    # a grid triangulation: collinear boundary points are not reflex turns
    g = np.array([[i, j] for i in range(5) for j in range(5)], dtype=np.float64)
    gt = []
    for i in range(4):
        for j in range(4):
            a, b, c, e = 5 * i + j, 5 * i + j + 1, 5 * (i + 1) + j, 5 * (i + 1) + j + 1
            gt += [[a, c, b], [b, c, e]]
    assert pkg.SimplexMesh.from_arrays(g, np.array(gt, dtype=np.int32)).convex()


def test_import_rejects_bad_arrays(pkg, orc):
    x = orc.synth_centres(50, 2)
    _, tri, nbr = qhull(x)
    for bad_tri, bad_nbr in ((np.where(tri == 0, 99, tri), nbr),                       # vertex id out of range
                             (tri, np.where(nbr == nbr.max(), len(tri) + 3, nbr)),     # neighbour id out of range
                             (np.vstack([tri[0][[0, 0, 2]], tri[1:]]), None),          # repeated vertex
                             (tri, np.roll(nbr, 1, axis=1))):                          # links not mutual / wrong edge
        with pytest.raises(pkg.capi.GslError):
            pkg.SimplexMesh.from_arrays(x, bad_tri, bad_nbr)
    with pytest.raises(pkg.capi.GslError):
        pkg.SimplexMesh.from_arrays(x, np.vstack([tri, tri[:1], tri[:1]]))             # an edge shared by three triangles


def test_export_of_a_tree_keeps_leaves_and_links(pkg, orc):
    n = 400
    x = orc.synth_centres(n, 2)
    t = pkg.SimplexTree(2, n)
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    mesh = pkg.SimplexMesh.from_tree(t)
    types, pidx, links = t.arrays()
    sh = t.shuffle()
    nodes, tri, nbr = mesh.tree_nodes(), mesh.triangles(), mesh.neighbours()
    leaves_in_hull = [k for k in range(t.n_nodes) if types[k] == 0 and (pidx.reshape(-1, 3)[k] >= 0).all()]
    assert list(nodes) == leaves_in_hull                         # every leaf made of data points, in node order
    assert np.array_equal(tri, sh[pidx.reshape(-1, 3)[nodes]])   # same vertex ORDER, insertion index -> data row
    where = {int(k): i for i, k in enumerate(nodes)}
    want = np.array([[where.get(int(l), -1) if l > 0 else -1 for l in links.reshape(-1, 3)[k]] for k in nodes])
    assert np.array_equal(nbr, want)
    # Euler: a triangulation of n points with h >= 3 hull vertices has 2n - 2 - h triangles
    assert mesh.n_triangles <= 2 * n - 5
    shift, scale = mesh.geometry()
    assert np.array_equal(np.concatenate([shift, scale]), t.geom()[6:10])


def test_mesh_checkpoint_round_trip_and_validation(pkg, orc, tmp_path):
    """simplex_mesh_fwrite / _fread: arrays, geometry and the convexity flag come back; truncated, foreign and corrupted
    files (vertex id out of range, one-sided neighbour link) are refused, not crashed on."""
    x = orc.synth_centres(300, 2)
    _, tri, nbr = qhull(x)
    m = pkg.SimplexMesh.from_arrays(x, tri, nbr)
    path = tmp_path / "mesh.bin"
    assert m.fwrite(path) == 0
    r = pkg.SimplexMesh.fread(path)
    assert r is not None and r.n_triangles == m.n_triangles and r.convex() == m.convex()
    assert np.array_equal(r.triangles(), tri) and np.array_equal(r.neighbours(), nbr)
    sa, ca = m.geometry(); sb, cb = r.geometry()
    assert np.array_equal(sa, sb) and np.array_equal(ca, cb)
    blob = path.read_bytes()

    def load(b):
        p = tmp_path / "bad.bin"
        p.write_bytes(bytes(b))
        try:
            return pkg.SimplexMesh.fread(p)
        except pkg.capi.GslError:
            return None
    assert load(blob[: len(blob) // 2]) is None and load(b"nonsense" * 8) is None
    bad = bytearray(blob)
    off_tri = 8 + 32
    bad[off_tri:off_tri + 4] = (10 ** 6).to_bytes(4, "little")                    # vertex id out of range
    assert load(bad) is None
    bad = bytearray(blob)
    off_nbr = off_tri + 12 * len(tri)
    first = int.from_bytes(blob[off_nbr:off_nbr + 4], "little", signed=True)
    other = (first + 1) % len(tri) if first >= 0 else 0
    bad[off_nbr:off_nbr + 4] = other.to_bytes(4, "little", signed=True)            # a link its neighbour does not answer
    assert load(bad) is None
    t = pkg.SimplexTree(2, 300)
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    e = pkg.SimplexMesh.from_tree(t)
    assert e.fwrite(tmp_path / "exported.bin") == 0
    e2 = pkg.SimplexMesh.fread(tmp_path / "exported.bin")
    assert np.array_equal(e2.tree_nodes(), e.tree_nodes()) and np.array_equal(e2.triangles(), e.triangles())
