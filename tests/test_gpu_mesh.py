"""-m gpu: imported triangulations -- grid seed + walk over the leaf adjacency (csrc/hip/bary.hip, "Imported
triangulations").  PARITY UNPINNED for the locate step (the reference has no import, README:28-31); pinned here by
 (a) the DAG path itself: a mesh exported from a simplex_tree returns the leaf and the BITS the certified DAG walk /
     the oracle's reference walk return wherever the containing leaf is unique (random targets), and
 (b) the reference's per-triangle arithmetic restated on explicit vertices (oracle_mesh_*): the returned triangle
     contains the target under contains_point's closed rule and the value is interp_point's, bit for bit -- also for
     targets ON edges and vertices, where any containing triangle is a correct answer;
 (c) QHull output through scipy.spatial.Delaunay (a real import), checked the same way and against scipy's own
     linear interpolant."""
import numpy as np
import pytest

from gpu_util import bits

pytestmark = pytest.mark.gpu


def build_tree(pkg, orc, n, seed_shift=0.0):
    x = orc.synth_centres(n, 2) + seed_shift
    t = pkg.SimplexTree(2, n)
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    o = orc.Tree(2, n)
    assert o.init(x, flags=0, seed=0) == 0
    return x, t, o


@pytest.mark.parametrize("n,m", [(300, 5000), (20000, 200000)])
def test_exported_mesh_matches_the_dag_path_bitwise(pkg, orc, n, m):
    x, t, o = build_tree(pkg, orc, n)
    f = orc.synth_response(x)
    mesh = pkg.SimplexMesh.from_tree(t)
    nodes, tri = mesh.tree_nodes(), mesh.triangles()
    dev = mesh.device_alloc(0)
    assert dev.set_response(f) == 0
    y = orc.synth_targets(0, m, 2)
    st, vals, idx = dev.eval_many(y)
    ovals, oleaf = o.eval_many(x, f, y)                       # the reference walk (oracle)
    types, pidx, _ = t.arrays()
    in_hull = (pidx.reshape(-1, 3)[oleaf] >= 0).all(axis=1)   # leaves without cage vertices
    assert in_hull.sum() > 0.9 * m
    assert st == (pkg.GSL_SUCCESS if in_hull.all() else pkg.GSL_EDOM)
    assert (idx[~in_hull] == -1).all() and np.isnan(vals[~in_hull]).all()
    assert np.array_equal(nodes[idx[in_hull]], oleaf[in_hull])                         # same leaf ...
    assert np.array_equal(bits(vals[in_hull]), bits(ovals[in_hull]))                   # ... same bits
    # batch independence: a shuffled half of the batch gives the same bits per target
    p = np.random.default_rng(1).permutation(m)[: m // 2]
    st2, vals2, idx2 = dev.eval_many(np.ascontiguousarray(y[p]))
    assert np.array_equal(idx2, idx[p]) and np.array_equal(bits(vals2[in_hull[p]]), bits(vals[p][in_hull[p]]))


def test_targets_on_edges_and_vertices_follow_the_closed_rule(pkg, orc):
    n = 600
    x, t, o = build_tree(pkg, orc, n)
    f = orc.synth_response(x)
    mesh = pkg.SimplexMesh.from_tree(t)
    tri = mesh.triangles()
    shift, scale = mesh.geometry()
    dev = mesh.device_alloc(0)
    assert dev.set_response(f) == 0
    rng = np.random.default_rng(3)
    pick = rng.choice(len(tri), 400, replace=False)
    mids = 0.5 * (x[tri[pick, 0]] + x[tri[pick, 1]])           # edge midpoints (exact or one rounding away from the edge)
    verts = x[rng.choice(n, 300, replace=False)]                # data points: vertices of ~6 triangles each
    y = np.ascontiguousarray(np.vstack([mids, verts]))
    st, vals, idx = dev.eval_many(y)
    # A point within rounding of an edge can fail the floating-point closed test of BOTH triangles (the reference's
    # walk closes that gap with its "least violating child", linear_simplex.c:374-400); the mesh walk applies the same
    # rule: the returned triangle contains the target or misses it by rounding only (<= 1e-9 in barycentric units).
    assert (idx >= 0).all() and st == pkg.GSL_SUCCESS
    gaps = 0
    for k in range(len(y)):
        tk = tri[idx[k]]
        if not orc.mesh_contains(x, shift, scale, tk, y[k]):
            c = orc.mesh_coords(x, shift, scale, tk, y[k])
            c3 = np.array([c[0], c[1], 1.0 - (c[0] + c[1])])
            assert max((-c3).max(), (c3 - 1.0).max()) <= 1e-12
            gaps += 1
        assert vals[k] == orc.mesh_interp(x, shift, scale, tk, f, y[k])
    print(f"{gaps} of {len(y)} on-edge / on-vertex targets fall into a rounding gap")
    # a vertex target returns the vertex's datum (one coordinate is exactly 1 or the last weight is 1 - 0)
    got = vals[len(mids):]
    want = f[[int(np.argmin(((x - v) ** 2).sum(1))) for v in verts]]
    assert np.abs(got - want).max() <= 1e-13 * np.abs(f).max()


@pytest.mark.parametrize("n,m", [(2000, 40000)])
def test_qhull_import_through_scipy(pkg, orc, n, m):
    from scipy.interpolate import LinearNDInterpolator
    from scipy.spatial import Delaunay
    x = orc.synth_centres(n, 2) * [3.0, 0.5] + [10.0, -2.0]
    f = orc.synth_response(x)
    d = Delaunay(x)
    mesh = pkg.SimplexMesh.from_arrays(x, d.simplices, d.neighbors)
    shift, scale = mesh.geometry()
    tri = mesh.triangles()
    dev = mesh.device_alloc(0)
    assert dev.set_response(f) == 0
    y = np.ascontiguousarray(orc.synth_targets(0, m, 2) * [3.0, 0.5] + [10.0, -2.0])
    y[-20:] += 100.0                                           # far outside
    st, vals, idx = dev.eval_many(y)
    assert st == pkg.GSL_EDOM and (idx[-20:] == -1).all() and np.isnan(vals[-20:]).all()
    inside = idx >= 0
    sfind = d.find_simplex(y)
    assert np.array_equal(inside, sfind >= 0) or (inside != (sfind >= 0)).sum() <= 3      # hull-edge rounding
    ref = LinearNDInterpolator(d, f)(y)
    both = inside & (sfind >= 0)
    assert np.abs(vals[both] - ref[both]).max() <= 1e-12 * np.abs(f).max()
    for k in np.nonzero(inside)[0][::97]:                       # the reference's arithmetic, bit for bit
        assert orc.mesh_contains(x, shift, scale, tri[idx[k]], y[k])
        assert vals[k] == orc.mesh_interp(x, shift, scale, tri[idx[k]], f, y[k])
    for k in np.nonzero(inside)[0][::1999]:                     # exhaustive oracle search agrees that the triangle is unique
        first, cnt = orc.mesh_locate(x, shift, scale, tri, y[k])
        assert cnt >= 1 and (cnt > 1 or first == idx[k])


def test_nonconvex_mesh_and_resident_buffers(pkg, orc):
    """A mesh with a hole: the walk runs into hull edges although the target is inside another part; with
    convex = 0 such targets go to the exhaustive scan (smallest containing index).  Also the resident-buffer entry."""
    import torch
    from scipy.spatial import Delaunay
    n = 1500
    x = orc.synth_centres(n, 2)
    f = 2 * x[:, 0] - 3 * x[:, 1] + 0.5                         # linear: the interpolant reproduces it
    d = Delaunay(x)
    cen = x[d.simplices].mean(axis=1)
    keep = ~((np.abs(cen[:, 0] - 0.5) < 0.2) & (np.abs(cen[:, 1] - 0.5) < 0.2))      # punch a square hole
    tri = np.ascontiguousarray(d.simplices[keep].astype(np.int32))
    mesh = pkg.SimplexMesh.from_arrays(x, tri)                  # neighbours derived; hole edges become hull edges
    assert not mesh.convex()                                    # decided at import from the boundary loops
    shift, scale = mesh.geometry()
    dev = mesh.device_alloc(0)
    assert dev.set_response(f) == 0
    m = 20000
    y = orc.synth_targets(0, m, 2)
    ty = torch.from_numpy(y).cuda()
    tv = torch.empty(m, dtype=torch.float64, device="cuda")
    ti = torch.empty(m, dtype=torch.int32, device="cuda")
    assert dev.eval_resident(ty.data_ptr(), m, 2, tv.data_ptr(), ti.data_ptr()) == 0
    assert pkg.lib().gsl_sinterp_hip_sync(dev.ctx_handle()) == 0
    torch.cuda.synchronize()
    vals, idx = tv.cpu().numpy(), ti.cpu().numpy()
    inside = idx >= 0
    assert 0.7 * m < inside.sum() < m
    lin = 2 * y[:, 0] - 3 * y[:, 1] + 0.5
    assert np.abs(vals[inside] - lin[inside]).max() < 5e-15
    for k in range(0, m, 211):                                  # in or out, as the exhaustive oracle search says
        first, cnt = orc.mesh_locate(x, shift, scale, tri, y[k])
        assert (first >= 0) == bool(inside[k])
        if first >= 0:
            assert orc.mesh_contains(x, shift, scale, tri[idx[k]], y[k])


def test_mesh_facade_type_groups_and_checkpoint(pkg, orc, tmp_path):
    """Round 4: imported triangulations behind the gsl_sinterp facade (gsl_sinterp_linear_mesh + gsl_sinterp_set_triangulation),
    replicated over a device group ([0, 0, 0]: one broadcast of the raw arrays, every member packs its own records) and
    checkpointed; the sorted-target route (>= 4096 targets, two-level from 2^18) returns the bits of the per-target route."""
    from scipy.spatial import Delaunay
    n, m = 4000, 300_000
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    y = orc.synth_targets(0, m, 2)
    d = Delaunay(x)
    tri = d.simplices.astype(np.int32)
    mesh = pkg.SimplexMesh.from_arrays(x, tri, d.neighbors.astype(np.int32))
    dev = mesh.device_alloc(0)
    assert dev.set_response(f) == 0
    st0, v0, t0 = dev.eval_many(y)
    small = [dev.eval_many(np.ascontiguousarray(y[i:i + 3000])) for i in range(0, 30000, 3000)]     # < 4096: unsorted route
    assert all(s[0] in (0, pkg.GSL_EDOM) for s in small)
    assert np.array_equal(np.concatenate([s[1] for s in small]).view(np.uint64), v0[:30000].view(np.uint64))
    assert np.array_equal(np.concatenate([s[2] for s in small]), t0[:30000])
    grp = mesh.device_alloc_multi([0, 0, 0])
    assert grp.n_devices() == 3 and grp.set_response(f) == 0
    st1, v1, t1 = grp.eval_many(y)
    assert st1 == st0 and np.array_equal(v1.view(np.uint64), v0.view(np.uint64)) and np.array_equal(t1, t0)
    # the facade type
    s = pkg.Sinterp("linear_mesh", 2, n, 0)
    assert s.name() == "linear-imported-triangulation"
    assert s.init(x, f) == pkg.GSL_EINVAL                      # no triangulation yet
    assert s.set_triangulation(tri) == 0 and s.init(x, f) == 0
    st2, v2, t2 = s.eval_many(y, want_leaf=True)
    assert st2 == st0 and np.array_equal(v2.view(np.uint64), v0.view(np.uint64)) and np.array_equal(t2, t0)
    path = tmp_path / "mesh_interp.bin"
    assert s.fwrite(path) == 0
    r = pkg.Sinterp("linear_mesh", 2, n, 0)
    assert r.set_device_list([0, 0]) == 0 and r.fread(path) == 0
    st3, v3, t3 = r.eval_many(y, want_leaf=True)
    assert st3 == st0 and np.array_equal(v3.view(np.uint64), v0.view(np.uint64)) and np.array_equal(t3, t0)
    assert pkg.Sinterp("linear_simplex", 2, n, 0).fread(path) == pkg.capi.GSL_EBADLEN


def test_cyclic_non_delaunay_import_has_bounded_cost(pkg, orc):
    """A valid but badly shaped (non-Delaunay) triangulation can make the straight walk circle; such targets leave the walk
    after 64 + 4 G steps and are resolved by the exhaustive scan.  The cost must stay bounded: a fan of long slivers
    around one hub, 20 000 targets, finishes in well under a second and agrees with the exhaustive oracle search."""
    import time
    k = 720
    ang = np.linspace(0.0, 2.0 * np.pi, k, endpoint=False)
    ring = np.stack([np.cos(ang), np.sin(ang)], axis=1) * (1.0 + 0.3 * np.sin(7 * ang))[:, None]
    x = np.vstack([[0.0, 0.0], ring])
    tri = np.array([[0, 1 + i, 1 + (i + 1) % k] for i in range(k)], dtype=np.int32)       # slivers around the hub
    f = 1.0 + 2.0 * x[:, 0] - x[:, 1]
    mesh = pkg.SimplexMesh.from_arrays(x, tri)
    assert not mesh.convex()
    dev = mesh.device_alloc(0)
    assert dev.set_response(f) == 0
    rng = np.random.default_rng(5)
    y = rng.uniform(-0.6, 0.6, size=(20000, 2))
    dev.eval_many(y)                                             # warm-up (allocations)
    t0 = time.perf_counter()
    st, v, t = dev.eval_many(y)
    dt = time.perf_counter() - t0
    assert st in (0, pkg.GSL_EDOM) and dt < 1.0, dt
    inside = t >= 0
    assert inside.mean() > 0.95
    assert np.abs(v[inside] - (1.0 + 2.0 * y[inside, 0] - y[inside, 1])).max() < 1e-12
    shift, scale = mesh.geometry()
    for q in range(0, 20000, 997):
        first, cnt = orc.mesh_locate(x, shift, scale, tri, y[q])
        assert (first >= 0) == bool(inside[q])
