"""ctypes wrapper over oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_PD = C.POINTER(C.c_double)


class _Tree(C.Structure):
    _fields_ = [("dim", C.c_int), ("n_nodes", C.c_int), ("cap_nodes", C.c_int), ("type", C.POINTER(C.c_int)),
                ("pidx", C.POINTER(C.c_int)), ("links", C.POINTER(C.c_int)), ("seed", _PD),
                ("n_points", C.c_int), ("max_points", C.c_int), ("shift", _PD), ("scale", _PD), ("min", _PD),
                ("max", _PD), ("shuffle", C.POINTER(C.c_size_t)), ("acc_mat", _PD),
                ("acc_perm", C.POINTER(C.c_size_t)), ("acc_coords", _PD), ("acc_current", C.c_int),
                ("stat_tests", C.c_long), ("stat_depth", C.c_long), ("stat_maxdepth", C.c_long),
                ("stat_fallbacks", C.c_long)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("GSL_SINTERP_ORACLE_LIBRARY") or os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR])
        L = C.CDLL(path, mode=C.RTLD_LOCAL)
        L.oracle_tree_alloc.restype = C.POINTER(_Tree)
        L.oracle_mt_alloc.restype = C.c_void_p
        L.oracle_mt_get.restype = C.c_ulong
        L.oracle_mt_get.argtypes = [C.c_void_p]
        L.oracle_mt_uniform_int.restype = C.c_ulong
        L.oracle_mt_uniform_int.argtypes = [C.c_void_p, C.c_ulong]
        L.oracle_mt_free.argtypes = [C.c_void_p]
        L.oracle_interp_point.restype = C.c_double
        L.oracle_mesh_interp.restype = C.c_double
        L.oracle_tree_hash.restype = C.c_uint64
        L.oracle_rbf_phi.restype = C.c_double
        L.oracle_rbf_phi.argtypes = [C.c_int, C.c_double, C.c_double]
        L.oracle_splitmix64.restype = C.c_uint64
        L.oracle_splitmix64.argtypes = [C.c_uint64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_PD) if a is not None else None


def _sz(v):
    return C.c_size_t(int(v))


class Tree:
    def __init__(self, dim, n_points):
        self.t = lib().oracle_tree_alloc(dim, n_points)
        self.dim = dim

    @property
    def c(self):
        return self.t.contents

    def init(self, data=None, vmin=None, vmax=None, flags=0, seed=None):
        rng = lib().oracle_mt_alloc(C.c_ulong(seed)) if seed is not None else None
        n = data.shape[0] if data is not None else 0
        tda = data.strides[0] // 8 if data is not None else self.dim
        st = lib().oracle_tree_init(self.t, _p(data), _sz(n), _sz(tda), _p(vmin), _p(vmax), flags,
                                    C.c_void_p(rng) if rng else None)
        if rng:
            lib().oracle_mt_free(C.c_void_p(rng))
        return st

    def find_leaf(self, data, point):
        pt = np.ascontiguousarray(point, dtype=np.float64)
        tda = data.strides[0] // 8 if data is not None else self.dim
        return lib().oracle_find_leaf(self.t, _p(data), _sz(tda), _p(pt))

    def insert_point(self, leaf, data):
        return lib().oracle_insert_point(self.t, leaf, _p(data), _sz(data.strides[0] // 8))

    def in_hypersphere(self, node, data, idx):
        return lib().oracle_in_hypersphere(self.t, node, _p(data), _sz(data.strides[0] // 8), idx)

    def interp_point(self, leaf, data, response, point):
        pt = np.ascontiguousarray(point, dtype=np.float64)
        tda = data.strides[0] // 8 if data is not None else self.dim
        rs = response.strides[0] // 8 if response is not None else 1
        return lib().oracle_interp_point(self.t, leaf, _p(data), _sz(tda), _p(response), _sz(rs), _p(pt))

    def eval_many(self, data, response, targets):
        m = targets.shape[0]
        vals = np.empty(m)
        leaf = np.empty(m, dtype=np.int32)
        lib().oracle_bary_eval_many(self.t, _p(data), _sz(data.strides[0] // 8), _p(response),
                                    _sz(response.strides[0] // 8), _p(targets), _sz(m), _sz(targets.strides[0] // 8),
                                    _p(vals), leaf.ctypes.data_as(C.POINTER(C.c_int)))
        return vals, leaf

    def arrays(self):
        c = self.c
        n, w = c.n_nodes, self.dim + 1
        return (np.ctypeslib.as_array(c.type, (n,)).copy(), np.ctypeslib.as_array(c.pidx, (n * w,)).copy(),
                np.ctypeslib.as_array(c.links, (n * w,)).copy())

    def shuffle(self):
        c = self.c
        return np.ctypeslib.as_array(c.shuffle, (max(c.max_points, 1),)).copy()[: c.n_points]

    def geom(self):
        c, d = self.c, self.dim
        seed = np.ctypeslib.as_array(c.seed, ((d + 1) * d,)).copy()
        return np.concatenate([seed, [c.shift[i] for i in range(d)], [c.scale[i] for i in range(d)]])

    def vertices(self, node):
        c = self.c
        w = self.dim + 1
        return [c.pidx[node * w + i] for i in range(w)]

    def data_rows(self, node):
        c = self.c
        return [int(c.shuffle[v]) if v >= 0 else v for v in self.vertices(node)]

    def check_leaf_nodes(self):
        return lib().oracle_check_leaf_nodes(self.t)

    def check_delaunay(self, data):
        return lib().oracle_check_delaunay(self.t, _p(data), _sz(data.strides[0] // 8))

    def output_triangulation(self, data, response, standardize, lines=None, points=None, circles=None):
        enc = lambda p: str(p).encode() if p is not None else None
        rs = response.strides[0] // 8 if response is not None else 1
        return lib().oracle_output_triangulation(self.t, _p(data), _sz(data.strides[0] // 8), _p(response), _sz(rs),
                                                 int(standardize), enc(lines), enc(points), enc(circles))

    def hash(self):
        return lib().oracle_tree_hash(self.t)

    def reset_stats(self):
        c = self.c
        c.stat_tests = c.stat_depth = c.stat_maxdepth = c.stat_fallbacks = 0

    def __del__(self):
        try:
            lib().oracle_tree_free(self.t)
        except Exception:
            pass


# ---- dense linalg
def lu_decomp(a):
    a = np.array(a, dtype=np.float64, order="C")
    n = a.shape[0]
    perm = np.zeros(n, dtype=np.uintp)
    sg = C.c_int(0)
    lib().oracle_lu_decomp(_sz(n), _p(a), _sz(n), perm.ctypes.data_as(C.POINTER(C.c_size_t)), C.byref(sg))
    return a, perm, sg.value


def lu_solve(lu, perm, b):
    x = np.array(b, dtype=np.float64)
    st = lib().oracle_lu_svx(_sz(lu.shape[0]), _p(lu), _sz(lu.shape[1]), perm.ctypes.data_as(C.POINTER(C.c_size_t)), _p(x))
    return st, x


def cholesky_decomp1(a):
    a = np.array(a, dtype=np.float64, order="C")
    st = lib().oracle_cholesky_decomp1(_sz(a.shape[0]), _p(a), _sz(a.shape[1]))
    return st, a


def cholesky_solve(llt, b):
    x = np.array(b, dtype=np.float64)
    lib().oracle_cholesky_svx(_sz(llt.shape[0]), _p(llt), _sz(llt.shape[1]), _p(x))
    return x


def cholesky_decomp2(a):
    a = np.array(a, dtype=np.float64, order="C")
    s = np.empty(a.shape[0])
    st = lib().oracle_cholesky_decomp2(_sz(a.shape[0]), _p(a), _sz(a.shape[1]), _p(s))
    return st, a, s


def cholesky_solve2(llt, s, b):
    x = np.array(b, dtype=np.float64)
    lib().oracle_cholesky_svx2(_sz(llt.shape[0]), _p(llt), _sz(llt.shape[1]), _p(s), _p(x))
    return x


def cholesky_rcond(llt):
    n = llt.shape[0]
    r = C.c_double(0)
    work = np.zeros(3 * n)
    lib().oracle_cholesky_rcond(_sz(n), _p(llt), _sz(llt.shape[1]), C.byref(r), _p(work))
    return r.value


def lu_refine(a, lu, perm, b, x):
    x = np.array(x, dtype=np.float64)
    work = np.empty(len(x))
    st = lib().oracle_lu_refine(_sz(len(x)), _p(a), _sz(a.shape[1]), _p(lu), _sz(lu.shape[1]),
                                perm.ctypes.data_as(C.POINTER(C.c_size_t)), _p(np.ascontiguousarray(b)), _p(x), _p(work))
    return st, x


def pcholesky_decomp(a):
    a = np.array(a, dtype=np.float64, order="C")
    perm = np.zeros(a.shape[0], dtype=np.uintp)
    st = lib().oracle_pcholesky_decomp(_sz(a.shape[0]), _p(a), _sz(a.shape[1]), perm.ctypes.data_as(C.POINTER(C.c_size_t)))
    return st, a, perm


def pcholesky_solve(ldlt, perm, b):
    x = np.array(b, dtype=np.float64)
    lib().oracle_pcholesky_svx(_sz(ldlt.shape[0]), _p(ldlt), _sz(ldlt.shape[1]), perm.ctypes.data_as(C.POINTER(C.c_size_t)), _p(x))
    return x


def pcholesky_decomp2(a):
    a = np.array(a, dtype=np.float64, order="C")
    perm = np.zeros(a.shape[0], dtype=np.uintp)
    s = np.empty(a.shape[0])
    st = lib().oracle_pcholesky_decomp2(_sz(a.shape[0]), _p(a), _sz(a.shape[1]), perm.ctypes.data_as(C.POINTER(C.c_size_t)), _p(s))
    return st, a, perm, s


def pcholesky_solve2(ldlt, perm, s, b):
    x = np.array(b, dtype=np.float64)
    lib().oracle_pcholesky_svx2(_sz(ldlt.shape[0]), _p(ldlt), _sz(ldlt.shape[1]), perm.ctypes.data_as(C.POINTER(C.c_size_t)), _p(s), _p(x))
    return x


def pcholesky_rcond(ldlt, perm):
    n = ldlt.shape[0]
    r = C.c_double(0)
    work = np.zeros(3 * n)
    lib().oracle_pcholesky_rcond(_sz(n), _p(ldlt), _sz(ldlt.shape[1]), perm.ctypes.data_as(C.POINTER(C.c_size_t)), C.byref(r), _p(work))
    return r.value


# ---- RBF harness
def rbf_fill(kind, eps, x):
    n, d = x.shape
    phi = np.empty((n, n))
    lib().oracle_rbf_fill(kind, C.c_double(eps), _p(x), _sz(n), d, _sz(x.strides[0] // 8), _p(phi), _sz(n))
    return phi


def rbf_solve(kind, eps, x, f):
    n, d = x.shape
    w = np.empty(n)
    st = lib().oracle_rbf_solve(kind, C.c_double(eps), _p(x), _sz(n), d, _sz(x.strides[0] // 8), _p(f), _p(w))
    assert st == 0, st
    return w


def krige_solve(kind, eps, nugget, x, f):
    n, d = x.shape
    w = np.empty(n)
    mu = C.c_double(0)
    st = lib().oracle_krige_solve(kind, C.c_double(eps), C.c_double(nugget), _p(x), _sz(n), d, _sz(x.strides[0] // 8), _p(f), _p(w),
                                  C.byref(mu))
    assert st == 0, st
    return w, mu.value


def rbf_solve_affine(kind, eps, x, f):
    """thin-plate spline with its affine tail: weights w and c_0 .. c_dim (raw coordinates)"""
    n, d = x.shape
    w = np.empty(n)
    c = np.zeros(d + 1)
    st = lib().oracle_rbf_solve_affine(kind, C.c_double(eps), _p(x), _sz(n), d, _sz(x.strides[0] // 8), _p(f), _p(w), _p(c))
    assert st == 0, st
    return w, c


def rbf_eval_affine(kind, eps, c, x, w, y):
    n, d = x.shape
    m = y.shape[0]
    s = np.empty(m)
    c = np.ascontiguousarray(c, dtype=np.float64)
    lib().oracle_rbf_eval_affine(kind, C.c_double(eps), _p(c), _p(x), _sz(n), d, _sz(x.strides[0] // 8), _p(w), _p(y), _sz(m),
                                 _sz(y.strides[0] // 8), _p(s))
    return s


def krige_eval(kind, eps, mean, x, w, y):
    return rbf_eval(kind, eps, x, w, y) + mean


def rbf_eval(kind, eps, x, w, y):
    n, d = x.shape
    m = y.shape[0]
    s = np.empty(m)
    lib().oracle_rbf_eval(kind, C.c_double(eps), _p(x), _sz(n), d, _sz(x.strides[0] // 8), _p(w), _p(y), _sz(m),
                          _sz(y.strides[0] // 8), _p(s))
    return s


# ---- synthetic clouds
def synth_centres(n, dim):
    x = np.empty((n, dim))
    lib().oracle_synth_centres(_p(x), _sz(n), dim)
    return x


def synth_targets(first, m, dim):
    y = np.empty((m, dim))
    lib().oracle_synth_targets(_p(y), _sz(first), _sz(m), dim)
    return y


def synth_response(x):
    f = np.empty(x.shape[0])
    lib().oracle_synth_response(_p(x), _sz(x.shape[0]), x.shape[1], _p(f))
    return f


def gaussian_eps(n, dim):
    return 2.0 * n ** (1.0 / dim)


# ---- imported triangulations: the reference's per-triangle arithmetic on explicit vertex rows
def mesh_contains(data, shift, scale, tri, point):
    d = np.ascontiguousarray(data, dtype=np.float64)
    t = np.ascontiguousarray(tri, dtype=np.int32)
    return bool(lib().oracle_mesh_contains(_p(d), _sz(d.shape[1]), _p(np.ascontiguousarray(shift, dtype=np.float64)),
                                           _p(np.ascontiguousarray(scale, dtype=np.float64)),
                                           t.ctypes.data_as(C.POINTER(C.c_int)), _p(np.ascontiguousarray(point, dtype=np.float64))))


def mesh_coords(data, shift, scale, tri, point):
    d = np.ascontiguousarray(data, dtype=np.float64)
    t = np.ascontiguousarray(tri, dtype=np.int32)
    c = np.zeros(2)
    lib().oracle_mesh_coords(_p(d), _sz(d.shape[1]), _p(np.ascontiguousarray(shift, dtype=np.float64)),
                             _p(np.ascontiguousarray(scale, dtype=np.float64)), t.ctypes.data_as(C.POINTER(C.c_int)),
                             _p(np.ascontiguousarray(point, dtype=np.float64)), _p(c))
    return c


def mesh_interp(data, shift, scale, tri, response, point):
    d = np.ascontiguousarray(data, dtype=np.float64)
    t = np.ascontiguousarray(tri, dtype=np.int32)
    r = np.ascontiguousarray(response, dtype=np.float64)
    return float(lib().oracle_mesh_interp(_p(d), _sz(d.shape[1]), _p(np.ascontiguousarray(shift, dtype=np.float64)),
                                          _p(np.ascontiguousarray(scale, dtype=np.float64)),
                                          t.ctypes.data_as(C.POINTER(C.c_int)), _p(r), _sz(1),
                                          _p(np.ascontiguousarray(point, dtype=np.float64))))


def mesh_locate(data, shift, scale, tris, point):
    """(smallest containing triangle or -1, number of containing triangles) by exhaustive search"""
    d = np.ascontiguousarray(data, dtype=np.float64)
    t = np.ascontiguousarray(tris, dtype=np.int32)
    n = C.c_int(0)
    first = lib().oracle_mesh_locate(_p(d), _sz(d.shape[1]), _p(np.ascontiguousarray(shift, dtype=np.float64)),
                                     _p(np.ascontiguousarray(scale, dtype=np.float64)), t.ctypes.data_as(C.POINTER(C.c_int)),
                                     _sz(len(t)), _p(np.ascontiguousarray(point, dtype=np.float64)), C.byref(n))
    return int(first), n.value
