"""ctypes bindings of include/gsl_sinterp.h and include/gsl_sinterp_hip.h.

Mirrors the C structs (LP64) and wraps the entry points with numpy / torch
friendly helpers.  Device pointers are plain integers (``tensor.data_ptr()``).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

GSL_SUCCESS, GSL_FAILURE = 0, -1
GSL_EDOM, GSL_EFAULT, GSL_EINVAL, GSL_EFAILED, GSL_ENOMEM = 1, 3, 4, 5, 8
GSL_EBADLEN, GSL_ENOTSQR, GSL_EUNIMPL = 19, 20, 24
RBF_GAUSSIAN, RBF_TPS, RBF_WENDLAND = 0, 1, 2
SOLVER_DEFAULT, SOLVER_CHOLESKY2, SOLVER_PCHOLESKY, SOLVER_LU_REFINE = 0, 1, 2, 3
TREE_DEFAULT, TREE_NOSTANDARDIZE, TREE_ISOSCALE = 0, 1, 2
TREE_RECORD_BYTES, TREE_LEAFTAB_BYTES = 64, 32


def library_path():
    """The product library.  GSL_SINTERP_LIBRARY points the bindings at another BUILD of the same
    sources (the sanitizer build `make asan` of the host C, tests/test_sanitizers.py) -- never at a
    different implementation: there is no CPU fallback behind this switch."""
    return os.environ.get("GSL_SINTERP_LIBRARY") or os.path.join(_HERE, "libgsl_sinterp.so")


_lib = None


def lib():
    """Load libgsl_sinterp.so; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no pure-Python or CPU fallback)")
        _lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        _declare(_lib)
        _lib.gsl_set_error_handler_off()
    return _lib


# ------------------------------------------------------------------ structs
class gsl_block(C.Structure):
    _fields_ = [("size", C.c_size_t), ("data", C.POINTER(C.c_double))]


class gsl_vector(C.Structure):
    _fields_ = [("size", C.c_size_t), ("stride", C.c_size_t), ("data", C.POINTER(C.c_double)),
                ("block", C.POINTER(gsl_block)), ("owner", C.c_int)]


class gsl_matrix(C.Structure):
    _fields_ = [("size1", C.c_size_t), ("size2", C.c_size_t), ("tda", C.c_size_t),
                ("data", C.POINTER(C.c_double)), ("block", C.POINTER(gsl_block)), ("owner", C.c_int)]


class gsl_permutation(C.Structure):
    _fields_ = [("size", C.c_size_t), ("data", C.POINTER(C.c_size_t))]


class simplex_tree_node(C.Structure):
    _fields_ = [("points", C.c_int), ("links", C.c_int), ("type", C.c_uint, 2)]


class simplex_tree_accel(C.Structure):
    _fields_ = [("simplex_matrix", C.POINTER(gsl_matrix)), ("perm", C.POINTER(gsl_permutation)),
                ("coords", C.POINTER(gsl_vector)), ("current_simplex", C.c_int)]


class simplex_tree(C.Structure):
    _fields_ = [
        ("n_simplexes", C.c_int), ("max_simplexes", C.c_int), ("simplexes", C.POINTER(simplex_tree_node)),
        ("n_pidx", C.c_int), ("max_pidx", C.c_int), ("pidx", C.POINTER(C.c_int)),
        ("n_links", C.c_int), ("max_links", C.c_int), ("links", C.POINTER(C.c_int)),
        ("seed_points", C.POINTER(gsl_matrix)), ("n_points", C.c_int), ("max_points", C.c_int), ("dim", C.c_int),
        ("shift", C.POINTER(gsl_vector)), ("scale", C.POINTER(gsl_vector)),
        ("min", C.POINTER(gsl_vector)), ("max", C.POINTER(gsl_vector)),
        ("shuffle", C.POINTER(gsl_permutation)), ("accel", C.POINTER(simplex_tree_accel)),
        ("new_simplexes", C.POINTER(C.c_int)), ("old_neighbors1", C.POINTER(C.c_int)),
        ("old_neighbors2", C.POINTER(C.c_int)), ("left_out", C.POINTER(C.c_int)),
        ("tmp_points1", C.POINTER(C.c_int)), ("tmp_vec1", C.POINTER(gsl_vector)),
        ("tmp_vec2", C.POINTER(gsl_vector)), ("tmp_mat", C.POINTER(gsl_matrix)),
    ]


class gsl_sinterp(C.Structure):
    _fields_ = [("type", C.c_void_p), ("dim", C.c_size_t), ("size", C.c_size_t), ("device", C.c_int),
                ("shape", C.c_double), ("init_flags", C.c_int), ("rng", C.c_void_p), ("state", C.c_void_p),
                ("n_devices", C.c_int), ("devices", C.c_int * 64), ("solver", C.c_int), ("want_rcond", C.c_int),
                ("rcond", C.c_double), ("route", C.c_int), ("nugget", C.c_double)]


_vp, _i, _sz, _d = C.c_void_p, C.c_int, C.c_size_t, C.c_double
_pd, _pi = C.POINTER(C.c_double), C.POINTER(C.c_int)
_pm, _pv, _pt = C.POINTER(gsl_matrix), C.POINTER(gsl_vector), C.POINTER(simplex_tree)

# name -> (restype, argtypes); this table is also what tests check the .so exports
SIGNATURES = {
    # --- include/gsl_sinterp_hip.h
    "gsl_sinterp_hip_device_count": (_i, []),
    "gsl_sinterp_hip_ctx_create": (_i, [C.POINTER(_vp), _i, _vp]),
    "gsl_sinterp_hip_ctx_own_stream": (_i, [_vp]),
    "gsl_sinterp_hip_ctx_destroy": (None, [_vp]),
    "gsl_sinterp_hip_sync": (_i, [_vp]),
    "gsl_sinterp_hip_last_error": (C.c_char_p, [_vp]),
    "gsl_sinterp_hip_malloc": (_i, [_vp, C.POINTER(_vp), _sz]),
    "gsl_sinterp_hip_free": (_i, [_vp, _vp]),
    "gsl_sinterp_hip_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "gsl_sinterp_hip_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "gsl_sinterp_hip_timer_start": (_i, [_vp]),
    "gsl_sinterp_hip_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "gsl_sinterp_hip_tree_pack": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _pd, _vp]),
    "gsl_sinterp_hip_tree_bind": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "gsl_sinterp_hip_bary_eval": (_i, [_vp, _i, _vp, _vp, _pd, _vp, _sz, _sz, _vp, _vp, C.POINTER(C.c_longlong)]),
    "gsl_sinterp_hip_tree_check": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _pd, _i, C.POINTER(C.c_longlong),
                                        C.POINTER(C.c_longlong), _pi]),
    "gsl_sinterp_hip_rbf_fill": (_i, [_vp, _i, _d, _vp, _sz, _i, _sz, _vp, _sz]),
    "gsl_sinterp_hip_cholesky_decomp1": (_i, [_vp, _sz, _vp, _sz, _pi]),
    "gsl_sinterp_hip_cholesky_svx": (_i, [_vp, _sz, _vp, _sz, _vp]),
    "gsl_sinterp_hip_lu_decomp": (_i, [_vp, _sz, _vp, _sz, _vp, _pi]),
    "gsl_sinterp_hip_lu_svx": (_i, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "gsl_sinterp_hip_cholesky_decomp2": (_i, [_vp, _sz, _vp, _sz, _vp, _pi]),
    "gsl_sinterp_hip_cholesky_svx2": (_i, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "gsl_sinterp_hip_cholesky_rcond": (_i, [_vp, _sz, _vp, _sz, _pd]),
    "gsl_sinterp_hip_lu_refine": (_i, [_vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp]),
    "gsl_sinterp_hip_pcholesky_decomp": (_i, [_vp, _sz, _vp, _sz, _vp]),
    "gsl_sinterp_hip_pcholesky_svx": (_i, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "gsl_sinterp_hip_pcholesky_decomp2": (_i, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "gsl_sinterp_hip_pcholesky_svx2": (_i, [_vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "gsl_sinterp_hip_pcholesky_rcond": (_i, [_vp, _sz, _vp, _sz, _vp, _pd]),
    "gsl_sinterp_hip_rbf_solve_ex": (_i, [_vp, _i, _d, _vp, _sz, _i, _sz, _vp, _sz, _vp, _i, _pd, _pi]),
    "gsl_sinterp_hip_rbf_eval": (_i, [_vp, _i, _d, _vp, _sz, _i, _sz, _vp, _vp, _sz, _sz, _vp]),
    "gsl_sinterp_hip_rbf_eval_model": (_i, [_vp, _i, _d, _vp, _sz, _i, _sz, _vp, _vp, _sz, _sz, _vp, C.c_uint64]),
    "gsl_sinterp_hip_ctx_device": (_i, [_vp]),
    "gsl_sinterp_hip_rbf_solve": (_i, [_vp, _i, _d, _vp, _sz, _i, _sz, _vp, _sz, _vp, _pi]),
    "gsl_sinterp_hip_gemm_minus": (_i, [_vp, _sz, _sz, _sz, _vp, _sz, _vp, _sz, _i, _vp, _sz, _i]),
    "gsl_sinterp_hip_grid_targets": (_i, [_vp, _d, _d, _sz, _d, _d, _sz, _vp]),
    "gsl_sinterp_hip_synth_unit": (_i, [_vp, C.c_uint64, C.c_uint64, _d, _d, _vp, _sz]),
    # device groups (multi-GPU target shards)
    "gsl_sinterp_hip_group_create": (_i, [C.POINTER(_vp), _pi, _i]),
    "gsl_sinterp_hip_group_destroy": (None, [_vp]),
    "gsl_sinterp_hip_group_size": (_i, [_vp]),
    "gsl_sinterp_hip_group_device": (_i, [_vp, _i]),
    "gsl_sinterp_hip_group_ctx": (_vp, [_vp, _i]),
    "gsl_sinterp_hip_group_transport": (C.c_char_p, [_vp]),
    "gsl_sinterp_hip_group_last_error": (C.c_char_p, [_vp]),
    "gsl_sinterp_hip_group_broadcast": (_i, [_vp, C.POINTER(_vp), _sz]),
    "gsl_sinterp_hip_shard_bounds": (None, [_sz, _i, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "gsl_sinterp_hip_h2d_async": (_i, [_vp, _vp, _vp, _sz]),
    "gsl_sinterp_hip_d2h_async": (_i, [_vp, _vp, _vp, _sz]),
    "gsl_sinterp_hip_host_alloc": (_i, [C.POINTER(_vp), _sz]),
    "gsl_sinterp_hip_host_free": (None, [_vp]),
    # --- include/gsl_sinterp.h part 1 (reference symbols)
    "simplex_tree_node_alloc": (_i, [_pt]),
    "simplex_tree_alloc": (_pt, [_i, _i]),
    "simplex_tree_init": (_i, [_pt, _pm, _pv, _pv, _i, _vp]),
    "simplex_tree_free": (None, [_pt]),
    "simplex_tree_accel_alloc": (_vp, [_i]),
    "simplex_tree_accel_free": (None, [_vp]),
    "point_in_simplex": (_i, [_pt, _i, _i]),
    "find_leaf": (_i, [_pt, _pm, _pv, _vp]),
    "_find_leaf": (_i, [_pt, _i, _pm, _pv, _vp]),
    "insert_point": (_i, [_pt, _i, _pm, _pv, _vp]),
    "in_hypersphere": (_i, [_pt, _i, _pm, _i, _vp]),
    "in_hypersphere_points": (_i, [_pt, _pi, _pm, _i, _vp]),
    "calculate_hypersphere": (_i, [_pt, _i, _pm, _pv, _pd, _vp]),
    "calculate_hypersphere_points": (_i, [_pt, _pi, _pm, _pv, _pd, _vp]),
    "calculate_bary_coords": (_i, [_pt, _i, _pm, _pv, _vp]),
    "contains_point": (_i, [_pt, _i, _pm, _pv, _vp]),
    "interp_point": (_d, [_pt, _i, _pm, _pv, _pv, _vp]),
    "delaunay": (_i, [_pt, _i, _pm, _i, _vp]),
    # --- part 2
    "simplex_tree_device_alloc": (_vp, [_pt, _pm, _i]),
    "simplex_tree_device_free": (None, [_vp]),
    "simplex_tree_device_set_response": (_i, [_vp, _pv]),
    "simplex_tree_device_eval_many": (_i, [_vp, _pm, _pv, _pi]),
    "simplex_tree_device_eval_resident": (_i, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "simplex_tree_device_ctx": (_vp, [_vp]),
    "check_leaf_nodes": (None, [_pt, _vp]),
    "check_delaunay": (_i, [_pt, _pm]),
    "output_triangulation": (None, [_pt, _pm, _pv, _i, C.c_char_p, C.c_char_p, C.c_char_p]),
    "simplex_tree_fwrite": (_i, [_vp, _pt]),
    "simplex_tree_fread": (_pt, [_vp]),
    "simplex_tree_check_device": (_i, [_pt, _pm, _i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "simplex_tree_device_alloc_multi": (_vp, [_pt, _pm, _pi, _i]),
    "simplex_tree_device_n_devices": (_i, [_vp]),
    "simplex_tree_device_transport": (C.c_char_p, [_vp]),
    # --- part 2b: imported triangulations
    "simplex_mesh_import": (_vp, [_pm, _pi, _pi, _sz]),
    "simplex_mesh_from_tree": (_vp, [_pt, _pm]),
    "simplex_mesh_free": (None, [_vp]),
    "simplex_mesh_n_triangles": (_sz, [_vp]),
    "simplex_mesh_n_points": (_sz, [_vp]),
    "simplex_mesh_triangles": (_pi, [_vp]),
    "simplex_mesh_neighbours": (_pi, [_vp]),
    "simplex_mesh_tree_nodes": (_pi, [_vp]),
    "simplex_mesh_geometry": (None, [_vp, _pd, _pd]),
    "simplex_mesh_set_convex": (None, [_vp, _i]),
    "simplex_mesh_convex": (_i, [_vp]),
    "simplex_mesh_device_alloc": (_vp, [_vp, _i]),
    "simplex_mesh_device_alloc_multi": (_vp, [_vp, _pi, _i]),
    "simplex_mesh_device_n_devices": (_i, [_vp]),
    "simplex_mesh_fwrite": (_i, [_vp, _vp]),
    "simplex_mesh_fread": (_vp, [_vp]),
    "gsl_sinterp_set_triangulation": (_i, [C.POINTER(gsl_sinterp), _pi, _pi, _sz]),
    "simplex_mesh_device_free": (None, [_vp]),
    "simplex_mesh_device_set_response": (_i, [_vp, _pv]),
    "simplex_mesh_device_eval_many": (_i, [_vp, _pm, _pv, _pi]),
    "simplex_mesh_device_eval_resident": (_i, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "simplex_mesh_device_ctx": (_vp, [_vp]),
    "gsl_sinterp_hip_mesh_pack": (_i, [_vp, _i, _vp, _vp, _i, _vp, _pd, _i, _vp, _vp]),
    "gsl_sinterp_hip_mesh_eval": (_i, [_vp, _i, _vp, _vp, _vp, _i, _pd, _i, _vp, _sz, _sz, _vp, _vp, C.POINTER(C.c_longlong)]),
    # --- part 3
    "gsl_sinterp_alloc": (C.POINTER(gsl_sinterp), [_vp, _sz, _sz]),
    "gsl_sinterp_set_device": (_i, [C.POINTER(gsl_sinterp), _i]),
    "gsl_sinterp_set_shape": (_i, [C.POINTER(gsl_sinterp), _d]),
    "gsl_sinterp_set_solver": (_i, [C.POINTER(gsl_sinterp), _i]),
    "gsl_sinterp_set_nugget": (_i, [C.POINTER(gsl_sinterp), _d]),
    "gsl_sinterp_mean": (_i, [C.POINTER(gsl_sinterp), _pd]),
    "gsl_sinterp_poly": (_i, [C.POINTER(gsl_sinterp), _pv]),
    "gsl_sinterp_hip_pipe_create": (_i, [_vp, C.POINTER(_vp)]),
    "gsl_sinterp_hip_pipe_destroy": (None, [_vp]),
    "gsl_sinterp_hip_pipe_upload": (_i, [_vp, _vp, _vp, _sz]),
    "gsl_sinterp_hip_pipe_mark": (_i, [_vp, _pi]),
    "gsl_sinterp_hip_pipe_download": (_i, [_vp, _i, _vp, _vp, _sz]),
    "gsl_sinterp_hip_pipe_sync": (_i, [_vp]),
    "gsl_sinterp_hip_bary_last_queue": (_i, [_vp, C.POINTER(C.c_uint), _pi]),
    "gsl_sinterp_hip_count_negative": (_i, [_vp, _vp, _sz, C.POINTER(C.c_longlong)]),
    "gsl_sinterp_hip_rbf_solve_affine": (_i, [_vp, _i, _d, _vp, _sz, _i, _sz, _vp, _sz, _vp, _pd, _pi]),
    "gsl_sinterp_hip_rbf_eval_affine": (_i, [_vp, _i, _d, _pd, _vp, _sz, _i, _sz, _vp, _vp, _sz, _sz, _vp, C.c_uint64]),
    "gsl_sinterp_hip_krige_solve": (_i, [_vp, _i, _d, _d, _vp, _sz, _i, _sz, _vp, _sz, _vp, _pd, _pi]),
    "gsl_sinterp_hip_krige_eval": (_i, [_vp, _i, _d, _d, _vp, _sz, _i, _sz, _vp, _vp, _sz, _sz, _vp, C.c_uint64]),
    "gsl_sinterp_set_rcond": (_i, [C.POINTER(gsl_sinterp), _i]),
    "gsl_sinterp_rcond": (_i, [C.POINTER(gsl_sinterp), _pd]),
    "gsl_sinterp_route": (_i, [C.POINTER(gsl_sinterp)]),
    "gsl_sinterp_set_devices": (_i, [C.POINTER(gsl_sinterp), _i]),
    "gsl_sinterp_set_device_list": (_i, [C.POINTER(gsl_sinterp), _pi, _i]),
    "gsl_sinterp_n_devices": (_i, [C.POINTER(gsl_sinterp)]),
    "gsl_sinterp_set_tree_options": (_i, [C.POINTER(gsl_sinterp), _i, _vp]),
    "gsl_sinterp_init": (_i, [C.POINTER(gsl_sinterp), _pm, _pv]),
    "gsl_sinterp_name": (C.c_char_p, [C.POINTER(gsl_sinterp)]),
    "gsl_sinterp_min_size": (C.c_uint, [C.POINTER(gsl_sinterp)]),
    "gsl_sinterp_eval_e": (_i, [C.POINTER(gsl_sinterp), _pv, _pd]),
    "gsl_sinterp_eval": (_d, [C.POINTER(gsl_sinterp), _pv]),
    "gsl_sinterp_eval_many": (_i, [C.POINTER(gsl_sinterp), _pm, _pv, _pi]),
    "gsl_sinterp_eval_resident": (_i, [C.POINTER(gsl_sinterp), _vp, _sz, _sz, _vp, _vp]),
    "gsl_sinterp_eval_grid": (_i, [C.POINTER(gsl_sinterp), _pv, _pv, _pm]),
    "gsl_sinterp_fprintf_grid": (_i, [_vp, _pv, _pv, _pm]),
    "gsl_sinterp_fwrite": (_i, [_vp, C.POINTER(gsl_sinterp)]),
    "gsl_sinterp_fread": (_i, [_vp, C.POINTER(gsl_sinterp)]),
    "gsl_sinterp_get_weights": (_i, [C.POINTER(gsl_sinterp), _pv]),
    "gsl_sinterp_free": (None, [C.POINTER(gsl_sinterp)]),
    # --- compat slice used by the bindings
    "gsl_set_error_handler_off": (_vp, []),
    "gsl_rng_alloc": (_vp, [_vp]),
    "gsl_rng_set": (None, [_vp, C.c_ulong]),
    "gsl_rng_free": (None, [_vp]),
    "gsl_rng_get": (C.c_ulong, [_vp]),
    "gsl_rng_uniform_int": (C.c_ulong, [_vp, C.c_ulong]),
}
DATA_SYMBOLS = ["gsl_sinterp_kriging", "gsl_sinterp_linear_mesh", "gsl_sinterp_rbf_gaussian", "gsl_sinterp_rbf_tps", "gsl_sinterp_rbf_tps_affine", "gsl_sinterp_rbf_wendland", "gsl_sinterp_linear_simplex",
                "gsl_rng_mt19937", "gsl_rng_default"]


def _declare(L):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


def _ptr(symbol):
    """value of an exported `const T *symbol` variable"""
    return C.c_void_p.in_dll(lib(), symbol).value


# ------------------------------------------------------------------ C stdio (FILE * arguments)
_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]


class CFile:
    """`with CFile(path, "wb") as fp:` -> a FILE * for the library's fwrite / fread / fprintf entries."""

    def __init__(self, path, mode):
        self.fp = _libc.fopen(str(path).encode(), mode.encode())
        if not self.fp:
            raise OSError(f"fopen({path!r}, {mode!r}) failed")

    def __enter__(self):
        return C.c_void_p(self.fp)

    def __exit__(self, *exc):
        _libc.fclose(C.c_void_p(self.fp))
        self.fp = None


# ------------------------------------------------------------------ views
def as_matrix(a):
    """gsl_matrix view of a 2-D float64 numpy array (row stride honoured)."""
    assert a.dtype == np.float64 and a.ndim == 2 and (a.size == 0 or a.strides[1] == 8)
    m = gsl_matrix()
    m.size1, m.size2 = a.shape[0], a.shape[1]
    m.tda = a.strides[0] // 8 if a.shape[0] > 1 and a.size else max(a.shape[1], 1)
    m.data = a.ctypes.data_as(_pd)
    m.block, m.owner = None, 0
    m._keep = a
    return m


def as_vector(a):
    assert a.dtype == np.float64 and a.ndim == 1
    v = gsl_vector()
    v.size, v.stride = a.shape[0], (a.strides[0] // 8 if a.shape[0] > 1 else 1)
    v.data = a.ctypes.data_as(_pd)
    v.block, v.owner = None, 0
    v._keep = a
    return v


class GslError(RuntimeError):
    def __init__(self, status, what=""):
        super().__init__(f"GSL status {status} {what}")
        self.status = status


def check(status, ctx=None):
    if status != GSL_SUCCESS:
        msg = lib().gsl_sinterp_hip_last_error(ctx).decode() if ctx else ""
        raise GslError(status, msg)


# ------------------------------------------------------------------ HIP context
class HipContext:
    """gsl_sinterp_hip_ctx bound to one device and (optionally) torch's current stream."""

    def __init__(self, device=0, stream=None):
        L = lib()
        if L.gsl_sinterp_hip_device_count() <= 0:
            raise RuntimeError("no HIP device visible: the gfx950 kernels cannot run (no CPU fallback exists)")
        self._h = C.c_void_p()
        st = L.gsl_sinterp_hip_ctx_create(C.byref(self._h), device, stream)
        if st != GSL_SUCCESS:
            raise GslError(st, "gsl_sinterp_hip_ctx_create")
        self.device = device

    @classmethod
    def on_torch_stream(cls, device=0):
        import torch
        torch.cuda.set_device(device)
        return cls(device, C.c_void_p(torch.cuda.current_stream(device).cuda_stream))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            lib().gsl_sinterp_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib().gsl_sinterp_hip_sync(self._h), self._h)

    # thin wrappers: every pointer argument is an int device address
    def tree_pack(self, n_nodes, d_type, d_pidx, d_links, n_points, d_points, geom, d_records):
        g = np.ascontiguousarray(geom, dtype=np.float64)
        check(lib().gsl_sinterp_hip_tree_pack(self._h, n_nodes, d_type, d_pidx, d_links, n_points, d_points,
                                              g.ctypes.data_as(_pd), d_records), self._h)

    def tree_bind(self, n_nodes, d_pidx, n_points, d_response, d_leaftab):
        check(lib().gsl_sinterp_hip_tree_bind(self._h, n_nodes, d_pidx, n_points, d_response, d_leaftab), self._h)

    def bary_eval(self, n_nodes, d_records, d_leaftab, scale, d_targets, m, ttda, d_values, d_leaf=None,
                  count_outside=False):
        s = np.ascontiguousarray(scale, dtype=np.float64)
        cnt = C.c_longlong(0)
        st = lib().gsl_sinterp_hip_bary_eval(self._h, n_nodes, d_records, d_leaftab, s.ctypes.data_as(_pd), d_targets,
                                             m, ttda, d_values, d_leaf, C.byref(cnt) if count_outside else None)
        if st not in (GSL_SUCCESS, GSL_EDOM):
            check(st, self._h)
        return cnt.value

    def rbf_fill(self, kind, eps, d_x, n, dim, xtda, d_phi, lda):
        check(lib().gsl_sinterp_hip_rbf_fill(self._h, kind, eps, d_x, n, dim, xtda, d_phi, lda), self._h)

    def cholesky_decomp1(self, n, d_a, lda):
        info = C.c_int(0)
        st = lib().gsl_sinterp_hip_cholesky_decomp1(self._h, n, d_a, lda, C.byref(info))
        return st, info.value

    def cholesky_svx(self, n, d_llt, lda, d_x):
        check(lib().gsl_sinterp_hip_cholesky_svx(self._h, n, d_llt, lda, d_x), self._h)

    def lu_decomp(self, n, d_a, lda, d_perm):
        sg = C.c_int(0)
        check(lib().gsl_sinterp_hip_lu_decomp(self._h, n, d_a, lda, d_perm, C.byref(sg)), self._h)
        return sg.value

    def lu_svx(self, n, d_lu, lda, d_perm, d_x):
        return lib().gsl_sinterp_hip_lu_svx(self._h, n, d_lu, lda, d_perm, d_x)

    def rbf_eval(self, kind, eps, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, model_id=0):
        """model_id != 0: the caller vouches that (d_x, d_w) do not change while it uses this id, so the sweep's
        per-model preprocessing is cached in the context (gsl_sinterp_hip_rbf_eval_model)"""
        check(lib().gsl_sinterp_hip_rbf_eval_model(self._h, kind, eps, d_x, n, dim, xtda, d_w, d_y, m, ytda, d_s, model_id), self._h)

    def device(self):
        return lib().gsl_sinterp_hip_ctx_device(self._h)

    def cholesky_decomp2(self, n, d_a, lda, d_s):
        info = C.c_int(0)
        st = lib().gsl_sinterp_hip_cholesky_decomp2(self._h, n, d_a, lda, d_s, C.byref(info))
        return st, info.value

    def cholesky_svx2(self, n, d_llt, lda, d_s, d_x):
        check(lib().gsl_sinterp_hip_cholesky_svx2(self._h, n, d_llt, lda, d_s, d_x), self._h)

    def cholesky_rcond(self, n, d_llt, lda):
        r = C.c_double(0)
        check(lib().gsl_sinterp_hip_cholesky_rcond(self._h, n, d_llt, lda, C.byref(r)), self._h)
        return r.value

    def lu_refine(self, n, d_a, lda, d_lu, ldlu, d_perm, d_b, d_x, d_work):
        return lib().gsl_sinterp_hip_lu_refine(self._h, n, d_a, lda, d_lu, ldlu, d_perm, d_b, d_x, d_work)

    def pcholesky_decomp(self, n, d_a, lda, d_perm):
        check(lib().gsl_sinterp_hip_pcholesky_decomp(self._h, n, d_a, lda, d_perm), self._h)

    def pcholesky_svx(self, n, d_ldlt, lda, d_perm, d_x):
        check(lib().gsl_sinterp_hip_pcholesky_svx(self._h, n, d_ldlt, lda, d_perm, d_x), self._h)

    def pcholesky_decomp2(self, n, d_a, lda, d_perm, d_s):
        check(lib().gsl_sinterp_hip_pcholesky_decomp2(self._h, n, d_a, lda, d_perm, d_s), self._h)

    def pcholesky_svx2(self, n, d_ldlt, lda, d_perm, d_s, d_x):
        check(lib().gsl_sinterp_hip_pcholesky_svx2(self._h, n, d_ldlt, lda, d_perm, d_s, d_x), self._h)

    def pcholesky_rcond(self, n, d_ldlt, lda, d_perm):
        r = C.c_double(0)
        check(lib().gsl_sinterp_hip_pcholesky_rcond(self._h, n, d_ldlt, lda, d_perm, C.byref(r)), self._h)
        return r.value

    def rbf_solve_ex(self, kind, eps, d_x, n, dim, xtda, d_phi, lda, d_w, solver, want_rcond=False):
        route, rc = C.c_int(0), C.c_double(0)
        st = lib().gsl_sinterp_hip_rbf_solve_ex(self._h, kind, eps, d_x, n, dim, xtda, d_phi, lda, d_w, solver,
                                                C.byref(rc) if want_rcond else None, C.byref(route))
        return st, route.value, rc.value

    def rbf_solve(self, kind, eps, d_x, n, dim, xtda, d_phi, lda, d_w):
        route = C.c_int(0)
        st = lib().gsl_sinterp_hip_rbf_solve(self._h, kind, eps, d_x, n, dim, xtda, d_phi, lda, d_w, C.byref(route))
        return st, route.value

    def gemm_minus(self, m, n, k, d_a, lda, d_b, ldb, b_is_kn, d_c, ldc, lower_only=0):
        check(lib().gsl_sinterp_hip_gemm_minus(self._h, m, n, k, d_a, lda, d_b, ldb, b_is_kn, d_c, ldc, lower_only),
              self._h)

    def synth_unit(self, seed, first, offset, span, d_out, count):
        check(lib().gsl_sinterp_hip_synth_unit(self._h, seed, first, offset, span, d_out, count), self._h)

    def timer_start(self):
        check(lib().gsl_sinterp_hip_timer_start(self._h), self._h)

    def timer_stop(self):
        ms = C.c_float(0)
        check(lib().gsl_sinterp_hip_timer_stop(self._h, C.byref(ms)), self._h)
        return ms.value


# ------------------------------------------------------------------ host tree
class Rng:
    """gsl_rng (mt19937) from the library's compat slice."""

    def __init__(self, seed=0):
        self._h = lib().gsl_rng_alloc(_ptr("gsl_rng_mt19937"))
        lib().gsl_rng_set(self._h, seed)

    def close(self):
        if self._h:
            lib().gsl_rng_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SimplexTree:
    """The reference's simplex_tree_* API (host, 2-D) over numpy arrays."""

    def __init__(self, dim, n_points):
        self._t = lib().simplex_tree_alloc(dim, n_points)
        if not self._t:
            raise GslError(GSL_EUNIMPL, "simplex_tree_alloc")
        self.data = None
        self._mat = None

    def init(self, data=None, vmin=None, vmax=None, flags=0, rng=None):
        self.data = data
        self._mat = as_matrix(data) if data is not None else None
        mn = as_vector(np.ascontiguousarray(vmin, dtype=np.float64)) if vmin is not None else None
        mx = as_vector(np.ascontiguousarray(vmax, dtype=np.float64)) if vmax is not None else None
        return lib().simplex_tree_init(self._t, C.byref(self._mat) if self._mat is not None else None,
                                       C.byref(mn) if mn is not None else None,
                                       C.byref(mx) if mx is not None else None, flags,
                                       rng._h if rng is not None else None)

    def set_data(self, data):
        self.data = data
        self._mat = as_matrix(data)

    @property
    def c(self):
        return self._t.contents

    def _m(self):
        return C.byref(self._mat) if self._mat is not None else None

    def find_leaf(self, point):
        p = np.ascontiguousarray(point, dtype=np.float64)
        return lib().find_leaf(self._t, self._m(), C.byref(as_vector(p)), None)

    def insert_point(self, leaf):
        return lib().insert_point(self._t, leaf, self._m(), None, None)

    def in_hypersphere(self, node, idx):
        return lib().in_hypersphere(self._t, node, self._m(), idx, None)

    def contains_point(self, node, point):
        p = np.ascontiguousarray(point, dtype=np.float64)
        return lib().contains_point(self._t, node, self._m(), C.byref(as_vector(p)), None)

    def interp_point(self, leaf, response, point):
        p = np.ascontiguousarray(point, dtype=np.float64)
        r = as_vector(response) if response is not None else None
        return lib().interp_point(self._t, leaf, self._m(), C.byref(r) if r is not None else None,
                                  C.byref(as_vector(p)), None)

    # flat copies of the DAG arrays
    @property
    def n_nodes(self):
        return self.c.n_simplexes

    def arrays(self):
        t = self.c
        n = t.n_simplexes
        nodes = np.ctypeslib.as_array(C.cast(t.simplexes, C.POINTER(C.c_int)), (n, 3)).copy()
        types = (nodes[:, 2] & 3).astype(np.int32)
        pidx = np.ctypeslib.as_array(t.pidx, (3 * n,)).copy().astype(np.int32)
        links = np.ctypeslib.as_array(t.links, (3 * n,)).copy().astype(np.int32)
        assert (nodes[:, 0] == 3 * np.arange(n)).all() and (nodes[:, 1] == 3 * np.arange(n)).all()
        return types, pidx, links

    def shuffle(self):
        t = self.c
        return np.ctypeslib.as_array(t.shuffle.contents.data, (max(t.max_points, 1),)).copy()[: t.n_points]

    def geom(self):
        t = self.c
        sp = t.seed_points.contents
        seed = np.ctypeslib.as_array(sp.data, (3, sp.tda))[:, :2].copy().reshape(-1)
        sh = np.array([t.shift.contents.data[0], t.shift.contents.data[1]])
        sc = np.array([t.scale.contents.data[0], t.scale.contents.data[1]])
        return np.concatenate([seed, sh, sc])

    def device_alloc(self, device=0):
        h = lib().simplex_tree_device_alloc(self._t, self._m(), device)
        if not h:
            raise GslError(GSL_EFAILED, "simplex_tree_device_alloc")
        return DeviceTree(h)

    def fwrite(self, path):
        with CFile(path, "wb") as fp:
            return lib().simplex_tree_fwrite(fp, self._t)

    @classmethod
    def fread(cls, path, data=None):
        with CFile(path, "rb") as fp:
            t = lib().simplex_tree_fread(fp)
        if not t:
            return None
        self = cls.__new__(cls)
        self._t, self.data, self._mat = t, None, None
        if data is not None:
            self.set_data(data)
        return self

    def output_triangulation(self, response, standardize, lines=None, points=None, circles=None):
        enc = lambda p: str(p).encode() if p is not None else None
        r = as_vector(response) if response is not None else None
        lib().output_triangulation(self._t, self._m(), C.byref(r) if r is not None else None, int(standardize),
                                   enc(lines), enc(points), enc(circles))

    def leaf_walk_order(self):
        """node ids in the order check_leaf_nodes visits them"""
        order = []
        CB = C.CFUNCTYPE(None, _pt, C.c_int)
        cb = CB(lambda tree, node: order.append(node))
        lib().check_leaf_nodes(self._t, C.cast(cb, C.c_void_p))
        return order

    def check_device(self, device=0):
        """(verdict, leaf_violations, delaunay_violations): check_leaf_nodes + check_delaunay on the GPU."""
        lv, dv = C.c_longlong(0), C.c_longlong(0)
        ok = lib().simplex_tree_check_device(self._t, self._m(), device, C.byref(lv), C.byref(dv))
        return ok, lv.value, dv.value

    def device_alloc_multi(self, devices):
        arr = (C.c_int * len(devices))(*devices)
        h = lib().simplex_tree_device_alloc_multi(self._t, self._m(), arr, len(devices))
        if not h:
            raise GslError(GSL_EFAILED, "simplex_tree_device_alloc_multi")
        return DeviceTree(h)

    def close(self):
        if self._t:
            lib().simplex_tree_free(self._t)
            self._t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceTree:
    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    def set_response(self, response):
        return lib().simplex_tree_device_set_response(self._h, C.byref(as_vector(response)))

    def eval_many(self, targets, want_leaf=True, out=None):
        """out = (values, leaf) preallocated arrays (timing loops: fresh arrays pay first-touch page faults in the copy)"""
        m = targets.shape[0]
        vals = np.empty(m, dtype=np.float64) if out is None else out[0]
        leaf = (np.empty(m, dtype=np.int32) if out is None else out[1]) if want_leaf else None
        st = lib().simplex_tree_device_eval_many(self._h, C.byref(as_matrix(targets)), C.byref(as_vector(vals)),
                                                 leaf.ctypes.data_as(_pi) if want_leaf else None)
        return st, vals, leaf

    def eval_resident(self, d_targets, m, ttda, d_values, d_leaf):
        return lib().simplex_tree_device_eval_resident(self._h, d_targets, m, ttda, d_values, d_leaf)

    def n_devices(self):
        return lib().simplex_tree_device_n_devices(self._h)

    def transport(self):
        return lib().simplex_tree_device_transport(self._h).decode()

    def close(self):
        if self._h:
            lib().simplex_tree_device_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SimplexMesh:
    """An imported triangulation (QHull / CGAL style arrays) or the final triangulation of a SimplexTree."""

    def __init__(self, handle, keep=None):
        if not handle:
            raise GslError(GSL_EINVAL, "simplex_mesh")
        self._h = C.c_void_p(handle)
        self._keep = keep

    @classmethod
    def from_arrays(cls, points, triangles, neighbours=None):
        pts = np.ascontiguousarray(points, dtype=np.float64)
        tri = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        nbr = None if neighbours is None else np.ascontiguousarray(neighbours, dtype=np.int32).reshape(-1, 3)
        h = lib().simplex_mesh_import(C.byref(as_matrix(pts)), tri.ctypes.data_as(_pi),
                                      nbr.ctypes.data_as(_pi) if nbr is not None else None, len(tri))
        return cls(h)

    @classmethod
    def from_tree(cls, tree):
        return cls(lib().simplex_mesh_from_tree(tree._t, tree._m()), keep=tree)

    @property
    def n_triangles(self):
        return int(lib().simplex_mesh_n_triangles(self._h))

    def _ints(self, fn, width):
        p = fn(self._h)
        if not p:
            return None
        n = self.n_triangles
        return np.ctypeslib.as_array(p, shape=(n * width,)).reshape(n, width).copy() if width > 1 else \
            np.ctypeslib.as_array(p, shape=(n,)).copy()

    def triangles(self):
        return self._ints(lib().simplex_mesh_triangles, 3)

    def neighbours(self):
        return self._ints(lib().simplex_mesh_neighbours, 3)

    def tree_nodes(self):
        return self._ints(lib().simplex_mesh_tree_nodes, 1)

    def geometry(self):
        sh, sc = (C.c_double * 2)(), (C.c_double * 2)()
        lib().simplex_mesh_geometry(self._h, sh, sc)
        return np.array(sh[:]), np.array(sc[:])

    def set_convex(self, convex):
        lib().simplex_mesh_set_convex(self._h, int(convex))

    def convex(self):
        return bool(lib().simplex_mesh_convex(self._h))

    def device_alloc(self, device=0):
        h = lib().simplex_mesh_device_alloc(self._h, device)
        if not h:
            raise GslError(GSL_EFAILED, "simplex_mesh_device_alloc")
        return DeviceMesh(h, self)

    def device_alloc_multi(self, devices):
        arr = (C.c_int * len(devices))(*devices)
        h = lib().simplex_mesh_device_alloc_multi(self._h, arr, len(devices))
        if not h:
            raise GslError(GSL_EFAILED, "simplex_mesh_device_alloc_multi")
        return DeviceMesh(h, self)

    def fwrite(self, path):
        with CFile(path, "wb") as fp:
            return lib().simplex_mesh_fwrite(fp, self._h)

    @classmethod
    def fread(cls, path):
        with CFile(path, "rb") as fp:
            h = lib().simplex_mesh_fread(fp)
        return cls(h) if h else None

    def close(self):
        if self._h:
            lib().simplex_mesh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceMesh:
    def __init__(self, handle, mesh):
        self._h = C.c_void_p(handle)
        self._mesh = mesh

    def set_response(self, response):
        return lib().simplex_mesh_device_set_response(self._h, C.byref(as_vector(response)))

    def eval_many(self, targets, out=None):
        m = targets.shape[0]
        vals = np.empty(m, dtype=np.float64) if out is None else out[0]
        tri = np.empty(m, dtype=np.int32) if out is None else out[1]
        st = lib().simplex_mesh_device_eval_many(self._h, C.byref(as_matrix(targets)), C.byref(as_vector(vals)), tri.ctypes.data_as(_pi))
        return st, vals, tri

    def n_devices(self):
        return lib().simplex_mesh_device_n_devices(self._h)

    def eval_resident(self, d_targets, m, ttda, d_values, d_tri):
        return lib().simplex_mesh_device_eval_resident(self._h, d_targets, m, ttda, d_values, d_tri)

    def ctx_handle(self):
        return lib().simplex_mesh_device_ctx(self._h)

    def close(self):
        if self._h:
            lib().simplex_mesh_device_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ facade
class Sinterp:
    TYPES = {"gaussian": "gsl_sinterp_rbf_gaussian", "tps": "gsl_sinterp_rbf_tps", "tps_affine": "gsl_sinterp_rbf_tps_affine",
             "wendland": "gsl_sinterp_rbf_wendland", "linear_mesh": "gsl_sinterp_linear_mesh",
             "linear_simplex": "gsl_sinterp_linear_simplex", "kriging": "gsl_sinterp_kriging"}

    def __init__(self, kind, dim, size, device=0):
        self._p = lib().gsl_sinterp_alloc(_ptr(self.TYPES[kind]), dim, size)
        if not self._p:
            raise GslError(GSL_EINVAL, "gsl_sinterp_alloc")
        lib().gsl_sinterp_set_device(self._p, device)
        self._rng = None

    def set_shape(self, eps):
        return lib().gsl_sinterp_set_shape(self._p, eps)

    def set_nugget(self, nugget):
        return lib().gsl_sinterp_set_nugget(self._p, nugget)

    def mean(self):
        v = C.c_double(0)
        st = lib().gsl_sinterp_mean(self._p, C.byref(v))
        return st, v.value

    def poly(self):
        c = np.zeros(self._p.contents.dim + 1, dtype=np.float64)
        st = lib().gsl_sinterp_poly(self._p, C.byref(as_vector(c)))
        return st, c

    def set_solver(self, solver):
        return lib().gsl_sinterp_set_solver(self._p, solver)

    def set_rcond(self, want=True):
        return lib().gsl_sinterp_set_rcond(self._p, int(want))

    def rcond(self):
        r = C.c_double(0)
        st = lib().gsl_sinterp_rcond(self._p, C.byref(r))
        return st, r.value

    def route(self):
        return lib().gsl_sinterp_route(self._p)

    def set_devices(self, n):
        return lib().gsl_sinterp_set_devices(self._p, n)

    def set_device_list(self, devices):
        arr = (C.c_int * len(devices))(*devices)
        return lib().gsl_sinterp_set_device_list(self._p, arr, len(devices))

    def n_devices(self):
        return lib().gsl_sinterp_n_devices(self._p)

    def set_triangulation(self, triangles, neighbours=None):
        tri = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        nbr = None if neighbours is None else np.ascontiguousarray(neighbours, dtype=np.int32).reshape(-1, 3)
        return lib().gsl_sinterp_set_triangulation(self._p, tri.ctypes.data_as(_pi), nbr.ctypes.data_as(_pi) if nbr is not None else None,
                                                   len(tri))

    def set_tree_options(self, flags, rng):
        self._rng = rng
        return lib().gsl_sinterp_set_tree_options(self._p, flags, rng._h if rng is not None else None)

    def init(self, x, f):
        return lib().gsl_sinterp_init(self._p, C.byref(as_matrix(x)), C.byref(as_vector(f)))

    def name(self):
        return lib().gsl_sinterp_name(self._p).decode()

    def eval_e(self, y):
        out = C.c_double(0)
        st = lib().gsl_sinterp_eval_e(self._p, C.byref(as_vector(np.ascontiguousarray(y, dtype=np.float64))),
                                      C.byref(out))
        return st, out.value

    def eval_many(self, y, want_leaf=False, out=None):
        m = y.shape[0]
        s = np.empty(m, dtype=np.float64) if out is None else out
        leaf = np.empty(m, dtype=np.int32) if want_leaf else None
        st = lib().gsl_sinterp_eval_many(self._p, C.byref(as_matrix(y)), C.byref(as_vector(s)),
                                         leaf.ctypes.data_as(_pi) if want_leaf else None)
        return st, s, leaf

    def eval_resident(self, d_y, m, ytda, d_s, d_leaf=None):
        return lib().gsl_sinterp_eval_resident(self._p, d_y, m, ytda, d_s, d_leaf)

    def eval_grid(self, vmin, vmax, n0, n1):
        grid = np.empty((n0, n1), dtype=np.float64)
        st = lib().gsl_sinterp_eval_grid(self._p, C.byref(as_vector(np.ascontiguousarray(vmin, dtype=np.float64))),
                                         C.byref(as_vector(np.ascontiguousarray(vmax, dtype=np.float64))),
                                         C.byref(as_matrix(grid)))
        return st, grid

    def fwrite(self, path):
        with CFile(path, "wb") as fp:
            return lib().gsl_sinterp_fwrite(fp, self._p)

    def fread(self, path):
        with CFile(path, "rb") as fp:
            return lib().gsl_sinterp_fread(fp, self._p)

    def weights(self):
        w = np.empty(self._p.contents.size, dtype=np.float64)
        st = lib().gsl_sinterp_get_weights(self._p, C.byref(as_vector(w)))
        return st, w

    def close(self):
        if self._p:
            lib().gsl_sinterp_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
