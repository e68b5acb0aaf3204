"""-m gpu: the multi-GPU entries of the C API (gsl_sinterp_set_devices / GSL_SINTERP_DEVICES,
simplex_tree_device_alloc_multi, gsl_sinterp_hip_group_*) on a ONE-GPU box.

A device list may name an ordinal several times: [0, 0, 0] builds three contexts with three streams on
GPU 0, so the whole multi-device machinery -- per-member model buffers, the replication step, the
contiguous target shards, the asynchronous H2D / sweep / D2H of every member and the per-shard
copy-back -- runs and is checked against the single-device result, bit for bit (barycentric) / to the
last bit as well (RBF: same kernels, same per-target criterion).  RCCL refuses duplicate ordinals, so
these groups replicate with peer copies; the RCCL binding itself (dlopen, ncclCommInitAll,
ncclBroadcast inside a group call) is exercised with a one-member communicator.  What a one-GPU box
cannot show is a scaling curve: none has been measured."""
import ctypes as C

import numpy as np
import pytest
import torch

from gpu_util import bits, dev, ptr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0, 0, 0, 0, 0]])
@pytest.mark.parametrize("kind,dim,n,m", [("gaussian", 2, 1500, 20011), ("tps", 2, 900, 5003), ("gaussian", 3, 1100, 3),
                                          ("wendland", 2, 1300, 9001)])
def test_facade_rbf_sharded_equals_single_device(pkg, orc, devices, kind, dim, n, m):
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    y = orc.synth_targets(0, m, dim)
    one = pkg.Sinterp(kind, dim, n, 0)
    assert one.init(x, f) == 0
    st, want, _ = one.eval_many(y)
    assert st == 0
    multi = pkg.Sinterp(kind, dim, n, 0)
    assert multi.set_device_list(devices) == 0 and multi.n_devices() == len(devices)
    assert multi.init(x, f) == 0
    st, got, _ = multi.eval_many(y)
    assert st == 0
    assert np.array_equal(bits(got), bits(want))
    # weights come from member 0; a second batch reuses the cached staging / shard buffers
    st, w1 = one.weights()
    st2, w2 = multi.weights()
    assert st == 0 and st2 == 0 and np.array_equal(w1, w2)
    st, got2, _ = multi.eval_many(y[: m // 2 + 1])
    assert st == 0 and np.array_equal(bits(got2), bits(want[: m // 2 + 1]))
    # oracle parity of the sharded result (north star tolerance)
    kid = {"gaussian": 0, "tps": 1, "wendland": 2}[kind]
    eps = 0.125 * n ** (1.0 / dim) if kind == "wendland" else orc.gaussian_eps(n, dim)
    ref = orc.rbf_eval(kid, eps, x, orc.rbf_solve(kid, eps, x, f), y)
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0]])
def test_simplex_tree_device_multi_bitexact(pkg, orc, devices):
    n, m = 6000, 50021
    x = orc.synth_centres(n, 2) * np.array([3.0, 0.5]) + np.array([-1.0, 10.0])
    f = orc.synth_response(x)
    y = orc.synth_targets(0, m, 2) * np.array([3.0, 0.5]) + np.array([-1.0, 10.0])
    t = pkg.SimplexTree(2, n)
    assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    o = orc.Tree(2, n)
    assert o.init(x, flags=0, seed=0) == 0
    d = t.device_alloc_multi(devices)
    assert d.n_devices() == len(devices) and d.transport() == "peer-copy"
    for resp in (f, 2 * x[:, 0] - x[:, 1]):
        assert d.set_response(resp) == 0
        st, vals, leaf = d.eval_many(y)
        ovals, oleaf = o.eval_many(x, resp, y)
        assert st == 0 and np.array_equal(leaf, oleaf) and np.array_equal(bits(vals), bits(ovals))
    # a target outside the cage in the LAST shard: reported (EDOM, leaf -1, NaN), the other shards intact
    y2 = y[:1000].copy()
    y2[-1] = [1e9, -1e9]
    st, vals, leaf = d.eval_many(y2)
    assert st == pkg.capi.GSL_EDOM and leaf[-1] == -1 and np.isnan(vals[-1])
    ovals, oleaf = o.eval_many(x, 2 * x[:, 0] - x[:, 1], np.ascontiguousarray(y2[:-1]))
    assert np.array_equal(leaf[:-1], oleaf) and np.array_equal(bits(vals[:-1]), bits(ovals))
    # fewer targets than members: trailing shards are empty
    st, vals, leaf = d.eval_many(y[:1])
    assert st == 0 and leaf[0] == oleaf[0]
    # the facade's linear-simplex type over the same list
    s = pkg.Sinterp("linear_simplex", 2, n, 0)
    assert s.set_tree_options(0, pkg.capi.Rng(0)) == 0 and s.set_device_list(devices) == 0
    assert s.init(x, f) == 0
    st, vals, leaf = s.eval_many(y, want_leaf=True)
    ovals, oleaf = o.eval_many(x, f, y)
    assert st == 0 and np.array_equal(leaf, oleaf) and np.array_equal(bits(vals), bits(ovals))


def test_env_device_list_is_the_default(pkg, orc, monkeypatch):
    monkeypatch.setenv("GSL_SINTERP_DEVICES", "0,0,0")
    s = pkg.Sinterp.__new__(pkg.Sinterp)
    s._p = pkg.lib().gsl_sinterp_alloc(pkg.capi._ptr("gsl_sinterp_rbf_gaussian"), 2, 300)
    s._rng = None
    assert s.n_devices() == 3 and list(s._p.contents.devices[:3]) == [0, 0, 0]
    monkeypatch.setenv("GSL_SINTERP_DEVICES", "1")
    s1 = pkg.Sinterp.__new__(pkg.Sinterp)
    s1._p = pkg.lib().gsl_sinterp_alloc(pkg.capi._ptr("gsl_sinterp_rbf_gaussian"), 2, 300)
    s1._rng = None
    assert s1.n_devices() == 1
    x = orc.synth_centres(300, 2)
    f = orc.synth_response(x)
    assert s.init(x, f) == 0
    y = orc.synth_targets(0, 1000, 2)
    st, got, _ = s.eval_many(y)
    assert s1.init(x, f) == 0
    st1, want, _ = s1.eval_many(y)
    assert st == 0 and st1 == 0 and np.array_equal(bits(got), bits(want))
    # an ordinal that does not exist: refused (GSL_EFAILED through the handler), never silently remapped
    bad = pkg.Sinterp("gaussian", 2, 300, 0)
    assert bad.set_device_list([0, 63]) == 0
    assert bad.init(x, f) == pkg.capi.GSL_EFAILED


@pytest.mark.parametrize("single_rccl", ["0", "1"])
def test_group_broadcast_and_rccl_binding(pkg, monkeypatch, single_rccl):
    """gsl_sinterp_hip_group_*: replication from member 0 reaches every member's buffer.  With
    GSL_SINTERP_RCCL_SINGLE=1 a one-member group goes through the real RCCL binding (dlopen of librccl,
    ncclCommInitAll on this GPU, ncclBroadcast inside ncclGroupStart/End on the member's stream)."""
    L = pkg.lib()
    monkeypatch.setenv("GSL_SINTERP_RCCL_SINGLE", single_rccl)
    devices = [0] if single_rccl == "1" else [0, 0, 0]
    arr = (C.c_int * len(devices))(*devices)
    grp = C.c_void_p()
    assert L.gsl_sinterp_hip_group_create(C.byref(grp), arr, len(devices)) == 0, L.gsl_sinterp_hip_group_last_error(grp)
    try:
        assert L.gsl_sinterp_hip_group_size(grp) == len(devices)
        assert L.gsl_sinterp_hip_group_transport(grp).decode() == ("rccl" if single_rccl == "1" else "peer-copy")
        nbytes = 8 * 100003
        src = torch.arange(100003, dtype=torch.float64, device="cuda") * 0.5
        bufs = [src] + [torch.zeros_like(src) for _ in devices[1:]]
        torch.cuda.synchronize()
        ptrs = (C.c_void_p * len(devices))(*[b.data_ptr() for b in bufs])
        assert L.gsl_sinterp_hip_group_broadcast(grp, ptrs, nbytes) == 0, L.gsl_sinterp_hip_group_last_error(grp)
        for i in range(len(devices)):
            assert L.gsl_sinterp_hip_sync(L.gsl_sinterp_hip_group_ctx(grp, i)) == 0
        for b in bufs:
            assert torch.equal(b, src)
    finally:
        L.gsl_sinterp_hip_group_destroy(grp)


def test_two_contexts_spin_wait_kernels_do_not_overlap(pkg, orc):
    """Two contexts with private streams on one device, each factorising (stream-K GEMM + dataflow sweeps:
    kernels that spin on sibling workgroups).  The per-device exclusive-section chain serialises them on the
    GPU; both finish and agree with a lone run bit for bit (co-residency hazard)."""
    n = 2048
    x = orc.synth_centres(n, 2)
    f = orc.synth_response(x)
    eps = orc.gaussian_eps(n, 2)
    ctxs = [pkg.HipContext(0, None) for _ in range(2)]
    for c in ctxs:
        assert pkg.lib().gsl_sinterp_hip_ctx_own_stream(c.handle) == 0
    d_x = dev(x)
    ws, phis = [], []
    torch.cuda.synchronize()
    for rep in range(3):
        ws = [dev(f) for _ in ctxs]
        phis = [torch.empty((n, n), dtype=torch.float64, device="cuda") for _ in ctxs]
        torch.cuda.synchronize()
        for c, w, p in zip(ctxs, ws, phis):              # enqueued back to back on two streams
            st, route = c.rbf_solve(0, eps, ptr(d_x), n, 2, 2, ptr(p), n, ptr(w))
            assert st == 0 and route == 1
        for c in ctxs:
            c.sync()
        assert torch.equal(ws[0], ws[1])
    want = orc.rbf_solve(0, eps, x, f)
    assert np.abs(ws[0].cpu().numpy() - want).max() <= 1e-10 * np.abs(want).max()
    # fully asynchronous entries (no host read-back in between): two large stream-K GEMMs enqueued on the two
    # streams with nothing to separate them but the exclusive-section chain
    h = 2048
    a = torch.randn((h, h), dtype=torch.float64, device="cuda")
    c0 = torch.randn((h, h), dtype=torch.float64, device="cuda")
    lone = c0.clone()
    torch.cuda.synchronize()
    ctxs[0].gemm_minus(h, h, h, ptr(a), h, ptr(a), h, 0, ptr(lone), h, 1)
    ctxs[0].sync()
    cs = [c0.clone(), c0.clone()]
    torch.cuda.synchronize()
    for _ in range(4):
        for c, cc in zip(ctxs, cs):
            c.gemm_minus(h, h, h, ptr(a), h, ptr(a), h, 0, ptr(cc), h, 1)
    for c in ctxs:
        c.sync()
    want4 = c0.clone()
    for _ in range(4):
        ctxs[0].gemm_minus(h, h, h, ptr(a), h, ptr(a), h, 0, ptr(want4), h, 1)
    ctxs[0].sync()
    assert torch.equal(cs[0], want4) and torch.equal(cs[1], want4)
