"""Cycle stamps inside the first (tallest) base panel of the pivoted LU (needs `make prof` and
GSL_SINTERP_LIBRARY=.../libgsl_sinterp_prof.so).  usage: python tools/lu_ts.py [N]"""
import sys, ctypes, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.empty((n, 2), dtype=torch.float64, device="cuda")
ctx.synth_unit(0xC0FFEE01, 0, 0.0, 1.0, x.data_ptr(), 2 * n)
phi = torch.empty((n, n), dtype=torch.float64, device="cuda")
perm = torch.empty(n, dtype=torch.int32, device="cuda")
for rep in range(2):
    ctx.rbf_fill(1, 0.0, x.data_ptr(), n, 2, 2, phi.data_ptr(), n)
    ctx.sync()
    ctx.timer_start()
    ctx.lu_decomp(n, phi.data_ptr(), n, perm.data_ptr())
    ms = ctx.timer_stop()
print("N = %d: LU_decomp %.2f ms" % (n, ms))
out = (ctypes.c_ulonglong * 128)()
pkg.capi.lib().gsl_sinterp_hip_debug_lu_ts(out)
t = np.array(list(out), dtype=np.int64)
print("ticks: total %d, load->col0 argmax %d, store %d" % (t[61] - t[0], t[2] - t[0], t[61] - t[60]))
names = ["local argmax", "wave reduce", "barrier + 2nd level", "publish + barrier", "swap", "divide + update"]
for j in range(8):
    b = 2 + 6 * j
    prev = t[b - 1] if j else t[0]
    seg = [t[b] - prev] + [t[b + i] - t[b + i - 1] for i in range(1, 6)]
    print("col %d: " % j + ", ".join("%s %d" % (nm, v) for nm, v in zip(names, seg)) + "  | total %d" % sum(seg))

lc = (ctypes.c_ulonglong * (6 * 64 + 2))()
if hasattr(pkg.capi.lib(), "gsl_sinterp_hip_debug_lc_ts"):
    pkg.capi.lib().gsl_sinterp_hip_debug_lc_ts(lc)
    t = np.array(list(lc), dtype=np.int64)
    print("cooperative panel (first, tallest): load %d ticks, 64 columns %d, store %d" % (t[0] - t[384], t[6 * 63 + 5] - t[0], t[385] - t[6 * 63 + 5]))
    names = ["reduce+barrier", "publish", "poll", "lds+barrier", "swap+update"]
    seg = np.array([[t[6 * j + i + 1] - t[6 * j + i] for i in range(5)] for j in range(64)])
    print("mean ticks per column: " + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, seg.mean(axis=0))) + " | total %.0f" % seg.sum(axis=1).mean())
    for j in (0, 1, 31, 62, 63):
        print("  col %2d: %s" % (j, seg[j].tolist()))
