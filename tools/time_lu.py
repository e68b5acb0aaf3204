"""Times the pivoted-LU route (gsl_sinterp_hip_lu_decomp) at size N (default 4096) on the thin-plate matrix of C2,
with row stride lda = N + PAD (default 0: the power-of-two stride the bench uses).
usage: python tools/time_lu.py [N] [PAD]"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lda = n + pad
x = torch.empty((n, 2), dtype=torch.float64, device="cuda")
ctx.synth_unit(0xC0FFEE01, 0, 0.0, 1.0, x.data_ptr(), 2 * n)
phi = torch.empty((n, lda), dtype=torch.float64, device="cuda")
perm = torch.empty(n, dtype=torch.int32, device="cuda")
for rep in range(3):
    ctx.rbf_fill(1, 0.0, x.data_ptr(), n, 2, 2, phi.data_ptr(), lda)
    ctx.sync()
    ctx.timer_start()
    ctx.lu_decomp(n, phi.data_ptr(), lda, perm.data_ptr())
    ms = ctx.timer_stop()
    print("N = %d lda = %d: LU_decomp %.2f ms = %.2f TFLOP/s" % (n, lda, ms, 2.0 * n ** 3 / 3.0 / ms / 1e9))
