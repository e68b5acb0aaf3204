"""Standalone timing of the fp64 MFMA GEMM building block (gsl_sinterp_hip_gemm_minus)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
def run(m, n, k, kn, lower, reps=5):
    A = torch.randn((m, k), dtype=torch.float64, device='cuda')
    B = torch.randn((k, n) if kn else (n, k), dtype=torch.float64, device='cuda')
    Cm = torch.randn((m, n), dtype=torch.float64, device='cuda')
    ref = None
    if m * n * k <= 2048 ** 3:
        ref = Cm - (A @ B if kn else A @ B.T)
    ctx.gemm_minus(m, n, k, A.data_ptr(), k, B.data_ptr(), B.shape[1], kn, Cm.data_ptr(), n, lower)
    torch.cuda.synchronize()
    if ref is not None:
        got = Cm if not lower else torch.tril(Cm)
        want = ref if not lower else torch.tril(ref)
        err = float((got - want).abs().max())
    else:
        err = float('nan')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx.gemm_minus(m, n, k, A.data_ptr(), k, B.data_ptr(), B.shape[1], kn, Cm.data_ptr(), n, lower)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * m * n * k * (0.5 * (1 + 128.0 / n) if lower else 1.0)
    print(f"m={m} n={n} k={k} {'KN' if kn else 'NT'} lower={lower}: {ms:.3f} ms  {fl/ms/1e9:.2f} TF  err={err:.2e}", flush=True)
import sys as _s
shapes = [(2048, 2048, 2048, 0, 0), (2048, 2048, 2048, 0, 1), (1024, 1024, 1024, 0, 1), (3072, 1024, 1024, 0, 1),
          (1024, 256, 512, 0, 0), (4096, 4096, 4096, 0, 0), (4096, 4096, 4096, 0, 1),
          (8192, 8192, 8192, 0, 1), (8192, 8192, 8192, 0, 0), (16128, 256, 256, 0, 1), (8192, 128, 128, 0, 1),
          (12288, 4096, 4096, 0, 1), (14336, 2048, 2048, 0, 1), (15360, 1024, 1024, 0, 1), (8192, 512, 512, 0, 1)]
for args in shapes:
    run(*args, reps=3)
