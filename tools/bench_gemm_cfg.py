"""Developer tool: the top-level trailing update (h^3, lower) under the tile configurations GSL_SINTERP_GEMM_CFG selects.
usage: GSL_SINTERP_GEMM_CFG=<1|3|4> python tools/bench_gemm_cfg.py [h]"""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
h = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for (m, n, k) in ((h, h, h), (h, h // 8, h // 8), (h // 2, h // 2, h // 2)):
    a = torch.randn((m, k), dtype=torch.float64, device="cuda")
    b = torch.randn((n, k), dtype=torch.float64, device="cuda")
    c = torch.randn((m, n), dtype=torch.float64, device="cuda")
    c0 = c.clone()
    ctx.gemm_minus(m, n, k, a.data_ptr(), k, b.data_ptr(), k, 0, c.data_ptr(), n, 1)
    torch.cuda.synchronize()
    err = float("nan")
    if m <= 4096:
        want = torch.tril(c0 - a @ b.T) if m == n else None
        if want is not None:
            err = float((torch.tril(c) - want).abs().max())
    ctx.timer_start()
    for _ in range(3):
        ctx.gemm_minus(m, n, k, a.data_ptr(), k, b.data_ptr(), k, 0, c.data_ptr(), n, 1)
    ms = ctx.timer_stop() / 3
    fl = 2.0 * k * (n * (n + 1) / 2.0 + (m - n) * n)
    print(f"cfg={os.environ.get('GSL_SINTERP_GEMM_CFG','default')} m={m} n={n} k={k}: {ms:.3f} ms {fl/ms/1e9:.2f} TFLOP/s err={err:.2e}", flush=True)
