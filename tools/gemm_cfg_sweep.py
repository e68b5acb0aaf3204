"""Developer tool: the trailing-update shapes of the K <= 1024 levels of the N = 16384 factorisation under each tile
configuration the dispatcher can be forced into (GSL_SINTERP_GEMM_CFG = 0 default rule, 1 = 128x128, 2 = 64x64)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    ctx = pkg.HipContext.on_torch_stream(0)
    N = 16384
    a = torch.randn((N, N), dtype=torch.float64, device="cuda")
    for (m, n, k) in [(8192, 128, 128), (8192, 256, 256), (8192, 512, 512), (8192, 1024, 1024), (4096, 512, 512), (14336, 512, 512),
                      (12288, 1024, 1024), (4096, 256, 256), (14336, 256, 256)]:
        pa = a.data_ptr() + ((N - m) * N) * 8
        pc = pa + k * 8
        ctx.gemm_minus(m, n, k, pa, N, pa, N, 0, pc, N, 1)
        ctx.timer_start()
        for _ in range(5):
            ctx.gemm_minus(m, n, k, pa, N, pa, N, 0, pc, N, 1)
        ms = ctx.timer_stop() / 5
        fl = 2.0 * k * (n * (n + 1) / 2.0 + (m - n) * n)
        print(f"  m={m:6d} n={n:5d} k={k:5d}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.2f} TFLOP/s")
else:
    for cfg in ("", "1", "2"):
        env = dict(os.environ)
        if cfg:
            env["GSL_SINTERP_GEMM_CFG"] = cfg
        print("cfg", cfg or "default rule")
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, cwd=ROOT)
