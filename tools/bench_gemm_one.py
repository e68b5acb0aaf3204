"""One shape of the stream-K GEMM (default: the top-level update of C3, 8192^3 lower), 1 warm-up + 2 timed launches;
used under rocprofv3 --pmc to read the fabric traffic of that launch (tools/pmc_fetch_gemm.sh)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
h = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
a = torch.randn((h, h), dtype=torch.float64, device="cuda")
c = torch.randn((h, h), dtype=torch.float64, device="cuda")
ctx.gemm_minus(h, h, h, a.data_ptr(), h, a.data_ptr(), h, 0, c.data_ptr(), h, 1)
ctx.timer_start()
for _ in range(2):
    ctx.gemm_minus(h, h, h, a.data_ptr(), h, a.data_ptr(), h, 0, c.data_ptr(), h, 1)
ms = ctx.timer_stop() / 2
print("h =", h, "ms =", ms, "TFLOP/s =", 2.0 * h * (h * (h + 1) / 2.0) / ms / 1e9)
