"""Times every trailing-update shape of the recursive Cholesky at size N (default 16384) on its own:
the launches chol_panel (csrc/hip/chol.hip) issues, grouped by recursion level, with achieved TFLOP/s
(algorithmic flops: lower part only on the square block).  Tells which levels of the recursion sit
furthest below the MFMA roofline.   usage: python tools/chol_levels.py [N]"""
import sys, collections
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
PB = 128

def split(w):
    unit = PB if w > PB else 32
    w1 = ((w // 2 + unit - 1) // unit) * unit
    return w1 if w1 < w else w - unit

shapes = []
def panel(j0, w):
    if w <= PB:
        return
    w1 = split(w)
    panel(j0, w1)
    r0, w2 = j0 + w1, w - w1
    shapes.append((N - r0, w2, w1))
    panel(r0, w2)
panel(0, N)
a = torch.randn((N, N), dtype=torch.float64, device="cuda")
agg = collections.OrderedDict()
for (m, n, k) in shapes:
    # operands: A = rows x k panel (lda = N), C = rows x n block right of it
    pa = a.data_ptr() + ((N - m) * N) * 8
    pc = pa + k * 8
    ctx.gemm_minus(m, n, k, pa, N, pa, N, 0, pc, N, 1)
    ctx.timer_start()
    reps = 2 if k >= 2048 else 3
    for _ in range(reps):
        ctx.gemm_minus(m, n, k, pa, N, pa, N, 0, pc, N, 1)
    ms = ctx.timer_stop() / reps
    fl = 2.0 * k * (n * (n + 1) / 2.0 + (m - n) * n)
    d = agg.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1; d[1] += ms; d[2] += fl
print("N =", N)
tot_ms = tot_fl = 0.0
for k, (cnt, ms, fl) in sorted(agg.items(), reverse=True):
    print("K=%5d  launches %4d  total %8.3f ms  %7.2f TFLOP/s  (%5.1f %% of the factorisation flops)" % (k, cnt, ms, fl / ms / 1e9, 100 * fl / (N ** 3 / 3.0)))
    tot_ms += ms; tot_fl += fl
print("all updates: %.3f ms, %.2f TFLOP/s" % (tot_ms, tot_fl / tot_ms / 1e9))
