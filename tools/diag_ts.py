import sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.HipContext.on_torch_stream(0)
n = 128
rng = np.random.default_rng(n)
m = rng.random((n, n)); a = np.tril(m) + np.tril(m, -1).T + 10.0 * n * np.eye(n)
d0 = torch.from_numpy(a).cuda()
for _ in range(3):
    d = d0.clone(); st, info = ctx.cholesky_decomp1(n, d.data_ptr(), n)
torch.cuda.synchronize()
lib = pkg.capi.lib()
out = (ctypes.c_ulonglong * 80)()
lib.gsl_sinterp_hip_debug_diag_ts(out)
t = np.array(list(out), dtype=np.int64)
names = {0: 'start', 1: 'loaded', 2: 'P(0) done', 18: 'factored', 19: 'end'}
for jb in range(3):
    names[4 + jb * 4] = 'T(%d) done' % jb
    names[5 + jb * 4] = 'U1(%d) done' % jb
    names[6 + jb * 4] = 'P(%d) || U2(%d) done' % (jb + 1, jb)
prev = t[0]
for i in sorted(names):
    print('%-16s %8d ticks  (+%d)' % (names[i], t[i]-t[0], t[i]-prev)); prev = t[i]

print('per-column ticks of the last potrf32 (jb=3):', [int(t[33+j]-t[32+j]) for j in range(31)])
