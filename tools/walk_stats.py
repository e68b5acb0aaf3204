"""Developer tool: walk-length statistics of bary_walk_kernel on C5 (needs `make prof`).

    GSL_SINTERP_LIBRARY=gsl-scattered-interpolation_amd/libgsl_sinterp_prof.so python tools/walk_stats.py [n] [m]
"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
ctx = pkg.HipContext.on_torch_stream(0)
f64 = torch.float64
rng = np.random.default_rng(0)
xh = rng.random((n, 2)); fh = np.sin(3 * xh[:, 0]) + xh[:, 1]
tree = pkg.SimplexTree(2, n)
assert tree.init(xh, flags=0, rng=pkg.capi.Rng(0)) == 0
nn = tree.n_nodes
rec = torch.empty(nn * 64, dtype=torch.uint8, device="cuda"); tab = torch.empty(nn * 32, dtype=torch.uint8, device="cuda")
types, pidx, links = tree.arrays(); sh = tree.shuffle()
d_type, d_pidx, d_links = (torch.from_numpy(a).cuda() for a in (types, pidx, links))
d_pts, d_resp = torch.from_numpy(xh[sh]).cuda(), torch.from_numpy(fh[sh]).cuda()
ctx.tree_pack(nn, d_type.data_ptr(), d_pidx.data_ptr(), d_links.data_ptr(), n, d_pts.data_ptr(), tree.geom(), rec.data_ptr())
ctx.tree_bind(nn, d_pidx.data_ptr(), n, d_resp.data_ptr(), tab.data_ptr())
d_y = torch.from_numpy(rng.random((m, 2))).cuda()
d_v = torch.empty(m, dtype=f64, device="cuda"); d_l = torch.empty(m, dtype=torch.int32, device="cuda")
lib = pkg.capi.lib()
out = (ctypes.c_ulonglong * 40)()
ctx.bary_eval(nn, rec.data_ptr(), tab.data_ptr(), tree.geom()[8:10], d_y.data_ptr(), m, 2, d_v.data_ptr(), d_l.data_ptr())
torch.cuda.synchronize()
lib.gsl_sinterp_hip_debug_walk_stats(out, 1)
ctx.bary_eval(nn, rec.data_ptr(), tab.data_ptr(), tree.geom()[8:10], d_y.data_ptr(), m, 2, d_v.data_ptr(), d_l.data_ptr())
torch.cuda.synchronize()
lib.gsl_sinterp_hip_debug_walk_stats(out, 1)
s = [int(v) for v in out]
print(f"nodes {nn}  targets {m}")
for name, a, b in (("step", 0, 1), ("refill", 2, 3)):
    print(f"{name:7s} wave-iterations {s[a]:10d}  lanes {s[b]:11d}  mean active lanes {s[b] / max(s[a], 1):5.1f} of 64")
