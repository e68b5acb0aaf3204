import sys, ctypes, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g, oracle_lib as orc
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
x = orc.synth_centres(n, 2); f = orc.synth_response(x)
t = pkg.SimplexTree(2, n); assert t.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
d = t.device_alloc(0); assert d.set_response(f) == 0
y = orc.synth_targets(0, 200000, 2)
st, v, l = d.eval_many(y)
q, lw = ctypes.c_uint(0), ctypes.c_int(0)
pkg.lib().gsl_sinterp_hip_bary_last_queue(pkg.lib().simplex_tree_device_ctx(d._h), ctypes.byref(q), ctypes.byref(lw))
print("st", st, "queue", q.value, "lw", lw.value)
