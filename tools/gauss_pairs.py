"""Developer tool: pair counts of the culled Gaussian sweep at the bench shapes (C3: 3-D N = 16384, M = 10^6; C4: 2-D N = 8192,
M = 10^7), from the counters of the prof build.  usage (GPU box):
    GSL_SINTERP_LIBRARY=gsl-scattered-interpolation_amd/libgsl_sinterp_prof.so python tools/gauss_pairs.py > profiles/r04_gauss_pairs.json
evaluated_pair_lanes: lane-slots of the exp2 evaluation the kernel issued (every lane of a wave pays for a centre that any of its
targets takes); useful_pairs: pairs inside the 2^-72 cut-off; staged_pairs: centres that survived the tile culling x targets."""
import ctypes, json, os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import torch
import __graft_entry__ as g
pkg = g.load_package()
lib = pkg.capi.lib()
ctx = pkg.HipContext.on_torch_stream(0)
out = {}
for name, dim, n, m in (("C3", 3, 16384, 1_000_000), ("C4", 2, 8192, 10_000_000)):
    eps = 2.0 * n ** (1.0 / dim)
    d_x = torch.empty((n, dim), dtype=torch.float64, device="cuda")
    d_y = torch.empty((m, dim), dtype=torch.float64, device="cuda")
    d_s = torch.empty(m, dtype=torch.float64, device="cuda")
    d_w = torch.ones(n, dtype=torch.float64, device="cuda")
    ctx.synth_unit(0xC0FFEE01, 0, 0.0, 1.0, d_x.data_ptr(), n * dim)
    ctx.synth_unit(0xC0FFEE02, 0, 0.02, 0.96, d_y.data_ptr(), m * dim)
    ctx.rbf_eval(pkg.RBF_GAUSSIAN, eps, d_x.data_ptr(), n, dim, dim, d_w.data_ptr(), d_y.data_ptr(), m, dim, d_s.data_ptr())
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 4)()
    assert lib.gsl_sinterp_hip_debug_cull_stats(buf, 1) == 0
    ctx.rbf_eval(pkg.RBF_GAUSSIAN, eps, d_x.data_ptr(), n, dim, dim, d_w.data_ptr(), d_y.data_ptr(), m, dim, d_s.data_ptr())
    torch.cuda.synchronize()
    assert lib.gsl_sinterp_hip_debug_cull_stats(buf, 1) == 0
    ev, useful, staged = int(buf[0]), int(buf[1]), int(buf[2])
    out[name] = {"n": n, "dim": dim, "m": m, "algorithmic_pairs": n * m, "staged_pairs": staged, "evaluated_pair_lanes": ev,
                 "useful_pairs": useful, "evaluated_per_target": ev / m, "useful_per_target": useful / m,
                 "lane_efficiency": useful / ev if ev else None}
print(json.dumps(out, indent=1))
