#!/usr/bin/env python3
"""bench.py -- headline benchmark of the scattered-interpolation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5|C1]

One "step" = one full pass of the hot path over one batch of synthetic input that is
already resident in HBM (SURVEY.md 8(d) clouds, generated on the device):
  RBF configs (C1-C4): Phi fill -> dense factorisation -> two triangular solves (rank 0)
                       -> RCCL broadcast of the weight vector -> N x M evaluation sweep of
                          this rank's shard of targets
  barycentric (C5):    locate + interpolate this rank's shard over the (host-built, already
                       mirrored) Delaunay history DAG
`value` = targets interpolated by all ranks per second of step time (max over ranks), in
M points/s.  Default workload = BASELINE.json configs[1] (C2: 2-D, N=4096 TPS, M=1M per GPU).

For N > 1 the driver launches this file with torch.distributed.run (one rank per GPU).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 matrix = vector peak; 256 CU x 128 flop/clk x 2.4 GHz

CONFIGS = {
    # name: (kind, dim, n_centres, m_targets, sharding)   sharding: per_gpu = weak, total = strong
    "C1": dict(kind="gaussian", dim=2, n=512, m=10_000, shard="per_gpu",
               label="C1: 2-D N=512 Gaussian RBF, M=10k targets"),
    "C2": dict(kind="tps", dim=2, n=4096, m=1_000_000, shard="per_gpu",
               label="C2: 2-D N=4096 thin-plate-spline RBF, M=1M targets per GPU"),
    "C3": dict(kind="gaussian", dim=3, n=16384, m=1_000_000, shard="per_gpu",
               label="C3: 3-D N=16384 Gaussian RBF (16k x 16k fp64 Cholesky), M=1M targets per GPU"),
    "C4": dict(kind="gaussian", dim=2, n=8192, m=10_000_000, shard="total",
               label="C4: 2-D N=8192 Gaussian RBF, M=10M targets sharded over the GPUs"),
    "C5": dict(kind="bary", dim=2, n=50_000, m=10_000_000, shard="total",
               label="C5: 2-D N=50000 barycentric over host-built Delaunay DAG, M=10M targets sharded"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
    assert torch.cuda.is_available(), "bench.py needs a GPU: the hot path has no CPU fallback"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)

    pkg = g.load_package()
    cfg = CONFIGS[args.config]
    ctx = pkg.HipContext.on_torch_stream(local_rank)
    dim, n = cfg["dim"], cfg["n"]
    if cfg["shard"] == "per_gpu":                       # weak scaling: fixed work per GPU
        first, m_rank, m_total = rank * cfg["m"], cfg["m"], cfg["m"] * world
    else:                                               # strong scaling: one target set, sharded
        first, m_rank = pkg.sharding.shard_bounds(cfg["m"], world, rank)
        m_total = cfg["m"]

    f64 = torch.float64
    # ---- synthetic inputs, generated in HBM (same generator as oracle/oracle_synth.c)
    d_x = torch.empty((n, dim), dtype=f64, device="cuda")
    d_y = torch.empty((m_rank, dim), dtype=f64, device="cuda")
    d_s = torch.empty(m_rank, dtype=f64, device="cuda")
    ctx.synth_unit(0xC0FFEE01, 0, 0.0, 1.0, d_x.data_ptr(), n * dim)
    ctx.synth_unit(0xC0FFEE02, first * dim, 0.02, 0.96, d_y.data_ptr(), m_rank * dim)
    d_f = torch.zeros(n, dtype=f64, device="cuda")
    for c in range(dim):
        d_f += torch.sin(3.0 * (c + 1) * d_x[:, c])
    torch.cuda.synchronize()

    phases = {}

    def timed(name, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        phases.setdefault(name, []).append((e0, e1))

    if cfg["kind"] == "bary":
        # host-built triangulation (one-off, not part of the step), mirrored once into HBM
        xh = d_x.cpu().numpy()
        fh = d_f.cpu().numpy()
        t0 = time.time()
        tree = pkg.SimplexTree(2, n)
        assert tree.init(xh, flags=0, rng=pkg.capi.Rng(0)) == 0
        build_s = time.time() - t0
        nn = tree.n_nodes
        rec = torch.empty(nn * 64, dtype=torch.uint8, device="cuda")
        tab = torch.empty(nn * 32, dtype=torch.uint8, device="cuda")
        if rank == 0:
            types, pidx, links = tree.arrays()
            sh = tree.shuffle()
            d_type, d_pidx, d_links = (torch.from_numpy(a).cuda() for a in (types, pidx, links))
            d_pts, d_resp = torch.from_numpy(xh[sh]).cuda(), torch.from_numpy(fh[sh]).cuda()
            ctx.tree_pack(nn, d_type.data_ptr(), d_pidx.data_ptr(), d_links.data_ptr(), n, d_pts.data_ptr(),
                          tree.geom(), rec.data_ptr())
            ctx.tree_bind(nn, d_pidx.data_ptr(), n, d_resp.data_ptr(), tab.data_ptr())
        pkg.sharding.broadcast_model([rec, tab], 0)      # model replication: one broadcast of the packed DAG
        scale = tree.geom()[8:10]
        d_leaf = torch.empty(m_rank, dtype=torch.int32, device="cuda")

        def step():
            timed("bary_eval", lambda: ctx.bary_eval(nn, rec.data_ptr(), tab.data_ptr(), scale, d_y.data_ptr(), m_rank,
                                                       2, d_s.data_ptr(), d_leaf.data_ptr()))
        extra = {"dag_nodes": nn, "host_build_s": round(build_s, 3)}
        dominant = "bary_eval"
    else:
        kind = pkg.RBF_GAUSSIAN if cfg["kind"] == "gaussian" else pkg.RBF_TPS
        eps = 2.0 * n ** (1.0 / dim)
        d_phi = torch.empty((n, n), dtype=f64, device="cuda") if rank == 0 else None
        d_w = torch.empty(n, dtype=f64, device="cuda")
        d_perm = torch.empty(n, dtype=torch.int32, device="cuda")

        route_seen = {}

        def step():
            if rank == 0:
                d_w.copy_(d_f)

                def init():                               # fill + factorisation + triangular solves
                    st, route = ctx.rbf_solve(kind, eps, d_x.data_ptr(), n, dim, dim, d_phi.data_ptr(), n, d_w.data_ptr())
                    assert st == 0, (st, route, pkg.lib().gsl_sinterp_hip_last_error(ctx.handle))
                    route_seen["route"] = route
                timed("init", init)
            if world > 1:
                timed("bcast", lambda: dist.broadcast(d_w, 0))      # RCCL over xGMI: the only data-path collective
            timed("eval", lambda: ctx.rbf_eval(kind, eps, d_x.data_ptr(), n, dim, dim, d_w.data_ptr(), d_y.data_ptr(),
                                               m_rank, dim, d_s.data_ptr()))
        extra = {"eps": eps, "route": route_seen}
        dominant = None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    phases.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=f64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ph_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in phases.items()}
    ms_per_step = elapsed / args.steps * 1e3
    value = m_total * args.steps / elapsed / 1e6

    # ---- sanity of the result that was just produced (not timed)
    sample = d_s[: min(m_rank, 4096)].cpu().numpy()
    assert np.isfinite(sample).all(), "non-finite interpolated values"
    verify = None
    if cfg["kind"] != "bary":
        # the weights of the LAST timed step must interpolate the data: s(x_i) = f_i at a sample of the
        # centres (catches a step that ran fast because it computed garbage, e.g. a broken graph replay)
        ns = min(n, 2048)
        d_chk = torch.empty(ns, dtype=f64, device="cuda")
        ctx.rbf_eval(kind, eps, d_x.data_ptr(), n, dim, dim, d_w.data_ptr(), d_x.data_ptr(), ns, dim, d_chk.data_ptr())
        torch.cuda.synchronize()
        verify = float((d_chk - d_f[:ns]).abs().max()) / float(d_f.abs().max())
        assert verify < 1e-6, f"interpolant does not reproduce the data after the timed steps: {verify:.3e}"
    else:
        # barycentric: the first values of this rank's shard against the library's host per-point entries
        # (find_leaf + interp_point; the reference's own API, bit-identical to the oracle per tests/test_host_tree.py)
        nchk = 300
        yh = d_y[:nchk].cpu().numpy()
        got_v, got_l = d_s[:nchk].cpu().numpy(), d_leaf[:nchk].cpu().numpy()
        bad = 0
        for i in range(nchk):
            leaf = tree.find_leaf(yh[i])
            bad += int(leaf != got_l[i] or tree.interp_point(leaf, fh, yh[i]) != got_v[i])
        verify = float(bad)
        assert bad == 0, f"{bad} of {nchk} GPU barycentric results differ from the host walk"

    if rank == 0:
        out = {
            "metric": "M interpolated points/sec (whole hot path: fill + solve + eval sweep per step)",
            "value": round(value, 4), "unit": "M points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if cfg["shard"] == "per_gpu" else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["label"], "config": args.config, "n_centres": n, "dim": dim,
                       "targets_per_gpu": m_rank, "targets_total": m_total,
                       "parallelism": f"target shards x{world}, weights broadcast (RCCL)" if world > 1 else "1 GPU"},
            "phase_ms": {k: round(v, 4) for k, v in ph_ms.items()},
            "verify_after_timed_steps": verify,
        }
        gemm = time_top_gemm(pkg, ctx, n) if cfg["kind"] != "bary" else None
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc_path):      # HBM bytes per launch from the committed rocprofv3 --pmc passes (not live)
            pmc = json.load(open(pmc_path)).get(args.config, {})
        out.update(rooflines(cfg, n, dim, m_rank, ph_ms, dominant, extra, gemm, pmc))
        mfma_path = os.path.join(ROOT, "profiles", "r01_pmc_mfma.json")
        if os.path.exists(mfma_path):     # MFMA-pipe busy fraction of the top GEMM launch from the committed counter pass (not live)
            mb = json.load(open(mfma_path)).get(args.config, {}).get("gemm_minus_streamk_kernel")
            for key in ("roofline", "roofline_other"):
                if mb and out.get(key) and out[key].get("bound") == "mfma":
                    out[key]["mfma_busy_pmc"] = mb["mfma_busy"]
        out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, n, dim)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def time_top_gemm(pkg, ctx, n, reps=3):
    """HIP-event timing of ONE launch of the factorisation's dominant kernel: the top-level
    trailing update of the recursive Cholesky, C[n/2 x n/2] -= A A^T (lower part), K = n/2."""
    import torch
    h = (n // 2 // 128) * 128
    if h < 256:
        return None
    a = torch.randn((h, h), dtype=torch.float64, device="cuda")
    c = torch.randn((h, h), dtype=torch.float64, device="cuda")
    ctx.gemm_minus(h, h, h, a.data_ptr(), h, a.data_ptr(), h, 0, c.data_ptr(), h, 1)
    ctx.timer_start()
    for _ in range(reps):
        ctx.gemm_minus(h, h, h, a.data_ptr(), h, a.data_ptr(), h, 0, c.data_ptr(), h, 1)
    ms = ctx.timer_stop() / reps
    flops = 2.0 * h * (h * (h + 1) / 2.0)                           # algorithmic: the lower triangle incl. diagonal, K = h
    return {"h": h, "ms": ms, "tflops": flops / ms / 1e9}


def rooflines(cfg, n, dim, m_rank, ph, dominant, extra=None, gemm=None, pmc=None):
    """Roofline objects from live HIP-event timings.  Algorithmic work per SURVEY.md 8(d)."""
    res = {}
    pmc = pmc or {}
    if cfg["kind"] == "bary":
        t = ph["bary_eval"] * 1e-3
        by = 28.0 * m_rank                               # 16 B target in, 8 B value + 4 B leaf out
        res["roofline"] = {"kernel": "bary_eval_kernel (+ cell sort of the targets)", "bound": "hbm",
                           "achieved": round(by / t / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(by / t / 1e9 / HBM_PEAK_GBS, 6), "traffic": pmc.get("bary_eval_kernel"),
                           "note": "latency-bound DAG walk (~65 dependent 64-B gathers per target, DAG resident in "
                                   "Infinity Cache); algorithmic bytes = 28 B/target"}
        return res
    route = (extra or {}).get("route", {}).get("route", 1)
    flops = 2.0 * n ** 3 / 3.0 if route == 3 else (n ** 3) / 3.0      # LU vs Cholesky factorisation
    tf = ph["init"] * 1e-3
    te = ph["eval"] * 1e-3
    by = (8.0 * dim + 8.0) * m_rank
    r_gemm = None
    if gemm:
        r_gemm = {"kernel": "gemm_minus_streamk_kernel<256,128,64,64> (top-level trailing update, %d^3 lower)" % gemm["h"],
                  "bound": "mfma", "achieved": round(gemm["tflops"], 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": round(gemm["tflops"] / FP64_PEAK_TFLOPS, 5), "launch_ms": round(gemm["ms"], 4),
                  "traffic": pmc.get("gemm_minus_streamk_kernel")}
    pair_ops = n * m_rank
    gauss = cfg["kind"] == "gaussian"
    ek = "rbf_eval_gauss_cull_kernel" if gauss else "rbf_eval_kernel"
    r_eval = {"kernel": ek, "bound": "hbm", "achieved": round(by / te / 1e9, 3), "peak": HBM_PEAK_GBS,
              "unit": "GB/s", "frac": round(by / te / 1e9 / HBM_PEAK_GBS, 6), "traffic": pmc.get(ek),
              "pair_evals_per_s": round(pair_ops / te, 1),
              "note": "fp64-VALU bound by construction (N pair-evals per 8d+8 B): the HBM fraction is ~1e-3; "
                      "pair_evals_per_s is the meaningful rate"}
    if gauss:
        r_eval["note"] += ("; Gaussian: tiles of centres beyond the 2^-72 cut-off are culled, so the algorithmic "
                           "pair rate (N*M/t) exceeds what the VALUs could evaluate pair by pair")
    else:
        # thin-plate sweep: ~20 VALU instructions per pair (2-D; ISA count of the inner loop), issue peak =
        # 256 CU x 4 SIMD x 16 lanes x 2.4 GHz lane-instructions/s
        ipp = 17 + 3 * (dim - 1)
        peak = 256 * 4 * 16 * 2.4e9
        r_eval["valu_issue"] = {"instr_per_pair": ipp, "achieved_lane_instr_per_s": round(pair_ops * ipp / te, 1),
                                "peak_lane_instr_per_s": peak, "frac": round(pair_ops * ipp / te / peak, 4)}
    # dominant single kernel: the GEMM when the factorisation (mostly GEMM) outweighs the sweep
    gemm_dominant = r_gemm is not None and 0.55 * tf > te
    res["roofline"] = r_gemm if gemm_dominant else r_eval
    res["roofline_other"] = r_eval if gemm_dominant else r_gemm
    res["init_as_a_unit"] = {"flops": flops, "tflops": round(flops / tf / 1e12, 4),
                             "frac_of_fp64_mfma_peak": round(flops / tf / 1e12 / FP64_PEAK_TFLOPS, 5),
                             "route": {1: "cholesky", 2: "shifted-SPD cholesky + Woodbury", 3: "pivoted LU"}.get(route, "?")}
    res["solve_gflops"] = round(flops / tf / 1e9, 2)
    res["eval_only_mpts"] = round(m_rank / te / 1e6, 3)
    return res


def cpu_baseline(cfg, n, dim):
    """The CPU oracle (reference-order C restatement) timed on this box's host cores, one
    thread, on a bounded sample of the same workload (kind: "port")."""
    import oracle_lib as orc
    cores = 1
    if cfg["kind"] == "bary":
        x = orc.synth_centres(n, 2)
        f = orc.synth_response(x)
        t = orc.Tree(2, n)
        t0 = time.perf_counter()
        assert t.init(x, flags=0, seed=0) == 0
        build = time.perf_counter() - t0
        ms = 200_000
        y = orc.synth_targets(0, ms, 2)
        t0 = time.perf_counter()
        t.eval_many(x, f, y)
        dt = time.perf_counter() - t0
        return {"value": round(ms / dt / 1e6, 5), "unit": "M points/s", "cores": cores, "kind": "port",
                "sample": f"first {ms} of the targets, N={n}; host DAG build {build:.2f} s (one-off, excluded)"}
    kind = 0 if cfg["kind"] == "gaussian" else 1
    eps = orc.gaussian_eps(n, dim)
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    # factorisation sample: N capped so the unblocked reference-order solver takes a few seconds
    ns = min(n, 1536)
    xs, fs = np.ascontiguousarray(x[:ns]), np.ascontiguousarray(f[:ns])
    phi = orc.rbf_fill(kind, orc.gaussian_eps(ns, dim), xs)
    t0 = time.perf_counter()
    if kind == 0:
        st, llt = orc.cholesky_decomp1(phi)
        w = orc.cholesky_solve(llt, fs)
    else:
        lu, perm, _ = orc.lu_decomp(phi)
        st, w = orc.lu_solve(lu, perm, fs)
    dts = time.perf_counter() - t0
    fl = (ns ** 3 / 3.0) if kind == 0 else (2.0 * ns ** 3 / 3.0)
    # eval sample: ~10 s of single-core work at ~50 M pair-evals/s
    ms = max(1000, int(5.0e8 // n))
    y = orc.synth_targets(0, ms, dim)
    wfull = np.resize(w, n)
    t0 = time.perf_counter()
    orc.rbf_eval(kind, eps, x, wfull, y)
    dte = time.perf_counter() - t0
    return {"value": round(ms / dte / 1e6, 6), "unit": "M points/s", "cores": cores, "kind": "port",
            "sample": f"eval sweep of the first {ms} targets against all N={n} centres "
                      f"({n * ms / dte / 1e6:.1f} M pair-evals/s); factor+solve at N={ns}: {dts:.2f} s",
            "solve_gflops": round(fl / dts / 1e9, 3)}


if __name__ == "__main__":
    main()
