#!/usr/bin/env python3
"""bench.py -- headline benchmark of the scattered-interpolation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C2|C4|C5|C1|C4W] [--only]

One "step" = one full pass of the hot path over one batch of synthetic input that is
already resident in HBM (SURVEY.md 8(d) clouds, generated on the device):
  RBF configs (C1-C4): Phi fill -> dense factorisation -> two triangular solves (rank 0)
                       -> RCCL broadcast of the weight vector -> N x M evaluation sweep of
                          this rank's shard of targets
  barycentric (C5):    locate + interpolate this rank's shard over the (host-built, already
                       mirrored) Delaunay history DAG
`value` = targets interpolated by all ranks per second of step time (max over ranks), in
M points/s.

With --gpus 1 the driver-timed line is the LARGEST single-GPU configuration of BASELINE.json, C3 (3-D, N=16384
Gaussian: the 16k x 16k fp64 Cholesky the north star's MFMA target is quoted on, M=1M targets); with --gpus N > 1 it
is C4, the strong-scaling configuration BASELINE.json quotes for 8 GPUs (10^7 targets sharded over the ranks: init
on rank 0 is serial and reported as such, `eval_only_mpts_aggregate` = all targets / slowest rank's sweep);
its `roofline` is the dominant kernel, the top-level trailing update of the factorisation (fp64 MFMA).
The other GPU configurations (C2, C4, C5) run in the same invocation -- a few hundred ms in all -- and are
reported under `extra.other_configs`, each with its own roofline object (`--only` skips them).

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
VALU_PEAK_LANE_INSTR = 256 * 4 * 16 * 2.4e9   # fp64 VALU issue: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz lane-instructions/s
PMC_ROUND = "r04"            # committed rocprofv3 --pmc passes the static counter figures are read from

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
    # beyond BASELINE.json's configs (never part of the default line): the compactly supported kernel, C4's shape
    "C4W": dict(kind="wendland", dim=2, n=8192, m=10_000_000, shard="total",
                label="C4W: 2-D N=8192 Wendland-C2 RBF (support of 8 mean spacings), M=10M targets sharded over the GPUs"),
    "C5": dict(kind="bary", dim=2, n=50_000, m=10_000_000, shard="total",
               label="C5: 2-D N=50000 barycentric over host-built Delaunay DAG, M=10M targets sharded"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="headline configuration; default C3 on one GPU, C4 (strong scaling, the configuration "
                         "BASELINE.json quotes for 8 GPUs) on several")
    ap.add_argument("--only", action="store_true", help="run the headline configuration only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


class Env:
    """Process-wide state shared by the configurations of one invocation."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        import __graft_entry__ as g
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # Rehearsal aids for a ONE-GPU box (never set by the driver): BENCH_FORCE_DEVICE puts every rank on that
        # ordinal and BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU), so the whole
        # multi-rank control flow -- shards, rank-0-only init, broadcasts, barriers, max-reduction, rank-0 JSON --
        # runs with N processes sharing one card; the numbers of such a run mean nothing.
        forced = os.environ.get("BENCH_FORCE_DEVICE")
        if forced is not None:
            self.local_rank = int(forced)
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        self.rehearsal = forced is not None or backend != "nccl"
        if self.world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(self.local_rank)
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))   # nccl == RCCL on ROCm
            else:
                dist.init_process_group(backend=backend)
        assert torch.cuda.is_available(), "bench.py needs a GPU: the hot path has no CPU fallback"
        assert self.world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={self.world}"
        torch.cuda.set_device(self.local_rank)
        self.pkg = g.load_package()
        self.ctx = self.pkg.HipContext.on_torch_stream(self.local_rank)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def run_config(env, name, steps, warmup):
    """Time `steps` steps of configuration `name` (after `warmup` untimed ones) between two barriers;
    returns the result dict of this configuration (identical on every rank up to the max-reduction)."""
    torch, dist, pkg, ctx = env.torch, env.dist, env.pkg, env.ctx
    world, rank = env.world, env.rank
    cfg = CONFIGS[name]
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

    def timed(pname, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        phases.setdefault(pname, []).append((e0, e1))

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
        # model replication as in the library's device groups: rank 0's flat DAG arrays are broadcast
        # (13 MB at C5), every rank packs its own records -- and with them the full-resolution jump table, which
        # a rank that only received packed records would have to rebuild per batch at a coarser resolution
        types, pidx, links = tree.arrays()
        sh = tree.shuffle()
        flat = [types, pidx, links, xh[sh], fh[sh]]
        d_type, d_pidx, d_links, d_pts, d_resp = (torch.from_numpy(a).cuda() if rank == 0 else
                                                  torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, device="cuda") for a in flat)
        pkg.sharding.broadcast_model([d_type, d_pidx, d_links, d_pts, d_resp], 0)
        ctx.timer_start()
        ctx.tree_pack(nn, d_type.data_ptr(), d_pidx.data_ptr(), d_links.data_ptr(), n, d_pts.data_ptr(),
                      tree.geom(), rec.data_ptr())
        pack_ms = ctx.timer_stop()
        ctx.tree_bind(nn, d_pidx.data_ptr(), n, d_resp.data_ptr(), tab.data_ptr())
        scale = tree.geom()[8:10]
        d_leaf = torch.empty(m_rank, dtype=torch.int32, device="cuda")

        def step():
            timed("bary_eval", lambda: ctx.bary_eval(nn, rec.data_ptr(), tab.data_ptr(), scale, d_y.data_ptr(), m_rank,
                                                       2, d_s.data_ptr(), d_leaf.data_ptr()))
        extra = {"dag_nodes": nn, "host_build_s": round(build_s, 3),
                 "tree_pack_ms_once_per_tree": None if pack_ms is None else round(pack_ms, 3),
                 "tree_pack_note": "node records + per-cell jump table (jump_build_kernel), once per tree, outside the step"}
    else:
        kind = {"gaussian": pkg.RBF_GAUSSIAN, "tps": pkg.RBF_TPS, "wendland": pkg.RBF_WENDLAND}[cfg["kind"]]
        eps = (0.125 if cfg["kind"] == "wendland" else 2.0) * n ** (1.0 / dim)
        d_phi = torch.empty((n, n), dtype=f64, device="cuda") if rank == 0 else None
        d_w = torch.empty(n, dtype=f64, device="cuda")
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

    for _ in range(warmup):
        step()
    env.barrier()
    phases.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    env.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=f64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ph_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in phases.items()}
    if world > 1 and "init" not in ph_ms and cfg["kind"] != "bary":       # ranks > 0 do not factor
        ph_ms["init"] = float("nan")
    ms_per_step = elapsed / steps * 1e3
    value = m_total * steps / elapsed / 1e6
    # per-phase MAX over ranks (a rank without the phase contributes 0): the aggregate eval-only rate of the job is
    # all targets / the slowest rank's sweep, and the init of rank 0 is visibly serial (Amdahl) instead of hidden
    names = ("init", "bcast", "eval", "bary_eval")
    tmax = torch.tensor([ph_ms.get(k, 0.0) if ph_ms.get(k, 0.0) == ph_ms.get(k, 0.0) else 0.0 for k in names], dtype=f64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ph_max = {k: float(v) for k, v in zip(names, tmax.tolist()) if v > 0.0}

    # ---- sanity of the result that was just produced (not timed)
    sample = d_s[: min(m_rank, 4096)].cpu().numpy()
    assert np.isfinite(sample).all(), "non-finite interpolated values"
    if cfg["kind"] == "tps" and rank == 0:
        # the reference's own route for this kernel class (gsl_linalg_LU_decomp + _svx, linalg/lu.c:59-201), forced, timed
        # beside the default shifted-SPD route: 1 warm-up + 2 timed inits
        os.environ["GSL_SINTERP_FORCE_LU"] = "1"
        try:
            lu_ms = []
            for rep in range(3):
                d_w.copy_(d_f)
                ctx.timer_start()
                st, route = ctx.rbf_solve(kind, eps, d_x.data_ptr(), n, dim, dim, d_phi.data_ptr(), n, d_w.data_ptr())
                ms = ctx.timer_stop()
                assert st == 0 and route == 3, (st, route)
                if rep:
                    lu_ms.append(ms)
            extra["init_ms_reference_route_pivoted_lu"] = round(float(np.mean(lu_ms)), 3)
        finally:
            del os.environ["GSL_SINTERP_FORCE_LU"]
        d_w.copy_(d_f)                                   # leave the default route's weights for the verification below
        st, route = ctx.rbf_solve(kind, eps, d_x.data_ptr(), n, dim, dim, d_phi.data_ptr(), n, d_w.data_ptr())
        assert st == 0
    if cfg["kind"] == "tps" and world > 1:
        dist.broadcast(d_w, 0)
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
        nchk = min(300, m_rank)
        yh = d_y[:nchk].cpu().numpy()
        got_v, got_l = d_s[:nchk].cpu().numpy(), d_leaf[:nchk].cpu().numpy()
        bad = 0
        for i in range(nchk):
            leaf = tree.find_leaf(yh[i])
            bad += int(leaf != got_l[i] or tree.interp_point(leaf, fh, yh[i]) != got_v[i])
        verify = float(bad)
        assert bad == 0, f"{bad} of {nchk} GPU barycentric results differ from the host walk"

    res = {
        "workload": cfg["label"], "config": name, "n_centres": n, "dim": dim,
        "targets_per_gpu": m_rank, "targets_total": m_total,
        "scaling": "weak" if cfg["shard"] == "per_gpu" else "strong",
        "value_mpts": round(value, 4), "ms_per_step": round(ms_per_step, 4), "steps": steps, "warmup": warmup,
        "phase_ms": {k: round(v, 4) for k, v in ph_ms.items()},
        "phase_ms_max_over_ranks": {k: round(v, 4) for k, v in ph_max.items()},
        "verify_after_timed_steps": verify, "extra": extra,
    }
    t_eval = ph_max.get("bary_eval", ph_max.get("eval"))
    if t_eval:
        res["eval_only_mpts_aggregate"] = round(m_total / (t_eval * 1e-3) / 1e6, 3)   # all ranks' targets / slowest rank's sweep
        res["eval_speedup_vs_1gpu_expected_from"] = (
            "divide by `eval_only_mpts_aggregate` of the SAME config in the --gpus 1 run (headline or extra.other_configs); "
            "whole-step `value_mpts` additionally carries rank 0's serial init (phase_ms_max_over_ranks.init) and the broadcast")
    if rank == 0:
        gemm = time_top_gemm(ctx, n) if cfg["kind"] != "bary" and n >= 2048 else None
        res.update(rooflines(cfg, name, n, dim, m_rank, ph_ms, extra, gemm))
    del d_x, d_y, d_s
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    env = Env(args)
    # One GPU: C3, the largest single-GPU configuration (the 16k x 16k Cholesky the MFMA target is quoted on).
    # Several GPUs: C4, the STRONG-scaling configuration BASELINE.json quotes for 8 GPUs (fixed 10^7 targets sharded
    # over the ranks).  C3 with a fixed M per GPU would read ~N x by construction while rank 0's ~30 ms init pins the
    # step and N-1 GPUs idle (round-2 review); it is still run, labelled weak, under other_configs.
    headline = args.config or ("C3" if env.world == 1 else "C4")
    head = run_config(env, headline, args.steps, args.warmup)
    others = {}
    if not args.only:
        for name in ("C2", "C3", "C4", "C5"):
            if name != headline and not (name == "C3" and headline != "C3" and env.world == 1):
                others[name] = run_config(env, name, min(args.steps, 3), min(args.warmup, 1))
    if env.rank == 0:
        world = env.world
        out = {
            "metric": "M interpolated points/sec (whole hot path: fill + solve + eval sweep per step)",
            "value": head["value_mpts"], "unit": "M points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": head["scaling"],
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": head["workload"], "config": head["config"], "n_centres": head["n_centres"],
                       "dim": head["dim"], "targets_per_gpu": head["targets_per_gpu"], "targets_total": head["targets_total"],
                       "parallelism": f"target shards x{world}, weights broadcast (RCCL)" if world > 1 else "1 GPU"},
            "phase_ms": head["phase_ms"], "phase_ms_max_over_ranks": head["phase_ms_max_over_ranks"],
            "verify_after_timed_steps": head["verify_after_timed_steps"],
        }
        # the points of the strong-scaling curves, in every run (N = 1 included) so that the curve can be assembled from
        # the per-N lines: whole step and eval only, C4 and C5 (no multi-GPU curve has been MEASURED by the builder:
        # development boxes have one GPU)
        allc = dict(others, **{headline: head})
        out["strong_scaling_points"] = {
            k: {"n_gpus": world, "whole_step_mpts": allc[k]["value_mpts"], "eval_only_mpts_aggregate": allc[k].get("eval_only_mpts_aggregate"),
                "init_ms_rank0": allc[k]["phase_ms_max_over_ranks"].get("init")}
            for k in ("C4", "C5") if k in allc}
        for k in ("roofline", "roofline_other", "init_as_a_unit", "solve_gflops", "eval_only_mpts", "eval_only_mpts_aggregate",
                  "eval_speedup_vs_1gpu_expected_from"):
            if k in head:
                out[k] = head[k]
        out["extra"] = dict(head["extra"], other_configs=others)
        if env.rehearsal:
            out["extra"]["REHEARSAL"] = "ranks share one GPU / non-RCCL backend: control-flow check only, the numbers are meaningless"
        if world == 1 and not args.only:
            try:
                out["facade_host_mpts"] = {k: facade_host_path(env, k) for k in ("C4", "C5")}
            except Exception as exc:                      # never let the side measurement take the line down
                out["facade_host_mpts"] = {"error": repr(exc)}
            try:
                out["extra"]["imported_mesh_C5M"] = mesh_figure(env)
            except Exception as exc:
                out["extra"]["imported_mesh_C5M"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(CONFIGS[headline], head["n_centres"], head["dim"], head["targets_total"])
        print(json.dumps(out))
    if env.world > 1:
        env.dist.destroy_process_group()


def facade_host_path(env, name, reps=3):
    """The drop-in entries fed with HOST matrices (gsl_sinterp_eval_many / simplex_tree_device_eval_many): chunks of the
    batch pipelined over the copy pipe (H2D | sweep | D2H).  Reported beside the PCIe bound measured in this run
    (hipMemcpy of the same pageable arrays, both directions one after the other) -- never `value`."""
    import numpy as np
    torch, pkg, ctx = env.torch, env.pkg, env.ctx
    cfg = CONFIGS[name]
    dim, n, m = cfg["dim"], cfg["n"], cfg["m"]
    d_x = torch.empty((n, dim), dtype=torch.float64, device="cuda")
    d_y = torch.empty((m, dim), dtype=torch.float64, device="cuda")
    ctx.synth_unit(0xC0FFEE01, 0, 0.0, 1.0, d_x.data_ptr(), n * dim)
    ctx.synth_unit(0xC0FFEE02, 0, 0.02, 0.96, d_y.data_ptr(), m * dim)
    x = d_x.cpu().numpy()
    y = d_y.cpu().numpy()
    f = np.sin(3.0 * x[:, 0]) + np.sin(6.0 * x[:, 1]) + (np.sin(9.0 * x[:, 2]) if dim == 3 else 0.0)
    vals = np.zeros(m)
    leaf = np.zeros(m, dtype=np.int32)
    if cfg["kind"] == "bary":
        tree = pkg.SimplexTree(2, n)
        assert tree.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
        dev = tree.device_alloc(0)
        assert dev.set_response(f) == 0
        run = lambda: dev.eval_many(y, out=(vals, leaf))[0]
        bytes_up, bytes_down = 16.0 * m, 12.0 * m
    else:
        s = pkg.Sinterp("gaussian", dim, n, 0)
        assert s.init(x, f) == 0
        run = lambda: s.eval_many(y, out=vals)[0]
        bytes_up, bytes_down = 8.0 * dim * m, 8.0 * m
    assert run() == 0
    t0 = time.perf_counter()
    for _ in range(reps):
        assert run() == 0
    ms = (time.perf_counter() - t0) / reps * 1e3
    # the PCIe bound of the same pageable arrays: plain copies, one direction after the other
    d_v = torch.empty(m, dtype=torch.float64, device="cuda")
    ty, tv = torch.from_numpy(y), torch.from_numpy(vals)
    tl = torch.from_numpy(leaf)
    d_l = torch.empty(m, dtype=torch.int32, device="cuda")
    d_y.copy_(ty); tv.copy_(d_v); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        d_y.copy_(ty)
        tv.copy_(d_v)
        if cfg["kind"] == "bary":
            tl.copy_(d_l)
        torch.cuda.synchronize()
    ms_pcie = (time.perf_counter() - t0) / reps * 1e3
    return {"config": name, "ms_per_batch": round(ms, 3), "mpts": round(m / ms / 1e3, 2), "bytes_per_target": (bytes_up + bytes_down) / m,
            "pcie_copy_ms_same_arrays": round(ms_pcie, 3), "pcie_bound_mpts": round(m / ms_pcie / 1e3, 2),
            "frac_of_pcie_bound": round(ms_pcie / ms, 4),
            "note": "host gsl_matrix in, host gsl_vector out through the C facade (pageable memory, dense rows); the bound is "
                    "torch's H2D + D2H of the same arrays, serial, measured in this run"}


def mesh_figure(env, steps=10):
    """Imported triangulations at C5's shape: the final triangulation of the N = 50 000 tree exported as a mesh
    (simplex_mesh_from_tree), M = 10^7 resident targets through the seed grid + leaf-adjacency walk; checked against the
    DAG path on a sample (same leaf -> same bits)."""
    import numpy as np
    torch, pkg, ctx = env.torch, env.pkg, env.ctx
    n, m = CONFIGS["C5"]["n"], CONFIGS["C5"]["m"]
    d_x = torch.empty((n, 2), dtype=torch.float64, device="cuda")
    d_y = torch.empty((m, 2), dtype=torch.float64, device="cuda")
    ctx.synth_unit(0xC0FFEE01, 0, 0.0, 1.0, d_x.data_ptr(), n * 2)
    ctx.synth_unit(0xC0FFEE02, 0, 0.02, 0.96, d_y.data_ptr(), m * 2)
    x = d_x.cpu().numpy()
    f = np.sin(3.0 * x[:, 0]) + np.sin(6.0 * x[:, 1])
    tree = pkg.SimplexTree(2, n)
    assert tree.init(x, flags=0, rng=pkg.capi.Rng(0)) == 0
    mesh = pkg.SimplexMesh.from_tree(tree)
    dev = mesh.device_alloc(0)
    assert dev.set_response(f) == 0
    d_v = torch.empty(m, dtype=torch.float64, device="cuda")
    d_t = torch.empty(m, dtype=torch.int32, device="cuda")
    run = lambda: dev.eval_resident(d_y.data_ptr(), m, 2, d_v.data_ptr(), d_t.data_ptr())
    assert run() == 0
    pkg.lib().gsl_sinterp_hip_sync(dev.ctx_handle())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        assert run() == 0
    pkg.lib().gsl_sinterp_hip_sync(dev.ctx_handle())
    ms = (time.perf_counter() - t0) / steps * 1e3
    # the DAG path on a sample: identical values wherever the mesh found a triangle
    tdev = tree.device_alloc(0)
    assert tdev.set_response(f) == 0
    idx = np.arange(0, m, 9973)
    ys = np.ascontiguousarray(d_y.cpu().numpy()[idx])
    _, dag_v, _ = tdev.eval_many(ys)
    got = d_v.cpu().numpy()[idx]
    tri = d_t.cpu().numpy()[idx]
    same = bool(np.array_equal(got[tri >= 0].view(np.uint64), dag_v[tri >= 0].view(np.uint64)))
    return {"workload": "C5M: the final triangulation of C5's tree exported as an imported mesh, M = 10M resident targets",
            "n_triangles": mesh.n_triangles, "ms_per_step": round(ms, 4), "mpts": round(m / ms / 1e3, 2),
            "values_equal_dag_path_on_sample": same, "targets_outside_mesh_in_sample": int((tri < 0).sum())}


def time_top_gemm(ctx, n, reps=3):
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


def committed_pmc(name, kernel):
    """Counter figures of `kernel` for configuration `name` from the committed rocprofv3 --pmc passes
    (separate profiling runs of this same command; NOT measured in the run that printed the line)."""
    out = {}
    for key, fname in (("hbm_traffic_bytes_per_launch", f"{PMC_ROUND}_pmc_traffic.json"), ("mfma_busy", f"{PMC_ROUND}_pmc_mfma.json")):
        path = os.path.join(ROOT, "profiles", fname)
        if os.path.exists(path):
            v = json.load(open(path)).get(name, {}).get(kernel)
            if v is not None:
                out[key] = v["mfma_busy"] if isinstance(v, dict) and "mfma_busy" in v else v
                out.setdefault("source", []).append("profiles/" + fname)
    if out:
        out["note"] = "committed rocprofv3 --pmc pass of this command, not measured in this run"
    return out or None


TPS_VALU_PER_PAIR = {1: 17.25, 2: 19.25, 3: 21.25}       # rbf_eval_kernel<TPS, dim, 2>: VALU instructions of the inner loop / 4 pairs
GAUSS_VALU_PER_PAIR = {2: (6.5, 20.0), 3: (9.0, 20.0)}    # rbf_eval_gauss_cull_kernel: (cut-off pre-test, exp2 evaluation) per pair-slot


def gauss_pair_counts(name, n, dim, m_rank):
    """Pair counts of the culled sweep from the committed counter run (profiles/r04_gauss_pairs.json: the prof build's
    counters on the same synthetic clouds), scaled to this rank's share of the targets."""
    path = os.path.join(ROOT, "profiles", "r04_gauss_pairs.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path)).get(name)
    if not rec or rec["n"] != n or rec["dim"] != dim:
        return None
    f = m_rank / rec["m"]
    return {"staged_pairs": rec["staged_pairs"] * f, "evaluated_pair_lanes": rec["evaluated_pair_lanes"] * f,
            "useful_pairs": rec["useful_pairs"] * f, "source": "profiles/r04_gauss_pairs.json"}


def rooflines(cfg, name, n, dim, m_rank, ph, extra=None, gemm=None):
    """Roofline objects from live HIP-event timings.  Algorithmic work per SURVEY.md 8(d)."""
    res = {}
    if cfg["kind"] == "bary":
        t = ph["bary_eval"] * 1e-3
        by = 28.0 * m_rank                               # 16 B target in, 8 B value + 4 B leaf out
        pmc = committed_pmc(name, "leafwalk_kernel")
        res["roofline"] = {"kernel": "leafwalk_kernel (+ exact kernel on the ~1 % the margin test leaves, two-level reorder / un-sort of the targets)", "bound": "hbm",
                           "achieved": round(by / t / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(by / t / 1e9 / HBM_PEAK_GBS, 6),
                           "traffic": pmc["hbm_traffic_bytes_per_launch"] if pmc and "hbm_traffic_bytes_per_launch" in pmc else None,
                           "traffic_source": pmc["source"] if pmc else None,
                           "note": "the algorithmic HBM stream is 28 B/target; of the ~1.1 ms step 0.35 is the two-level reorder of the "
                                   "targets (five streaming passes, 104 B/target at ~3.3 TB/s), 0.14 the random gather of the un-sort, "
                                   "~0.5 the certified leaf walk (the walk over the leaves' adjacency from the grid seed: latency of "
                                   "dependent 64-byte gathers from L2 / Infinity Cache; the margin test against the leaf's ~13 lines: "
                                   "per-lane load rate), 0.09 the exact DAG kernel on the ~1 % of targets the margin leaves -- not HBM"}
        res["eval_only_mpts"] = round(m_rank / t / 1e6, 3)
        return res
    route = (extra or {}).get("route", {}).get("route", 1)
    flops = 2.0 * n ** 3 / 3.0 if route == 3 else (n ** 3) / 3.0      # LU vs Cholesky factorisation
    tf = ph.get("init", float("nan")) * 1e-3
    te = ph["eval"] * 1e-3
    by = (8.0 * dim + 8.0) * m_rank
    r_gemm = None
    if gemm:
        pmc = committed_pmc(name, "gemm_minus_streamk_kernel")
        r_gemm = {"kernel": "gemm_minus_streamk_kernel<256,128,64,64> (top-level trailing update, %d^3 lower)" % gemm["h"],
                  "bound": "mfma", "achieved": round(gemm["tflops"], 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": round(gemm["tflops"] / FP64_PEAK_TFLOPS, 5), "launch_ms": round(gemm["ms"], 4),
                  "traffic": pmc.get("hbm_traffic_bytes_per_launch") if pmc else None,
                  "committed_pmc": pmc}
    pair_ops = n * m_rank
    gauss = cfg["kind"] in ("gaussian", "wendland")        # the culled sweep
    ek = "rbf_eval_gauss_cull_kernel" if gauss else "rbf_eval_kernel"
    hbm = {"achieved": round(by / te / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / te / 1e9 / HBM_PEAK_GBS, 6),
           "note": "algorithmic bytes 8d+8 per target; ~1e-3 of the HBM roofline by construction (N pair evaluations per target)"}
    pmc = committed_pmc(name, ek)
    pairs = gauss_pair_counts(name, n, dim, m_rank) if gauss else None
    if gauss and pairs:
        # culled sweep: VALU-issue bound on what survives the culling.  Lane-instructions = staged pair-slots x the
        # cut-off pre-test + evaluated pair-slots x the exp2 evaluation (both counted in the ISA of the inner loop),
        # pair counts from the committed counter run of the prof build (tools/gauss_pairs.py, same clouds)
        pre, ev = GAUSS_VALU_PER_PAIR[dim]
        lane_instr = pairs["staged_pairs"] * pre + pairs["evaluated_pair_lanes"] * ev
        ach = lane_instr / te
        r_eval = {"kernel": ek + " (+ two-level reorder of the targets and un-sort, inside the timed phase)", "bound": "valu",
                  "achieved": round(ach, 1), "peak": VALU_PEAK_LANE_INSTR, "unit": "fp64 lane-instructions/s",
                  "frac": round(ach / VALU_PEAK_LANE_INSTR, 4),
                  "valu_instr_per_staged_pair": pre, "valu_instr_per_evaluated_pair": ev,
                  "pairs": pairs, "algorithmic_pair_evals_per_s": round(pair_ops / te, 1),
                  "traffic": pmc.get("hbm_traffic_bytes_per_launch") if pmc else None, "committed_pmc": pmc, "hbm": hbm,
                  "note": "the tile culling leaves %.1f%% of the N x M pairs staged, the per-centre wave test evaluates %.2f%%; "
                          "%.0f%% of the evaluated lane slots are pairs inside the 2^-72 cut-off" %
                          (100.0 * pairs["staged_pairs"] / pair_ops, 100.0 * pairs["evaluated_pair_lanes"] / pair_ops,
                           100.0 * pairs["useful_pairs"] / max(pairs["evaluated_pair_lanes"], 1))}
    elif gauss:
        r_eval = dict(hbm, kernel=ek + " (+ cell sort of the targets)", bound="hbm",
                      traffic=pmc.get("hbm_traffic_bytes_per_launch") if pmc else None, committed_pmc=pmc,
                      algorithmic_pair_evals_per_s=round(pair_ops / te, 1))
        r_eval["note"] = ("VALU-bound on the surviving (un-culled) pairs, not HBM-bound (no committed pair counts for this shape): " + hbm["note"])
    else:
        # thin-plate sweep: VALU-issue bound.  VALU instructions per pair counted in the ISA of the inner loop
        # (2 centres x 2 targets per iteration; 2-D: 77 / 4 -- 15 fp64 + 4.25 32-bit ops)
        ipp = TPS_VALU_PER_PAIR[dim]
        ach = pair_ops * ipp / te
        r_eval = {"kernel": ek, "bound": "valu", "achieved": round(ach, 1), "peak": VALU_PEAK_LANE_INSTR,
                  "unit": "fp64 lane-instructions/s", "frac": round(ach / VALU_PEAK_LANE_INSTR, 4),
                  "instr_per_pair": ipp, "pair_evals_per_s": round(pair_ops / te, 1),
                  "traffic": pmc.get("hbm_traffic_bytes_per_launch") if pmc else None, "committed_pmc": pmc, "hbm": hbm}
    # dominant single kernel: the GEMM when the factorisation (mostly GEMM) outweighs the sweep
    gemm_dominant = r_gemm is not None and tf == tf and 0.55 * tf > te
    res["roofline"] = r_gemm if gemm_dominant else r_eval
    res["roofline_other"] = r_eval if gemm_dominant else r_gemm
    if tf == tf:
        res["init_as_a_unit"] = {"flops": flops, "tflops": round(flops / tf / 1e12, 4),
                                 "frac_of_fp64_mfma_peak": round(flops / tf / 1e12 / FP64_PEAK_TFLOPS, 5),
                                 "route": {1: "cholesky", 2: "shifted-SPD cholesky + Woodbury", 3: "pivoted LU"}.get(route, "?")}
        if gemm_dominant:
            # the object a reader wants first: fill + factorisation + solves as ONE unit against the fp64 MFMA peak; the
            # single top-level launch (37.5 % of the flops, timed alone on random operands) stays beside it
            res["roofline"] = dict(res["roofline"], init_as_a_unit=res["init_as_a_unit"],
                                   frac_kernel=res["roofline"]["frac"], frac_init_as_a_unit=res["init_as_a_unit"]["frac_of_fp64_mfma_peak"])
        res["solve_gflops"] = round(flops / tf / 1e9, 2)
    res["eval_only_mpts"] = round(m_rank / te / 1e6, 3)
    return res


def cpu_baseline(cfg, n, dim, m_total):
    """The CPU oracle (reference-order C restatement, kind "port") timed on this box's host cores on a
    bounded sample of the SAME step the GPU `value` covers: fill + factorisation + solves + sweep.
    (i) one thread -- the reference is single-threaded -- gives `value`; (ii) the sweep split over all
    host cores gives `all_cores` (the factorisation stays serial: the reference's gaxpy Cholesky /
    unblocked LU are Level-2 chains).  The factorisation is timed at N in {2048, 4096} and extrapolated
    with N^3; the fill and the sweep are timed on a sample and scaled linearly."""
    import concurrent.futures as cf
    import oracle_lib as orc
    # the GPU box gives one GPU a 16-core share (cpu_count reports the whole host)
    ncores = max(1, min(len(os.sched_getaffinity(0)), 16))

    def par_map(fn, y):
        chunks = np.array_split(np.arange(len(y)), ncores)
        with cf.ThreadPoolExecutor(ncores) as ex:      # ctypes releases the GIL inside the oracle
            list(ex.map(lambda idx: fn(np.ascontiguousarray(y[idx])), [c for c in chunks if len(c)]))

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
        return {"value": round(ms / dt / 1e6, 5), "unit": "M points/s", "cores": 1, "kind": "port",
                "sample": f"first {ms} of the {m_total} targets, N={n}; host DAG build {build:.2f} s (one-off, excluded on both sides)",
                "all_cores": {"cores": ncores, "note": "not measured: the oracle tree keeps its scratch inside the tree like the "
                                                       "reference (linear_simplex.h:51-58), so one tree cannot be walked by several threads"}}
    kind = {"gaussian": 0, "tps": 1, "wendland": 2}[cfg["kind"]]
    eps = 0.125 * n ** (1.0 / dim) if kind == 2 else orc.gaussian_eps(n, dim)
    x = orc.synth_centres(n, dim)
    f = orc.synth_response(x)
    # --- factorisation + solves at N in {2048, 4096}, reference order, one thread; N^3 extrapolation
    fact = {}
    for ns in ((2048, 4096) if n >= 2048 else (n,)):      # a configuration below 2048 centres is measured whole
        if ns > n:
            continue
        xs, fs = np.ascontiguousarray(x[:ns]), np.ascontiguousarray(f[:ns])
        t0 = time.perf_counter()
        phi = orc.rbf_fill(kind, 0.125 * ns ** (1.0 / dim) if kind == 2 else orc.gaussian_eps(ns, dim), xs)
        tfill = time.perf_counter() - t0
        t0 = time.perf_counter()
        if kind != 1:
            st, llt = orc.cholesky_decomp1(phi)
            w = orc.cholesky_solve(llt, fs)
        else:
            lu, perm, _ = orc.lu_decomp(phi)
            st, w = orc.lu_solve(lu, perm, fs)
        dts = time.perf_counter() - t0
        fl = (ns ** 3 / 3.0) if kind != 1 else (2.0 * ns ** 3 / 3.0)
        fact[ns] = {"fill_s": round(tfill, 3), "factor_solve_s": round(dts, 3), "gflops": round(fl / dts / 1e9, 3)}
    ns_max = max(fact)
    fill_full = fact[ns_max]["fill_s"] * (n / ns_max) ** 2
    solve_full = fact[ns_max]["factor_solve_s"] * (n / ns_max) ** 3
    # --- sweep sample: ~10 s of single-core work at ~50 M pair-evals/s
    ms = max(1000, int(5.0e8 // n))
    y = orc.synth_targets(0, ms, dim)
    wfull = np.resize(w, n)
    t0 = time.perf_counter()
    orc.rbf_eval(kind, eps, x, wfull, y)
    dte = time.perf_counter() - t0
    eval_full_1 = dte * (m_total / ms)
    ms_par = ms * min(ncores, 8)
    ypar = orc.synth_targets(0, ms_par, dim)
    t0 = time.perf_counter()
    par_map(lambda yc: orc.rbf_eval(kind, eps, x, wfull, yc), ypar)
    dtp = time.perf_counter() - t0
    eval_full_p = dtp * (m_total / ms_par)
    step1 = fill_full + solve_full + eval_full_1
    stepp = fill_full + solve_full + eval_full_p
    return {"value": round(m_total / step1 / 1e6, 6), "unit": "M points/s", "cores": 1, "kind": "port",
            "sample": f"whole step extrapolated from: fill + factor + solve at N={sorted(fact)} (N^2 / N^3 scaling from N={ns_max} "
                      f"to N={n}: fill {fill_full:.1f} s, factor+solve {solve_full:.1f} s) and the sweep of the first {ms} of {m_total} "
                      f"targets against all N={n} centres ({n * ms / dte / 1e6:.1f} M pair-evals/s -> {eval_full_1:.1f} s)",
            "eval_only_mpts": round(ms / dte / 1e6, 6), "factorisation": fact,
            "solve_gflops": fact[ns_max]["gflops"],
            "all_cores": {"cores": ncores, "value": round(m_total / stepp / 1e6, 6), "eval_only_mpts": round(ms_par / dtp / 1e6, 6),
                          "note": f"sweep of {ms_par} targets split over {ncores} threads; fill/factorisation serial as in the reference"}}


if __name__ == "__main__":
    main()
