"""world_size-2 gloo test of the multi-GPU path's plumbing (runs on CPU, no GPU needed):
target shards cover every target exactly once, the model broadcast delivers rank 0's
weights, and the gathered shards equal the single-process result.  The per-shard compute
stand-in is the CPU oracle (tests may use it); on the GPU box the same plumbing drives
the HIP eval sweep in bench.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, m_total, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    import oracle_lib as orc
    pkg = g.load_package()
    n, dim = 200, 2
    x = orc.synth_centres(n, dim)
    eps = orc.gaussian_eps(n, dim)
    w = torch.zeros(n, dtype=torch.float64)
    if rank == 0:                                   # only rank 0 solves
        w = torch.from_numpy(orc.rbf_solve(0, eps, x, orc.synth_response(x)))
    pkg.sharding.broadcast_model([w], 0)
    first, count = pkg.sharding.shard_bounds(m_total, world, rank)
    y = orc.synth_targets(first, count, dim) if count else np.zeros((0, dim))
    local = torch.from_numpy(orc.rbf_eval(0, eps, x, w.numpy(), y)) if count else torch.zeros(0, dtype=torch.float64)
    full = pkg.sharding.gather_shards(local, m_total, 0)
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("m_total", [1000, 1001, 1])
def test_two_rank_sharded_eval_matches_single_process(tmp_path, orc, m_total):
    out = str(tmp_path / "full.npy")
    port = 29500 + (os.getpid() + m_total) % 2000
    mp.spawn(_worker, args=(2, port, m_total, out), nprocs=2, join=True)
    got = np.load(out)
    n, dim = 200, 2
    x = orc.synth_centres(n, dim)
    eps = orc.gaussian_eps(n, dim)
    w = orc.rbf_solve(0, eps, x, orc.synth_response(x))
    want = orc.rbf_eval(0, eps, x, w, orc.synth_targets(0, m_total, dim))
    assert got.shape == (m_total,) and np.array_equal(got, want)


def test_shard_bounds_cover_exactly_once(pkg):
    for m in (0, 1, 7, 8, 9, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [pkg.sharding.shard_bounds(m, world, r) for r in range(world)]
            covered = sum(c for _, c in spans)
            assert covered == m
            nxt = 0
            for f, c in spans:
                assert c == 0 or f == nxt
                nxt = f + c if c else nxt
