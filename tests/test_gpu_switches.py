"""-m gpu: the diagnostic switches of INTEGRATION.md select simpler kernel variants (one tile per workgroup
instead of stream-K, one launch per sweep block instead of the dataflow kernel, no hipGraphs, ...).  They are
read once per process, so each variant runs the relevant parity tests in a child interpreter (one at a
time: the GPU box allows few concurrent GPU processes)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LINALG = ["tests/test_gpu_linalg.py", "-k", "cholesky or gemm or graph_replays or single_launch"]
RBF_INIT = ["tests/test_gpu_rbf.py", "-k", "repeated_init or tps_solver_routes or facade_rbf"]
RBF_SWEEP = ["tests/test_gpu_rbf.py", "-k", "culling or linearity or facade_rbf"]
BARY = ["tests/test_gpu_bary.py"]
LU = ["tests/test_gpu_linalg.py", "-k", "lu_"]

CASES = [
    ("GSL_SINTERP_NO_GRAPH", LINALG), ("GSL_SINTERP_NO_GRAPH", RBF_INIT),
    ("GSL_SINTERP_NO_STREAMK", LINALG), ("GSL_SINTERP_NO_HYBRID_SK", LINALG), ("GSL_SINTERP_NO_GEMM8", LINALG), ("GSL_SINTERP_NO_GEMM64", LINALG), ("GSL_SINTERP_NO_GEMM_GROUP", LINALG), ("GSL_SINTERP_NO_GEMM_PIPE", LINALG), ("GSL_SINTERP_NO_KN_STREAMK", LINALG),
    ("GSL_SINTERP_NO_FUSED_SHIFT", RBF_INIT), ("GSL_SINTERP_NO_LU_COOP", LU), ("GSL_SINTERP_NO_TRSM64", LU), ("GSL_SINTERP_NO_GRAPH", LU), ("GSL_SINTERP_NO_FOLD", LINALG), ("GSL_SINTERP_NO_FOLD", RBF_INIT),
    ("GSL_SINTERP_NO_DMA_GEMM", LINALG), ("GSL_SINTERP_NO_PANEL128", LINALG), ("GSL_SINTERP_NO_PANEL128", RBF_INIT),
    ("GSL_SINTERP_NO_DATAFLOW_TRSV", LINALG), ("GSL_SINTERP_NO_DATAFLOW_TRSV", RBF_INIT),
    ("GSL_SINTERP_NO_SORT", RBF_SWEEP), ("GSL_SINTERP_NO_CULL", RBF_SWEEP), ("GSL_SINTERP_SERIAL_CELL_ORDER", RBF_SWEEP),
    ("GSL_SINTERP_SORT_LEVELS", BARY), ("GSL_SINTERP_SORT_LEVELS", RBF_SWEEP),      # "1": the one-level (atomic) target sort for every batch size
    ("GSL_SINTERP_NO_SORT", BARY), ("GSL_SINTERP_NO_JUMP", BARY), ("GSL_SINTERP_NO_FASTDIV", BARY), ("GSL_SINTERP_NO_AFFINE_WALK", BARY), ("GSL_SINTERP_NO_SIDE_STREAM", BARY), ("GSL_SINTERP_NO_LEAFWALK", BARY),
]


@pytest.mark.parametrize("switch,selection", CASES, ids=[f"{s}-{sel[0].split('_')[-1][:-3]}{i}" for i, (s, sel) in enumerate(CASES)])
def test_variant_passes_parity(switch, selection):
    env = dict(os.environ)
    env[switch] = "1"
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + selection
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, f"{switch}=1: {' '.join(selection)}\n" + r.stdout[-3000:]
