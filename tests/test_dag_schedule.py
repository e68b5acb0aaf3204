"""CPU: the list scheduler of the opt-in task-DAG Cholesky (csrc/hip/chol_dag_sched.h) -- every list it builds must pass
its own replay check (dag_check_schedule: device claim rule, unit durations, every block fully updated and solved, no
two updates of one block in flight, nobody waits forever), for block counts from 2 to 128 and for cost models far from
the calibrated one (the list order comes from a SIMULATED execution; the check must hold whatever the real durations)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "chol_dag_study", "simulate.cpp")
INC = os.path.join(ROOT, "gsl-scattered-interpolation_amd", "csrc", "hip")


@pytest.fixture(scope="module")
def simulate(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dag") / "simulate")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", INC, SRC, "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("T", [2, 3, 4, 5, 8, 13, 32, 64, 128])
def test_default_list_passes_the_replay_check(simulate, T):
    out = subprocess.run([simulate, str(T)], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("check 0"), out


@pytest.mark.parametrize("args", [
    ["potrf=1", "chain_trsm=1", "chain_syrk=1"],                      # a chain far faster than the workers
    ["potrf=500", "chain_trsm=200"],                                  # ... and far slower
    ["step256=0.1", "step128=0.1", "upd_fixed256=0", "upd_fixed128=0", "fused_fixed=0", "fused_trsm=0.5"],
    ["kcb=1"], ["kcb=16"], ["near_rows=0"], ["near_rows=64", "urgent_rows=64"], ["express=0"], ["express=200"],
])
def test_lists_from_distorted_cost_models_pass_too(simulate, args):
    for T in (7, 40):
        out = subprocess.run([simulate, str(T)] + args, check=True, capture_output=True, text=True).stdout
        assert out.strip().endswith("check 0"), out
