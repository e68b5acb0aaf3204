"""Compile a plain C program against include/gsl_sinterp.h + libgsl_sinterp.so (what a user of the
reference does with linear_simplex.h) and check the reference's known answers from C."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_links_and_reproduces_known_answers(pkg, tmp_path):
    libdir = os.path.dirname(pkg.library_path())
    exe = str(tmp_path / "dropin")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g"] if os.environ.get("GSL_SINTERP_ASAN") else []
    libname = os.path.basename(pkg.library_path())[3:-3]           # gsl_sinterp, or gsl_sinterp_asan under the sanitizer run
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", *san, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "dropin_known_answers.c"), "-o", exe,
                           "-L", libdir, "-l" + libname, "-lm", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "weather_stations.csv")], text=True)
    assert "ok" in out


def test_c_shard_rule_unit(pkg, tmp_path):
    """CPU-side unit test, in C, of the shard / gather bookkeeping of the multi-GPU entries."""
    libdir = os.path.dirname(pkg.library_path())
    exe = str(tmp_path / "shard_rule")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g"] if os.environ.get("GSL_SINTERP_ASAN") else []
    libname = os.path.basename(pkg.library_path())[3:-3]
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", *san, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "shard_rule.c"), "-o", exe,
                           "-L", libdir, "-l" + libname, "-lm", "-Wl,-rpath," + libdir])
    assert "ok" in subprocess.check_output([exe], text=True)
