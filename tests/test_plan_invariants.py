"""Host-side planning of the engine (msm_pipeline.h make_plan): invariants over the whole supported size range,
including pair counts far beyond what the GPU tests run (up to 2^31 - 1)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_invariants(tmp_path):
    exe = str(tmp_path / "t_plan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "constantine_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c_api", "t_plan.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "plans ok" in out.stdout, out.stdout[-2000:]
