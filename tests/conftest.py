import os
import sys

import pytest

# A process gets four hardware queues by default; every engine context brings three streams, and streams beyond the queues share one --
# dependent launches then start 2-5 x later (tools/cu_mask_probe.hip section 3).  INTEGRATION.md asks embedding hosts to raise the limit
# before the HIP runtime starts; the test process does what a host does.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a HIP device: on a box without one they are skipped with that reason instead of failing one by one
    (ctt_hip_msm_available() is the library's own probe and never aborts)."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    try:
        from constantine_amd import _lib
        have = _lib.lib().ctt_hip_msm_available() == 1
    except Exception:   # the library itself is missing: let the tests fail loudly
        return
    if not have:
        skip = pytest.mark.skip(reason="no HIP device on this box (ctt_hip_msm_available() == 0)")
        for it in gpu_items:
            it.add_marker(skip)
