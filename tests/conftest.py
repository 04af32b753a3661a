import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's torch ops within the CPUs this process owns: torch's default of one OpenMP thread per VISIBLE CPU (128 on
    # the GPU boxes, 16 CPUs' worth of quota) gets the whole process throttled by the container's CFS quota (puflow_amd/_host.py)
    from puflow_amd._host import limit_host_threads
    limit_host_threads()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
