import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle is most of the GPU suite's wall time, and on a host with hundreds of logical cores torch's default thread count
    # makes it several times SLOWER (bench.py's cpu_baseline tuning: 16 threads 31 k edges/s, 128 threads 6.5 k edges/s)
    try:
        import torch

        if (os.cpu_count() or 1) > 16:
            torch.set_num_threads(16)
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """Write the achieved parity maxima of this session (tests/parity_record.py) next to the other GPU-box outputs."""
    import json

    from tests import parity_record

    if not parity_record.RECORDS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_r06.json"), "w") as f:
        json.dump({"exitstatus": int(exitstatus), "records": parity_record.RECORDS}, f, indent=1)
