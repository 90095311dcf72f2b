import os
import sys

import pytest

# More than three frames in flight need more hardware queues than HIP's default of 4 per process (include/pt.h,
# pt_set_frames_in_flight); HIP reads the variable when it initialises, so it is set before anything imports torch.
# test_frames_in_flight_are_capped_by_the_hardware_queues covers a process that leaves it unset.
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 16:   # the GPU box exports HIP's default, 4
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_binding import load_oracle
    return load_oracle()
