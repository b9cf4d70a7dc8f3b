import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the weight-stationary 64 -> 64 kernel (csrc/conv_c64.hip) takes layers of >= 1024 row-steps in production; the parity tests lower the gate so
# that small shapes run it too (read once by the library, on its first migan_c64_conv_ok call)
os.environ.setdefault("MIGAN_C64_MIN_STEPS", "8")
# A/B knobs of the library (csrc/common.h MIGAN_KNOB) are re-read on every call in the test processes: one process runs a kernel family under
# several settings (e.g. MIGAN_DMA_TAPS9 = 0 / 2: tap-outer and tap-inner K order of the 3x3 layers on the same small shapes)
os.environ.setdefault("MIGAN_TEST_KNOBS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
