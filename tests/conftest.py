import os
import sys

import pytest

os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # reference's libgomp workers must not spin between calls
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")


@pytest.fixture(scope="session")
def ref():
    """The compiled reference (oracle/_ref/libbsc_ref.so)."""
    from oracle.refbind import Ref, REF_SO
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libbsc_ref.so not built (make -C oracle ref; needs /root/reference)")
    return Ref()
