import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")
    # a fresh checkout has no in-tree libelimaloc_hip.so (it is git-ignored): build it once (hipcc cross-compiles without a GPU)
    lib = os.path.join(ROOT, "elimaloc_amd", "libelimaloc_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "elimaloc_amd", "csrc")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O
