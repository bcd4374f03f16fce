import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Make sure the native pieces exist before any test imports them: the HIP extension
    (hipcc cross-compiles without a GPU) and the C oracle.  Both are no-ops when up to date."""
    try:
        import __graft_entry__
        __graft_entry__.build()
    except Exception as e:                               # noqa: BLE001
        print("warning: build() failed before the test session: %r" % (e,))
    # GPU session: bring torch (used by the column-partition tests for device buffers and
    # torch.distributed) in NOW.  Imported for the first time two hundred GPU tests into the
    # process it crashed inside its own import once in eight runs of the suite (segmentation fault
    # in `import torch`, nothing of this package on the stack); at start-up it never has.
    mexpr = session.config.getoption("markexpr", "") or ""
    if "gpu" in mexpr and "not gpu" not in mexpr:
        try:
            import torch  # noqa: F401
        except Exception as e:                           # noqa: BLE001
            print("warning: torch is not importable: %r" % (e,))


@pytest.fixture(scope="session")
def golden():
    from tests import goldens
    return goldens.load()
