import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # (round-5 review: the suite must stay well inside the driver's step limit.  `-m gpu` still runs everything;
    # the full-size / multi-GB cases -- ten seconds and more each, a third of the suite's time -- can be run apart:
    # `-m "gpu and not slow"` and `-m "gpu and slow"` are two shards of about equal length)
    config.addinivalue_line("markers", "slow: a gpu test of a full-size shape (>= 10 s); included by -m gpu")


def pytest_collection_modifyitems(config, items):
    """GPU tests run kernels that wait for each other (persistent look-ahead, resident solve, P2P
    exchanges).  Their waits are bounded or deadlock-free by construction -- but should a change
    break that, a test must END (pytest-timeout's thread method terminates the process, which tears
    the GPU context down) instead of holding the GPU box until the lease is killed."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


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


@pytest.fixture
def hooks_lib():
    """The TEST build of the library (fault-injection hooks compiled in, -DMI355X_TEST_HOOKS) in
    place of the product library for the duration of one test: the product library neither exports
    the hooks nor holds the code behind them.  Every handle the test makes dies before the swap
    back."""
    import gc
    from tests.helpers import lp_amd
    ctx = lp_amd().capi.test_build()
    L = ctx.__enter__()
    try:
        yield L
    finally:
        gc.collect()
        ctx.__exit__(None, None, None)
