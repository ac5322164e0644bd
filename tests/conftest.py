import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--parity-report", action="store_true", default=False,
                     help="measurement run: helpers.close() / check() write their numbers to the parity log WITHOUT asserting.  The session "
                          "prints a banner and exits non-zero whatever the tests did — a report run cannot be mistaken for a passing one.")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if config.getoption("--parity-report", default=False):
        import helpers
        helpers.REPORT_ONLY = True


@pytest.fixture(autouse=True)
def _cpu_threads_pinned():
    """The CPU suite (oracle + op-graph interpreter) is bound by torch's intra-op threads: 91 tests take 3.5 min in one process on an idle
    8-core box (profiles/r06_cpu_suite_one_process.log) and 20+ when the cores are shared with another job or when torch sizes its pool from a
    host that has more cores than this container may use.  Pin the pool to the cores this process can run on, and undo whatever a test
    (or a library it imports) changed before the next one."""
    import torch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    want = max(1, min(avail, 16))
    if torch.get_num_threads() != want:
        torch.set_num_threads(want)
    yield
    if torch.get_num_threads() != want:
        torch.set_num_threads(want)


def pytest_sessionfinish(session, exitstatus):
    if session.config.getoption("--parity-report", default=False):
        print("\n" + "=" * 100 + "\n  --parity-report: tolerance assertions were DISABLED for this session (numbers in the parity log).\n"
              "  This is a measurement run, not a test result: exit status forced to 3.\n" + "=" * 100)
        session.exitstatus = 3


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
