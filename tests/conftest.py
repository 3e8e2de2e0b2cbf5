import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


# ---- the GPU suite's time budget: the driver's step limit is 900 s, the suite is kept at <= 450 s on a cold
# box.  Every run prints the seconds per test FILE (where a new parametrisation went) and the total against the budget; soak-like
# parametrisations belong under tools/probes/soak_*, not here.
GPU_SUITE_BUDGET_S = 450.0
_file_seconds = {}


def pytest_runtest_logreport(report):
    if report.when in ('setup', 'call', 'teardown'):
        f = report.nodeid.split('::', 1)[0]
        _file_seconds[f] = _file_seconds.get(f, 0.0) + float(getattr(report, 'duration', 0.0))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _file_seconds:
        return
    total = sum(_file_seconds.values())
    tr = terminalreporter
    tr.write_sep('=', 'seconds per test file')
    for f, s in sorted(_file_seconds.items(), key=lambda kv: -kv[1]):
        tr.write_line('%8.1f s  %s' % (s, f))
    gpu = 'gpu' in (config.getoption('-m') or '') and 'not gpu' not in (config.getoption('-m') or '')
    tr.write_line('%8.1f s  total%s' % (total, (' (GPU suite budget %.0f s: %s)' % (GPU_SUITE_BUDGET_S, 'inside' if total <= GPU_SUITE_BUDGET_S else 'OVER -- move soak-like cases to tools/probes')) if gpu else ''))
