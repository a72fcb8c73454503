"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """Compile libsrgan_hip.so when it is missing or was built from other kernel sources (judged by content -- the
    source id compiled into the library -- because file times do not survive the copy to the GPU box).  hipcc
    cross-compiles for gfx950 without a GPU.  The product itself never builds on demand: it fails loudly instead."""
    try:
        import srgan_amd  # noqa: F401
        from srgan_amd import _build
        if not _build.is_current():
            _build.build()
    except Exception as error:           # the tests that need the library then report the real problem
        print(f'[conftest] could not build libsrgan_hip.so: {error}', file=sys.stderr)


# Facts a reader of the `-q` tail needs (which batch the bench-size parity test really ran at, which collective backends
# ran): tests append lines here and the terminal summary prints them after the pass / fail counts.
PARITY_NOTES = []


def pytest_terminal_summary(terminalreporter):
    for line in PARITY_NOTES:
        terminalreporter.write_line('[parity] ' + line)
    skipped = terminalreporter.stats.get('skipped', [])
    for report in skipped:                                   # a skip is never silent, even under -q
        reason = report.longrepr[2] if isinstance(report.longrepr, tuple) else str(report.longrepr)
        terminalreporter.write_line(f'[skipped] {report.nodeid}: {reason}')
