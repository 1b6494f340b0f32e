import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle


def pytest_runtest_setup(item):
    # the package reads its GSVC_* switches once, at import (gsvc_amd/switches.py): every test starts from the environment as it is now
    from gsvc_amd import switches
    switches.reload()


@pytest.fixture(autouse=True)
def _switches_follow_the_environment(monkeypatch):
    """``monkeypatch.setenv`` / ``delenv`` of a GSVC_* switch inside a test takes effect at once (the switches are re-read)."""
    from gsvc_amd import switches
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def _setenv(name, value, *a, **k):
        setenv(name, value, *a, **k)
        switches.reload()

    def _delenv(name, *a, **k):
        delenv(name, *a, **k)
        switches.reload()

    monkeypatch.setenv, monkeypatch.delenv = _setenv, _delenv
    yield


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Every skip with its reason at the end of the run, whatever the verbosity (`-q` shows a skip as a bare "s"), and which host
    modules ran compiled: a record of the run should say why something did not run and what did."""
    skipped = terminalreporter.stats.get("skipped", [])
    if skipped:
        terminalreporter.write_sep("-", "skipped (reason)")
        for rep in skipped:
            reason = rep.longrepr[2] if isinstance(rep.longrepr, tuple) and len(rep.longrepr) == 3 else str(rep.longrepr)
            terminalreporter.write_line(f"{rep.nodeid}: {reason}")
    try:
        import gsvc_amd
        ch = gsvc_amd.compiled_host()
        terminalreporter.write_line(f"gsvc_amd host modules compiled: {len(ch)} ({', '.join(sorted(ch)) if ch else 'plain Python'})")
    except Exception:  # noqa: BLE001
        pass
