import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import helpers
    return helpers.load_oracle()


@pytest.fixture(scope="session")
def hostlib():
    import helpers
    return helpers.load_host_primitives()


@pytest.fixture
def switch():
    """switch(name, value) sets an XHE_* switch of the library for one test.  The library reads its environment once (when the
    first context is created), so every change is followed by xhe_settings_reload(); the old values come back afterwards."""
    import os
    from xelis_he_amd._capi import load_library
    lib = load_library()
    saved = {}

    def set_switch(name, value):
        saved.setdefault(name, os.environ.get(name))
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(value)
        lib.xhe_settings_reload()

    yield set_switch
    for name, old in saved.items():
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old
    lib.xhe_settings_reload()
