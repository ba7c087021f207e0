"""pytest configuration: `gpu` marks tests that need a real MI355X (run with
`-m gpu` on the GPU box); everything else runs on CPU with `-m "not gpu"`."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()  # builds oracle/_build/libg2s_oracle.so on first use
    return oracle_lib


@pytest.fixture(scope="session")
def product():
    lib_path = os.path.join(ROOT, "gap2seq_amd", "libg2s_hip.so")
    if not os.path.exists(lib_path):
        import __graft_entry__
        __graft_entry__.build()
    from gap2seq_amd import lib
    lib.load_library()
    return lib


def pytest_runtest_logstart(nodeid, location):
    """G2S_TEST_PROGRESS=path: append the id of every test as it starts (diagnostics on the GPU box:
    the file is closed after every line, so it survives a run that has to be abandoned)."""
    path = os.environ.get("G2S_TEST_PROGRESS")
    if path:
        with open(path, "a") as f:
            f.write(nodeid + "\n")
