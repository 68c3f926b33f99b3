import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real AMD GPU (run with `-m gpu` on MI355X)")


def _device_count() -> int:
    try:
        from embiggen_amd import _lib

        return _lib.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    if _device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no AMD GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure both native pieces exist (hipcc cross-compiles without a GPU)."""
    from embiggen_amd import _lib
    from oracle import oracle

    _lib.build()
    oracle.build()


@pytest.fixture(scope="session")
def karate():
    import embiggen_amd as E

    return E.karate_club()


@pytest.fixture(scope="session")
def karate_oracle(karate):
    from oracle import oracle as O

    return O.OracleGraph(karate.row_ptr, karate.col_idx)
