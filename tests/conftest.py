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
    from oracle.pyoracle import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.pyoracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libcroaring_ref.so not prebuilt here")
    return Ref()


@pytest.fixture(scope="session")
def engine():
    # No skip: on the GPU box a missing library or device must fail loudly.
    import torch  # noqa: F401  (first, so the engine binds to the HIP runtime torch loads)
    import croaring_amd
    eng = croaring_amd.Engine()
    yield eng
    eng.close()
