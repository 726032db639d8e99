import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a HIP device skips the gpu-marked tests (the drop-in symbols return
    NULL there and the reference's validators would dereference it).  An explicit `-m gpu` selection is never
    skipped: on the GPU box a missing device or library must fail loudly, not pass by skipping."""
    if "gpu" in (config.getoption("markexpr") or ""):
        return
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device on this box (select with -m gpu to make that an error)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


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
