import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    import bvh_pkg
    return bvh_pkg.load()


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def require_ref(present, what: str) -> None:
    """The reference-side pins (oracle/_ref/: binaries built from /root/reference by oracle/Makefile, git-ignored, shipped to the GPU box with the
    snapshot) must not vanish silently: a missing piece FAILS the test unless BVH_ALLOW_NO_REF=1 says the run is knowingly without them."""
    if present:
        return
    msg = f"{what} is missing (make -C oracle ref, in the container that has /root/reference); set BVH_ALLOW_NO_REF=1 to skip the reference-side checks"
    if os.environ.get("BVH_ALLOW_NO_REF") == "1":
        pytest.skip(msg)
    pytest.fail(msg)


@pytest.fixture
def sched_opts(ctx):
    """sched_opts(hploc="block", lbvh="block", sort_knobs=8) sets bvh_ctx options on the session ctx for one test; all are reset afterwards.
    (The library reads no environment variables: schedulers are chosen by size unless the host overrides them per context.)"""
    def setter(**kw):
        for k, v in kw.items():
            ctx.set_option(k, v)
    yield setter
    for k in ("hploc", "lbvh", "sort_knobs", "ploc"):
        ctx.set_option(k, 0)


GOLDEN = os.path.join(ROOT, "tests", "golden")
