import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_finish(session):
    """A full `-m gpu` run: the expensive oracle halves of the parity tests (tests/oracle_jobs.py; every module lists its own in ORACLE_JOBS) start NOW, in child
    processes beside the tests, in the order the modules run.  Single tests (fewer than 30 selected GPU items) compute theirs inline as before."""
    gpu_items = [it for it in session.items if it.get_closest_marker("gpu")]
    if len(gpu_items) < 30 or os.environ.get("SRUKF_NO_ORACLE_POOL"):
        return
    from oracle_jobs import OraclePool
    pool = OraclePool()
    seen = []
    for it in gpu_items:
        mod = it.module
        if mod in seen:
            continue
        seen.append(mod)
        for kind, kw in getattr(mod, "ORACLE_JOBS", []):
            pool.submit(kind, **kw)
    session.config._oracle_pool = pool


def pytest_sessionfinish(session, exitstatus):
    pool = getattr(session.config, "_oracle_pool", None)
    if pool is not None:
        pool.close()


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def oracle_pool(request):
    """results of the oracle jobs started at collection time (tests/oracle_jobs.py); a job that was not started runs inline in get()"""
    pool = getattr(request.config, "_oracle_pool", None)
    if pool is None:
        from oracle_jobs import OraclePool
        pool = OraclePool(workers=1)
        request.addfinalizer(pool.close)
    return pool


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {f[:-4]: np.load(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".npz")}


@pytest.fixture(scope="session")
def srukf(pkg):
    """The HIP path through the C-ABI.  Fails loudly (no fallback) if the library is missing."""
    pkg.srukf.load_library()
    return pkg.srukf
