"""pytest configuration: `gpu` marker, shared fixtures (oracle / product libraries)."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

# The checkers (oracle/) are OpenMP code and several of them run at once in the GPU suite (the background worker of the full-size
# tests next to the test in progress) on a CPU quota: libgomp's idle threads then spin the quota away (default: 300 000 spins).
# Measured on 8 CPUs with two 8-thread oracle runs side by side: 18.2 s each by default, 12.4 s with GOMP_SPINCOUNT=30000 - and
# 5.9 s against 6.9 s for one run alone.  Read by libgomp when the first checker library is loaded; child processes inherit it.
import os
os.environ.setdefault("GOMP_SPINCOUNT", "30000")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU comparison")
    config.addinivalue_line("markers", "fullsize_background(part): the test reads the results of tests/fullsize_worker.py, started right after collection")
    config.addinivalue_line("markers", "gpu_timing: compares timings - must not share the GPU with the background worker's product runs (moved towards the end of the suite)")


# ---- the two full-size Ravone-project tests: their checker runs (minutes of oracle time) happen in a background process that is
# started as soon as the tests are known to be selected, and the tests themselves run LAST - the oracle's minutes pass while the rest
# of the suite keeps the GPU busy (round 3's suite spent 202 of its 790 s waiting for exactly this)
_BACKGROUND = {}


def pytest_collection_modifyitems(config, items):
    late = [it for it in items if it.get_closest_marker("fullsize_background")]
    timing = [it for it in items if it.get_closest_marker("gpu_timing")] if late else []      # (the worker's GPU phase is over long before the end of the suite)
    if late:
        items[:] = [it for it in items if it not in late and it not in timing] + timing + late


def pytest_collection_finish(session):
    import subprocess
    import tempfile
    if session.config.option.collectonly:
        return
    parts = sorted({it.get_closest_marker("fullsize_background").args[0] for it in session.items if it.get_closest_marker("fullsize_background")})
    if not parts or _BACKGROUND:
        return
    out = Path(tempfile.mkdtemp(prefix="sf3d_fullsize_")) / "results.json"
    log = open(out.with_suffix(".log"), "w")
    proc = subprocess.Popen([sys.executable, str(ROOT / "tests" / "fullsize_worker.py"), str(out), *parts], stdout=log, stderr=subprocess.STDOUT)
    _BACKGROUND.update(proc=proc, out=out, log=log, parts=parts)


@pytest.fixture(scope="session")
def fullsize_results():
    """results of tests/fullsize_worker.py (waits for the background process: it has had the whole suite's time to finish)"""
    import json
    if not _BACKGROUND:
        pytest.fail("tests/fullsize_worker.py was not started (pytest_collection_finish did not see a fullsize_background test)")
    proc, out = _BACKGROUND["proc"], _BACKGROUND["out"]
    import time
    t_wait = time.time()
    try:
        rc = proc.wait(timeout=1500)
    except Exception:
        proc.kill()
        raise
    t_wait = time.time() - t_wait
    _BACKGROUND["log"].close()
    text = out.with_suffix(".log").read_text()[-3000:]
    assert out.exists(), f"fullsize_worker wrote no results (rc {rc}):\n{text}"
    res = json.loads(out.read_text())
    res["_rc"], res["_log"], res["_waited_s"] = rc, text, t_wait
    scratch = os.environ.get("GRAFT_REPO_ROOT")
    if scratch and os.path.isdir(os.path.join(scratch, "gpurun_out")):      # on the GPU box: how long the suite waited for the worker, and what its parts took
        with open(os.path.join(scratch, "gpurun_out", "fullsize_worker_times.txt"), "a") as f:
            f.write(json.dumps({"waited_s": t_wait, **{k: {q: v[q] for q in ("seconds_product", "seconds_total")} for k, v in res.items() if isinstance(v, dict) and "seconds_total" in v}}) + "\n")
    return res


def pytest_sessionfinish(session, exitstatus):
    proc = _BACKGROUND.get("proc")
    if proc is not None and proc.poll() is None:          # (-x stopped the session early: do not leave the worker behind)
        proc.kill()


@pytest.fixture(scope="session")
def oracle():
    from criteria3d_amd import build
    from tests import checkers
    if not checkers.ORACLE_LIB.exists():
        build.build_oracle(with_reference=False)
    return checkers.load_oracle()


@pytest.fixture(scope="session")
def product():
    """The HIP library.  Fails (never skips, never falls back) when it is missing."""
    from criteria3d_amd import build, capi
    build.build_product()
    return capi.load_product()
