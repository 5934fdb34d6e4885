"""Background worker of the two FULL-SIZE Ravone-project tests (tests/test_gpu_ravone_project.py, tests/test_gpu_heat.py): the
product's runs (seconds of GPU time) and the checkers' runs (minutes of CPU time: the oracle takes ~0.7 s per computeStep at 5.85 M
nodes) as a process of its own, started by tests/conftest.py right after collection so that the oracle's minutes pass WHILE the rest
of the GPU suite runs.  Only comparison metrics are written (JSON); the asserts live in the tests.
usage: python tests/fullsize_worker.py <out.json> [water] [heat]"""
import json
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from criteria3d_amd import capi, catchment as cm  # noqa: E402
from tests import checkers  # noqa: E402
from tests.scenarios import ravone_project_model  # noqa: E402

COUNTERS = ("attempts", "accepted", "approximations", "sweeps", "courant_rejections", "linear_failures", "restores")


def rel(a, b, floor=1e-9):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def segment(sf, m, H0, dt0, steps, threads=16, heat=None, T0=None, hour=1, more=0):
    """build, take over (H, [T,] dt) through the state setters, `steps` computeStep calls of the dry hour (+ `more`, uninterrupted)"""
    sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
    cm.build(sf, m, threads=threads, heat=heat)
    sf.set_total_potential_bulk(0, H0)
    if T0 is not None:
        sf.set_temperature_bulk(0, T0)
    sf.check(sf.lib.sf3d_set_time_step(dt0), "set_time_step")
    sf.check(sf.lib.sf3d_initialize_balance(), "initialize_balance")
    if heat is not None:
        cm.apply_heat_forcing(sf, m, hour)
    base, hbase = sf.counters(), (sf.heat_counters() if heat is not None else None)
    out = []
    dts_all = []
    for n in (steps, more):
        if n <= 0:
            continue
        _, dts = cm.run_hour(sf, m, 0.0, max_steps=n)
        dts_all += dts
        c = sf.counters()
        r = {"dts": np.array(dts_all), "snap": cm.snapshot(sf, m), "work": {k: c[k] - base[k] for k in c}}
        if heat is not None:
            h = sf.heat_counters()
            r["heat_work"] = {k: h[k] - hbase[k] for k in h}
            r["T"] = sf.temperature(0, m.n)[m.ns:]
        out.append(r)
    return out


def compare(g, o):
    """metrics of one compared segment: product (g) against a checker (o)"""
    res = {"H_bits_equal": bool(np.array_equal(g["snap"]["H"], o["snap"]["H"])), "Se_bits_equal": bool(np.array_equal(g["snap"]["Se"], o["snap"]["Se"])),
           "steps": int(len(o["dts"])), "dts_equal": bool(len(g["dts"]) == len(o["dts"]) and np.allclose(g["dts"], o["dts"], rtol=1e-12, atol=0)),
           "rel_H": rel(g["snap"]["H"], o["snap"]["H"]), "abs_Se": float(np.max(np.abs(g["snap"]["Se"] - o["snap"]["Se"]))),
           "scalars": {q: [float(g["snap"][q]), float(o["snap"][q])] for q in ("total_water", "storage", "runoff", "drainage", "lateral")},
           "work_product": {k: int(g["work"][k]) for k in COUNTERS}, "work_checker": {k: int(o["work"][k]) for k in COUNTERS}}
    if "T" in g:
        res["rel_T"] = rel(g["T"], o["T"])
        res["heat_work_product"], res["heat_work_checker"] = g["heat_work"], o["heat_work"]
    return res


def water_product(product):
    """the product's runs of tests/test_gpu_ravone_project.py::test_project_full_size_runoff_regime_matches_oracle"""
    m = ravone_project_model(None)
    t0 = time.time()
    product.check(product.lib.sf3d_reset_solver_state(), "reset")
    cm.build(product, m)
    n0, _ = cm.run_hour(product, m, 25.0)
    warm = product.counters()
    H0, dt0 = product.total_potential(0, m.n), product.lib.sf3d_get_time_step()
    g50, g300 = segment(product, m, H0, dt0, 50, more=250)            # 300 uninterrupted steps, looked at after 50 and after 300
    cm.run_hour(product, m, 0.0, max_steps=100)
    H1, dt1 = product.total_potential(0, m.n), product.lib.sf3d_get_time_step()
    late, = segment(product, m, H1, dt1, 100)
    product.lib.sf3d_clean()
    return dict(m=m, n0=n0, warm=warm, H0=H0, dt0=dt0, H1=H1, dt1=dt1, g50=g50, g300=g300, late=late, t0=t0, t_gpu=time.time() - t0,
                faithful=product.lib.sf3d_libm_set() == 1)


def water_checkers(p, results):
    m, H0, dt0, H1, dt1 = p["m"], p["H0"], p["dt0"], p["H1"], p["dt1"]
    t0 = time.time()
    oracle, second = checkers.load_oracle(), checkers.load_oracle_copy("second")
    if p["faithful"]:
        # default build (the reference C library's elementary functions): ONE run of the glibc oracle - the pin - for all 300
        # uninterrupted steps, looked at after 50 and after 300
        with ThreadPoolExecutor(2) as pool:          # (ctypes calls release the interpreter lock)
            j_pin = pool.submit(segment, oracle, m, H0, dt0, 50, 8, None, None, 1, 250)
            j_late = pool.submit(segment, second, m, H1, dt1, 100, 4)
            o50, o300 = j_pin.result(); o_late, = j_late.result()
        long_checker, libs = "glibc oracle", (oracle, second)
    else:
        # a -DSF3D_LIBM_GLIBC=0 build: the 300 steps against the oracle's fast-math twin, the first 50 against the glibc oracle
        twin = checkers.load_oracle_fastmath()
        with ThreadPoolExecutor(3) as pool:
            j_twin = pool.submit(segment, twin, m, H0, dt0, 300, 6)
            j_pin = pool.submit(segment, oracle, m, H0, dt0, 50, 3)
            j_late = pool.submit(segment, second, m, H1, dt1, 100, 3)
            o50, = j_pin.result(); o_late, = j_late.result(); o300, = j_twin.result()
        long_checker, libs = "fast-math twin", (oracle, twin, second)
    for sf in libs:
        sf.lib.sf3d_clean()
    results["water"] = {"nodes": int(m.n), "surface_nodes": int(m.ns), "hour0_steps": int(p["n0"]), "hour0_courant_rejections": int(p["warm"]["courant_rejections"]),
                        "finite": bool(np.all(np.isfinite(H0)) and np.all(np.isfinite(H1))),
                        "pin": compare(p["g50"], o50), "uninterrupted": compare(p["g300"], o300), "uninterrupted_checker": long_checker,
                        "late": compare(p["late"], o_late),
                        "seconds_product": p["t_gpu"], "seconds_total": p["t_gpu"] + time.time() - t0}


HEAT = cm.Heat(water=True, latent=True, save_mode=0)


def heat_product(product):
    """the product's runs of tests/test_gpu_heat.py::test_heat_project_full_size_fifty_steps"""
    m = cm.with_heat_surface(ravone_project_model(None))
    t0 = time.time()
    product.check(product.lib.sf3d_reset_solver_state(), "reset")
    cm.build(product, m, heat=HEAT)
    cm.apply_heat_forcing(product, m, 0)
    n0, _ = cm.run_hour(product, m, 20.0)
    H0, T0, dt0 = product.total_potential(0, m.n), product.temperature(0, m.n), product.lib.sf3d_get_time_step()
    g, = segment(product, m, H0, dt0, 50, heat=HEAT, T0=T0)
    product.lib.sf3d_clean()
    return dict(m=m, n0=n0, H0=H0, T0=T0, dt0=dt0, g=g, t_gpu=time.time() - t0)


def heat_checker(p, results):
    m, H0, T0 = p["m"], p["H0"], p["T0"]
    t0 = time.time()
    oracle = checkers.load_oracle()
    o, = segment(oracle, m, H0, p["dt0"], 50, threads=12, heat=HEAT, T0=T0)
    oracle.lib.sf3d_clean()
    results["heat"] = {"nodes": int(m.n), "hour0_steps": int(p["n0"]), "finite": bool(np.all(np.isfinite(H0)) and np.all(np.isfinite(T0[m.ns:]))),
                       "fifty": compare(p["g"], o), "seconds_product": p["t_gpu"], "seconds_total": p["t_gpu"] + time.time() - t0}


def release_gpu():
    """The product's runs are done: give the device back (context, queues, streams) - minutes of checker time follow, during which
    the suite's multi-rank tests start several GPU processes of their own on the same device."""
    import ctypes
    try:
        ctypes.CDLL("libamdhip64.so").hipDeviceReset()
    except OSError:
        pass


def main():
    out = Path(sys.argv[1])
    parts = sys.argv[2:] or ["water", "heat"]
    product = capi.load_product()
    results = {}
    try:
        # every product run first (under a minute of GPU time in all, at the start of the suite), then the GPU is released and the
        # checkers take their minutes of CPU time
        pw = water_product(product) if "water" in parts else None
        ph = heat_product(product) if "heat" in parts else None
        del product
        release_gpu()
        if pw is not None:
            water_checkers(pw, results)
            out.write_text(json.dumps(results))
        if ph is not None:
            heat_checker(ph, results)
    finally:
        out.write_text(json.dumps(results))          # what was finished is reported even if a later part fails


if __name__ == "__main__":
    main()
    sys.stdout.flush(); sys.stderr.flush()
    import os
    os._exit(0)          # (the device was reset under the product library: skip its static destructors)
