"""Device build of the kernels' elementary functions == host build of the same source text, bit for bit - so the CPU tests speak for
the kernels: tests/test_glibcmath.py (default build: the reference C library's log / exp / pow / cbrt reproduced operation by operation,
host build == libm on > 10^7 arguments per function) or tests/test_fastmath.py (-DSF3D_LIBM_GLIBC=0: the 0.50-ulp routines).  On a
default build this file therefore also compares the DEVICE with the C library of the box directly."""
import numpy as np
import pytest

from criteria3d_amd import capi

import sys, os
sys.path.insert(0, os.path.dirname(__file__))
from test_fastmath import exp_samples, fm, pow_samples, samples  # noqa: F401  (fixture)
from test_glibcmath import cbrt_ranges, exp_ranges, gl, log_ranges, pow_ranges  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def host(product, request):
    """host build of the set of routines the loaded product was compiled with: host("log", x), host("pow", x, y), ..."""
    if product.lib.sf3d_libm_set() == 1:
        g = request.getfixturevalue("gl")
        return lambda name, *a: g.call("gl_" + name, *a)
    f = request.getfixturevalue("fm")
    return lambda name, *a: f("fm_" + name, *a)


def dev1(product, which, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    fn = {"log": product.lib.sf3d_device_log, "exp": product.lib.sf3d_device_exp, "cbrt": product.lib.sf3d_device_cbrt}[which]
    product.check(fn(x.size, x.ctypes.data_as(capi.pd), y.ctypes.data_as(capi.pd)), "device_" + which)
    return y


def dev_pow(product, x, y):
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(x)
    product.check(product.lib.sf3d_device_pow(x.size, x.ctypes.data_as(capi.pd), y.ctypes.data_as(capi.pd), out.ctypes.data_as(capi.pd)), "device_pow")
    return out


def same_bits(a, b):
    """bit-identical, a nan equal to any nan (the payload of an invalid operation is the hardware's)"""
    nan = np.isnan(a) & np.isnan(b)
    return np.array_equal(a.view(np.int64)[~nan], b.view(np.int64)[~nan])


def test_box_libm_is_the_library_the_kernels_reproduce():
    """The water tests assert the BITS of a live oracle - which calls whatever libm this box has.  The identity holds where that is the
    library csrc/sf3d_glibcmath.inc reproduces (glibc 2.35, x86-64 FMA variants).  On another box the water tests fall back to
    WATER_RTOL (tests/tolerances.py: water_nodes_exact) and THIS test says why - one failure instead of 150."""
    from tests.tolerances import libm_probe
    ok, msg = libm_probe()
    assert ok, f"this box's libm is not the one the kernels' tables reproduce ({msg}): the live-oracle water tests ran at WATER_RTOL instead of bit identity"
    print(msg)


def test_default_build_is_the_faithful_set(product):
    """the shipped build evaluates the reference C library's functions (sf3d_glibcmath.inc); the 0.50-ulp set is a build option"""
    if os.environ.get("SF3D_PRODUCT_LIB"):
        pytest.skip("an alternative build is loaded on purpose")
    assert product.lib.sf3d_libm_set() == 1


def test_device_functions_are_the_c_librarys_on_every_range(product, gl):  # noqa: F811
    """default build: device == host build == glibc on the ranges of tests/test_glibcmath.py (special values, subnormal results,
    negative bases and random bit patterns included), 10^6 arguments per range"""
    if product.lib.sf3d_libm_set() != 1:
        pytest.skip("a -DSF3D_LIBM_GLIBC=0 build is loaded")
    n = 1_000_000
    for name, x in log_ranges(seed=31, n=n).items():
        assert same_bits(dev1(product, "log", x), gl.call("gl_log_libm", x)), name
    for name, x in exp_ranges(seed=32, n=n).items():
        assert same_bits(dev1(product, "exp", x), gl.call("gl_exp_libm", x)), name
    for name, x in cbrt_ranges(seed=33, n=n).items():
        assert same_bits(dev1(product, "cbrt", x), gl.call("gl_cbrt_libm", x)), name
    for name, (x, y) in pow_ranges(seed=34, n=n).items():
        assert same_bits(dev_pow(product, x, y), gl.call("gl_pow_libm", x, y)), name


def test_device_log_equals_host_build(product, host):
    for name, x in samples(seed=3, n=1_000_000).items():
        assert np.array_equal(dev1(product, "log", x).view(np.int64), host("log", x).view(np.int64)), name


def test_device_log_special_values(product):
    x = np.array([0.0, -1.0, np.inf, np.nan, 5e-324, 1.0])
    y = np.empty_like(x)
    product.check(product.lib.sf3d_device_log(x.size, x.ctypes.data_as(capi.pd), y.ctypes.data_as(capi.pd)), "device_log")
    assert y[0] == -np.inf and np.isnan(y[1]) and y[2] == np.inf and np.isnan(y[3]) and abs(y[4] - np.log(5e-324)) < 1e-12 and y[5] == 0.0


def test_device_pow_equals_host_build(product, host):
    for name, (x, y) in pow_samples(seed=7, n=1_000_000).items():
        assert np.array_equal(dev_pow(product, x, y).view(np.int64), host("pow", x, y).view(np.int64)), name


def test_device_pow_special_values(product, host):
    bases = np.array([0.0, 1.0, np.inf, np.nan, 5e-324, 1e-310, 0.5, 2.0, 1e300, 1e-300])
    exps = np.array([0.0, np.inf, -np.inf, np.nan, 2.5, -2.5, 1.0, 1e300, -1e300])
    x, y = [np.ascontiguousarray(a.ravel()) for a in np.meshgrid(bases, exps)]
    assert np.array_equal(dev_pow(product, x, y), host("pow", x, y), equal_nan=True)


def test_device_exp_equals_host_build(product, host):
    for name, x in exp_samples(seed=11, n=1_000_000).items():
        assert same_bits(dev1(product, "exp", x), host("exp", x)), name


def test_device_cbrt_equals_host_build(product, host):
    """the cbrt of the runoff links' Manning term: device == host build of the same text, bit for bit"""
    rng = np.random.default_rng(21)
    for name, x in {"depth^2": np.exp(rng.uniform(-24, 6, 1_000_000)), "whole range": np.exp(rng.uniform(-700, 700, 200_000)),
                    "edges": np.array([0.0, 1.0, 8.0, 27.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf])}.items():
        assert np.array_equal(dev1(product, "cbrt", x).view(np.int64), host("cbrt", x).view(np.int64)), name


def test_sweep_norm_is_the_exactly_rounded_sum_in_every_association(product):  # noqa: F811
    """The norm of a Jacobi iteration (water.cpp:592-600) is a sum of N non-negative terms; k_sweep, the paired passes, k_sweep_bnd
    and the ranks of a sharded run add them in different associations.  As double-doubles (sf3d_physics.inc) every association
    must round the SAME exact sum once: held here against math.fsum (exactly rounded) for terms of the sizes the solver sees
    (1e-12 ... 1 per node), for both association shapes and block counts from 1 to 2 048, and for a permuted input.  The plain-double
    sums of the same associations are reported next to it: they differ from one another, which is why the pair exists."""
    import math
    rng = np.random.default_rng(20241003)
    n = 1 << 20
    x = np.ascontiguousarray(10.0 ** rng.uniform(-12, 0, n) * rng.uniform(0.5, 1.0, n))
    x[rng.integers(0, n, n // 8)] = 0.0               # converged nodes
    exact = math.fsum(x.tolist())
    plain = set()
    for terms in (x, np.ascontiguousarray(x[rng.permutation(n)])):
        for assoc in (0, 1):
            for blocks in (1, 7, 256, 921, 2048):
                out = np.zeros(2)
                product.check(product.lib.sf3d_device_norm_sum(n, terms.ctypes.data_as(capi.pd), blocks, assoc, out.ctypes.data_as(capi.pd)), "device_norm_sum")
                assert out[0] == exact, (assoc, blocks, out[0], exact)
                assert abs(out[1] - exact) < 1e-9 * exact
                plain.add(float(out[1]))
    assert len(plain) > 1, "the plain-double sums of twenty associations all agree: the contrast leg measures nothing"
    # the degenerate sizes
    out = np.zeros(2)
    product.check(product.lib.sf3d_device_norm_sum(0, None, 4, 0, out.ctypes.data_as(capi.pd)), "device_norm_sum")
    assert out[0] == 0.0
    one = np.array([0.1])
    product.check(product.lib.sf3d_device_norm_sum(1, one.ctypes.data_as(capi.pd), 3, 1, out.ctypes.data_as(capi.pd)), "device_norm_sum")
    assert out[0] == 0.1
