"""Device build of the table-driven logarithm == host build of the same source, bit for bit (so the CPU accuracy
tests of tests/test_fastmath.py speak for the kernels)."""
import numpy as np
import pytest

from criteria3d_amd import capi

import sys, os
sys.path.insert(0, os.path.dirname(__file__))
from test_fastmath import exp_samples, fm, pow_samples, samples  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_device_log_equals_host_build(product, fm):  # noqa: F811
    for name, x in samples(seed=3, n=1_000_000).items():
        x = np.ascontiguousarray(x)
        y = np.empty_like(x)
        product.check(product.lib.sf3d_device_log(x.size, x.ctypes.data_as(capi.pd), y.ctypes.data_as(capi.pd)), "device_log")
        assert np.array_equal(y.view(np.int64), fm("fm_log", x).view(np.int64)), name


def test_device_log_special_values(product):
    x = np.array([0.0, -1.0, np.inf, np.nan, 5e-324, 1.0])
    y = np.empty_like(x)
    product.check(product.lib.sf3d_device_log(x.size, x.ctypes.data_as(capi.pd), y.ctypes.data_as(capi.pd)), "device_log")
    assert y[0] == -np.inf and np.isnan(y[1]) and y[2] == np.inf and np.isnan(y[3]) and abs(y[4] - np.log(5e-324)) < 1e-12 and y[5] == 0.0


def test_device_pow_equals_host_build(product, fm):  # noqa: F811
    for name, (x, y) in pow_samples(seed=7, n=1_000_000).items():
        x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
        out = np.empty_like(x)
        product.check(product.lib.sf3d_device_pow(x.size, x.ctypes.data_as(capi.pd), y.ctypes.data_as(capi.pd), out.ctypes.data_as(capi.pd)), "device_pow")
        assert np.array_equal(out.view(np.int64), fm("fm_pow", x, y).view(np.int64)), name


def test_device_pow_special_values(product, fm):  # noqa: F811
    bases = np.array([0.0, 1.0, np.inf, np.nan, 5e-324, 1e-310, 0.5, 2.0, 1e300, 1e-300])
    exps = np.array([0.0, np.inf, -np.inf, np.nan, 2.5, -2.5, 1.0, 1e300, -1e300])
    x, y = [np.ascontiguousarray(a.ravel()) for a in np.meshgrid(bases, exps)]
    out = np.empty_like(x)
    product.check(product.lib.sf3d_device_pow(x.size, x.ctypes.data_as(capi.pd), y.ctypes.data_as(capi.pd), out.ctypes.data_as(capi.pd)), "device_pow")
    assert np.array_equal(out, fm("fm_pow", x, y), equal_nan=True)


def test_device_exp_equals_host_build(product, fm):  # noqa: F811
    for name, x in exp_samples(seed=11, n=1_000_000).items():
        x = np.ascontiguousarray(x)
        y = np.empty_like(x)
        product.check(product.lib.sf3d_device_exp(x.size, x.ctypes.data_as(capi.pd), y.ctypes.data_as(capi.pd)), "device_exp")
        assert np.array_equal(y.view(np.int64), fm("fm_exp", x).view(np.int64)), name


def test_device_cbrt_equals_host_build(product, fm):  # noqa: F811
    """the cbrt of the runoff links' Manning term: device == host build of the same text, bit for bit"""
    rng = np.random.default_rng(21)
    for name, x in {"depth^2": np.exp(rng.uniform(-24, 6, 1_000_000)), "whole range": np.exp(rng.uniform(-700, 700, 200_000)),
                    "edges": np.array([0.0, 1.0, 8.0, 27.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf])}.items():
        x = np.ascontiguousarray(x)
        y = np.empty_like(x)
        product.check(product.lib.sf3d_device_cbrt(x.size, x.ctypes.data_as(capi.pd), y.ctypes.data_as(capi.pd)), "device_cbrt")
        assert np.array_equal(y.view(np.int64), fm("fm_cbrt", x).view(np.int64)), name


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
