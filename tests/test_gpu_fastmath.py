"""Device build of the table-driven logarithm == host build of the same source, bit for bit (so the CPU accuracy
tests of tests/test_fastmath.py speak for the kernels)."""
import numpy as np
import pytest

from criteria3d_amd import capi

import sys, os
sys.path.insert(0, os.path.dirname(__file__))
from test_fastmath import fm, samples  # noqa: F401  (fixture)

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
