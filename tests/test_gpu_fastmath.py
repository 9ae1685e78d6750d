"""gr_div64 / gr_rcp64 / gr_sqrt64 (csrc/lw_kernels.hpp): the division, reciprocal and square root of the fp64 RRTMG_SW instantiation
(hardware v_rcp_f64 / v_rsq_f64 + Newton steps + one residual correction) against the correctly rounded results, in ulps."""
import ctypes

import numpy as np
import pytest

from geosradiation_gridcomp_amd import _lib

pytestmark = pytest.mark.gpu


def _run(a, b):
    L = _lib.lib()
    L.geosrad_dbg_fast64.restype = ctypes.c_int
    L.geosrad_dbg_fast64.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5
    n = a.size
    q, r, s = (np.empty(n) for _ in range(3))
    rc = L.geosrad_dbg_fast64(n, a.ctypes.data, b.ctypes.data, q.ctypes.data, r.ctypes.data, s.ctypes.data)
    assert rc == 0
    return q, r, s


def _ulps(x, ref):
    return np.abs(x - ref) / np.spacing(np.abs(ref))


def test_fast64_within_one_ulp():
    rng = np.random.default_rng(7)
    n = 1 << 20
    # the operands of the optics: 1e-12 .. 1e12 (optical depths, albedos, cosines, two-stream denominators), both signs for the quotient
    a = (10.0 ** rng.uniform(-12, 12, n)) * rng.choice([-1.0, 1.0], n)
    b = 10.0 ** rng.uniform(-12, 12, n)
    a[:4] = [0.0, 1.0, -0.0, 3.0]
    b[:4] = [1.0, 1.0, 2.0, 3.0]
    q, r, s = _run(a, b)
    assert _ulps(q[a != 0], (a / b)[a != 0]).max() <= 1.0
    assert _ulps(r, 1.0 / b).max() <= 1.0
    assert _ulps(s, np.sqrt(b)).max() <= 1.0
    assert q[0] == 0.0 and q[1] == 1.0 and q[3] == 1.0 and r[2] == 0.5
    # share of correctly rounded results (information for the record; the bound above is what the kernels rely on)
    assert (q == a / b).mean() > 0.9 and (s == np.sqrt(b)).mean() > 0.9


def test_fast64_sqrt_of_zero_and_exact_squares():
    b = np.array([0.0, 1.0, 4.0, 9.0, 0.25, 1e-300, 1e300])
    q, r, s = _run(np.ones_like(b), b)
    assert np.array_equal(s[:5], [0.0, 1.0, 2.0, 3.0, 0.5])
    assert _ulps(s[5:], np.sqrt(b[5:])).max() <= 1.0
