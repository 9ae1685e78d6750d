"""GPU parity tests of the on-chip RRTMG_LW band sweeps (k_lw_cols, lw_cols_kernels.hpp: workers = (layer, column), the (layer, g-point)
intermediates in LDS, a sweep wave for the vertical recurrences) - the north-star mapping, selected with GEOSRAD_LW_PATH=cols.
Same gates as the default path (tests/test_gpu_lw.py): the reference's golden vectors, the pinned oracle, batching invisible."""
import os
import numpy as np
import pytest
from tests.conftest import load_golden, GOLDEN_CASES, FLUX, sub_columns

pytestmark = pytest.mark.gpu
TOL_FLUX = {4: 2e-3, 8: 1e-6}
TOL_DFDT = {4: 2e-5, 8: 1e-8}


@pytest.fixture(scope="module")
def cols_ctx():
    from geosradiation_gridcomp_amd.api import Context
    old = os.environ.get("GEOSRAD_LW_PATH")
    os.environ["GEOSRAD_LW_PATH"] = "cols"             # read by geosrad_create
    try:
        ctxs = {4: Context(4), 8: Context(8)}
    finally:
        if old is None:
            del os.environ["GEOSRAD_LW_PATH"]
        else:
            os.environ["GEOSRAD_LW_PATH"] = old
    yield ctxs
    for c in ctxs.values():
        c.close()


def _kind(rk):
    return "r4" if rk == 4 else "r8"


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("rk", [8, 4])
def test_cols_fluxes_match_reference_golden(cols_ctx, name, rk):
    ctx = cols_ctx[rk]
    inp, g, ih = load_golden(name)
    kind = _kind(rk)
    ctx.set_inhomogeneity(ih)
    bo = np.ones(16, dtype=np.int32) if f"{kind}_olrb" in g else None
    o = ctx.rrtmg_lw_columns(inp, band_output=bo)
    ctx.set_inhomogeneity(0)
    for k in FLUX:
        tol = TOL_DFDT[rk] if "dTs" in k else TOL_FLUX[rk]
        err = np.abs(o[k].astype(np.float64) - g[f"{kind}_{k}"].astype(np.float64)).max()
        assert err <= tol, (k, err)
    assert np.abs(o["clearCounts"] - g[f"{kind}_clearCounts"]).max() <= (0 if rk == 8 else 1)
    if bo is not None:
        assert np.abs(o["olrb"] - g[f"{kind}_olrb"]).max() <= TOL_FLUX[rk]
        assert np.abs(o["dolrb_dTs"] - g[f"{kind}_dolrb_dTs"]).max() <= TOL_DFDT[rk]


@pytest.mark.parametrize("rk", [8, 4])
def test_cols_taumol_tap_matches_reference(cols_ctx, rk):
    ctx = cols_ctx[rk]
    inp, g, _ = load_golden("lw_cloudy_ih2_137")
    kind = _kind(rk)
    taug, pfr = ctx.rrtmg_lw_taumol(sub_columns(inp, 2))
    rt = 1e-11 if rk == 8 else 5e-4
    np.testing.assert_allclose(taug, g[f"{kind}_taug2"], rtol=rt, atol=1e-30 if rk == 8 else 1e-12)
    np.testing.assert_allclose(pfr, g[f"{kind}_pfracs2"], rtol=rt, atol=0)


@pytest.mark.parametrize("rk", [8, 4])
def test_cols_equals_default_path_and_batching_is_invisible(cols_ctx, gpu_ctx, rk):
    """Fresh columns (ragged count: the last block of each class is partly filled), against the default band kernels and against the
    pinned oracle; a column's result does not depend on its batch or its neighbours (bitwise)."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    ctx = cols_ctx[rk]
    kind = _kind(rk)
    n = 333
    inp = synth.make_columns(n, 72, start=86_000, aerosol=True, cloudy_frac=0.55)
    ctx.set_inhomogeneity(1); gpu_ctx[rk].set_inhomogeneity(1); clib.set_inhomogeneity(1, kind)
    try:
        a = ctx.rrtmg_lw_columns(inp, band_output=np.ones(16, dtype=np.int32))
        a2 = ctx.rrtmg_lw_columns(inp, band_output=np.ones(16, dtype=np.int32))
        ctx.set_chunk(128)
        b = ctx.rrtmg_lw_columns(inp, band_output=np.ones(16, dtype=np.int32))
        ctx.set_chunk(131072)
        shard = {k: (np.ascontiguousarray(v[..., 100:205]) if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == n else v) for k, v in inp.items()}
        s = ctx.rrtmg_lw_columns(shard, band_output=np.ones(16, dtype=np.int32))
        d = gpu_ctx[rk].rrtmg_lw_columns(inp, band_output=np.ones(16, dtype=np.int32))
        r = clib.rrtmg_lw(sub_columns(inp, 48), kind)
    finally:
        ctx.set_chunk(131072)
        ctx.set_inhomogeneity(0); gpu_ctx[rk].set_inhomogeneity(0); clib.set_inhomogeneity(0, kind)
    for k in FLUX + ("clearCounts", "olrb", "dolrb_dTs"):
        np.testing.assert_array_equal(a[k], a2[k], err_msg=k)                 # run to run
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)                  # batches of 128 + 128 + 77
        if k in ("olrb", "dolrb_dTs"):
            np.testing.assert_array_equal(s[k], a[k][100:205], err_msg=k)
        else:
            np.testing.assert_array_equal(s[k], a[k][..., 100:205], err_msg=k)  # the same columns alone
    for k in FLUX:
        tol = (TOL_DFDT if "dTs" in k else TOL_FLUX)[rk]
        # the two device paths perform the same operations in the same order; fused multiply-adds may be contracted differently
        assert np.abs(a[k].astype(np.float64) - d[k]).max() <= 0.1 * tol, k
        assert np.abs(a[k][:, :48].astype(np.float64) - r[k]).max() <= tol, k
    np.testing.assert_array_equal(a["clearCounts"], d["clearCounts"])
