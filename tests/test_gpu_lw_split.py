"""GPU tests of the two-kernel RRTMG_LW band sweeps (GEOSRAD_LW_PATH=split: k_lw_cells + k_lw_sweep, lw_split_kernels.hpp): the
k-distribution as a layer-parallel kernel of its own, the vertical recurrences from the parked indices.  The arithmetic per cell is
k_lw_bands' (the compiler's choice of fused multiply-adds aside): the fluxes must agree with the default path's to rounding - 1e-11 W m-2
in fp64 -, on clear, cloudy, aerosol, 137-layer and ragged batches; plus the reference's golden vectors through this path."""
import os
import numpy as np
import pytest
from tests.conftest import load_golden, GOLDEN_CASES, FLUX

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def split_ctx():
    from geosradiation_gridcomp_amd.api import Context
    old = os.environ.get("GEOSRAD_LW_PATH")
    os.environ["GEOSRAD_LW_PATH"] = "split"             # read by geosrad_create
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


@pytest.mark.parametrize("rk", [8, 4])
@pytest.mark.parametrize("case", [dict(ncol=700, nlay=72, cloudy_frac=0.6, aerosol=True), dict(ncol=333, nlay=72, cloudy_frac=0.0, aerosol=False),
                                  dict(ncol=300, nlay=137, cloudy_frac=0.9, aerosol=True), dict(ncol=1500, nlay=72, cloudy_frac=1.0, aerosol=False)])
def test_split_path_agrees_with_the_default_path(gpu_ctx, split_ctx, rk, case):
    from geosradiation_gridcomp_amd import synth
    inp = synth.make_columns(start=9100, **case)
    bo = np.ones(16, dtype=np.int32)
    for ih in (1, 0):
        gpu_ctx[rk].set_inhomogeneity(ih); split_ctx[rk].set_inhomogeneity(ih)
        a = gpu_ctx[rk].rrtmg_lw_columns(inp, band_output=bo)
        b = split_ctx[rk].rrtmg_lw_columns(inp, band_output=bo)
        np.testing.assert_array_equal(a["clearCounts"], b["clearCounts"])
        for k in FLUX + ("olrb", "dolrb_dTs"):
            err = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()
            # fp32: a product rounded once instead of twice can move a cell to the neighbouring entry of the 10 000-entry transmittance table
            assert err <= ((1e-11 if rk == 8 else 1e-3) if "dTs" not in k else (1e-13 if rk == 8 else 1e-5)), (k, ih, err)
    gpu_ctx[rk].set_inhomogeneity(0); split_ctx[rk].set_inhomogeneity(0)


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("rk", [8, 4])
def test_split_fluxes_match_reference_golden(split_ctx, name, rk):
    ctx = split_ctx[rk]
    inp, g, ih = load_golden(name)
    kind = "r4" if rk == 4 else "r8"
    ctx.set_inhomogeneity(ih)
    o = ctx.rrtmg_lw_columns(inp)
    ctx.set_inhomogeneity(0)
    for k in FLUX:
        tol = (2e-5 if rk == 4 else 1e-8) if "dTs" in k else (2e-3 if rk == 4 else 1e-6)
        assert np.abs(o[k].astype(np.float64) - g[f"{kind}_{k}"].astype(np.float64)).max() <= tol, k
