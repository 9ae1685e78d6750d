"""CPU: pin the plain-C oracle (oracle/lw_oracle_impl.h) against (a) the committed golden vectors produced by
the reference itself and (b), when oracle/_ref is present, the reference's Fortran run live."""
import numpy as np
import pytest
from tests.conftest import load_golden, GOLDEN_CASES, FLUX, sub_columns
from oracle import clib, reflib


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("kind", ["r4", "r8"])
def test_oracle_matches_golden_bitwise(name, kind):
    inp, g, ih = load_golden(name)
    clib.set_inhomogeneity(ih, kind)
    bo = np.ones(16, dtype=np.int32) if f"{kind}_olrb" in g else None
    o = clib.rrtmg_lw(inp, kind, band_output=bo, intermediates=True)
    assert o["rc"] == 0
    for k in FLUX:
        np.testing.assert_array_equal(o[k], g[f"{kind}_{k}"], err_msg=k)     # bit-exact
    np.testing.assert_array_equal(o["clearCounts"], g[f"{kind}_clearCounts"])
    if bo is not None:
        np.testing.assert_array_equal(o["olrb"], g[f"{kind}_olrb"])
        np.testing.assert_array_equal(o["dolrb_dTs"], g[f"{kind}_dolrb_dTs"])
    np.testing.assert_array_equal(o["taug"][:2], g[f"{kind}_taug2"])
    np.testing.assert_array_equal(o["pfracs"][:2], g[f"{kind}_pfracs2"])
    s4 = sub_columns(inp, 4)
    for tag, nsub, so in (("lw", 140, (1, 2, 3, 4)), ("sw", 112, (4, 3, 2, 1))):
        cl, ci, cw = clib.mcica(s4["zm"], s4["alat"], int(inp["dyofyr"]), s4["play"], s4["cldf"], s4["ciwp"], s4["clwp"], nsub,
                                seed_order=so, prec=kind)
        np.testing.assert_array_equal(cl, g[f"{kind}_mc_{tag}_cldy"])
        np.testing.assert_array_equal(ci, g[f"{kind}_mc_{tag}_ciwp"])
        np.testing.assert_array_equal(cw, g[f"{kind}_mc_{tag}_clwp"])
    clib.set_inhomogeneity(0, kind)


def test_kiss_known_answers():
    # the only known-answer test in the reference: cloud_subcol_gen.F90:579-605
    assert np.float32(clib.kiss_to_real(-2**31, "f32")) == np.float32(8.9406967e-08)
    assert np.float32(clib.kiss_to_real(2**31 - 1, "f32")) == np.float32(0.9999999)
    r = clib.kiss_stream((1, 2, 3, 4), 1000, "f32")
    assert r.min() > 0.0 and r.max() < 1.0
    # stream is a pure function of the seeds; fp64 stream is the same integers mapped in double
    r8 = clib.kiss_stream((1, 2, 3, 4), 1000, "f64")
    assert np.abs(r8 - r).max() < 1e-7


def test_oracle_input_checks():
    inp, _, _ = load_golden("lw_clear_72")
    bad = dict(inp); bad["tlay"] = inp["tlay"].copy(); bad["tlay"][3, 2] = -1.0
    assert clib.rrtmg_lw(bad, "r4")["rc"] == 101                 # negative values in input: tlay
    bad = dict(inp); bad["play"] = inp["play"].copy(); bad["play"][40, 1] = 500.0   # lower-atm layer above upper
    assert clib.rrtmg_lw(bad, "r4")["rc"] == 1                   # RRTMG LW pressure misordering
    bad = dict(inp); bad["cloudLM"] = 5; bad["cloudMH"] = 5
    assert clib.rrtmg_lw(bad, "r4")["rc"] == 5                   # invalid pressure super-layers!


@pytest.mark.skipif(not reflib.available("r4"), reason="oracle/_ref not built (needs /root/reference + flang)")
@pytest.mark.parametrize("kind", ["r4", "r8"])
def test_oracle_matches_live_reference(kind):
    from geosradiation_gridcomp_amd import synth
    inp = synth.make_columns(12, 72, start=777, aerosol=True, cloudy_frac=0.7)
    for ih in (1, 0):
        reflib.set_inhomogeneity(ih, kind); clib.set_inhomogeneity(ih, kind)
        r = reflib.rrtmg_lw(inp, kind); o = clib.rrtmg_lw(inp, kind, intermediates=True)
        for k in FLUX:
            np.testing.assert_array_equal(o[k], r[k])
        np.testing.assert_array_equal(o["clearCounts"], r["clearCounts"])
        t = reflib.lw_setcoef_taumol(inp, kind)
        np.testing.assert_array_equal(o["taug"], t["taug"])
        np.testing.assert_array_equal(o["pfracs"], t["pfracs"])
    z = np.random.default_rng(0).random(500)
    s = np.random.default_rng(1).choice([0.5, 0.71, 1.0], 500)
    reflib.set_inhomogeneity(1, kind); clib.set_inhomogeneity(1, kind)
    np.testing.assert_array_equal(clib.zcw_lookup(z, s, kind), reflib.zcw_lookup(z, s, kind))
    reflib.set_inhomogeneity(0, kind); clib.set_inhomogeneity(0, kind)


@pytest.mark.parametrize("kind", ["r4", "r8"])
@pytest.mark.parametrize("ice", [0, 1, 2, 4])
def test_oracle_cldprmc_iceflags_match_golden_bitwise(kind, ice):
    """LW cldprmc with the ice parameterisations other than GEOS's default 3 (rrtmg_lw_cldprmc.F90:138-226,270-316): fluxes of the
    reference itself for iceflag 0, 1, 2, 4 (tests/golden/lw_iceflags_72.npz, made by running oracle/_ref)."""
    inp, g, ih = load_golden("lw_iceflags_72")
    clib.set_inhomogeneity(ih, kind)
    o = clib.rrtmg_lw(inp, kind, iceflg=ice)
    clib.set_inhomogeneity(0, kind)
    assert o["rc"] == 0
    for k in FLUX:
        np.testing.assert_array_equal(o[k], g[f"{kind}_ice{ice}_{k}"], err_msg=k)
    np.testing.assert_array_equal(o["clearCounts"], g[f"{kind}_ice{ice}_clearCounts"])
    if ice != 0:     # the parameterisations really differ on these columns
        assert np.abs(g[f"{kind}_ice{ice}_uflx"] - g[f"{kind}_ice0_uflx"]).max() > 0.05
