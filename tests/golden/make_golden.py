"""tests/golden/make_golden.py -- regenerate the committed golden vectors by RUNNING THE REFERENCE
(oracle/_ref = the reference's Fortran compiled unmodified from /root/reference, see oracle/Makefile).

    python -m tests.golden.make_golden

Each .npz holds the generator parameters (inputs are re-created bit-identically by
geosradiation_gridcomp_amd.synth.make_columns), the reference outputs in both precisions (r4 = GEOS
default real, r8 = -fdefault-real-8) and a few intermediates.  Data only: no reference source text.
"""
import os
import numpy as np
from geosradiation_gridcomp_amd import synth
from oracle import reflib

HERE = os.path.dirname(os.path.abspath(__file__))
FLUX = ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs")

CASES = {
    # name: (make_columns kwargs, ih, band_output?)
    "lw_clear_72": (dict(ncol=16, nlay=72, aerosol=False, cloudy_frac=0.0), 0, False),
    "lw_aer_72": (dict(ncol=16, nlay=72, aerosol=True, cloudy_frac=0.0), 0, True),
    "lw_cloudy_ih1_72": (dict(ncol=24, nlay=72, aerosol=True, cloudy_frac=0.8), 1, True),
    "lw_cloudy_ih0_72": (dict(ncol=16, nlay=72, aerosol=False, cloudy_frac=0.8, start=1000), 0, False),
    "lw_cloudy_ih2_137": (dict(ncol=8, nlay=137, aerosol=True, cloudy_frac=0.9, start=5000), 2, False),
}


def main():
    for name, (kw, ih, bo) in CASES.items():
        inp = synth.make_columns(**kw)
        out = {"kw_json": np.array(repr(kw)), "ih": np.int32(ih)}
        band_output = np.ones(16, dtype=np.int32) if bo else None
        for kind in ("r4", "r8"):
            reflib.set_inhomogeneity(ih, kind)
            r = reflib.rrtmg_lw(inp, kind, band_output=band_output)
            for k in FLUX:
                out[f"{kind}_{k}"] = r[k]
            out[f"{kind}_clearCounts"] = r["clearCounts"]
            if bo:
                out[f"{kind}_olrb"] = r["olrb"]; out[f"{kind}_dolrb_dTs"] = r["dolrb_dTs"]
            # intermediates for the first 2 columns
            sub = {k: (v[..., :2] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == kw["ncol"] else v) for k, v in inp.items()}
            t = reflib.lw_setcoef_taumol(sub, kind)
            out[f"{kind}_taug2"] = t["taug"]; out[f"{kind}_pfracs2"] = t["pfracs"]
            # McICA sub-columns of the first 4 columns (LW seeding + SW-like seeding with 112 sub-columns)
            sub4 = {k: (v[..., :4] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == kw["ncol"] else v) for k, v in inp.items()}
            for tag, nsub, so in (("lw", 140, (1, 2, 3, 4)), ("sw", 112, (4, 3, 2, 1))):
                cl, ci_s, cl_s = reflib.mcica(sub4["zm"], sub4["alat"], int(inp["dyofyr"]), sub4["play"], sub4["cldf"], sub4["ciwp"],
                                              sub4["clwp"], nsub, seed_order=so, kind=kind)
                out[f"{kind}_mc_{tag}_cldy"] = cl.astype(np.uint8)
                out[f"{kind}_mc_{tag}_ciwp"] = ci_s; out[f"{kind}_mc_{tag}_clwp"] = cl_s
            reflib.set_inhomogeneity(0, kind)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
