"""oracle/make_tables.py -- (re)generate the coefficient-table blobs shipped in
geosradiation_gridcomp_amd/data/ by RUNNING the reference's own initialisation code (oracle/_ref, i.e.
rrtmg_lw_ini / set_inhomogeneity compiled unmodified from /root/reference) and dumping the resulting
module state.  Needs oracle/_ref (hence /root/reference + flang); the blobs themselves are committed
so the GPU box never needs the reference.

  python -m oracle.make_tables
"""
import os
import numpy as np
from oracle import reflib
from geosradiation_gridcomp_amd.tableblob import read_blob, write_blob

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geosradiation_gridcomp_amd", "data")


def add_avg_cycle(path, kind):
    """AvgCyc11 of the Mg and SB indices (NRLSSI2.F90:60-117: `mgavgcyc`, `sbavgcyc`, 134 values each, module-private) recovered
    through the public interpolate_indices(): at solcycfr = 0 / 1 it returns elements 1 / 134, at the node (n-2) * intrvl_len +
    intrvl_len_hf element n.  Checked: the r8 recovery IS a 6- (Mg) / 4-decimal (SB) number - the source literal - and the r4
    recovery is that literal rounded to fp32, for all 134 elements.  Appended to the SW blob (isolvar = 1 needs them)."""
    dt = reflib.dtype_of(kind)
    il = dt(1.0) / dt(132); ilh = dt(0.5) * il
    pts = [dt(0.0)] + [dt(n - 2) * il + ilh for n in range(2, 134)] + [dt(1.0)]
    m8 = np.array([reflib.nrlssi2_interp(p, "r8") for p in [0.0] + [(n - 2) / 132 + 0.5 / 132 for n in range(2, 134)] + [1.0]], dtype=np.float64)
    lit_mg, lit_sb = np.round(m8[:, 0], 6), np.round(m8[:, 1], 4)
    assert np.abs(lit_mg - m8[:, 0]).max() < 1e-15 and np.abs(lit_sb - m8[:, 1]).max() < 1e-11
    mk = np.array([reflib.nrlssi2_interp(p, kind) for p in pts], dtype=dt)
    assert np.array_equal(mk[:, 0], lit_mg.astype(dt)) and np.array_equal(mk[:, 1], lit_sb.astype(dt))
    rb, arr = read_blob(path)
    arr["mgavgcyc"] = lit_mg.astype(dt); arr["sbavgcyc"] = lit_sb.astype(dt)
    write_blob(path, rb, arr)


def main():
    os.makedirs(DATA, exist_ok=True)
    for kind in ("r4", "r8"):
        reflib.dump_lw_tables(os.path.join(DATA, f"rrtmg_lw_{kind}.grtb"), kind)
        reflib.dump_sw_tables(os.path.join(DATA, f"rrtmg_sw_{kind}.grtb"), kind)
        add_avg_cycle(os.path.join(DATA, f"rrtmg_sw_{kind}.grtb"), kind)
        reflib.dump_chou_lw_tables(os.path.join(DATA, f"chou_lw_{kind}.grtb"), kind)
        reflib.dump_chou_sw_tables(os.path.join(DATA, f"chou_sw_{kind}.grtb"), kind)
    # condensate-inhomogeneity tables: beta (ih=1) and gamma (ih=2).  r4 is recovered bit-exactly through the
    # public zcw_lookup(); the r8 table is the decimal literal of each entry parsed as double, obtained by
    # snapping the (<=2e-14 rel.) r8 recovery to the shortest decimal of the exact r4 value.
    for ih, nm in ((1, "beta"), (2, "gamma")):
        p4 = os.path.join(DATA, f"xcw_{nm}_r4.grtb")
        reflib.dump_xcw(p4, ih, "r4")
        _, a = read_blob(p4)
        assert int(a["inexact_points"]) == 0
        p8 = os.path.join(DATA, f"xcw_{nm}_r8.grtb")
        reflib.dump_xcw(p8, ih, "r8")
        _, b = read_blob(p8)
        snap = np.array([float(str(v)) for v in a["xcw"].ravel(order="F")], dtype=np.float64).reshape(a["xcw"].shape, order="F")
        rel = np.abs(snap - b["xcw"]) / np.maximum(np.abs(snap), 1e-300)
        assert rel.max() < 1e-12, rel.max()
        write_blob(p8, 8, {"ih": np.int32(ih), "inexact_points": np.int32(0), "xcw": snap})
        print(nm, "r8 snapped; max rel diff of raw recovery", rel.max())
    reflib.set_inhomogeneity(0, "r4"); reflib.set_inhomogeneity(0, "r8")
    for f in sorted(os.listdir(DATA)):
        print(f, os.path.getsize(os.path.join(DATA, f)))


if __name__ == "__main__":
    main()
