"""Reader for the "GRTB" coefficient-table blobs (format documented in oracle/ref_glue.F90).

A blob is a flat list of named arrays (Fortran column-major) produced by running the reference's own
initialisation routines (`rrtmg_lw_ini`, `rrtmg_sw_ini`, `set_inhomogeneity`) and dumping the module
state: it is the *data* the reference keeps in `rrlw_kgNN`, `rrlw_tbl`, `rrlw_wvn`, ... after the
256->140 (LW) / 224->112 (SW) g-point reduction.
"""
import numpy as np


def read_blob(path):
    """Return (real_bytes, {name: ndarray}) with arrays in Fortran order (shape = Fortran dims)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:4] != b"GRTB":
        raise ValueError(f"{path}: not a GRTB blob")
    ver, rb = np.frombuffer(buf, dtype="<i4", count=2, offset=4)
    if ver != 1:
        raise ValueError(f"{path}: unsupported GRTB version {ver}")
    off = 12
    out = {}
    while True:
        name = buf[off:off + 32].split(b"\0")[0].decode()
        kind, ndim = np.frombuffer(buf, dtype="<i4", count=2, offset=off + 32)
        dims = np.frombuffer(buf, dtype="<i4", count=4, offset=off + 40)
        off += 56
        if name == "END":
            break
        n = int(np.prod(dims[:ndim])) if ndim > 0 else 1
        dt = {4: "<f4", 8: "<f8", -4: "<i4"}[int(kind)]
        nbytes = n * abs(int(kind))
        a = np.frombuffer(buf, dtype=dt, count=n, offset=off).copy()
        if ndim > 0:
            a = a.reshape(tuple(int(d) for d in dims[:ndim]), order="F")
        else:
            a = a.reshape(())
        out[name] = a
        off += nbytes + ((8 - nbytes % 8) % 8)
    return int(rb), out


def write_blob(path, real_bytes, arrays):
    """Write {name: ndarray} as a GRTB blob (arrays given in Fortran-shape, any memory order)."""
    with open(path, "wb") as f:
        f.write(b"GRTB")
        f.write(np.array([1, real_bytes], dtype="<i4").tobytes())

        def entry(name, kind, ndim, dims, data):
            f.write(name.encode().ljust(32, b"\0"))
            f.write(np.array([kind, ndim] + list(dims), dtype="<i4").tobytes())
            f.write(data)
            f.write(b"\0" * ((8 - len(data) % 8) % 8))

        for name, a in arrays.items():
            a = np.asarray(a)
            if a.dtype.kind in "iu":
                kind, dt = -4, "<i4"
            else:
                kind, dt = real_bytes, ("<f4" if real_bytes == 4 else "<f8")
            dims = list(a.shape) + [1] * (4 - a.ndim)
            entry(name, kind, a.ndim, dims, np.asfortranarray(a.astype(dt)).tobytes(order="F"))
        entry("END", 0, 0, [0, 0, 0, 0], b"")
