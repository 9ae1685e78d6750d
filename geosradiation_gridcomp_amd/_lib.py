"""Loader for the C-ABI shared library (include/geosrad.h) and its build recipe.

The product has no CPU path: if libgeosrad.so cannot be loaded, or no HIP device is visible, everything
here raises.  (oracle/ is test infrastructure and is never imported from this package.)
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.environ.get("GEOSRAD_LIB") or os.path.join(CSRC, "libgeosrad.so")   # override only for A/B kernel experiments
DATA = os.path.join(HERE, "data")

EXPORTS = [
    "geosrad_create", "geosrad_destroy", "geosrad_last_error", "geosrad_real_kind", "geosrad_set_chunk",
    "geosrad_workspace_bytes", "geosrad_set_tables_lw", "geosrad_load_tables_lw", "geosrad_set_inhomogeneity",
    "geosrad_load_inhomogeneity", "geosrad_set_corr_lengths", "geosrad_rrtmg_lw", "geosrad_rrtmg_lw_dev",
    "geosrad_check", "geosrad_profile", "geosrad_profile_read", "geosrad_kernel_name", "geosrad_rrtmg_lw_taumol", "geosrad_mcica", "geosrad_clearcounts",
    "geosrad_set_tables_sw", "geosrad_load_tables_sw", "geosrad_rrtmg_sw", "geosrad_rrtmg_sw_dev", "geosrad_rrtmg_sw_taumol", "geosrad_mcica_dev",
    "geosrad_set_tables_chou_lw", "geosrad_load_tables_chou_lw", "geosrad_irrad", "geosrad_irrad_dev",
    "geosrad_lw_driver_rrtmg_dev", "geosrad_sw_driver_rrtmg_dev", "geosrad_lw_update_flx_dev", "geosrad_sw_update_export_dev",
    "geosrad_rad_tendencies_dev", "geosrad_lw_chou_post_dev", "geosrad_rrtmg_lw_rats_dev", "geosrad_lw_driver_rrtmg_rats_dev", "geosrad_lw_update_rats_dev", "geosrad_lw_update_bands_dev", "geosrad_sw_update_surface_dev",
    "geosrad_set_tables_chou_sw", "geosrad_load_tables_chou_sw", "geosrad_sorad", "geosrad_sorad_dev",
    "geosrad_read_table", "geosrad_rrtmg_sw_cldprmc",
]

_lib = None


def build(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "geosrad.h"))
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in srcs):
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # one source, three objects compiled in parallel: fp32 kernels, fp64 kernels, the extern "C" layer
    src = os.path.join(CSRC, "geosrad.hip")
    objdir = os.path.join(os.path.dirname(HERE), "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    jobs = []
    for part in (4, 8, 0):
        obj = os.path.join(objdir, f"geosrad_part{part}.o")
        jobs.append((obj, subprocess.Popen([hipcc, *flags, f"-DGEOSRAD_PART={part}", "-c", src, "-o", obj])))
    for obj, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, pr.args)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[o for o, _ in jobs], "-o", SO])
    return SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise RuntimeError(f"{SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        # PyTorch ships its own copy of the HIP runtime; when it is going to be used in this process (device tensors and streams
        # handed to the `_dev` entry points) it has to be the one this library binds to, so it is loaded first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(SO)
        L.geosrad_last_error.restype = ctypes.c_char_p
        L.geosrad_last_error.argtypes = [ctypes.c_void_p]
        L.geosrad_workspace_bytes.restype = ctypes.c_size_t
        L.geosrad_workspace_bytes.argtypes = [ctypes.c_void_p]
        L.geosrad_kernel_name.restype = ctypes.c_char_p
        L.geosrad_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]
        _lib = L
    return _lib
