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
    "geosrad_create", "geosrad_create_multi", "geosrad_pick_device", "geosrad_destroy", "geosrad_last_error", "geosrad_real_kind", "geosrad_set_chunk",
    "geosrad_workspace_bytes", "geosrad_set_tables_lw", "geosrad_load_tables_lw", "geosrad_set_inhomogeneity",
    "geosrad_load_inhomogeneity", "geosrad_set_corr_lengths", "geosrad_rrtmg_lw", "geosrad_rrtmg_lw_dev",
    "geosrad_check", "geosrad_profile", "geosrad_profile_read", "geosrad_kernel_name", "geosrad_kernel_label", "geosrad_rrtmg_lw_taumol", "geosrad_mcica", "geosrad_clearcounts",
    "geosrad_set_tables_sw", "geosrad_load_tables_sw", "geosrad_rrtmg_sw", "geosrad_rrtmg_sw_dev", "geosrad_rrtmg_sw_taumol", "geosrad_mcica_dev",
    "geosrad_set_tables_chou_lw", "geosrad_load_tables_chou_lw", "geosrad_irrad", "geosrad_irrad_dev",
    "geosrad_lw_driver_rrtmg_dev", "geosrad_sw_driver_rrtmg_dev", "geosrad_lw_update_flx_dev", "geosrad_sw_update_export_dev",
    "geosrad_rad_tendencies_dev", "geosrad_lw_chou_post_dev", "geosrad_sw_driver_chou_dev", "geosrad_rrtmg_lw_rats_dev", "geosrad_lw_driver_rrtmg_rats_dev", "geosrad_lw_update_rats_dev", "geosrad_lw_update_bands_dev", "geosrad_sw_update_surface_dev",
    "geosrad_set_tables_chou_sw", "geosrad_load_tables_chou_sw", "geosrad_sorad", "geosrad_sorad_dev",
    "geosrad_read_table", "geosrad_rrtmg_sw_cldprmc", "geosrad_lit_index_dev", "geosrad_lit_pack_dev", "geosrad_lit_unpack_dev", "geosrad_dbg_fast64",
]

_lib = None


def build(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Objects (compiled in parallel, each rebuilt only when one of
    the files it includes is newer): geosrad.hip three times - fp32 kernels, fp64 kernels, the extern "C" layer - and lw_cols.hip (the
    on-chip RRTMG_LW band sweeps) and sw_reform.hip (the default RRTMG_SW band sweeps) twice each - fp32, fp64."""
    inc = os.path.join(os.path.dirname(HERE), "include", "geosrad.h")
    hpp = {f: os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")}
    cols_only = {"lw_cols_kernels.hpp", "sw_reform_kernels.hpp", "lw_split_kernels.hpp"}      # headers geosrad.hip does not include
    deps_main = [os.path.join(CSRC, "geosrad.hip"), inc] + [p for f, p in hpp.items() if f not in cols_only]
    deps_cols = [os.path.join(CSRC, "lw_cols.hip")] + [hpp[f] for f in ("lw_cols_kernels.hpp", "lw_cols.hpp", "lw_kernels.hpp", "lw_device.hpp")]
    deps_split = [os.path.join(CSRC, "lw_split.hip")] + [hpp[f] for f in ("lw_split_kernels.hpp", "lw_split.hpp", "lw_kernels.hpp", "lw_device.hpp")]
    deps_swq = [os.path.join(CSRC, "sw_reform.hip")] + [hpp[f] for f in ("sw_reform_kernels.hpp", "sw_reform.hpp", "sw_kernels.hpp", "lw_kernels.hpp", "lw_device.hpp")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(os.path.dirname(HERE), "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    # -fno-slp-vectorize: the SLP vectorizer pairs the fp32 arithmetic of neighbouring g-points into v_pk_* instructions, which issue at half
    # the rate of the plain ones on gfx950 (profiles/tools/ubench/valu_rate.hip: v_pk_fma_f32 4.9 against v_fma_f32 2.4 cycles) and need their
    # operands moved into register pairs: the SW sweeps lose a third of their instructions and 30 VGPRs without it
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]
    units = [("geosrad.hip", part, deps_main) for part in (4, 8, 0)] + [("lw_cols.hip", part, deps_cols) for part in (4, 8)] \
        + [("sw_reform.hip", part, deps_swq) for part in (4, 8)] + [("lw_split.hip", part, deps_split) for part in (4, 8)]
    extra = os.environ.get("GEOSRAD_HIPCC_FLAGS", "").split()            # kernel experiments (A/B builds) only
    jobs, objs = [], []
    for src, part, deps in units:
        obj = os.path.join(objdir, f"{src[:-4]}_part{part}.o")
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in deps):
            jobs.append(subprocess.Popen([hipcc, *flags, *extra, f"-DGEOSRAD_PART={part}", "-c", os.path.join(CSRC, src), "-o", obj]))
    for pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, pr.args)
    if jobs or not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(o) for o in objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", SO])
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
        L.geosrad_kernel_label.restype = ctypes.c_char_p
        L.geosrad_kernel_label.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.geosrad_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]
        L.geosrad_create_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int]
        L.geosrad_pick_device.argtypes = [ctypes.c_int]
        _lib = L
    return _lib
