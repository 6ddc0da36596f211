"""ctypes binding of the C ABI (include/uvic_gpu.h -> csrc/libuvic_gpu.so).

There is no CPU path: importing works anywhere (so that the symbol table can be
checked on a GPU-less host), but creating a model instance without the HIP
library or without a GPU raises immediately.
"""
from __future__ import annotations

import ctypes
import re
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
HEADER = PKG.parent / "include" / "uvic_gpu.h"
LIBPATH = PKG / "csrc" / "libuvic_gpu.so"


class UvicGpuError(RuntimeError):
    pass


def _parse_fields():
    text = HEADER.read_text()
    body = text[text.index("enum uvic_field {") + len("enum uvic_field {"):]
    body = body[:body.index("};")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for tok in body.split(","):
        tok = tok.strip().split("=")[0].strip()
        if tok:
            names.append(tok)
    assert names[-1] == "UVIC_F_COUNT"
    return {n[len("UVIC_F_"):].lower(): i for i, n in enumerate(names[:-1])}


FIELD = _parse_fields()

EXPORTS = [
    "uvic_gpu_last_error", "uvic_gpu_abi_version", "uvic_gpu_create", "uvic_gpu_destroy", "uvic_gpu_upload",
    "uvic_gpu_download", "uvic_gpu_upload_rows", "uvic_gpu_download_rows", "uvic_gpu_field_elems",
    "uvic_gpu_field_devptr", "uvic_gpu_stream", "uvic_gpu_set_params", "uvic_gpu_set_shard", "uvic_gpu_isopyc",
    "uvic_gpu_transport", "uvic_gpu_convect", "uvic_gpu_tracer", "uvic_gpu_rotate", "uvic_gpu_sync",
    "uvic_gpu_profile", "uvic_gpu_profile_live", "uvic_gpu_profile_read", "uvic_gpu_step_async", "uvic_gpu_step_pre_async", "uvic_gpu_convect_async", "uvic_gpu_set_mobi", "uvic_gpu_set_mobi_opt", "uvic_gpu_mobi_options_flat", "uvic_gpu_set_mobi_flat", "uvic_gpu_mobi", "uvic_gpu_prefetch_sources",
    "uvic_gpu_set_mixing", "uvic_gpu_set_exact", "uvic_gpu_adv_vel", "uvic_gpu_set_vmix_params", "uvic_gpu_vmixc", "uvic_gpu_set_filter", "uvic_gpu_prefetch_isopyc",
    "uvic_gpu_step_lookahead", "uvic_gpu_download_level", "uvic_gpu_set_mobi_step", "uvic_gpu_pin_host", "uvic_gpu_halo_elems", "uvic_gpu_halo_buffer", "uvic_gpu_halo_pack", "uvic_gpu_halo_unpack",
    "uvic_gpu_push_setup", "uvic_gpu_push_export", "uvic_gpu_push_open", "uvic_gpu_push_exchange",
    "uvic_gpu_step_lookahead_at", "uvic_gpu_prefetch_sources_at", "uvic_gpu_set_host_sync", "uvic_gpu_sbc_config",
    "uvic_gpu_sbc_transfer", "uvic_gpu_overlay_step", "uvic_gpu_overlay_inputs", "uvic_gpu_overlay_velocities", "uvic_gpu_overlay_momentum",
    "uvic_gpu_set_clinic_params", "uvic_gpu_state", "uvic_gpu_clinic", "uvic_gpu_set_filter_u",
    "uvic_gpu_state_async", "uvic_gpu_clinic_async",
    "uvic_gpu_tmm_create", "uvic_gpu_tmm_set_mobi", "uvic_gpu_tmm_sources", "uvic_gpu_rotate_u", "uvic_gpu_add_ext_mode", "uvic_gpu_adv_vel_async",
    "uvic_gpu_momentum_async", "uvic_gpu_momentum_wait", "uvic_gpu_unpin_host", "uvic_gpu_set_option", "uvic_gpu_set_tsi", "uvic_gpu_tsi_read", "uvic_gpu_tsi_ektot", "uvic_gpu_set_tavg", "uvic_gpu_tavg_read",
]


class Dims(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("imt", "jmt", "km", "nt", "nsrc", "ntnpzd")]


class ClinicParams(ctypes.Structure):
    """uvic_clinic_params (include/uvic_gpu.h)."""
    _fields_ = [(n, ctypes.c_double) for n in ("c2dtuv", "grav", "rho0r", "kappa_m", "cdbot")]


class Params(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_double) for n in ("c2dtts", "aidif", "diff_cet", "diff_cnt", "slmxr", "ahisop", "athkdf")]
                + [("diff_cbt_has_k33", ctypes.c_int32), ("pad_", ctypes.c_int32)])


class VmixParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("kappa_h", "zetar", "ogamma", "gravrho0r")]


_lib = None


def load():
    """Load libuvic_gpu.so; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    import os
    path = Path(os.environ.get("UVIC_GPU_LIB", LIBPATH))   # development override (kernel variants)
    if not path.exists():
        raise UvicGpuError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # PyTorch ships its own copy of the HIP runtime.  A process that uses both must let torch bring its runtime up
    # first (the library then binds to that one); the other way round torch finds "no HIP GPUs".  So if torch is
    # already imported, initialise its device layer before the library is loaded.
    import sys
    if "torch" in sys.modules:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:   # torch without a usable GPU: the library will say so itself
            pass
    lib = ctypes.CDLL(str(path))
    lib.uvic_gpu_last_error.restype = ctypes.c_char_p
    lib.uvic_gpu_field_elems.restype = ctypes.c_int64
    lib.uvic_gpu_field_elems.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.uvic_gpu_field_devptr.restype = ctypes.c_void_p
    lib.uvic_gpu_field_devptr.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.uvic_gpu_stream.restype = ctypes.c_void_p
    lib.uvic_gpu_stream.argtypes = [ctypes.c_void_p]
    lib.uvic_gpu_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(Dims), ctypes.c_int]
    for fn in ("uvic_gpu_upload", "uvic_gpu_download"):
        getattr(lib, fn).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    for fn in ("uvic_gpu_upload_rows", "uvic_gpu_download_rows"):
        getattr(lib, fn).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.uvic_gpu_set_params.argtypes = [ctypes.c_void_p, ctypes.POINTER(Params)]
    lib.uvic_gpu_set_shard.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4
    for fn in ("uvic_gpu_destroy", "uvic_gpu_isopyc", "uvic_gpu_transport", "uvic_gpu_convect", "uvic_gpu_tracer",
               "uvic_gpu_rotate", "uvic_gpu_sync", "uvic_gpu_step_async", "uvic_gpu_step_pre_async",
               "uvic_gpu_convect_async", "uvic_gpu_mobi"):
        getattr(lib, fn).argtypes = [ctypes.c_void_p]
    lib.uvic_gpu_prefetch_sources.argtypes = [ctypes.c_void_p, ctypes.c_double]
    lib.uvic_gpu_set_mixing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.uvic_gpu_set_exact.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.uvic_gpu_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    lib.uvic_gpu_set_tsi.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.uvic_gpu_tsi_read.argtypes = [ctypes.c_void_p] * 5
    lib.uvic_gpu_tsi_ektot.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    lib.uvic_gpu_set_mobi.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.uvic_gpu_set_mobi_opt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.uvic_gpu_profile.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    lib.uvic_gpu_set_vmix_params.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.uvic_gpu_set_filter.argtypes = [ctypes.c_void_p, ctypes.c_double] + [ctypes.c_int] * 5
    lib.uvic_gpu_set_filter_u.argtypes = [ctypes.c_void_p, ctypes.c_double] + [ctypes.c_int] * 5
    lib.uvic_gpu_set_clinic_params.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.uvic_gpu_tmm_create.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6
    lib.uvic_gpu_tmm_set_mobi.argtypes = [ctypes.c_void_p] * 5
    lib.uvic_gpu_tmm_sources.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double] + [ctypes.c_void_p] * 6
    lib.uvic_gpu_adv_vel_async.argtypes = [ctypes.c_void_p]
    lib.uvic_gpu_momentum_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    lib.uvic_gpu_momentum_wait.argtypes = [ctypes.c_void_p]
    lib.uvic_gpu_unpin_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.uvic_gpu_rotate_u.argtypes = [ctypes.c_void_p]
    lib.uvic_gpu_add_ext_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.uvic_gpu_state.argtypes = [ctypes.c_void_p]
    lib.uvic_gpu_state_async.argtypes = [ctypes.c_void_p]
    lib.uvic_gpu_clinic.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
    lib.uvic_gpu_clinic_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
    lib.uvic_gpu_profile_live.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.uvic_gpu_download_level.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.uvic_gpu_set_mobi_step.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double] + [ctypes.c_void_p] * 4
    lib.uvic_gpu_pin_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    lib.uvic_gpu_step_lookahead.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    lib.uvic_gpu_step_lookahead_at.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                               ctypes.c_double, ctypes.c_double, ctypes.c_int]
    lib.uvic_gpu_prefetch_sources_at.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    lib.uvic_gpu_set_host_sync.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.uvic_gpu_sbc_config.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.uvic_gpu_sbc_transfer.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.uvic_gpu_overlay_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.uvic_gpu_halo_elems.restype = ctypes.c_int64
    lib.uvic_gpu_halo_elems.argtypes = [ctypes.c_void_p]
    lib.uvic_gpu_halo_buffer.restype = ctypes.c_void_p
    lib.uvic_gpu_halo_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.uvic_gpu_halo_pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.uvic_gpu_halo_unpack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.uvic_gpu_overlay_inputs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6
    lib.uvic_gpu_set_tavg.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.uvic_gpu_tavg_read.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
    lib.uvic_gpu_overlay_velocities.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.uvic_gpu_overlay_momentum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double] + [ctypes.c_void_p] * 4
    lib.uvic_gpu_push_setup.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.uvic_gpu_push_export.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.uvic_gpu_push_open.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.uvic_gpu_push_exchange.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.uvic_gpu_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise UvicGpuError(f"{what} failed (status {rc}): {load().uvic_gpu_last_error().decode()}")
