"""ctypes driver for oracle/_ref (the reference Fortran compiled by build_ref.py).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never by the product package.

`RefLib(cfg, imt, jmt, km)` loads libuvicref_<cfg>_<imt>x<jmt>x<km>.so with
lazy binding and exposes every COMMON-block variable of the reference as a
numpy view (`ref.v['t']`, Fortran order, the reference's own lower bounds are
available in `ref.lb['t']`).  Reference subroutines are called through
`ref.call('tracer', joff, js, je, is, ie)`; every argument is passed by
reference as the F77 implicit interface requires.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_DT = {(1, 8): np.float64, (1, 4): np.float32, (2, 4): np.int32, (2, 8): np.int64,
       (3, 4): np.int32, (3, 8): np.int64}


def lib_path(cfg: str, imt: int, jmt: int, km: int, shim: bool = False) -> Path:
    """`shim=True`: the same reference build with `tracer` replaced by the package's
    Fortran overlay (uvic2.9_amd/fortran/tracer_gpu.F) linked against libuvic_gpu.so."""
    kind = "uvicshim" if shim else "uvicref"
    return HERE / "_ref" / f"lib{kind}_{cfg}_{imt}x{jmt}x{km}.so"


def available(cfg: str, imt: int, jmt: int, km: int, shim: bool = False) -> bool:
    return lib_path(cfg, imt, jmt, km, shim).exists()


class RefLib:
    def __init__(self, cfg: str, imt: int, jmt: int, km: int, shim: bool = False):
        path = lib_path(cfg, imt, jmt, km, shim)
        if not path.exists():
            raise FileNotFoundError(f"{path} missing: run `python oracle/build_ref.py` in the build container")
        self.cfg, self.imt, self.jmt, self.km = cfg, imt, jmt, km
        self.lib = ctypes.CDLL(str(path), mode=os.RTLD_LAZY | os.RTLD_LOCAL)
        self.lib.orc_reset()
        self.lib.orc_register_all_()
        self.v: dict[str, np.ndarray] = {}
        self.lb: dict[str, tuple] = {}
        self.block: dict[str, str] = {}
        n = self.lib.orc_count()
        name = ctypes.create_string_buffer(64)
        addr = ctypes.c_void_p()
        eb, tc, rank = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        shape = (ctypes.c_int * 7)()
        lb = (ctypes.c_int * 7)()
        for i in range(n):
            self.lib.orc_get(i, name, ctypes.byref(addr), ctypes.byref(eb), ctypes.byref(tc),
                             ctypes.byref(rank), shape, lb)
            full = name.value.decode()
            blk, var = full.split(":")
            dt = _DT[(tc.value, eb.value)]
            shp = tuple(shape[k] for k in range(rank.value)) or (1,)
            cnt = int(np.prod(shp))
            if cnt == 0:
                continue
            buf = (ctypes.c_char * (cnt * eb.value)).from_address(addr.value)
            arr = np.frombuffer(buf, dtype=dt, count=cnt).reshape(shp, order="F")
            if var in self.v:  # same name in two blocks: keep first, expose the other qualified
                self.v[full] = arr
                self.lb[full] = tuple(lb[k] for k in range(rank.value))
                continue
            self.v[var] = arr
            self.lb[var] = tuple(lb[k] for k in range(rank.value))
            self.block[var] = blk

    # scalars -----------------------------------------------------------------
    def get(self, name):
        return self.v[name].reshape(-1)[0]

    def set(self, name, value):
        self.v[name].reshape(-1)[0] = value

    # calling reference procedures -------------------------------------------
    def call(self, name: str, *args):
        """Call reference subroutine `name` (F77: all args by reference)."""
        fn = getattr(self.lib, name.lower() + "_")
        fn.restype = None
        keep, cargs = [], []
        for a in args:
            if isinstance(a, np.ndarray):
                cargs.append(ctypes.c_void_p(a.ctypes.data))
                keep.append(a)
            elif isinstance(a, (bool, np.bool_)):
                c = ctypes.c_int(1 if a else 0)
                keep.append(c)
                cargs.append(ctypes.byref(c))
            elif isinstance(a, (int, np.integer)):
                c = ctypes.c_int(int(a))
                keep.append(c)
                cargs.append(ctypes.byref(c))
            elif isinstance(a, (float, np.floating)):
                c = ctypes.c_double(float(a))
                keep.append(c)
                cargs.append(ctypes.byref(c))
            elif isinstance(a, ctypes._SimpleCData):
                keep.append(a)
                cargs.append(ctypes.byref(a))
            else:
                raise TypeError(type(a))
        fn(*cargs)
        return keep
