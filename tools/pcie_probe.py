#!/usr/bin/env python3
"""Host<->device copy rates through the library's own upload/download entry points, page-locked or not, at the sizes the
resident overlay moves every step."""
import ctypes, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from uvic29_amd.tracer import TracerModel
from uvic29_amd.capi import FIELD
m = TracerModel(102, 102, 19, 30, 28, 0)
lib, h = m.lib, m.h
for pinned in (False, True):
    a = np.zeros((102, 19, 102), order="F")
    t = np.zeros((102, 19, 102, 30), order="F")
    if pinned:
        lib.uvic_gpu_pin_host(h, a.ctypes.data_as(ctypes.c_void_p), a.nbytes)
        lib.uvic_gpu_pin_host(h, t.ctypes.data_as(ctypes.c_void_p), t.nbytes)
    for name, arr, f, cnt in (("adv_vet 1.58 MB", a, FIELD["adv_vet"], a.size), ("t x2 3.2 MB", t, FIELD["t_tau"], 2 * a.size)):
        for up in (True, False):
            fn = lib.uvic_gpu_upload if up else lib.uvic_gpu_download
            fn(h, f, arr.ctypes.data_as(ctypes.c_void_p), 0, cnt)
            t0 = time.perf_counter()
            for _ in range(50):
                fn(h, f, arr.ctypes.data_as(ctypes.c_void_p), 0, cnt)
            dt = (time.perf_counter() - t0) / 50
            print(f"{'pinned' if pinned else 'pageable'} {name} {'H2D' if up else 'D2H'}: {dt*1e3:.3f} ms = {cnt*8/dt/1e9:.1f} GB/s")
