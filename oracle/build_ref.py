#!/usr/bin/env python3
"""Build oracle/_ref/libuvicref_<cfg>_<imt>x<jmt>x<km>.so from the reference
sources WHERE THEY LIE under /root/reference.

TEST INFRASTRUCTURE ONLY.  Nothing produced here is shipped or imported by the
product (uvic2.9_amd/); only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may load the resulting library.

Recipe (which mirrors what the reference's own
`mk` driver does, /root/reference/mk:2326-2400):

  1. cpp -traditional -P every header of the source directories in overlay
     order (later directory wins, mk:1289-1321) into a scratch inc/ directory;
  2. cpp -traditional -P the few .F files of the tracer path;
  3. in the *scratch copies only*: substitute the grid-size parameter line of
     size.h (imt, jmt, km are compile-time parameters,
     updates/09/source/common/size.h:27) and the one declaration that flang
     rejects (`mi` is REAL but used as a subscript, updates/09/source/mom/tracer.F:35,539);
  4. flang -fdefault-real-8 (== ifort -r8, run/mk.ver:51) -O2 -ffp-contract=off
     -fno-automatic (static locals, the ifort default for arrays: tracer.F:121 `src`
     must be zero on land columns);
  5. link everything plus oracle/ref/harness.c and the generated registration
     file into one shared object.  Symbols that only the un-buildable parts of
     the model would resolve (netCDF wrappers, the other component models)
     stay undefined; the library is loaded with lazy binding and those entry
     points are never called.  No stand-in is written for them.

Scratch files are deleted after the link: only the .so stays in oracle/_ref/.
"""
from __future__ import annotations

import argparse
import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path(os.environ.get("UVIC_REFERENCE", "/root/reference"))
OUT = HERE / "_ref"
FLANG = "/opt/rocm/lib/llvm/bin/flang"

BASE = ("O_mom O_fct O_isopycmix O_gent_mcwilliams O_cyclic O_consthmix O_constvmix "
        "O_fullconvect O_embm O_ice O_fourfil O_time_averages O_tidal_kv "
        "O_save_convection").split()

# option sets; every one of them was compile-verified
CONFIGS = {
    # BASELINE config 1: physics only, nt=2
    "p2": BASE,
    # option set E, nt=13
    "e13": BASE + "O_mobi O_mobi_o2 O_mobi_iron O_carbon O_mobi_caco3".split(),
    # option set F = E + alkalinity + nitrogen, nt=18
    "f18": BASE + "O_mobi O_mobi_o2 O_mobi_iron O_carbon O_mobi_caco3 O_mobi_alk O_mobi_nitrogen".split(),
    # option set C == BASELINE config 4, nt=30
    "c30": BASE + ("O_mobi O_mobi_o2 O_mobi_iron O_carbon O_mobi_alk O_mobi_nitrogen "
                   "O_carbon_13 O_carbon_14 O_mobi_nitrogen_15").split(),
    # shipped run/mk.in set, nt=37
    "s37": BASE + ("O_mobi O_carbon O_mobi_alk O_mobi_o2 O_mobi_nitrogen O_mobi_caco3 "
                   "O_mobi_iron O_mobi_silicon O_mobi_nitrogen_15 O_carbon_13 "
                   "O_carbon_14").split(),
    # momentum row: physics only, nt=2, with the three options of run/mk.in that shape
    # `clinic` (explicit Coriolis term, 3-D viscosity coefficients, surface velocities handed to the ice model)
    "m2": BASE + "O_stream_function O_anisotropic_viscosity O_ice_evp".split(),
    # the same without the two options that are additions of updates/09 and of the ice model: one viscosity per row
    # (hmixc.h: amc_north(jmt), visc_ceu scalar), no isbcu/asbcu -- the other branch of the clinic overlay
    "m2i": BASE + ["O_stream_function"],
}

# option set C with the momentum routines and the time-step monitor of run/mk.in (tsiperts steps: tbar, travar, dtabs,
# dc14bar of diagt1 and ektot of clinic, which the overlays form on the device)
CONFIGS["t30"] = CONFIGS["c30"] + "O_stream_function O_anisotropic_viscosity O_ice_evp O_time_step_monitor".split()

# the reference's second boundary: -DO_TMM turns `tracer` into the column-batch source operator of the
# Transport-Matrix-Method driver: imt = batch size, jmt = 1 (u09/common/size.h:26-30)
CONFIGS["tmm30"] = CONFIGS["c30"] + ["O_TMM"]

# ... and compiles without the transport routines (their row arithmetic has no meaning with jmt = 1)
ONLY_SOURCES = {
    "tmm30": ["updates/09/source/mom/tracer.F", "updates/09/source/mom/mobi.F", "updates/09/source/common/co2calc.F",
              "source/common/util.F", "updates/09/source/common/iomngr.F", "updates/09/source/common/file_names.F",
              "updates/09/source/common/UVic_ESCM.F", "source/mom/denscoef.F"],
}

# sources compiled for some configurations only
EXTRA_SOURCES = {
    "m2": ["updates/09/source/mom/clinic.F", "source/common/filuv.F", "updates/09/source/mom/setvbc.F",
           "updates/09/source/mom/loadmw.F"],
    "m2i": ["updates/09/source/mom/clinic.F", "source/common/filuv.F", "updates/09/source/mom/setvbc.F",
            "updates/09/source/mom/loadmw.F"],
    "t30": ["updates/09/source/mom/clinic.F", "source/common/filuv.F", "updates/09/source/mom/setvbc.F",
            "updates/09/source/mom/loadmw.F"],
}

HDR_DIRS = ["source/common", "source/mom", "source/embm", "source/ice",
            "updates/09/source/common", "updates/09/source/mom", "updates/09/source/embm"]

SOURCES = [
    "updates/09/source/mom/tracer.F", "updates/09/source/mom/tracer_adv_flx.F",
    "updates/09/source/mom/isopyc.F", "source/mom/invtri.F",
    "updates/09/source/mom/mobi.F", "updates/09/source/common/co2calc.F",
    "source/mom/convect.F", "updates/09/source/mom/set_sbc.F",
    "source/common/util.F", "source/common/filt.F", "source/common/filtr.F", "source/common/findex.F",
    "updates/09/source/common/iomngr.F", "updates/09/source/common/file_names.F",
    "updates/09/source/common/UVic_ESCM.F",
    "source/mom/state.F", "source/mom/denscoef.F", "source/mom/adv_vel.F",
    "updates/09/source/mom/vmixc.F", "updates/09/source/mom/hmixc.F",
]

SIZE_RE = re.compile(r"parameter \(imt=\s*102, jmt=\s*102, km=\s*19\)")
TMM_SIZE_RE = re.compile(r"parameter \(imt=\s*2000, jmt=\s*1, km=\s*19\)")


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, encoding="latin-1", **kw)
    return r


def lib_name(cfg: str, imt: int, jmt: int, km: int) -> Path:
    return OUT / f"libuvicref_{cfg}_{imt}x{jmt}x{km}.so"


SHIM_DIR = HERE.parent / "uvic2.9_amd" / "fortran"
GPU_LIB_DIR = HERE.parent / "uvic2.9_amd" / "csrc"


def shim_lib_name(cfg: str, imt: int, jmt: int, km: int) -> Path:
    return OUT / f"libuvicshim_{cfg}_{imt}x{jmt}x{km}.so"


def build(cfg: str, imt: int, jmt: int, km: int, keep: bool = False, verbose: bool = False, shim: bool = False) -> Path:
    if not REF.exists():
        raise SystemExit(f"reference not present at {REF}; oracle/_ref can only be built in the build container")
    target = shim_lib_name(cfg, imt, jmt, km) if shim else lib_name(cfg, imt, jmt, km)
    work = OUT / "build" / target.stem
    if work.exists():
        shutil.rmtree(work)
    (work / "inc").mkdir(parents=True)
    defs = [f"-D{o}" for o in CONFIGS[cfg]]
    size_line = f"parameter (imt={imt}, jmt={jmt}, km={km})"

    def patch(text: str) -> str:
        return TMM_SIZE_RE.sub(size_line, SIZE_RE.sub(size_line, text))

    # 1. headers, overlay order
    for d in HDR_DIRS:
        for h in sorted((REF / d).glob("*.h")):
            r = run(["cpp", "-traditional", "-P", *defs, str(h)])
            (work / "inc" / h.name).write_text(patch(r.stdout), encoding="latin-1")
    # 2. sources
    incs = []
    for d in reversed(HDR_DIRS):
        incs += ["-I", str(REF / d)]
    objs = []
    for s in ONLY_SOURCES.get(cfg, SOURCES + EXTRA_SOURCES.get(cfg, [])):
        src = REF / s
        r = run(["cpp", "-traditional", "-P", *incs, *defs, str(src)])
        text = patch(r.stdout)
        if src.name == "tracer.F":
            # `mi` (month index) is declared REAL but used as an array subscript
            text = text.replace("bctz, mi, yrtime", "bctz, yrtime")
            text = text.replace("parameter (fe_n = 14)", "parameter (fe_n = 14)\n      integer mi")
        (work / (src.stem + ".f")).write_text(text, encoding="latin-1")
    if shim:
        # the drop-in test: the package's Fortran overlay (uvic2.9_amd/fortran) provides `tracer`;
        # it is preprocessed like any model source (it #includes the reference's headers)
        # (mixing_gpu.F: `isopyc` and `vmixc` as calls the device makes unnecessary under UVIC_RESIDENT=3)
        shim_files = ["tracer_gpu.F", "mixing_gpu.F"] + (["clinic_gpu.F"] if any(x.endswith("/clinic.F") for x in EXTRA_SOURCES.get(cfg, [])) else [])
        for f in shim_files:
            r = run(["cpp", "-traditional", "-P", *incs, *defs, str(SHIM_DIR / f)])
            (work / (Path(f).stem + ".f")).write_text(patch(r.stdout), encoding="latin-1")
    # 3. generated registration harness (includes only; no reference text)
    sys.path.insert(0, str(HERE / "ref"))
    import gen_harness
    gen_harness.emit(work / "inc", work / "orc_reg.f")
    # 4. compile
    fflags = ["-fdefault-real-8", "-ffixed-form", "-ffixed-line-length-132", "-O2",
              "-ffp-contract=off", "-fno-automatic", "-fPIC", f"-I{work / 'inc'}"]
    if shim:
        r = run([FLANG, "-fdefault-real-8", "-O2", "-fPIC", f"-J{work}", "-c", str(SHIM_DIR / "uvic_gpu_mod.F90"),
                 "-o", str(work / "uvic_gpu_mod.o")])
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-4000:])
            raise SystemExit("flang failed on uvic_gpu_mod.F90")
        objs.append(str(work / "uvic_gpu_mod.o"))
        fflags.append(f"-I{work}")
    for f in sorted(work.glob("*.f")):
        o = f.with_suffix(".o")
        r = run([FLANG, *fflags, "-c", str(f), "-o", str(o)])
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-4000:])
            raise SystemExit(f"flang failed on {f.name}")
        if verbose:
            print("  compiled", f.name)
        objs.append(str(o))
        if shim and f.name in ("tracer.f", "clinic.f", "isopyc.f", "vmixc.f", "adv_vel.f", "state.f"):
            # keep the reference routine reachable as `tracer_cpu` / `clinic_cpu` / ... (diagnostic time steps)
            nm = f.stem
            rr = run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--redefine-sym", f"{nm}_={nm}_cpu_", str(o)])
            if rr.returncode != 0:
                raise SystemExit(rr.stderr)
    r = run(["gcc", "-O2", "-fPIC", "-c", str(HERE / "ref" / "harness.c"), "-o", str(work / "harness.o")])
    if r.returncode != 0:
        raise SystemExit(r.stderr)
    objs.append(str(work / "harness.o"))
    # 5. link
    extra = []
    if shim:
        extra = [f"-L{GPU_LIB_DIR}", "-luvic_gpu", "-Wl,-rpath,$ORIGIN/../../uvic2.9_amd/csrc"]
    r = run([FLANG, "-shared", "-o", str(target), *objs, "-Wl,-z,lazy", *extra])
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        raise SystemExit("link failed")
    if not keep:
        shutil.rmtree(work)
        try:
            (OUT / "build").rmdir()
        except OSError:
            pass
    return target


DEFAULT_BUILDS = [
    ("p2", 14, 14, 6), ("c30", 14, 14, 6),
    ("p2", 102, 102, 19), ("c30", 102, 102, 19),
    ("f18", 14, 14, 6), ("s37", 14, 14, 6),       # MOBI option sets F and run/mk.in's (tests/test_mobi_sets.py)
    ("m2", 14, 14, 6), ("m2", 102, 102, 19),      # momentum step: clinic, filuv, setvbc (tests/test_clinic.py)
    ("m2i", 14, 14, 6),                           # clinic with one viscosity per row, without O_ice_evp
    ("tmm30", 64, 1, 6),                          # `tracer` as the O_TMM column-batch operator (tests/test_tmm.py)
    ("t30", 14, 14, 6),                           # option set C + clinic + O_time_step_monitor (tsiperts steps on the device)
]


SHIM_BUILDS = [("p2", 14, 14, 6), ("c30", 14, 14, 6), ("c30", 102, 102, 19), ("f18", 14, 14, 6), ("s37", 14, 14, 6),
               ("m2", 14, 14, 6), ("m2", 102, 102, 19), ("m2i", 14, 14, 6), ("t30", 14, 14, 6),
               ("t30", 102, 102, 19)]     # bench.py: both overlays with the shipped switches (overlay_baseline(..., ocean_loop=True))


def build_default(force: bool = False, verbose: bool = False, jobs: int = 4):
    """Every library the tests use, `jobs` builds at a time (each is a chain of cpp/flang processes in its own scratch
    directory; about five minutes one after the other)."""
    from concurrent.futures import ThreadPoolExecutor
    todo, built = [], []
    if (GPU_LIB_DIR / "libuvic_gpu.so").exists():
        for cfg, imt, jmt, km in SHIM_BUILDS:
            t = shim_lib_name(cfg, imt, jmt, km)
            srcs = [SHIM_DIR / "tracer_gpu.F", SHIM_DIR / "clinic_gpu.F", SHIM_DIR / "mixing_gpu.F", SHIM_DIR / "uvic_gpu_mod.F90"]
            built.append(t)
            if t.exists() and not force and all(t.stat().st_mtime >= s_.stat().st_mtime for s_ in srcs):
                continue
            todo.append((cfg, imt, jmt, km, True))
    for cfg, imt, jmt, km in DEFAULT_BUILDS:
        t = lib_name(cfg, imt, jmt, km)
        built.append(t)
        if t.exists() and not force:
            continue
        todo.append((cfg, imt, jmt, km, False))
    sys.path.insert(0, str(HERE / "ref"))
    import gen_harness  # noqa: F401  (imported once, before the worker threads use it)

    def one(job):
        cfg, imt, jmt, km, shim = job
        if verbose:
            print("building", (shim_lib_name if shim else lib_name)(cfg, imt, jmt, km).name, flush=True)
        return build(cfg, imt, jmt, km, shim=shim)

    todo.sort(key=lambda j: -j[1] * j[2] * j[3])     # the large grids first
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(one, todo))
    return built


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default=None, choices=list(CONFIGS))
    ap.add_argument("--imt", type=int, default=102)
    ap.add_argument("--jmt", type=int, default=102)
    ap.add_argument("--km", type=int, default=19)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--shim", action="store_true", help="link the package's Fortran overlay in place of `tracer`")
    a = ap.parse_args()
    if a.cfg is None:
        for t in build_default(force=a.force, verbose=True):
            print(t)
    else:
        print(build(a.cfg, a.imt, a.jmt, a.km, keep=a.keep, verbose=True, shim=a.shim))
