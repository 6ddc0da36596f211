"""Build libuvic_gpu.so (gfx950) in-tree with hipcc.  No JIT cache: the .so
lives next to the sources so that it travels to the GPU box with the tree."""
from __future__ import annotations

import os
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libuvic_gpu.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def sources():
    return [CSRC / "uvic_gpu.hip"] + sorted(CSRC.glob("*.hpp")) + sorted(CSRC.glob("*.h")) + \
        [CSRC.parent.parent / "include" / "uvic_gpu.h"]


def build(force: bool = False, verbose: bool = False) -> Path:
    srcs = sources()
    if LIB.exists() and not force and all(LIB.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return LIB
    cmd = [HIPCC, *FLAGS, "-o", str(LIB), str(CSRC / "uvic_gpu.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
