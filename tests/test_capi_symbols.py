"""The C-ABI library loads on a GPU-less host and exports every symbol that
include/uvic_gpu.h declares (no compute calls here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    lib_path = ROOT / "uvic2.9_amd" / "csrc" / "libuvic_gpu.so"
    if not lib_path.exists():
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(str(lib_path))
    header = (ROOT / "include" / "uvic_gpu.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(uvic_gpu_\w+)\s*\(", header))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/uvic_gpu.h but not exported"
    assert lib.uvic_gpu_abi_version() == 9


def test_field_table_matches_header():
    from uvic29_amd import capi
    assert capi.FIELD["dxt"] == 0 and "diff_cbt" in capi.FIELD and "t_taup1" in capi.FIELD


def test_product_has_no_cpu_path(monkeypatch):
    from uvic29_amd import capi
    monkeypatch.setattr(capi, "LIBPATH", Path("/nonexistent/libuvic_gpu.so"))
    monkeypatch.setattr(capi, "_lib", None)
    with pytest.raises(capi.UvicGpuError):
        capi.load()
