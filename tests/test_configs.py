import numpy as np
import pytest

from uvic29_amd import OPTION_SETS
import refmodel


def test_c30_counts():
    c = OPTION_SETS["c30"]
    assert (c.nt, c.nsrc, c.ntnpzd) == (30, 28, 25)
    assert OPTION_SETS["p2"].nt == 2


@pytest.mark.parametrize("cfg", ["p2", "c30"])
def test_index_arrays_bit_exact_vs_reference(cfg):
    if not refmodel.available(cfg, 14, 14, 6):
        pytest.skip("oracle/_ref not built")
    ref = refmodel.RefLib(cfg, 14, 14, 6)
    ref.call("tracer_init")
    c = OPTION_SETS[cfg]
    assert np.array_equal(ref.v["itrc"], np.array(c.itrc(), dtype=np.int32))
    for name in c.tracers:
        key = {"temp": "itemp", "salt": "isalt", "c14": "ic14"}.get(name, "i" + name)
        assert int(ref.get(key)) == c.index(name), name
