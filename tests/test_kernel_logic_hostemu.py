"""Kernel LOGIC on the CPU: the package's kernel sources compiled for the host
with the barrier-phase emulation of tests/hostemu (no GPU in the build
container) must reproduce the oracle bit-for-bit -- including longitude
chunking with cyclic halos and different workgroup sizes."""
import numpy as np
import pytest

from uvic29_amd import synthetic
import emu
import oracle_c


@pytest.mark.parametrize("cfg,dims,nchunk,nth", [("p2", (14, 14, 6), 1, 64), ("p2", (14, 14, 6), 3, 7),
                                                 ("c30", (14, 14, 6), 2, 32), ("p2", (30, 22, 9), 4, 100)])
def test_emulated_kernels_equal_oracle(cfg, dims, nchunk, nth):
    oc = synthetic.make_ocean(cfg, *dims)
    km = dims[2]
    if km in (6, 19, 32):
        to, so, c = synthetic.load_eos(km)
    else:  # any smooth coefficients will do for a logic test
        to0, so0, c0 = synthetic.load_eos(19)
        to, so, c = to0[:km].copy(), so0[:km].copy(), np.asfortranarray(c0[:km].copy())
    src = None
    if oc.cfg.nsrc:
        rng = np.random.default_rng(2029)
        src = np.asfortranarray(rng.standard_normal(dims[:1] + (km, dims[1], oc.cfg.nsrc)) * 1e-9 * oc.topo.tmask[..., None])
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c, src=src)
    em = emu.EmuOcean(oc, to, so, c, src=src)
    orc.isopyc(); orc.add_k33(); em.isopyc()
    for n in ("alphai", "betai", "ddxt", "ddyt", "ddzt", "Ai_ez", "Ai_nz", "Ai_bx", "Ai_by", "K11", "K22", "K33",
              "adv_vetiso", "adv_vbtiso"):
        assert np.array_equal(em.a[n], orc.a[n]), n
    imt, jmt = dims[0], dims[1]
    assert np.array_equal(em.a["adv_vntiso"][:imt - 1], orc.a["adv_vntiso"][:imt - 1])
    tp = orc.transport()
    em.transport(nchunk=nchunk, nthreads=nth)
    before = em.a["t_taup1"].copy(order="F")
    em.convect()
    assert np.array_equal(em.a["t_taup1"][:, :, 1:jmt - 1], tp[:, :, 1:jmt - 1])
    # the two-pass convection used on the GPU (segments from T,S, replay per tracer)
    em.a["t_taup1"][...] = before
    em.convect(twopass=True)
    assert np.array_equal(em.a["t_taup1"][:, :, 1:jmt - 1], tp[:, :, 1:jmt - 1])
    assert (em.a["cv_nseg"] > 0).any() or cfg == "c30"
