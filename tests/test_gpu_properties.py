"""Size-independent properties of the production (column-kernel) tracer step at BASELINE's full sizes,
where the scalar oracle is too slow to be the only check: conservation of tracer content, land and
cyclic-image handling, independence of the passive tracers from one another."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

pytestmark = pytest.mark.gpu


def _step(imt, jmt, km, nt, perm=None, src=None, zero_flux=False):
    from uvic29_amd import performance_set, synthetic
    from uvic29_amd.tracer import TracerModel
    cfg = performance_set(nt)
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    if perm is not None:                         # reorder the passive tracers
        for name in ("t_taum1", "t_tau", "stf", "btf"):
            setattr(oc, name, np.asfortranarray(getattr(oc, name)[..., perm]))
    if zero_flux:     # closed system: no surface/bottom fluxes, no resolved flow (the synthetic u has w != 0 at the sea floor)
        oc.stf = np.zeros_like(oc.stf)
        oc.btf = np.zeros_like(oc.btf)
        oc.adv_vet, oc.adv_vnt, oc.adv_vbt = np.zeros_like(oc.adv_vet), np.zeros_like(oc.adv_vnt), np.zeros_like(oc.adv_vbt)
    to, so, c = synthetic.load_eos(km)
    m = TracerModel(imt, jmt, km, nt, cfg.nsrc, 0, device=0)
    m.load_ocean(oc, to, so, c, src=src)
    m.isopyc(); m.tracer()
    out = m.download("t_taup1")
    m.close()
    return oc, out


@pytest.mark.parametrize("dims,nt", [((102, 102, 19), 8), ((202, 202, 32), 4)])
def test_tracer_content_is_conserved(dims, nt):
    """Isopycnal and vertical diffusion, the flux-corrected advection by the Gent-McWilliams velocities (which
    vanish on every boundary) and convective mixing move tracer around but create none: with zero surface and
    bottom fluxes and no resolved flow, sum(t dV) over the ocean is unchanged to the rounding of the sum."""
    imt, jmt, km = dims
    oc, tp = _step(imt, jmt, km, nt, src=np.zeros((imt, km, jmt, nt - 2), order="F"), zero_flux=True)
    g, topo = oc.grid, oc.topo
    vol = (g.dxt[:, None, None] * g.dzt[None, :, None] * (g.cst * g.dyt)[None, None, :] * topo.tmask)[1:-1, :, 1:-1]
    for n in range(nt):
        before = (oc.t_taum1[1:-1, :, 1:-1, n] * vol).sum()
        after = (tp[1:-1, :, 1:-1, n] * vol).sum()
        scale = (np.abs(oc.t_taum1[1:-1, :, 1:-1, n]) * vol).sum()
        assert abs(after - before) <= 2e-12 * scale, (n, after - before, scale)


def test_land_stays_zero_and_cyclic_images_are_copies():
    oc, tp = _step(102, 102, 19, 8)
    land = oc.topo.tmask[:, :, 1:-1] == 0
    assert np.all(tp[:, :, 1:-1][land] == 0.0)
    assert np.array_equal(tp[0, :, 1:-1], tp[100, :, 1:-1]) and np.array_equal(tp[101, :, 1:-1], tp[1, :, 1:-1])
    assert np.isfinite(tp).all()


def test_passive_tracers_do_not_see_each_other():
    """Transport of tracer n depends on T,S (through the mixing tensor and convection) and on tracer n only:
    permuting the passive tracers permutes the result, bit for bit."""
    nt = 8
    perm = np.array([0, 1, 7, 5, 3, 6, 4, 2])
    _, a = _step(102, 102, 19, nt)
    _, b = _step(102, 102, 19, nt, perm=perm)
    assert np.array_equal(b[:, :, 1:-1], a[:, :, 1:-1][..., perm])


def test_level_download_and_halo_staging_move_the_right_rows():
    """uvic_gpu_download_level (one tracer, and every tracer at once) and the halo staging of a latitude slab
    (pack rows js..js+1 / je-1..je, unpack into js-2..js-1 / je+1..je+2) against plain array indexing."""
    import ctypes
    from uvic29_amd import synthetic
    from uvic29_amd.capi import check
    from uvic29_amd.tracer import TracerModel
    dims = (14, 14, 6)
    oc = synthetic.make_ocean("c30", *dims)
    to, so, c = synthetic.load_eos(dims[2])
    m = TracerModel(*dims, oc.cfg.nt, oc.cfg.nsrc, 0)
    m.load_ocean(oc, to, so, c)
    rng = np.random.default_rng(7)
    t = np.asfortranarray(rng.standard_normal((dims[0], dims[2], dims[1], oc.cfg.nt)))
    m.upload("t_taup1", t)
    from uvic29_amd.capi import FIELD
    one = np.zeros((dims[0], dims[1]), order="F")
    check(m.lib.uvic_gpu_download_level(m.h, FIELD["t_taup1"], 5, 3, one.ctypes.data_as(ctypes.c_void_p)), "download_level")
    assert np.array_equal(one, t[:, 2, :, 4])
    every = np.zeros((dims[0], dims[1], oc.cfg.nt), order="F")
    check(m.lib.uvic_gpu_download_level(m.h, FIELD["t_taup1"], 0, 1, every.ctypes.data_as(ctypes.c_void_p)), "download_level")
    assert np.array_equal(every, t[:, 0, :, :])
    # halo staging: what is packed as "send south" is unpacked as "received from the north" of the slab below, etc.
    m.set_shard(js=5, je=9)
    n = int(m.lib.uvic_gpu_halo_elems(m.h))
    assert n == 2 * dims[0] * dims[2] * oc.cfg.nt
    check(m.lib.uvic_gpu_halo_pack(m.h, 1, 1), "halo_pack")
    m.sync()
    hip = ctypes.CDLL("libamdhip64.so")        # plain copies of the staging buffers (device pointers from the ABI)
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    ptr = [m.lib.uvic_gpu_halo_buffer(m.h, q) for q in range(4)]

    def fetch(q):
        a = np.zeros(n)
        assert hip.hipMemcpy(a.ctypes.data_as(ctypes.c_void_p), ptr[q], n * 8, 2) == 0      # device to host
        return a.reshape((oc.cfg.nt, 2, dims[0] * dims[2]))
    send_s, send_n = fetch(0), fetch(1)
    rows = lambda j: t[:, :, j - 1, :].reshape((dims[0] * dims[2], oc.cfg.nt), order="F").T     # row j (1-based) as (nt, imt*km)
    assert np.array_equal(send_s[:, 0], rows(5)) and np.array_equal(send_s[:, 1], rows(6))
    assert np.array_equal(send_n[:, 0], rows(8)) and np.array_equal(send_n[:, 1], rows(9))
    assert hip.hipMemcpy(ptr[2], ptr[1], n * 8, 3) == 0 and hip.hipMemcpy(ptr[3], ptr[0], n * 8, 3) == 0   # pretend the neighbours sent these
    check(m.lib.uvic_gpu_halo_unpack(m.h, 1, 1), "halo_unpack")
    m.sync()
    got = m.download("t_taup1")
    want = t.copy()
    want[:, :, 2:4] = t[:, :, 7:9]        # rows 3,4 <- what was packed from rows 8,9
    want[:, :, 9:11] = t[:, :, 4:6]       # rows 10,11 <- rows 5,6
    assert np.array_equal(got, want)
    m.close()
