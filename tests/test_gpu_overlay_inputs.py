"""uvic_gpu_overlay_inputs (include/uvic_gpu.h): the step's inputs -- velocities, diff_cbt, surface and bottom fluxes -- in
one call, copied beside the main stream into the device copy the previous step does not read, with adv_vbt formed on the
device.  Steps fed this way must equal, bit for bit, steps fed by six plain uploads that send adv_vbt as well."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NSTEP = 5


def _inputs(ocean, step):
    """velocities and fluxes that change from step to step, continuity kept (synthetic.advective_velocities)"""
    from uvic29_amd import synthetic
    g = ocean.grid
    vet, vnt, vbt = synthetic.advective_velocities(g, ocean.u * (1.0 + 0.05 * np.sin(0.7 * step)))
    rng = np.random.default_rng(100 + step)
    cbt = np.asfortranarray(ocean.diff_cbt_bg * (1.0 + 0.2 * rng.random(ocean.diff_cbt_bg.shape)))
    stf = np.asfortranarray(ocean.stf * (1.0 + 0.1 * step) + 1e-9 * rng.standard_normal(ocean.stf.shape) * (ocean.topo.kmt > 0)[..., None])
    btf = np.asfortranarray(1e-10 * rng.standard_normal(ocean.btf.shape) * (ocean.topo.kmt > 0)[..., None])
    return [np.asfortranarray(a) for a in (vet, vnt, vbt, cbt, stf, btf)]


def _run(cfg_name, grid, one_call, streams=2):
    from uvic29_amd import OPTION_SETS, synthetic
    from uvic29_amd.capi import check
    from uvic29_amd.tracer import TimeLoop, TracerModel
    imt, jmt, km = grid
    cfg = OPTION_SETS[cfg_name]
    ocean = synthetic.make_ocean(cfg, imt, jmt, km)
    to, so, c = synthetic.load_eos(km)
    m = TracerModel(imt, jmt, km, cfg.nt, cfg.nsrc, cfg.ntnpzd, device=0)
    m.load_ocean(ocean, to, so, c)
    if cfg.ntnpzd:
        m.set_mobi(ocean)
    m.set_params(diff_cbt_has_k33=1)
    if one_call:
        m.set_option("upload_streams", streams)
        m.set_host_sync(False)
    loop = TimeLoop(m, ocean.params.dtts, nmix=3)
    keep = []
    for step in range(NSTEP):
        vet, vnt, vbt, cbt, stf, btf = _inputs(ocean, step)
        if one_call:
            # the memory window's shapes: adv_vet and diff_cbt without row 1 (jsmw = 2), diff_cbt without row jmt (jemw = jmt-1)
            host = [np.asfortranarray(vet[:, :, 1:]), vnt, np.asfortranarray(cbt[:, :, 1:jmt - 1]), stf, btf]
            keep.append(host)           # the copies are not waited for: the arrays must outlive the step
            p = [a.ctypes.data_as(ctypes.c_void_p) for a in host]
            check(m.lib.uvic_gpu_overlay_inputs(m.h, 2, jmt - 1, p[0], p[1], None, p[2], p[3], p[4]), "overlay_inputs")
        else:
            m.upload("adv_vet", vet); m.upload("adv_vnt", vnt); m.upload("adv_vbt", vbt)
            m.upload_rows("diff_cbt", np.asfortranarray(cbt[:, :, 1:jmt - 1]), 2, jmt - 1)
            m.upload("stf", stf); m.upload("btf", btf)
        loop.step()
    m.sync()
    out = m.download("t_tau").copy()
    m.close()
    return out


@pytest.mark.parametrize("cfg,grid,streams", [("c30", (14, 14, 6), 2), ("c30", (102, 102, 19), 2), ("c30", (14, 14, 6), 1), ("p2", (14, 14, 6), 2)])
def test_inputs_in_one_call_equal_plain_uploads(cfg, grid, streams):
    ref = _run(cfg, grid, False)
    got = _run(cfg, grid, True, streams)
    assert np.isfinite(ref).all()
    assert np.array_equal(got, ref)


def test_inputs_without_velocities_need_velocities_formed_on_the_device():
    """adv_vet/adv_vnt may be left out only after uvic_gpu_overlay_velocities has formed them on the device for the coming
    step: otherwise the call is refused with a message, not computed through with whatever the buffers hold."""
    from uvic29_amd import OPTION_SETS, synthetic
    from uvic29_amd.tracer import TracerModel
    cfg = OPTION_SETS["p2"]
    ocean = synthetic.make_ocean(cfg, 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    m = TracerModel(14, 14, 6, cfg.nt, cfg.nsrc, cfg.ntnpzd, device=0)
    m.load_ocean(ocean, to, so, c)
    cbt = np.asfortranarray(ocean.diff_cbt_bg[:, :, 1:13])
    p = [a.ctypes.data_as(ctypes.c_void_p) for a in (cbt, ocean.stf, ocean.btf)]
    assert m.lib.uvic_gpu_overlay_inputs(m.h, 2, 13, None, None, None, p[0], p[1], p[2]) != 0
    assert "uvic_gpu_overlay_velocities" in m.last_error()
    vet = np.asfortranarray(ocean.adv_vet[:, :, 1:])
    assert m.lib.uvic_gpu_overlay_inputs(m.h, 2, 13, vet.ctypes.data_as(ctypes.c_void_p), None, None, p[0], p[1], p[2]) != 0   # one of the pair
    m.close()
