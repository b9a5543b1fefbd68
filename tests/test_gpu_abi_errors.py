"""Error behaviour of the GPU-side ABI calls (status codes instead of the reference's
unwrap()/panic)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_render_before_setup_and_bad_arguments(rwr, suzanne):
    with rwr.Context(0) as ctx:
        cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera())
        with pytest.raises(rwr.RwrError) as ei:
            ctx.render(cam_inv)
        assert ei.value.code == rwr.ERR_NOT_READY                      # no resize yet
        ctx.resize(64, 64)
        with pytest.raises(rwr.RwrError) as ei:
            ctx.render(cam_inv)
        assert ei.value.code == rwr.ERR_NOT_READY                      # no scene yet
        with pytest.raises(rwr.RwrError) as ei:
            ctx.resize(0, 10)
        assert ei.value.code == rwr.ERR_INVALID_ARGUMENT
        bad = dict(suzanne, faces=suzanne["faces"].copy())
        bad["faces"]["indices"][5, 1] = 100000                          # the shader would read out of bounds
        with pytest.raises(rwr.RwrError) as ei:
            ctx.upload_model(bad)
        assert ei.value.code == rwr.ERR_INVALID_ARGUMENT and "out of range" in ei.value.message
        ctx.upload_model(suzanne)
        ctx.render(cam_inv)
        with pytest.raises(rwr.RwrError) as ei:                         # aux planes were not requested
            ctx.readback(aux=True)
        assert ei.value.code == rwr.ERR_NOT_READY
        with pytest.raises(rwr.RwrError) as ei:
            ctx.render(cam_inv, rows=(10, 200))
        assert ei.value.code == rwr.ERR_INVALID_ARGUMENT
        for params, code in ((rwr.make_params(spp=0), rwr.ERR_INVALID_ARGUMENT), (rwr.make_params(max_bounces=2), rwr.ERR_UNSUPPORTED),
                             (rwr.make_params(spp=2, flags=rwr.FLAG_USE_BVH), rwr.ERR_UNSUPPORTED)):
            with pytest.raises(rwr.RwrError) as ei:
                ctx.render(cam_inv, params)
            assert ei.value.code == code
        with pytest.raises(rwr.RwrError):
            ctx.set_spheres(rwr.make_spheres([((0, 0, -3), 0.5)] * 9))  # > RWR_MAX_SPHERES
        ctx.render(cam_inv)                                             # still usable after the errors
        assert ctx.readback()["color"].any()
    with pytest.raises(rwr.RwrError) as ei:
        rwr.Context(99)
    assert ei.value.code == rwr.ERR_INVALID_ARGUMENT


def test_degenerate_camera_uniform_renders_without_fault(rwr, orc, gpu_ctx, suzanne):
    """A singular / non-finite camera uniform must not fault the GPU: culling switches itself off
    and the result still equals the oracle's (NaN comparisons included)."""
    w, h = 64, 48
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h))
    flat = cam_inv.copy()
    flat["proj_inv"][0][0][:] = 0.0                                     # x no longer reaches the ray: every column sees the same rays
    gpu_ctx.upload_model(suzanne); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(rwr.make_spheres()); gpu_ctx.resize(w, h)
    gpu_ctx.render(flat, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
    got = gpu_ctx.readback(aux=True)
    want = orc.render_frame(flat.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), suzanne)
    assert np.array_equal(got["obj_id"], want["obj_id"]) and np.array_equal(got["depth"].view(np.uint32), want["depth"].view(np.uint32))
    nan = cam_inv.copy()
    nan["viewmodel_inv"][0][1][1] = np.nan
    gpu_ctx.render(nan, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
    got = gpu_ctx.readback(aux=True)
    want = orc.render_frame(nan.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), suzanne)
    assert np.array_equal(got["obj_id"], want["obj_id"])
