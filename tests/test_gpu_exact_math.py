"""The frame kernel's short exact forms (rwr_device.h: to_non_linear_depth_fast, normalize3_fast) must
return the very bits of the shader expressions they replace (compute.wgsl:78-80, :162).  The library
checks them on the GPU itself: every float of the depth form's domain, and 2^30 pseudo-random vectors
over the whole domain of the normalize form."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_short_exact_forms_match_the_ieee_expressions_bit_for_bit(rwr, gpu_ctx):
    depth_n, depth_bad, vec_n, vec_bad = gpu_ctx.selftest_exact_math(normalize_count=1 << 30, seed=7)
    # positive normal floats below 2^126: biased exponents 1..252, 2^23 mantissas each
    assert depth_n == 252 * (1 << 23)
    assert depth_bad == 0
    assert vec_n > 0.7 * (1 << 30)      # most generated vectors are in the domain
    assert vec_bad == 0


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [2.0 ** 45, 2.0 ** -45, 1.0])
def test_ray_directions_outside_the_short_forms_domain_take_the_ieee_path(rwr, orc, gpu_ctx, suzanne, scale):
    """A camera uniform whose un-normalised ray directions leave [2^-40, 2^40] (viewmodel_inv scaled by 2^+-45:
    same rays after normalize) makes every wave fall back to the compiler's divisions; frames must stay
    bit-identical to the oracle either way."""
    w, h = 130, 70
    cam = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.2, 0.1, 0.4), target=(0, 0, -1), aspect=w / h)).copy()
    cam["viewmodel_inv"][0][:3, :3] *= np.float32(scale)
    spheres = rwr.make_spheres(rwr.REFERENCE_SPHERES)
    gpu_ctx.upload_model(suzanne); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(spheres); gpu_ctx.resize(w, h)
    gpu_ctx.render(cam, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
    got = gpu_ctx.readback(aux=True)
    want = orc.render_frame(cam.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), spheres.view(orc.SPHERE_DTYPE), suzanne)
    assert np.array_equal(got["obj_id"], want["obj_id"]) and (got["obj_id"] >= 0).mean() > 0.5
    assert np.array_equal(got["hit_t"].view(np.uint32), want["hit_t"].view(np.uint32))
    assert np.array_equal(got["depth"].view(np.uint32), want["depth"].view(np.uint32))
    assert np.abs(got["color_f32"] - want["color_f32"]).max() <= 1e-4
