"""Normal-mapped shading (extension; north_star "diffuse/normal-map shading", res/cube.mtl:13 `map_Bump cube-normal.png`,
which the reference itself never loads: /root/reference/src/resources.rs:187-213).  Definition: oracle/rt_oracle.c
normal_mapped().  Gated by RWR_FLAG_NORMAL_MAP: without the flag the frame is the reference's, with it only the COLOUR
of mesh hits changes — ids, distances and depth stay bit-exact against the oracle, colour within 1e-4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
COLOR_TOL = 1e-4


def _gpu(rwr, ctx, model, cam_inv, w, h, params, spheres=None):
    ctx.upload_model(model)
    ctx.set_instances(None)
    ctx.set_spheres(rwr.make_spheres() if spheres is None else spheres)
    ctx.resize(w, h)
    ctx.render(cam_inv, params)
    return ctx.readback(aux=True)


def _same_visibility(a, b):
    for k in ("obj_id", "hit_t", "depth"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k


@pytest.mark.parametrize("extra", [0, "bvh", "one_pixel"])
def test_cube_from_outside_matches_the_oracle(rwr, orc, gpu_ctx, cube, extra):
    assert cube["normal_map"] is not None and cube["normal_map"].shape[2] == 4
    w, h = 192, 144
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(2.2, 1.7, 3.1), target=(0, 0, 0), aspect=w / h))
    flags = rwr.FLAG_AUX_OUTPUTS | {0: 0, "bvh": rwr.FLAG_USE_BVH, "one_pixel": rwr.FLAG_ONE_PIXEL_PER_LANE}[extra]
    got = _gpu(rwr, gpu_ctx, cube, cam_inv, w, h, rwr.make_params(flags=flags | rwr.FLAG_NORMAL_MAP))
    flat = _gpu(rwr, gpu_ctx, cube, cam_inv, w, h, rwr.make_params(flags=flags))
    oc = cam_inv.view(orc.CAMERA_INV_DTYPE)
    want = orc.render_path(oc, orc.make_screen(w, h), orc.make_params(1, 0, flags=orc.FLAG_NORMAL_MAP), orc.make_spheres(), cube)
    want_flat = orc.render_frame(oc, orc.make_screen(w, h), orc.make_spheres(), cube)
    _same_visibility(got, want)
    _same_visibility(flat, want_flat)                                   # without the flag: the reference's frame
    assert np.abs(flat["color_f32"] - want_flat["color_f32"]).max() <= COLOR_TOL
    assert np.abs(got["color_f32"] - want["color_f32"]).max() <= COLOR_TOL
    assert np.abs(got["color"].astype(int) - want["color"].astype(int)).max() <= 1
    mesh = got["obj_id"] >= 0
    assert mesh.mean() > 0.03          # (the reference camera sees wider than its fovy: z = -0.495)
    # the map is not a no-op here, and it only touches mesh pixels.  (cube.mtl has Ka = 1: clamped colours saturate, so the
    # difference is looked for in the unclamped float plane.)
    assert np.abs(got["color_f32"] - flat["color_f32"])[mesh].max() > 0.05
    assert np.array_equal(got["color_f32"][~mesh], flat["color_f32"][~mesh])


def test_path_traced_with_the_map(rwr, orc, gpu_ctx, cube):
    w, h = 96, 72
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(2.2, 1.7, 3.1), target=(0, 0, 0), aspect=w / h))
    sph = rwr.make_spheres([((1.6, 1.2, 1.4), 0.5)])
    got = _gpu(rwr, gpu_ctx, cube, cam_inv, w, h, rwr.make_params(spp=3, max_bounces=1, seed=4, flags=rwr.FLAG_AUX_OUTPUTS | rwr.FLAG_NORMAL_MAP), sph)
    want = orc.render_path(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_params(3, 1, seed=4, flags=orc.FLAG_NORMAL_MAP),
                           sph.view(orc.SPHERE_DTYPE), cube)
    _same_visibility(got, want)
    assert np.abs(got["color_f32"] - want["color_f32"]).max() <= COLOR_TOL


def test_no_map_no_effect_and_removal(rwr, orc, gpu_ctx, suzanne, cube):
    w, h = 128, 72
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0, 0, 3), aspect=w / h))
    assert suzanne["normal_map"] is None                               # suzanne.mtl names no map_Bump
    a = _gpu(rwr, gpu_ctx, suzanne, cam_inv, w, h, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS | rwr.FLAG_NORMAL_MAP))
    b = _gpu(rwr, gpu_ctx, suzanne, cam_inv, w, h, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
    assert np.array_equal(a["color_f32"], b["color_f32"]) and np.array_equal(a["color"], b["color"])
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(2.2, 1.7, 3.1), target=(0, 0, 0), aspect=w / h))
    flat = _gpu(rwr, gpu_ctx, cube, cam_inv, w, h, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
    gpu_ctx.set_normal_map(0, None)                                    # the map goes away: the flag finds nothing
    gpu_ctx.render(cam_inv, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS | rwr.FLAG_NORMAL_MAP))
    assert np.array_equal(gpu_ctx.readback(aux=True)["color_f32"], flat["color_f32"])
    with pytest.raises(rwr.RwrError):
        gpu_ctx.set_normal_map(5, cube["normal_map"])                  # no such part


def test_map_of_another_size_and_two_parts(rwr, orc, ref_loader, gpu_ctx, cube, suzanne):
    """A second part with its own diffuse texture and a normal map of a different size than its diffuse texture."""
    rng = np.random.default_rng(5)
    other = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in suzanne.items()}
    other["vertices"] = other["vertices"].copy()
    other["vertices"]["position"] += np.float32([2.0, 0.2, 0.0])
    other["normal_map"] = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    other["normal_map"][..., 2] |= 0x80                                # mostly outward-pointing normals
    parts = [cube, other]
    w, h = 160, 90
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(3.5, 1.7, 4.1), target=(1, 0, 0), aspect=w / h))
    gpu_ctx.upload_parts(parts); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(rwr.make_spheres()); gpu_ctx.resize(w, h)
    gpu_ctx.render(cam_inv, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS | rwr.FLAG_NORMAL_MAP))
    got = gpu_ctx.readback(aux=True)
    want = orc.render_path(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_params(1, 0, flags=orc.FLAG_NORMAL_MAP),
                           orc.make_spheres(), parts)
    _same_visibility(got, want)
    assert np.abs(got["color_f32"] - want["color_f32"]).max() <= COLOR_TOL
    ids = got["obj_id"]
    assert (ids >= len(cube["faces"])).any() and ((ids >= 0) & (ids < len(cube["faces"]))).any()   # both parts are in view
