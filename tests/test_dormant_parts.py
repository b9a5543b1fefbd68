"""the parts the reference carries but never dispatches — the single-triangle model
(src/models/triangle/) and pixelToRay_ortho (triangle_list/compute.wgsl:166-174).  Oracle known answers
on the CPU, GPU parity through the C ABI."""
import numpy as np
import pytest

TRI = ((-1.0, -1.0, -2.0), (1.0, -1.0, -2.0), (0.0, 1.0, -2.0))          # N = (0, 0, 4), faces +z
SMALL = ((0.5, -0.4, 1.5), (1.4, -0.5, 1.6), (1.0, 0.45, 1.4))           # in front of the mesh; |N| < 1: unsaturated shading


def test_oracle_single_triangle_known_answers(orc, cube):
    empty = dict(cube, faces=cube["faces"][:0])
    w, h = 64, 48
    cam = orc.camera_build_inv_uniform(orc.make_camera(aspect=w / h, eye=(0, 0, 3), target=(0, 0, -1)))
    out = orc.render_frame_ex(cam, orc.make_screen(w, h), orc.make_spheres([]), orc.make_triangles([TRI]), empty, ortho=True)
    # orthographic rays run along -z from z = 3: every hit is 5 away, and the 10 x 10 window (origin +- 5 nds)
    # shows the triangle's 2 x 2 footprint: area 2 of 100 -> 2 % of the pixels
    hit = out["obj_id"] == -10
    assert np.all(out["hit_t"][hit] == 5.0) and abs(hit.mean() - 0.02) < 0.005
    assert np.array_equal(np.unique(out["obj_id"]), [-10, -1])
    # shading (triangle/compute.wgsl:171-187) with the UN-NORMALISED normal (0, 0, 4): no diffuse (the light
    # comes from +y / -x / -z), specular 0.5 * (4 * cos)^32
    nl = -np.array([1.0, -5.0, 1.0]) / np.sqrt(27.0)
    half = nl - np.array([0.0, 0.0, -1.0]); half /= np.linalg.norm(half)
    spec = 0.5 * (4.0 * half[2]) ** 32
    got = out["color_f32"][hit][0]
    np.testing.assert_allclose(got[:3], [0.1 + spec, spec, spec], rtol=2e-5)
    assert got[3] == 2.0 and np.all(out["color"][hit] == 255)
    d = out["depth"][hit][0]
    assert np.float32(1.0) - np.float32(orc.to_non_linear_depth(5.0)) == d
    # perspective rays: the centre ray hits the same plane at (nearly) the same distance
    persp = orc.render_frame_ex(cam, orc.make_screen(w, h), orc.make_spheres([]), orc.make_triangles([TRI]), empty)
    assert persp["obj_id"][h // 2, w // 2] == -10 and abs(persp["hit_t"][h // 2, w // 2] - 5.0) < 0.01
    # nothing dormant switched on: the plain frame, bit for bit
    a = orc.render_frame(cam, orc.make_screen(w, h), orc.make_spheres(), cube)
    b = orc.render_frame_ex(cam, orc.make_screen(w, h), orc.make_spheres(), orc.make_triangles(), cube)
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_host_triangle_layout(rwr):
    t = rwr.make_triangles([TRI])
    assert t.dtype.itemsize == 48 and t["p1"][0].tolist() == [1.0, -1.0, -2.0]


@pytest.mark.gpu
@pytest.mark.parametrize("ortho", [False, True])
def test_dormant_parts_match_oracle(rwr, orc, gpu_ctx, suzanne, ortho):
    w, h = 161, 90            # odd width: partial tiles
    cam = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.3, 0.2, 3.0), target=(0, 0, -1), aspect=w / h))
    spheres = rwr.make_spheres(rwr.REFERENCE_SPHERES)
    tris = rwr.make_triangles([SMALL, ((-1.6, -0.2, 1.0), (-1.0, -0.3, 0.8), (-1.3, 0.5, 1.2))])
    flags = rwr.FLAG_AUX_OUTPUTS | (rwr.FLAG_ORTHO_RAYS if ortho else 0)
    gpu_ctx.upload_model(suzanne); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(spheres); gpu_ctx.resize(w, h)
    try:
        gpu_ctx.set_triangles(tris)
        gpu_ctx.render(cam, rwr.make_params(flags=flags))
        got = gpu_ctx.readback(aux=True)
        want = orc.render_frame_ex(cam.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), spheres.view(orc.SPHERE_DTYPE),
                                   tris.view(orc.TRIANGLE_DTYPE), suzanne, ortho=ortho)
        assert np.array_equal(got["obj_id"], want["obj_id"])
        assert np.array_equal(got["hit_t"].view(np.uint32), want["hit_t"].view(np.uint32))
        assert np.array_equal(got["depth"].view(np.uint32), want["depth"].view(np.uint32))
        ids = set(np.unique(got["obj_id"]).tolist())
        assert -10 in ids and any(i >= 0 for i in ids)            # a single triangle and the mesh are visible
        assert -11 in ids
        # colour: 1e-4 absolute, and relative where the un-normalised normal drives x^32 beyond 1
        diff = np.abs(got["color_f32"] - want["color_f32"])
        assert np.all(diff <= 1e-4 + 1e-5 * np.abs(want["color_f32"]))
        assert np.abs(got["color"].astype(int) - want["color"].astype(int)).max() <= 1
        # the dormant parts apply to the reference frame only
        with pytest.raises(rwr.RwrError):
            gpu_ctx.render(cam, rwr.make_params(spp=2, flags=flags))
        with pytest.raises(rwr.RwrError):
            gpu_ctx.render(cam, rwr.make_params(flags=flags | rwr.FLAG_USE_BVH))
        with pytest.raises(rwr.RwrError):
            gpu_ctx.set_triangles(rwr.make_triangles([TRI] * 9))
        # without triangles and ortho the frame is the plain frame again
        gpu_ctx.set_triangles(rwr.make_triangles())
        gpu_ctx.render(cam, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
        plain = gpu_ctx.readback(aux=True)
        ref = orc.render_frame(cam.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), spheres.view(orc.SPHERE_DTYPE), suzanne)
        assert np.array_equal(plain["obj_id"], ref["obj_id"]) and np.array_equal(plain["depth"], ref["depth"])
    finally:
        gpu_ctx.set_triangles(rwr.make_triangles())
