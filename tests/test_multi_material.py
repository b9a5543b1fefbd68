"""Extension ("full loader fidelity"): scenes of several meshes, each with its own
material and texture — everything resources::load_model_compute returns in Model{meshes, materials}
(the reference's TriangleList binds meshes[0]/materials[0] only, triangle_list.rs:212-245)."""
import numpy as np
import pytest


def _write_two_object_scene(d):
    from PIL import Image
    rng = np.random.default_rng(5)
    Image.fromarray(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8), "RGB").save(d / "a.png")
    Image.fromarray(rng.integers(0, 256, (8, 32, 4), dtype=np.uint8), "RGBA").save(d / "b.png")
    (d / "two.mtl").write_text(
        "newmtl red\nKa 0.20 0.02 0.02\nKd 1 1 1\nKs 0.5 0.5 0.5\nmap_Kd a.png\n"
        "newmtl blue\nKa 0.02 0.02 0.30\nKd 1 1 1\nKs 0.1 0.2 0.3\nmap_Kd b.png\n")
    (d / "two.obj").write_text(
        "mtllib two.mtl\n"
        "o left\nv -2 -1 -4\nv 0 -1 -4\nv 0 1 -4\nv -2 1 -4\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "usemtl red\nf 1/1 2/2 3/3 4/4\n"
        "o right\nv -0.5 -1 -5\nv 2 -1 -3\nv 2 1 -3\nv -0.5 1 -5\n"
        "usemtl blue\nf 5/1 6/2 7/3\nf 5/1 7/3 8/4\n")


def test_loader_exposes_every_part_with_its_material(rwr, ref_loader, tmp_path):
    _write_two_object_scene(tmp_path)
    got = rwr.load_model_parts("two.obj", str(tmp_path))
    want = ref_loader.load_model_parts(str(tmp_path), "two.obj")
    assert len(got) == len(want) == 2
    for a, b in zip(got, want):
        assert a["vertices"].tobytes() == b["vertices"].tobytes() and a["faces"].tobytes() == b["faces"].tobytes()
        assert a["material"].tobytes() == b["material"].tobytes() and np.array_equal(a["texture"], b["texture"])
    np.testing.assert_allclose(got[1]["material"]["ambient"][0], (0.02, 0.02, 0.30))
    assert got[0]["texture"].shape == (16, 16, 4) and got[1]["texture"].shape == (8, 32, 4)
    first = rwr.load_model_compute("two.obj", str(tmp_path))            # the reference's view: meshes[0] / materials[0]
    assert first["vertices"].tobytes() == got[0]["vertices"].tobytes() and first["n_meshes"] == 2 and first["n_materials"] == 2


@pytest.mark.gpu
def test_two_materials_match_oracle(rwr, orc, gpu_ctx, tmp_path):
    _write_two_object_scene(tmp_path)
    parts = rwr.load_model_parts("two.obj", str(tmp_path))
    w, h = 120, 72
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.2, 0.1, 0.5), target=(0, 0, -4), aspect=w / h))
    spheres = rwr.make_spheres([((0.6, 0.5, -3.5), 0.4)])
    for spp, bounces in ((1, 0), (3, 1)):
        gpu_ctx.upload_parts(parts); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(spheres); gpu_ctx.resize(w, h)
        gpu_ctx.render(cam_inv, rwr.make_params(spp=spp, max_bounces=bounces, seed=2, flags=rwr.FLAG_AUX_OUTPUTS))
        got = gpu_ctx.readback(aux=True)
        want = orc.render_path(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_params(spp, bounces, seed=2),
                               spheres.view(orc.SPHERE_DTYPE), parts)
        assert np.array_equal(got["obj_id"], want["obj_id"])
        assert np.array_equal(got["depth"].view(np.uint32), want["depth"].view(np.uint32))
        assert np.abs(got["color_f32"] - want["color_f32"]).max() <= 1e-4
        if spp == 1:
            single = got
    ids = set(np.unique(single["obj_id"]).tolist())
    assert {0, 1, 2, 3} <= ids                                         # faces of both parts are visible (2 + 2 triangles)
    # the two parts really are shaded with different materials: on the one-sample frame (no pixel mixes
    # objects) every pixel of a part is at least its own ambient term
    left = single["color_f32"][single["obj_id"] == 0][:, :3].min(0)
    right = single["color_f32"][single["obj_id"] == 2][:, :3].min(0)
    assert left[0] >= 0.2 - 1e-5 and right[2] >= 0.3 - 1e-5
    assert right[0] < 0.2 or left[2] < 0.3                             # and they are not the same material


@pytest.mark.gpu
def test_suzanne_plus_cube_instanced(rwr, orc, gpu_ctx, suzanne, cube):
    """Two real assets as one scene, instanced: face index = instance * total_faces + face."""
    parts = [suzanne, cube]
    inst = rwr.make_instance_grid(2, 3.0)
    w, h = 96, 64
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(-1.5, 1.0, 6.0), target=(-1.5, 0, 0), aspect=w / h))
    gpu_ctx.upload_parts(parts); gpu_ctx.set_instances(inst); gpu_ctx.set_spheres(rwr.make_spheres()); gpu_ctx.resize(w, h)
    gpu_ctx.render(cam_inv, rwr.make_params(spp=2, max_bounces=1, seed=8, flags=rwr.FLAG_AUX_OUTPUTS))
    got = gpu_ctx.readback(aux=True)
    want = orc.render_path(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_params(2, 1, seed=8),
                           orc.make_spheres(), parts, instances=inst.view(orc.INSTANCE_DTYPE))
    assert np.array_equal(got["obj_id"], want["obj_id"])
    assert np.array_equal(got["hit_t"].view(np.uint32), want["hit_t"].view(np.uint32))
    assert np.abs(got["color_f32"] - want["color_f32"]).max() <= 1e-4
    total = 111 + 428
    local = got["obj_id"][got["obj_id"] >= 0] % total
    assert (local < 111).any() and (local >= 111).any()
    gpu_ctx.set_instances(None)
