"""Golden frames (tests/golden/frames.npz, produced by tests/golden/make_golden.py).
CPU: the oracle and the host-side camera code still reproduce them.
GPU: the HIP path reproduces them through the C ABI (ids / distance / depth
bit-exact, colour within 1e-4, RGBA8 within 1 LSB)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames.npz")


@pytest.fixture(scope="module")
def golden():
    z = np.load(GOLDEN, allow_pickle=False)
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/")[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


def test_golden_inventory(golden):
    assert set(golden) == {"suzanne_reference_camera", "suzanne_s_x15", "suzanne_oblique_spheres",
                           "cube_reference_camera", "cube_outside"}
    g = golden["suzanne_oblique_spheres"]
    assert (g["obj_id"] >= 0).any() and (g["obj_id"] < -1).any() and (g["obj_id"] == -1).any()


def test_oracle_reproduces_golden(orc, ref_loader, res_dir, golden):
    for name, g in golden.items():
        w, h = map(int, g["size"])
        model = ref_loader.load_model_compute(res_dir, str(g["scene"]))
        cam_inv = g["camera_inv"].view(orc.CAMERA_INV_DTYPE)
        out = orc.render_frame(cam_inv, orc.make_screen(w, h), g["spheres"].view(orc.SPHERE_DTYPE), model)
        assert np.array_equal(out["obj_id"], g["obj_id"].astype(np.int32)), name
        for k in ("hit_t", "depth", "color", "color_f32"):
            assert np.array_equal(out[k].view(np.uint8), g[k].view(np.uint8)), (name, k)


def test_host_camera_reproduces_golden_uniform(rwr, golden):
    for name, g in golden.items():
        cam = g["camera"].view(rwr.CAMERA_DTYPE)
        assert rwr.camera_build_inv_uniform(cam).tobytes() == g["camera_inv"].tobytes(), name


@pytest.mark.gpu
def test_gpu_reproduces_golden(rwr, gpu_ctx, golden, suzanne, cube):
    # scene arrays come from the loader the golden frames were made with (the cube's JPEG texture
    # decodes +-2 LSB differently in other decoders; decoder parity is tests/test_host_surface.py)
    models = {"suzanne_lowpoly.obj": suzanne, "cube.obj": cube}
    for name, g in golden.items():
        w, h = map(int, g["size"])
        gpu_ctx.upload_model(models[str(g["scene"])])
        gpu_ctx.set_instances(None)
        gpu_ctx.set_spheres(g["spheres"].view(rwr.SPHERE_DTYPE))
        gpu_ctx.resize(w, h)
        gpu_ctx.render(g["camera_inv"].view(rwr.CAMERA_INV_DTYPE), rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
        got = gpu_ctx.readback(aux=True)
        assert np.array_equal(got["obj_id"], g["obj_id"].astype(np.int32)), name
        assert np.array_equal(got["hit_t"].view(np.uint32), g["hit_t"].view(np.uint32)), name
        assert np.array_equal(got["depth"].view(np.uint32), g["depth"].view(np.uint32)), name
        assert np.abs(got["color_f32"] - g["color_f32"]).max() <= 1e-4, name
        assert np.abs(got["color"].astype(int) - g["color"].astype(int)).max() <= 1, name


# ---- frozen outputs of the extended integrator and of the reference's dormant parts (make_golden_ext.py) ----
GOLDEN_EXT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames_ext.npz")


@pytest.fixture(scope="module")
def golden_ext():
    z = np.load(GOLDEN_EXT, allow_pickle=False)
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/")[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


def _oracle_ext(orc, g, model):
    cam_inv = g["camera_inv"].view(orc.CAMERA_INV_DTYPE)
    spheres = g["spheres"].view(orc.SPHERE_DTYPE)
    if str(g["kind"]) == "path":
        w, h, spp, bounces, seed, side = map(int, g["size"])
        inst = g["instances"].view(orc.INSTANCE_DTYPE) if side else None
        return orc.render_path(cam_inv, orc.make_screen(w, h), orc.make_params(spp, bounces, seed=seed), spheres, model, instances=inst)
    w, h, ortho = map(int, g["size"])
    return orc.render_frame_ex(cam_inv, orc.make_screen(w, h), spheres, g["triangles"].view(orc.TRIANGLE_DTYPE), model, ortho=bool(ortho))


def test_oracle_reproduces_extended_golden(orc, ref_loader, res_dir, golden_ext):
    assert set(golden_ext) == {"path_inside_3spp_bounce", "path_outside_4spp", "path_grid2_2spp_bounce",
                               "dormant_triangles_perspective", "dormant_triangles_ortho"}
    model = ref_loader.load_model_compute(res_dir, "suzanne_lowpoly.obj")
    for name, g in golden_ext.items():
        out = _oracle_ext(orc, g, model)
        for k in ("obj_id", "hit_t", "depth", "color", "color_f32"):
            assert np.array_equal(out[k].view(np.uint8), g[k].view(np.uint8)), (name, k)
    assert (golden_ext["dormant_triangles_ortho"]["obj_id"] <= -10).any()
    assert (golden_ext["path_grid2_2spp_bounce"]["obj_id"] >= 111).any()     # a face of an instance other than the first


@pytest.mark.gpu
def test_gpu_reproduces_extended_golden(rwr, gpu_ctx, golden_ext, suzanne):
    try:
        for name, g in golden_ext.items():
            gpu_ctx.upload_model(suzanne)
            gpu_ctx.set_spheres(g["spheres"].view(rwr.SPHERE_DTYPE))
            cam_inv = g["camera_inv"].view(rwr.CAMERA_INV_DTYPE)
            if str(g["kind"]) == "path":
                w, h, spp, bounces, seed, side = map(int, g["size"])
                gpu_ctx.set_triangles(rwr.make_triangles())
                gpu_ctx.set_instances(g["instances"].view(rwr.INSTANCE_DTYPE) if side else None)
                params = rwr.make_params(spp=spp, max_bounces=bounces, seed=seed, flags=rwr.FLAG_AUX_OUTPUTS)
            else:
                w, h, ortho = map(int, g["size"])
                gpu_ctx.set_instances(None)
                gpu_ctx.set_triangles(g["triangles"].view(rwr.TRIANGLE_DTYPE))
                params = rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS | (rwr.FLAG_ORTHO_RAYS if ortho else 0))
            gpu_ctx.resize(w, h)
            gpu_ctx.render(cam_inv, params)
            got = gpu_ctx.readback(aux=True)
            assert np.array_equal(got["obj_id"], g["obj_id"]), name
            assert np.array_equal(got["hit_t"].view(np.uint32), g["hit_t"].view(np.uint32)), name
            assert np.array_equal(got["depth"].view(np.uint32), g["depth"].view(np.uint32)), name
            assert np.all(np.abs(got["color_f32"] - g["color_f32"]) <= 1e-4 + 1e-5 * np.abs(g["color_f32"])), name
            assert np.abs(got["color"].astype(int) - g["color"].astype(int)).max() <= 1, name
    finally:
        gpu_ctx.set_triangles(rwr.make_triangles())
        gpu_ctx.set_instances(None)
