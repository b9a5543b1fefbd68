"""Screen-bin lists sized by a count pass (csrc/kernels_primary.hip k_bin_faces / k_bin_scan): a mesh of more than
50 000 faces at 3840x2160 through the two-pixel frame kernel (binned candidate lists) must give the very frame the
per-ray BVH kernel gives — and the oracle's on the rows it is asked for — and a frame whose lists do not fit the
buffer yet (the kernels then walk the whole scene) must not differ from the next one, which has the room."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _heightfield(ref_loader, n, tex):
    """n x n quads (2 n^2 faces) of a bumpy sheet in front of the camera, shared vertices."""
    g = np.linspace(-1.0, 1.0, n + 1, dtype=np.float64)
    x, y = np.meshgrid(g, g)
    z = -4.0 + 0.25 * np.sin(5.0 * x) * np.cos(4.0 * y) + 0.6 * x
    verts = np.zeros((n + 1) * (n + 1), ref_loader.VERTEX_DTYPE)
    verts["position"] = np.stack([2.2 * x, 1.3 * y, z], -1).reshape(-1, 3).astype(np.float32)
    verts["tex_coords"] = np.stack([(x + 1) / 2, (y + 1) / 2], -1).reshape(-1, 2).astype(np.float32)
    i = np.arange(n)[:, None] * (n + 1) + np.arange(n)[None, :]
    a, b, c, d = i, i + 1, i + n + 1, i + n + 2
    faces = np.zeros(2 * n * n, ref_loader.FACE_DTYPE)
    faces["indices"] = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([b, d, c], -1).reshape(-1, 3)]).astype(np.uint32)
    mat = np.zeros(1, ref_loader.MATERIAL_DTYPE)
    mat["ambient"], mat["diffuse"], mat["specular"] = 0.05, 0.8, 0.3
    return {"vertices": verts, "faces": faces, "material": mat, "texture": tex}


def _with_env(env, fn):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_fifty_thousand_faces_at_4k_binned_equals_bvh_and_oracle(rwr, orc, ref_loader, suzanne):
    model = _heightfield(ref_loader, 160, suzanne["texture"])
    assert len(model["faces"]) == 51200
    w, h = 3840, 2160
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h))

    def render():
        with rwr.Context(0) as ctx:   # RWR_AUTO_BVH_FACE_PX=0: the context must not pick the BVH kernel by itself
            ctx.upload_model(model)
            ctx.set_spheres(rwr.make_spheres())
            ctx.resize(w, h)
            ctx.render(cam_inv, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
            binned = ctx.readback(aux=True)
            ctx.render(cam_inv, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))       # second frame: capacity settled
            binned2 = ctx.readback(aux=True)
            ctx.render(cam_inv, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS | rwr.FLAG_USE_BVH))
            return binned, binned2, ctx.readback(aux=True)

    binned, binned2, bvh = _with_env({"RWR_AUTO_BVH_FACE_PX": "0"}, render)
    assert (binned["obj_id"] >= 0).mean() > 0.05      # (the reference camera sees wider than its fovy: z = -0.495)
    for k in ("obj_id", "hit_t", "depth", "color", "color_f32"):
        assert np.array_equal(binned[k].view(np.uint8), bvh[k].view(np.uint8)), k
        assert np.array_equal(binned[k].view(np.uint8), binned2[k].view(np.uint8)), k
    for r0, r1 in ((1080, 1081), (300, 301)):
        want = orc.render_path(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_params(1, 0), orc.make_spheres(),
                               model, rows=(r0, r1))
        for k in ("obj_id", "hit_t", "depth"):
            assert np.array_equal(binned[k][r0:r1].view(np.uint8), want[k][r0:r1].view(np.uint8)), (r0, k)
        assert np.abs(binned["color_f32"][r0:r1] - want["color_f32"][r0:r1]).max() <= 1e-4


def test_lists_that_do_not_fit_fall_back_to_the_whole_scene(rwr, orc, cube):
    """RWR_BIN_CAPACITY=16: the first frame's lists (hundreds of entries) cannot fit, every bin is marked 'no list' and the
    kernels walk all 428 faces; the buffer then grows and the second frame is binned.  Same frame both times."""
    w, h = 400, 300
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(2.2, 1.7, 3.1), target=(0, 0, 0), aspect=w / h))

    def render():
        with rwr.Context(0) as ctx:
            ctx.upload_model(cube)
            ctx.set_spheres(rwr.make_spheres())
            ctx.resize(w, h)
            frames = []
            for _ in range(3):
                ctx.render(cam_inv, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
                frames.append(ctx.readback(aux=True))
            return frames

    frames = _with_env({"RWR_BIN_CAPACITY": "16", "RWR_AUTO_BVH_FACE_PX": "0"}, render)
    want = orc.render_frame(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), cube)
    for f in frames:
        for k in ("obj_id", "hit_t", "depth"):
            assert np.array_equal(f[k].view(np.uint8), want[k].view(np.uint8)), k
        assert np.array_equal(f["color_f32"], frames[0]["color_f32"])
        assert np.abs(f["color_f32"] - want["color_f32"]).max() <= 1e-4
