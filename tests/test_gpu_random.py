"""Randomised parity: triangle soups, random cameras and sphere sets, odd frame sizes — the
conservative culling (per-frame face records, block/tile rectangles, sphere silhouette bounds)
and the BVH must never change a pixel relative to the oracle's brute-force loops."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _soup(ref_loader, rng, n_faces, extent, tri_size, tex):
    verts = np.zeros(3 * n_faces, ref_loader.VERTEX_DTYPE)
    centers = rng.uniform(-extent, extent, (n_faces, 1, 3))
    verts["position"] = (centers + rng.normal(0, tri_size, (n_faces, 3, 3))).reshape(-1, 3).astype(np.float32)
    verts["tex_coords"] = rng.uniform(-0.2, 1.2, (3 * n_faces, 2)).astype(np.float32)   # also exercises ClampToEdge
    faces = np.zeros(n_faces, ref_loader.FACE_DTYPE)
    faces["indices"] = np.arange(3 * n_faces, dtype=np.uint32).reshape(-1, 3)
    mat = np.zeros(1, ref_loader.MATERIAL_DTYPE)
    mat["ambient"], mat["diffuse"], mat["specular"] = 0.05, 0.8, 0.3
    return {"vertices": verts, "faces": faces, "material": mat, "texture": tex}


def _compare(got, want, tol=1e-4):
    assert np.array_equal(got["obj_id"], want["obj_id"])
    assert np.array_equal(got["hit_t"].view(np.uint32), want["hit_t"].view(np.uint32))
    assert np.array_equal(got["depth"].view(np.uint32), want["depth"].view(np.uint32))
    assert np.abs(got["color_f32"] - want["color_f32"]).max() <= tol


@pytest.mark.parametrize("seed", range(6))
def test_random_soup_frame(rwr, orc, ref_loader, gpu_ctx, suzanne, seed):
    rng = np.random.default_rng(1000 + seed)
    n_faces = int(rng.choice([1, 7, 64, 65, 300, 777]))              # around the 64 / 256 batch boundaries too
    model = _soup(ref_loader, rng, n_faces, extent=2.5, tri_size=float(rng.choice([0.05, 0.4, 1.5])), tex=suzanne["texture"])
    w, h = int(rng.integers(17, 140)), int(rng.integers(9, 100))
    eye = rng.uniform(-3, 3, 3)
    target = rng.uniform(-1, 1, 3)
    cam = rwr.make_camera(eye=eye, target=target, aspect=w / h, fovy=float(rng.uniform(20, 100)))
    cam_inv = rwr.camera_build_inv_uniform(cam)
    spheres = rwr.make_spheres([(tuple(rng.uniform(-2, 2, 3)), float(rng.uniform(0.1, 1.2))) for _ in range(int(rng.integers(0, 5)))])
    gpu_ctx.upload_model(model); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(spheres); gpu_ctx.resize(w, h)
    gpu_ctx.render(cam_inv, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
    got = gpu_ctx.readback(aux=True)
    want = orc.render_frame(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), spheres.view(orc.SPHERE_DTYPE), model)
    _compare(got, want)


@pytest.mark.parametrize("seed", range(4))
def test_random_soup_path(rwr, orc, ref_loader, gpu_ctx, suzanne, seed):
    rng = np.random.default_rng(2000 + seed)
    n_faces = int(rng.choice([3, 40, 129, 400]))
    model = _soup(ref_loader, rng, n_faces, extent=2.0, tri_size=float(rng.choice([0.2, 0.8])), tex=suzanne["texture"])
    w, h = int(rng.integers(20, 90)), int(rng.integers(12, 60))
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=rng.uniform(-3, 3, 3), target=rng.uniform(-0.5, 0.5, 3), aspect=w / h))
    spheres = rwr.make_spheres([(tuple(rng.uniform(-2, 2, 3)), float(rng.uniform(0.2, 0.9))) for _ in range(int(rng.integers(0, 3)))])
    spp = int(rng.choice([1, 3]))
    gpu_ctx.upload_model(model); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(spheres); gpu_ctx.resize(w, h)
    gpu_ctx.render(cam_inv, rwr.make_params(spp=spp, max_bounces=1, seed=seed, flags=rwr.FLAG_AUX_OUTPUTS))
    got = gpu_ctx.readback(aux=True)
    want = orc.render_path(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_params(spp, 1, seed=seed),
                           spheres.view(orc.SPHERE_DTYPE), model)
    _compare(got, want)


def test_axis_aligned_geometry_and_rays(rwr, orc, ref_loader, gpu_ctx, suzanne):
    """Axis-parallel faces and rays (zero direction components, exact ties, coplanar duplicates):
    the BVH slab test sees inf / NaN here and must stay conservative."""
    quad = lambda z, s: [(-s, -s, z), (s, -s, z), (s, s, z), (-s, -s, z), (s, s, z), (-s, s, z)]
    pos = quad(-3.0, 1.0) + quad(-3.0, 1.0) + quad(-5.0, 4.0) + [(1, -2, -4), (1, 2, -4), (1, 0, -1)]   # duplicate quad: exact t ties
    verts = np.zeros(len(pos), ref_loader.VERTEX_DTYPE)
    verts["position"] = pos
    verts["tex_coords"] = np.tile([(0, 0), (1, 0), (1, 1)], (len(pos) // 3, 1))
    faces = np.zeros(len(pos) // 3, ref_loader.FACE_DTYPE)
    faces["indices"] = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
    mat = np.zeros(1, ref_loader.MATERIAL_DTYPE); mat["ambient"] = 0.1; mat["specular"] = 0.2
    model = {"vertices": verts, "faces": faces, "material": mat, "texture": suzanne["texture"]}
    w, h = 65, 33   # odd: the centre pixel's ray is exactly (0, 0, -1)
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h))
    gpu_ctx.upload_model(model); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(rwr.make_spheres([])); gpu_ctx.resize(w, h)
    for params, oparams in ((rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS), orc.make_params(1, 0)),
                            (rwr.make_params(spp=2, max_bounces=1, seed=4, flags=rwr.FLAG_AUX_OUTPUTS), orc.make_params(2, 1, seed=4))):
        gpu_ctx.render(cam_inv, params)
        got = gpu_ctx.readback(aux=True)
        want = orc.render_path(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), oparams, orc.make_spheres([]), model)
        _compare(got, want)
    assert got["obj_id"][h // 2, w // 2] == 0     # duplicate faces: the lower index wins the tie


def test_skewed_scene_deep_sah_tree_is_rebuilt(rwr, orc, ref_loader, gpu_ctx, suzanne):
    """A dense cluster plus far outliers: the binned-SAH tree of this 6 000-face scene is 16 levels deep, beyond what
    the traversal stacks are sized for (csrc/bvh.hpp kBvhMaxDepth = 12), so the context rebuilds it with object-median
    splits (8 levels).  The BVH kernels — the reference frame with RWR_FLAG_USE_BVH, and bounce rays — must still
    find the brute-force winner everywhere."""
    rng = np.random.default_rng(42)
    n = 6000
    scale = np.where(np.arange(n) % 50 == 0, 1000.0 * 2.0 ** (np.arange(n) % 13), 1.0)[:, None, None]
    centers = scale * rng.uniform(-1, 1, (n, 1, 3))
    model = _soup(ref_loader, rng, n, extent=1.0, tri_size=0.05, tex=suzanne["texture"])
    model["vertices"]["position"] = (centers + rng.normal(0, 0.05, (n, 3, 3))).reshape(-1, 3).astype(np.float32)
    w, h = 96, 64
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.3, 0.2, 1.9), target=(0, 0, 0), aspect=w / h))
    gpu_ctx.upload_model(model); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(rwr.make_spheres()); gpu_ctx.resize(w, h)
    gpu_ctx.render(cam_inv, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS | rwr.FLAG_USE_BVH))
    got = gpu_ctx.readback(aux=True)
    want = orc.render_frame(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), model)
    _compare(got, want)
    assert (got["obj_id"] >= 0).mean() > 0.08
    gpu_ctx.render(cam_inv, rwr.make_params(spp=2, max_bounces=1, seed=3, flags=rwr.FLAG_AUX_OUTPUTS))
    got = gpu_ctx.readback(aux=True)
    want = orc.render_path(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_params(2, 1, seed=3), orc.make_spheres(), model)
    _compare(got, want)
