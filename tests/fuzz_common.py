"""Randomised parity against the CPU oracle (test infrastructure): random triangle soups and the real meshes,
cameras (fields of view from 1 to 175 degrees), up to 8 spheres, frame sizes from 1x1, 1-3 frames in flight, a
quarter of the frames path-traced (some with rigid instances or a second part with its own material), a tenth with
single-triangle passes / orthographic rays.  Every frame must match the oracle bit for bit in object ids, hit
distances and depth, and within 1e-4 in colour.  Used by tests/test_gpu_fuzz.py (bounded, in the driver's -m gpu
run) and tools/fuzz_parity.py (as long as you like).  What this validates above all: the conservative culling
margins (csrc/rwr_cull.h) and the BVH / packet traversal never change which surface a ray sees."""
import time

import numpy as np


def soup(ref_loader, rng, n_faces, extent, tri_size, tex):
    verts = np.zeros(3 * n_faces, ref_loader.VERTEX_DTYPE)
    centers = rng.uniform(-extent, extent, (n_faces, 1, 3))
    verts["position"] = (centers + rng.normal(0, tri_size, (n_faces, 3, 3))).reshape(-1, 3).astype(np.float32)
    verts["tex_coords"] = rng.uniform(-0.2, 1.2, (3 * n_faces, 2)).astype(np.float32)   # also exercises ClampToEdge
    faces = np.zeros(n_faces, ref_loader.FACE_DTYPE)
    faces["indices"] = np.arange(3 * n_faces, dtype=np.uint32).reshape(-1, 3)
    mat = np.zeros(1, ref_loader.MATERIAL_DTYPE)
    mat["ambient"], mat["diffuse"], mat["specular"] = 0.05, 0.8, 0.3
    return {"vertices": verts, "faces": faces, "material": mat, "texture": tex}


def run(r, orc, ref_loader, ctx, seed, seconds, path_fraction=0.25, many_samples=False, verbose=False):
    """r: the product package; ctx: an r.Context.  Returns (frames, path-traced frames, dormant frames, worst colour difference)."""
    _soup = soup
    t_end = time.time() + seconds
    rng = np.random.default_rng(seed)
    meshes = [ref_loader.load_model_compute(r.RES_DIR, n) for n in ("suzanne_lowpoly.obj", "cube.obj")]
    tex = meshes[0]["texture"]
    n_frames = n_path = n_dormant = 0
    t_note = time.time()
    worst = 0.0
    while time.time() < t_end:
        n_faces = int(rng.choice([1, 2, 7, 63, 64, 65, 128, 129, 255, 256, 257, 300, 777, 1500]))
        kind = rng.random()
        if kind < 0.5:
            model = _soup(ref_loader, rng, n_faces, extent=float(rng.choice([0.5, 2.5, 8.0])), tri_size=float(rng.choice([0.02, 0.05, 0.4, 1.5, 6.0])), tex=tex)
        else:   # real meshes: exactly shared edges and vertices (inclusive edge tests, lowest-index ties)
            model = meshes[int(rng.integers(0, len(meshes)))]
            n_faces = len(model["faces"])
        w, h = int(rng.integers(1, 260)), int(rng.integers(1, 150))
        cam = r.make_camera(eye=rng.uniform(-4, 4, 3) * float(rng.choice([0.1, 1.0, 1.0, 6.0])), target=rng.uniform(-1, 1, 3), aspect=w / h,
                            fovy=float(rng.choice([rng.uniform(15, 110), rng.uniform(1, 15), rng.uniform(110, 175)])))
        ci = r.camera_build_inv_uniform(cam)
        spheres = r.make_spheres([(tuple(rng.uniform(-3, 3, 3)), float(rng.uniform(0.05, 1.5))) for _ in range(int(rng.integers(0, 9)))])
        parts = None
        if kind < 0.5 and rng.random() < 0.3 and n_faces < 400:   # a second part with its own material and texture
            tex2 = rng.integers(0, 256, (int(rng.integers(1, 40)), int(rng.integers(1, 40)), 4), dtype=np.uint8)
            other = _soup(ref_loader, rng, int(rng.integers(1, 200)), extent=2.0, tri_size=0.5, tex=tex2)
            other["material"]["ambient"], other["material"]["specular"] = rng.uniform(0, 0.3, 3), rng.uniform(0, 1, 3)
            parts = [model, other]
            ctx.upload_parts(parts)
        else:
            ctx.upload_model(model)
        ctx.set_spheres(spheres); ctx.resize(w, h)
        ctx.set_frames_in_flight(int(rng.integers(1, 4)))
        path = (rng.random() < path_fraction or parts is not None) and w * h * n_faces < 4e6   # several parts: the oracle's path renderer
        if parts is not None and not path:
            ctx.upload_model(model); parts = None
        dormant = (not path) and rng.random() < 0.12 and w * h * n_faces < 2e7
        inst = None
        if path and rng.random() < 0.4 and n_faces < 400:   # rigid instances: rotation about y + translation
            k = int(rng.integers(2, 6))
            inst = np.zeros(k, dtype=r.INSTANCE_DTYPE)
            for i in range(k):
                a = rng.uniform(0, 2 * np.pi); c_, s_ = np.float32(np.cos(a)), np.float32(np.sin(a))
                m = np.eye(4, dtype=np.float32)
                m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c_, -s_, s_, c_       # column-major m[col][row]
                m[3, :3] = rng.uniform(-3, 3, 3)
                inst["model"][i] = m
        ctx.set_instances(inst)
        ctx.set_triangles(r.make_triangles())
        if dormant:
            tris = r.make_triangles([tuple(tuple(rng.uniform(-2, 2, 3)) for _ in range(3)) for _ in range(int(rng.integers(0, 4)))])
            ortho = bool(rng.random() < 0.5)
            ctx.set_triangles(tris)
            ctx.render(ci, r.make_params(flags=r.FLAG_AUX_OUTPUTS | (r.FLAG_ORTHO_RAYS if ortho else 0)))
            want = orc.render_frame_ex(ci.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), spheres.view(orc.SPHERE_DTYPE),
                                       tris.view(orc.TRIANGLE_DTYPE), model, ortho=ortho)
            n_dormant += 1
        elif path:
            spp, b, sd = int(rng.choice([1, 1, 2, 3])), int(rng.integers(0, 2)), int(rng.integers(0, 1000))
            if many_samples and w * h * n_faces < 4e5:   # several launch groups, well-filled ray pools
                spp, b = int(rng.choice([5, 17, 20, 33])), 1
            ctx.render(ci, r.make_params(spp=spp, max_bounces=b, seed=sd, flags=r.FLAG_AUX_OUTPUTS))
            want = orc.render_path(ci.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_params(spp, b, seed=sd),
                                   spheres.view(orc.SPHERE_DTYPE), parts if parts is not None else model,
                                   instances=None if inst is None else inst.view(orc.INSTANCE_DTYPE))
            n_path += 1
        else:
            ctx.render(ci, r.make_params(flags=r.FLAG_AUX_OUTPUTS))
            want = orc.render_frame(ci.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), spheres.view(orc.SPHERE_DTYPE), model)
        got = ctx.readback(aux=True)
        tag = (seed, n_frames, n_faces, w, h, path, dormant, None if inst is None else len(inst))
        assert np.array_equal(got["obj_id"], want["obj_id"]), tag
        assert np.array_equal(got["hit_t"].view(np.uint32), want["hit_t"].view(np.uint32)), tag
        assert np.array_equal(got["depth"].view(np.uint32), want["depth"].view(np.uint32)), tag
        d = float(np.abs(got["color_f32"] - want["color_f32"]).max())
        if dormant:
            # the single-triangle shading raises dot(half_dir, N) with an UN-NORMALISED N to the 32nd power: values up
            # to 1e30 whose relative conditioning is 32 x that of the dot product; compared relative to the value there
            big = np.abs(want["color_f32"]) > 1.0
            same = got["color_f32"] == want["color_f32"]          # includes x^32 overflowing to +inf on both sides
            with np.errstate(invalid="ignore"):
                rel = np.where(same, 0.0, np.abs(got["color_f32"] - want["color_f32"]) / np.maximum(np.abs(want["color_f32"]), 1.0))
            assert float(rel[big].max() if big.any() else 0.0) <= 5e-3, tag
            d = float(np.abs(got["color_f32"] - want["color_f32"])[~big].max()) if (~big).any() else 0.0
        assert d <= 1e-4, tag + (d,)
        worst = max(worst, d)
        n_frames += 1
        if verbose and time.time() - t_note > 30.0:
            t_note = time.time()
            print(f"... {n_frames} frames so far", flush=True)
    return n_frames, n_path, n_dormant, worst
