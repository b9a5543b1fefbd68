"""Generates tests/golden/frames_ext.npz from the CPU oracle: frozen outputs of the parts that have no
counterpart in the reference's dispatched frame — the extended integrator (samples, one bounce, instances;
DESIGN.md §6) and the reference's dormant parts (single-triangle passes, pixelToRay_ortho).
Oracle outputs, not reference outputs ("parity unpinned").

    python tests/golden/make_golden_ext.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc, ref_loader  # noqa: E402

RES = os.path.join(ROOT, "rust-wgpu-raytracing_amd", "res")
TRIS = [((0.5, -0.4, 1.5), (1.4, -0.5, 1.6), (1.0, 0.45, 1.4)), ((-1.6, -0.2, 1.0), (-1.0, -0.3, 0.8), (-1.3, 0.5, 1.2))]

PATH_CASES = {
    # name: (camera kwargs, (w, h), spp, bounces, seed, instance grid side or 0)
    "path_inside_3spp_bounce": (dict(eye=(0, 0, 0), target=(0, 0, -1)), (64, 36), 3, 1, 11, 0),
    "path_outside_4spp": (dict(eye=(0, 0, 3), target=(0, 0, -1)), (64, 36), 4, 0, 5, 0),
    "path_grid2_2spp_bounce": (dict(eye=(-1.5, 1.0, 6.0), target=(-1.5, 0, 0)), (64, 40), 2, 1, 8, 2),
}
DORMANT_CASES = {
    # name: (camera kwargs, (w, h), ortho)
    "dormant_triangles_perspective": (dict(eye=(0.3, 0.2, 3.0), target=(0, 0, -1)), (72, 40), False),
    "dormant_triangles_ortho": (dict(eye=(0.3, 0.2, 3.0), target=(0, 0, -1)), (72, 40), True),
}


def instance_grid(side, spacing):
    """lib.rs:400-421: side x side grid in the xz plane, `spacing` apart, identity rotation."""
    inst = np.zeros(side * side, dtype=orc.INSTANCE_DTYPE)
    k = 0
    for z in range(side):
        for x in range(side):
            m = np.eye(4, dtype=np.float32)
            m[3, 0] = spacing * (x - side / 2.0)   # column-major: m[col][row]; translation in column 3
            m[3, 2] = spacing * (z - side / 2.0)
            inst["model"][k] = m
            k += 1
    return inst


def main():
    model = ref_loader.load_model_compute(RES, "suzanne_lowpoly.obj")
    out = {}
    for name, (cam_kw, (w, h), spp, bounces, seed, side) in PATH_CASES.items():
        cam_inv = orc.camera_build_inv_uniform(orc.make_camera(aspect=w / h, **cam_kw))
        sph = orc.make_spheres(orc.REFERENCE_SPHERES)
        inst = instance_grid(side, 3.0) if side else None
        r = orc.render_path(cam_inv, orc.make_screen(w, h), orc.make_params(spp, bounces, seed=seed), sph, model, instances=inst)
        out[f"{name}/kind"] = np.array("path")
        out[f"{name}/camera_inv"] = cam_inv.view(np.uint8)
        out[f"{name}/spheres"] = sph.view(np.uint8)
        out[f"{name}/size"] = np.array([w, h, spp, bounces, seed, side], np.int32)
        if inst is not None:
            out[f"{name}/instances"] = inst.view(np.uint8)
        for k in ("obj_id", "hit_t", "depth", "color", "color_f32"):
            out[f"{name}/{k}"] = r[k]
        print(name, (w, h), "mesh px", int((r["obj_id"] >= 0).sum()))
    for name, (cam_kw, (w, h), ortho) in DORMANT_CASES.items():
        cam_inv = orc.camera_build_inv_uniform(orc.make_camera(aspect=w / h, **cam_kw))
        sph = orc.make_spheres(orc.REFERENCE_SPHERES)
        tris = orc.make_triangles(TRIS)
        r = orc.render_frame_ex(cam_inv, orc.make_screen(w, h), sph, tris, model, ortho=ortho)
        out[f"{name}/kind"] = np.array("dormant")
        out[f"{name}/camera_inv"] = cam_inv.view(np.uint8)
        out[f"{name}/spheres"] = sph.view(np.uint8)
        out[f"{name}/triangles"] = tris.view(np.uint8)
        out[f"{name}/size"] = np.array([w, h, 1 if ortho else 0], np.int32)
        for k in ("obj_id", "hit_t", "depth", "color", "color_f32"):
            out[f"{name}/{k}"] = r[k]
        print(name, (w, h), "triangle px", int((r["obj_id"] <= -10).sum()), "mesh px", int((r["obj_id"] >= 0).sum()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frames_ext.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
