"""Generates tests/golden/frames.npz from the CPU oracle (oracle/rt_oracle.c).

The reference cannot run here and ships no golden images, so these
vectors are ORACLE outputs ("parity unpinned" against the reference itself): they
freeze the oracle's behaviour for regression (CPU suite) and give the GPU suite
fixed inputs/outputs that do not depend on the oracle being importable.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc, ref_loader  # noqa: E402

RES = os.path.join(ROOT, "rust-wgpu-raytracing_amd", "res")

CASES = {
    # name: (scene, spheres, camera kwargs, (w, h))
    "suzanne_reference_camera": ("suzanne_lowpoly.obj", orc.REFERENCE_SPHERES, dict(eye=(0, 0, 0), target=(0, 0, -1)), (96, 54)),
    "suzanne_s_x15": ("suzanne_lowpoly.obj", orc.REFERENCE_SPHERES, dict(eye=(0, 0, 3), target=(0, 0, -1)), (96, 54)),
    "suzanne_oblique_spheres": ("suzanne_lowpoly.obj", orc.REFERENCE_SPHERES, dict(eye=(2.4, 0.9, 1.0), target=(0.3, 0.3, -2.0)), (80, 60)),
    "cube_reference_camera": ("cube.obj", orc.REFERENCE_SPHERES, dict(eye=(0, 0, 0), target=(0, 0, -1)), (64, 64)),
    "cube_outside": ("cube.obj", [((1.6, 1.2, 1.4), 0.5)], dict(eye=(2.2, 1.7, 3.1), target=(0, 0, 0)), (72, 48)),
}


def main():
    out = {}
    for name, (scene, spheres, cam_kw, (w, h)) in CASES.items():
        model = ref_loader.load_model_compute(RES, scene)
        cam = orc.make_camera(aspect=w / h, **cam_kw)
        cam_inv = orc.camera_build_inv_uniform(cam)
        sph = orc.make_spheres(spheres)
        r = orc.render_frame(cam_inv, orc.make_screen(w, h), sph, model)
        out[f"{name}/scene"] = np.array(scene)
        out[f"{name}/camera"] = cam.view(np.uint8)
        out[f"{name}/camera_inv"] = cam_inv.view(np.uint8)
        out[f"{name}/spheres"] = sph.view(np.uint8)
        out[f"{name}/size"] = np.array([w, h], np.int32)
        out[f"{name}/obj_id"] = r["obj_id"].astype(np.int16)
        out[f"{name}/hit_t"] = r["hit_t"]
        out[f"{name}/depth"] = r["depth"]
        out[f"{name}/color"] = r["color"]
        out[f"{name}/color_f32"] = r["color_f32"]
        print(name, (w, h), "mesh px", int((r["obj_id"] >= 0).sum()), "sphere px", int((r["obj_id"] < -1).sum()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frames.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
