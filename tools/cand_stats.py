"""Debug: candidate-face statistics of the per-wave cull of the frame kernel (needs a GPU)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
r = g.load_package()
DEBUG_COUNTS = 1 << 16
for scene, eye, (w, h) in [("suzanne_lowpoly.obj", (0, 0, 0), (1920, 1080)), ("suzanne_lowpoly.obj", (0, 0, 3), (1920, 1080)),
                           ("cube.obj", (0, 0, 0), (256, 256))]:
    m = r.load_model_compute(scene)
    with r.Context(0) as ctx:
        ctx.upload_model(m); ctx.set_spheres(r.make_spheres()); ctx.resize(w, h)
        ci = r.camera_build_inv_uniform(r.make_camera(eye=eye, aspect=w / h))
        ctx.render(ci, r.make_params(flags=r.FLAG_AUX_OUTPUTS | DEBUG_COUNTS))
        o = ctx.readback(aux=True)
    listed = o["obj_id"][::4, ::32].astype(float); tested = o["hit_t"][::4, ::32]   # one sample per 32x4 tile
    print(scene, eye, "faces", len(m["faces"]), "faces past the tile cull per wave mean/max", listed.mean(), listed.max(),
          "exact tests per wave mean/max", tested.mean(), tested.max())
