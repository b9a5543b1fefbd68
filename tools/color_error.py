"""Debug: largest colour difference between the GPU frame and the oracle at BASELINE configs[1] (needs a GPU)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
r = g.load_package()
from oracle import oracle as orc, ref_loader
w, h = 1920, 1080
for scene, eye in (("suzanne_lowpoly.obj", (0, 0, 0)), ("suzanne_lowpoly.obj", (0, 0, 3)), ("cube.obj", (0, 0, 0))):
    model = ref_loader.load_model_compute(r.RES_DIR, scene)
    cam = orc.camera_build_inv_uniform(orc.make_camera(aspect=w / h, eye=eye, target=(0, 0, -1)))
    want = orc.render_frame(cam, orc.make_screen(w, h), orc.make_spheres(), model)
    with r.Context(0) as ctx:
        ctx.upload_model(model);   # the oracle's own decode of the texture (JPEG decoders differ by an LSB or two)
        ctx.set_spheres(r.make_spheres()); ctx.resize(w, h)
        ctx.render(cam.view(r.CAMERA_INV_DTYPE) if hasattr(r, "CAMERA_INV_DTYPE") else cam, r.make_params(flags=r.FLAG_AUX_OUTPUTS))
        got = ctx.readback(aux=True)
    d = np.abs(got["color_f32"] - want["color_f32"]).max()
    d8 = np.abs(got["color"].astype(int) - want["color"].astype(int))
    print(f"{scene} eye {eye}: ids equal {np.array_equal(got['obj_id'], want['obj_id'])}, depth bits equal "
          f"{np.array_equal(got['depth'].view(np.uint32), want['depth'].view(np.uint32))}, hit_t bits equal "
          f"{np.array_equal(got['hit_t'].view(np.uint32), want['hit_t'].view(np.uint32))}, max |dcolour| {d:.3e}, "
          f"rgba8 max diff {d8.max()}, bytes differing {(d8 > 0).mean() * 100:.4f} %")
    if d > 1e-4:
        diff = np.abs(got["color_f32"] - want["color_f32"]).max(-1)
        ys, xs = np.nonzero(diff > 1e-4)
        print("   pixels over 1e-4:", len(ys), "faces:", np.unique(got["obj_id"][ys, xs])[:20])
        y, x = np.unravel_index(diff.argmax(), diff.shape)
        f = int(got["obj_id"][y, x])
        print("   worst", (x, y), "face", f, "gpu", got["color_f32"][y, x], "oracle", want["color_f32"][y, x], "t", got["hit_t"][y, x])
        V = model["vertices"]; F = model["faces"]["indices"][f]
        print("   corners", V["position"][F], "uv", V["tex_coords"][F])
