"""Debug: kernel time (HIP events on the launch stream) for scene variants (needs a GPU)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
r = g.load_package()
suz = r.load_model_compute("suzanne_lowpoly.obj")
cube = r.load_model_compute("cube.obj")
empty = dict(suz, faces=suz["faces"][:0])
N = 300
def run(label, model, spheres, eye, w, h, flags=0):
    with r.Context(0) as ctx:
        ctx.upload_model(model); ctx.set_spheres(r.make_spheres(spheres)); ctx.resize(w, h)
        ci = r.camera_build_inv_uniform(r.make_camera(eye=eye, aspect=w / h))
        call = ctx.render_call(ci, r.make_params(flags=flags), (0, h))
        for _ in range(20): call()
        ctx.synchronize()
        ctx.timer_begin()
        for _ in range(N): call()
        ms = ctx.timer_end()
    print(f"{label:44s} {w}x{h}: {ms / N * 1e3:8.2f} us/frame")
S = r.REFERENCE_SPHERES
for (w, h) in [(1920, 1080), (256, 256)]:
    run("empty scene (raygen+store)", empty, [], (0, 0, 0), w, h)
    run("2 spheres only", empty, S, (0, 0, 0), w, h)
    run("suzanne, eye 0 (inside), no spheres", suz, [], (0, 0, 0), w, h)
    run("suzanne, eye 0 (inside), 2 spheres", suz, S, (0, 0, 0), w, h)
    run("suzanne, eye (0,0,3)", suz, S, (0, 0, 3), w, h)
    run("suzanne, eye (0,0,40) (all misses)", suz, S, (0, 0, 40), w, h)
    run("suzanne, eye 0, NO_CULL", suz, S, (0, 0, 0), w, h, r.FLAG_NO_CULL)
    run("cube, eye 0", cube, S, (0, 0, 0), w, h)
    run("suzanne, eye 0, USE_BVH", suz, S, (0, 0, 0), w, h, r.FLAG_USE_BVH)
    run("suzanne, eye (0,0,40), USE_BVH", suz, S, (0, 0, 40), w, h, r.FLAG_USE_BVH)
    run("suzanne, eye 0, ONE_PIXEL_PER_LANE", suz, S, (0, 0, 0), w, h, r.FLAG_ONE_PIXEL_PER_LANE)
