"""Debug: frame kernel vs RWR_FLAG_USE_BVH over the eye distance (needs a GPU): where does a mesh become
'dense' (many faces per tile) enough for the per-ray BVH traversal to win?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
r = g.load_package()
N = 200
for scene in ("suzanne_lowpoly.obj", "cube.obj"):
    m = r.load_model_compute(scene)
    with r.Context(0) as ctx:
        ctx.upload_model(m); ctx.set_spheres(r.make_spheres(r.REFERENCE_SPHERES))
        for (w, h) in ((1920, 1080), (256, 256)):
            ctx.resize(w, h)
            for z in (0, 3, 5, 8, 12, 20, 40):
                ci = r.camera_build_inv_uniform(r.make_camera(eye=(0, 0, z), aspect=w / h))
                out = []
                for flags in (0, r.FLAG_USE_BVH):
                    call = ctx.render_call(ci, r.make_params(flags=flags), (0, h))
                    for _ in range(10): call()
                    ctx.synchronize(); ctx.timer_begin()
                    for _ in range(N): call()
                    out.append(ctx.timer_end() / N * 1e3)
                print(f"{scene} {w}x{h} eye z={z:3d}: tiles {out[0]:8.2f} us/frame, bvh {out[1]:8.2f} us/frame")
