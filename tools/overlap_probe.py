"""Debug: do frames rendered on two streams overlap?  Two contexts (own stream, own buffers), same scene;
frames alternate between them.  Needs a GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
r = g.load_package()
suz = r.load_model_compute("suzanne_lowpoly.obj")
w, h, N = 1920, 1080, 2000
ci = r.camera_build_inv_uniform(r.make_camera(eye=(0, 0, 0), aspect=w / h))
ctxs = [r.Context(0) for _ in range(3)]
calls = []
for c in ctxs:
    c.upload_model(suz); c.set_spheres(r.make_spheres(r.REFERENCE_SPHERES)); c.resize(w, h)
    calls.append(c.render_call(ci, r.make_params(), (0, h)))
for k in (1, 2, 3):
    for c in ctxs: c.synchronize()
    for i in range(60): calls[i % k]()
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for i in range(N): calls[i % k]()
    for c in ctxs: c.synchronize()
    dt = time.perf_counter() - t0
    print(f"{k} stream(s): {dt / N * 1e6:7.2f} us/frame, {w * h * N / dt / 1e9:6.1f} Gray/s")
