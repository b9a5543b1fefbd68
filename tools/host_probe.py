"""Debug: host cost of one rwr_render call (time to enqueue, not to execute).  Needs a GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
if "--torch" in sys.argv:   # as bench.py runs: torch initialised, a torch stream current
    import torch
    torch.cuda.set_device(0); _s = torch.cuda.Stream(); torch.cuda.set_stream(_s); torch.zeros(1, device="cuda")
r = g.load_package()
suz = r.load_model_compute("suzanne_lowpoly.obj")
ci = r.camera_build_inv_uniform(r.make_camera(eye=(0, 0, 0), aspect=16 / 9))
with r.Context(0) as ctx:
    ctx.upload_model(suz); ctx.set_spheres(r.make_spheres(r.REFERENCE_SPHERES))
    for (w, h) in ((64, 64), (1920, 1080)):
        ctx.resize(w, h)
        for fif in (1, 2, 3):
            ctx.set_frames_in_flight(fif)
            call = ctx.render_call(ci, r.make_params(), (0, h))
            for _ in range(50): call()
            ctx.synchronize()
            N = 3000
            t0 = time.perf_counter()
            for _ in range(N): call()
            t1 = time.perf_counter()
            ctx.synchronize()
            t2 = time.perf_counter()
            print(f"{w}x{h} frames_in_flight={fif}: enqueue {1e6 * (t1 - t0) / N:6.2f} us/call, total {1e6 * (t2 - t0) / N:6.2f} us/frame")
    ctx.set_frames_in_flight(1)
