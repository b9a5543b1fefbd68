"""Stress: frames in flight vs one at a time over many random cameras and sizes (needs a GPU).
Two contexts render the same camera sequence — one with 1 frame in flight, one with 2 or 3 and reads
delayed by up to the number of slots — and every frame must be identical."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
r = g.load_package()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
suz = r.load_model_compute("suzanne_lowpoly.obj")
cube = r.load_model_compute("cube.obj")
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 20.0
frames = 0
with r.Context(0) as a, r.Context(0) as b:
    while time.time() < t_end:
        model = suz if rng.random() < 0.6 else cube
        w, h = int(rng.integers(33, 700)), int(rng.integers(9, 400))
        fif = int(rng.integers(2, 4))
        for c in (a, b):
            c.upload_model(model); c.set_spheres(r.make_spheres(r.REFERENCE_SPHERES)); c.resize(w, h)
        a.set_frames_in_flight(1); b.set_frames_in_flight(fif)
        cams = [r.camera_build_inv_uniform(r.make_camera(eye=tuple(rng.uniform(-4, 4, 3)), target=tuple(rng.uniform(-1, 1, 3)), aspect=w / h))
                for _ in range(12)]
        # a mix of reference frames and path-traced ones (a slot owns a set of the integrator's accumulators and queues)
        params = [r.make_params() if rng.random() < 0.4 else
                  r.make_params(spp=int(rng.integers(1, 40)), max_bounces=int(rng.integers(0, 2)), seed=int(rng.integers(0, 1000)))
                  for _ in cams]
        want = []
        for c, pr in zip(cams, params):
            a.render(c, pr)
            want.append(a.readback())
        # b: queue a burst, read the last; then read every frame after queuing the next one(s)
        for i, c in enumerate(cams):
            b.render(c, params[i])
            if rng.random() < 0.5 or i == len(cams) - 1:
                got = b.readback()
                assert np.array_equal(got["color"], want[i]["color"]) and np.array_equal(got["depth"], want[i]["depth"]), (w, h, fif, i)
            frames += 1
print("ok:", frames, "frames compared")
