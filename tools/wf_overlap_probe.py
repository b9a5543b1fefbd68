"""Debug: would WAVEFRONT frames in flight pay?  Two contexts (own streams, own accumulators and ray queues), the same
scene; frames alternate between them, every context synchronised before its next frame (so that each learns how little the
frame shows).  If two contexts render 2 N frames much faster than one renders them one after the other, frames of the
integrator overlap usefully on the chip.  Needs a GPU.   usage: python tools/wf_overlap_probe.py [cfg4|cfg5|cfg3]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rwr = importlib.import_module("rust-wgpu-raytracing_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
cfg = bench.CONFIGS[name]
w, h = cfg["width"], cfg["height"]
cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, **cfg["camera"]))
params = rwr.make_params(spp=cfg["spp"], max_bounces=cfg["bounces"])
ctxs, calls = [], []
for _ in range(2):
    c = rwr.Context(0)
    c.upload_model(rwr.load_model_compute(cfg["scene"]))
    c.set_spheres(rwr.make_spheres())
    if cfg.get("instances"):
        c.set_instances(rwr.make_instance_grid(cfg["instances"], 3.0))
    c.resize(w, h)
    ctxs.append(c)
    calls.append(c.render_call(cam_inv, params, rows=(0, h)))
N = 40 if cfg["spp"] * w * h < 2e8 else 8
for k in (1, 2):
    for i in range(3 * k):
        calls[i % k](); ctxs[i % k].synchronize()
    t0 = time.perf_counter()
    for i in range(N * k):
        calls[i % k]()
    for c in ctxs: c.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {k} context(s): {dt / (N * k) * 1e3:.4f} ms per frame")
