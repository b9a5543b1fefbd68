#!/usr/bin/env python3
"""Renders one rank's share of a bench configuration a few times (for a kernel trace):
python tools/band_probe.py cfg3 0 135          rows [0, 135)
python tools/band_probe.py cfg5 strips 3 8     every 8th strip of 8 rows, from strip 3"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rwr = importlib.import_module("rust-wgpu-raytracing_amd")
name = sys.argv[1]
strips = (int(sys.argv[3]), int(sys.argv[4])) if sys.argv[2] == "strips" else None
rows = None if strips else (int(sys.argv[2]), int(sys.argv[3]))
cfg = bench.CONFIGS[name]
w, h = cfg["width"], cfg["height"]
ctx = rwr.Context(0)
ctx.upload_model(rwr.load_model_compute(cfg["scene"]))
ctx.set_spheres(rwr.make_spheres())
if cfg.get("instances"):
    ctx.set_instances(rwr.make_instance_grid(cfg["instances"], 3.0))
ctx.resize(w, h)
cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, **cfg["camera"]))
render = ctx.render_call(cam_inv, rwr.make_params(spp=cfg["spp"], max_bounces=cfg["bounces"]), rows=rows, strips=strips)
for _ in range(3):
    render()
    ctx.synchronize()   # (the host learns how little the frame shows from the frame before: DESIGN §4.2)
ctx.timer_begin()
for _ in range(6): render()
print(f"{name} {'strips %d + %d k' % strips if strips else 'rows [%d,%d)' % rows}: {ctx.timer_end() / 6:.4f} ms per frame")
ctx.close()
