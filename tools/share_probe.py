#!/usr/bin/env python3
"""One rank's share of a multi-GPU frame (every N-th strip), rendered the way bench.py --gpus N renders it — several frames in
flight — without the gather: ms per frame.  usage: tools/share_probe.py cfg5 8 [frames_in_flight]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as graft
import bench
rwr = graft.load_package()
cfg = bench.CONFIGS[sys.argv[1]]
n = int(sys.argv[2])
fif = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w, h = cfg["width"], cfg["height"]
ctx = rwr.Context(0)
ctx.upload_model(rwr.load_model_compute(cfg["scene"])); ctx.set_spheres(rwr.make_spheres())
if cfg.get("instances"):
    ctx.set_instances(rwr.make_instance_grid(cfg["instances"], 3.0))
ctx.resize(w, h)
ctx.set_frames_in_flight(fif)
cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, **cfg["camera"]))
params = rwr.make_params(spp=cfg["spp"], max_bounces=cfg["bounces"])
res = []
for r in (0, n // 2):
    render = ctx.render_call(cam_inv, params, strips=(r, n))
    for _ in range(6):
        render()
    ctx.synchronize()
    K = 30
    ctx.timer_begin()
    for _ in range(K):
        render()
    res.append(ctx.timer_end() / K)
print(f"{sys.argv[1]} share 1/{n}, {fif} frames in flight: " + " ".join(f"{x:.4f}" for x in res) + " ms per frame")
ctx.close()
