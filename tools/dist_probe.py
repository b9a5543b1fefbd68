#!/usr/bin/env python3
"""Where a gathered frame's time goes on one rank (world = 1 rehearsal): host enqueue cost per step of the render, the gather,
both; device time per step for 1..3 frames in flight.  usage: tools/dist_probe.py [cfg2|cfg5] [bands]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as graft
import bench

rwr = graft.load_package()
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
strips = not (len(sys.argv) > 2 and sys.argv[2] == "bands")
w, h = cfg["width"], cfg["height"]
model = rwr.load_model_compute(cfg["scene"])
cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, **cfg["camera"]))
params = rwr.make_params(spp=cfg["spp"], max_bounces=cfg["bounces"])
ctx = rwr.Context(0)
ctx.upload_model(model); ctx.set_spheres(rwr.make_spheres())
if cfg.get("instances"):
    ctx.set_instances(rwr.make_instance_grid(cfg["instances"], 3.0))
ctx.resize(w, h)
ctx.dist_init(0, 1, rwr.dist_get_unique_id())
render = ctx.render_call(cam_inv, params, strips=(0, 1)) if strips else ctx.render_call(cam_inv, params, rows=(0, h))
gather = ctx.dist_gather_call(0, strips=strips)
K = 300 if cfg["spp"] == 1 else 10
out = {}
for fif in (1, 2, 3):
    ctx.set_frames_in_flight(fif)
    for name, fn in (("render", lambda: render()), ("render+gather", lambda: (render(), gather()))):
        for _ in range(K // 4 + 2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out[f"fif{fif} {name}"] = {"host_enqueue_us": round((t1 - t0) / K * 1e6, 2), "us_per_step": round((t2 - t0) / K * 1e6, 2)}
print(json.dumps(out, indent=1))
ctx.dist_destroy(); ctx.close()
