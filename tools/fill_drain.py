#!/usr/bin/env python3
"""How a K-frame timed region of the headline config splits into a fixed part and a per-frame part.

For K in a few sizes: wall clock around (K frames + synchronize) and the HIP-event time of the same
region, each the median of 30 repeats; a least-squares line T = a + b*K through them gives the fixed cost
of a region (first launch's latency + the completion's way back to the host) and the steady frame time.
"""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import importlib
rwr = importlib.import_module("rust-wgpu-raytracing_amd")

def main():
    w, h = 1920, 1080
    ctx = rwr.Context(0)
    ctx.upload_model(rwr.load_model_compute("suzanne_lowpoly.obj"))
    ctx.set_spheres(rwr.make_spheres())
    ctx.resize(w, h)
    fif = int(os.environ.get("FIF", "2"))
    ctx.set_frames_in_flight(fif)
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h))
    render = ctx.render_call(cam_inv, rwr.make_params(spp=1, max_bounces=0), (0, h))
    for _ in range(200): render()
    torch.cuda.synchronize()
    rows = []
    mode = os.environ.get("MODE", "end")          # end | split | none
    timed = mode != "none"
    for K in (1, 2, 5, 10, 20, 40, 80, 160, 320):
        walls, devs, subs = [], [], []
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if timed: ctx.timer_begin()
            for _ in range(K): render()
            t1 = time.perf_counter()
            if mode == "split":
                ctx.timer_stop()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                dev = ctx.timer_elapsed()
            else:
                dev = ctx.timer_end() if timed else 0.0
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            walls.append((t2 - t0) * 1e6); devs.append(dev * 1e3); subs.append((t1 - t0) * 1e6)
        rows.append((K, statistics.median(walls), statistics.median(devs), statistics.median(subs), min(walls)))
        print(f"K={K:4d} wall {rows[-1][1]:9.1f} us  events {rows[-1][2]:9.1f} us  submit {rows[-1][3]:8.1f} us  min wall {rows[-1][4]:9.1f}", flush=True)
    ks = np.array([r[0] for r in rows if r[0] >= 10], float)
    for name, col in (("wall", 1), ("events", 2), ("submit", 3)):
        ys = np.array([r[col] for r in rows if r[0] >= 10])
        b, a = np.polyfit(ks, ys, 1)
        print(f"{name}: fixed {a:.1f} us + {b:.2f} us/frame")

if __name__ == "__main__":
    main()
