#!/usr/bin/env python3
"""Rehearsal of the N-GPU row-band split on ONE GPU: renders every band of a configuration by itself and times it.
The slowest share bounds an N-GPU frame (plus the gather): contiguous row bands (rwr_render_rows) against every N-th strip of
8 rows (rwr_render_strips).
usage: python tools/band_balance.py [cfg5|cfg3|cfg4|cfg2] [N ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
rwr = importlib.import_module("rust-wgpu-raytracing_amd")

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
    worlds = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
    cfg = bench.CONFIGS[name]
    w, h = cfg["width"], cfg["height"]
    ctx = rwr.Context(0)
    ctx.upload_model(rwr.load_model_compute(cfg["scene"]))
    ctx.set_spheres(rwr.make_spheres())
    if cfg.get("instances"):
        ctx.set_instances(rwr.make_instance_grid(cfg["instances"], 3.0))
    ctx.resize(w, h)
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, **cfg["camera"]))
    params = rwr.make_params(spp=cfg["spp"], max_bounces=cfg["bounces"])
    reps = 200 if cfg["spp"] == 1 else 6
    def timed(rows=None, strips=None):
        render = ctx.render_call(cam_inv, params, rows=rows, strips=strips)
        for _ in range(3):
            render()
            ctx.synchronize()   # (the host learns how little the frame shows from the frame before: DESIGN §4.2)
        ctx.timer_begin()
        for _ in range(reps): render()
        return ctx.timer_end() / reps
    whole = timed(rows=(0, h))
    print(f"{name}: whole frame {whole:.4f} ms")
    for n in worlds:
        t = [timed(rows=rwr.dist_band(r, n, h)) for r in range(n)]
        print(f"  N={n}: contiguous bands     {' '.join(f'{x:.4f}' for x in t)} ms; slowest {max(t):.4f}, mean {sum(t)/n:.4f}; "
              f"whole / slowest = {whole / max(t):.2f} of {n}")
        t = [timed(strips=(r, n)) for r in range(n)]
        print(f"  N={n}: interleaved strips   {' '.join(f'{x:.4f}' for x in t)} ms; slowest {max(t):.4f}, mean {sum(t)/n:.4f}; "
              f"whole / slowest = {whole / max(t):.2f} of {n}")

if __name__ == "__main__":
    main()
