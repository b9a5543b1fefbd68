#!/usr/bin/env python3
"""A/B for the reference frame's launches (reference: ONE queue.submit per frame, /root/reference/src/lib.rs:1226):
two launches per frame (k_frame_setup -> k_primary_p2: RWR_FUSED_SETUP=0) against ONE (the frame kernel's first workgroups make
the records, the others wait for them: the default).  Workloads: cfg2, cfg2b, one rank's share of either in an 8-GPU frame;
1 and 2 frames in flight; fixed and moving camera; every variant twice, same box.  Prints us per frame on the device and the
host's enqueue cost, and checks that the frames are the same bytes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as graft
import bench

rwr = graft.load_package()
K = 2000
out = []
frames_equal = True
ref_frames = {}
for graph in (0, 1):
    os.environ["RWR_FUSED_SETUP"] = str(graph)
    for name, cfgname, strips in (("cfg2", "cfg2", None), ("cfg2b", "cfg2b", None), ("cfg2 1/2 strips", "cfg2", (0, 2)), ("cfg2 1/4 strips", "cfg2", (0, 4)), ("cfg2 1/8 strips", "cfg2", (0, 8)), ("cfg2b 1/8 strips", "cfg2b", (0, 8))):
        cfg = bench.CONFIGS[cfgname]
        w, h = cfg["width"], cfg["height"]
        ctx = rwr.Context(0)
        ctx.upload_model(rwr.load_model_compute(cfg["scene"]))
        ctx.set_spheres(rwr.make_spheres())
        ctx.resize(w, h)
        params = rwr.make_params()
        cams = [rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, eye=tuple(np.add(cfg["camera"]["eye"], (0.01 * k, 0, 0))), target=cfg["camera"]["target"])) for k in range(4)]
        for fif in (1, 2):
            ctx.set_frames_in_flight(fif)
            for moving in (False, True):
                calls = [ctx.render_call(c, params, strips=strips) if strips else ctx.render_call(c, params) for c in (cams if moving else cams[:1])]
                for i in range(K // 4):
                    calls[i % len(calls)]()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.timer_begin()
                for i in range(K):
                    calls[i % len(calls)]()
                t1 = time.perf_counter()
                ms = ctx.timer_end()
                out.append(dict(workload=name, fused=graph, frames_in_flight=fif, camera="moving" if moving else "fixed",
                                us_per_frame=round(ms * 1e3 / K, 2), host_enqueue_us=round((t1 - t0) / K * 1e6, 2)))
                print(out[-1], flush=True)
        # the frames are the same bytes either way
        ctx.set_frames_in_flight(2)
        for k, c in enumerate(cams):
            ctx.render(c, params, strips=strips) if strips else ctx.render(c, params)
            got = ctx.readback()["color"]
            if (name, k) not in ref_frames:
                ref_frames[(name, k)] = got
            elif not np.array_equal(got, ref_frames[(name, k)]):
                frames_equal = False
        ctx.close()
print(json.dumps({"frames_equal_with_one_and_two_launches": frames_equal, "rows": out}))
