#!/usr/bin/env python3
"""Headline benchmark: Mray/s and ms/frame, suzanne_lowpoly at 1920x1080 (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one frame of the reference's render loop (clear + 2 sphere passes +
mesh pass, /root/reference/src/lib.rs:1024-1184) through the C ABI of
include/rwr_hip.h.  Scene, camera and targets are resident in HBM before the
timed region.  With N > 1 the frame is split into N interleaved sets of 8-row strips (one
process per GPU) and every step ends with ONE RCCL gather of the finished strips
to rank 0 (north_star), issued by the library itself (rwr_dist_gather_strips_rgba8; torch.distributed
only carries the communicator id and the barriers, over gloo) — total work is fixed, so
"scaling" is "strong".

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — dominant kernel against BOTH roofs: the binding one (VALU issue: the
                 kernel's SQ_ACTIVE_INST_VALU from the committed PMC pass x 4 cycles /
                 SIMDs / the shader clock measured live in this run) and HBM (algorithmic
                 bytes = 8 B/pixel: RGBA8 + R32F store); the duration is the
                 average time per launch of the timed region, from HIP events on the
                 launch stream(s).  No per-launch event sits inside the timed region;
  cpu_baseline — the CPU oracle (a port of the reference shaders; the reference's
                 own wgpu/llvmpipe path cannot be built here) timed on the host
                 cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3    # vector FP32 spec
ALGO_BYTES_PER_PIXEL = 8    # 4 B RGBA8 + 4 B R32F depth, each pixel stored once

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(scene="suzanne_lowpoly.obj", width=1920, height=1080, spp=1, bounces=0,
                 camera=dict(eye=(0, 0, 0), target=(0, 0, -1)),
                 label="suzanne_lowpoly.obj 1920x1080 1spp primary rays, reference camera + 2 spheres (configs[1])"),
    # the same scene from outside (15 x 'S'): 3 % of the pixels hit the mesh
    "cfg2b": dict(scene="suzanne_lowpoly.obj", width=1920, height=1080, spp=1, bounces=0,
                  camera=dict(eye=(0, 0, 3), target=(0, 0, -1)),
                  label="suzanne_lowpoly.obj 1920x1080 1spp primary rays, eye (0,0,3) + 2 spheres"),
    "cfg1": dict(scene="cube.obj", width=256, height=256, spp=1, bounces=0,
                 camera=dict(eye=(0, 0, 0), target=(0, 0, -1)),
                 label="cube.obj 256x256 1spp primary rays (configs[0])"),
    # BASELINE.json configs[2..4]: the wavefront integrator (extensions; parity cases, not the headline)
    "cfg3": dict(scene="suzanne_lowpoly.obj", width=1920, height=1080, spp=64, bounces=1, steps=20, warmup=4,
                 camera=dict(eye=(0, 0, 0), target=(0, 0, -1)),
                 label="suzanne_lowpoly.obj 1920x1080 64spp + 1 diffuse bounce, wavefront (configs[2])"),
    "cfg4": dict(scene="suzanne_lowpoly.obj", width=3840, height=2160, spp=16, bounces=1, instances=4, steps=20, warmup=4,
                 camera=dict(eye=(0, 0, 12), target=(0, 0, -1)),
                 label="suzanne_lowpoly.obj x16 instanced 3840x2160 16spp + 1 bounce, BVH (configs[3])"),
    "cfg5": dict(scene="suzanne_lowpoly.obj", width=3840, height=2160, spp=64, bounces=1, instances=4, steps=10, warmup=4,
                 camera=dict(eye=(0, 0, 12), target=(0, 0, -1)),
                 label="suzanne_lowpoly.obj x16 instanced 3840x2160 64spp + 1 bounce (configs[4])"),
}


def cpu_baseline(cfg, budget_s: float) -> dict:
    """Times the CPU oracle on whole frames of the same workload (checker code used
    as the reported CPU baseline, never as the product path)."""
    from oracle import oracle as orc, ref_loader
    import __graft_entry__ as graft

    rwr = graft.load_package()
    model = ref_loader.load_model_compute(rwr.RES_DIR, cfg["scene"])
    w, h = cfg["width"], cfg["height"]
    cam_inv = orc.camera_build_inv_uniform(orc.make_camera(aspect=w / h, **cfg["camera"]))
    screen, spheres = orc.make_screen(w, h), orc.make_spheres()
    cores = orc.usable_cores()          # affinity / cgroup share of the box, not the machine's thread count
    orc.set_num_threads(cores)
    orc.render_frame(cam_inv, screen, spheres, model, want_aux=False)  # warm-up (page in, thread pool)
    times = []
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        orc.render_frame(cam_inv, screen, spheres, model, want_aux=False)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start >= budget_s or len(times) >= 64:
            break
    times.sort()
    med = times[len(times) // 2]
    return {
        "value": round(w * h / med / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port", "host_threads_visible": os.cpu_count(),
        "ms_per_frame": round(med * 1e3, 2),
        "sample": f"{len(times)} full {w}x{h} frames of the same workload (median), brute-force per-pixel loop as in the WGSL, "
                  f"gcc -O3 -march=x86-64-v3 -ffp-contract=off + OpenMP rows; reference wgpu/llvmpipe path not buildable here (no Rust toolchain, no Vulkan ICD)",
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 disables)")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="target sets / streams the context alternates between (default: 2 for the single-GPU "
                         "frame kernel, 1 with a gather or the wavefront integrator)")
    ap.add_argument("--skip-serial", action="store_true",
                    help="leave out the one-frame-at-a-time segment after the timed region (profiling runs: every frame of the "
                         "process then runs the same schedule)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the gather path even with one rank (rehearsal on a 1-GPU box)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    # defaults: 16 us frames need ~1000 of them before the clocks and caches have settled (50: 4 % slower)
    if args.steps is None:
        args.steps = cfg.get("steps", 4000)
    if args.warmup is None:
        args.warmup = cfg.get("warmup", 1000)

    import torch
    import __graft_entry__ as graft

    rwr = graft.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the render path has no CPU fallback", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # torch.distributed is the CONTROL plane only (gloo over TCP): it carries RCCL's unique id from rank 0 to
        # the others, the barriers around the timed region and the max-over-ranks of the timings.  The frame's data
        # path — the single gather of finished bands — is RCCL called by librwr_hip.so itself (rwr_dist_*).
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    w, h = cfg["width"], cfg["height"]
    # Partition of the frame across ranks: every world-th strip of 8 rows (rwr_render_strips) — each rank gets the same share of
    # whatever part of the screen the scene covers (contiguous row bands leave six of eight ranks idle on configs[4]'s frame:
    # tools/band_balance.py).  One rank: the whole frame.
    n_strips = (h + rwr.STRIP_ROWS - 1) // rwr.STRIP_ROWS
    my_rows = sum(min(rwr.STRIP_ROWS, h - s * rwr.STRIP_ROWS) for s in range(rank, n_strips, world))
    model = rwr.load_model_compute(cfg["scene"])
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, **cfg["camera"]))
    params = rwr.make_params(spp=cfg["spp"], max_bounces=cfg["bounces"])

    ctx = rwr.Context(local_rank)
    info = ctx.device_info()
    ctx.upload_model(model)
    ctx.set_spheres(rwr.make_spheres())
    if cfg.get("instances"):
        ctx.set_instances(rwr.make_instance_grid(cfg["instances"], 3.0))   # lib.rs:400-421 grid, 3.0 apart
    ctx.resize(w, h)
    if use_dist:
        uid = torch.zeros(rwr.DIST_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            uid = torch.tensor(list(rwr.dist_get_unique_id()), dtype=torch.uint8)
        dist.broadcast(uid, src=0)
        ctx.dist_init(rank, world, bytes(uid.tolist()))   # RCCL communicator of this rank's context

    # Frames in flight: like a swapchain, the context owns two sets of targets and alternates between
    # them, so one frame's kernel ramps up while the previous frame's last waves drain (at 1080p about
    # 40 % of a lone frame kernel is its first and last waves' latency chain; a slot also owns a set of the
    # wavefront integrator's accumulators and ray queues, so its frames overlap the same way).  With a gather the root has one
    # receive buffer: 1.
    primary_only = cfg["spp"] == 1 and cfg["bounces"] == 0
    fif = args.frames_in_flight if args.frames_in_flight else (1 if use_dist else 2)
    ctx.set_frames_in_flight(fif)

    render = ctx.render_call(cam_inv, params, strips=(rank, world)) if use_dist else ctx.render_call(cam_inv, params, rows=(0, h))

    gather = ctx.dist_gather_call(0, strips=True) if use_dist else None

    def step():
        render()
        if use_dist:
            gather()   # the frame's single collective: every rank's strips to rank 0 (RCCL over xGMI), stream-ordered after the render

    def barrier():
        if use_dist:
            dist.barrier()

    # Shader clock and VALU issue cost under load, for the VALU roof (about 15 ms of arithmetic on every CU).  It runs
    # here, before the warm-up steps, like the scene upload and the BVH build: setup, not a step.
    clk = ctx.measure_valu_clock(8)

    for _ in range(args.warmup):   # (wavefront configs: at least one frame per slot, or a slot's queues are allocated inside the timed region)
        step()
    torch.cuda.synchronize()
    barrier()

    # timed region: exactly K steps.  Nothing but the frames themselves is enqueued inside it: the two
    # HIP events of timer_begin / timer_stop sit on the launch stream before the first and after the last frame.
    ctx.set_kernel_timing(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.timer_begin()            # HIP events on the launch stream(s); timer_stop joins every frame in flight
    for _ in range(args.steps):
        step()
    ctx.timer_stop()             # enqueues the end event; the one host wait of the region is the synchronize below
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = ctx.timer_elapsed()
    barrier()

    # ---- everything below is OUTSIDE the timed region --------------------------------------------
    # (a) duration of the dominant kernel's own dispatches (start/stop timestamps of the dispatch, as a
    #     profiler reports them) while frames are pipelined exactly as in the timed region
    n_probe = max(64, min(512, args.steps)) if primary_only else max(2, min(8, args.steps))
    ctx.set_kernel_timing(4 if primary_only else 1)
    probe_us = int(max(200.0, min(50000.0, 0.6 * n_probe * dev_ms * 1e3 / args.steps)))
    if rank == 0:
        ctx.clock_probe_start(probe_us)   # one idle wave on a side stream: the shader clock WHILE these frames render
    for _ in range(n_probe):
        step()
    kernel_us, kernel_samples = ctx.kernel_timing_stats()
    ctx.set_kernel_timing(0)
    workload_mhz = ctx.clock_probe_read() if rank == 0 else None
    # (b) the same frames one at a time (no overlap between frames): the latency of a frame and the
    #     duration of a lone kernel launch
    serial_ms_per_frame, serial_kernel_us = None, None
    if fif > 1 and not args.skip_serial:
        ctx.set_frames_in_flight(1)
        n_serial = max(50, min(500, args.steps // 4)) if primary_only else max(3, min(20, args.steps))
        for _ in range(10 if primary_only else 1):
            step()
        ctx.set_kernel_timing(max(1, n_serial // 64))
        ctx.timer_begin()
        for _ in range(n_serial):
            step()
        serial_ms_per_frame = ctx.timer_end() / n_serial
        serial_kernel_us, _ = ctx.kernel_timing_stats()
        ctx.set_kernel_timing(0)
    render_only_ms = None
    if use_dist:
        # outside the timed region: the same share WITHOUT the gather, so that the line shows how the frame's time
        # splits between rendering (which scales with the rank count) and the single collective (which does not)
        n_ro = max(20, min(300, args.steps // 4))
        ctx.timer_begin()
        for _ in range(n_ro):
            render()
        render_only_ms = ctx.timer_end() / n_ro
        t = torch.tensor([elapsed, dev_ms, render_only_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms, render_only_ms = float(t[0]), float(t[1]), float(t[2])

    # path segments actually traced: W*H*spp primary rays + the bounce rays the queue carried
    primary_rays, bounce_rays = ctx.last_render_stats()
    seg = torch.tensor([primary_rays + bounce_rays], dtype=torch.float64)
    if use_dist:
        dist.all_reduce(seg, op=dist.ReduceOp.SUM)
    rays_per_frame = float(seg[0])
    ms_per_step = elapsed * 1e3 / args.steps
    value = rays_per_frame / (elapsed / args.steps) / 1e6

    gathered_ok = None
    if use_dist and rank == 0:
        # the gathered frame must equal what a single GPU renders (checked once, outside the timed region)
        got = ctx.dist_readback()                        # waits for the last gather
        ctx.render(cam_inv, params)                      # the whole frame on this rank alone
        gathered_ok = bool((got == ctx.readback()["color"]).all())
    out = None
    if rank == 0:
        # Average duration of one launch of the dominant kernel(s) over the timed region: HIP events on the launch
        # stream(s) around exactly K steps / K.  With two frames in flight the dispatches overlap, so a single
        # dispatch's own start-to-stop time (launch_us_pipelined, what a kernel trace lists) is LONGER than this;
        # it cannot serve as a per-step cost and is reported for reference only.
        launch_s = dev_ms * 1e-3 / args.steps
        primary_only_cfg = cfg["spp"] == 1 and cfg["bounces"] == 0
        if primary_only_cfg:
            # dominant kernel: the fused frame kernel — one launch per step on this rank's share (plus the
            # 13-workgroup k_frame_setup that precedes it on the same stream)
            algo_bytes = ALGO_BYTES_PER_PIXEL * w * my_rows
            kernel = "k_primary_p2"
            note = ("VALU-bound: the scene (face records + 4 MiB linear-float texture) is cache resident and HBM sees only the "
                    "8 B/pixel store (RGBA8 + R32F, each pixel once), so hbm.frac is small by construction (SURVEY §8(d)); "
                    "frac = VALU issue time of one launch (SQ_ACTIVE_INST_VALU x 4 cycles / SIMDs / measured shader clock) / "
                    "duration_us; duration_us = HIP-event time of the timed region / K launches")
        else:
            # wavefront: SURVEY §8(d) contract figure, 96 B per path segment + 20 B per pixel per frame
            algo_bytes = int(96 * rays_per_frame / world + 20 * w * my_rows)
            kernel = "k_wf_primary + k_wf_sort + k_wf_trace_packet + k_wf_trace_lane + k_wf_resolve (all launches of one frame)"
            note = ("VALU-bound (shading and BVH traversal).  hbm.achieved uses SURVEY §8(d)'s contract figure, 96 B per path "
                    "segment (ray + hit record, written and read) + 20 B per pixel, which this design deliberately does not move "
                    "(primary rays and hit records stay in registers) — so that fraction can exceed 1 and is kept only for "
                    "comparability; hbm.design_bytes_per_step is what THIS design's algorithm moves (per bounce ray: 32-B record "
                    "written and read, 2-B direction bin and 2-B sorted slot written and read; per pixel: 4 u64 sums added to by the "
                    "primary stage and 3 by the trace kernels in every launch group, all 4 read + cleared by the resolve, 4 B RGBA8 — "
                    "an upper bound: tiles that see nothing are never touched), and traffic is what the counters saw")
        # Counters of the dominant kernel(s) per step from the committed PMC passes (they cannot be collected live
        # inside this process); valid only for the exact workload they were measured on.
        counters = None
        if world == 1:
            try:
                import glob
                with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))[-1]) as fh:
                    counters = json.load(fh).get(args.config)
            except (OSError, IndexError, ValueError):
                counters = None
        traffic = counters.get("traffic_bytes_per_step") if counters else None
        n_simd = 4 * info["cu_count"]
        valu = None
        if counters and counters.get("SQ_ACTIVE_INST_VALU") and clk:
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs (MI355X_MICROARCH.md constants table)
            valu_us = counters["SQ_ACTIVE_INST_VALU"] * 4.0 / n_simd / workload_mhz
            valu = {"issue_us_per_step": round(valu_us, 3), "SQ_ACTIVE_INST_VALU": counters["SQ_ACTIVE_INST_VALU"],
                    "SQ_INSTS_VALU": counters.get("SQ_INSTS_VALU"), "simds": n_simd,
                    "shader_mhz_while_rendering": round(workload_mhz, 1), "shader_mhz_under_fma_load": round(clk["shader_mhz"], 1),
                    "cycles_per_wave64_v_fma_f32": round(clk["cycles_per_v_fma_f32"], 3),
                    "cycles_per_wave64_v_pk_fma_f32": round(clk["cycles_per_v_pk_fma_f32"], 3),
                    "frac": round(valu_us * 1e-6 / launch_s, 4), "counters_from": counters.get("source")}
        hbm_achieved = algo_bytes / launch_s / 1e9
        hbm = {"achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 5),
               "algorithmic_bytes_per_step": algo_bytes}
        if not primary_only_cfg:
            n_groups = -(-cfg["spp"] // (64 if fif > 1 else 32))   # launch groups: 64 samples with frames in flight, else 32
            design = int(float(bounce_rays) * (2 * 32 + 2 * 2 + 2 * 2) + w * my_rows * (n_groups * 7 * 16 + 4 * 16 + 4))
            hbm.update({"design_bytes_per_step": design, "design_frac": round(design / launch_s / 1e9 / HBM_PEAK_GBS, 5),
                        "traffic_frac": round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 5) if traffic else None})
        if valu:
            roofline = {"bound": "valu", "kernel": kernel, "achieved": valu["issue_us_per_step"], "peak": round(launch_s * 1e6, 3),
                        "unit": "us of VALU issue per step / us per step", "frac": valu["frac"], "traffic": traffic}
        else:   # no counter record for this workload: only the HBM figure can be stated
            roofline = {"bound": "hbm", "kernel": kernel, "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": hbm["frac"], "traffic": traffic}
        roofline.update({
            "valu": valu, "hbm": hbm, "valu_frac": valu["frac"] if valu else None, "hbm_frac": hbm["frac"],
            "duration_us": round(launch_s * 1e6, 3), "launch_us_pipelined": round(kernel_us, 3) if kernel_samples else None,
            "launch_us_pipelined_samples": kernel_samples, "timed_region_instrumented": False, "note": note,
        })
        if serial_kernel_us:
            roofline["launch_us_serial"] = round(serial_kernel_us, 3)
        out = {
            "metric": "Mray/s", "value": round(value, 2), "unit": "Mray/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "ms_per_frame": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["label"], "width": w, "height": h, "spp": cfg["spp"], "bounces": cfg["bounces"],
                       "faces": int(len(model["faces"])), "partition": f"{world} interleaved sets of 8-row strips + 1 RCCL gather" if world > 1 else "single GPU",
                       "frames_in_flight": fif, "device": info["name"]},
            "roofline": roofline,
        }
        if serial_ms_per_frame is not None:
            out["ms_per_frame_one_in_flight"] = round(serial_ms_per_frame, 5)   # frame latency: each frame waits for the previous
        if render_only_ms is not None:
            out["ms_per_frame_render_only"] = round(render_only_ms, 5)   # slowest rank's share, no gather (not timed above)
        if gathered_ok is not None:
            out["config"]["gathered_frame_ok"] = gathered_ok
    if use_dist:
        ctx.dist_destroy()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
