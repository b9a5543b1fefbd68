#!/usr/bin/env python3
"""Headline benchmark: Mray/s and ms/frame, suzanne_lowpoly at 1920x1080 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks as fresh child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config loop                           (the reference's redraw loop: controller + uniform + render per frame)

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
    # the reference's redraw loop (State::update + State::render, /root/reference/src/lib.rs:994-1010,1335-1337)
    # on configs[1]'s scene: every frame CircleCameraController::update_camera (circle_camera_control.rs:76-105) with a scripted
    # key, CameraInvUniform::update_view_proj (lib.rs:105-111), rwr_render.  The keys S, W, D, A in turn (exactly periodic) wobble the camera about
    # the reference pose, so every frame has a new uniform and much the same work as cfg2.
    "loop": dict(scene="suzanne_lowpoly.obj", width=1920, height=1080, spp=1, bounces=0,
                 camera=dict(eye=(0, 0, 0), target=(0, 0, -1)), loop_keys="SWDA",
                 label="suzanne_lowpoly.obj 1920x1080 1spp, moving camera: controller update + inverse uniform + render per frame (SURVEY §8(f)1)"),
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


def spawn_ranks(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (this parent has made no
    GPU call and imports neither torch nor the library), one per GPU, with the environment torch.distributed.run would give
    them; forward rank 0's JSON line; fail if any rank fails."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    for ln in (out0 or "").splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return max(c if c > 0 else 1 for c in codes if c)
    if not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 disables)")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="frame slots (targets, per-frame records, stream, gather buffers) the context alternates between; default 2 for the "
                         "reference frame, 3 with a gather and for the wavefront configurations")
    ap.add_argument("--skip-serial", action="store_true",
                    help="leave out the one-frame-at-a-time segment after the timed region (profiling runs: every frame of the "
                         "process then runs the same schedule)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the gather path even with one rank (rehearsal on a 1-GPU box)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])   # before torch or the library are imported: the parent never touches a GPU
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
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
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
    # wavefront integrator's accumulators and ray queues, so its frames overlap the same way; and a set of gather buffers, so
    # the gather of one frame runs beside the render of the next).
    primary_only = cfg["spp"] == 1 and cfg["bounces"] == 0
    # Three with a gather: pack -> exchange -> deal-out of a frame is a longer chain than its render, and the exchange's host cost
    # (one grouped RCCL call, 20 us) sits between the launches — measured on a one-rank rehearsal: 56.8 / 49.1 / 39.2 us per frame
    # with 1 / 2 / 3 slots (profiles/r03_bench_cfg2_force_dist*.json).
    # Three for the wavefront configurations too (a frame there is a chain of a dozen launches with latency-bound ends: a third
    # frame fills more of them — configs[2] 7.62 -> 7.48 ms, configs[4]'s frame 1.74 -> 1.69, configs[3] +-0, one box).
    fif = args.frames_in_flight if args.frames_in_flight else (3 if (use_dist or not primary_only) else 2)
    ctx.set_frames_in_flight(fif)

    loop_keys = [dict(W=rwr.KEY_FORWARD, S=rwr.KEY_BACKWARD, A=rwr.KEY_LEFT, D=rwr.KEY_RIGHT)[k] for k in cfg.get("loop_keys", "")]
    loop_cam = rwr.make_camera(aspect=w / h, **cfg["camera"]) if loop_keys else None
    if loop_keys and use_dist:
        print("bench.py: --config loop is a single-GPU measurement", file=sys.stderr)
        return 2
    if loop_keys:
        render = ctx.loop_call(loop_cam, loop_keys, params)   # controller update + inverse uniform + rwr_render, every frame
    elif use_dist:
        render = ctx.render_call(cam_inv, params, strips=(rank, world))
    else:
        render = ctx.render_call(cam_inv, params, rows=(0, h))

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
    loop_stats = None
    if loop_keys:
        # the host side of a frame alone (controller update + inverse uniform, no render), and what a frame costs the host
        # thread when it only enqueues (K frames issued back to back, the device drained before and after)
        ctx.set_frames_in_flight(fif)
        host_only = ctx.loop_call(loop_cam.copy(), loop_keys, params, render=False)
        n_host = 20000
        t_h = time.perf_counter()
        for _ in range(n_host):
            host_only()
        host_us = (time.perf_counter() - t_h) / n_host * 1e6
        n_enq = max(100, min(400, args.steps))
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        for _ in range(n_enq):
            step()
        enqueue_us = (time.perf_counter() - t_h) / n_enq * 1e6
        torch.cuda.synchronize()
        loop_stats = {"host_us_per_frame_update_and_uniform": round(host_us, 3), "host_us_per_frame_enqueue_total": round(enqueue_us, 3),
                      "keys": cfg["loop_keys"], "eye_after": [round(float(v), 4) for v in loop_cam["eye"][0]],
                      "note": "host_us_per_frame_update_and_uniform = rwr_circle_controller_update + rwr_camera_build_inv_uniform through ctypes, no render; "
                              "host_us_per_frame_enqueue_total = the host thread's time per frame when it issues frames back to back without waiting "
                              "(update + uniform + rwr_render's launches); ms_per_step is the device-side frame rate with a new uniform every frame"}
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
                    "frac = VALU issue time of one launch (SQ_ACTIVE_INST_VALU x 4 cycles / SIMDs / 2400 MHz spec clock; "
                    "valu.frac_at_probe_clock uses the shader clock a probe wave measured while the frames rendered) / "
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
        # inside this process); valid only for the exact code, workload and schedule they were measured on — the file
        # records the hash of the library's sources and the schedule, and the line says when this run is something else.
        counters, counters_tree, counters_file = None, None, None
        csrc_tree = rwr.csrc_tree()
        overrides = sorted(k for k in os.environ if k.startswith("RWR_"))
        if world == 1:
            try:
                import glob
                counters_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))[-1]
                with open(counters_file) as fh:
                    doc = json.load(fh)
                counters = doc.get("cfg2" if args.config == "loop" else args.config)   # (the loop's frames wobble about cfg2's pose)
                counters_tree = doc.get("_csrc_tree")
            except (OSError, IndexError, ValueError):
                counters = None
        stale_why = []
        if counters:
            if counters_tree != csrc_tree:
                stale_why.append(f"library sources changed since the counters were collected ({counters_tree} -> {csrc_tree})")
            if overrides:
                stale_why.append("environment overrides " + ",".join(overrides))
            if counters.get("frames_in_flight", 2) != fif:
                stale_why.append(f"frames in flight {fif}, counters collected with {counters.get('frames_in_flight', 2)}")
            if args.config == "loop":
                stale_why.append("moving camera: counters are cfg2's (fixed reference pose)")
        traffic = counters.get("traffic_bytes_per_step") if counters else None
        n_simd = 4 * info["cu_count"]
        valu = None
        SPEC_MHZ = 2400.0   # MI355X_MICROARCH.md: peak engine clock
        if counters and counters.get("SQ_ACTIVE_INST_VALU") and clk:
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs (MI355X_MICROARCH.md constants table).  Issue TIME needs
            # a clock: the chip's specified 2.4 GHz gives the fraction reported as roofline.frac (reproducible from profiles/ and
            # this line alone); the clock a probe wave measured while these frames rendered gives frac_at_probe_clock beside it.
            cycles_per_simd = counters["SQ_ACTIVE_INST_VALU"] * 4.0 / n_simd
            valu_us_spec = cycles_per_simd / SPEC_MHZ
            probe_ok = bool(workload_mhz and workload_mhz > 0.0)
            valu_us_probe = cycles_per_simd / workload_mhz if probe_ok else None
            valu = {"issue_us_per_step": round(valu_us_spec, 3), "issue_us_per_step_at_probe_clock": round(valu_us_probe, 3) if probe_ok else None,
                    "SQ_ACTIVE_INST_VALU": counters["SQ_ACTIVE_INST_VALU"],
                    "SQ_INSTS_VALU": counters.get("SQ_INSTS_VALU"), "simds": n_simd, "cycles_per_simd": round(cycles_per_simd, 1),
                    "spec_mhz": SPEC_MHZ, "shader_mhz_while_rendering": round(workload_mhz, 1) if probe_ok else None,
                    "shader_mhz_under_fma_load": round(clk["shader_mhz"], 1),
                    "cycles_per_wave64_v_fma_f32": round(clk["cycles_per_v_fma_f32"], 3),
                    "cycles_per_wave64_v_pk_fma_f32": round(clk["cycles_per_v_pk_fma_f32"], 3),
                    "frac": round(valu_us_spec * 1e-6 / launch_s, 4), "frac_at_2400mhz": round(valu_us_spec * 1e-6 / launch_s, 4),
                    "frac_at_probe_clock": round(valu_us_probe * 1e-6 / launch_s, 4) if probe_ok else None,
                    "counters_from": counters.get("source"), "kernel": counters.get("dominant_kernel")}
            if counters.get("dominant_kernel") and primary_only_cfg:
                kernel = counters["dominant_kernel"]
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
            "csrc_tree": csrc_tree, "counters_tree": counters_tree, "counters_file": os.path.relpath(counters_file, ROOT) if counters_file else None,
            "counters_stale": bool(stale_why) if counters else None, "counters_stale_why": stale_why or None,
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
        if loop_stats is not None:
            out["loop"] = loop_stats
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
