"""Debug: frame time vs frame size for the reference camera inside suzanne (needs a GPU).
Separates the per-frame fixed cost (setup kernel, launch ramp, tail) from the per-pixel cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
r = g.load_package()
suz = r.load_model_compute("suzanne_lowpoly.obj")
N = 300
with r.Context(0) as ctx:
    ctx.upload_model(suz); ctx.set_spheres(r.make_spheres(r.REFERENCE_SPHERES))
    for (w, h) in [(1920, 136), (1920, 272), (1920, 544), (1920, 1080), (1920, 2160), (3840, 2160), (3840, 4320)]:
        ctx.resize(w, h)
        ci = r.camera_build_inv_uniform(r.make_camera(eye=(0, 0, 0), aspect=16 / 9))
        call = ctx.render_call(ci, r.make_params(), (0, h))
        for _ in range(20): call()
        ctx.synchronize()
        ctx.set_kernel_timing(1)
        ctx.timer_begin()
        for _ in range(N): call()
        ms = ctx.timer_end()
        k_us, n = ctx.kernel_timing_stats()
        ctx.set_kernel_timing(0)
        wgs = ((w + 63) // 64) * ((h + 7) // 8)
        print(f"{w}x{h}: {ms / N * 1e3:8.2f} us/frame, kernel {k_us:8.2f} us, {wgs} workgroups ({wgs / 256:.1f}/CU), {w * h / (ms / N * 1e-3) / 1e9:.1f} Gray/s")
# latency floor of one wave round: scene variants at 2 workgroups per CU
empty = dict(suz, faces=suz["faces"][:0])
for label, model, spheres, eye in [("empty scene (ray generation + stores)", empty, [], (0, 0, 0)),
                                   ("suzanne from (0,0,40): all misses", suz, [], (0, 0, 40)),
                                   ("suzanne from inside", suz, [], (0, 0, 0))]:
    with r.Context(0) as ctx:
        ctx.upload_model(model); ctx.set_spheres(r.make_spheres(spheres)); ctx.resize(1920, 136)
        ci = r.camera_build_inv_uniform(r.make_camera(eye=eye, aspect=16 / 9))
        call = ctx.render_call(ci, r.make_params(), (0, 136))
        for _ in range(20): call()
        ctx.synchronize()
        ctx.set_kernel_timing(1)
        for _ in range(N): call()
        k_us, n = ctx.kernel_timing_stats()
        print(f"1920x136 {label:42s}: kernel {k_us:6.2f} us")
