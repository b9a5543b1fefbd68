"""rwr_ctx_set_frames_in_flight: frames alternate between target sets / streams; every frame must be
the frame a single-slot context renders, and the frame rendered last is the one read back."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_frames_in_flight_give_the_same_frames(rwr, gpu_ctx, suzanne):
    w, h = 200, 120
    cams = [rwr.camera_build_inv_uniform(rwr.make_camera(eye=e, aspect=w / h)) for e in ((0, 0, 0), (0, 0, 3), (1.5, 0.5, 2.5), (0, 1, 4))]
    gpu_ctx.upload_model(suzanne); gpu_ctx.set_spheres(rwr.make_spheres(rwr.REFERENCE_SPHERES)); gpu_ctx.resize(w, h)
    gpu_ctx.set_frames_in_flight(1)
    want = []
    for c in cams:
        gpu_ctx.render(c, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
        want.append(gpu_ctx.readback(aux=True))
    try:
        for n in (2, 3):
            gpu_ctx.set_frames_in_flight(n)
            # read every frame right after it was queued
            for c, ref in zip(cams, want):
                gpu_ctx.render(c, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS))
                got = gpu_ctx.readback(aux=True)
                for k in ("color", "depth", "obj_id", "hit_t", "color_f32"):
                    assert np.array_equal(got[k], ref[k]), (n, k)
            # queue them all, read the last: it is the last camera's frame
            for c in cams:
                gpu_ctx.render(c, rwr.make_params())
            got = gpu_ctx.readback()
            assert np.array_equal(got["color"], want[-1]["color"]) and np.array_equal(got["depth"], want[-1]["depth"])
            # a path-traced frame between primary frames (its slot's own accumulators and queues)
            gpu_ctx.render(cams[1], rwr.make_params())
            gpu_ctx.render(cams[0], rwr.make_params(spp=2, max_bounces=1, seed=3))
            pt = gpu_ctx.readback()
            gpu_ctx.render(cams[1], rwr.make_params())
            after = gpu_ctx.readback()
            assert np.array_equal(after["color"], want[1]["color"])
            gpu_ctx.set_frames_in_flight(1)
            gpu_ctx.render(cams[0], rwr.make_params(spp=2, max_bounces=1, seed=3))
            assert np.array_equal(gpu_ctx.readback()["color"], pt["color"])
    finally:
        gpu_ctx.set_frames_in_flight(1)
    with pytest.raises(rwr.RwrError):
        gpu_ctx.set_frames_in_flight(0)
    with pytest.raises(rwr.RwrError):
        gpu_ctx.set_frames_in_flight(4)


@pytest.mark.gpu
def test_path_traced_frames_in_flight_give_the_same_frames(rwr, suzanne):
    """A frame slot owns a whole set of the wavefront integrator's accumulators and ray queues: path-traced frames queued
    back to back in two or three slots (different cameras, sample counts, with and without a bounce, reference frames in
    between) are each the frame a single-slot context renders — the one read last, and every one when read one by one."""
    w, h = 200, 96
    seq = [((0, 0, 0), 5, 1), ((0, 0, 3), 3, 1), ((1.5, 0.5, 2.5), 1, 0), ((0, 1, 4), 4, 0), ((0.2, 0, 2.8), 40, 1), ((0, 0, 3), 2, 1),
           ((0, 0, 0), 1, 0), ((1, 1, 3), 6, 1)]
    frames = [(rwr.camera_build_inv_uniform(rwr.make_camera(eye=e, aspect=w / h)), rwr.make_params(spp=spp, max_bounces=b, seed=7, flags=rwr.FLAG_AUX_OUTPUTS))
              for e, spp, b in seq]
    with rwr.Context(0) as ctx:
        ctx.upload_model(suzanne); ctx.set_spheres(rwr.make_spheres()); ctx.resize(w, h)
        want = []
        for cam, params in frames:
            ctx.render(cam, params)
            want.append(ctx.readback(aux=True))
        for n in (2, 3):
            ctx.set_frames_in_flight(n)
            for rounds in range(2):                     # the second round reuses every slot's buffers
                for i, (cam, params) in enumerate(frames):   # read one by one
                    ctx.render(cam, params)
                    got = ctx.readback(aux=True)
                    for k in ("color", "depth", "obj_id", "hit_t", "color_f32"):
                        assert np.array_equal(got[k].view(np.uint8), want[i][k].view(np.uint8)), (n, i, k)
            for upto in (3, 5, len(frames)):            # queue several, read the last
                for cam, params in frames[:upto]:
                    ctx.render(cam, params)
                got = ctx.readback(aux=True)
                for k in ("color", "depth", "obj_id", "hit_t", "color_f32"):
                    assert np.array_equal(got[k].view(np.uint8), want[upto - 1][k].view(np.uint8)), (n, upto, k)
            ctx.resize(w + 8, h)                        # a resize in between: every slot's buffers follow
            ctx.resize(w, h)
        ctx.set_frames_in_flight(1)
        ctx.render(*frames[0])
        assert np.array_equal(ctx.readback()["color"], want[0]["color"])


@pytest.mark.gpu
def test_auto_bvh_for_small_projected_faces_is_invisible(rwr, orc, gpu_ctx, cube):
    """A binned mesh whose faces project far below a tile (cube.obj from 20 units away) is rendered by the
    per-ray BVH kernel without being asked to; the frame must be the one the tile kernels produce."""
    w, h = 320, 200
    cam = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.4, 0.3, 20.0), target=(0, 0, -1), aspect=w / h))
    gpu_ctx.upload_model(cube); gpu_ctx.set_instances(None); gpu_ctx.set_spheres(rwr.make_spheres(rwr.REFERENCE_SPHERES))
    gpu_ctx.resize(w, h)
    frames = []
    for flags in (0, rwr.FLAG_ONE_PIXEL_PER_LANE, rwr.FLAG_USE_BVH, rwr.FLAG_NO_CULL):
        gpu_ctx.render(cam, rwr.make_params(flags=flags | rwr.FLAG_AUX_OUTPUTS))
        frames.append(gpu_ctx.readback(aux=True))
    for f in frames[1:]:
        for k in ("obj_id", "hit_t", "depth", "color", "color_f32"):
            assert np.array_equal(frames[0][k], f[k]), k
    want = orc.render_frame(cam.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h),
                            rwr.make_spheres(rwr.REFERENCE_SPHERES).view(orc.SPHERE_DTYPE), cube)
    assert np.array_equal(frames[0]["obj_id"], want["obj_id"]) and (frames[0]["obj_id"] >= 0).any()
    assert np.array_equal(frames[0]["depth"].view(np.uint32), want["depth"].view(np.uint32))


@pytest.mark.gpu
def test_frame_graph_knob_gives_the_same_frames(rwr, suzanne, monkeypatch):
    """RWR_FRAME_GRAPH=1 (A/B knob, DESIGN §4.1: the reference frame's two launches as one hipGraph per slot, replayed while
    the camera stands still and updated in place when it moves) changes how a frame is launched, never its bytes."""
    w, h = 320, 180
    cams = [rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.05 * k, 0.0, 3.0 - 0.1 * k), aspect=w / h)) for k in range(4)]
    frames = {}
    for graph in ("0", "1"):
        monkeypatch.setenv("RWR_FRAME_GRAPH", graph)
        with rwr.Context(0) as ctx:
            ctx.upload_model(suzanne)
            ctx.set_spheres(rwr.make_spheres())
            ctx.resize(w, h)
            ctx.set_frames_in_flight(2)
            got = []
            for cam in cams + cams[:2] + [cams[1]] * 3:      # moving, then standing still (replay), across both slots
                ctx.render(cam, rwr.make_params())
                out = ctx.readback()
                got.append((out["color"].copy(), out["depth"].copy()))
            frames[graph] = got
    for (c0, d0), (c1, d1) in zip(frames["0"], frames["1"]):
        assert np.array_equal(c0, c1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["0", "1"])
def test_one_launch_per_frame_gives_the_same_frames(rwr, orc, suzanne, monkeypatch, fused):
    """The frame kernel's fused form (its first workgroups make the frame's records, the others wait for them: ONE launch per
    frame, the default for small frames with frames in flight) against two launches (RWR_FUSED_SETUP=0) and against the oracle:
    whole frames, strips of a multi-GPU frame, a moving camera, 1-3 frame slots."""
    w, h = 328, 181
    monkeypatch.setenv("RWR_FUSED_SETUP", fused)
    cams = [rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.07 * k, 0.02 * k, 3.0 - 0.15 * k), aspect=w / h)) for k in range(5)]
    want = orc.render_frame(cams[2].view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), suzanne)
    with rwr.Context(0) as ctx:
        ctx.upload_model(suzanne)
        ctx.set_spheres(rwr.make_spheres())
        ctx.resize(w, h)
        for fif in (1, 2, 3):
            ctx.set_frames_in_flight(fif)
            for rep in range(3):
                for k, cam in enumerate(cams):
                    ctx.render(cam, rwr.make_params())
                    if k == 2:
                        got = ctx.readback()
                        assert np.array_equal(got["depth"].view(np.uint32), want["depth"].view(np.uint32)), (fif, rep)
                        assert np.abs(got["color"].astype(int) - want["color"].astype(int)).max() <= 1
            full = ctx.readback()["color"]        # the frame of cams[4]
            asm = np.zeros_like(full)
            for r in range(3):                    # the same frame as three ranks' strips
                ctx.render(cams[4], rwr.make_params(), strips=(r, 3))
                part = ctx.readback()["color"]
                rows = [y for s in range(r, (h + 7) // 8, 3) for y in range(8 * s, min(h, 8 * s + 8))]
                asm[rows] = part[rows]
            assert np.array_equal(asm, full), fif
        ctx.synchronize()


@pytest.mark.gpu
def test_wide_per_lane_kernel_is_picked_and_gives_the_same_frame(rwr, suzanne, monkeypatch):
    """configs[3]'s scene at full size with frames in flight: from the second frame on the context traces the sparse pools with
    the per-lane kernel's WIDE form (1 024-thread workgroups sharing one LDS copy of the 508 nodelets, one work item per pool) —
    picked by itself from the previous frame's pool counts.  Same bytes as the single-slot frame and as the frame with the
    WIDE form switched off."""
    w, h = 3840, 2160
    cam = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0, 0, 12), aspect=w / h))
    params = rwr.make_params(spp=6, max_bounces=1, seed=9)
    frames = {}
    for label, env, fif in (("single", None, 1), ("auto", None, 2), ("off", "0", 2), ("forced", "1", 1)):
        if env is None:
            monkeypatch.delenv("RWR_WF_WIDE_LANE", raising=False)
        else:
            monkeypatch.setenv("RWR_WF_WIDE_LANE", env)
        with rwr.Context(0) as ctx:
            ctx.upload_model(suzanne)
            ctx.set_spheres(rwr.make_spheres())
            ctx.set_instances(rwr.make_instance_grid(4, 3.0))
            ctx.resize(w, h)
            ctx.set_frames_in_flight(fif)
            for _ in range(4):          # (the pool counts arrive a frame late: the later frames take the WIDE form)
                ctx.render(cam, params)
            out = ctx.readback()
            frames[label] = (out["color"].copy(), out["depth"].copy(), ctx.last_render_stats())
    ref = frames["single"]
    assert ref[0].any() and ref[2][1] > 100000          # a real frame with bounce rays
    for label in ("auto", "off", "forced"):
        assert np.array_equal(frames[label][0], ref[0]), label
        assert np.array_equal(frames[label][1].view(np.uint32), ref[1].view(np.uint32)), label
        assert frames[label][2] == ref[2], label
