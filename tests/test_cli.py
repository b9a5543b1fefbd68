"""The headless host program rwr_render (C++ above the C ABI; the reference's run() loop,
/root/reference/src/lib.rs:1233-1352 with State::{new,input,update,render})."""
import os
import re
import subprocess

import numpy as np
import pytest


def _exe(rwr):
    exe = os.path.join(os.path.dirname(rwr.LIB_PATH), "..", "bin", "rwr_render")
    if not os.path.exists(exe):
        rwr.build()
    return os.path.abspath(exe)


def test_cli_help_and_argument_errors(rwr):
    exe = _exe(rwr)
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--keys" in r.stdout
    assert subprocess.run([exe], capture_output=True).returncode == 2                        # --res is required
    assert subprocess.run([exe, "--res", rwr.RES_DIR, "--keys", "Q*3"], capture_output=True).returncode == 2
    if rwr.device_count() == 0:
        r = subprocess.run([exe, "--res", rwr.RES_DIR, "--size", "32x32"], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr                           # no CPU fallback


@pytest.mark.gpu
def test_cli_scripted_camera_loop_matches_oracle(rwr, orc, suzanne, tmp_path):
    """scripted key events drive update()+render() over a frame sequence;
    the camera after the script equals the controller restatement and the last frame equals
    the oracle's frame for that camera."""
    exe = _exe(rwr)
    out = str(tmp_path / "frame.png")
    w, h = 160, 120
    r = subprocess.run([exe, "--res", rwr.RES_DIR, "--size", f"{w}x{h}", "--keys", "S*15,D*4,W*2,-*1", "--frames", "2", "--out", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = re.search(r"frames (\d+)\s+eye \(([-\d.e]+), ([-\d.e]+), ([-\d.e]+)\)", r.stdout)
    assert m and int(m.group(1)) == 15 + 4 + 2 + 1 + 2
    cam = orc.make_camera(aspect=w / h)
    for keys, n in ((orc.KEY_BACKWARD, 15), (orc.KEY_RIGHT, 4), (orc.KEY_FORWARD, 2), (0, 3)):
        for _ in range(n):
            cam = orc.controller_update(cam, keys)
    np.testing.assert_allclose([float(m.group(i)) for i in (2, 3, 4)], cam["eye"][0], atol=2e-6)
    want = orc.render_frame(orc.camera_build_inv_uniform(cam), orc.make_screen(w, h), orc.make_spheres(), suzanne)
    got = rwr.decode_image_rgba8(open(out, "rb").read()).astype(int)[::-1]                   # PNG row 0 = top = framebuffer row h-1
    lin = want["color"].astype(float) / 255
    enc = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1 / 2.4) - 0.055)
    enc[..., 3] = lin[..., 3]
    d = np.abs(got - np.rint(enc * 255))
    assert d.max() <= 3 and (d > 0).mean() < 0.01   # +-1 LSB in linear RGBA8 can move the sRGB byte by up to 3 near black


@pytest.mark.gpu
def test_cli_path_mode_and_cube(rwr, tmp_path):
    exe = _exe(rwr)
    out = str(tmp_path / "cube.png")
    r = subprocess.run([exe, "--res", rwr.RES_DIR, "--scene", "cube.obj", "--size", "96x96", "--keys", "S*20", "--spp", "4", "--bounces", "1",
                        "--out", out, "--time"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    img = rwr.decode_image_rgba8(open(out, "rb").read())
    assert img.shape == (96, 96, 4) and (img[..., 3] == 255).any() and "ms/frame" in r.stdout
    r = subprocess.run([exe, "--res", rwr.RES_DIR, "--scene", "missing.obj"], capture_output=True, text=True)
    assert r.returncode == 1 and "error -4" in r.stderr                                   # RWR_ERR_IO, not a panic


@pytest.mark.gpu
def test_cli_resize_keeps_the_old_aspect(rwr, orc, suzanne, tmp_path):
    """SURVEY §8(f) rank 1, the resize path: State::resize recomputes camera.aspect from the size BEFORE the resize
    (/root/reference/src/lib.rs:772-777: line 774 runs before 776-777), re-creates the targets at the new size and
    rewrites the screen uniform (:966-973).  After WxH -> W'xH' the frame is W'xH' pixels seen with aspect W/H; after a
    second resize the aspect is W'/H'."""
    exe = _exe(rwr)

    def run(resizes, frames):
        out = str(tmp_path / "resized.png")
        cmd = [exe, "--res", rwr.RES_DIR, "--size", "160x120", "--keys", "S*15", "--frames", str(frames), "--out", out]
        for r in resizes:
            cmd += ["--resize", r]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        m = re.search(r"size (\d+)x(\d+)\s+aspect ([-\d.e]+)", r.stdout)
        return int(m.group(1)), int(m.group(2)), float(m.group(3)), rwr.decode_image_rgba8(open(out, "rb").read()).astype(int)[::-1]

    def want(w, h, aspect):
        cam = orc.make_camera(aspect=aspect)
        for _ in range(15):
            cam = orc.controller_update(cam, orc.KEY_BACKWARD)
        ref = orc.render_frame(orc.camera_build_inv_uniform(cam), orc.make_screen(w, h), orc.make_spheres(), suzanne)
        lin = ref["color"].astype(float) / 255
        enc = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1 / 2.4) - 0.055)
        enc[..., 3] = lin[..., 3]
        return np.rint(enc * 255)

    for resizes, frames, (ew, eh, easp) in (
            (["200x100@16"], 3, (200, 100, np.float32(160) / np.float32(120))),                 # one resize: old aspect 4:3
            (["200x100@16", "96x128@17"], 4, (96, 128, np.float32(200) / np.float32(100))),    # second resize: aspect 2:1
    ):
        w, h, aspect, got = run(resizes, frames)
        assert (w, h) == (ew, eh) and abs(aspect - float(easp)) < 1e-6
        assert got.shape == (eh, ew, 4)
        d = np.abs(got - want(ew, eh, float(easp)))
        assert d.max() <= 3 and (d > 0).mean() < 0.01
        # and NOT the frame a "correct" resize would give (aspect from the new size) — the quirk is visible
        assert np.abs(got - want(ew, eh, ew / eh)).max() > 3
