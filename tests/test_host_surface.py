"""CPU-only tests of the boundary and the host-side surface (no compute calls):
the C-ABI library loads and exports every symbol include/rwr_hip.h declares; PODs
match the reference's #[repr(C)] layouts; camera / controller / loader / decoders
agree with the oracle's independent restatements; error behaviour."""
import ctypes as C
import io
import os

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(rwr):
    names = rwr.exported_symbols_declared_in_header()
    assert len(names) >= 30 and "rwr_render" in names and "rwr_load_model_compute" in names
    lib = rwr.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_pod_layouts_match_reference_structs(rwr):
    assert rwr.CAMERA_INV_DTYPE.itemsize == 144 and rwr.CAMERA_INV_DTYPE.fields["origin"][1] == 128  # lib.rs:86-93
    assert rwr.VERTEX_DTYPE.itemsize == 32 and rwr.VERTEX_DTYPE.fields["tex_coords"][1] == 16         # model.rs:45-51
    assert rwr.FACE_DTYPE.itemsize == 16                                                              # model.rs:65-69
    assert rwr.MATERIAL_DTYPE.itemsize == 48 and rwr.MATERIAL_DTYPE.fields["specular"][1] == 32       # triangle_list.rs:24-33
    assert rwr.SPHERE_DTYPE.itemsize == 16 and rwr.INSTANCE_DTYPE.itemsize == 64 and rwr.SCREEN_DTYPE.itemsize == 8


def test_no_gpu_fails_loudly(rwr):
    """There is no CPU fallback: without a device, context creation reports RWR_ERR_HIP."""
    if rwr.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(rwr.RwrError) as ei:
        rwr.Context(0)
    assert ei.value.code == rwr.ERR_HIP and "HIP device" in ei.value.message


@pytest.mark.parametrize("kw", [dict(), dict(eye=(0, 0, 3)), dict(eye=(1.5, -0.7, 2.2), target=(0.1, 0.2, -0.3), fovy=47.0, znear=0.3, zfar=50.0),
                                dict(eye=(0, 5, 0.001), target=(0, 0, 0))])
def test_camera_uniform_is_bit_identical_to_oracle(rwr, orc, kw):
    cam = rwr.make_camera(aspect=1920 / 1080, **kw)
    a = rwr.camera_build_inv_uniform(cam)
    b = orc.camera_build_inv_uniform(cam.view(orc.CAMERA_DTYPE))
    assert a.tobytes() == b.tobytes()


def test_singular_camera_is_an_error_not_a_panic(rwr):
    cam = rwr.make_camera(eye=(0, 0, 0), target=(0, 1, 0), up=(0, 1, 0))   # up parallel to the view direction
    with pytest.raises(rwr.RwrError) as ei:
        rwr.camera_build_inv_uniform(cam)
    assert ei.value.code == rwr.ERR_INVALID_ARGUMENT


def test_controller_matches_oracle_over_a_key_script(rwr, orc):
    script = [rwr.KEY_BACKWARD] * 15 + [rwr.KEY_RIGHT] * 7 + [rwr.KEY_FORWARD | rwr.KEY_LEFT] * 9 + [rwr.KEY_UP, rwr.KEY_DOWN, 0] + \
             [rwr.KEY_FORWARD] * 40 + [rwr.KEY_LEFT | rwr.KEY_RIGHT] * 3
    a = rwr.make_camera()
    b = a.view(orc.CAMERA_DTYPE).copy()
    for keys in script:
        a = rwr.circle_controller_update(a, keys)
        b = orc.controller_update(b, keys & 15)
        assert a.tobytes() == b.tobytes()
    # forward stops once |target - eye| <= speed (circle_camera_control.rs:83); it may end arbitrarily close
    assert np.linalg.norm(a["eye"][0] - a["target"][0]) <= 0.2 + 1e-6


@pytest.mark.parametrize("name,nv,nf", [("suzanne_lowpoly.obj", 333, 111), ("cube.obj", 277, 428)])
def test_loader_matches_oracle_loader(rwr, ref_loader, name, nv, nf):
    a = rwr.load_model_compute(name)
    b = ref_loader.load_model_compute(rwr.RES_DIR, name)
    assert len(a["vertices"]) == nv and len(a["faces"]) == nf       # SURVEY §2 asset counts
    assert a["vertices"].tobytes() == b["vertices"].tobytes()
    assert a["faces"].tobytes() == b["faces"].tobytes()
    assert a["material"].tobytes() == b["material"].tobytes()
    assert a["n_meshes"] == 1 and a["n_materials"] == 1
    if name.endswith("suzanne_lowpoly.obj"):
        assert np.array_equal(a["texture"], b["texture"])           # PNG: lossless, exact
        np.testing.assert_allclose(a["material"]["ambient"][0], 0.01)
        np.testing.assert_allclose(a["material"]["specular"][0], 0.170455)
    else:
        d = np.abs(a["texture"].astype(int) - b["texture"].astype(int))
        assert d.max() <= 3 and (d > 1).mean() < 0.01                # JPEG: decoder dependent (parity unpinned)


def _png_bytes(arr, mode, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr, mode).save(buf, format="PNG", **kw)
    return buf.getvalue()


def test_png_decoder_variants_match_pillow(rwr):
    from PIL import Image
    rng = np.random.default_rng(7)
    cases = [(rng.integers(0, 256, (37, 53, 4), dtype=np.uint8), "RGBA"), (rng.integers(0, 256, (16, 31, 3), dtype=np.uint8), "RGB"),
             (rng.integers(0, 256, (9, 70), dtype=np.uint8), "L"), (rng.integers(0, 256, (12, 12, 2), dtype=np.uint8), "LA")]
    for arr, mode in cases:
        for level in (0, 6, 9):
            data = _png_bytes(arr, mode, compress_level=level)
            want = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
            assert np.array_equal(rwr.decode_image_rgba8(data), want), (mode, level)
    # palette + tRNS
    pal = Image.fromarray(rng.integers(0, 256, (20, 20, 3), dtype=np.uint8), "RGB").quantize(16)
    buf = io.BytesIO(); pal.save(buf, format="PNG", transparency=3)
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"))
    assert np.array_equal(rwr.decode_image_rgba8(buf.getvalue()), want)
    # 16-bit grey: image-rs narrows with (v + 128) / 257
    g16 = rng.integers(0, 65536, (8, 8), dtype=np.uint16)
    buf = io.BytesIO(); Image.fromarray(g16, "I;16").save(buf, format="PNG")
    got = rwr.decode_image_rgba8(buf.getvalue())
    assert np.array_equal(got[..., 0], ((g16.astype(np.uint32) + 128) // 257).astype(np.uint8)) and (got[..., 3] == 255).all()
    # 1-bit
    bw = (rng.integers(0, 2, (10, 13)) * 255).astype(np.uint8)
    buf = io.BytesIO(); Image.fromarray(bw, "L").convert("1").save(buf, format="PNG")
    assert np.array_equal(rwr.decode_image_rgba8(buf.getvalue())[..., 0], bw)


def test_jpeg_decoder_variants_close_to_pillow(rwr):
    from PIL import Image
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:61, 0:83]
    img = np.stack([(xx * 3) % 256, (yy * 4) % 256, ((xx + yy) * 2) % 256], -1).astype(np.uint8)
    img = (img * 0.7 + rng.integers(0, 60, img.shape)).astype(np.uint8)
    for subsampling in (0, 1, 2):          # 4:4:4, 4:2:2, 4:2:0
        buf = io.BytesIO(); Image.fromarray(img, "RGB").save(buf, format="JPEG", quality=90, subsampling=subsampling)
        want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA")).astype(int)
        got = rwr.decode_image_rgba8(buf.getvalue()).astype(int)
        assert got.shape == want.shape
        d = np.abs(got - want)
        assert d.max() <= 6 and d.mean() < 0.6, (subsampling, d.max(), d.mean())
    buf = io.BytesIO(); Image.fromarray(img[..., 0], "L").save(buf, format="JPEG", quality=85)
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA")).astype(int)
    assert np.abs(rwr.decode_image_rgba8(buf.getvalue()).astype(int) - want).max() <= 2
    # progressive JPEG is a clean error, not a crash
    buf = io.BytesIO(); Image.fromarray(img, "RGB").save(buf, format="JPEG", progressive=True)
    with pytest.raises(rwr.RwrError) as ei:
        rwr.decode_image_rgba8(buf.getvalue())
    assert ei.value.code == rwr.ERR_PARSE


def test_decoder_rejects_garbage_and_truncation(rwr):
    with pytest.raises(rwr.RwrError):
        rwr.decode_image_rgba8(b"not an image at all")
    data = open(os.path.join(rwr.RES_DIR, "suzanne_diffuse.png"), "rb").read()
    for cut in (7, 40, len(data) // 2, len(data) - 20):
        with pytest.raises(rwr.RwrError) as ei:
            rwr.decode_image_rgba8(data[:cut])
        assert ei.value.code == rwr.ERR_PARSE
    corrupt = bytearray(data); corrupt[5000] ^= 0xFF         # CRC catches it
    with pytest.raises(rwr.RwrError):
        rwr.decode_image_rgba8(bytes(corrupt))
    jpg = open(os.path.join(rwr.RES_DIR, "cube-diffuse.jpg"), "rb").read()
    with pytest.raises(rwr.RwrError):
        rwr.decode_image_rgba8(jpg[:300])


def test_decoder_bounds_hostile_inputs(rwr):
    """Untrusted files: a deflate bomb must not inflate past the image's own size, a JPEG header may not demand gigabytes,
    a truncated SOS segment may not be read past its end."""
    import struct
    import zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    # 8x8 grey PNG whose IDAT inflates to 64 MiB of zeros (stored size: a few tens of KB)
    bomb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 8, 0, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(bytes(64 << 20), 9)) + chunk(b"IEND", b"")
    assert len(bomb) < 200_000
    with pytest.raises(rwr.RwrError) as ei:
        rwr.decode_image_rgba8(bomb)
    assert ei.value.code == rwr.ERR_PARSE
    # the same header with an honest stream still decodes
    ok = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 8, 0, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(bytes(9 * 8))) + chunk(b"IEND", b"")
    assert rwr.decode_image_rgba8(ok).shape == (8, 8, 4)

    jpg = bytearray(open(os.path.join(rwr.RES_DIR, "cube-diffuse.jpg"), "rb").read())
    sof = jpg.find(b"\xff\xc0")
    assert sof > 0
    huge = bytearray(jpg)
    huge[sof + 5:sof + 9] = b"\xff\xff\xff\xff"          # 65535 x 65535
    with pytest.raises(rwr.RwrError) as ei:
        rwr.decode_image_rgba8(bytes(huge))
    assert ei.value.code == rwr.ERR_PARSE and "too large" in ei.value.message
    sos = jpg.find(b"\xff\xda")
    cut = bytes(jpg[:sos]) + b"\xff\xda\x00\x02"          # an SOS segment of length 2: no payload at all, end of file
    with pytest.raises(rwr.RwrError):
        rwr.decode_image_rgba8(cut)


def test_decoders_under_address_sanitizer(rwr, tmp_path):
    """The hand-written PNG / JPEG decoders read untrusted files: the two reference textures, truncations and 60 randomly
    damaged copies go through them in a host build with -fsanitize=address,undefined (tools/asan_codec.cpp)."""
    import random
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "asan_codec")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-I", os.path.join(root, "include"), os.path.join(root, "tools", "asan_codec.cpp"), "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("no sanitizer runtime")
    assert r.returncode == 0, r.stderr
    rnd = random.Random(7)
    files = []
    for name in ("cube-diffuse.jpg", "suzanne_diffuse.png", "cube-normal.png"):
        data = open(os.path.join(rwr.RES_DIR, name), "rb").read()
        files.append(os.path.join(rwr.RES_DIR, name))
        for k in range(20):
            b = bytearray(data)
            if k % 4 == 0:
                b = b[:rnd.randrange(1, len(b))]
            else:
                for _ in range(rnd.randrange(1, 12)):
                    b[rnd.randrange(len(b))] = rnd.randrange(256)
            path = str(tmp_path / f"{k}_{name}")
            open(path, "wb").write(bytes(b))
            files.append(path)
    r = subprocess.run([exe] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count(": ok") >= 3 and "rejected" in r.stdout


def test_loader_exposes_the_normal_map(rwr, ref_loader):
    """map_Bump (cube.mtl:13) is parsed and decoded for the normal-mapped shading extension; suzanne names none."""
    from PIL import Image
    cube = rwr.load_model_compute("cube.obj")
    want = np.asarray(Image.open(os.path.join(rwr.RES_DIR, "cube-normal.png")).convert("RGBA"), dtype=np.uint8)
    assert cube["normal_map"] is not None and np.array_equal(cube["normal_map"], want)
    assert np.array_equal(ref_loader.load_model_compute(rwr.RES_DIR, "cube.obj")["normal_map"], want)
    assert rwr.load_model_compute("suzanne_lowpoly.obj")["normal_map"] is None
    assert rwr.load_model_parts("cube.obj")[0]["normal_map"] is not None


def test_loader_error_behaviour(rwr, tmp_path):
    with pytest.raises(rwr.RwrError) as ei:                          # anyhow::Error from fs::read_to_string
        rwr.load_model_compute("does_not_exist.obj")
    assert ei.value.code == rwr.ERR_IO
    d = tmp_path
    (d / "a.mtl").write_text("newmtl m\nKa 1 1 1\nKs 0 0 0\nmap_Kd missing.png\n")
    (d / "a.obj").write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl m\nf 1/1 2/2 3/3\n")
    with pytest.raises(rwr.RwrError) as ei:                          # texture load fails -> propagated (resources.rs:189)
        rwr.load_model_compute("a.obj", str(d))
    assert ei.value.code == rwr.ERR_IO
    from PIL import Image
    Image.new("RGBA", (2, 2), (10, 20, 30, 255)).save(d / "t.png")
    (d / "b.mtl").write_text("newmtl m\nKa 0.5 0.25 0.125\nKd 1 1 1\nKs 0.1 0.2 0.3\nNs 12\nmap_Kd t.png\n")
    (d / "novt.obj").write_text("mtllib b.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 3\n")
    with pytest.raises(rwr.RwrError) as ei:                          # resources.rs:226 would index-panic
        rwr.load_model_compute("novt.obj", str(d))
    assert ei.value.code == rwr.ERR_PARSE
    (d / "bad.obj").write_text("mtllib b.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nusemtl m\nf 1/1 2/1 9/1\n")
    with pytest.raises(rwr.RwrError) as ei:
        rwr.load_model_compute("bad.obj", str(d))
    assert ei.value.code == rwr.ERR_PARSE and "out of range" in ei.value.message


def test_loader_tobj_semantics(rwr, ref_loader, tmp_path):
    """single_index dedup in first-use order, fan triangulation of a quad, negative indices,
    only the first o/g group is consumed (meshes[0])."""
    from PIL import Image
    d = tmp_path
    Image.new("RGB", (4, 4), (200, 100, 50)).save(d / "t.png")
    (d / "q.mtl").write_text("newmtl first\nKa 0.5 0.25 0.125\nKd 1 1 1\nKs 0.1 0.2 0.3\nmap_Kd t.png\nnewmtl second\nKa 9 9 9\nmap_Kd t.png\n")
    (d / "q.obj").write_text(
        "mtllib q.mtl\no quad\nv -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 1\n"
        "usemtl first\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4/1/1 -2/3/1 -1/2/1\no other\nv 5 5 5\nf 1/1/1 2/2/1 5/3/1\n")
    a = rwr.load_model_compute("q.obj", str(d))
    b = ref_loader.load_model_compute(str(d), "q.obj")
    assert a["vertices"].tobytes() == b["vertices"].tobytes() and a["faces"].tobytes() == b["faces"].tobytes()
    assert a["n_meshes"] == 2 and a["n_materials"] == 2
    assert a["faces"]["indices"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 2, 4]]   # quad fan, then (v1/vt1, v3/vt3, v4/vt2) -> one new vertex
    assert len(a["vertices"]) == 5
    np.testing.assert_allclose(a["material"]["ambient"][0], (0.5, 0.25, 0.125))


def test_instance_grid_formula(rwr):
    """lib.rs:400-421 with NUM_INSTANCES_PER_ROW = 4, SPACE_BETWEEN = 3 (the instanced configs)."""
    g = rwr.make_instance_grid(4, 3.0)["model"]
    assert g.shape == (16, 4, 4)
    pos = g[:, 3, :3]
    xs = [3.0 * (x - 2.0) for x in range(4)]
    want = np.array([(x, 0.0, z) for z in xs for x in xs], np.float32)
    assert np.array_equal(pos, want)
    ident = [i for i in range(16) if np.array_equal(pos[i], (0, 0, 0))]
    assert len(ident) == 1 and np.array_equal(g[ident[0]], np.eye(4, dtype=np.float32))
    for m in g:                                                       # rigid: R^T R = I, det = +1
        r = m[:3, :3].astype(np.float64)
        np.testing.assert_allclose(r @ r.T, np.eye(3), atol=2e-6)
        assert np.linalg.det(r) == pytest.approx(1.0, abs=1e-5)
    # 45 deg about normalize(position): the axis itself is unchanged by the rotation
    i = 0
    axis = pos[i] / np.linalg.norm(pos[i])
    np.testing.assert_allclose(g[i][:3, :3].T @ axis, axis, atol=1e-6)
    assert np.trace(g[i][:3, :3]) == pytest.approx(1 + 2 * np.cos(np.pi / 4), abs=1e-5)


def test_png_writer_roundtrip(rwr, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (33, 47, 4), dtype=np.uint8)
    path = str(tmp_path / "o.png")
    rwr.write_png(path, img, flip_vertical=True)
    back = np.asarray(Image.open(path).convert("RGBA"))
    assert np.array_equal(back, img[::-1])                           # framebuffer row 0 is the bottom row
    assert np.array_equal(rwr.decode_image_rgba8(open(path, "rb").read()), img[::-1])
    rwr.write_png(path, img, flip_vertical=False, encode_srgb=True)
    back = np.asarray(Image.open(path).convert("RGBA")).astype(float) / 255
    lin = img.astype(float) / 255
    enc = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
    np.testing.assert_allclose(back[..., :3], enc[..., :3], atol=0.5 / 255 + 1e-9)
    assert np.array_equal(back[..., 3] * 255, img[..., 3])
