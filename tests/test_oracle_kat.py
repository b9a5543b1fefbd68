"""Known-answer tests that pin the CPU oracle (oracle/rt_oracle.c).

The reference holds no tests or golden vectors for this path, so the
oracle is "parity unpinned" against the reference itself; what pins it here is
(1) analytic answers derived by hand from the shader source and (2) the
provisional values SURVEY.md §8(c) recorded from an independent NumPy
restatement of the same shaders.
"""
import math

import numpy as np
import pytest


def test_pod_sizes(orc, ref_loader):
    assert orc.CAMERA_INV_DTYPE.itemsize == 144      # lib.rs:86-93
    assert orc.SCREEN_DTYPE.itemsize == 8            # lib.rs:216-221
    assert orc.SPHERE_DTYPE.itemsize == 16           # sphere.rs:10-15
    assert ref_loader.VERTEX_DTYPE.itemsize == 32    # model.rs:45-51
    assert ref_loader.FACE_DTYPE.itemsize == 16      # model.rs:65-69
    assert ref_loader.MATERIAL_DTYPE.itemsize == 48  # triangle_list.rs:24-33


def test_to_non_linear_depth_end_points(orc):
    # compute.wgsl:78-80 with kNear = 0.01, kFar = 100
    assert orc.to_non_linear_depth(0.01) == pytest.approx(0.0, abs=1e-6)
    assert orc.to_non_linear_depth(100.0) == pytest.approx(1.0, abs=1e-6)
    assert orc.to_non_linear_depth(1.0) == pytest.approx((1 - 100) / (0.01 - 100), rel=1e-6)
    assert orc.to_non_linear_depth(0.005) < 0.0  # nearer than kNear: negative, still passes the depth test


def test_view_space_ray_quirk(orc):
    """proj_inv = G * P^-1 (lib.rs:109) puts the view-space ray at z = -0.5 + 0.5/zfar = -0.495."""
    cam = orc.make_camera(aspect=192 / 108)
    ci = orc.camera_build_inv_uniform(cam)
    _, d, vv = orc.pixel_to_ray(ci, orc.make_screen(192, 108), 96, 54)
    assert vv[2] == pytest.approx(-0.495, abs=1e-6)
    np.testing.assert_allclose(vv[:2], (0.00534588, 0.00534582), atol=2e-8)
    f = 1.0 / math.tan(math.radians(30.0))
    # closed form: v = (xn*aspect/f, yn/f, -0.495)
    xn = 2 * 96.5 / 192 - 1
    assert vv[0] == pytest.approx(xn * (192 / 108) / f, rel=1e-5)
    assert np.linalg.norm(d) == pytest.approx(1.0, abs=1e-6)
    # effective vertical half-FOV: atan(tan(30deg)/0.495) = 49.39 deg
    _, dtop, _ = orc.pixel_to_ray(ci, orc.make_screen(192, 108), 96, 107, 0.5, 1.0)
    assert math.degrees(math.atan2(dtop[1], -dtop[2])) == pytest.approx(49.39, abs=0.05)


def test_camera_inverse_matches_numpy(orc):
    cam = orc.make_camera(eye=(1.5, -0.7, 2.2), target=(0.1, 0.2, -0.3), up=(0, 1, 0), aspect=1.6, fovy=47.0, znear=0.3, zfar=50.0)
    ci = orc.camera_build_inv_uniform(cam)
    eye, tgt, up = np.array(cam["eye"][0], float), np.array(cam["target"][0], float), np.array(cam["up"][0], float)
    f = (tgt - eye) / np.linalg.norm(tgt - eye)
    s = np.cross(f, up); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    view = np.eye(4)
    view[0, :3], view[1, :3], view[2, :3] = s, u, -f
    view[:3, 3] = -view[:3, :3] @ eye
    ft = 1 / math.tan(math.radians(47.0) / 2)
    proj = np.array([[ft / 1.6, 0, 0, 0], [0, ft, 0, 0], [0, 0, (50 + .3) / (.3 - 50), 2 * 50 * .3 / (.3 - 50)], [0, 0, -1, 0]])
    G = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, .5, .5], [0, 0, 0, 1]])
    np.testing.assert_allclose(ci["viewmodel_inv"][0].T, np.linalg.inv(view), atol=2e-6)   # stored [col][row]
    np.testing.assert_allclose(ci["proj_inv"][0].T, G @ np.linalg.inv(proj), rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(ci["origin"][0], cam["eye"][0])


def test_single_triangle_centroid_and_edges(orc):
    p0, p1, p2 = (0, 0, -2), (1, 0, -2), (0, 1, -2)
    cen = np.array([1 / 3, 1 / 3, -2.0], np.float32)
    d = cen / np.linalg.norm(cen)
    hit, t, n, b = orc.triangle_ray_intersect(p0, p1, p2, (0, 0, 0), d)
    assert hit and t == pytest.approx(np.linalg.norm(cen), rel=1e-6)
    np.testing.assert_allclose(b, (1 / 3, 1 / 3, 1 / 3), atol=1e-6)    # weights of (p0,p1,p2)
    np.testing.assert_allclose(n, (0, 0, 1), atol=1e-7)                # flipped towards the ray (compute.wgsl:140)
    # barycentric[0] weighs p0: a ray at p0 gives (1,0,0)
    hit, _, _, b = orc.triangle_ray_intersect(p0, p1, p2, (0, 0, 0), (0, 0, -1))
    assert hit and b[0] == pytest.approx(1.0) and b[1] == 0.0 and b[2] == pytest.approx(0.0)
    # behind the origin: rejected by t < 0 (compute.wgsl:105)
    assert not orc.triangle_ray_intersect(p0, p1, p2, (0, 0, 0), (0, 0, 1))[0]
    # parallel: |N.D| < 1e-6 (compute.wgsl:94)
    assert not orc.triangle_ray_intersect(p0, p1, p2, (0, 0, 0), (1, 0, 0))[0]
    # the epsilon is scale dependent (N carries 2*area): a tiny triangle is invisible head-on
    tiny = [(0, 0, -2), (5e-4, 0, -2), (0, 5e-4, -2)]
    assert not orc.triangle_ray_intersect(*tiny, (1e-4, 1e-4, 0), (0, 0, -1))[0]


def test_shared_edge_is_inclusive_and_first_face_wins(orc, ref_loader):
    """Edges are inclusive (>= 0 passes, compute.wgsl:118-138): a ray exactly on a shared
    edge hits both faces at the same t, and the strict '<' (compute.wgsl:198) keeps the
    lower face index."""
    quad = [(-1, -1, -3), (1, -1, -3), (1, 1, -3), (-1, 1, -3)]
    for tri in ([quad[0], quad[1], quad[2]], [quad[0], quad[2], quad[3]]):
        hit, t, _, _ = orc.triangle_ray_intersect(*tri, (0, 0, 0), (0, 0, -1))   # on the diagonal
        assert hit and t == 3.0
    verts = np.zeros(4, ref_loader.VERTEX_DTYPE)
    verts["position"] = quad
    for order, expect in (([(0, 1, 2), (0, 2, 3)], 0), ([(0, 2, 3), (0, 1, 2)], 0)):
        faces = np.zeros(2, ref_loader.FACE_DTYPE)
        faces["indices"] = order
        model = {"vertices": verts, "faces": faces, "material": np.zeros(1, ref_loader.MATERIAL_DTYPE),
                 "texture": np.full((2, 2, 4), 255, np.uint8)}
        cam_inv = orc.camera_build_inv_uniform(orc.make_camera())
        out = orc.render_frame(cam_inv, orc.make_screen(4, 4), orc.make_spheres([]), model)
        # pixel centres (1.5, 1.5) and (2.5, 2.5) lie exactly on the diagonal x = y
        assert out["obj_id"][1, 1] == expect and out["obj_id"][2, 2] == expect


def test_far_clip_and_near_pass(orc, ref_loader):
    """depth >= 1 - depth_input drops hits with t >= kFar = 100 on a cleared target; t < kNear passes."""
    verts = np.zeros(3, ref_loader.VERTEX_DTYPE)
    faces = np.zeros(1, ref_loader.FACE_DTYPE)
    faces["indices"] = [(0, 1, 2)]
    mat = np.zeros(1, ref_loader.MATERIAL_DTYPE)
    mat["ambient"] = 0.25
    cam_inv = orc.camera_build_inv_uniform(orc.make_camera())
    for z, visible in ((-90.0, True), (-101.0, False), (-0.005, True)):
        s = abs(z) * 3
        verts["position"] = [(-s, -s, z), (s, -s, z), (0, s, z)]
        model = {"vertices": verts, "faces": faces, "material": mat, "texture": np.zeros((1, 1, 4), np.uint8)}
        out = orc.render_frame(cam_inv, orc.make_screen(8, 8), orc.make_spheres([]), model)
        assert (out["obj_id"][4, 4] == 0) == visible, z
    assert out["depth"][4, 4] > 1.0          # t < kNear: stored value 1 - depth exceeds 1


def test_sphere_centre_ray(orc):
    hit, t, n = orc.sphere_ray_intersect((0, 0, -5), 1.0, (0, 0, 0), (0, 0, -1))
    assert hit and t == pytest.approx(4.0) and tuple(n) == (0, 0, 1)
    hit, t, n = orc.sphere_ray_intersect((0, 0, -5), 1.0, (0, 0, -5), (0, 0, -1))   # from inside: far root
    assert hit and t == pytest.approx(1.0)
    assert not orc.sphere_ray_intersect((0, 0, 5), 1.0, (0, 0, 0), (0, 0, -1))[0]   # behind
    assert not orc.sphere_ray_intersect((3, 0, -5), 1.0, (0, 0, 0), (0, 0, -1))[0]  # miss


def test_srgb_table_and_bilinear_sampler(orc):
    lut = orc.srgb_lut()
    assert lut[0] == 0.0 and lut[255] == 1.0
    assert lut[10] == pytest.approx(10 / 255 / 12.92, rel=1e-6)
    assert lut[128] == pytest.approx(((128 / 255 + 0.055) / 1.055) ** 2.4, rel=1e-6)
    tex = np.zeros((2, 2, 4), np.uint8)
    tex[0, 0] = (255, 0, 0, 255); tex[0, 1] = (0, 255, 0, 255); tex[1, 0] = (0, 0, 255, 255); tex[1, 1] = (255, 255, 255, 255)
    np.testing.assert_allclose(orc.tex_sample(tex, 0.25, 0.25), (1, 0, 0), atol=1e-7)     # texel centre
    np.testing.assert_allclose(orc.tex_sample(tex, 0.5, 0.5), (0.5, 0.5, 0.5), atol=1e-7)  # decode-then-filter
    np.testing.assert_allclose(orc.tex_sample(tex, -3.0, 0.25), (1, 0, 0), atol=1e-7)      # ClampToEdge
    np.testing.assert_allclose(orc.tex_sample(tex, 7.0, 9.0), (1, 1, 1), atol=1e-7)


def test_unorm8_store(orc):
    L = orc.lib()
    assert [L.or_unorm8(v) for v in (-1.0, 0.0, 0.5, 1.0, 2.0, float("nan"))] == [0, 0, 128, 255, 255, 0]
    # exact ties round to the EVEN byte (the oracle's stated store rule = v_cvt_pk_u8_f32 on the GPU)
    ties = 0
    for k in range(255):
        c = np.float32(k + 0.5) / np.float32(255.0)
        if np.float32(c * np.float32(255.0)) == np.float32(k + 0.5):
            ties += 1
            assert L.or_unorm8(float(c)) == (k if k % 2 == 0 else k + 1), k
    assert ties > 20


def test_controller_s_times_15(orc):
    """circle_camera_control.rs:86-88: 15 x 'S' from the default camera puts the eye at (0,0,3)."""
    cam = orc.make_camera()
    for _ in range(15):
        cam = orc.controller_update(cam, orc.KEY_BACKWARD)
    np.testing.assert_allclose(cam["eye"][0], (0, 0, 3.0), atol=1e-6)
    # forward is refused when closer than `speed` to the target (:83)
    near = orc.make_camera(eye=(0, 0, -0.9))
    assert np.array_equal(orc.controller_update(near, orc.KEY_FORWARD)["eye"], near["eye"])
    # orbiting keeps the distance to the target (:95-104)
    right = orc.controller_update(cam, orc.KEY_RIGHT)
    assert np.linalg.norm(right["eye"][0] - right["target"][0]) == pytest.approx(4.0, abs=1e-5)
    assert right["eye"][0][0] < 0  # 'D' moves the eye along -right (target - (forward + right*speed))


# ---- provisional known answers from an independent NumPy restatement ------------
def test_survey_kat_cube_config1(orc, cube):
    cam_inv = orc.camera_build_inv_uniform(orc.make_camera(aspect=1.0))
    out = orc.render_frame(cam_inv, orc.make_screen(256, 256), orc.make_spheres(), cube)
    assert (out["obj_id"] >= 0).sum() == 65536
    assert (out["color"] == 255).all()                      # Ka = 1.0
    assert out["obj_id"][128, 128] == 427 and out["hit_t"][128, 128] == pytest.approx(1.000021, abs=2e-6)
    assert out["depth"][128, 128] == pytest.approx(0.0099007, abs=1e-6)
    assert out["obj_id"][10, 10] == 114 and out["hit_t"][10, 10] == pytest.approx(1.642745, abs=2e-6)
    assert out["obj_id"][50, 200] == 217 and out["hit_t"][50, 200] == pytest.approx(1.391098, abs=2e-6)


@pytest.fixture(scope="module")
def cfg2_small(orc, suzanne):
    # the 1920x1080 frame sub-sampled: same NDC for pixel (10x+4.5)/1920 -> use 192x108 instead
    cam_inv = orc.camera_build_inv_uniform(orc.make_camera(aspect=1920 / 1080))
    return orc.render_frame(cam_inv, orc.make_screen(1920, 1080), orc.make_spheres(), suzanne)


def test_survey_kat_suzanne_config2(cfg2_small):
    out = cfg2_small
    # 100 % hit (camera inside the mesh); the literal unfused evaluation leaves ONE crack pixel
    # on the shared edge of faces 94 / 51 at (1859, 392)
    assert (out["obj_id"] >= 0).sum() == 1920 * 1080 - 1 and out["obj_id"][392, 1859] == -1
    assert out["obj_id"][540, 960] == 40 and out["hit_t"][540, 960] == pytest.approx(0.708794, abs=1e-6)
    np.testing.assert_allclose(out["color_f32"][540, 960, :3], (0.1375972, 0.0674279, 0.0622155), atol=1e-6)
    assert out["depth"][540, 960] == pytest.approx(0.0140099, abs=1e-6)
    assert out["obj_id"][100, 100] == 95 and out["hit_t"][100, 100] == pytest.approx(0.476577, abs=1e-6)
    np.testing.assert_allclose(out["color_f32"][100, 100, :3], (0.035251, 0.017300, 0.015966), atol=1e-6)
    assert out["obj_id"][900, 1800] == 40 and out["hit_t"][900, 1800] == pytest.approx(0.884595, abs=1e-6)
    mean = np.clip(out["color_f32"][..., :3], 0, 1).reshape(-1, 3).mean(0)
    np.testing.assert_allclose(mean, (0.12034, 0.07311, 0.06953), atol=1e-5)
    assert (out["color"][..., 3] == 255).sum() == 1920 * 1080 - 1   # alpha 2.0 saturates; the crack pixel stays (0,0,0,0)


def test_survey_kat_suzanne_outside_view(orc, suzanne):
    cam_inv = orc.camera_build_inv_uniform(orc.make_camera(eye=(0, 0, 3), aspect=1920 / 1080))
    out = orc.render_frame(cam_inv, orc.make_screen(1920, 1080), orc.make_spheres(), suzanne)
    assert (out["obj_id"] >= 0).sum() == 64618               # 3.12 %
    assert not (out["obj_id"] < -1).any()                    # both spheres hidden behind the mesh
    assert out["obj_id"][540, 960] == 26 and out["hit_t"][540, 960] == pytest.approx(2.252415, abs=2e-6)
    np.testing.assert_allclose(out["color_f32"][540, 960, :3], (0.465575, 0.404066, 0.235748), atol=1e-6)
    assert out["obj_id"][600, 930] == 17 and out["hit_t"][600, 930] == pytest.approx(2.253628, abs=2e-6)
    assert out["obj_id"][300, 300] == -1 and not out["color"][300, 300].any() and out["depth"][300, 300] == 0.0
