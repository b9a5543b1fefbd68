"""ORACLE (test infrastructure only): ctypes front-end of oracle/lib/librt_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  See the header of rt_oracle.c for what is restated and for the
"parity unpinned" statement.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "librt_oracle.so")

CAMERA_INV_DTYPE = np.dtype(
    [("viewmodel_inv", "<f4", (4, 4)), ("proj_inv", "<f4", (4, 4)), ("origin", "<f4", (3,)), ("_padding", "<u4")]
)  # lib.rs:86-93, 144 B; matrices are [col][row]
SCREEN_DTYPE = np.dtype([("width", "<u4"), ("height", "<u4")])
SPHERE_DTYPE = np.dtype([("center", "<f4", (3,)), ("radius", "<f4")])
CAMERA_DTYPE = np.dtype(
    [("eye", "<f4", (3,)), ("target", "<f4", (3,)), ("up", "<f4", (3,)),
     ("aspect", "<f4"), ("fovy", "<f4"), ("znear", "<f4"), ("zfar", "<f4")]
)
assert CAMERA_INV_DTYPE.itemsize == 144 and SPHERE_DTYPE.itemsize == 16 and CAMERA_DTYPE.itemsize == 52

KEY_FORWARD, KEY_BACKWARD, KEY_LEFT, KEY_RIGHT = 1, 2, 4, 8

# reference scene constants, src/lib.rs:352-361 and 532-534
REFERENCE_SPHERES = [((0.6, 0.5, -4.0), 0.4), ((0.4, 0.4, -3.0), 0.4)]
CONTROLLER_SPEED = 0.2


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rt_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.or_to_non_linear_depth.restype = C.c_float
        _lib.or_to_non_linear_depth.argtypes = [C.c_float]
        _lib.or_unorm8.restype = C.c_uint8
        _lib.or_unorm8.argtypes = [C.c_float]
        _lib.or_render_frame.restype = C.c_int
        _lib.or_camera_build_inv_uniform.restype = C.c_int
        _lib.or_triangle_ray_intersect.restype = C.c_int
        _lib.or_sphere_ray_intersect.restype = C.c_int
        _lib.or_sphere_ray_intersect.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.or_num_threads.restype = C.c_int
        _lib.or_controller_update.argtypes = [C.c_float, C.c_uint32, C.c_void_p]
        _lib.or_tex_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        _lib.or_pixel_to_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_camera(eye=(0, 0, 0), target=(0, 0, -1), up=(0, 1, 0), aspect=1.0, fovy=60.0, znear=0.1, zfar=100.0):
    """Default values = src/lib.rs:352-360."""
    cam = np.zeros(1, dtype=CAMERA_DTYPE)
    cam["eye"], cam["target"], cam["up"] = eye, target, up
    cam["aspect"], cam["fovy"], cam["znear"], cam["zfar"] = aspect, fovy, znear, zfar
    return cam


def camera_build_inv_uniform(cam: np.ndarray) -> np.ndarray:
    out = np.zeros(1, dtype=CAMERA_INV_DTYPE)
    rc = lib().or_camera_build_inv_uniform(_p(cam), _p(out))
    if rc != 0:
        raise ValueError("singular camera matrix")
    return out


def controller_update(cam: np.ndarray, keys: int, speed: float = CONTROLLER_SPEED) -> np.ndarray:
    cam = cam.copy()
    lib().or_controller_update(speed, keys, _p(cam))
    return cam


def make_screen(w: int, h: int) -> np.ndarray:
    s = np.zeros(1, dtype=SCREEN_DTYPE)
    s["width"], s["height"] = w, h
    return s


def make_spheres(spec=REFERENCE_SPHERES) -> np.ndarray:
    s = np.zeros(len(spec), dtype=SPHERE_DTYPE)
    for i, (c, r) in enumerate(spec):
        s[i]["center"], s[i]["radius"] = c, r
    return s


def pixel_to_ray(cam_inv, screen, x, y, jx=0.5, jy=0.5):
    o = np.zeros(3, np.float32); d = np.zeros(3, np.float32); vv = np.zeros(4, np.float32)
    lib().or_pixel_to_ray(_p(cam_inv), _p(screen), x, y, jx, jy, _p(o), _p(d), _p(vv))
    return o, d, vv


def triangle_ray_intersect(p0, p1, p2, origin, direction):
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (p0, p1, p2, origin, direction)]
    t = C.c_float(0); n = np.zeros(3, np.float32); b = np.zeros(3, np.float32)
    hit = lib().or_triangle_ray_intersect(*[_p(v) for v in a], C.byref(t), _p(n), _p(b))
    return (bool(hit), t.value, n, b)


def sphere_ray_intersect(center, radius, origin, direction):
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (center, origin, direction)]
    t = C.c_float(0); n = np.zeros(3, np.float32)
    hit = lib().or_sphere_ray_intersect(_p(a[0]), radius, _p(a[1]), _p(a[2]), C.cast(C.byref(t), C.c_void_p), _p(n))
    return (bool(hit), t.value, n)


def to_non_linear_depth(t: float) -> float:
    return lib().or_to_non_linear_depth(t)


def srgb_lut() -> np.ndarray:
    lut = np.zeros(256, np.float32)
    lib().or_srgb_lut(_p(lut))
    return lut


def tex_sample(tex_rgba8: np.ndarray, u: float, v: float) -> np.ndarray:
    out = np.zeros(3, np.float32)
    h, w = tex_rgba8.shape[:2]
    lib().or_tex_sample(_p(np.ascontiguousarray(tex_rgba8)), w, h, u, v, _p(out))
    return out


def render_frame(cam_inv, screen, spheres, model, want_aux=True) -> dict:
    """State::render (lib.rs:1024-1184) on the CPU.  `model` is the dict returned by
    ref_loader.load_model_compute (or any dict with the same four arrays)."""
    w, h = int(screen["width"][0]), int(screen["height"][0])
    color = np.zeros((h, w, 4), np.uint8)
    depth = np.zeros((h, w), np.float32)
    color_f = np.zeros((h, w, 4), np.float32) if want_aux else None
    obj_id = np.zeros((h, w), np.int32) if want_aux else None
    hit_t = np.zeros((h, w), np.float32) if want_aux else None
    verts, faces = model["vertices"], model["faces"]
    tex = np.ascontiguousarray(model["texture"])
    rc = lib().or_render_frame(
        _p(cam_inv), _p(screen), _p(spheres), C.c_uint32(len(spheres)),
        _p(verts), C.c_uint32(len(verts)), _p(faces), C.c_uint32(len(faces)),
        _p(model["material"]), _p(tex), C.c_uint32(tex.shape[1]), C.c_uint32(tex.shape[0]),
        _p(color), _p(depth), _p(color_f), _p(obj_id), _p(hit_t))
    if rc != 0:
        raise MemoryError("or_render_frame")
    return {"color": color, "depth": depth, "color_f32": color_f, "obj_id": obj_id, "hit_t": hit_t}


TRIANGLE_DTYPE = np.dtype([("p0", "<f4", (3,)), ("pad0", "<f4"), ("p1", "<f4", (3,)), ("pad1", "<f4"),
                           ("p2", "<f4", (3,)), ("pad2", "<f4")])  # TriangleBufferData, models/triangle/triangle.rs:10-19


def make_triangles(tris=()) -> np.ndarray:
    out = np.zeros(len(tris), dtype=TRIANGLE_DTYPE)
    for i, (p0, p1, p2) in enumerate(tris):
        out["p0"][i], out["p1"][i], out["p2"][i] = p0, p1, p2
    return out


def render_frame_ex(cam_inv, screen, spheres, triangles, model, ortho=False) -> dict:
    """or_render_frame plus the reference's dormant parts: single-triangle passes (after the spheres) and
    pixelToRay_ortho for every pass."""
    w, h = int(screen["width"][0]), int(screen["height"][0])
    color = np.zeros((h, w, 4), np.uint8)
    depth = np.zeros((h, w), np.float32)
    color_f = np.zeros((h, w, 4), np.float32)
    obj_id = np.zeros((h, w), np.int32)
    hit_t = np.zeros((h, w), np.float32)
    verts, faces = model["vertices"], model["faces"]
    tex = np.ascontiguousarray(model["texture"])
    triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE_DTYPE)
    rc = lib().or_render_frame_ex(
        _p(cam_inv), _p(screen), _p(spheres), C.c_uint32(len(spheres)), _p(triangles), C.c_uint32(len(triangles)),
        _p(verts), C.c_uint32(len(verts)), _p(faces), C.c_uint32(len(faces)),
        _p(model["material"]), _p(tex), C.c_uint32(tex.shape[1]), C.c_uint32(tex.shape[0]), C.c_uint32(1 if ortho else 0),
        _p(color), _p(depth), _p(color_f), _p(obj_id), _p(hit_t))
    if rc != 0:
        raise MemoryError("or_render_frame_ex")
    return {"color": color, "depth": depth, "color_f32": color_f, "obj_id": obj_id, "hit_t": hit_t}


PARAMS_DTYPE = np.dtype([("spp", "<u4"), ("max_bounces", "<u4"), ("seed", "<u4"), ("flags", "<u4")])
INSTANCE_DTYPE = np.dtype([("model", "<f4", (4, 4))])


def make_params(spp=1, max_bounces=0, seed=0, flags=0) -> np.ndarray:
    p = np.zeros(1, dtype=PARAMS_DTYPE)
    p["spp"], p["max_bounces"], p["seed"], p["flags"] = spp, max_bounces, seed, flags
    return p


def rng_hash(pixel: int, sample: int, dim: int, seed: int) -> int:
    f = lib().or_rng_hash
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint32] * 4
    return f(pixel, sample, dim, seed)


def bounce_direction(n, pixel: int, sample: int, seed: int) -> np.ndarray:
    n = np.ascontiguousarray(n, dtype=np.float32)
    out = np.zeros(3, np.float32)
    f = lib().or_bounce_direction
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    f(_p(n), pixel, sample, seed, _p(out))
    return out


def concat_parts(parts: list) -> dict:
    """Flattens [part, ...] (each a model dict) into one vertex/face list with a per-face material id,
    part order then face order (the multi-material extension's face numbering)."""
    from . import ref_loader as rl
    verts = np.concatenate([p["vertices"] for p in parts]) if parts else np.zeros(0, rl.VERTEX_DTYPE)
    faces, fmat, off = [], [], 0
    for k, p in enumerate(parts):
        f = p["faces"].copy()
        f["indices"] += off
        off += len(p["vertices"])
        faces.append(f)
        fmat.append(np.full(len(f), k, np.uint32))
    return {"vertices": verts, "faces": np.concatenate(faces) if faces else np.zeros(0, rl.FACE_DTYPE),
            "face_material": np.concatenate(fmat) if fmat else np.zeros(0, np.uint32),
            "materials": np.concatenate([p["material"] for p in parts]),
            "textures": [np.ascontiguousarray(p["texture"]) for p in parts],
            "normal_textures": [None if p.get("normal_map") is None else np.ascontiguousarray(p["normal_map"]) for p in parts]}


FLAG_NORMAL_MAP = 1 << 4


def render_path(cam_inv, screen, params, spheres, model, instances=None, rows=None) -> dict:
    """The extended integrator (spp / one bounce / instances / several materials / normal maps) on the CPU, brute
    force.  `model` is one model dict or a list of them (parts); a part's optional "normal_map" (RGBA8 array, linear) is
    used when params has FLAG_NORMAL_MAP."""
    if isinstance(model, (list, tuple)):
        return _render_path_mm(cam_inv, screen, params, spheres, concat_parts(list(model)), instances, rows)
    if int(params["flags"][0]) & FLAG_NORMAL_MAP and model.get("normal_map") is not None:
        return _render_path_mm(cam_inv, screen, params, spheres, concat_parts([model]), instances, rows)
    w, h = int(screen["width"][0]), int(screen["height"][0])
    r0, r1 = rows if rows is not None else (0, h)
    color = np.zeros((h, w, 4), np.uint8)
    depth = np.zeros((h, w), np.float32)
    color_f = np.zeros((h, w, 4), np.float32)
    obj_id = np.full((h, w), -1, np.int32)
    hit_t = np.zeros((h, w), np.float32)
    verts, faces = model["vertices"], model["faces"]
    tex = np.ascontiguousarray(model["texture"])
    n_inst = 0 if instances is None else len(instances)
    inst = None if n_inst == 0 else np.ascontiguousarray(instances, dtype=INSTANCE_DTYPE)
    f = lib().or_render_path
    f.restype = C.c_int
    rc = f(_p(cam_inv), _p(screen), _p(params), _p(spheres), C.c_uint32(len(spheres)),
           _p(verts), C.c_uint32(len(verts)), _p(faces), C.c_uint32(len(faces)),
           _p(inst), C.c_uint32(n_inst), _p(model["material"]), _p(tex), C.c_uint32(tex.shape[1]), C.c_uint32(tex.shape[0]),
           C.c_uint32(r0), C.c_uint32(r1), _p(color), _p(depth), _p(color_f), _p(obj_id), _p(hit_t))
    if rc != 0:
        raise MemoryError("or_render_path")
    return {"color": color, "depth": depth, "color_f32": color_f, "obj_id": obj_id, "hit_t": hit_t}


def _render_path_mm(cam_inv, screen, params, spheres, scene, instances, rows) -> dict:
    w, h = int(screen["width"][0]), int(screen["height"][0])
    r0, r1 = rows if rows is not None else (0, h)
    color = np.zeros((h, w, 4), np.uint8)
    depth = np.zeros((h, w), np.float32)
    color_f = np.zeros((h, w, 4), np.float32)
    obj_id = np.full((h, w), -1, np.int32)
    hit_t = np.zeros((h, w), np.float32)
    texs = scene["textures"]
    n_mat = len(texs)
    ptrs = (C.c_void_p * n_mat)(*[t.ctypes.data for t in texs])
    ws = np.array([t.shape[1] for t in texs], np.uint32)
    hs = np.array([t.shape[0] for t in texs], np.uint32)
    n_inst = 0 if instances is None else len(instances)
    inst = None if n_inst == 0 else np.ascontiguousarray(instances, dtype=INSTANCE_DTYPE)
    mats = np.ascontiguousarray(scene["materials"])
    fmat = np.ascontiguousarray(scene["face_material"], dtype=np.uint32)
    nts = scene.get("normal_textures") or [None] * n_mat
    nptrs = (C.c_void_p * n_mat)(*[None if t is None else t.ctypes.data for t in nts])
    nws = np.array([0 if t is None else t.shape[1] for t in nts], np.uint32)
    nhs = np.array([0 if t is None else t.shape[0] for t in nts], np.uint32)
    f = lib().or_render_path_nm
    f.restype = C.c_int
    rc = f(_p(cam_inv), _p(screen), _p(params), _p(spheres), C.c_uint32(len(spheres)),
           _p(scene["vertices"]), C.c_uint32(len(scene["vertices"])), _p(scene["faces"]), C.c_uint32(len(scene["faces"])),
           _p(inst), C.c_uint32(n_inst), _p(mats), C.c_uint32(n_mat), _p(fmat), ptrs, _p(ws), _p(hs), nptrs, _p(nws), _p(nhs),
           C.c_uint32(r0), C.c_uint32(r1), _p(color), _p(depth), _p(color_f), _p(obj_id), _p(hit_t))
    if rc != 0:
        raise MemoryError("or_render_path_nm")
    return {"color": color, "depth": depth, "color_f32": color_f, "obj_id": obj_id, "hit_t": hit_t}


def num_threads() -> int:
    return lib().or_num_threads()


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, p = int(fq.read()), int(fp.read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, n)


def set_num_threads(n: int) -> None:
    lib().or_set_num_threads(int(n))
