"""Python driver for librwr_hip.so (tests, bench.py, smoke) — thin ctypes over the C ABI.

The product is the shared library declared in include/rwr_hip.h (HIP kernels for
gfx950 + host code in C++); this module only marshals numpy arrays across that
boundary.  There is NO CPU fallback: if the library is missing or no GPU is
visible, construction of a `Context` raises.

The directory name contains '-', so it is imported through
`__graft_entry__.load_package()` under the module name `rwr_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RWR_HIP_LIB") or os.path.join(_HERE, "lib", "librwr_hip.so")  # override: A/B runs of two builds
RES_DIR = os.path.join(_HERE, "res")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "rwr_hip.h")

# ---- PODs (layout-identical to include/rwr_hip.h and the reference's #[repr(C)] structs)
CAMERA_INV_DTYPE = np.dtype(
    [("viewmodel_inv", "<f4", (4, 4)), ("proj_inv", "<f4", (4, 4)), ("origin", "<f4", (3,)), ("_padding", "<u4")])
SCREEN_DTYPE = np.dtype([("width", "<u4"), ("height", "<u4")])
VERTEX_DTYPE = np.dtype([("position", "<f4", (3,)), ("pad0", "<f4"), ("tex_coords", "<f4", (2,)), ("pad1", "<f4", (2,))])
FACE_DTYPE = np.dtype([("indices", "<u4", (3,)), ("pad0", "<u4")])
MATERIAL_DTYPE = np.dtype([("ambient", "<f4", (3,)), ("pad0", "<f4"), ("diffuse", "<f4", (3,)), ("pad1", "<f4"),
                           ("specular", "<f4", (3,)), ("pad2", "<f4")])
SPHERE_DTYPE = np.dtype([("center", "<f4", (3,)), ("radius", "<f4")])
INSTANCE_DTYPE = np.dtype([("model", "<f4", (4, 4))])
TRIANGLE_DTYPE = np.dtype([("p0", "<f4", (3,)), ("pad0", "<f4"), ("p1", "<f4", (3,)), ("pad1", "<f4"),
                           ("p2", "<f4", (3,)), ("pad2", "<f4")])  # rwr_triangle_buffer_data, 48 B
CAMERA_DTYPE = np.dtype([("eye", "<f4", (3,)), ("target", "<f4", (3,)), ("up", "<f4", (3,)),
                         ("aspect", "<f4"), ("fovy", "<f4"), ("znear", "<f4"), ("zfar", "<f4")])
PARAMS_DTYPE = np.dtype([("spp", "<u4"), ("max_bounces", "<u4"), ("seed", "<u4"), ("flags", "<u4")])
assert (CAMERA_INV_DTYPE.itemsize, VERTEX_DTYPE.itemsize, FACE_DTYPE.itemsize, MATERIAL_DTYPE.itemsize,
        SPHERE_DTYPE.itemsize, INSTANCE_DTYPE.itemsize, CAMERA_DTYPE.itemsize) == (144, 32, 16, 48, 16, 64, 52)

FLAG_AUX_OUTPUTS, FLAG_NO_CULL, FLAG_USE_BVH = 1, 2, 4
# internal debug flags (csrc/rwr_internal.h, not part of include/rwr_hip.h)
FLAG_ORTHO_RAYS = 1 << 3
FLAG_NORMAL_MAP = 1 << 4
FLAG_DEBUG_COUNTS, FLAG_ONE_PIXEL_PER_LANE = 1 << 16, 1 << 17
KEY_FORWARD, KEY_BACKWARD, KEY_LEFT, KEY_RIGHT, KEY_UP, KEY_DOWN = 1, 2, 4, 8, 16, 32
OK, ERR_INVALID_ARGUMENT, ERR_HIP, ERR_NOT_READY, ERR_IO, ERR_PARSE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6

# reference scene literals: src/lib.rs:352-361, 532-534
REFERENCE_SPHERES = [((0.6, 0.5, -4.0), 0.4), ((0.4, 0.4, -3.0), 0.4)]
CONTROLLER_SPEED = 0.2


class RwrError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rwr error {code}: {message}")
        self.code = code
        self.message = message


def build(force: bool = False) -> str:
    """Compile librwr_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", _HERE, "-j4"] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building librwr_hip.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Loads librwr_hip.so.  torch is imported first so that the process binds ONE
    libamdhip64 (torch ships its own copy with the same SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the render path)")
    try:
        import torch  # noqa: F401  (HIP runtime provider)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int, C.c_float
    L.rwr_last_error_string.restype = C.c_char_p
    L.rwr_ctx_get_stream.restype = vp
    for name in ("rwr_model_vertices", "rwr_model_faces", "rwr_model_material", "rwr_model_texture_rgba8"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    sigs = {
        "rwr_ctx_create": [i32, vp], "rwr_ctx_destroy": [vp], "rwr_device_count": [vp],
        "rwr_ctx_device_info": [vp, vp, C.c_size_t, vp, vp], "rwr_ctx_set_stream": [vp, vp], "rwr_ctx_get_stream": [vp],
        "rwr_scene_upload_mesh": [vp, vp, u32, vp, u32, vp, vp, u32, u32],
        "rwr_scene_clear": [vp], "rwr_scene_add_mesh": [vp, vp, u32, vp, u32, vp, vp, u32, u32], "rwr_scene_commit": [vp],
        "rwr_model_part_count": [vp, vp], "rwr_model_part": [vp, u32, vp, vp, vp, vp, vp, vp, vp, vp],
        "rwr_scene_upload_model_all": [vp, vp], "rwr_model_part_normal_map": [vp, u32, vp, vp, vp],
        "rwr_scene_set_normal_map": [vp, u32, vp, u32, u32], "rwr_scene_part_count": [vp, vp],
        "rwr_scene_set_spheres": [vp, vp, u32], "rwr_scene_set_triangles": [vp, vp, u32], "rwr_scene_set_instances": [vp, vp, u32],
        "rwr_resize": [vp, vp], "rwr_render": [vp, vp, vp], "rwr_render_rows": [vp, vp, vp, u32, u32], "rwr_render_strips": [vp, vp, vp, u32, u32],
        "rwr_synchronize": [vp], "rwr_readback": [vp, vp, vp, vp, vp, vp], "rwr_get_device_targets": [vp, vp, vp],
        "rwr_timer_begin": [vp], "rwr_timer_end": [vp, vp], "rwr_timer_stop": [vp], "rwr_timer_elapsed": [vp, vp], "rwr_last_render_stats": [vp, vp, vp],
        "rwr_camera_build_inv_uniform": [vp, vp], "rwr_circle_controller_update": [f32, u32, vp],
        "rwr_load_model_compute": [C.c_char_p, C.c_char_p, vp], "rwr_model_free": [vp],
        "rwr_model_info": [vp, vp, vp, vp, vp, vp, vp], "rwr_scene_upload_model": [vp, vp],
        "rwr_decode_image_rgba8": [vp, C.c_size_t, vp, vp, vp], "rwr_free": [vp],
        "rwr_make_instance_grid": [u32, f32, vp],
        "rwr_write_png_rgba8": [C.c_char_p, vp, u32, u32, i32, i32],
        "rwr_ctx_set_kernel_timing": [vp, u32], "rwr_kernel_timing_stats": [vp, vp, vp],
        "rwr_selftest_exact_math": [vp, u32, u32, vp], "rwr_ctx_set_frames_in_flight": [vp, u32],
        "rwr_dist_get_unique_id": [vp], "rwr_dist_init": [vp, i32, i32, vp], "rwr_dist_band": [u32, u32, u32, vp, vp],
        "rwr_dist_gather_rgba8": [vp, i32], "rwr_dist_gather_strips_rgba8": [vp, i32], "rwr_dist_frame": [vp, vp], "rwr_dist_readback": [vp, vp],
        "rwr_dist_barrier": [vp], "rwr_dist_destroy": [vp],
        "rwr_dist_strip_layout": [u32, u32, u32, vp], "rwr_dist_host_pack_strips": [u32, u32, u32, u32, vp, vp],
        "rwr_dist_host_deal_strips": [u32, u32, u32, vp, vp],
        "rwr_dist_loopback_deposit": [vp, u32, u32, i32], "rwr_dist_loopback_finish": [vp, u32, i32],
        "rwr_measure_valu_clock": [vp, u32, vp], "rwr_clock_probe_start": [vp, u32], "rwr_clock_probe_read": [vp, vp],
    }
    for name, argtypes in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        if name not in ("rwr_ctx_destroy", "rwr_model_free", "rwr_free", "rwr_ctx_get_stream"):
            fn.restype = C.c_int
        elif name != "rwr_ctx_get_stream":
            fn.restype = None
    _lib = L
    return L


def _check(rc: int):
    if rc != OK:
        raise RwrError(rc, lib().rwr_last_error_string().decode("utf-8", "replace"))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ host surface --
def make_camera(eye=(0, 0, 0), target=(0, 0, -1), up=(0, 1, 0), aspect=1.0, fovy=60.0, znear=0.1, zfar=100.0):
    """Camera literal of src/lib.rs:352-360 by default."""
    cam = np.zeros(1, dtype=CAMERA_DTYPE)
    cam["eye"], cam["target"], cam["up"] = eye, target, up
    cam["aspect"], cam["fovy"], cam["znear"], cam["zfar"] = aspect, fovy, znear, zfar
    return cam


def camera_build_inv_uniform(cam: np.ndarray) -> np.ndarray:
    out = np.zeros(1, dtype=CAMERA_INV_DTYPE)
    _check(lib().rwr_camera_build_inv_uniform(_p(cam), _p(out)))
    return out


def circle_controller_update(cam: np.ndarray, keys: int, speed: float = CONTROLLER_SPEED) -> np.ndarray:
    cam = cam.copy()
    _check(lib().rwr_circle_controller_update(speed, keys, _p(cam)))
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


def make_triangles(spec=()) -> np.ndarray:
    t = np.zeros(len(spec), dtype=TRIANGLE_DTYPE)
    for i, (p0, p1, p2) in enumerate(spec):
        t[i]["p0"], t[i]["p1"], t[i]["p2"] = p0, p1, p2
    return t


def make_params(spp=1, max_bounces=0, seed=0, flags=0) -> np.ndarray:
    p = np.zeros(1, dtype=PARAMS_DTYPE)
    p["spp"], p["max_bounces"], p["seed"], p["flags"] = spp, max_bounces, seed, flags
    return p


def make_instance_grid(per_row: int, space_between: float = 3.0) -> np.ndarray:
    out = np.zeros(per_row * per_row, dtype=INSTANCE_DTYPE)
    _check(lib().rwr_make_instance_grid(per_row, space_between, _p(out)))
    return out


def decode_image_rgba8(data: bytes) -> np.ndarray:
    buf = np.frombuffer(data, dtype=np.uint8)
    out = C.c_void_p()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib().rwr_decode_image_rgba8(_p(buf), len(data), C.byref(out), C.byref(w), C.byref(h)))
    try:
        arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        lib().rwr_free(out)
    return arr


def write_png(path: str, rgba8: np.ndarray, flip_vertical: bool = True, encode_srgb: bool = False):
    rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w = rgba8.shape[:2]
    _check(lib().rwr_write_png_rgba8(path.encode(), _p(rgba8), w, h, int(flip_vertical), int(encode_srgb)))


def load_model_compute(file_name: str, res_dir: str = RES_DIR) -> dict:
    """resources::load_model_compute through the C ABI; returns numpy copies of
    meshes[0] / materials[0] (what TriangleList consumes)."""
    L = lib()
    h = C.c_void_p()
    _check(L.rwr_load_model_compute(res_dir.encode(), file_name.encode(), C.byref(h)))
    try:
        n = [C.c_uint32() for _ in range(6)]
        _check(L.rwr_model_info(h, *[C.byref(x) for x in n]))
        n_meshes, n_materials, n_verts, n_faces, tw, th = [x.value for x in n]
        verts = np.frombuffer(C.string_at(L.rwr_model_vertices(h), n_verts * 32), dtype=VERTEX_DTYPE).copy()
        faces = np.frombuffer(C.string_at(L.rwr_model_faces(h), n_faces * 16), dtype=FACE_DTYPE).copy()
        material = np.frombuffer(C.string_at(L.rwr_model_material(h), 48), dtype=MATERIAL_DTYPE).copy()
        tex = np.frombuffer(C.string_at(L.rwr_model_texture_rgba8(h), tw * th * 4), dtype=np.uint8).reshape(th, tw, 4).copy()
        nmap = _part_normal_map(L, h, 0)
    finally:
        L.rwr_model_free(h)
    return {"vertices": verts, "faces": faces, "material": material, "texture": tex, "normal_map": nmap,
            "n_meshes": n_meshes, "n_materials": n_materials}


def _part_normal_map(L, h, part):
    """The decoded map_Bump image of a part's material (extension), or None."""
    pn, nw, nh = C.c_void_p(), C.c_uint32(), C.c_uint32()
    _check(L.rwr_model_part_normal_map(h, part, C.byref(pn), C.byref(nw), C.byref(nh)))
    if not pn.value:
        return None
    return np.frombuffer(C.string_at(pn, nw.value * nh.value * 4), dtype=np.uint8).reshape(nh.value, nw.value, 4).copy()



def load_model_parts(file_name: str, res_dir: str = RES_DIR) -> list:
    """Extension: every mesh of the file with its own material, as a list of model dicts."""
    L = lib()
    h = C.c_void_p()
    _check(L.rwr_load_model_compute(res_dir.encode(), file_name.encode(), C.byref(h)))
    parts = []
    try:
        n = C.c_uint32()
        _check(L.rwr_model_part_count(h, C.byref(n)))
        for i in range(n.value):
            pv, pf, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()
            nv, nf, tw, th = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
            mat = np.zeros(1, dtype=MATERIAL_DTYPE)
            _check(L.rwr_model_part(h, i, C.byref(pv), C.byref(nv), C.byref(pf), C.byref(nf), _p(mat), C.byref(pt), C.byref(tw), C.byref(th)))
            parts.append({
                "vertices": np.frombuffer(C.string_at(pv, nv.value * 32), dtype=VERTEX_DTYPE).copy(),
                "faces": np.frombuffer(C.string_at(pf, nf.value * 16), dtype=FACE_DTYPE).copy(),
                "material": mat,
                "texture": np.frombuffer(C.string_at(pt, tw.value * th.value * 4), dtype=np.uint8).reshape(th.value, tw.value, 4).copy(),
                "normal_map": _part_normal_map(L, h, i),
            })
    finally:
        L.rwr_model_free(h)
    return parts


# ----------------------------------------------------------------- multi-GPU frames --
DIST_ID_BYTES = 128
STRIP_ROWS = 8   # RWR_STRIP_ROWS


def dist_get_unique_id() -> bytes:
    """RCCL unique id (rank 0 creates it, the launcher hands it to every rank)."""
    buf = (C.c_uint8 * DIST_ID_BYTES)()
    _check(lib().rwr_dist_get_unique_id(buf))
    return bytes(buf)


def dist_band(rank: int, world: int, height: int) -> tuple[int, int]:
    a, b = C.c_uint32(), C.c_uint32()
    _check(lib().rwr_dist_band(rank, world, height, C.byref(a), C.byref(b)))
    return a.value, b.value


class _StripLayout(C.Structure):   # rwr_strip_layout
    _fields_ = [(n, C.c_uint32) for n in ("n_strips", "strips", "rows", "recv_row", "recv_rows_total", "owns_tail")]


def dist_strip_layout(rank: int, world: int, height: int) -> dict:
    """The interleaved partition's gather layout as the library uses it (rwr_dist_strip_layout; host arithmetic, no GPU)."""
    out = _StripLayout()
    _check(lib().rwr_dist_strip_layout(rank, world, height, C.byref(out)))
    return {n: int(getattr(out, n)) for n, _ in _StripLayout._fields_}


def dist_host_pack_strips(rank: int, world: int, frame: np.ndarray) -> np.ndarray:
    """rank's message (its strips back to back) cut from a whole (H, W, 4) uint8 frame in HOST memory, by the library's layout."""
    frame = np.ascontiguousarray(frame, np.uint8)
    h, w = frame.shape[:2]
    lay = dist_strip_layout(rank, world, h)
    msg = np.zeros((max(1, lay["strips"] * STRIP_ROWS), w, 4), np.uint8)
    _check(lib().rwr_dist_host_pack_strips(rank, world, w, h, _p(frame), _p(msg)))
    return msg[:lay["rows"]]


def dist_host_deal_strips(world: int, recv: np.ndarray, height: int) -> np.ndarray:
    """The frame assembled from the root's receive buffer ((recv_rows_total, W, 4) uint8, HOST memory), by the library's layout."""
    recv = np.ascontiguousarray(recv, np.uint8)
    w = recv.shape[1]
    assert recv.shape[0] == dist_strip_layout(0, world, height)["recv_rows_total"]
    frame = np.zeros((height, w, 4), np.uint8)
    _check(lib().rwr_dist_host_deal_strips(world, w, height, _p(recv), _p(frame)))
    return frame


# ------------------------------------------------------------------------ context --
def device_count() -> int:
    n = C.c_int()
    lib().rwr_device_count(C.byref(n))
    return n.value


class Context:
    """One GPU, one HIP stream (rwr_context)."""

    def __init__(self, device_id: int = 0):
        self._h = C.c_void_p()
        _check(lib().rwr_ctx_create(device_id, C.byref(self._h)))
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None):
            lib().rwr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cu, ws = C.c_int(), C.c_int()
        _check(lib().rwr_ctx_device_info(self._h, name, 256, C.byref(cu), C.byref(ws)))
        return {"name": name.value.decode(), "cu_count": cu.value, "wave_size": ws.value}

    def set_stream(self, hip_stream: int | None):
        _check(lib().rwr_ctx_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def upload_mesh(self, vertices, faces, material, texture):
        vertices = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        faces = np.ascontiguousarray(faces, dtype=FACE_DTYPE)
        material = np.ascontiguousarray(material, dtype=MATERIAL_DTYPE)
        texture = np.ascontiguousarray(texture, dtype=np.uint8)
        th, tw = texture.shape[:2] if texture.ndim == 3 else (0, 0)
        _check(lib().rwr_scene_upload_mesh(self._h, _p(vertices), len(vertices), _p(faces), len(faces),
                                           _p(material), _p(texture), tw, th))

    def upload_model(self, model: dict):
        self.upload_mesh(model["vertices"], model["faces"], model["material"], model["texture"])
        if model.get("normal_map") is not None and len(model["faces"]):
            self.set_normal_map(0, model["normal_map"])

    def set_normal_map(self, part: int, rgba8_linear):
        """Extension: the normal map (RGBA8, linear) of scene part `part`; None removes it.  Used by FLAG_NORMAL_MAP renders."""
        if rgba8_linear is None:
            _check(lib().rwr_scene_set_normal_map(self._h, part, None, 0, 0))
            return
        t = np.ascontiguousarray(rgba8_linear, dtype=np.uint8)
        _check(lib().rwr_scene_set_normal_map(self._h, part, _p(t), t.shape[1], t.shape[0]))

    def upload_parts(self, parts: list):
        """Extension: a scene of several meshes, each with its own material and texture."""
        _check(lib().rwr_scene_clear(self._h))
        for m in parts:
            vertices = np.ascontiguousarray(m["vertices"], dtype=VERTEX_DTYPE)
            faces = np.ascontiguousarray(m["faces"], dtype=FACE_DTYPE)
            material = np.ascontiguousarray(m["material"], dtype=MATERIAL_DTYPE)
            texture = np.ascontiguousarray(m["texture"], dtype=np.uint8)
            th, tw = texture.shape[:2]
            _check(lib().rwr_scene_add_mesh(self._h, _p(vertices), len(vertices), _p(faces), len(faces), _p(material),
                                            _p(texture), tw, th))
            if m.get("normal_map") is not None and len(faces):
                n = C.c_uint32()
                _check(lib().rwr_scene_part_count(self._h, C.byref(n)))
                self.set_normal_map(n.value - 1, m["normal_map"])
        _check(lib().rwr_scene_commit(self._h))

    def set_spheres(self, spheres):
        spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        _check(lib().rwr_scene_set_spheres(self._h, _p(spheres) if len(spheres) else None, len(spheres)))

    def set_triangles(self, triangles):
        """Single-triangle passes (the reference's dormant models/triangle), after the spheres."""
        triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE_DTYPE)
        _check(lib().rwr_scene_set_triangles(self._h, _p(triangles) if len(triangles) else None, len(triangles)))

    def set_instances(self, instances):
        n = 0 if instances is None else len(instances)
        arr = None if n == 0 else np.ascontiguousarray(instances, dtype=INSTANCE_DTYPE)
        _check(lib().rwr_scene_set_instances(self._h, _p(arr), n))

    def resize(self, width: int, height: int):
        _check(lib().rwr_resize(self._h, _p(make_screen(width, height))))
        self.width, self.height = width, height

    def render(self, cam_inv, params=None, rows=None, strips=None):
        """rows = (row_begin, row_end): a band (rwr_render_rows); strips = (first_strip, strip_stride): every strip_stride-th
        8-row strip (rwr_render_strips); neither: the whole frame."""
        if strips is not None:
            _check(lib().rwr_render_strips(self._h, _p(cam_inv), _p(params), strips[0], strips[1]))
        elif rows is None:
            _check(lib().rwr_render(self._h, _p(cam_inv), _p(params)))
        else:
            _check(lib().rwr_render_rows(self._h, _p(cam_inv), _p(params), rows[0], rows[1]))

    def render_call(self, cam_inv, params, rows=None, strips=None):
        """A zero-argument callable that enqueues one frame; arguments are marshalled
        once so that the per-frame host cost is one foreign call."""
        fn, h = (lib().rwr_render_strips if strips is not None else lib().rwr_render_rows), self._h
        a, b = _p(cam_inv), _p(params)
        if strips is None and rows is None:
            rows = (0, self.height)
        r0, r1 = (C.c_uint32(strips[0]), C.c_uint32(strips[1])) if strips is not None else (C.c_uint32(rows[0]), C.c_uint32(rows[1]))

        def call(_keep=(cam_inv, params)):
            rc = fn(h, a, b, r0, r1)
            if rc:
                _check(rc)

        return call

    def loop_call(self, cam: np.ndarray, keys: list[int], params, speed: float = CONTROLLER_SPEED, render: bool = True):
        """One iteration of the reference's redraw loop (State::update + State::render, src/lib.rs:994-1010,1335-1337) per call:
        CircleCameraController::update_camera with the next key mask of `keys` (cycled), CameraInvUniform::update_view_proj,
        rwr_render.  Arguments are marshalled once; `cam` is advanced in place.  render=False: the host side alone."""
        L, h = lib(), self._h
        upd, inv, ren = L.rwr_circle_controller_update, L.rwr_camera_build_inv_uniform, L.rwr_render
        uni = np.zeros(1, dtype=CAMERA_INV_DTYPE)
        pc, pu, pp = _p(cam), _p(uni), _p(params)
        sp = C.c_float(speed)
        masks = [C.c_uint32(k) for k in keys]
        state = [0]

        def call(_keep=(cam, uni, params)):
            i = state[0]
            state[0] = i + 1 if i + 1 < len(masks) else 0
            rc = upd(sp, masks[i], pc) or inv(pc, pu) or (ren(h, pu, pp) if render else 0)
            if rc:
                _check(rc)

        return call

    def synchronize(self):
        _check(lib().rwr_synchronize(self._h))

    def readback(self, aux: bool = False) -> dict:
        h, w = self.height, self.width
        out = {"color": np.zeros((h, w, 4), np.uint8), "depth": np.zeros((h, w), np.float32)}
        if aux:
            out["color_f32"] = np.zeros((h, w, 4), np.float32)
            out["obj_id"] = np.zeros((h, w), np.int32)
            out["hit_t"] = np.zeros((h, w), np.float32)
        _check(lib().rwr_readback(self._h, _p(out["color"]), _p(out["depth"]), _p(out.get("color_f32")),
                                  _p(out.get("obj_id")), _p(out.get("hit_t"))))
        return out

    def device_targets(self) -> tuple[int, int]:
        a, b = C.c_void_p(), C.c_void_p()
        _check(lib().rwr_get_device_targets(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timer_begin(self):
        _check(lib().rwr_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float()
        _check(lib().rwr_timer_end(self._h, C.byref(ms)))
        return ms.value

    def timer_stop(self):
        """Enqueue the end event of the interval without waiting for it (rwr_timer_stop)."""
        _check(lib().rwr_timer_stop(self._h))

    def timer_elapsed(self) -> float:
        ms = C.c_float()
        _check(lib().rwr_timer_elapsed(self._h, C.byref(ms)))
        return ms.value

    def set_frames_in_flight(self, n: int):
        _check(lib().rwr_ctx_set_frames_in_flight(self._h, n))

    def set_kernel_timing(self, every_n: int):
        _check(lib().rwr_ctx_set_kernel_timing(self._h, every_n))

    def kernel_timing_stats(self) -> tuple[float, int]:
        mean, n = C.c_double(), C.c_uint32()
        _check(lib().rwr_kernel_timing_stats(self._h, C.byref(mean), C.byref(n)))
        return mean.value, n.value

    def selftest_exact_math(self, normalize_count: int = 1 << 30, seed: int = 1) -> tuple[int, int, int, int]:
        out = (C.c_uint64 * 4)()
        _check(lib().rwr_selftest_exact_math(self._h, normalize_count, seed, out))
        return tuple(int(v) for v in out)

    def measure_valu_clock(self, waves_per_simd: int = 8) -> dict:
        """Shader clock (MHz) and cycles a SIMD spends per wave64 v_fma_f32 / v_pk_fma_f32, measured under load."""
        out = (C.c_double * 4)()
        _check(lib().rwr_measure_valu_clock(self._h, waves_per_simd, out))
        return {"shader_mhz": out[0], "cycles_per_v_fma_f32": out[1], "cycles_per_v_pk_fma_f32": out[2], "shader_mhz_pk": out[3]}

    def dist_init(self, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == DIST_ID_BYTES
        buf = (C.c_uint8 * DIST_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().rwr_dist_init(self._h, rank, world, buf))

    def dist_gather(self, root: int = 0):
        """The frame's single collective: every rank's finished RGBA8 band to `root` (RCCL, stream-ordered)."""
        _check(lib().rwr_dist_gather_rgba8(self._h, root))

    def dist_gather_call(self, root: int = 0, strips: bool = False):
        """strips: the ranks rendered the interleaved partition (render(..., strips=(rank, world)))"""
        fn, h, r = (lib().rwr_dist_gather_strips_rgba8 if strips else lib().rwr_dist_gather_rgba8), self._h, C.c_int(root)

        def call():
            rc = fn(h, r)
            if rc:
                _check(rc)

        return call

    def dist_loopback_deposit(self, rank: int, world: int, strips: bool = True):
        """One-GPU self-test of the gather: `rank`'s side on the frame just rendered, a device copy in place of Send/Recv."""
        _check(lib().rwr_dist_loopback_deposit(self._h, rank, world, 1 if strips else 0))

    def dist_loopback_finish(self, world: int, strips: bool = True):
        _check(lib().rwr_dist_loopback_finish(self._h, world, 1 if strips else 0))

    def dist_readback(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.uint8)
        _check(lib().rwr_dist_readback(self._h, _p(out)))
        return out

    def dist_barrier(self):
        _check(lib().rwr_dist_barrier(self._h))

    def dist_destroy(self):
        _check(lib().rwr_dist_destroy(self._h))

    def clock_probe_start(self, micros: int):
        _check(lib().rwr_clock_probe_start(self._h, micros))

    def clock_probe_read(self) -> float:
        mhz = C.c_double()
        _check(lib().rwr_clock_probe_read(self._h, C.byref(mhz)))
        return mhz.value

    def last_render_stats(self) -> tuple[int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().rwr_last_render_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


def csrc_tree() -> str:
    """Hash of the sources librwr_hip.so is built from (csrc/, host/, include/rwr_hip.h, Makefile): ties a set of profiler
    counters (profiles/r*_counters.json `_csrc_tree`) to the code they were measured on, with or without a .git directory."""
    import hashlib

    h = hashlib.sha256()
    files = []
    for sub in ("csrc", "host"):
        d = os.path.join(_HERE, sub)
        files += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hpp", ".hip", ".cpp"))]
    files += [os.path.join(_HERE, "Makefile"), HEADER_PATH]
    for f in sorted(files):
        h.update(os.path.relpath(f, _HERE).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def exported_symbols_declared_in_header() -> list[str]:
    """Names of every RWR_API function declared in include/rwr_hip.h."""
    import re

    with open(HEADER_PATH, "r") as fh:
        text = fh.read()
    return sorted(set(re.findall(r"RWR_API\s+[^;(]*?\b(rwr_[a-z0-9_]+)\s*\(", text)))
