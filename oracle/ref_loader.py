"""ORACLE (test infrastructure only): pure-Python restatement of the asset path.

Follows /root/reference/src/resources.rs:163-264 (`load_model_compute`) and the
behaviour of its un-vendored dependencies it relies on:

* tobj 3.2.5 `load_obj_buf(triangulate=true, single_index=true)` + `load_mtl_buf`
  (call site resources.rs:173-185): one output vertex per unique `v/vt/vn`
  index triple in first-use order, faces in file order, polygons fan-
  triangulated, a new model at every `o`/`g`/`usemtl` that follows faces.
* image 0.24.6 `load_from_memory(..).to_rgba8()` (texture.rs:104,114-115):
  decoded here with Pillow.  PNG is lossless, so any decoder gives the same
  bytes; JPEG decoders differ by +-1-2 LSB (parity unpinned there).

Nothing here is imported by the product package.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

VERTEX_DTYPE = np.dtype(
    [("position", "<f4", (3,)), ("pad0", "<f4"), ("tex_coords", "<f4", (2,)), ("pad1", "<f4", (2,))]
)  # model.rs:45-63, 32 B
FACE_DTYPE = np.dtype([("indices", "<u4", (3,)), ("pad0", "<u4")])  # model.rs:65-79, 16 B
MATERIAL_DTYPE = np.dtype(
    [("ambient", "<f4", (3,)), ("pad0", "<f4"), ("diffuse", "<f4", (3,)), ("pad1", "<f4"),
     ("specular", "<f4", (3,)), ("pad2", "<f4")]
)  # triangle_list.rs:24-33, 48 B

assert VERTEX_DTYPE.itemsize == 32 and FACE_DTYPE.itemsize == 16 and MATERIAL_DTYPE.itemsize == 48


@dataclass
class RefMaterial:
    name: str = ""
    ambient: tuple = (0.0, 0.0, 0.0)
    diffuse: tuple = (0.0, 0.0, 0.0)
    specular: tuple = (0.0, 0.0, 0.0)
    shininess: float = 0.0
    diffuse_texture: str = ""
    normal_texture: str = ""


@dataclass
class RefMesh:
    name: str = ""
    positions: list = field(default_factory=list)  # flat xyz
    texcoords: list = field(default_factory=list)  # flat uv
    normals: list = field(default_factory=list)
    indices: list = field(default_factory=list)
    material_id: int | None = None


def parse_mtl(text: str) -> list[RefMaterial]:
    mats: list[RefMaterial] = []
    cur: RefMaterial | None = None
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        parts = line.split()
        key, args = parts[0], parts[1:]
        if key == "newmtl":
            cur = RefMaterial(name=" ".join(args))
            mats.append(cur)
        elif cur is None:
            continue
        elif key == "Ka":
            cur.ambient = tuple(float(a) for a in args[:3])
        elif key == "Kd":
            cur.diffuse = tuple(float(a) for a in args[:3])
        elif key == "Ks":
            cur.specular = tuple(float(a) for a in args[:3])
        elif key == "Ns":
            cur.shininess = float(args[0])
        elif key == "map_Kd":
            cur.diffuse_texture = args[-1]
        elif key in ("map_Bump", "map_bump", "bump"):
            cur.normal_texture = args[-1]
    return mats


def _fix_index(tok: str, count: int) -> int:
    i = int(tok)
    return i - 1 if i > 0 else count + i


def parse_obj(text: str):
    """Returns (list[RefMesh], list[str] mtllibs, list[str] usemtl name per mesh)."""
    pos: list[float] = []
    tex: list[float] = []
    nor: list[float] = []
    meshes: list[RefMesh] = []
    mtllibs: list[str] = []
    mesh_mtl: list[str | None] = []

    cur_name = "unnamed_object"
    cur_mtl: str | None = None
    cur_faces: list[list[tuple[int, int, int]]] = []

    def flush():
        nonlocal cur_faces
        if not cur_faces:
            return
        mesh = RefMesh(name=cur_name)
        index_map: dict[tuple[int, int, int], int] = {}
        for face in cur_faces:
            tris = [(face[0], face[i], face[i + 1]) for i in range(1, len(face) - 1)]
            for tri in tris:
                for vert in tri:
                    hit = index_map.get(vert)
                    if hit is not None:
                        mesh.indices.append(hit)
                        continue
                    v, vt, vn = vert
                    mesh.positions.extend(pos[3 * v:3 * v + 3])
                    if tex and vt >= 0:
                        mesh.texcoords.extend(tex[2 * vt:2 * vt + 2])
                    if nor and vn >= 0:
                        mesh.normals.extend(nor[3 * vn:3 * vn + 3])
                    nxt = len(index_map)
                    mesh.indices.append(nxt)
                    index_map[vert] = nxt
        meshes.append(mesh)
        mesh_mtl.append(cur_mtl)
        cur_faces = []

    for raw in text.splitlines():
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        parts = line.split()
        key, args = parts[0], parts[1:]
        if key == "v":
            pos.extend(float(a) for a in args[:3])
        elif key == "vt":
            tex.extend([float(args[0]), float(args[1]) if len(args) > 1 else 0.0])
        elif key == "vn":
            nor.extend(float(a) for a in args[:3])
        elif key == "f":
            face = []
            for tok in args:
                f = tok.split("/")
                v = _fix_index(f[0], len(pos) // 3)
                vt = _fix_index(f[1], len(tex) // 2) if len(f) > 1 and f[1] else -1
                vn = _fix_index(f[2], len(nor) // 3) if len(f) > 2 and f[2] else -1
                face.append((v, vt, vn))
            if len(face) >= 3:
                cur_faces.append(face)
        elif key in ("o", "g"):
            flush()
            cur_name = " ".join(args) if args else "unnamed_object"
        elif key == "usemtl":
            name = " ".join(args)
            if cur_faces and name != cur_mtl:
                flush()
            cur_mtl = name
        elif key == "mtllib":
            mtllibs.append(" ".join(args))
    flush()
    return meshes, mtllibs, mesh_mtl


def decode_image_rgba8(path: str) -> np.ndarray:
    from PIL import Image

    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))


def load_model_parts(res_dir: str, file_name: str) -> list:
    """EXTENSION: every mesh of the file with ITS material (mesh.material, resources.rs:257), as a list
    of model dicts — what a renderer that consumed all of Model{meshes, materials} would draw."""
    with open(os.path.join(res_dir, file_name), "r") as fh:
        meshes, mtllibs, mesh_mtl = parse_obj(fh.read())
    materials: list[RefMaterial] = []
    for lib in mtllibs:
        with open(os.path.join(res_dir, lib), "r") as fh:
            materials.extend(parse_mtl(fh.read()))
    parts = []
    for m, mtl_name in zip(meshes, mesh_mtl):
        n_verts = len(m.positions) // 3
        if len(m.texcoords) < 2 * n_verts:
            raise IndexError("mesh has no texture coordinates (resources.rs:226 index panic)")
        mid = 0
        for k, mat in enumerate(materials):
            if mat.name == mtl_name:
                mid = k
        mat0 = materials[mid]
        verts = np.zeros(n_verts, dtype=VERTEX_DTYPE)
        verts["position"] = np.asarray(m.positions, dtype=np.float32).reshape(-1, 3)
        verts["tex_coords"] = np.asarray(m.texcoords, dtype=np.float32).reshape(-1, 2)
        faces = np.zeros(len(m.indices) // 3, dtype=FACE_DTYPE)
        faces["indices"] = np.asarray(m.indices, dtype=np.uint32).reshape(-1, 3)
        material = np.zeros(1, dtype=MATERIAL_DTYPE)
        material["ambient"], material["diffuse"], material["specular"] = mat0.ambient, mat0.diffuse, mat0.specular
        parts.append({"vertices": verts, "faces": faces, "material": material,
                      "texture": decode_image_rgba8(os.path.join(res_dir, mat0.diffuse_texture))})
    return parts


def load_model_compute(res_dir: str, file_name: str) -> dict:
    """resources.rs:163-264.  Only meshes[0]/materials[0] are consumed downstream
    (triangle_list.rs:212-245), and that is what is returned."""
    with open(os.path.join(res_dir, file_name), "r") as fh:
        meshes, mtllibs, mesh_mtl = parse_obj(fh.read())
    materials: list[RefMaterial] = []
    for lib in mtllibs:
        with open(os.path.join(res_dir, lib), "r") as fh:
            materials.extend(parse_mtl(fh.read()))
    if not meshes:
        raise ValueError("no meshes in " + file_name)
    if not materials:
        raise ValueError("no materials for " + file_name)
    m = meshes[0]
    n_verts = len(m.positions) // 3
    if len(m.texcoords) < 2 * n_verts:
        raise IndexError("mesh has no texture coordinates (resources.rs:226 index panic)")
    verts = np.zeros(n_verts, dtype=VERTEX_DTYPE)
    verts["position"] = np.asarray(m.positions, dtype=np.float32).reshape(-1, 3)
    verts["tex_coords"] = np.asarray(m.texcoords, dtype=np.float32).reshape(-1, 2)
    faces = np.zeros(len(m.indices) // 3, dtype=FACE_DTYPE)
    faces["indices"] = np.asarray(m.indices, dtype=np.uint32).reshape(-1, 3)
    mat0 = materials[0]
    material = np.zeros(1, dtype=MATERIAL_DTYPE)
    material["ambient"] = mat0.ambient
    material["diffuse"] = mat0.diffuse
    material["specular"] = mat0.specular
    tex = decode_image_rgba8(os.path.join(res_dir, mat0.diffuse_texture))
    # map_Bump (cube.mtl:13): the reference never loads it (resources.rs:187-213); decoded here for the normal-mapped
    # shading EXTENSION only (oracle.render_path with FLAG_NORMAL_MAP).  Linear RGBA8: vectors, not colours.
    nmap = None
    if mat0.normal_texture and os.path.exists(os.path.join(res_dir, mat0.normal_texture)):
        nmap = decode_image_rgba8(os.path.join(res_dir, mat0.normal_texture))
    return {
        "vertices": verts,
        "faces": faces,
        "material": material,
        "texture": tex,  # (H, W, 4) uint8, row 0 = top of the image file
        "material_name": mat0.name,
        "diffuse_texture": mat0.diffuse_texture,
        "normal_texture": mat0.normal_texture,
        "normal_map": nmap,
        "n_meshes": len(meshes),
        "n_materials": len(materials),
    }
