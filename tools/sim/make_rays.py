"""Generates bounce rays of a wavefront pass for the traversal simulator (tools/sim/sim_bvh.cpp).
Statistics only (numpy RNG, float64 geometry): not a parity tool.
usage: python tools/sim/make_rays.py cfg3|cfg4 out_prefix [max_rows]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc, ref_loader
import __graft_entry__ as g

rwr = g.load_package()
cfg, out = sys.argv[1], sys.argv[2]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 64
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
suz = ref_loader.load_model_compute(rwr.RES_DIR, "suzanne_lowpoly.obj")
if cfg == "cfg3":
    w, h, eye, inst = 1920, 1080, (0, 0, 0), None
else:
    w, h, eye, inst = 3840, 2160, (0, 0, 12), rwr.make_instance_grid(4, 3.0)
cam = orc.camera_build_inv_uniform(orc.make_camera(eye=eye, aspect=w / h))
r0 = (h // 2 - rows // 2) // 8 * 8
r1 = r0 + rows
res = orc.render_path(cam, orc.make_screen(w, h), orc.make_params(1, 0), orc.make_spheres(), suz, rows=(r0, r1),
                      instances=None if inst is None else inst.view(orc.INSTANCE_DTYPE))
obj, t = res["obj_id"][r0:r1], res["hit_t"][r0:r1]
# world-space triangles
V = suz["vertices"]["position"].astype(np.float64)
F = suz["faces"]["indices"]
tri = V[F]  # n,3,3
if inst is not None:
    M = inst["model"].astype(np.float64)  # [i][col][row]
    tris = []
    for m in M:
        R = m[:3, :3].T  # row-major 3x3
        T = m[3, :3]
        tris.append(tri @ R.T + T)
    tri = np.concatenate(tris)
tri32 = tri.astype(np.float32).reshape(-1, 9)
# primary rays (float64 restatement of pixelToRay)
ys, xs = np.mgrid[r0:r1, 0:w]
xn = 2.0 * (xs + 0.5) / w - 1.0
yn = 2.0 * (ys + 0.5) / h - 1.0
P = cam["proj_inv"][0].astype(np.float64)  # [col][row]
VM = cam["viewmodel_inv"][0].astype(np.float64)
v = P[0][None, None, :] * xn[..., None] + P[1][None, None, :] * yn[..., None] + P[2] + P[3]
v[..., 3] = 0
wv = v[..., 0:1] * VM[0] + v[..., 1:2] * VM[1] + v[..., 2:3] * VM[2]
D = wv[..., :3]
D /= np.linalg.norm(D, axis=-1, keepdims=True)
O = np.array(cam["origin"][0], np.float64)
hit = obj >= 0
Pw = O + t[..., None] * D
N = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
N /= np.linalg.norm(N, axis=-1, keepdims=True)
n = N[np.maximum(obj, 0)]
flip = (n * D).sum(-1) > 0
n[flip] *= -1
rng = np.random.default_rng(1)
s = np.where(n[..., 2] >= 0, 1.0, -1.0)
aa = -1.0 / (s + n[..., 2])
bb = n[..., 0] * n[..., 1] * aa
b1 = np.stack([1 + s * n[..., 0] ** 2 * aa, s * bb, -s * n[..., 0]], -1)
b2 = np.stack([bb, s + n[..., 1] ** 2 * aa, -n[..., 1]], -1)
O1 = Pw + 1e-4 * n
allr = []
for k in range(S):
    u1, u2 = rng.random(obj.shape), rng.random(obj.shape)
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    a, b, dz = r * np.cos(phi), r * np.sin(phi), np.sqrt(1 - u1)
    D1 = a[..., None] * b1 + b[..., None] * b2 + dz[..., None] * n
    D1 /= np.linalg.norm(D1, axis=-1, keepdims=True)
    allr.append(np.concatenate([O1, D1], -1).astype(np.float32))
rays = allr[0]
np.save(out + "_rays.npy", rays)
np.save(out + "_hit.npy", hit)
tri32.tofile(out + "_tris.bin")
obj.astype(np.int32).tofile(out + "_obj.bin")   # the face every pixel's bounce ray starts on (policy F of the simulator)
# raw dump for the C++ side: int32 rows, w, then hit mask (u8), rays (f32)
with open(out + "_rays.bin", "wb") as fh:
    np.array([rays.shape[0], rays.shape[1], tri32.shape[0], S], np.int32).tofile(fh)
    hit.astype(np.uint8).tofile(fh)
    for r_ in allr:
        r_.tofile(fh)
print(cfg, "rows", r0, r1, "hit fraction", hit.mean(), "tris", tri32.shape[0])
