"""What can ANY ordering of a bounce stream buy the closest-hit kernel on a scene whose BVH outgrows the L2s?

    python tools/bench_ray_order.py SCENE.json [width height spp]

Builds a second-bounce-like ray set on the scene itself — camera rays through the film, their hits (igd_traverse), one ray
from every hit point into a lobe around the direction it came from — and times igd_traverse (HIP-event kernel time) over the
SAME rays in different orders:
  * stream order (pixel-major: what a wavefront's streams carry, k_shade appends window after window),
  * the same with 256-ray windows grouped by direction octant (what k_shade does today),
  * a random permutation (no coherence at all: the floor),
  * globally by (octant, Morton code of the origin)   — raysort.hip's default key,
  * globally by (direction cell on the octahedral map, Morton code of the origin), several widths,
  * camera rays themselves (the coherent ceiling: neighbours in the stream are neighbours on the film).
Prints Mrays/s per order. Measurement tool, not part of the product."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ignis_amd import Device, LoadedScene

path = sys.argv[1]
W, H, SPP = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 4)
scene = LoadedScene.from_file(path, W, H)
t = scene.tables.contents if hasattr(scene.tables, "contents") else scene.tables
cam = t.camera
eye = np.array(list(cam.eye), np.float32)
fwd = np.array(list(cam.dir), np.float32)
up = np.array(list(cam.up), np.float32)
right = np.cross(fwd, up)
right /= np.linalg.norm(right)
bmin, bmax = np.array(list(t.bbox_min), np.float32), np.array(list(t.bbox_max), np.float32)
fov = float(cam.fov)
sx = np.tan(fov / 2)
sy = sx * H / W
rng = np.random.default_rng(5)
n = W * H * SPP
pix = np.repeat(np.arange(W * H), SPP)
px = (pix % W + rng.random(n)) / W * 2 - 1
py = 1 - (pix // W + rng.random(n)) / H * 2
d = fwd[None, :] + (px * sx)[:, None] * right[None, :] + (py * sy)[:, None] * up[None, :]
d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


def ray_list(org, dirs, tmin=1e-3):
    m = len(dirs)
    return np.concatenate([np.broadcast_to(org, (m, 3)) if org.ndim == 1 else org, dirs, np.full((m, 1), tmin, np.float32), np.full((m, 1), 3.4e38, np.float32)], axis=1).astype(np.float32)


dev = Device(0, acquire_stats=2)
dev.assign_scene(scene)
cam_rays = ray_list(eye, d)
r0 = dev.traverse(cam_rays, flags=1, repeat=3)
print(f"{'camera rays (the coherent ceiling)':58s} {n / r0['kernel_ms'] / 1e3:9.1f} Mrays/s   hit rate {np.mean(r0['prim_id'] >= 0):.3f}", flush=True)
hit = r0["prim_id"] >= 0
P = (eye[None, :] + d * r0["t"][:, None])[hit]
back = -d[hit]
m = len(P)
rnd = rng.normal(size=(m, 3)).astype(np.float32)
rnd /= np.linalg.norm(rnd, axis=1, keepdims=True)
bd = back + 0.95 * rnd  # a lobe around the way back (the surface normal is not known here; a ray into the surface just ends early)
bd = (bd / np.linalg.norm(bd, axis=1, keepdims=True)).astype(np.float32)
borg = (P + 1e-3 * back).astype(np.float32)
rays = ray_list(borg, bd)


def morton3(p, bits):
    q = np.clip(((p - bmin) / np.maximum(bmax - bmin, 1e-20) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(p), np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


def octa(v, bits):
    s = 1.0 / np.abs(v).sum(axis=1)
    x, y = v[:, 0] * s, v[:, 1] * s
    neg = v[:, 2] < 0
    ox = (1 - np.abs(y)) * np.where(x < 0, -1, 1)
    oy = (1 - np.abs(x)) * np.where(y < 0, -1, 1)
    x, y = np.where(neg, ox, x), np.where(neg, oy, y)
    qu = np.clip(((x + 1) * 0.5 * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    qv = np.clip(((y + 1) * 0.5 * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(v), np.int64)
    for b in range(bits):
        code |= ((qu >> b) & 1) << (2 * b)
        code |= ((qv >> b) & 1) << (2 * b + 1)
    return code


octant = (bd[:, 0] < 0).astype(np.int64) | ((bd[:, 1] < 0).astype(np.int64) << 1) | ((bd[:, 2] < 0).astype(np.int64) << 2)
mw = m // 256 * 256
win = np.concatenate([np.concatenate([w[np.argsort(octant[w], kind="stable")] for w in np.arange(mw).reshape(-1, 256)]), np.arange(mw, m)])
orders = {
    "bounce rays, stream (pixel) order": np.arange(m),
    "  256-ray windows grouped by octant (k_shade today)": win,
    "  random permutation (the floor)": rng.permutation(m),
    "  (octant, morton(origin, 7 bits))": np.argsort((octant << 21) | morton3(borg, 7), kind="stable"),
    "  (morton(origin, 7 bits), octant)": np.argsort((morton3(borg, 7) << 3) | octant, kind="stable"),
    "  (dir 4+4 bits, morton(origin, 6))": np.argsort((octa(bd, 4) << 18) | morton3(borg, 6), kind="stable"),
    "  (dir 6+6 bits, morton(origin, 6))": np.argsort((octa(bd, 6) << 18) | morton3(borg, 6), kind="stable"),
    "  (morton(origin, 5), dir 4+4 bits)": np.argsort((morton3(borg, 5) << 8) | octa(bd, 4), kind="stable"),
    "  (morton(origin, 3), dir 5+5 bits, morton low 4)": np.argsort((((morton3(borg, 7) >> 12) << 10 | octa(bd, 5)) << 12) | (morton3(borg, 7) & 0xFFF), kind="stable"),
}
for name, idx in orders.items():
    dev.reset_stats()
    r = dev.traverse(rays[idx], flags=4, repeat=3)
    st = dev.stats()
    lanes = sum(st["section_lanes"][:3]) / max(1, 64 * sum(st["section_passes"][:3]))
    print(f"{name:58s} {m / r['kernel_ms'] / 1e3:9.1f} Mrays/s   {r['kernel_ms']:8.3f} ms   nodes/ray/run {st['nodes_primary'] / m:.2f}   section lane use {lanes:.3f}", flush=True)
