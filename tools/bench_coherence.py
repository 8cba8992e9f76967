"""How much does ray order matter for the closest-hit kernel? Same incoherent ray set (origins uniform in the room,
directions uniform on the sphere — second-bounce-like), traversed in random order, grouped by direction octant, and
sorted by (octant, Morton code of the origin). igd_traverse, HIP-event kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ignis_amd import Device, LoadedScene

scene = LoadedScene.from_file("scenes/diamond_scene.json", 1920, 1080)
dev = Device(0)
dev.assign_scene(scene)
rng = np.random.default_rng(3)
n = 1 << 23
org = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([org, d, np.full((n, 1), 1e-3, np.float32), np.full((n, 1), 3.4e38, np.float32)], axis=1).astype(np.float32)
octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)


def morton(p, bits):
    q = np.clip(((p + 1) * 0.5 * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(p), np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


orders = {
    "random": np.arange(n),
    "octant, windows of 256": np.concatenate([w[np.argsort(octant[w], kind="stable")] for w in np.arange(n).reshape(-1, 256)]),
    "octant, windows of 4096": np.concatenate([w[np.argsort(octant[w], kind="stable")] for w in np.arange(n).reshape(-1, 4096)]),
    "octant, global": np.argsort(octant, kind="stable"),
    "octant + morton(origin, 4 bits)": np.argsort(octant * (1 << 12) + morton(org, 4), kind="stable"),
    "octant + morton(origin, 7 bits)": np.argsort(octant * (1 << 21) + morton(org, 7), kind="stable"),
}
for name, idx in orders.items():
    r = dev.traverse(rays[idx], flags=4, repeat=5)
    print(f"{name:36s} {r['kernel_ms'] * 1e3:9.1f} us  {n / r['kernel_ms'] / 1e3:9.1f} Mrays/s", flush=True)
