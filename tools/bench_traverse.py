"""Micro-benchmark of the traversal kernels on ray lists (igd_traverse, HIP-event kernel time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from ignis_amd import Device, LoadedScene

W, H = 1920, 1080
scene = LoadedScene.from_file("scenes/diamond_scene.json", W, H)
dev = Device(0)
dev.assign_scene(scene)
cam, _ = oracle.generate_rays(scene, 1, W, H, 0, W * H, seed=1)
rng = np.random.default_rng(3)
n = W * H
org = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
inc = np.concatenate([org, d, np.full((n, 1), 1e-3, np.float32), np.full((n, 1), 3.4e38, np.float32)], axis=1).astype(np.float32)
for name, rays in (("camera", cam), ("incoherent", inc)):
    for m in (1, 64, 4096, 65536, n):
        r = dev.traverse(rays[:m], flags=1, repeat=10)
        print(f"{name:10s} n={m:8d} kernel {r['kernel_ms']*1e3:9.1f} us  {m / r['kernel_ms'] / 1e3:9.1f} Mrays/s", flush=True)
    r = dev.traverse(rays, flags=8, any_hit=True, repeat=10)
    print(f"{name:10s} any-hit n={n} kernel {r['kernel_ms']*1e3:9.1f} us  {n / r['kernel_ms'] / 1e3:9.1f} Mrays/s", flush=True)
