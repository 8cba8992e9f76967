"""Closest-hit / any-hit traversal time against the launch size (igd_traverse on random rays inside the diamond scene's box, HIP-event
kernel time): what a launch costs beyond its rays. usage: python tools/traverse_sizes.py [max_log2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ignis_amd import Device, LoadedScene

W, H = 1920, 1080
scene = LoadedScene.from_file(os.path.join(os.path.dirname(__file__), "..", "scenes", "diamond_scene.json"), W, H)
dev = Device(0)
dev.assign_scene(scene)
top = int(sys.argv[1]) if len(sys.argv) > 1 else 25
rng = np.random.default_rng(3)
n = 1 << top
org = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([org, d, np.full((n, 1), 1e-3, np.float32), np.full((n, 1), 3.4e38, np.float32)], axis=1).astype(np.float32)
for any_hit in (False, True):
    pts = []
    for lg in range(10, top + 1):
        m = 1 << lg
        r = dev.traverse(rays[:m], flags=8 if any_hit else 1, any_hit=any_hit, repeat=8)
        us = r["kernel_ms"] * 1e3
        pts.append((m, us))
        print(f"{'any-hit' if any_hit else 'closest'} n=2^{lg:2d} {us:9.1f} us  {m / us:9.1f} Mrays/s", flush=True)
    (m1, t1), (m2, t2) = pts[-1], pts[-3]
    b = (t1 - t2) / (m1 - m2)
    print(f"# slope {1 / b:.0f} Mrays/s, intercept {t1 - b * m1:.0f} us")
