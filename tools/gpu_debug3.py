import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from ignis_amd import Device, LoadedScene
scene = LoadedScene.from_file("scenes/many_point_lights_hip.json", 64, 64)
dev = Device(0, acquire_stats=2)
dev.assign_scene(scene)
rng = np.random.default_rng(7)
n = 1 << 17
org = rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([org, d, np.full((n, 1), 1e-3, np.float32), np.full((n, 1), 3.4e38, np.float32)], axis=1).astype(np.float32)
a = dev.traverse(rays, flags=4); b = oracle.trace(scene, rays, flags=4)
for k in ("ent_id","prim_id"): print(k, np.array_equal(a[k], b[k]), int((a[k]!=b[k]).sum()))
for k in ("t","u","v"): print(k, np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), int((a[k].view(np.uint32)!=b[k].view(np.uint32)).sum()))
st = dev.stats(); print({k: (st[k], b["stats"][k]) for k in ("nodes","tris","leaves")}, "max_stack", b["stats"]["max_stack"])
bad = np.nonzero(a["prim_id"] != b["prim_id"])[0][:5]
for i in bad: print(i, rays[i], a["ent_id"][i], a["prim_id"][i], a["t"][i], "|", b["ent_id"][i], b["prim_id"][i], b["t"][i])
