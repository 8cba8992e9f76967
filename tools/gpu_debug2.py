import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from ignis_amd import Device, LoadedScene
w, h, spi = 160, 120, 4
scene = LoadedScene.from_file("scenes/many_point_lights_hip.json", w, h)
dev = Device(0, acquire_stats=2)
dev.assign_scene(scene); dev.resize(w, h)
dev.render(spi, w, h, iteration=0, seed=4)
fb = dev.framebuffer(); st = dev.stats()
ref, s = oracle.render(scene, spi, w, h, iteration=0, seed=4)
print("rel", np.linalg.norm(fb-ref)/np.linalg.norm(ref))
for k in ("camera_rays","bounce_rays","shadow_rays","unoccluded","nodes","tris","leaves"): print(k, st[k], s[k])
d = np.abs(fb-ref).sum(axis=2)
ys, xs = np.nonzero(d > 1e-4)
print("diff pixels", len(ys), list(zip(ys[:10].tolist(), xs[:10].tolist())))
for y, x in list(zip(ys[:5], xs[:5])): print(y, x, fb[y,x], ref[y,x])
# primary hits
rays, _ = oracle.generate_rays(scene, spi, w, h, 0, w*h*spi, seed=4)
a = dev.traverse(rays, flags=1); b = oracle.trace(scene, rays, flags=1)
print("hits equal", np.array_equal(a["ent_id"], b["ent_id"]), np.array_equal(a["prim_id"], b["prim_id"]), np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32)), "max_stack", b["stats"]["max_stack"])
