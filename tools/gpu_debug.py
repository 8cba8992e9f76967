"""Step-by-step GPU bring-up with progress prints (each stage flushed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def log(*a):
    print(f"[{time.strftime('%H:%M:%S')}]", *a, flush=True)

log("import")
from ignis_amd import Device, LoadedScene, device
import oracle
log("device_count", device.device_count())
scene = LoadedScene.from_file("scenes/diamond_scene.json", 64, 64)
dev = Device(0, acquire_stats=2)
log("created")
dev.assign_scene(scene)
log("scene assigned")
rays, _ = oracle.generate_rays(scene, 1, 64, 64, 0, 64 * 64, seed=3)
for n in (1, 64, 4096):
    log("traverse", n)
    got = dev.traverse(rays[:n], flags=1)
    ref = oracle.trace(scene, rays[:n], flags=1)
    log("  ids equal:", np.array_equal(got["ent_id"], ref["ent_id"]), np.array_equal(got["prim_id"], ref["prim_id"]),
        "t bits equal:", np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32)), "ms", got["kernel_ms"])
log("stats", dev.stats())
log("render 64x64 spi 1")
dev.render(1, 64, 64, seed=3)
fb = dev.framebuffer()
ref, st = oracle.render(scene, 1, 64, 64, seed=3)
log("  rel L2", float(np.linalg.norm(fb - ref) / np.linalg.norm(ref)), dev.stats(), st)
